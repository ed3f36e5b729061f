import sys, os, resource, time
sys.path.insert(0,'/root/repo')
import numpy as np
from sitrack_amd import ncio
lvl=sys.argv[1]; os.environ['SITRK_NC_COMPLEVEL']=lvl
Nb=1_100_000; Nt=int(sys.argv[2])
rng=np.random.default_rng(1)
t=np.arange(Nt)*3600+850608000
ids=np.arange(Nb)+300534062025510
st=ncio.CloudBuoysStream('/tmp/rss_%s.nc'%lvl,t,ids,corigin='X')
base=rng.normal(size=Nb)*1000
t0=time.time()
for k in range(Nt):
    y=base+k*0.36; st.put(k,y,y+1,y*0.01+70,y*0.02,np.ones(Nb,'i1'))
    if k%20==0: print(k, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss//1024,'MB', flush=True)
st.close()
print('done', time.time()-t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss//1024,'MB', os.path.getsize('/tmp/rss_%s.nc'%lvl)//(1<<20),'MB file')
