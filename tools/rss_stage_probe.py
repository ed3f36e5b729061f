# where the command line's resident memory comes from (GPU box): VmRSS after each stage of a minimal session
import os, sys
sys.path.insert(0, os.getcwd())
def rss(tag):
    v = [l.split()[1] for l in open('/proc/self/status') if l.startswith(('VmRSS', 'VmHWM'))]
    print("%-40s VmHWM %6d MB  VmRSS %6d MB" % (tag, int(v[0]) // 1024, int(v[1]) // 1024), flush=True)
rss("python")
import numpy as np
rss("numpy")
from sitrack_amd import _lib, ncio, h5lite
rss("sitrack_amd imported")
L = _lib.lib()
rss("libsitrk + HIP runtime loaded")
ctx = _lib.Context(0)
rss("context created (HIP initialised)")
from sitrack_amd import synthetic as syn
g = syn.make_grid(60, 70, dkm=10., warp=1.0)
ctx.set_grid(g["Yf"], g["Xf"], g["Yu"], g["Xu"], g["Yv"], g["Xv"], g["tmask"])
ctx.alloc_records(32, np.float32)
rss("grid + 32 slots")
n = 1_100_000
_, yx = syn.make_buoys(g, n, seed=1, frac=0.8)
found, ji = ctx.find_cells(yx, syn.nearest_t_index(g, yx).astype(np.int32))
ctx.set_buoys(yx[found], ji[found])
rss("1.1e6 buoys set")
u, v, s = syn.make_fields(g, K=2)
ctx.push_record(0, u[0], v[0], s[0]); ctx.step(0, 0)
p = ctx.fetch_record(0, latlon=True)
rss("one step + fetch_record with lat/lon")
h5lite._load(); h5lite._load_hl()
rss("libhdf5 loaded")
