import json, os, subprocess, sys, tempfile
root = os.getcwd()
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_driver as td
tmp = tempfile.mkdtemp()
c = td.make_case(tmp, nrec=int(os.environ.get('NREC','200')), nP=1_150_000)
launcher = ("import json, resource, sys; sys.path.insert(0, %r); from sitrack_amd import driver as drv; "
            "out = drv.main(sys.argv[1:]); "
            "print('RESULT ' + json.dumps({'nP': int(out['nP']), 'hwm': [l for l in open('/proc/self/status') if l.startswith(('VmHWM','VmRSS'))]}))" % root)
for lvl, arenas in (("1", None), ("9", None)):
    env = dict(os.environ, SITRK_NC_COMPLEVEL=lvl)
    if arenas: env["MALLOC_ARENA_MAX"] = arenas
    r = subprocess.run([sys.executable, "-c", launcher, "-i", c["si3"], "-m", c["mm"], "-s", c["seed"], "-N", "TEST4", "-F"], cwd=tmp, env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print("level", lvl, "arenas", arenas, line[-1][:400] if line else r.stderr[-800:], flush=True)
