#!/bin/bash
# Runs on the GPU box: how the small configuration (C2, 512x512) scales with the number of buoys and with the knobs that move
# work between global memory and LDS -- time per record ~ flat in the buoy count means one wave's dependent chain sets the pace.
OUT=gpurun_out/${1:-c2probe}; mkdir -p $OUT
for nb in 25000 50000 100000 200000 400000 800000; do
  python3 bench.py --config c2 --buoys $nb --steps 960 --warmup 64 --no-cpu-baseline --only-fused > $OUT/nb_$nb.json 2> $OUT/nb_$nb.err
  python3 - $OUT/nb_$nb.json $nb <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("buoys %8s  %.3e p-steps/s  %.3f us/record  launch %.1f us" % (sys.argv[2], d["value"], 1e3*d["ms_per_step"], 1e3*r["avg_launch_ms"]))
PY
done
for t in patch_kb=0 xcd_group=0 fuse8 fuse16; do
  case $t in fuse*) X="--fuse ${t#fuse}";; *) X="--tune $t";; esac
  python3 bench.py --config c2 --steps 960 --warmup 64 --no-cpu-baseline --only-fused $X > $OUT/t_$t.json 2> $OUT/t_$t.err
  python3 - $OUT/t_$t.json $t <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("tune %-12s  %.3e p-steps/s  %.3f us/record" % (sys.argv[2], d["value"], 1e3*d["ms_per_step"]))
PY
done
