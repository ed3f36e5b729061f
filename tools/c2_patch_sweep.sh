#!/bin/bash
# GPU box: the small configuration (C2) against the size of the fused kernel's LDS patch and its margin (occupancy is irrelevant there)
OUT=gpurun_out/${1:-c2patch}; mkdir -p $OUT
for t in "patch_kb=16,patch_margin=8" "patch_kb=24,patch_margin=8" "patch_kb=32,patch_margin=10" "patch_kb=48,patch_margin=12" "patch_kb=63,patch_margin=16" "patch_kb=63,patch_margin=10" "patch_kb=32,patch_margin=6" "patch_kb=16,patch_margin=4"; do
  for nb in 100000 50000; do
    python3 bench.py --config c2 --buoys $nb --steps 960 --warmup 64 --no-cpu-baseline --only-fused --tune $t > $OUT/x.json 2> $OUT/x.err
    python3 - $OUT/x.json $nb "$t" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("buoys %7s %-30s %.3e p-steps/s  %.3f us/record" % (sys.argv[2], sys.argv[3], d["value"], 1e3*d["ms_per_step"]))
PY
  done
done
