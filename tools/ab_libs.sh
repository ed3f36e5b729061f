#!/bin/bash
# Runs on the GPU box: bench.py once per variant library in build_ab/ (tools/build_variant.sh), interleaved over ROUNDS rounds.
#   tools/ab_libs.sh <tag> "<variant names>" [bench args...]
TAG=$1; VARS=$2; shift 2
ARGS=${@:---steps 640 --warmup 64 --no-cpu-baseline --no-c2}
mkdir -p gpurun_out
for r in 1 2; do
  for v in $VARS; do
    SITRK_LIB_PATH=$PWD/build_ab/libsitrk_$v.so python bench.py $ARGS > gpurun_out/${TAG}_${v}_$r.json 2> gpurun_out/${TAG}_${v}_$r.err || { echo "$v failed"; tail -3 gpurun_out/${TAG}_${v}_$r.err; }
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_${v}_$r.json").read().strip().splitlines()[-1])
    print("%-12s round $r  fused %.4e  (%.3f ms/launch)  per-record %.4e" % ("$v", d["value"], d["roofline"]["avg_launch_ms"], d.get("per_record_launch",{}).get("value",0)))
except Exception as e:
    print("$v", "no result", e)
PY
  done
done
