#!/bin/bash
# GPU box: variant libraries (build_ab/) on the default-length run AND on the round driver's own command (one 20-record launch).
#   tools/ab_libs20.sh <tag> "<variants>"
TAG=$1; VARS=$2
mkdir -p gpurun_out/$TAG
for r in 1 2 3; do
  for v in $VARS; do
    for form in long drv; do
      if [ $form = long ]; then A="--steps 640 --warmup 64 --no-cpu-baseline --no-c2 --only-fused"; else A="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-c2 --only-fused"; fi
      SITRK_LIB_PATH=$PWD/build_ab/libsitrk_$v.so python3 bench.py $A > gpurun_out/$TAG/${v}_${form}_$r.json 2> gpurun_out/$TAG/${v}_${form}_$r.err || { echo "$v failed"; tail -3 gpurun_out/$TAG/${v}_${form}_$r.err; }
      python3 - gpurun_out/$TAG/${v}_${form}_$r.json $v $form $r <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-10s %-4s round %s  %.4e p-steps/s  (%.3f ms/launch, %.1f records/launch)" % (sys.argv[2], sys.argv[3], sys.argv[4], d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["records_per_launch"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
    done
  done
done
