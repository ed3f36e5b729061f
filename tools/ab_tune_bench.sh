#!/bin/bash
# Runs on the GPU box: bench.py once per --tune setting (product library), two rounds.   tools/ab_tune_bench.sh <tag> "<tune1>;<tune2>;..." [bench args]
TAG=$1; IFS=';' read -ra SETS <<< "$2"; shift 2
ARGS=${@:---steps 640 --warmup 64 --no-cpu-baseline --no-c2}
mkdir -p gpurun_out
for r in 1 2; do
  n=0
  for t in "${SETS[@]}"; do
    n=$((n+1))
    python bench.py $ARGS ${t:+--tune $t} > gpurun_out/${TAG}_${n}_${r}.json 2> gpurun_out/${TAG}_${n}_${r}.err || { echo "[$t] failed"; tail -3 gpurun_out/${TAG}_${n}_${r}.err; }
    python - "$t" gpurun_out/${TAG}_${n}_${r}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print("%-34s fused %.4e  (%.3f ms/launch)  per-record %.4e" % (sys.argv[1] or "(defaults)", d["value"], d["roofline"]["avg_launch_ms"], d.get("per_record_launch", {}).get("value", 0)))
except Exception as e:
    print(sys.argv[1], "no result", e)
PY
  done
done
