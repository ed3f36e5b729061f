#!/bin/bash
# Runs on the GPU box: time + FETCH_SIZE / WRITE_SIZE / L2 hit of the fused kernel for a list of --tune settings.
#   tools/fetch_ab.sh <tag> "<tune1>;<tune2>;..."
TAG=$1; IFS=';' read -ra SETS <<< "$2"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/fetch_$TAG; mkdir -p $OUT
n=0
for t in "${SETS[@]}"; do
  n=$((n+1))
  python bench.py --steps 640 --warmup 64 --no-cpu-baseline --no-c2 --only-fused ${t:+--tune $t} > $OUT/t$n.json 2> $OUT/t$n.err
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    g=$(echo $grp | tr ' ' '+')
    timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/t${n}_$g -- python3 bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-c2 --only-fused ${t:+--tune $t} > $OUT/t${n}_$g.log 2>&1 || echo "pmc $grp failed for [$t]"
  done
  python3 - "$t" $OUT t$n <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
t, out, n = sys.argv[1:4]
acc = defaultdict(list)
for f in glob.glob(os.path.join(out, n + "_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "advect_run_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
d = json.loads(open(os.path.join(out, n + ".json")).read().strip().splitlines()[-1])
rd, wr = m.get("FETCH_SIZE", float("nan")) * 2048 / 1e9, m.get("WRITE_SIZE", float("nan")) * 1024 / 1e9
hit = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0), 1)
print("%-30s %.4e p-steps/s  %.3f ms/launch   read %.2f GB + write %.2f GB per 32-record launch   L2 hit %.3f"
      % (t or "(defaults)", d["value"], d["roofline"]["avg_launch_ms"], rd, wr, hit))
PY
done
