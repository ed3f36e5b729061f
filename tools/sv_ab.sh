#!/bin/bash
# GPU box: time of the Survive kernel per resident 4096^2 record for variant builds (tools/build_variant.sh svR_D -DSITRK_SV_R=.. -DSITRK_SV_D=..)
export TMPDIR=/tmp
for v in "$@"; do
  OUT=$PWD/gpurun_out/svab/$v; mkdir -p $OUT
  L=""; [ "$v" != "tree" ] && L=$PWD/build_ab/libsitrk_$v.so
  SITRK_LIB_PATH=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 32 --warmup 32 --only-fused --no-cpu-baseline --no-c2 > $OUT/kt.log 2>&1
  python3 - $OUT $v <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "survive" in r["Name"]: print("%-10s %s calls avg %.2f us min %.2f max %.2f" % (sys.argv[2], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
