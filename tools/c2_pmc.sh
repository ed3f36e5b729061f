#!/bin/bash
# GPU box: where one wave's time goes on the small configuration -- SQ wait / issue counters of the fused kernel.  tools/c2_pmc.sh <tag> [bench args]
TAG=${1:-c2pmc}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ARGS=${@:---config c2 --buoys 50000 --steps 256 --warmup 64 --no-cpu-baseline --only-fused --tune lat_max=0}
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU"; do
  name=$(echo $grp | tr ' ' '+' | cut -c1-60)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > "$OUT/pmc_$name.log" 2>&1 || { tail -5 "$OUT/pmc_$name.log"; echo "pass failed: $grp"; }
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS > "$OUT/kt.log" 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob(sys.argv[1]+"/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "advect_run_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-28s n=%3d mean %.6g" % (k, len(acc[k]), sum(acc[k])/len(acc[k])))
for f in glob.glob(sys.argv[1]+"/kt/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "advect_run" in r["Name"]: print("kernel avg us", float(r["AverageNs"])/1e3, "calls", r["Calls"])
PY
