#!/bin/bash
# Runs on the GPU box: separate rocprofv3 --pmc passes of a short bench.py run for each variant library in build_ab/,
# then prints the per-dispatch averages of advect_run_kernel side by side.   tools/pmc_ab.sh <tag> "<variants>"
TAG=$1; VARS=$2
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
         "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM"
         "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"
         "TCC_HIT_sum TCC_MISS_sum"
         "FETCH_SIZE" "GRBM_GUI_ACTIVE")
for v in $VARS; do
  export SITRK_LIB_PATH=$PWD/build_ab/libsitrk_$v.so
  g=0
  for grp in "${GROUPS_[@]}"; do
    g=$((g+1))
    # PMC_ONLY="1 2 10" restricts the passes to those groups
    if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $g "; then continue; fi
    # (a counter group the hardware cannot collect aborts rocprofv3 and leaves the child hanging: bound every pass)
    timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/${v}_g$g -- python3 bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-c2 --only-fused > $OUT/${v}_g$g.log 2>&1 || { echo "pmc pass $v g$g ($grp) failed"; grep -m2 "error code\|Error" $OUT/${v}_g$g.log; }
    echo "$v g$g done"
  done
done
python3 - "$OUT" $VARS <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, vars_ = sys.argv[1], sys.argv[2:]
res = {}
for v in vars_:
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, v + "_g*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "advect_run_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[v] = {k: sum(x) / len(x) for k, x in acc.items()}
names = sorted(set().union(*[set(r) for r in res.values()]))
print("%-40s" % "counter (avg per fused dispatch)" + "".join("%16s" % v for v in vars_))
for n in names:
    print("%-40s" % n + "".join("%16.5g" % res[v].get(n, float("nan")) for v in vars_))
PY
