#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/
# PMC passes are collected on their own (no trace domains), one counter group per pass, as
# MI355X_MICROARCH.md (HBM / rocprofv3 PMC slots) prescribes.
set -o pipefail
TAG=${1:-r01}; shift
ARGS=${@:---steps 128 --warmup 32 --no-cpu-baseline --no-c2 --only-fused}      # multiples of the 32 records per launch: every fused dispatch is a full one
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cp sitrack_amd/libsitrk.isa.json "$OUT/" 2>/dev/null || echo "no libsitrk.isa.json (run make -C sitrack_amd/csrc): the profile cannot be tied to the binary"
echo "== kernel trace" 
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS > "$OUT/kt.log" 2>&1 || { tail -20 "$OUT/kt.log"; exit 1; }
tail -1 "$OUT/kt.log"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32"; do
  name=$(echo $grp | tr ' ' '+')
  echo "== pmc $grp"
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py ${PMC_ARGS:---steps 64 --warmup 32} --no-cpu-baseline --no-c2 ${PMC_FUSED:---only-fused} $PMC_EXTRA > "$OUT/pmc_$name.log" 2>&1 || { tail -5 "$OUT/pmc_$name.log"; echo "pmc pass $grp failed (continuing)"; }
done
find "$OUT" -name "*.csv" | head -50
du -sh "$OUT"
