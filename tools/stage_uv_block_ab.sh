#!/bin/bash
# (needs profiles/r04ba_stage_uv_kernel.patch applied: the knob stage_uv is not in the shipped library -- DESIGN 3.2 item 39)
# GPU box: the staged form with 64- / 128-thread workgroups (one / two waves behind each per-record barrier).  tools/stage_uv_block_ab.sh <tag>
TAG=${1:-r04ax}; OUT=gpurun_out/${TAG}_stage_uv_block.txt; : > $OUT
for r in 1 2; do for v in b64 b128; do for t in "stage_uv=0" "stage_uv=1" "stage_uv=0,patch_kb=8" "stage_uv=1,patch_kb=8" "stage_uv=1,patch_kb=4"; do
  SITRK_LIB_PATH=$PWD/build_ab/libsitrk_$v.so python3 bench.py --config c2 --steps 1000 --warmup 50 --no-cpu-baseline --no-c2 --only-fused --check --tune $t 2>> $OUT.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('c2', '$v', '$t', 'round', $r, '%.4e' % d['value'])" >> $OUT
done; done; done
sort $OUT; grep -c "check OK" $OUT.err
