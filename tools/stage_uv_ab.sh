#!/bin/bash
# (needs profiles/r04ba_stage_uv_kernel.patch applied: the knob stage_uv is not in the shipped library -- DESIGN 3.2 item 39)
# GPU box: the staged small-set form of the fused loop (knob stage_uv) against the plain one.   tools/stage_uv_ab.sh <tag>
TAG=${1:-r04aw}; OUT=gpurun_out/${TAG}_stage_uv.txt; : > $OUT
for r in 1 2 3; do for s in 0 1; do
  python3 bench.py --config c2 --steps 1000 --warmup 50 --no-cpu-baseline --no-c2 --only-fused --check --tune stage_uv=$s 2>> $OUT.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('c2 stage_uv', $s, 'round', $r, '%.4e' % d['value'])" >> $OUT
done; done
for nb in 25000 400000 1000000; do for s in 0 1; do
  python3 bench.py --config c2 --buoys $nb --steps 1000 --warmup 50 --no-cpu-baseline --no-c2 --only-fused --check --tune stage_uv=$s 2>> $OUT.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('512x512 buoys', $nb, 'stage_uv', $s, '%.4e' % d['value'])" >> $OUT
done; done
for s in 0 1; do
  python3 bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-c2 --only-fused --check --tune stage_uv=$s 2>> $OUT.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('c3 stage_uv', $s, '%.4e' % d['value'])" >> $OUT
done
cat $OUT; grep -c "check OK" $OUT.err
