# GPU box: Survive rows-per-wave A/B (variants built by tools/build_variant.sh <name> -D...; the -DSITRK_SV_ROWS_BATCH experiment of round 4 was not kept: see DESIGN 3 kernel table)
for r in 1 2 3 4; do for v in ${SV_VARS:-r16 r28 r44}; do
SITRK_LIB_PATH=$PWD/build_ab/libsitrk_$v.so python3 tools/sv_box_bench.py 2>/dev/null | grep box_batch16 | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$v', $r, d['us_per_record'])"
done; done
