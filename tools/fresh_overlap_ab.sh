#!/bin/bash
# GPU box: the fresh-records leg with 64 resident records (launches of 32 while the other 32 are committed on the ingest stream)
# against the default (commits on the compute stream), per library variant.   tools/fresh_overlap_ab.sh <tag> "<variants>"
# (the svlow / cuN variants of profiles/r04an_fresh_overlap_ab.txt created the ingest stream with hipStreamCreateWithPriority(lowest) /
#  hipExtStreamCreateWithCUMask(N low bits) in sitrk_create: two-line hooks that were removed again after the measurement)
TAG=$1; VARS=$2
OUT=gpurun_out/${TAG}_fresh_overlap.txt; : > $OUT
for v in $VARS; do for mode in "" "--fresh-overlap"; do
  SITRK_LIB_PATH=$PWD/build_ab/libsitrk_$v.so python3 bench.py --records 64 --steps 1024 --warmup 64 --no-c2 --no-cpu-baseline --no-c4-shard --no-e2e-upload --check $mode 2>> $OUT.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d['fresh_records']
print('$v', '${mode:-sync}', 'value %.4e fresh %.4e' % (d['value'], d['value_fresh_records']), 'records/launch', f.get('records_per_launch'), 'survive_us', round(d['survive_us_per_record'],2))" >> $OUT
done; done
cat $OUT; grep -c "check OK" $OUT.err
