#!/bin/bash
# GPU box: the end-to-end regime of bench.py in its transport variants, each checked against the oracle.  tools/e2e_ab.sh <tag> [variants...]
TAG=${1:-r04}; shift
OUT=gpurun_out/${TAG}_e2e.txt
: > $OUT
if [ $# -eq 0 ]; then set -- "" "--e2e-rows" "--e2e-full" "--e2e-library" "--e2e-library --tune fill_threads=8" "--e2e-library --e2e-rows" "--e2e-library --e2e-full"; fi
for v in "$@"; do
  echo "== bench.py --regime e2e $v" >> $OUT
  python3 bench.py --regime e2e --steps 200 --warmup 20 --check --no-cpu-baseline $v 2>> $OUT | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'upload_bytes_per_step': d['config']['e2e_upload_bytes_per_step'], 'GBps': d['config']['e2e_upload_bytes_per_step']/d['ms_per_step']/1e6}))" >> $OUT
done
grep -v amdgpu.ids $OUT
