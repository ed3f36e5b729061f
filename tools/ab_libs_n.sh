#!/bin/bash
# GPU box: like tools/ab_libs.sh with N interleaved rounds and a mean per variant.   tools/ab_libs_n.sh <tag> <rounds> "<variants>" [bench args...]
TAG=$1; N=$2; VARS=$3; shift 3
ARGS=${@:---steps 2048 --warmup 64 --no-cpu-baseline --no-c2 --only-fused}
mkdir -p gpurun_out
: > gpurun_out/${TAG}_ab.txt
for r in $(seq 1 $N); do
  for v in $VARS; do
    SITRK_LIB_PATH=$PWD/build_ab/libsitrk_$v.so python3 bench.py $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', $r, d['value'], d['roofline']['avg_launch_ms'])" >> gpurun_out/${TAG}_ab.txt
  done
done
python3 - gpurun_out/${TAG}_ab.txt <<'PY'
import sys, collections
v=collections.defaultdict(list)
for l in open(sys.argv[1]):
    a=l.split(); v[a[0]].append(float(a[2]))
base=sum(v[next(iter(v))])/len(v[next(iter(v))])
for k,x in v.items():
    m=sum(x)/len(x); print("%-12s mean %.4e  (%+.2f %% vs %s)  min %.4e max %.4e  n=%d" % (k, m, 100*(m/base-1), next(iter(v)), min(x), max(x), len(x)))
PY
