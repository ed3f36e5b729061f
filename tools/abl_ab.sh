#!/bin/bash
# GPU box: ablation variants of the crossing path (WRONG results by design; timing only): what each part costs a wave per record
for v in ${VARS:-tree abl_nodiag abl_noedge abl_noboth}; do
  L=""; [ "$v" != "tree" ] && L=$PWD/build_ab/libsitrk_$v.so
  for cfg in "--config c2 --buoys 50000" "--config c2" "--config c3"; do
    SITRK_LIB_PATH=$L python3 bench.py $cfg --steps 640 --warmup 64 --no-cpu-baseline --only-fused --no-c2 > /tmp/x.json 2>/tmp/x.err
    python3 - /tmp/x.json $v "$cfg" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s %-26s %.3e p-steps/s  %.3f us/record  alive %s" % (sys.argv[2], sys.argv[3], d["value"], 1e3*d["ms_per_step"], d["config"]["alive_after"]))
PY
  done
done
