for r in 1 2; do
for v in tree blk128 blk64 blk512; do
  L=""; [ "$v" != "tree" ] && L=$PWD/build_ab/libsitrk_$v.so
  for nb in 100000 50000; do
    SITRK_LIB_PATH=$L python3 bench.py --config c2 --buoys $nb --steps 960 --warmup 64 --no-cpu-baseline --only-fused > /tmp/x.json 2>/tmp/x.err
    python3 - /tmp/x.json $v $nb <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-8s buoys %7s  %.3e p-steps/s  %.3f us/record" % (sys.argv[2], sys.argv[3], d["value"], 1e3*d["ms_per_step"]))
PY
  done
done
done
