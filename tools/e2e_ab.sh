mkdir -p gpurun_out/r03t
for x in "" "--e2e-full" "--e2e-library" "--e2e-library --e2e-full"; do
  python3 bench.py --regime e2e --steps 200 --warmup 20 --check --no-cpu-baseline --no-c2 $x > gpurun_out/r03t/x.json 2> gpurun_out/r03t/x.err
  python3 - "$x" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03t/x.json").read().strip().splitlines()[-1])
ok="check OK" in open("gpurun_out/r03t/x.err").read()
print("%-28s %.3e p-steps/s  %.3f ms/step  %.0f MB/step  %s" % (sys.argv[1] or "(torch pinned, row bands)", d["value"], d["ms_per_step"], d["config"]["e2e_upload_bytes_per_step"]/1e6, "check OK" if ok else "NO CHECK"))
PY
done
