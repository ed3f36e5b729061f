#!/bin/bash
# A/B timing of alternative builds of libsitrk.so on the GPU box (via gpurun): each build_ab/libsitrk_<tag>.so is
# copied over sitrack_amd/libsitrk.so in the box's scratch copy and bench.py is run with it.
# Usage: tools/ab_libs.sh <out-prefix> <tag> [<tag> ...]   -> gpurun_out/<out-prefix>_<tag>.json
set -o pipefail
PFX=$1; shift
cp sitrack_amd/libsitrk.so /tmp/libsitrk_orig.so
for tag in "$@"; do
  cp build_ab/libsitrk_$tag.so sitrack_amd/libsitrk.so || exit 1
  python3 bench.py --no-cpu-baseline $AB_ARGS > gpurun_out/${PFX}_$tag.json 2> gpurun_out/${PFX}_$tag.err || { tail -5 gpurun_out/${PFX}_$tag.err; exit 1; }
  python3 - "$tag" gpurun_out/${PFX}_$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
p = d.get("per_record_launch", {})
print("%-10s fused %.4e (%.4f ms/record)   per-record %.4e (%.4f ms)" % (sys.argv[1], d["value"], d["ms_per_step"], p.get("value", 0), p.get("ms_per_step", 0)), flush=True)
PY
done
cp /tmp/libsitrk_orig.so sitrack_amd/libsitrk.so
