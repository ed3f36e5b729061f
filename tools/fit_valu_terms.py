#!/usr/bin/env python3
"""VALU (and SALU) instructions per wave of the fused kernel as  fixed-per-launch + per-record x records,  fitted from
rocprofv3 counter summaries of launches of DIFFERENT lengths (tools/summarize_prof.py ... <config> <records per launch>), so
that bench.py can price the launch length that actually ran (the driver's `--steps 20` is one launch of 20 records, the
default run launches of 32) instead of reading the per-record figure off one launch length.

    python tools/fit_valu_terms.py c3 profiles/r03x_c3_20_summary.json profiles/r03x_c3_32_summary.json [...]

Writes valu_per_wave_fixed / valu_per_wave_per_record / salu_* / valu64_frac into profiles/traffic.json[<config>_fused].
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = list(sys.argv[1:])
    isa = None
    if "--isa" in argv:                      # tools/isa_valu_classes.py --json: the hot-block share of 64-bit-class instructions
        k = argv.index("--isa")
        isa = json.load(open(argv[k + 1]))
        isa_name = os.path.basename(argv[k + 1])
        del argv[k:k + 2]
    config, files = argv[0], argv[1:]
    pts = [json.load(open(f)) for f in files]
    n = np.array([p["records_per_launch"] for p in pts], dtype=float)
    out = {}
    for key, name in (("valu_insts_per_wave", "valu"), ("salu_insts_per_wave", "salu")):
        if not all(key in p for p in pts):
            continue
        y = np.array([p[key] for p in pts], dtype=float)
        A = np.stack([np.ones_like(n), n], axis=1)
        (fixed, per), *_ = np.linalg.lstsq(A, y, rcond=None)
        out[name + "_per_wave_fixed"], out[name + "_per_wave_per_record"] = float(fixed), float(per)
        print("%s per wave = %.1f + %.2f x records   (points: %s)" % (name.upper(), fixed, per, ", ".join("%d: %.1f" % (a, b) for a, b in zip(n, y))))
    w = [p["valu64_frac_counters"] for p in pts if "valu64_frac_counters" in p]
    if w:
        # the per-class counters see fp64 arithmetic and 64-bit integer ops only -- not fp64 compares, conversions, 64-bit moves:
        # a LOWER bound of the share of instructions that hold the SIMD 4 cycles
        out["valu64_frac_counters_lower_bound"] = float(np.mean(w))
        out["valu64_frac"] = float(np.mean(w))
        out["valu64_frac_source"] = "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 + _INT64 over SQ_INSTS_VALU, " + ", ".join(os.path.basename(f) for f in files)
    if isa is not None:
        out["valu64_frac"] = float(isa["hot_blocks"]["valu64_frac"])
        out["valu64_frac_source"] = ("ISA listing, blocks a wave executes per record (tools/isa_valu_classes.py -> %s): %.0f of %.0f VALU "
                                     "instructions are of the 64-bit classes (fp64 arithmetic, compares, conversions, 64-bit integer/moves)"
                                     % (isa_name, isa["hot_blocks"]["valu64"], isa["hot_blocks"]["valu"]))
    out["valu_terms_source"] = ", ".join(os.path.basename(f) for f in files)
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    allt = json.load(open(tj))
    allt.setdefault(config + "_fused", {}).update(out)
    json.dump(allt, open(tj, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
