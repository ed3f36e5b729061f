#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory into profiles/<tag>_<config>_summary.{md,json}.

    python tools/summarize_prof.py gpurun_out/prof_r01 r01 [config]

Kernel times come from `rocprofv3 --kernel-trace --stats`; HBM traffic from the separate
--pmc passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE
and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced
streaming read (x2), WRITE_SIZE is exact.  The x2 is checked in-run on build_geo_kernel, whose
byte count is known (reads 6 fp64 arrays = 48 B/cell, writes 48 B/cell).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "")
    if "rocprim" in n:
        return "rocprim::radix_sort(" + ("onesweep_iteration" if "onesweep_iteration" in n else "histogram") + ")"
    return n.split("(")[0]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    config = sys.argv[3] if len(sys.argv) > 3 else "c3"
    rec_per_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 32     # records every profiled fused dispatch advanced
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_md = os.path.join(root, "profiles", "%s_%s_summary.md" % (tag, config))
    stats = {}
    for f in glob.glob(os.path.join(src, "kt", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
            if k in stats:
                stats[k]["calls"] += calls
                stats[k]["total_ns"] += tot
            else:
                stats[k] = {"calls": calls, "total_ns": tot, "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    counters = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> [values per dispatch]
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            counters[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["# rocprofv3 summary `%s` (%s)" % (tag, config), "",
             "Command: `tools/profile_gpu.sh %s` = `rocprofv3 --kernel-trace --stats -- python3 bench.py ...` plus one" % tag,
             "`rocprofv3 --pmc <group>` pass per counter group (no trace domains in the PMC passes).", "",
             "## Kernel times (`--kernel-trace --stats`)", "",
             "| kernel | calls | avg us | min us | max us | total ms |", "|---|---|---|---|---|---|"]
    for k, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"]):
        lines.append("| `%s` | %d | %.1f | %.1f | %.1f | %.3f |" % (k, s["calls"], s["total_ns"] / s["calls"] / 1e3,
                                                                   s["min_ns"] / 1e3, s["max_ns"] / 1e3, s["total_ns"] / 1e6))
    lines += ["", "## PMC counters, average per dispatch", "", "| kernel | counter | dispatches | mean |", "|---|---|---|---|"]
    avg = {}
    for k in sorted(counters):
        if k.startswith("__amd_rocclr"):
            continue                      # runtime fill/copy helpers
        for c in sorted(counters[k]):
            v = counters[k][c]
            avg.setdefault(k, {})[c] = sum(v) / len(v)
            if not ("advect_" in k or "build_geo" in k or "survive_mask" in k):
                continue                  # keep the table to the kernels that matter; all averages stay in `avg`
            lines.append("| `%s` | %s | %d | %.6g |" % (k, c, len(v), avg[k][c]))
    res = {"tag": tag, "config": config}
    adv = next((k for k in avg if "advect_run_kernel" in k), None) or next((k for k in avg if "advect_step_kernel" in k), None)
    fused = adv is not None and "advect_run_kernel" in adv
    lines += ["", "## HBM traffic of the dominant kernel", ""]
    if adv and "FETCH_SIZE" in avg[adv] and "WRITE_SIZE" in avg[adv]:
        a = avg[adv]
        rd = a["FETCH_SIZE"] * 1024 * 2
        wr = a["WRITE_SIZE"] * 1024
        res.update({"kernel": adv, "fetch_size_kib": a["FETCH_SIZE"], "write_size_kib": a["WRITE_SIZE"],
                    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "avg_kernel_us": stats[adv]["total_ns"] / stats[adv]["calls"] / 1e3 if adv in stats else None})
        lines += ["`%s`: FETCH_SIZE %.0f KiB x 1024 x 2 (gfx950 correction) = %.1f MB read; WRITE_SIZE %.0f KiB x 1024 = %.1f MB written;"
                  % (adv, a["FETCH_SIZE"], rd / 1e6, a["WRITE_SIZE"], wr / 1e6),
                  "**%.1f MB per launch**." % ((rd + wr) / 1e6)]
        if res["avg_kernel_us"]:
            lines.append("At %.1f us per launch that is %.0f GB/s of fabric-side traffic (Infinity-Cache hits are counted, not excluded)."
                         % (res["avg_kernel_us"], (rd + wr) / res["avg_kernel_us"] / 1e3))
        for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
            if c in a:
                res[c] = a[c]
        if "SQ_ACTIVE_INST_VALU" in a and "GRBM_GUI_ACTIVE" in a and res.get("avg_kernel_us"):
            # SQ_* count quad-cycles summed over waves; 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0
            res["sclk_ghz"] = cyc / (res["avg_kernel_us"] * 1e3)
            res["valu_busy_frac"] = a["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
            if "SQ_INSTS_VALU" in a and "SQ_WAVES" in a:
                res["valu_insts_per_wave"] = a["SQ_INSTS_VALU"] / a["SQ_WAVES"]
                if "SQ_INSTS_SALU" in a:
                    res["salu_insts_per_wave"] = a["SQ_INSTS_SALU"] / a["SQ_WAVES"]
                # instruction classes (their own PMC pass): 64-bit classes occupy the SIMD 4 cycles per wave64 instruction, 32-bit ones 2
                c64 = [a.get("SQ_INSTS_VALU_%s" % k) for k in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT64")]
                if all(v is not None for v in c64):
                    res["valu_class_counts_per_wave"] = {k: a.get("SQ_INSTS_VALU_%s" % k, 0.0) / a["SQ_WAVES"]
                                                         for k in ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT64", "CVT", "INT32", "ADD_F32")}
                    res["valu64_frac_counters"] = sum(c64) / a["SQ_INSTS_VALU"]
                    lines.append("Instruction classes: ADD/MUL/FMA/TRANS_F64 + INT64 = %.1f %% of SQ_INSTS_VALU (CVT %.1f %%, INT32 %.1f %%)."
                                 % (100 * res["valu64_frac_counters"], 100 * a.get("SQ_INSTS_VALU_CVT", 0) / a["SQ_INSTS_VALU"],
                                    100 * a.get("SQ_INSTS_VALU_INT32", 0) / a["SQ_INSTS_VALU"]))
                if fused:
                    res["records_per_launch"] = rec_per_launch
                    res["valu_per_wave_record"] = res["valu_insts_per_wave"] / rec_per_launch
                    lines.append("SQ_INSTS_VALU / SQ_WAVES / %d records per launch = **%.1f VALU instructions per wave per record**."
                                 % (rec_per_launch, res["valu_per_wave_record"]))
            lines.append("Instruction issue: SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs = %.0f cycles per SIMD of %.0f available "
                         "(GRBM_GUI_ACTIVE / 8) -> VALU busy %.0f %%; clock held %.2f GHz."
                         % (a["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0, cyc, 100 * res["valu_busy_frac"], res["sclk_ghz"]))
        if "TCC_HIT_sum" in a and "TCC_MISS_sum" in a:
            lines.append("L2 hit rate TCC_HIT/(HIT+MISS) = %.3f." % (a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"])))
    bg = next((k for k in avg if "build_geo_kernel" in k), None)
    if bg and "FETCH_SIZE" in avg[bg]:
        n = {"c3": 4096 * 4096, "c2": 512 * 512}.get(config)
        if n:
            ratio = (48.0 * n) / (avg[bg]["FETCH_SIZE"] * 1024)
            res["fetch_calibration_build_geo"] = ratio
            msg = ("Calibration on `build_geo_kernel` (known: reads 48 B/cell = %.1f MB): FETCH_SIZE x 1024 = %.1f MB -> true/reported = **%.3f** (the guide's x2)."
                   % (48.0 * n / 1e6, avg[bg]["FETCH_SIZE"] * 1024 / 1e6, ratio))
            if "WRITE_SIZE" in avg[bg]:
                msg += " WRITE_SIZE x 1024 = %.1f MB vs %.1f MB written." % (avg[bg]["WRITE_SIZE"] * 1024 / 1e6, 48.0 * n / 1e6)
            lines += ["", msg]
    os.makedirs(os.path.dirname(out_md), exist_ok=True)
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump(res, open(out_md.replace("_summary.md", "_summary.json"), "w"), indent=1)
    # bench.py reads profiles/traffic.json for roofline.traffic
    tj = os.path.join(root, "profiles", "traffic.json")
    allt = json.load(open(tj)) if os.path.exists(tj) else {}
    if "hbm_bytes_per_launch" in res:
        allt[config + "_fused" if fused else config] = {"hbm_bytes_per_launch": res["hbm_bytes_per_launch"], "source": os.path.basename(out_md),
                                                        "valu_busy_frac": res.get("valu_busy_frac"), "sclk_ghz": res.get("sclk_ghz"),
                                                        "valu_per_wave_record": res.get("valu_per_wave_record"),
                                                        "records_per_launch": res.get("records_per_launch")}
        prev = (json.load(open(tj)) if os.path.exists(tj) else {}).get(config + "_fused" if fused else config, {})
        # the fingerprint of the binary that was profiled (tools/profile_gpu.sh copies libsitrk.isa.json next to the traces):
        # bench.py marks its roofline stale when the library it loaded has another one
        isa_path = os.path.join(src, "libsitrk.isa.json")
        if fused and os.path.exists(isa_path):
            allt[config + "_fused"]["isa"] = json.load(open(isa_path))["kernels"].get("advect_run_kernel<float,1,false>")
        elif os.path.exists(isa_path):
            allt[config]["isa"] = json.load(open(isa_path))["kernels"].get("advect_step_kernel<float,1,false,512>")
        for k in ("valu_per_wave_fixed", "valu_per_wave_per_record", "valu_terms_source", "valu64_frac", "valu64_frac_source",
                  "salu_per_wave_fixed", "salu_per_wave_per_record", "valu_main_path_per_wave_record",
                  "valu_crossing_path_per_wave_record", "valu_paths_source", "crossing_rate", "valu64_frac_counters_lower_bound", "note"):
            if k in prev:                         # fitted by tools/fit_valu_terms.py from several launch lengths: keep
                allt[config + "_fused" if fused else config][k] = prev[k]
        json.dump(allt, open(tj, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
