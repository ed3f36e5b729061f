#!/usr/bin/env python3
"""Static class count of the VALU instructions of a kernel's main loop in the ISA listing (`make -C sitrack_amd/csrc asm`).

    python tools/isa_valu_classes.py sitrack_amd/csrc/sitrk.s advect_run_kernelIfLi1ELb0 [--json out.json]

On gfx950 a wave64 VALU instruction of the 64-bit classes (fp64 arithmetic, compares and conversions, 64-bit integer ops)
occupies its SIMD for 4 cycles, a 32-bit one for 2 (MI355X_MICROARCH.md: peak vector fp64 = 1/2 of fp32; cycle constants).
The issue ceiling bench.py prices the fused kernel against weights the instruction mix accordingly; this script is the
static cross-check of the dynamic per-class counters (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, _INT64, _CVT) that
tools/profile_gpu.sh collects.  Blocks are split into the loop's straight path (everything up to the crossing test) and the rest.
"""
import json
import re
import sys


def is64(m):
    if re.search(r"_(f64|b64|u64|i64)(_|$)", m) and not m.startswith("v_cmpx_class"):
        return True
    if m.startswith("v_cvt_") and "f64" in m:
        return True
    if m.startswith("v_mad_u64") or m.startswith("v_mad_i64") or m.startswith("v_div_") or m.startswith("v_rcp_f64"):
        return True
    return False


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") is False and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # the main loop = the LAST "Inner Loop Header" of the function (the earlier ones are the patch fill and reductions)
    hdr = [i for i, l in enumerate(body) if "Inner Loop Header" in l][-1]
    name = body[hdr].split(":")[0]
    m = re.match(r"\.LBB(\d+)_(\d+)", name)
    tag = "Header=BB%s_%s" % (m.group(1), m.group(2))
    loop = [l for i, l in enumerate(body) if i >= hdr and (i == hdr or tag in l or not l.startswith(".LBB"))]
    # cut at the first block after the header that is not in the loop
    out, inside = [], False
    for i in range(hdr, len(body)):
        l = body[i]
        if l.startswith(".LBB"):
            inside = (i == hdr) or (tag in l)
            if not inside and i > hdr:
                # blocks placed after the loop's last block end the scan only when no later block belongs to the loop
                if not any(tag in b for b in body[i:]):
                    break
            continue
        if inside:
            out.append(l.strip())
    cls = {"valu64": 0, "valu32": 0, "salu": 0, "vmem": 0, "lds": 0, "smem": 0, "other": 0}
    for l in out:
        if not l or l.startswith(";") or l.startswith("."):
            continue
        mn = l.split()[0]
        if mn.startswith("v_"):
            cls["valu64" if is64(mn) else "valu32"] += 1
        elif mn.startswith("s_load") or mn.startswith("s_buffer_load"):
            cls["smem"] += 1
        elif mn.startswith("s_"):
            cls["salu"] += 1
        elif mn.startswith("global_") or mn.startswith("buffer_") or mn.startswith("flat_") or mn.startswith("scratch_"):
            cls["vmem"] += 1
        elif mn.startswith("ds_"):
            cls["lds"] += 1
        else:
            cls["other"] += 1
    # ---- the blocks a C3 wave really executes per record ("hot"): everything but the fall-backs -- exact divisions
    # (v_div_*/v_rcp_f64), the plain IsInsideQuadrangle (v_min/max_f64), the crossing path through global memory (a block
    # that reads the crossing table from LDS AND loads points with global_load_dwordx4: lanes outside the patch) -- with the
    # four per-edge regions of the filtered cell test (the blocks with v_cmp_lt_i64 on the cross product) counted at 1/2:
    # a buoy's +x ray is level with the cell's right edge and, in warped cells, one more
    blocks, cur = [], None
    for i in range(hdr, len(body)):
        l = body[i]
        if l.startswith(".LBB") or l.startswith("; %bb."):
            if l.startswith(".LBB") and i > hdr and tag not in l and not any(tag in b for b in body[i:]):
                break
            cur = {"v64": 0, "v32": 0, "cold": False, "edge": False, "tab": False, "gld4": False}
            blocks.append(cur)
            continue
        t = l.strip()
        if cur is None or not t or t.startswith(";") or t.startswith("."):
            continue
        mn = t.split()[0]
        if mn.startswith("v_"):
            cur["v64" if is64(mn) else "v32"] += 1
            if mn.startswith(("v_div_", "v_rcp_f64", "v_min_f64", "v_max_f64")):
                cur["cold"] = True
            if mn.startswith("v_cmp_lt_i64"):
                cur["edge"] = True
        elif mn.startswith("ds_read_b128"):
            cur["tab"] = True
        elif mn.startswith("global_load_dwordx4"):
            cur["gld4"] = True
    h64 = h32 = 0.0
    skip = 0
    for b in blocks:
        if b["tab"] and b["gld4"] and b["v64"] > 20:
            skip = 2                                             # the global-memory crossing block and its lazy second half
            continue
        if skip and b["v64"] in (12, 0):
            skip -= 1
            continue
        skip = 0
        if b["cold"]:
            continue
        w = 0.5 if b["edge"] else 1.0
        h64 += w * b["v64"]; h32 += w * b["v32"]
    nv = cls["valu64"] + cls["valu32"]
    res = {"kernel": key, "loop_header": name, "static": cls, "valu": nv, "valu64_frac_static": cls["valu64"] / max(nv, 1),
           "cycles_per_valu_static": (4.0 * cls["valu64"] + 2.0 * cls["valu32"]) / max(nv, 1),
           "hot_blocks": {"valu64": h64, "valu32": h32, "valu": h64 + h32, "valu64_frac": h64 / max(h64 + h32, 1e-9),
                          "cycles_per_valu": (4.0 * h64 + 2.0 * h32) / max(h64 + h32, 1e-9)},
           "note": "static counts over every block of the loop (all paths, cold division blocks included); the dynamic mix comes from the PMC class counters"}
    print(json.dumps(res, indent=1))
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
