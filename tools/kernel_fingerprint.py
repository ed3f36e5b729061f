#!/usr/bin/env python3
"""Fingerprint of the kernels AS SHIPPED: disassembles the gfx950 code object inside libsitrk.so and writes, per kernel, a hash
of its instruction stream plus static instruction-class counts -> sitrack_amd/libsitrk.isa.json (run by the Makefile after linking).

    python3 tools/kernel_fingerprint.py sitrack_amd/libsitrk.so [out.json]

Why: bench.py's `roofline` prices the fused kernel with constants measured by rocprofv3 counter passes (instructions per wave and
record, share of 64-bit classes, held clock: profiles/traffic.json).  Those passes describe ONE binary.  tools/summarize_prof.py
stores the fingerprint of the binary it profiled next to the constants; bench.py compares it with the fingerprint of the library it
loaded and marks the roofline "stale" (and drops `frac`) when they differ -- an edited kernel cannot silently keep old constants.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("SITRK_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
KERNELS = {
    "advect_run_kernel<float,1,false>": "_ZN5sitrk17advect_run_kernelIfLi1ELb0EEEvNS_7RunArgsE",
    "advect_run_kernel<float,1,true>": "_ZN5sitrk17advect_run_kernelIfLi1ELb1EEEvNS_7RunArgsE",
    "advect_step_kernel<float,1,false,512>": "_ZN5sitrk18advect_step_kernelIfLi1ELb0ELi512EEEvNS_8StepArgsE",
    "survive_kill9_rows_kernel<float>": "_ZN5sitrk25survive_kill9_rows_kernelIfEEviiNS_5SvBoxENS_7SvBatchEPKaPKT_dPaPh",
}


def is64(m):
    """a wave64 VALU instruction of the 64-bit classes (fp64 arithmetic / compares / conversions, 64-bit integer ops and moves)"""
    if re.search(r"_(f64|b64|u64|i64)(_|$)", m) and not m.startswith("v_cmpx_class"):
        return True
    if m.startswith("v_cvt_") and "f64" in m:
        return True
    return m.startswith(("v_mad_u64", "v_mad_i64", "v_div_", "v_rcp_f64"))


def fingerprint(so_path):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path, os.path.join(tmp, "x.so")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], text=True)
        out = {}
        for name, sym in KERNELS.items():
            m = re.search(r"\s(\d+)\s+FUNC\s+\S+\s+\S+\s+\d+\s+" + re.escape(sym) + r"\s*$", syms, re.M)
            if not m:
                out[name] = None
                continue
            dis = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr",
                                           "--disassemble-symbols=" + sym, co], text=True)
            cls = {"valu": 0, "valu64": 0, "salu": 0, "smem": 0, "vmem": 0, "lds": 0, "other": 0}
            h = hashlib.sha256()
            for line in dis.splitlines():
                t = line.split("//")[0].strip()
                if not t or t.endswith(":") or t.startswith(("/", "Disassembly", sym)) or "file format" in t:
                    continue
                h.update((" ".join(t.split()) + "\n").encode())
                mn = t.split()[0]
                if mn.startswith("v_"):
                    cls["valu"] += 1
                    cls["valu64"] += 1 if is64(mn) else 0
                elif mn.startswith(("s_load", "s_buffer_load")):
                    cls["smem"] += 1
                elif mn.startswith("s_"):
                    cls["salu"] += 1
                elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
                    cls["vmem"] += 1
                elif mn.startswith("ds_"):
                    cls["lds"] += 1
                else:
                    cls["other"] += 1
            out[name] = {"sha256": h.hexdigest()[:16], "code_bytes": int(m.group(1)), "static": cls}
        return out


def main():
    so = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(so)[0] + ".isa.json"
    res = {"library": os.path.basename(so), "arch": "gfx950", "kernels": fingerprint(so),
           "note": "hash of the disassembled instruction stream (mnemonics + operands) and static class counts over the whole kernel"}
    json.dump(res, open(dst, "w"), indent=1)
    print("kernel fingerprints -> %s: %s" % (dst, ", ".join("%s %s" % (k.split("<")[0], (v or {}).get("sha256")) for k, v in res["kernels"].items())))


if __name__ == "__main__":
    main()
