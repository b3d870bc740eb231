#!/usr/bin/env python3
"""Build-time check of k_gemm_bf16_pp's counted waits against the code hipcc actually emitted.

The kernel issues the NEXT tile's prologue LDS-DMA before the current tile's epilogue and then waits with
`s_waitcnt vmcnt(8 + NST)` / `vmcnt(10 + NST)`, where NST is the number of vector-memory instructions one wave's epilogue
issues behind that DMA (dp_gemm_bf16_pp.h: 16 for the bf16 output, 32 for float32).  The source pins the order with
scheduling barriers; this script checks the count: it compiles the kernel to gfx950 assembly and, per variant, requires
exactly NST `global_store_dwordx4` in the function and the two counted waits to be present.  Exit code 0 = ok.

    python tools/check_pp_asm.py          (needs hipcc; about 10 s)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
VARIANTS = [("true", 0, 16), ("false", 0, 16), ("true", 1, 32), ("false", 1, 32)]      # ACT, EPI, NST


def main() -> int:
    src = ['#include "%s"' % os.path.join(REPO, "diffputer_amd", "csrc", "dp_gemm_bf16_pp.h"),
           "namespace dp { thread_local std::string g_err; }"]
    for act, epi, _ in VARIANTS:
        src.append("template __global__ void dp::k_gemm_bf16_pp<%s, %d>(const dp::bf16_t*, const dp::bf16_t*, const float*, void*, "
                   "int, int, int, int, dp::MidStep);" % (act, epi))
    with tempfile.TemporaryDirectory() as d:
        hip, asm = os.path.join(d, "pp.hip"), os.path.join(d, "pp.s")
        open(hip, "w").write("\n".join(src) + "\n")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", asm, hip],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(asm).read()
    bad = 0
    for act, epi, nst in VARIANTS:
        name = "_ZN2dp14k_gemm_bf16_ppILb%dELi%dEEE" % (1 if act == "true" else 0, epi)
        m = re.search(r"^%s\w*:[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(name), text, flags=re.S | re.M)
        if not m:
            print("kernel %s not found in the assembly" % name)
            bad += 1
            continue
        body = m.group(1)
        stores = len(re.findall(r"^\s*global_store_dwordx4\b", body, flags=re.M))
        w8 = len(re.findall(r"s_waitcnt vmcnt\(%d\)" % (8 + nst), body))
        w10 = len(re.findall(r"s_waitcnt vmcnt\(%d\)" % (10 + nst), body))
        scratch = len(re.findall(r"scratch_(load|store)", body))
        ok = stores == nst and w8 >= 2 and w10 >= 2 and scratch == 0
        print("k_gemm_bf16_pp<%s, %d>: %d epilogue stores (NST %d), vmcnt(%d) x%d, vmcnt(%d) x%d, %d scratch accesses -> %s"
              % (act, epi, stores, nst, 8 + nst, w8, 10 + nst, w10, scratch, "ok" if ok else "MISMATCH"))
        bad += 0 if ok else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
