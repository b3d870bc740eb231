#!/usr/bin/env python3
"""Build-time check of k_gemm_bf16_pp's counted waits against the code hipcc actually emitted.

The kernel issues the NEXT tile's prologue LDS-DMA before the current tile's epilogue and then waits with
`s_waitcnt vmcnt(8 + NST)` / `vmcnt(10 + NST)`, where NST is the number of vector-memory instructions one wave's epilogue
issues behind that DMA (dp_gemm_bf16_pp.h: 16 for the bf16 output, 32 for float32).  The source pins the order with
scheduling barriers; this script checks the count: it compiles the kernel to gfx950 assembly and, per variant, requires
exactly NST `global_store_dwordx4` in the function and the two counted waits to be present.  Exit code 0 = ok.

k_gemm_bf16_w4 (dp_gemm_bf16_w4.h) counts the same way -- four LDS-DMA instructions per phase, W4_NST = 32 epilogue stores per
wave, vmcnt(24) / vmcnt(24 + NST) -- and must not touch scratch memory at all (a spill reload is a vector-memory load: its
compiler-placed vmcnt(0) would drain the DMA stream).  Checked here: 32 stores, 4 DMA instructions between consecutive
barriers of the tile loop, each M0 write one MFMA ahead of its DMA, no scratch access, vmcnt(0) only before the stream starts
and at the kernel's end.

    python tools/check_pp_asm.py          (needs hipcc; about 15 s)
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
    src = ['#include "%s"' % os.path.join(REPO, "diffputer_amd", "csrc", "dp_gemm_bf16_w4.h"),
           "namespace dp { thread_local std::string g_err; }"]
    for act in ("true", "false"):
        src.append("template __global__ void dp::k_gemm_bf16_w4<%s>(const dp::bf16_t*, const dp::bf16_t*, const float*, dp::bf16_t*, "
                   "int, int, int, int, unsigned long long*);" % act)
    for act, epi, _ in VARIANTS:
        src.append("template __global__ void dp::k_gemm_bf16_pp<%s, %d>(const dp::bf16_t*, const dp::bf16_t*, const float*, void*, "
                   "int, int, int, int, dp::MidStep, dp::PpBatch);" % (act, epi))
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
    for act in ("true", "false"):
        name = "_ZN2dp14k_gemm_bf16_w4ILb%dEEE" % (1 if act == "true" else 0)
        m = re.search(r"^%s\w*:[^\n]*\n(.*?)\n\s*s_endpgm" % re.escape(name), text, flags=re.S | re.M)
        if not m:
            print("kernel %s not found in the assembly" % name)
            bad += 1
            continue
        lines = [l.strip() for l in m.group(1).split("\n") if l.strip() and not l.strip().startswith(";")]
        stores = sum(l.startswith("global_store_dwordx4") for l in lines)
        scratch = sum(l.startswith("scratch_") for l in lines)
        mfma = sum(l.startswith("v_mfma") for l in lines)
        # phases = runs of instructions that contain 32 MFMAs and end in s_barrier
        phases, dma, n_mfma, order_ok, m0_fresh = [], 0, 0, True, False
        for l in lines:
            if l.startswith("v_mfma"):
                n_mfma += 1
            elif l.startswith("global_load_lds"):
                dma += 1
                order_ok &= m0_fresh or n_mfma == 0          # inside a phase: an M0 write, then >= 1 MFMA, then the load
                m0_fresh = False
            elif re.match(r"s_(mov_b32|add_u32|add_i32) m0\b", l):
                m0_fresh = True
                m0_at = n_mfma
            elif l.startswith("s_barrier"):
                if n_mfma == 32:
                    phases.append(dma)
                dma, n_mfma, m0_fresh = 0, 0, False
        w0 = [i for i, l in enumerate(lines) if re.search(r"vmcnt\(0\)", l)]
        first_mfma = next(i for i, l in enumerate(lines) if l.startswith("v_mfma"))
        last_mfma = max(i for i, l in enumerate(lines) if l.startswith("v_mfma"))
        drains_inside = [i for i in w0 if first_mfma < i < last_mfma]
        w24 = sum("vmcnt(24)" in l for l in lines)
        w56 = sum("vmcnt(56)" in l for l in lines)
        ok = (stores == 32 and scratch == 0 and mfma == 768 and len(phases) == 24 and all(d == 4 for d in phases)
              and not drains_inside and w24 >= 18 and w56 == 6 and order_ok)
        print("k_gemm_bf16_w4<%s>: %d epilogue stores (NST 32), %d MFMA in %d phases with %s DMA each, vmcnt(24) x%d, vmcnt(56) x%d, "
              "%d vmcnt(0) inside the tile loop, %d scratch accesses -> %s"
              % (act, stores, mfma, len(phases), sorted(set(phases)), w24, w56, len(drains_inside), scratch, "ok" if ok else "MISMATCH"))
        bad += 0 if ok else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
