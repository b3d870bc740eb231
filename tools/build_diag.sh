#!/bin/sh
# Diagnostic build of the library with in-kernel stamps (never loaded by the package, tests or bench.py).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/diag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DDP_PP_STAMP \
    -o tools/diag/libdiffputer_hip_stamp.so diffputer_amd/csrc/diffputer_hip.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DDP_PP_STAMP -DWS_STAMP \
    -o tools/diag/libdp_ws_stamp.so diffputer_amd/csrc/diffputer_hip.hip
# k_gemm_bf16_ws2 without its global stores: timing only, results are garbage
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -DWS2_DIAG_NOSTORE \
    -o tools/diag/libdp_ws2_nostore.so diffputer_amd/csrc/diffputer_hip.hip
# issue probe: does VALU work of one wave run under another wave's MFMAs on the same SIMD?
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/diag/issue_probe tools/issue_probe.hip
# SiLU formulations for the GEMM epilogues: cycles per value and accuracy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/diag/silu_probe tools/silu_probe.hip
# which instruction classes of one wave make progress under its SIMD partner's MFMA stream
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/diag/overlap_probe tools/overlap_probe.hip
