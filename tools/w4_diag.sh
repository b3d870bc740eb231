#!/bin/sh
# timing-only ablations of k_gemm_bf16_w4: in-kernel stamps of the -DDP_PP_STAMP -DW4_DIAG_* builds (tools/diag/libdp_w4s_*.so)
set -e
cd "$GRAFT_REPO_ROOT"
for d in "$@"; do
    echo "== $d"
    DP_DIAG_LIB=tools/diag/libdp_w4s_$d.so timeout -k 10 120 python tools/pp_stamps.py --m 32768 --variant 3 | grep -E "round 1|in-kernel clock|launch"
done
