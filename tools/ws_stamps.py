#!/usr/bin/env python3
"""Where the weight-stationary GEMM's time goes: per wave and row group, cycles from kernel start to the row group's start,
K loop, epilogue (diagnostic build tools/diag/libdp_ws_stamp.so: -DDP_PP_STAMP -DWS_STAMP; see tools/build_diag.sh).

    DP_DIAG_LIB=tools/diag/libdp_ws_stamp.so python tools/ws_stamps.py [--m 32768 --n 2048 --k 256]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffputer_amd import _lib                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=32768)
ap.add_argument("--n", type=int, default=2048)
ap.add_argument("--k", type=int, default=256)
ap.add_argument("--ws", type=int, default=2, help="dp_tune ws_gemm: 2 = k_gemm_bf16_ws, 3 = k_gemm_bf16_ws2 (K = 256)")
ap.add_argument("--act", type=int, default=1)
ap.add_argument("--timeline", type=int, default=-1, help="print every stamp of this workgroup's waves, cycles from the workgroup's first stamp")
a = ap.parse_args()
_lib.LIB_PATH = os.environ.get("DP_DIAG_LIB", "tools/diag/libdp_ws_stamp.so")
_lib.SIGNATURES["dp_debug_stamps"] = (C.c_int, [C.c_void_p])
lib = _lib.load()
_lib.check(lib.dp_tune(b"ws_gemm", a.ws), "dp_tune")      # 2: K = 192 / 256 too on the weight-stationary kernel; 3: its asm form
from diffputer_amd.sampler import linear            # noqa: E402
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(a.m, a.k, device=dev, generator=g).bfloat16()
W = (torch.randn(a.n, a.k, device=dev, generator=g) / a.k ** 0.5).bfloat16()
b = torch.randn(a.n, device=dev, generator=g)
grid = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(grid * 8 * 16 * 4, dtype=torch.int64, device=dev)
for _ in range(20):
    linear(A, W, b, act=bool(a.act))
torch.cuda.synchronize()
_lib.check(lib.dp_debug_stamps(buf.data_ptr()), "dp_debug_stamps")
linear(A, W, b, act=bool(a.act))
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(grid, 8, 16, 4).astype(np.int64)
used = s[..., 1] > 0
n_rg = int(used[0, 0].sum())
print(f"M={a.m} N={a.n} K={a.k}: {n_rg} row groups per wave")
t0 = s[:, :, 0, 0]                                   # kernel start per wave
for i in range(n_rg):
    start, kl, ep = s[:, :, i, 1] - t0, s[:, :, i, 2] - s[:, :, i, 1], s[:, :, i, 3] - s[:, :, i, 2]
    for name, w in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        print(f"  rg {i} {name}: start {np.median(start[:, w]):8.0f}  K loop {np.median(kl[:, w]):7.0f}  epilogue {np.median(ep[:, w]):7.0f}")
end = s[:, :, n_rg - 1, 3] - t0
print(f"  wave lifetime: median {np.median(end):.0f} max {end.max():.0f} cycles")
if a.timeline >= 0:
    w = s[a.timeline]
    base = w[:, 0, 0].min()
    for wave in range(8):
        row = [f"t0 {w[wave, 0, 0] - base:6d}"]
        for i in range(16):
            if w[wave, i, 1] > 0:
                row.append(f"[K {w[wave, i, 1] - base:6d} E {w[wave, i, 2] - base:6d} end {w[wave, i, 3] - base:6d}]")
        print(f"  wave {wave}: " + " ".join(row))
