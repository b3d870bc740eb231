#!/usr/bin/env python3
"""Where a ping-pong GEMM workgroup spends its time: s_memtime stamps from the diagnostic build.

    hipcc ... -DDP_PP_STAMP -o tools/diag/libdiffputer_hip_stamp.so diffputer_amd/csrc/diffputer_hip.hip   (tools/build_diag.sh)
    python tools/pp_stamps.py [--m 65536 --n 2048 --k 2048]
Stamps per workgroup and round: 0 tile start, 1 prologue landed (first barrier passed), 2 K loop done, 3 epilogue issued;
4 = s_memrealtime (100 MHz) at 3.  Shares, not absolute times, are the result (the stamps fence the schedule).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=65536)
ap.add_argument("--n", type=int, default=2048)
ap.add_argument("--k", type=int, default=2048)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--w4-min-k", type=int, default=1024)
ap.add_argument("--variant", type=int, default=2, help="dp_tune gemm_variant: 2 = ping-pong kernel, 3 = four-wave kernel")
a = ap.parse_args()
ap2 = os.environ.get("DP_DIAG_LIB", os.path.join(REPO, "tools", "diag", "libdiffputer_hip_stamp.so"))
lib = C.CDLL(ap2)
lib.dp_last_error.restype = C.c_char_p
assert lib.dp_tune(b"gemm_variant", a.variant) == 0, lib.dp_last_error()
assert lib.dp_tune(b"w4_min_k", a.w4_min_k) == 0, lib.dp_last_error()
assert lib.dp_tune(b"ws_gemm", 0) == 0, lib.dp_last_error()
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(a.m, a.k, device=dev, generator=g).bfloat16()
W = (torch.randn(a.n, a.k, device=dev, generator=g) / a.k ** 0.5).bfloat16()
b = torch.randn(a.n, device=dev, generator=g)
out = torch.empty(a.m, a.n, device=dev, dtype=torch.bfloat16)
grid = torch.cuda.get_device_properties(0).multi_processor_count   # the kernel indexes the buffer by blockIdx.x < min(tiles, CUs)
buf = torch.zeros(grid * 16 * 2 * 8, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    rc = lib.dp_linear(1, C.c_void_p(A.data_ptr()), C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()),
                       C.c_int64(a.m), a.n, a.k, 1, 0, st)
    assert rc == 0, lib.dp_last_error()


for _ in range(a.iters):
    run()
torch.cuda.synchronize()
assert lib.dp_debug_stamps(C.c_void_p(buf.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
if os.environ.get("DP_COLD"):                                       # one launch after an idle second: DVFS headroom
    import time
    for pause in (0.05, 0.5, 2.0):
        time.sleep(pause)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        c = buf.cpu().numpy().reshape(grid, 16, 2, 8).astype(np.float64)
        clk = np.median((c[:, 7, 0, 3] - c[:, 0, 0, 0]) / (c[:, 7, 0, 4] - c[:, 0, 0, 4]) * 100) if c[0, 0, 0, 4] else float("nan")
        print(f"after {pause:.2f} s idle: launch {e0.elapsed_time(e1)*1e3:.1f} us = {2.0*a.m*a.n*a.k/e0.elapsed_time(e1)/1e9:.1f} TFLOP/s")
    for _ in range(a.iters):
        run()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
ms_b2b = e0.elapsed_time(e1) / 50                                   # back to back, as in the sampler's stream
s = buf.cpu().numpy().reshape(grid, 16, 2, 8).astype(np.float64)
rounds = int((s[0, :, 0, 0] > 0).sum())
print(f"M={a.m} N={a.n} K={a.k}: launch {ms*1e3:.1f} us = {2.0*a.m*a.n*a.k/ms/1e9:.1f} TFLOP/s, rounds {rounds}")
t0_all = s[:, 0, 0, 0].min()
for r in range(rounds):
    d = s[:, r]                                                       # [grid, 2 wave rows, 8]
    pro = d[:, :, 1] - d[:, :, 0]
    loop = d[:, :, 2] - d[:, :, 1]
    epi = d[:, :, 3] - d[:, :, 2]
    start = d[:, 0, 0] - t0_all
    print(f"round {r}: start {np.median(start):9.0f} (min {start.min():9.0f} max {start.max():9.0f}) | prologue wait med {np.median(pro):7.0f} max {pro.max():7.0f} | "
          f"K loop med {np.median(loop):7.0f} min {loop.min():7.0f} max {loop.max():7.0f} | epilogue med {np.median(epi):6.0f} max {epi.max():6.0f}  [cycles]")
if s[:, :, :, 6].max() > 0:                                             # -DDP_PP_PHASES build: cycles per phase type
    nk = a.k // 64
    for row in range(2):
        d = s[:, 1, row]
        loop = d[:, 2] - d[:, 1]
        c = [np.median(d[:, 5 + k]) / nk for k in range(3)]
        c.append(np.median(loop - d[:, 5] - d[:, 6] - d[:, 7]) / nk)
        print(f"wave row {row}: cycles per phase c0..c3 = " + " ".join(f"{x:.0f}" for x in c) + f"  (sum {sum(c):.0f}; MFMA-bound would be 512 each)")
if s[:, 0, 0, 5].max() > 1e6 and s[:, :, :, 6].max() == 0:             # plain stamp build: wall-clock (100 MHz) start of every tile
    t_first = s[:, 0, 0, 5]
    t_last_end = s[:, rounds - 1, 0, 4]
    t0 = t_first.min()
    print(f"wall clock: workgroups start over {(t_first.max() - t0) / 100:.1f} us (median {(np.median(t_first) - t0) / 100:.1f}), "
          f"end between {(t_last_end.min() - t0) / 100:.1f} and {(t_last_end.max() - t0) / 100:.1f} us (median {(np.median(t_last_end) - t0) / 100:.1f}); "
          f"launch by events {ms * 1e3:.1f} us alone, {ms_b2b * 1e3:.1f} us back to back")
if rounds > 1 and s[:, 0, 0, 5].max() > 1e6 and s[:, :, :, 6].max() == 0:
    # is a workgroup's speed a property of where it runs?  wall-clock duration of the whole kernel per workgroup, by XCD
    # (workgroup id % 8) and spread within an XCD; and the correlation of a workgroup's duration between two launches
    dur = (s[:, rounds - 1, 0, 4] - s[:, 0, 0, 5]) / 100.0
    by = [dur[x::8] for x in range(8)]
    print("per-XCD median duration [us]: " + " ".join(f"{np.median(b):.1f}" for b in by) +
          f" | within-XCD spread (p95-p5) {np.median([np.percentile(b, 95) - np.percentile(b, 5) for b in by]):.1f}")
    run(); torch.cuda.synchronize()
    s2 = buf.cpu().numpy().reshape(grid, 16, 2, 8).astype(np.float64)
    dur2 = (s2[:, rounds - 1, 0, 4] - s2[:, 0, 0, 5]) / 100.0
    print(f"correlation of per-workgroup duration between two launches: {np.corrcoef(dur, dur2)[0, 1]:.2f}")
tot_cycles = (s[:, rounds - 1, 0, 3] - s[:, 0, 0, 0])
tot_real = (s[:, rounds - 1, 0, 4] - s[:, 0, 0, 4])
print(f"kernel span per WG: med {np.median(tot_cycles):.0f} cycles")
if rounds > 1:
    dc = s[:, rounds - 1, 0, 3] - s[:, 0, 0, 3]
    dr = s[:, rounds - 1, 0, 4] - s[:, 0, 0, 4]
    print(f"in-kernel clock: {np.median(dc / dr) * 100:.0f} MHz")
    mhz = dc / dr * 100
    cyc = s[:, rounds - 1, 0, 3] - s[:, 0, 0, 0]
    print("per-XCD clock [MHz]:  " + " ".join(f"{np.median(mhz[x::8]):.0f}" for x in range(8)))
    print("per-XCD cycles [k]:   " + " ".join(f"{np.median(cyc[x::8]) / 1e3:.1f}" for x in range(8)))
nk = a.k // 64
print(f"ideal MFMA cycles per tile: {nk * 4 * 512}")
