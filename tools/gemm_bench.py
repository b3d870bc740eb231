#!/usr/bin/env python3
"""A/B the GEMM kernel variants in ONE process on random data (cdna_hip_programming.md §5.4 rules 24/25).

    python tools/gemm_bench.py [--m 65536] [--variants 1,2,3] [--rounds 7]
Shapes are the denoiser's three big layers (mlp.0, mlp.2, mlp.4) at hid=1024.  Prints median / best TFLOP/s.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffputer_amd import _lib                      # noqa: E402
from diffputer_amd.sampler import linear            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=65536)
ap.add_argument("--variants", default="1,2")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--check", action="store_true")
ap.add_argument("--only", default=None)
ap.add_argument("--act", type=int, default=1)
ap.add_argument("--ab", default=None, help="toggle a dp_tune key instead of gemm_variant: key:v0,v1 (e.g. ws_gemm:0,1)")
ap.add_argument("--shapes", default=None, help="extra shapes as N:K,N:K (replaces the default list)")
a = ap.parse_args()
if os.environ.get('DP_DIAG_LIB'):
    _lib.LIB_PATH = os.environ['DP_DIAG_LIB']                # A/B a diagnostic build on one box
    import ctypes
    _probe = ctypes.CDLL(_lib.LIB_PATH)                      # an older build may lack the newest entry points
    for _name in [n for n in _lib.SIGNATURES if not hasattr(_probe, n)]:
        del _lib.SIGNATURES[_name]
lib = _lib.load()
dev = "cuda:0"
variants = [int(v) for v in a.variants.split(",")]
ab_key = b"gemm_variant"
if a.ab:
    k, vs = a.ab.split(":")
    ab_key, variants = k.encode(), [int(v) for v in vs.split(",")]
shapes = [("mlp.0", 2048, 1024), ("mlp.2", 2048, 2048), ("mlp.4", 1024, 2048), ("proj256", 1024, 256), ("mlp.6/256", 256, 1024)]
if a.shapes:
    shapes = [(f"N{n}K{k}", int(n), int(k)) for n, k in (x.split(":") for x in a.shapes.split(","))]
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K in shapes:
    if a.only and name != a.only:
        continue
    A = torch.randn(a.m, K, device=dev, generator=g).bfloat16()
    W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev, generator=g)
    res = {v: [] for v in variants}
    outs = {}
    for r in range(a.rounds):
        for v in variants:
            _lib.check(lib.dp_tune(ab_key, v), "dp_tune")
            o = linear(A, W, b, act=bool(a.act))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                o = linear(A, W, b, act=bool(a.act))
            e1.record()
            torch.cuda.synchronize()
            res[v].append(2.0 * a.m * N * K * a.iters / (e0.elapsed_time(e1) * 1e-3) / 1e12)
            outs[v] = o
    line = f"{name:10s} M={a.m} N={N} K={K}: "
    for v in variants:
        xs = sorted(res[v])
        line += f" v{v} med {xs[len(xs)//2]:7.1f} best {xs[-1]:7.1f} |"
    if a.check and len(variants) > 1:
        ref = outs[variants[0]].float()
        for v in variants[1:]:
            line += f" maxdiff v{v} {float((outs[v].float() - ref).abs().max()):.3g}"
    print(line, flush=True)
