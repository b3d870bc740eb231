#!/usr/bin/env python3
"""What the vendor GEMM (hipBLASLt through torch.matmul) reaches on the denoiser's shapes: a yardstick for
k_gemm_bf16_pp, nothing the product links or calls.   python tools/blas_ref.py"""
import torch

dev = "cuda:0"
for name, (m, n, k) in {"mlp.0": (65536, 2048, 1024), "mlp.2": (65536, 2048, 2048), "mlp.4": (65536, 1024, 2048), "sq8k": (8192, 8192, 8192)}.items():
    a = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
    for _ in range(10):
        c = a @ w.t()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c = a @ w.t()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    print(f"{name:6s} M={m} N={n} K={k}: torch.matmul med {2.0*m*n*k/ts[2]/1e9:.1f} best {2.0*m*n*k/ts[0]/1e9:.1f} TFLOP/s (plain GEMM, no bias/SiLU)")
