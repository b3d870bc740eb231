#!/usr/bin/env python3
"""M-step (main.py:92-152) step time on the GPU: eager PyTorch vs the same step replayed from a HIP graph.
    python tools/mstep_bench.py [--d 9 --hid 1024 --batch 4096 --steps 200]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffputer_amd.model import MLPDiffusion, Model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=9)
ap.add_argument("--hid", type=int, default=1024)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
data = torch.randn(a.batch * 4, a.d, device=dev) * 0.5


def build(capturable):
    model = Model(denoise_fn=MLPDiffusion(a.d, a.hid), hid_dim=a.d).to(dev)
    lr = torch.tensor(1e-4, device=dev) if capturable else 1e-4
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0, capturable=capturable)
    return model, opt


def timeit(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3


model, opt = build(False)
xb = data[: a.batch]


def eager():
    loss = model(xb).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


print(f"d={a.d} hid={a.hid} batch={a.batch}: eager {timeit(eager):.3f} ms/step")

model, opt = build(False)


def eager_bf16():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = model(xb).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


print(f"d={a.d} hid={a.hid} batch={a.batch}: eager, bf16 autocast {timeit(eager_bf16):.3f} ms/step")

model, opt = build(True)
static_x = data[: a.batch].clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        loss = model(static_x).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = model(static_x).mean()
    static_loss.backward()
    opt.step()


def graphed():
    static_x.copy_(xb)
    g.replay()


print(f"d={a.d} hid={a.hid} batch={a.batch}: graph {timeit(graphed):.3f} ms/step (loss {float(static_loss):.4f})")
# ---- the native training step (dp_mstep_fwd_bwd: bf16 hi/lo split, three MFMA products per float32 product) + PyTorch's Adam ----
from diffputer_amd.mstep import NativeTrainStep  # noqa: E402

for fused in (False, True):
    model, _ = build(False)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0, fused=fused)
    step = NativeTrainStep(model)

    def native():
        step.loss_and_grads(xb)
        opt.step()

    print(f"d={a.d} hid={a.hid} batch={a.batch}: native fwd+bwd + Adam(fused={fused}) {timeit(native):.3f} ms/step")

model, _ = build(False)
step = NativeTrainStep(model)
print(f"d={a.d} hid={a.hid} batch={a.batch}: native fwd+bwd only {timeit(lambda: step.loss_and_grads(xb)):.3f} ms/step")
model, _ = build(False)


def eager_nograd_step():
    loss = model(xb).mean()
    model.zero_grad(set_to_none=True)
    loss.backward()


print(f"d={a.d} hid={a.hid} batch={a.batch}: eager fwd+bwd only {timeit(eager_nograd_step):.3f} ms/step")
flops = 3 * a.batch * (4.0 * a.d * a.hid + 16.0 * a.hid * a.hid + 4.0 * a.hid * a.hid)
print(f"GEMM work per step (fwd + 2x bwd, incl. the per-row time embedding) = {flops/1e12:.3f} TFLOP")
