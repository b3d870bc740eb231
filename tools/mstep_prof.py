#!/usr/bin/env python3
"""Runs the native training step (or PyTorch eager with --eager) a fixed number of times: the command profiled with
`rocprofv3 --kernel-trace --stats -- python3 tools/mstep_prof.py` for profiles/r3_mstep_*_kernel_stats.csv."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffputer_amd.model import MLPDiffusion, Model  # noqa: E402
from diffputer_amd.mstep import NativeTrainStep      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=34)
ap.add_argument("--hid", type=int, default=1024)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--eager", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(a.batch, a.d, device=dev) * 0.5
model = Model(denoise_fn=MLPDiffusion(a.d, a.hid), hid_dim=a.d).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0, fused=True)
step = None if a.eager else NativeTrainStep(model)
for _ in range(a.steps):
    if a.eager:
        loss = model(x)
        opt.zero_grad(set_to_none=True)
        loss.backward()
    else:
        step.loss_and_grads(x)
    opt.step()
torch.cuda.synchronize()
print("done")
