#!/usr/bin/env python3
"""Golden fixture for the M-step's loss: the reference's `Model.forward` (model.py:171-181) =
`EDMLoss.__call__` (diffusion_utils.py:200-215) through `Precond.forward` with PER-ROW sigma (model.py:148-165), and its
gradient, on one batch with the two Gaussian draws (`torch.randn(B)` for the noise levels, `torch.randn_like(y)` for the
noise) replayed from a seeded stream.

Build-container only (imports /root/reference/{model,diffusion_utils}.py).  Stores inputs (weight seed, batch, the two
draws) and outputs (loss, per-row sigma, per-cell loss, gradients of proj.weight, mlp.2.weight, time_embed.0.weight and
mlp.6.bias) in tests/golden/mstep_D12_H64.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_mstep_golden.py
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DIFFPUTER_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import diffusion_utils as ref_du            # noqa: E402  (the reference)
from model import MLPDiffusion, Model       # noqa: E402  (the reference)

from diffputer_amd import synth             # noqa: E402

d, hid, B, wseed = 12, 64, 48, 900
sd = synth.make_weights(d, hid, wseed, 1.0)
model = Model(denoise_fn=MLPDiffusion(d, hid), hid_dim=d)
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model.train()
X, mask = synth.make_table(B, d, seed=901)
x = torch.from_numpy((X * (1.0 - mask)).astype(np.float32))          # the zero-filled batch of EM iteration 0 (main.py:84-86)
rng = np.random.Generator(np.random.Philox(902))
z_sigma = rng.standard_normal(B, dtype=np.float32)
z_noise = rng.standard_normal((B, d), dtype=np.float32)

calls = []
saved = (torch.randn, torch.randn_like)
torch.randn = lambda *size, **kw: (calls.append("randn"), torch.from_numpy(z_sigma.copy()))[1]
torch.randn_like = lambda t, **kw: (calls.append("randn_like"), torch.from_numpy(z_noise.copy()))[1]
try:
    percell = model.loss_fn(model.denoise_fn_D, x)                    # [B, d], diffusion_utils.py:200-215
    calls.clear()
    loss = model(x)                                                   # model.py:178-181
finally:
    torch.randn, torch.randn_like = saved
assert calls == ["randn", "randn_like"], calls
loss.backward()
P = ref_du.EDMLoss()
sigma = np.exp(z_sigma * np.float32(P.P_std) + np.float32(P.P_mean))
g = {k: v.grad.numpy() for k, v in model.named_parameters()}
pre = synth.CKPT_PREFIX
out = os.path.join(REPO, "tests", "golden", f"mstep_D{d}_H{hid}.npz")
np.savez_compressed(out, d_in=d, hid=hid, B=B, wseed=wseed, gain=1.0, x=x.numpy(), z_sigma=z_sigma, z_noise=z_noise, sigma=sigma,
                    loss=np.float32(loss.item()), percell=percell.detach().numpy(),
                    **{"grad__" + k: g[pre + k] for k in ("proj.weight", "mlp.2.weight", "time_embed.0.weight", "mlp.6.bias")})
print(f"loss {loss.item():.6f}; wrote {os.path.relpath(out, REPO)} ({os.path.getsize(out) // 1024} KiB)")
