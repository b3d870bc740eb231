#!/usr/bin/env python3
"""Golden fixture for the M-step's loss: the reference's `Model.forward` (model.py:171-181) =
`EDMLoss.__call__` (diffusion_utils.py:200-215) through `Precond.forward` with PER-ROW sigma (model.py:148-165), and its
gradient, on one batch with the two Gaussian draws (`torch.randn(B)` for the noise levels, `torch.randn_like(y)` for the
noise) replayed from a seeded stream.

Build-container only (imports /root/reference/{model,diffusion_utils}.py).  Stores inputs (weight seed, batch, the two
draws) and outputs (loss, per-row sigma, per-cell loss, gradients of proj.weight, mlp.2.weight, time_embed.0.weight and
mlp.6.bias -- since round 3 all 14) in tests/golden/mstep_D12_H64.npz, and a second fixture at the reference's real width
and batch size, tests/golden/mstep_D34_H1024.npz (H = 1024, D = 34, B = 4096; weight gradients as strided samples).

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

P = ref_du.EDMLoss()
pre = synth.CKPT_PREFIX
LAYERS = ["proj", "mlp.0", "mlp.2", "mlp.4", "mlp.6", "time_embed.0", "time_embed.2"]
STRIDE = (37, 41)                                 # weight gradients of the wide fixture are stored as [::37, ::41] samples


def run(d, hid, B, wseed, tseed, zseed, full):
    """One batch through the reference's Model.forward / backward with replayed draws.  full=True stores all 14 gradients;
    else (the H = 1024 case: mlp.2.weight alone is 16 MB) the 7 bias gradients in full, every weight gradient as its
    [::37, ::41] sample plus its Frobenius norm and its largest magnitude."""
    sd = synth.make_weights(d, hid, wseed, 1.0)
    model = Model(denoise_fn=MLPDiffusion(d, hid), hid_dim=d)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train()
    X, mask = synth.make_table(B, d, seed=tseed)
    x = torch.from_numpy((X * (1.0 - mask)).astype(np.float32))      # the zero-filled batch of EM iteration 0 (main.py:84-86)
    rng = np.random.Generator(np.random.Philox(zseed))
    z_sigma = rng.standard_normal(B, dtype=np.float32)
    z_noise = rng.standard_normal((B, d), dtype=np.float32)
    calls = []
    saved = (torch.randn, torch.randn_like)
    torch.randn = lambda *size, **kw: (calls.append("randn"), torch.from_numpy(z_sigma.copy()))[1]
    torch.randn_like = lambda t, **kw: (calls.append("randn_like"), torch.from_numpy(z_noise.copy()))[1]
    try:
        percell = model.loss_fn(model.denoise_fn_D, x)               # [B, d], diffusion_utils.py:200-215
        calls.clear()
        loss = model(x)                                              # model.py:178-181
    finally:
        torch.randn, torch.randn_like = saved
    assert calls == ["randn", "randn_like"], calls
    loss.backward()
    sigma = np.exp(z_sigma * np.float32(P.P_std) + np.float32(P.P_mean))
    g = {k[len(pre):]: v.grad.numpy() for k, v in model.named_parameters()}
    arrays = dict(d_in=d, hid=hid, B=B, wseed=wseed, tseed=tseed, zseed=zseed, gain=1.0, sigma=sigma, loss=np.float32(loss.item()))
    if full:
        arrays.update(x=x.numpy(), z_sigma=z_sigma, z_noise=z_noise, percell=percell.detach().numpy())
        arrays.update({"grad__" + k: v for k, v in g.items()})
    else:                                                            # inputs are regenerated from the three seeds
        arrays.update(x_sum=np.float64(x.double().sum().item()), z_sum=np.float64(z_noise.astype(np.float64).sum() + z_sigma.astype(np.float64).sum()),
                      rowloss=percell.detach().mean(1).numpy(), stride=np.asarray(STRIDE))
        for k, v in g.items():
            if v.ndim == 1:
                arrays["grad__" + k] = v
            else:
                arrays["gsample__" + k] = v[::STRIDE[0], ::STRIDE[1]].copy()
                arrays["gnorm__" + k] = np.float64(np.sqrt((v.astype(np.float64) ** 2).sum()))
                arrays["gmax__" + k] = np.float32(np.abs(v).max())
    out = os.path.join(REPO, "tests", "golden", f"mstep_D{d}_H{hid}.npz")
    np.savez_compressed(out, **arrays)
    print(f"D={d} H={hid} B={B}: loss {loss.item():.6f}; wrote {os.path.relpath(out, REPO)} ({os.path.getsize(out) // 1024} KiB)")


run(12, 64, 48, 900, 901, 902, full=True)          # small: everything stored
run(34, 1024, 4096, 910, 911, 912, full=False)     # the reference's real width and batch (main.py:25,92), adult's D
