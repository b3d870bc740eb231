#!/usr/bin/env python3
"""A `model.pt` WRITTEN BY THE REFERENCE, and what the reference's sampler makes of it.

Build-container only (imports /root/reference/{model,diffusion_utils}.py; nothing of it is copied or shipped).
  1. the reference's `Model(MLPDiffusion(d_in, hid), hid_dim=d_in)` with torch's own default initialisation under
     torch.manual_seed(42), trained for a few Adam steps with the reference's loss (so the file is what an M-step leaves);
  2. `torch.save(model.state_dict(), path)` -- the statement of main.py:142, torch's zip format;
  3. a fresh reference model `load_state_dict(torch.load(path))` (main.py:170-173) and `impute_mask` on a small table with a
     replayed Gaussian stream.
-> tests/golden/model_ref_H64.pt (the file itself: a fixture of DATA in the reference's wire format) and
   tests/golden/model_ref_H64_io.npz (table, mask, noise seed, per-iteration x_t, output).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_ckpt_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
from make_goldens import NoiseReplay, checksum, ref_du, MLPDiffusion, Model, synth   # noqa: E402  (sets up the reference import)

OUT = os.path.join(REPO, "tests", "golden")
d_in, hid, S, R, n, nseed = 9, 64, 5, 20, 8, 4242

torch.manual_seed(42)
model = Model(denoise_fn=MLPDiffusion(d_in, hid), hid_dim=d_in)
X, mask = synth.make_table(256, d_in, seed=99)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
for _ in range(20):                                            # main.py:124-133
    loss = model(torch.from_numpy(X * (1.0 - mask)).float()).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
path = os.path.join(OUT, "model_ref_H64.pt")
torch.save(model.state_dict(), path)                           # main.py:142

fresh = Model(denoise_fn=MLPDiffusion(d_in, hid), hid_dim=d_in)
fresh.load_state_dict(torch.load(path))                        # main.py:173
fresh.eval()
net = fresh.denoise_fn_D
x_obs = torch.from_numpy((X[:n] * (1.0 - mask[:n])).astype(np.float32))
m = torch.from_numpy(mask[:n])
stream = synth.reference_noise(nseed, synth.n_draws(S, R), (n, d_in))
starts, saved = [], ref_du.sample_step


def spy(net_, num_steps, i, t_cur, t_next, x_next):
    starts.append(x_next.numpy().copy())
    return saved(net_, num_steps, i, t_cur, t_next, x_next)


ref_du.sample_step = spy
try:
    with torch.no_grad(), NoiseReplay(stream) as nr:
        out = ref_du.impute_mask(net, x_obs, m, n, d_in, S, "cpu")
finally:
    ref_du.sample_step = saved
assert nr.k == stream.shape[0]
trace = np.stack(starts[1:] + [out.numpy()])                   # x_t after every inner iteration
np.savez_compressed(os.path.join(OUT, "model_ref_H64_io.npz"), d_in=d_in, hid=hid, S=S, R=R, n=n, nseed=nseed,
                    noise_checksum=checksum(stream), x_obs=x_obs.numpy(), mask=mask[:n], trace=trace, out=out.numpy(),
                    keys=np.asarray(sorted(model.state_dict().keys())))
print(f"wrote {path} ({os.path.getsize(path)} bytes) and model_ref_H64_io.npz; |out| max {float(out.abs().max()):.3f}")
