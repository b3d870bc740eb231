"""CPU oracle of the M-step's training step -- TEST INFRASTRUCTURE, never the product path.

A numpy float32 restatement, with the backward pass written out by hand, of what the reference computes in
`loss = model(inputs); loss.backward()` (main.py:124-131):

    Model.forward            model.py:171-181          loss = EDMLoss(...).mean(-1).mean()
    EDMLoss.__call__         diffusion_utils.py:200-215 sigma = exp(1.2 z - 1.2); weight = (sigma^2 + sd^2) / (sigma sd)^2;
                                                       n = randn_like(y) sigma; loss = weight (D(y + n, sigma) - y)^2
    Precond.forward          model.py:148-165          D = c_skip x + c_out F(c_in x, ln(sigma) / 4), per-row sigma
    MLPDiffusion.forward     model.py:121-128          emb = time_embed(swap(map_noise(c_noise))); F = mlp(proj(x) + emb)
    PositionalEmbedding      model.py:17-30, :123      [cos | sin] swapped to [sin | cos]

The two Gaussian draws are explicit inputs (z_sigma = the `torch.randn(B)` of diffusion_utils.py:202, z_noise = the
`torch.randn_like(y)` of :208), so a run of the reference with those draws replayed is reproduced cell for cell.
Pinned by tests/test_oracle_golden.py to the reference's own loss, per-cell loss and gradients
(tests/golden/mstep_D12_H64.npz: all 14 tensors; mstep_D34_H1024.npz: H = 1024, batch 4096).  Only tests/ and
__graft_entry__.smoke() may import this module.
"""
from __future__ import annotations

import numpy as np

PREFIX = "denoise_fn_D.denoise_fn_F."
LAYERS = ["proj", "mlp.0", "mlp.2", "mlp.4", "mlp.6", "time_embed.0", "time_embed.2"]
f32 = np.float32


def _get(sd, name):
    for k in (name, PREFIX + name):
        if k in sd:
            return np.asarray(sd[k], dtype=f32)
    raise KeyError(name)


def _sigmoid(a):
    return (f32(1.0) / (f32(1.0) + np.exp(-a))).astype(f32)


def _silu(a):
    return (a * _sigmoid(a)).astype(f32)


def _dsilu(a):                                   # d/da [a sigmoid(a)]
    s = _sigmoid(a)
    return (s + a * s * (f32(1.0) - s)).astype(f32)


def loss_and_grads(sd, x, z_sigma, z_noise, P_mean=-1.2, P_std=1.2, sigma_data=0.5):
    """-> (loss, percell [B, D], grads {"<layer>.weight" / "<layer>.bias": array}) for one batch x [B, D]."""
    x = np.asarray(x, dtype=f32)
    B, D = x.shape
    W = {n: _get(sd, n + ".weight") for n in LAYERS}
    b = {n: _get(sd, n + ".bias") for n in LAYERS}
    H = W["proj"].shape[0]
    sd_ = f32(sigma_data)
    # ---- EDMLoss (diffusion_utils.py:202-209) ----
    sigma = np.exp(np.asarray(z_sigma, dtype=f32) * f32(P_std) + f32(P_mean)).astype(f32)[:, None]
    weight = ((sigma ** 2 + sd_ ** 2) / (sigma * sd_) ** 2).astype(f32)
    noisy = (x + np.asarray(z_noise, dtype=f32) * sigma).astype(f32)
    # ---- Precond (model.py:155-160) ----
    c_skip = (sd_ ** 2 / (sigma ** 2 + sd_ ** 2)).astype(f32)
    c_out = (sigma * sd_ / np.sqrt(sigma ** 2 + sd_ ** 2)).astype(f32)
    c_in = (f32(1.0) / np.sqrt(sd_ ** 2 + sigma ** 2)).astype(f32)
    c_noise = (np.log(sigma) / f32(4.0)).astype(f32)[:, 0]
    u = (c_in * noisy).astype(f32)
    # ---- MLPDiffusion (model.py:121-128) ----
    half = H // 2
    freqs = (f32(1.0) / f32(10000.0)) ** (np.arange(half, dtype=f32) / f32(half))
    ang = np.outer(c_noise, freqs.astype(f32)).astype(f32)
    pe = np.concatenate([np.sin(ang), np.cos(ang)], axis=1).astype(f32)       # [cos | sin] with the halves swapped (:123)
    at0 = pe @ W["time_embed.0"].T + b["time_embed.0"]
    t1 = _silu(at0)
    emb = t1 @ W["time_embed.2"].T + b["time_embed.2"]
    h0 = u @ W["proj"].T + b["proj"] + emb
    a1 = h0 @ W["mlp.0"].T + b["mlp.0"]; h1 = _silu(a1)
    a2 = h1 @ W["mlp.2"].T + b["mlp.2"]; h2 = _silu(a2)
    a3 = h2 @ W["mlp.4"].T + b["mlp.4"]; h3 = _silu(a3)
    F = h3 @ W["mlp.6"].T + b["mlp.6"]
    Dx = c_skip * noisy + c_out * F                                           # model.py:164
    diff = (Dx - x).astype(f32)
    percell = (weight * diff ** 2).astype(f32)                                # diffusion_utils.py:213
    loss = f32(percell.mean(axis=-1).mean())                                  # model.py:180-181
    # ---- backward ----
    g = {}
    dF = (f32(2.0 / (B * D)) * weight * diff * c_out).astype(f32)

    def lin(name, dY, X):                                                     # Y = X W^T + b
        g[name + ".weight"] = (dY.T @ X).astype(f32)
        g[name + ".bias"] = dY.sum(axis=0).astype(f32)
        return (dY @ W[name]).astype(f32)

    da3 = lin("mlp.6", dF, h3) * _dsilu(a3)
    da2 = lin("mlp.4", da3, h2) * _dsilu(a2)
    da1 = lin("mlp.2", da2, h1) * _dsilu(a1)
    dh0 = lin("mlp.0", da1, h0)
    lin("proj", dh0, u)                                                       # no gradient flows to the data
    dat0 = lin("time_embed.2", dh0, t1) * _dsilu(at0)
    lin("time_embed.0", dat0, pe)
    return loss, percell, g
