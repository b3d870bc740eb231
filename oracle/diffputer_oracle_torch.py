"""The CPU oracle's hot loop with torch CPU tensors -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

Why it exists: the reference's sampler is PyTorch on the CPU (its Linear layers are `addmm` on torch's BLAS, its SiLU
torch's vectorised kernel), and BASELINE.json asks for that path's speed on the GPU box's host cores.  The reference
cannot travel to the box, so `bench.py` times this restatement: the algorithm of `diffputer_oracle.impute_mask`
(same citations, same draw order, same float32 scalars from `sigma_schedule` / `precond_coeffs`) with the tensor
arithmetic done by the torch operators the reference calls.  Parity pin: PINNED through the numpy oracle -- 
`tests/test_oracle_golden.py` holds this file to the reference-generated trajectories as well.
"""
from __future__ import annotations

import numpy as np
import torch

from . import diffputer_oracle as orc

F32 = np.float32
_LAYERS = ("proj", "mlp.0", "mlp.2", "mlp.4", "mlp.6", "time_embed.0", "time_embed.2")


def to_torch(sd: dict) -> dict:
    """The 14 checkpoint tensors as torch CPU float32 tensors under their short names."""
    out = {}
    for name in _LAYERS:
        for part in ("weight", "bias"):
            out[f"{name}.{part}"] = torch.from_numpy(np.ascontiguousarray(orc._w(sd, f"{name}.{part}"), dtype=F32))
    return out


def denoiser(tsd: dict, x: torch.Tensor, sigma) -> torch.Tensor:
    """Precond.forward (model.py:148-165) around MLPDiffusion.forward (:121-128); sigma is one float32 level."""
    c_skip, c_out, c_in, c_noise = (float(v) for v in orc.precond_coeffs(F32(sigma)))
    half = tsd["time_embed.0.weight"].shape[1] // 2
    freqs = torch.pow(torch.tensor(1.0 / 10000, dtype=torch.float32), torch.arange(half, dtype=torch.float32) / half)
    a = torch.tensor([c_noise], dtype=torch.float32).outer(freqs)
    emb = torch.cat([a.sin(), a.cos()], dim=1)                                           # the swap of :123
    emb = torch.nn.functional.linear(torch.nn.functional.silu(
        torch.nn.functional.linear(emb, tsd["time_embed.0.weight"], tsd["time_embed.0.bias"])),
        tsd["time_embed.2.weight"], tsd["time_embed.2.bias"])
    h = torch.nn.functional.linear(x * c_in, tsd["proj.weight"], tsd["proj.bias"]) + emb
    for name in ("mlp.0", "mlp.2", "mlp.4"):
        h = torch.nn.functional.silu(torch.nn.functional.linear(h, tsd[f"{name}.weight"], tsd[f"{name}.bias"]))
    F_x = torch.nn.functional.linear(h, tsd["mlp.6.weight"], tsd["mlp.6.bias"])
    return x * c_skip + F_x * c_out


def sample_step(tsd: dict, num_steps: int, i: int, t_cur, t_next, x_cur: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """diffusion_utils.py:90-109 with the float32 scalars of the numpy oracle."""
    t_cur, t_next = F32(t_cur), F32(t_next)
    gamma = min(orc.S_CHURN / num_steps, np.sqrt(2.0) - 1.0)
    t_hat = F32(t_cur + F32(gamma) * t_cur)
    x_hat = x_cur + float(np.sqrt(t_hat * t_hat - t_cur * t_cur) * F32(orc.S_NOISE)) * eps
    d_cur = (x_hat - denoiser(tsd, x_hat, t_hat)) / float(t_hat)
    h = float(t_next - t_hat)
    x_next = x_hat + h * d_cur
    if i < num_steps - 1:
        d_prime = (x_next - denoiser(tsd, x_next, t_next)) / float(t_next)
        x_next = x_hat + h * (0.5 * d_cur + 0.5 * d_prime)
    return x_next


@torch.no_grad()
def impute_mask(sd: dict, x, mask, noise, num_steps: int = 50, n_resample: int = orc.N_RESAMPLE, trace: list | None = None):
    """diffusion_utils.py:23-68; arguments as `diffputer_oracle.impute_mask` (noise[k] = k-th draw, numpy or torch)."""
    tsd = to_torch(sd)
    x = torch.as_tensor(np.asarray(x, dtype=F32))
    m = torch.as_tensor(np.asarray(mask).astype(F32))
    t = orc.sigma_schedule(num_steps)
    draw = lambda k: torch.as_tensor(noise[k])
    k = 0
    x_t = draw(k) * float(t[0])
    k += 1
    for i in range(num_steps):
        t_cur, t_next = t[i], t[i + 1]
        reps = n_resample if i < num_steps - 1 else 1
        for j in range(reps):
            k += 1                                                        # n_curr: drawn, never used
            x_known = x + draw(k) * float(t_next)
            k += 1
            x_unknown = sample_step(tsd, num_steps, i, t_cur, t_next, x_t, draw(k))
            k += 1
            x_t = x_known * (1.0 - m) + x_unknown * m
            if i < num_steps - 1:
                n = draw(k) * float(np.sqrt(t_cur * t_cur - t_next * t_next))
                k += 1
                if j < reps - 1:
                    x_t = x_t + n
            if trace is not None:
                trace.append(x_t.numpy().copy())
    return x_t.numpy()
