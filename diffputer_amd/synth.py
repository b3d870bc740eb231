"""Seeded synthetic weights and tables for tests, golden fixtures and the benchmark.

Nothing here comes from the reference: the reference has no seeds and no synthetic
data generator.  These generators exist so that a fixture can
store a *seed* instead of 42 MB of weights, and so that `bench.py` can build the
BASELINE.json table shapes (18 389x9, 22 792x34, 27 750x51, 1Mx64, 10Mx256) offline.

All randomness is `numpy.random.Generator(numpy.random.Philox(seed))`, so the same
seed gives the same bytes here and on the GPU box (same image, same numpy).
"""
from __future__ import annotations

import numpy as np

# state-dict key prefix of the reference checkpoint format (main.py:142, model.py:146,175)
CKPT_PREFIX = "denoise_fn_D.denoise_fn_F."

# (name, out_features(in terms of d_in/hid), in_features) in state-dict order (model.py:102-119)
def layer_shapes(d_in: int, hid: int):
    return [
        ("proj", hid, d_in),
        ("mlp.0", 2 * hid, hid),
        ("mlp.2", 2 * hid, 2 * hid),
        ("mlp.4", hid, 2 * hid),
        ("mlp.6", d_in, hid),
        ("time_embed.0", hid, hid),
        ("time_embed.2", hid, hid),
    ]


def make_weights(d_in: int, hid: int, seed: int, gain: float = 1.0) -> dict:
    """Random denoiser weights with torch.nn.Linear-like scale, keyed like a reference
    checkpoint (`denoise_fn_D.denoise_fn_F.<layer>.{weight,bias}`), float32, `[out, in]`.

    `gain` multiplies the weight matrices (not the biases); gain > 1 makes the random
    network's output O(1) so that parity tests exercise the c_out*F term.
    """
    rng = np.random.Generator(np.random.Philox(seed))
    sd = {}
    for name, fan_out, fan_in in layer_shapes(d_in, hid):
        bound = 1.0 / np.sqrt(fan_in)
        w = rng.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32)
        b = rng.uniform(-bound, bound, size=(fan_out,)).astype(np.float32)
        sd[CKPT_PREFIX + name + ".weight"] = (w * np.float32(gain)).astype(np.float32)
        sd[CKPT_PREFIX + name + ".bias"] = b
    return sd


def make_table(n_rows: int, d: int, seed: int = 1234, miss_rate: float = 0.3,
               latent: int = 8, chunk: int = 1 << 18):
    """Synthetic normalised table + MCAR mask in the layout the sampler consumes
    (SURVEY.md §8 a9): X float32 [n, d] with per-column std 0.5 (the reference divides the
    standardised data by 2, main.py:58), mask bool [n, d] with True = missing
    (generate_mask.py:513-514 draws `rand < p`).

    X = z A + 0.1 eps with z ~ N(0, I_latent): correlated columns, so a trained denoiser has
    something to impute from.  Built in row chunks so 10M x 256 does not need 3 copies in RAM.
    """
    rng = np.random.Generator(np.random.Philox(seed))
    A = rng.standard_normal((latent, d)).astype(np.float32) / np.float32(np.sqrt(latent))
    # analytic per-column std of z A + 0.1 eps
    col_std = np.sqrt((A.astype(np.float64) ** 2).sum(0) + 0.01).astype(np.float32)
    X = np.empty((n_rows, d), dtype=np.float32)
    mask = np.empty((n_rows, d), dtype=bool)
    for r0 in range(0, n_rows, chunk):
        r1 = min(n_rows, r0 + chunk)
        z = rng.standard_normal((r1 - r0, latent), dtype=np.float32)
        e = rng.standard_normal((r1 - r0, d), dtype=np.float32)
        X[r0:r1] = (z @ A + np.float32(0.1) * e) / col_std / np.float32(2.0)
        mask[r0:r1] = rng.random((r1 - r0, d), dtype=np.float32) < np.float32(miss_rate)
    return X, mask


def reference_noise(seed: int, n_draws: int, shape) -> np.ndarray:
    """The k-th Gaussian draw of a replayed run, as float32 [n_draws, *shape].

    Replay contract: draw k of a run is the k-th call, in program order, of
    `torch.randn` / `torch.randn_like` inside `impute_mask` (diffusion_utils.py:25,44,45,96,52).
    """
    rng = np.random.Generator(np.random.Philox(seed))
    return rng.standard_normal((n_draws,) + tuple(shape), dtype=np.float32)


def n_draws(num_steps: int, n_resample: int) -> int:
    """1 initial + 4 per inner iteration of the first S-1 levels + 3 at the last level
    (diffusion_utils.py:25,44-52,59-63): 3 924 for S=50, R=20."""
    return 1 + (num_steps - 1) * n_resample * 4 + 3
