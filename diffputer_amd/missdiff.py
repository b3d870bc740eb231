"""The MissDiff baseline's sampler on the same kernels.

`baselines/Missdiff_SDE/diffusion_utils.py:24-60` is the reference's `impute_mask` with two differences: the inner
resampling count is `N = 10` (:38) and the loop body is guarded by `if i < num_steps - 1` with no else branch, so the
final level (Euler step to sigma = 0) never runs and `x_t` at sigma_min is returned.  Same signature as the reference's.
"""
from __future__ import annotations

from . import diffusion_utils as _du

N_RESAMPLE = 10


def impute_mask(net, x, mask, num_samples, dim, num_steps=50, device="cuda:0", *, noise=None, seed=None, dtype=None):
    if tuple(x.shape) != (num_samples, dim):
        raise ValueError(f"x has shape {tuple(x.shape)} but num_samples, dim = {num_samples}, {dim}")
    plan = _du.plan_for(net, num_steps, device, dtype, N_RESAMPLE)
    if seed is None:
        seed = next(_du._seed_counter)
    # the reference returns x_t unblended; dp_impute returns x_t*mask + x*(1-mask), so observed cells come back
    # as x rather than x + sigma_min * noise -- what every caller of this function does next anyway (main.py:182-183)
    out, _ = plan.impute_sum(x, mask, 1, seed=seed, noise=noise, final_level=False)
    return out
