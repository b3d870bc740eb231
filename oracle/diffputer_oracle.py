"""CPU ORACLE for DiffPuter's E-step sampler -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
file.  The product path (`diffputer_amd/`) never does: it fails loudly when the HIP library
is missing.

This is a numpy float32 restatement of the reference's algorithm, written from the reference's
behaviour (file:line citations are relative to /root/reference).  It takes the Gaussian draws
as an explicit input (`noise[k]` = the k-th `randn`/`randn_like` call of the reference, in
program order) so that a run of the reference with recorded noise can be replayed exactly.

Parity pin: PINNED.  `tools/make_goldens.py` imports the reference's `model.py` and
`diffusion_utils.py` in the build container, records the noise it draws, and writes the
fixtures under `tests/golden/`; `tests/test_oracle_golden.py` checks every function below
against them (<= 1e-5, the difference being BLAS summation order).  The reference itself has
no tests, golden vectors or seeds for this path.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

# diffusion_utils.py:15-21
SIGMA_MIN = 0.002
SIGMA_MAX = 80.0
RHO = 7
S_CHURN = 1.0
S_MIN = 0.0
S_MAX = float("inf")
S_NOISE = 1.0
# diffusion_utils.py:37 (a local variable there, not a flag)
N_RESAMPLE = 20
# model.py:137
SIGMA_DATA = 0.5

CKPT_PREFIX = "denoise_fn_D.denoise_fn_F."


def _w(sd: dict, name: str) -> np.ndarray:
    """Fetch a tensor of the 14-tensor checkpoint by its short name."""
    if name in sd:
        return sd[name]
    return sd[CKPT_PREFIX + name]


def sigma_schedule(num_steps: int, net_sigma_min: float = 0.0,
                   net_sigma_max: float = float("inf")) -> np.ndarray:
    """Karras rho=7 schedule with a trailing 0 (diffusion_utils.py:24,27-32).

    The reference evaluates it in float32 tensor arithmetic with Python-double scalars demoted
    to float32 at each binary op; reproduced op by op.  Returns float32 [num_steps + 1].
    """
    sigma_min = max(SIGMA_MIN, net_sigma_min)
    sigma_max = min(SIGMA_MAX, net_sigma_max)
    step = np.arange(num_steps, dtype=F32)
    a = F32(sigma_max ** (1.0 / RHO))
    b = F32(sigma_min ** (1.0 / RHO) - sigma_max ** (1.0 / RHO))
    base = a + (step / F32(num_steps - 1)) * b
    t = np.power(base, F32(RHO)).astype(F32)
    return np.concatenate([t, np.zeros(1, dtype=F32)])


def silu(x: np.ndarray) -> np.ndarray:
    """torch.nn.SiLU: x * sigmoid(x) (model.py:106,108,110,117)."""
    with np.errstate(over="ignore"):          # exp(-x) -> inf for very negative x gives -0.0, as torch does
        return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def positional_embedding(c_noise: np.ndarray, dim_t: int, max_positions: int = 10000) -> np.ndarray:
    """`PositionalEmbedding.forward` followed by the sin/cos swap of `MLPDiffusion.forward`
    (model.py:24-30 and :123): returns [len(c_noise), dim_t] = [sin(a) | cos(a)] with
    a = c_noise * (1/max_positions) ** (k / (dim_t/2)), k = 0..dim_t/2-1 (endpoint=False)."""
    c_noise = np.asarray(c_noise, dtype=F32).reshape(-1)
    half = dim_t // 2
    freqs = np.arange(half, dtype=F32) / F32(half)
    freqs = np.power(F32(1.0 / max_positions), freqs).astype(F32)
    a = np.outer(c_noise, freqs).astype(F32)
    return np.concatenate([np.sin(a), np.cos(a)], axis=1).astype(F32)


def time_embedding(sd: dict, c_noise: np.ndarray) -> np.ndarray:
    """emb = time_embed(swap(map_noise(c_noise))) (model.py:122-124): Linear, SiLU, Linear,
    no trailing activation.  Returns [len(c_noise), dim_t]."""
    w0, b0 = _w(sd, "time_embed.0.weight"), _w(sd, "time_embed.0.bias")
    w2, b2 = _w(sd, "time_embed.2.weight"), _w(sd, "time_embed.2.bias")
    e = positional_embedding(c_noise, w0.shape[1])
    h = silu(e @ w0.T + b0)
    return (h @ w2.T + b2).astype(F32)


def mlp_diffusion(sd: dict, x_in: np.ndarray, c_noise: np.ndarray) -> np.ndarray:
    """`MLPDiffusion.forward(x, noise_labels)` (model.py:121-128): proj(x) + emb feeds mlp.0
    with no activation in between; SiLU after mlp.0, mlp.2, mlp.4; mlp.6 is linear."""
    emb = time_embedding(sd, c_noise)                     # [1, H] in the E-step, [N, H] in the M-step
    h = x_in @ _w(sd, "proj.weight").T + _w(sd, "proj.bias")
    h = (h + emb).astype(F32)
    h = silu(h @ _w(sd, "mlp.0.weight").T + _w(sd, "mlp.0.bias"))
    h = silu(h @ _w(sd, "mlp.2.weight").T + _w(sd, "mlp.2.bias"))
    h = silu(h @ _w(sd, "mlp.4.weight").T + _w(sd, "mlp.4.bias"))
    return (h @ _w(sd, "mlp.6.weight").T + _w(sd, "mlp.6.bias")).astype(F32)


def precond_coeffs(sigma, sigma_data: float = SIGMA_DATA):
    """c_skip, c_out, c_in, c_noise of `Precond.forward` (model.py:155-158) in float32."""
    s = np.asarray(sigma, dtype=F32)
    sd2 = F32(sigma_data ** 2)
    c_skip = sd2 / (s * s + sd2)
    c_out = s * F32(sigma_data) / np.sqrt(s * s + sd2)
    c_in = F32(1.0) / np.sqrt(sd2 + s * s)
    c_noise = np.log(s) / F32(4.0)
    return c_skip.astype(F32), c_out.astype(F32), c_in.astype(F32), c_noise.astype(F32)


def precond(sd: dict, x: np.ndarray, sigma, return_F: bool = False):
    """`Precond.forward(x, sigma)` (model.py:148-165).  sigma: 0-dim (E-step) or [N] (M-step)."""
    x = np.asarray(x, dtype=F32)
    s = np.asarray(sigma, dtype=F32).reshape(-1, 1)
    c_skip, c_out, c_in, c_noise = precond_coeffs(s)
    F_x = mlp_diffusion(sd, (c_in * x).astype(F32), c_noise.reshape(-1))
    D_x = (c_skip * x + c_out * F_x).astype(F32)
    return (D_x, F_x) if return_F else D_x


def sample_step(sd: dict, num_steps: int, i: int, t_cur, t_next, x_cur: np.ndarray,
                eps: np.ndarray) -> np.ndarray:
    """One stochastic EDM step (diffusion_utils.py:90-109).  `eps` is the churn draw of :96."""
    t_cur = F32(t_cur)
    t_next = F32(t_next)
    gamma = min(S_CHURN / num_steps, np.sqrt(2.0) - 1.0) if S_MIN <= t_cur <= S_MAX else 0.0
    t_hat = F32(t_cur + F32(gamma) * t_cur)                               # :95 (round_sigma is identity, model.py:167)
    x_hat = (x_cur + np.sqrt(t_hat * t_hat - t_cur * t_cur) * F32(S_NOISE) * eps).astype(F32)   # :96
    denoised = precond(sd, x_hat, t_hat)                                  # :99
    d_cur = ((x_hat - denoised) / t_hat).astype(F32)                      # :100
    x_next = (x_hat + (t_next - t_hat) * d_cur).astype(F32)               # :101
    if i < num_steps - 1:                                                 # :104
        denoised = precond(sd, x_next, t_next)                            # :105
        d_prime = ((x_next - denoised) / t_next).astype(F32)              # :106
        x_next = (x_hat + (t_next - t_hat) * (F32(0.5) * d_cur + F32(0.5) * d_prime)).astype(F32)  # :107
    return x_next


def impute_mask(sd: dict, x: np.ndarray, mask: np.ndarray, noise, num_steps: int = 50,
                n_resample: int = N_RESAMPLE, trace: list | None = None, final_level: bool = True) -> np.ndarray:
    """Masked reverse diffusion with RePaint resampling (diffusion_utils.py:23-68).

    x      [N, D] float32, observed values (anything at missing cells; the reference passes 0)
    mask   [N, D] bool/int, 1 = MISSING
    noise  sequence of [N, D] float32 arrays; noise[k] is the k-th Gaussian draw of the reference:
           k=0 `torch.randn` (:25); then per inner iteration n_curr (:44, dead), n_prev (:45),
           churn eps (:96, inside sample_step), n (:52, drawn always, used iff j < R-1);
           the last level draws only the first three (:59-63).
    trace  if a list, x_t after every inner iteration is appended (float32 copies).
    final_level  False = the MissDiff baseline's variant (baselines/Missdiff_SDE/diffusion_utils.py:24-60, which uses
           n_resample = 10): its loop body is guarded by `if i < num_steps - 1` with no else branch, so the last level
           does nothing, draws nothing, and x_t at sigma_min is returned.
    """
    x = np.asarray(x, dtype=F32)
    m = np.asarray(mask).astype(np.int32).astype(F32)       # :34, used arithmetically at :50
    t = sigma_schedule(num_steps)
    k = 0
    x_t = (np.asarray(noise[k], dtype=F32) * t[0]).astype(F32)      # :25,:35
    k += 1
    for i in range(num_steps if final_level else num_steps - 1):
        t_cur, t_next = t[i], t[i + 1]
        reps = n_resample if i < num_steps - 1 else 1               # :41-43 / :58
        for j in range(reps):
            k += 1                                                  # n_curr: drawn, never used (:44/:59)
            n_prev = np.asarray(noise[k], dtype=F32) * t_next       # :45/:60
            k += 1
            x_known = (x + n_prev).astype(F32)                      # :47/:62
            eps = np.asarray(noise[k], dtype=F32)
            k += 1
            x_unknown = sample_step(sd, num_steps, i, t_cur, t_next, x_t, eps)   # :48/:63
            x_prev = (x_known * (F32(1.0) - m) + x_unknown * m).astype(F32)      # :50/:65
            if i < num_steps - 1:
                n = np.asarray(noise[k], dtype=F32) * np.sqrt(t_cur * t_cur - t_next * t_next)  # :52
                k += 1
                x_t = x_prev if j == reps - 1 else (x_prev + n).astype(F32)     # :54-57
            else:
                x_t = x_prev                                                    # :66
            if trace is not None:
                trace.append(x_t.copy())
    return x_t


def sample(sd: dict, noise, num_steps: int = 50) -> np.ndarray:
    """Unconditional EDM sampler (diffusion_utils.py:70-88; not called by main.py).
    noise[0] = the latents draw (:71), noise[1 + i] = the churn draw of level i."""
    t = sigma_schedule(num_steps)
    x_next = (np.asarray(noise[0], dtype=F32) * t[0]).astype(F32)          # :82
    for i in range(num_steps):
        x_next = sample_step(sd, num_steps, i, t[i], t[i + 1], x_next, np.asarray(noise[1 + i], dtype=F32))   # :86
    return x_next


def e_step(sd: dict, X: np.ndarray, mask: np.ndarray, noise_per_trial, num_trials: int,
           num_steps: int = 50, n_resample: int = N_RESAMPLE) -> np.ndarray:
    """The E-step driver around `impute_mask` (main.py:160-190): per trial zero-fill the
    missing cells (:164), sample, blend `rec*mask + x*(1-mask)` (:182-183); mean over trials
    (:188).  Returns the mean in the sampler's (x/2) scale; the reference then multiplies by 2
    (:190) before saving."""
    mf = np.asarray(mask).astype(F32)
    x_obs = ((F32(1.0) - mf) * np.asarray(X, dtype=F32)).astype(F32)
    recs = []
    for tr in range(num_trials):
        rec = impute_mask(sd, x_obs, mask, noise_per_trial[tr], num_steps, n_resample)
        recs.append((rec * mf + x_obs * (F32(1.0) - mf)).astype(F32))
    return np.stack(recs, 0).mean(0).astype(F32)


def masked_errors(pred: np.ndarray, truth: np.ndarray, mask: np.ndarray):
    """MAE / RMSE over masked (missing) numeric cells, the metric of `get_eval`
    (dataset.py:199-201): |pred - truth| and (pred - truth)^2 averaged over mask == 1."""
    mk = np.asarray(mask).astype(bool)
    d = (np.asarray(pred, dtype=np.float64) - np.asarray(truth, dtype=np.float64))[mk]
    return float(np.abs(d).mean()), float(np.sqrt((d ** 2).mean()))
