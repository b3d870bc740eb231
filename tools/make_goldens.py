#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE on CPU.

Build-container only: imports /root/reference/{model,diffusion_utils}.py (never copied into this
repo, never shipped to the GPU box), feeds it seeded weights (diffputer_amd.synth.make_weights)
and a recorded Gaussian stream (diffputer_amd.synth.reference_noise), and stores inputs +
outputs as small .npz files.  Fixtures hold data only -- no reference source.

Noise replay: `torch.randn`, `torch.randn_like` and the module-level alias
`diffusion_utils.randn_like` (diffusion_utils.py:13) are replaced for the duration of a call by a
function that returns the next array of a pre-generated [n_draws, N, D] float32 stream, so draw
k of the stream is the k-th Gaussian call of the reference in program order.  Fixtures store the
stream's seed (plus a float64 checksum), or the stream itself when it is small.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py [--only NAME]
"""
from __future__ import annotations

import argparse
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

OUT = os.path.join(REPO, "tests", "golden")


class NoiseReplay:
    """Context manager that makes the reference draw from `stream` ([n, N, D] float32)."""

    def __init__(self, stream: np.ndarray):
        self.stream = stream
        self.k = 0

    def _next(self, shape):
        a = self.stream[self.k]
        assert tuple(a.shape) == tuple(shape), (a.shape, shape)
        self.k += 1
        return torch.from_numpy(a.copy())

    def __enter__(self):
        self._saved = (torch.randn, torch.randn_like, ref_du.randn_like)
        torch.randn = lambda *size, **kw: self._next(size[0] if len(size) == 1 and not isinstance(size[0], int) else size)
        torch.randn_like = lambda t, **kw: self._next(t.shape)
        ref_du.randn_like = torch.randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, ref_du.randn_like = self._saved


def ref_net(d_in: int, hid: int, wseed: int, gain: float):
    """The reference's Precond(MLPDiffusion) with seeded weights loaded through its own
    state-dict keys (main.py:170-177)."""
    sd = synth.make_weights(d_in, hid, wseed, gain)
    model = Model(denoise_fn=MLPDiffusion(d_in, hid), hid_dim=d_in)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    return model.denoise_fn_D, sd


def checksum(a: np.ndarray) -> float:
    return float(np.asarray(a, dtype=np.float64).sum())


def save(name: str, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path, REPO)}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------------------------
def gold_schedule():
    """sigma schedule (diffusion_utils.py:24-32) for S in {5, 10, 50}: run impute_mask's own
    expression through the reference by calling it with a net that records t_steps."""
    out = {}
    for S in (5, 10, 50):
        step_indices = torch.arange(S, dtype=torch.float32)
        sigma_min = max(ref_du.SIGMA_MIN, 0)
        sigma_max = min(ref_du.SIGMA_MAX, float("inf"))
        # the reference's expression, evaluated by torch exactly as impute_mask does (:30-32)
        t_steps = (sigma_max ** (1 / ref_du.rho) + step_indices / (S - 1) * (
            sigma_min ** (1 / ref_du.rho) - sigma_max ** (1 / ref_du.rho))) ** ref_du.rho
        t_steps = torch.cat([t_steps, torch.zeros_like(t_steps[:1])])
        out[f"t_S{S}"] = t_steps.numpy()
    # cross-check against what impute_mask really uses: capture t_cur/t_next seen by sample_step
    seen = []
    saved = ref_du.sample_step
    net, _ = ref_net(3, 8, 1, 1.0)

    def spy(net_, num_steps, i, t_cur, t_next, x_next):
        seen.append((float(t_cur), float(t_next)))
        return saved(net_, num_steps, i, t_cur, t_next, x_next)

    ref_du.sample_step = spy
    try:
        S = 5
        stream = synth.reference_noise(0, synth.n_draws(S, 20), (2, 3))
        with NoiseReplay(stream):
            ref_du.impute_mask(net, torch.zeros(2, 3), torch.ones(2, 3, dtype=torch.bool), 2, 3, S, "cpu")
    finally:
        ref_du.sample_step = saved
    lv = [seen[0][0]] + [seen[20 * i][1] for i in range(S - 1)] + [seen[-1][1]]
    assert np.array_equal(np.asarray(lv, dtype=np.float32), out["t_S5"]), (lv, out["t_S5"])
    save("schedule", **out)


def gold_embedding():
    """time embedding for the 99 noise levels of an S=50 run, H in {64, 1024} (model.py:122-124)."""
    for hid in (64, 1024):
        net, _ = ref_net(9, hid, 100 + hid, 1.0)
        t = torch.from_numpy(np.load(os.path.join(OUT, "schedule.npz"))["t_S50"])
        gamma = min(ref_du.S_churn / 50, np.sqrt(2) - 1)
        t_hat = t[:-1] + gamma * t[:-1]
        sig = torch.cat([t_hat, t[1:-1]])                        # 50 + 49 = 99 levels
        c_noise = sig.log() / 4
        F = net.denoise_fn_F
        with torch.no_grad():
            emb = F.map_noise(c_noise)
            emb = emb.reshape(emb.shape[0], 2, -1).flip(1).reshape(*emb.shape)
            pos = emb.clone()
            emb = F.time_embed(emb)
        save(f"embedding_H{hid}", wseed=100 + hid, d_in=9, hid=hid, gain=1.0, sigma=sig.numpy(),
             posemb=pos.numpy(), emb=emb.numpy())


def gold_denoiser():
    """Single evaluations D(x, sigma), F(c_in x, c_noise) at the real H = 1024 (model.py:148-165)."""
    for d_in in (9, 64, 256):
        wseed, gain = 200 + d_in, 2.0
        net, _ = ref_net(d_in, 1024, wseed, gain)
        rng = np.random.Generator(np.random.Philox(300 + d_in))
        sigmas = np.asarray([81.6, 7.5, 0.6, 0.05, 0.00204], dtype=np.float32)
        xs, Ds, Fs = [], [], []
        for s in sigmas:
            x = (rng.standard_normal((16, d_in), dtype=np.float32) * np.float32(np.sqrt(s * s + 0.25)))
            with torch.no_grad():
                xt = torch.from_numpy(x)
                st = torch.tensor(s)
                D = net(xt, st)
                c_in = 1 / (0.25 + st ** 2).sqrt()
                Fx = net.denoise_fn_F(c_in * xt, (st.log() / 4).reshape(1))
            xs.append(x); Ds.append(D.numpy()); Fs.append(Fx.numpy())
        save(f"denoiser_D{d_in}_H1024", wseed=wseed, gain=gain, d_in=d_in, hid=1024, sigma=sigmas,
             x=np.stack(xs), D=np.stack(Ds), F=np.stack(Fs))
        print(f"    |F| mean {np.abs(np.stack(Fs)).mean():.3f}  |D| mean {np.abs(np.stack(Ds)).mean():.3f}")


def gold_sample_step():
    """sample_step (diffusion_utils.py:90-109) at a Heun level and at the last (Euler-only) level."""
    d_in, hid, S, wseed, gain = 9, 64, 10, 400, 2.0
    net, _ = ref_net(d_in, hid, wseed, gain)
    t = np.load(os.path.join(OUT, "schedule.npz"))["t_S10"]
    rng = np.random.Generator(np.random.Philox(401))
    cases = {}
    for i in (0, 4, 8, 9):
        x = rng.standard_normal((12, d_in), dtype=np.float32) * np.float32(max(t[i], 0.5))
        eps = rng.standard_normal((1, 12, d_in), dtype=np.float32)
        with NoiseReplay(eps), torch.no_grad():
            y = ref_du.sample_step(net, S, i, torch.tensor(t[i]), torch.tensor(t[i + 1]), torch.from_numpy(x))
        cases[f"x_{i}"] = x
        cases[f"eps_{i}"] = eps[0]
        cases[f"y_{i}"] = y.numpy()
    save("sample_step_D9_H64_S10", wseed=wseed, gain=gain, d_in=d_in, hid=hid, S=S,
         levels=np.asarray([0, 4, 8, 9]), **cases)


def run_impute(d_in, hid, S, n, wseed, gain, nseed, xseed, miss=0.3, trace_every=1):
    net, _ = ref_net(d_in, hid, wseed, gain)
    X, mask = synth.make_table(n, d_in, seed=xseed, miss_rate=miss)
    x_obs = (X * (1.0 - mask.astype(np.float32))).astype(np.float32)      # main.py:164
    stream = synth.reference_noise(nseed, synth.n_draws(S, 20), (n, d_in))
    starts = []
    saved = ref_du.sample_step

    def spy(net_, num_steps, i, t_cur, t_next, x_next):
        starts.append(x_next.numpy().copy())
        return saved(net_, num_steps, i, t_cur, t_next, x_next)

    ref_du.sample_step = spy
    try:
        with NoiseReplay(stream) as nr:
            out = ref_du.impute_mask(net, torch.from_numpy(x_obs), torch.from_numpy(mask), n, d_in, S, "cpu")
            assert nr.k == stream.shape[0], (nr.k, stream.shape)
    finally:
        ref_du.sample_step = saved
    out = out.numpy()
    # x_t after inner iteration k = the x_t handed to iteration k+1; the last one is the result
    trace = np.stack(starts[1:] + [out])
    idx = np.arange(trace.shape[0])
    keep = (idx % trace_every == trace_every - 1) | (idx == trace.shape[0] - 1)
    return dict(wseed=wseed, gain=gain, d_in=d_in, hid=hid, S=S, R=20, n=n, nseed=nseed,
                noise_checksum=checksum(stream), X=X, mask=mask, x_obs=x_obs, out=out,
                trace=trace[keep], trace_idx=idx[keep]), stream


def gold_impute():
    """impute_mask end to end (diffusion_utils.py:23-68) with per-iteration x_t."""
    # tiny: noise stored in the file, so the fixture does not depend on numpy's Philox stream
    g, stream = run_impute(9, 64, 5, 8, wseed=500, gain=2.0, nseed=501, xseed=502)
    save("impute_D9_H64_S5_N8", noise=stream, **g)
    # mixed-type width (adult stand-in D=34), wider net
    g, _ = run_impute(34, 256, 10, 16, wseed=510, gain=2.0, nseed=511, xseed=512)
    save("impute_D34_H256_S10_N16", **g)
    # full S=50 schedule at small H
    g, _ = run_impute(9, 64, 50, 8, wseed=520, gain=2.0, nseed=521, xseed=522)
    save("impute_D9_H64_S50_N8", **g)
    # the real network width, full schedule, few rows (GPU parity at H=1024)
    g, _ = run_impute(9, 1024, 50, 16, wseed=530, gain=2.0, nseed=531, xseed=532, trace_every=20)
    save("impute_D9_H1024_S50_N16", **g)
    g, _ = run_impute(64, 1024, 10, 32, wseed=540, gain=2.0, nseed=541, xseed=542, trace_every=20)
    save("impute_D64_H1024_S10_N32", **g)


def gold_impute_news():
    """news stand-in width (45 numeric + 3 + 3 binary-digit columns, D = 51 -- not a multiple of 4, so the
    step kernels run one cell per thread and the GEMMs pad K and N to 64)."""
    g, _ = run_impute(51, 256, 10, 16, wseed=550, gain=2.0, nseed=551, xseed=552)
    save("impute_D51_H256_S10_N16", **g)


def gold_estep():
    """The E-step driver of main.py:160-190 for T=3 trials: per-trial impute_mask, final blend,
    stack().mean(0).  Stored in the sampler's x/2 scale (before the *2 of main.py:190)."""
    d_in, hid, S, n, T = 9, 64, 5, 12, 3
    net, _ = ref_net(d_in, hid, 600, 2.0)
    X, mask = synth.make_table(n, d_in, seed=602)
    Xt, mt = torch.from_numpy(X), torch.from_numpy(mask)
    recs = []
    for trial in range(T):
        stream = synth.reference_noise(610 + trial, synth.n_draws(S, 20), (n, d_in))
        impute_X = (1. - mt.float()) * Xt                                           # main.py:164
        with NoiseReplay(stream):
            rec = ref_du.impute_mask(net, impute_X, mt, n, d_in, S, "cpu")           # main.py:180
        mask_int = mt.to(torch.float)
        recs.append(rec * mask_int + impute_X * (1 - mask_int))                     # main.py:182-183
    mean = torch.stack(recs, dim=0).mean(0)                                         # main.py:188
    save("estep_D9_H64_S5_T3", wseed=600, gain=2.0, d_in=d_in, hid=hid, S=S, R=20, n=n, T=T,
         nseeds=np.asarray([610, 611, 612]), X=X, mask=mask, mean=mean.numpy())


def gold_variants():
    """the unconditional sampler `sample` (diffusion_utils.py:70-88) and the MissDiff baseline's
    impute_mask (baselines/Missdiff_SDE/diffusion_utils.py:24-60: N = 10 and no final level)."""
    import importlib.util
    d_in, hid, S, n = 9, 64, 10, 8
    net, _ = ref_net(d_in, hid, 700, 2.0)
    stream = synth.reference_noise(701, 1 + S, (n, d_in))
    with NoiseReplay(stream) as nr:
        out = ref_du.sample(net, n, d_in, S, "cpu")
        assert nr.k == 1 + S
    save("sample_D9_H64_S10", wseed=700, gain=2.0, d_in=d_in, hid=hid, S=S, n=n, noise=stream, out=out.numpy())

    spec = importlib.util.spec_from_file_location("missdiff_du", os.path.join(REF, "baselines", "Missdiff_SDE", "diffusion_utils.py"))
    md = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(md)
    R = 10
    X, mask = synth.make_table(n, d_in, seed=712)
    x_obs = (X * (1.0 - mask.astype(np.float32))).astype(np.float32)
    stream = synth.reference_noise(711, 1 + 4 * R * (S - 1), (n, d_in))
    saved = md.randn_like
    with NoiseReplay(stream) as nr:
        md.randn_like = torch.randn_like                      # the module-level alias (:13) must see the patched function
        try:
            out = md.impute_mask(net, torch.from_numpy(x_obs), torch.from_numpy(mask), n, d_in, S, "cpu")
        finally:
            md.randn_like = saved
        assert nr.k == stream.shape[0], (nr.k, stream.shape)
    save("missdiff_D9_H64_S10", wseed=700, gain=2.0, d_in=d_in, hid=hid, S=S, R=R, n=n, noise=stream, X=X, mask=mask,
         x_obs=x_obs, out=out.numpy())


ALL = {"variants": gold_variants, "schedule": gold_schedule, "embedding": gold_embedding, "denoiser": gold_denoiser,
       "sample_step": gold_sample_step, "impute": gold_impute, "impute_news": gold_impute_news, "estep": gold_estep}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    for name, fn in ALL.items():
        if a.only and a.only != name:
            continue
        print(name)
        fn()
