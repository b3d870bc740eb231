"""End-to-end parity on a TRAINED checkpoint (BASELINE.json north_star: final imputation MAE/RMSE on held-out
masked cells within 1 % of the reference).

Fixture tests/golden/e2e_california_H256.npz (tools/make_e2e_golden.py): the reference's own M-step trained a
hid-256 denoiser on the california training table (MCAR 30 %, np.random.seed(0) masks), then its out-of-sample
E-step (main.py:212-246) ran on the test table
  * REPLAY: first 256 test rows, 1 trial, a seeded Gaussian stream replayed through torch.randn/randn_like
            -> the GPU path fed the same stream must reproduce the output per cell;
  * FREE:   all 2 044 rows, 5 trials, torch's generator under two seeds -> two MAE/RMSE pairs whose spread is
            the noise floor for comparing against our counter-based Philox noise.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from diffputer_amd import synth
from diffputer_amd.evalmetrics import mae_rmse

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def e2e():
    g = load_golden("e2e_california_H256")
    sd = {k[4:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}
    return g, sd


def _plan(sd, g, dtype):
    from diffputer_amd.sampler import SamplerPlan
    return SamplerPlan(sd, num_steps=int(g["S"]), n_resample=int(g["R"]), dtype=dtype, device=DEV)


def _replay_inputs(g):
    n, d, S, R = int(g["replay_rows"]), int(g["d_in"]), int(g["S"]), int(g["R"])
    noise = synth.reference_noise(int(g["nseed"]), synth.n_draws(S, R), (n, d))
    assert float(np.asarray(noise, dtype=np.float64).sum()) == float(g["noise_checksum"])
    X, m = g["X_test"][:n], g["mask_test"][:n]
    return X, m, (X * (1.0 - m)).astype(np.float32), noise


def test_replay_f32_per_cell(e2e):
    """float32 path, identical noise: every cell of the final imputation within 1e-3 relative (floor 1e-2 = 2 % of the
    data's std of 0.5)."""
    g, sd = e2e
    X, m, x_obs, noise = _replay_inputs(g)
    p = _plan(sd, g, "f32")
    out, _ = p.impute_sum(x_obs, m, 1, noise=noise)
    out = out.cpu().numpy()
    p.close()
    ref = g["replay_out"]
    err = np.abs(out - ref)
    lim = 1e-3 * np.maximum(np.abs(ref), 1e-2)
    assert np.all(err <= lim), (float(err.max()), float((err / lim).max()))
    assert np.array_equal(out[~m], x_obs[~m])
    mae, rmse = mae_rmse(out * 2, X * 2, m)
    assert abs(mae - g["replay_mae_rmse"][0]) <= 1e-4 * g["replay_mae_rmse"][0]
    assert abs(rmse - g["replay_mae_rmse"][1]) <= 1e-4 * g["replay_mae_rmse"][1]


@pytest.mark.parametrize("fold", [1, 0], ids=["fold", "nofold"])
def test_replay_bf16(e2e, fold):
    """bf16 GEMM operands, identical noise.  The gate is the north star's: MAE and RMSE on the masked cells within 1 % of
    the reference's on the same rows (measured on MI355X: +0.03 % / +0.03 %).  Per cell this briefly trained hid-256 net
    is less contractive than SURVEY.md §0's hid-1024 one: measured mean |d| 1.2e-3, 2-3 % of cells above 5e-3, worst
    cell 4.9e-2 (tools/e2e_diag2.py) -> bounds 3e-3 / 6 % / 2e-1."""
    from diffputer_amd import _lib
    g, sd = e2e
    X, m, x_obs, noise = _replay_inputs(g)
    lib = _lib.load()
    _lib.check(lib.dp_tune(b"fold_proj", fold), "dp_tune")
    try:
        p = _plan(sd, g, "bf16")
        out, _ = p.impute_sum(x_obs, m, 1, noise=noise)
        out = out.cpu().numpy()
        p.close()
    finally:
        _lib.check(lib.dp_tune(b"fold_proj", 1), "dp_tune")
    ref = g["replay_out"]
    mae, rmse = mae_rmse(out * 2, X * 2, m)
    assert abs(mae - g["replay_mae_rmse"][0]) <= 1e-2 * g["replay_mae_rmse"][0], (mae, g["replay_mae_rmse"])
    assert abs(rmse - g["replay_mae_rmse"][1]) <= 1e-2 * g["replay_mae_rmse"][1], (rmse, g["replay_mae_rmse"])
    e = np.abs(out - ref)[m]
    assert float(e.mean()) <= 3e-3 and float((e > 5e-3).mean()) <= 0.06 and float(e.max()) <= 2e-1, (e.mean(), (e > 5e-3).mean(), e.max())
    assert np.array_equal(out[~m], x_obs[~m])


# ---- the same at the reference's real width: hid_dim = 1024 (main.py:25), 600 epochs of the reference's M-step --------
@pytest.fixture(scope="module")
def e2e_wide():
    g = load_golden("e2e_california_H1024")
    sd = {k[4:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}
    return g, sd


@pytest.fixture(scope="module", params=[64, 256], ids=["rows64", "rows256"])
def e2e_wide_replay(request, e2e_wide):
    """The replay part of the wide fixture: the original 64 rows, or the 256-row run of tools/make_replay256.py (same
    checkpoint and held-out table, another noise stream; round 3: a larger sample for the bf16 bounds)."""
    g, sd = e2e_wide
    if request.param == 256:
        g = {**g, **load_golden("e2e_california_H1024_replay256")}
    return g, sd


def _level_sigma(g, t, k):
    """Noise level of x_t after inner iteration k (the next level's once a level's last repeat has run)."""
    S, R = int(g["S"]), int(g["R"])
    lvl = min(int(k) // R, S - 1)
    return float(t[lvl + 1]) if (int(k) % R == R - 1 or lvl == S - 1) else float(t[lvl])


def test_replay_f32_per_step_full_width(e2e_wide_replay):
    """north_star: per-step outputs within 1e-3 relative in fp32, identical noise -- on a TRAINED net at H = 1024, against
    the x_t the reference handed to each new noise level (every 20th inner iteration) and its final imputation."""
    g, sd = e2e_wide_replay
    X, m, x_obs, noise = _replay_inputs(g)
    p = _plan(sd, g, "f32")
    out, tr = p.impute_sum(x_obs, m, 1, noise=noise, trace=True)
    t = p.schedule()
    out, tr = out.cpu().numpy(), tr[:, 0].cpu().numpy()
    p.close()
    worst = 0.0
    for k, ref in zip(g["trace_idx"], g["trace"]):
        lim = 1e-3 * np.maximum(np.abs(ref), max(1e-2 * _level_sigma(g, t, k), 1e-3))
        err = np.abs(tr[int(k)] - ref)
        worst = max(worst, float((err / lim).max()))
        assert np.all(err <= lim), (int(k), float((err / lim).max()))
    ref = g["replay_out"]
    err = np.abs(out - ref)
    assert np.all(err <= 1e-3 * np.maximum(np.abs(ref), 1e-2)), float(err.max())
    assert np.array_equal(out[~m], x_obs[~m])
    mae, rmse = mae_rmse(out * 2, X * 2, m)
    assert abs(mae - g["replay_mae_rmse"][0]) <= 1e-4 * g["replay_mae_rmse"][0]
    assert abs(rmse - g["replay_mae_rmse"][1]) <= 1e-4 * g["replay_mae_rmse"][1]
    print(f"f32 H=1024 trained, {int(g['replay_rows'])} rows: worst per-step ratio to the 1e-3 bound {worst:.3f}, final max |d| {float(err.max()):.2e}")


@pytest.mark.parametrize("fold", [1, 0], ids=["fold", "nofold"])
def test_replay_bf16_full_width(e2e_wide_replay, fold):
    """bf16 GEMM operands (float32 accumulate, float32 sampler state) on the trained H = 1024 net, identical noise.
    Bounds stated from measurement on MI355X over 981 inner iterations (tools/bf16_trained_diag.py [256];
    profiles/r2_diag_bf16_trained_h1024.txt = 64 rows, profiles/r3_diag_bf16_trained_h1024_256rows.txt = 256 rows):
      typical chain        per step mean |d| 3e-5 (sigma > 1) rising to 2.2e-4 (sigma < 0.01); 99.9th percentile <= 1e-3 down
                           to sigma 0.7, 2.4e-2 at the end; final imputation mean 7.4e-4 (fold) / 5.4e-4
      sensitive chains     from sigma ~ 0.35 on a few rows amplify the rounding difference: 1 of 64 rows and 11 of 256 end
                           with a cell above SURVEY.md section 8d's 5e-3; per step the 99.9th percentile stays <= 2.4e-2
                           (fold) / 1.2e-2, the worst cell in flight is 3.4e-2 (fold) / 2.4e-1 (two GEMMs: ONE chain's
                           excursion at sigma 0.09, back to 2e-2 twenty iterations later -- the sampler is contractive);
                           worst final cell 2.5e-2 / 1.7e-2; 99th percentile of the final cells 1.4e-2 / 6.1e-3
      -> asserted: per step mean <= 4e-4, 99.9th percentile <= 4e-2, max <= 5e-1 (a bound on excursions, not on typical
         cells); final mean <= 1e-3, 99th percentile <= 2e-2, max <= 5e-2, at most 8 % of the rows with a cell above 5e-3
      MAE / RMSE on the masked cells   64 rows -0.04 % / +0.003 %; 256 rows +0.02 % / +0.06 % (fold), +0.03 % / +0.04 %
                                       -> asserted 1 %: the north star's gate, and the one that matters
    (north_star: "looser bound stated for bf16".)"""
    from diffputer_amd import _lib
    g, sd = e2e_wide_replay
    X, m, x_obs, noise = _replay_inputs(g)
    lib = _lib.load()
    _lib.check(lib.dp_tune(b"fold_proj", fold), "dp_tune")
    try:
        p = _plan(sd, g, "bf16")
        out, tr = p.impute_sum(x_obs, m, 1, noise=noise, trace=True)
        out, tr = out.cpu().numpy(), tr[:, 0].cpu().numpy()
        p.close()
    finally:
        _lib.check(lib.dp_tune(b"fold_proj", 1), "dp_tune")
    worst_max = worst_mean = 0.0
    for k, ref in zip(g["trace_idx"], g["trace"]):
        e = np.abs(tr[int(k)] - ref)
        worst_max, worst_mean = max(worst_max, float(e.max())), max(worst_mean, float(e.mean()))
        assert float(e.max()) <= 5e-1 and float(e.mean()) <= 4e-4 and float(np.quantile(e, 0.999)) <= 4e-2, (int(k), float(e.max()), float(e.mean()))
    ref = g["replay_out"]
    rows_off = float((np.abs(out - ref).max(1) > 5e-3).mean())
    e = np.abs(out - ref)[m]
    mae, rmse = mae_rmse(out * 2, X * 2, m)
    print(f"bf16 H=1024 trained, {int(g['replay_rows'])} rows ({'fold' if fold else 'nofold'}): per step max |d| {worst_max:.2e} mean {worst_mean:.2e}; final max "
          f"{float(e.max()):.2e} p99 {float(np.quantile(e, 0.99)):.2e} mean {float(e.mean()):.2e}; MAE {mae:.5f} vs "
          f"{float(g['replay_mae_rmse'][0]):.5f}, RMSE {rmse:.5f} vs {float(g['replay_mae_rmse'][1]):.5f}")
    assert float(np.quantile(e, 0.99)) <= 2e-2 and float(e.mean()) <= 1e-3 and float(e.max()) <= 5e-2 and rows_off <= 0.08, rows_off
    assert abs(mae - g["replay_mae_rmse"][0]) <= 1e-2 * g["replay_mae_rmse"][0]
    assert abs(rmse - g["replay_mae_rmse"][1]) <= 1e-2 * g["replay_mae_rmse"][1]
    assert np.array_equal(out[~m], x_obs[~m])


def _five_stats(rec, X, m):
    """[MAE, RMSE, MAE of errors clipped at 3, RMSE of the same, median error] over the masked cells, x2 scale (main.py:240-241):
    the columns of tests/golden/e2e_california_H1024_seeds.npz (tools/make_free_seeds.py)."""
    mae, rmse = mae_rmse(rec * 2, X * 2, m)
    err = np.abs((rec - X) * 2)[m]
    c = np.minimum(err, 3.0)
    return np.asarray([mae, rmse, c.mean(), np.sqrt((c ** 2).mean()), np.median(err)])


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_free_run_metrics_full_width(e2e_wide, dtype):
    """Production mode (Philox noise) on all 2 044 held-out rows, 5 trials, H = 1024, against TEN free runs of the reference
    (torch's generator under seeds 0..9, tools/make_free_seeds.py: per seed MAE, RMSE, their clipped forms and the median).
    MAE / RMSE of a free run are statistics of 5 518 cells x 5 trials with a heavy tail (a rare chain that wanders off moves
    RMSE by 5-25 % in the reference and here alike), so the comparison is between MEANS over seeds: ours over three Philox
    seeds, the reference's over its ten.
      * MAE, RMSE, median: |ours - ref| <= max(1 % of ref, 2 sd of the reference's seeds)
      * clipped MAE and clipped RMSE (errors capped at 3 = 6 sigma_data: the tail taken out): within 1 % of the reference,
        the north star's bound, with no allowance."""
    from diffputer_amd import estep
    g, sd = e2e_wide
    ref = load_golden("e2e_california_H1024_seeds")
    assert len(ref["seeds"]) >= 8 and int(ref["trials"]) == int(g["free_trials"])
    X, m, T = g["X_test"], g["mask_test"], int(g["free_trials"])
    p = _plan(sd, g, dtype)
    ours = np.stack([_five_stats(estep.impute_trials(p, np.asfortranarray(X), m, T, seed=2025 + k).cpu().numpy(), X, m) for k in range(3)])
    p.close()
    mean, sdev = ref["stats"].mean(0), ref["stats"].std(0, ddof=1)
    got = ours.mean(0)
    print(f"free run H=1024 {dtype}: ours (3 Philox seeds) {np.round(got, 4).tolist()} vs reference ({len(ref['seeds'])} seeds) "
          f"{np.round(mean, 4).tolist()} sd {np.round(sdev, 4).tolist()}")
    for j in (0, 1, 4):
        assert abs(got[j] - mean[j]) <= max(1e-2 * mean[j], 2 * sdev[j]), (ref["columns"][j], got[j], mean[j], sdev[j])
    for j in (2, 3):
        assert abs(got[j] - mean[j]) <= 1e-2 * mean[j], (ref["columns"][j], got[j], mean[j])
    zf = g["zero_fill_mae_rmse"]
    assert got[0] < 0.8 * zf[0] and got[1] < 0.9 * zf[1], "a trained denoiser must clearly beat zero-fill"


def _robust(rec, X, m, clip=3.0):
    """MAE, RMSE of the masked-cell errors clipped at `clip` (6 sigma_data in the x2 scale), and their median."""
    err = np.minimum(np.abs((rec - X) * 2)[m], clip)
    return np.asarray([err.mean(), np.sqrt((err ** 2).mean()), np.median(err)])


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_free_run_error_distribution(e2e, dtype):
    """Production mode (Philox noise, all 2 044 test rows, 5 trials, column-major input as pandas hands it over) against
    the reference's two free runs.  With this briefly trained net a few chains DIVERGE in the reference itself (its two
    seeds give RMSE 0.75 and 2.10, worst cell 13 vs 103; the HIP path shows the same rare cells), so plain MAE/RMSE of a
    free run is a property of which chains blew up.  Compared instead: MAE and RMSE of errors clipped at 3 and the
    median error -- each within 1 % of the reference's two-seed mean plus the reference's own two-seed spread."""
    from diffputer_amd import estep
    g, sd = e2e
    X, m, T = g["X_test"], g["mask_test"], int(g["free_trials"])
    p = _plan(sd, g, dtype)
    rec = estep.impute_trials(p, np.asfortranarray(X), m, T, seed=2024).cpu().numpy()
    p.close()
    assert np.allclose(rec[~m], X[~m], rtol=1e-6, atol=0)                     # (5 x) / 5 rounds, as stack().mean(0) does
    ours = _robust(rec, X, m)
    ref = np.stack([_robust(g["free_out"][i], X, m) for i in range(2)])
    mid, spread = ref.mean(0), np.abs(ref[0] - ref[1])
    assert np.all(np.abs(ours - mid) <= 1e-2 * mid + spread), (ours, ref)
    zf = g["zero_fill_mae_rmse"]
    assert ours[0] < 0.8 * zf[0] and ours[1] < 0.8 * zf[1], "a trained denoiser must clearly beat zero-fill"


def test_em_loop_and_wire_formats(tmp_path):
    """two EM iterations (PyTorch M-step, HIP E-step) on a small correlated table; the files written
    are the reference's (model.pt state dict, iter_k.npy in the x2 scale) and the imputation beats zero-fill."""
    from diffputer_amd import em, formats, synth
    from diffputer_amd.sampler import SamplerPlan
    X, mask = synth.make_table(3000, 8, seed=31, latent=2)
    Xt, mt = synth.make_table(500, 8, seed=32, latent=2)
    Xt = Xt  # a different draw of the same generator family is not the same distribution: score in-sample only
    res = em.run_em(X, mask, hid_dim=128, max_iter=2, num_trials=2, num_steps=10, device=DEV, ckpt_dir=str(tmp_path),
                    max_epochs=150, seed=3, log=lambda *_: None)
    zero_mae, zero_rmse = mae_rmse(np.zeros_like(X), X * 2, mask)
    # 150 epochs from a fresh init per iteration (the reference trains up to 10 001): iteration 0 must clearly beat
    # zero-fill, later ones at least match it
    assert res.mae_in[0] < 0.8 * zero_mae and res.rmse_in[0] < 0.8 * zero_rmse, (res.mae_in, zero_mae)
    assert all(m < 1.05 * zero_mae for m in res.mae_in) and len(res.mae_in) == 2, (res.mae_in, zero_mae)
    sd = formats.load_checkpoint(str(tmp_path / "1" / "model.pt"))
    assert len(sd) == 14 and sd["denoise_fn_D.denoise_fn_F.mlp.2.weight"].shape == (256, 256)
    p = SamplerPlan(sd, num_steps=10, n_resample=20, dtype="bf16", device=DEV)      # a reloaded checkpoint drives the sampler
    p.close()
    it2 = np.load(tmp_path / "iter_2.npy")
    assert it2.dtype == np.float32 and it2.shape == X.shape
    assert np.allclose(it2[~mask], (X * 2)[~mask], rtol=1e-5, atol=1e-6)           # observed cells pass through, x2 scale
    assert np.allclose(formats.load_iteration(str(tmp_path / "iter_2.npy")), it2 / 2)
    assert np.allclose(res.imputed.cpu().numpy(), it2)


def test_em_iteration_on_a_mixed_table():
    """BASELINE configs[1] in miniature: an adult-shaped MIXED table (6 numeric columns + 8 categorical ones of
    9/16/7/15/6/5/2/42 categories = 28 binary digits, D = 34; layout.py = dataset.py:70-157), normalised on the device
    with the observed-cell statistics (prep = dataset.mean_std + main.py:55-64), one EM iteration.  The categorical columns
    are functions of the numeric latent, so a working imputer recovers most masked categories from the observed cells."""
    from diffputer_amd import em, layout, prep
    rng = np.random.default_rng(41)
    N, n_cat = 4000, [9, 16, 7, 15, 6, 5, 2, 42]
    z = rng.standard_normal((N, 2)).astype(np.float32)
    num = (z @ rng.standard_normal((2, 6)).astype(np.float32) + 0.1 * rng.standard_normal((N, 6))).astype(np.float32)
    score = z @ rng.standard_normal((2, len(n_cat))).astype(np.float32)
    cat = np.stack([np.clip(((score[:, j] + 2.5) / 5.0 * n).astype(np.int64), 0, n - 1) for j, n in enumerate(n_cat)], axis=1)
    X, bins = layout.encode_table(num, cat, n_cat)
    assert X.shape == (N, 34)
    mask0 = rng.random((N, 6 + len(n_cat))) < 0.3                       # over the ORIGINAL columns
    mask = layout.extend_mask(mask0, 6, bins)
    Xd, md = torch.from_numpy(X).to(DEV), torch.from_numpy(mask).to(DEV)
    mean, std = prep.masked_mean_std(Xd, md)
    Xn = prep.normalize(Xd, mean, std)
    res = em.run_em(Xn, md, hid_dim=128, max_iter=1, num_trials=4, num_steps=10, device=DEV, max_epochs=200, seed=5,
                    num_num=6, log=lambda *_: None)
    zero_mae, _ = prep.masked_errors(torch.zeros_like(Xn), Xn * 2, md, num_num=6)
    assert res.mae_in[0] < 0.8 * zero_mae, (res.mae_in, zero_mae)      # numeric columns: get_eval's score
    rec = res.imputed / 2 * 2 * std + mean                              # main.py:196-199: back to the data scale
    got = layout.decode_bits(rec[:, 6:].cpu().numpy(), bins)
    miss = mask0[:, 6:]
    acc = float((got[miss] == cat[miss]).mean())
    base = float(np.mean([np.bincount(cat[:, j]).max() / N for j in range(len(n_cat))]))
    assert np.array_equal(got[~miss], cat[~miss])                       # observed categories pass through exactly
    assert acc > base + 0.1, (acc, base)                                # masked ones beat the majority-class guess
