"""Pin the CPU oracle (oracle/diffputer_oracle.py) to outputs of the reference itself.

The fixtures under tests/golden/ were produced by tools/make_goldens.py, which imports the
reference's model.py / diffusion_utils.py in the build container and replays a recorded noise
stream through them.  The reference has no tests or golden vectors of its own for this path
(SURVEY.md §4), so these are the pins.  Tolerance: 1e-5 (BLAS summation order, libm ulp).
"""
import numpy as np
import pytest

from conftest import golden_noise, load_golden
from diffputer_amd import synth
from oracle import diffputer_oracle as orc


def close(a, b, atol=1e-5, rtol=1e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    lim = atol + rtol * np.abs(b)
    assert a.shape == b.shape
    assert np.all(err <= lim), f"max err {err.max():.3e}, worst excess {(err - lim).max():.3e}"


def test_schedule_matches_reference():
    g = load_golden("schedule")
    for S in (5, 10, 50):
        t = orc.sigma_schedule(S)
        ref = g[f"t_S{S}"]
        assert t.dtype == np.float32 and t.shape == (S + 1,)
        # fp32 pow differs by at most an ulp or two between torch/Sleef and numpy/libm
        assert np.all(np.abs(t - ref) <= 2 * np.spacing(ref)), (t - ref)
        assert t[-1] == 0.0
    # values quoted in SURVEY.md §8 a7
    t = g["t_S50"]
    assert abs(t[0] - 79.99998474) < 1e-5 and abs(t[49] - 0.00199999963) < 1e-9


@pytest.mark.parametrize("hid", [64, 1024])
def test_time_embedding(hid):
    g = load_golden(f"embedding_H{hid}")
    sd = synth.make_weights(int(g["d_in"]), hid, int(g["wseed"]), float(g["gain"]))
    c_noise = np.log(g["sigma"]) / np.float32(4.0)
    close(orc.positional_embedding(c_noise, hid), g["posemb"], atol=2e-6)
    close(orc.time_embedding(sd, c_noise), g["emb"])


@pytest.mark.parametrize("d_in", [9, 64, 256])
def test_denoiser_single_eval(d_in):
    g = load_golden(f"denoiser_D{d_in}_H1024")
    sd = synth.make_weights(d_in, 1024, int(g["wseed"]), float(g["gain"]))
    for k, s in enumerate(g["sigma"]):
        D, F = orc.precond(sd, g["x"][k], s, return_F=True)
        close(F, g["F"][k])
        close(D, g["D"][k], atol=1e-5 * max(1.0, float(s)))


def test_sample_step():
    g = load_golden("sample_step_D9_H64_S10")
    sd = synth.make_weights(9, 64, int(g["wseed"]), float(g["gain"]))
    S = int(g["S"])
    t = orc.sigma_schedule(S)
    for i in g["levels"]:
        y = orc.sample_step(sd, S, int(i), t[i], t[i + 1], g[f"x_{i}"], g[f"eps_{i}"])
        close(y, g[f"y_{i}"], atol=1e-5 * max(1.0, float(t[i])))
    # last level (t_next = 0) returns the denoised sample itself
    i = S - 1
    th = np.float32(t[i] + np.float32(0.1) * t[i])
    x_hat = g[f"x_{i}"] + np.sqrt(th * th - t[i] * t[i]) * g[f"eps_{i}"]
    close(orc.precond(sd, x_hat, th), g[f"y_{i}"], atol=2e-6)


def _run_impute(name, noise=None):
    g = load_golden(name)
    sd = synth.make_weights(int(g["d_in"]), int(g["hid"]), int(g["wseed"]), float(g["gain"]))
    stream = g["noise"] if "noise" in g else golden_noise(g)
    trace = []
    out = orc.impute_mask(sd, g["x_obs"], g["mask"], stream, int(g["S"]), int(g["R"]), trace)
    return g, out, np.stack(trace)


@pytest.mark.parametrize("name", ["impute_D9_H64_S5_N8", "impute_D34_H256_S10_N16", "impute_D51_H256_S10_N16",
                                  "impute_D9_H64_S50_N8"])
def test_impute_mask_trajectory(name):
    g, out, trace = _run_impute(name)
    t = orc.sigma_schedule(int(g["S"]))
    assert trace.shape[0] == (int(g["S"]) - 1) * int(g["R"]) + 1
    # per-iteration tolerance scales with the noise level of that iteration
    for k, ref in zip(g["trace_idx"], g["trace"]):
        lvl = min(int(k) // int(g["R"]), int(g["S"]) - 1)
        close(trace[k], ref, atol=1e-5 * max(1.0, float(t[lvl])))
    close(out, g["out"], atol=1e-5)
    # observed cells come back bit-exactly (t_next = 0 at the last level)
    obs = ~g["mask"]
    assert np.array_equal(out[obs], g["x_obs"][obs])
    assert np.array_equal(g["out"][obs], g["x_obs"][obs])


@pytest.mark.parametrize("name", ["impute_D9_H64_S5_N8", "impute_D34_H256_S10_N16", "impute_D64_H1024_S10_N32"])
def test_torch_oracle_trajectory(name):
    """oracle/diffputer_oracle_torch.py (the restatement bench.py times as the CPU baseline: torch CPU operators, as the
    reference uses) against the same reference-generated trajectories."""
    from oracle import diffputer_oracle_torch as orct
    g = load_golden(name)
    sd = synth.make_weights(int(g["d_in"]), int(g["hid"]), int(g["wseed"]), float(g["gain"]))
    stream = g["noise"] if "noise" in g else golden_noise(g)
    trace = []
    out = orct.impute_mask(sd, g["x_obs"], g["mask"], stream, int(g["S"]), int(g["R"]), trace)
    close(out, g["out"], atol=2e-5)
    for k, ref in zip(g["trace_idx"], g["trace"]):
        close(trace[k], ref, atol=2e-5 * max(1.0, float(np.abs(ref).max())))
    obs = ~g["mask"]
    assert np.array_equal(out[obs], g["x_obs"][obs])


def test_impute_mask_noise_stored_equals_regenerated():
    g = load_golden("impute_D9_H64_S5_N8")
    assert np.array_equal(golden_noise(g), g["noise"])


@pytest.mark.parametrize("name", ["impute_D64_H1024_S10_N32"])
def test_impute_mask_full_width(name):
    g, out, trace = _run_impute(name)
    close(out, g["out"], atol=2e-5)
    for k, ref in zip(g["trace_idx"], g["trace"]):
        close(trace[k], ref, atol=2e-5 * max(1.0, float(np.abs(ref).max())))


def test_estep_driver():
    g = load_golden("estep_D9_H64_S5_T3")
    n, d, S, R, T = (int(g[k]) for k in ("n", "d_in", "S", "R", "T"))
    sd = synth.make_weights(d, int(g["hid"]), int(g["wseed"]), float(g["gain"]))
    noise = [synth.reference_noise(int(s), synth.n_draws(S, R), (n, d)) for s in g["nseeds"]]
    mean = orc.e_step(sd, g["X"], g["mask"], noise, T, S, R)
    close(mean, g["mean"])
    obs = ~g["mask"]
    close(mean[obs], g["X"][obs], atol=1e-7)


def test_draw_count():
    assert synth.n_draws(50, 20) == 3924        # SURVEY.md §0 [probed]


def test_trained_checkpoint_replay():
    """The oracle against the reference's E-step on a TRAINED california checkpoint (tools/make_e2e_golden.py):
    first 256 test rows, replayed noise; also pins the MAE/RMSE helper to the reference's get_eval numbers."""
    from diffputer_amd.evalmetrics import mae_rmse
    g = load_golden("e2e_california_H256")
    sd = {k[4:].replace("__", "."): v for k, v in g.items() if k.startswith("sd__")}
    n, d, S, R = int(g["replay_rows"]), int(g["d_in"]), int(g["S"]), int(g["R"])
    noise = synth.reference_noise(int(g["nseed"]), synth.n_draws(S, R), (n, d))
    assert float(np.asarray(noise, dtype=np.float64).sum()) == float(g["noise_checksum"])
    X, m = g["X_test"][:n], g["mask_test"][:n]
    x_obs = (X * (1.0 - m)).astype(np.float32)
    out = orc.impute_mask(sd, x_obs, m, noise, S, R)
    out = out * m + x_obs * (1.0 - m)                                   # main.py:232-233
    close(out, g["replay_out"], atol=2e-5, rtol=1e-4)
    mae, rmse = mae_rmse(out * 2, X * 2, m)
    assert abs(mae - g["replay_mae_rmse"][0]) < 1e-5 and abs(rmse - g["replay_mae_rmse"][1]) < 1e-5


def test_trained_checkpoint_replay_full_width():
    """The same at the reference's real width: tests/golden/e2e_california_H1024.npz (hid_dim = 1024, main.py:25; 600 epochs
    of the reference's M-step).  The oracle runs the first 12 of the 64 replay rows (rows are independent chains, so a row
    subset with the matching slice of the noise stream reproduces those rows) and is held to the reference's final output
    and to its per-iteration x_t (every 20th inner iteration = the state handed to each new noise level)."""
    g = load_golden("e2e_california_H1024")
    sd = {k[4:].replace("__", "."): v for k, v in g.items() if k.startswith("sd__")}
    n, d, S, R, k = int(g["replay_rows"]), int(g["d_in"]), int(g["S"]), int(g["R"]), 12
    noise = synth.reference_noise(int(g["nseed"]), synth.n_draws(S, R), (n, d))
    assert float(np.asarray(noise, dtype=np.float64).sum()) == float(g["noise_checksum"])
    X, m = g["X_test"][:k], g["mask_test"][:k]
    x_obs = (X * (1.0 - m)).astype(np.float32)
    trace = []
    out = orc.impute_mask(sd, x_obs, m, np.ascontiguousarray(noise[:, :k]), S, R, trace)
    close(out * m + x_obs * (1.0 - m), g["replay_out"][:k], atol=2e-5, rtol=1e-4)
    t = orc.sigma_schedule(S)
    for idx, ref in zip(g["trace_idx"], g["trace"]):
        lvl = min(int(idx) // R, S - 1)
        close(trace[int(idx)], ref[:k], atol=2e-5 * max(1.0, float(t[lvl + 1])), rtol=1e-4)


def test_variants_sample_and_missdiff():
    """`sample` and the MissDiff baseline's impute_mask (N = 10, no final level), both run from the reference."""
    g = load_golden("sample_D9_H64_S10")
    sd = synth.make_weights(int(g["d_in"]), int(g["hid"]), int(g["wseed"]), float(g["gain"]))
    close(orc.sample(sd, g["noise"], int(g["S"])), g["out"], atol=1e-5, rtol=1e-4)
    g = load_golden("missdiff_D9_H64_S10")
    out = orc.impute_mask(sd, g["x_obs"], g["mask"], g["noise"], int(g["S"]), int(g["R"]), final_level=False)
    close(out, g["out"], atol=1e-5, rtol=1e-4)


def test_reference_written_checkpoint():
    """tests/golden/model_ref_H64.pt is a file the REFERENCE wrote (`torch.save(model.state_dict())`,
    main.py:142, tools/make_ckpt_golden.py).  formats.load_checkpoint (weights_only=True: nothing in the file is executed)
    must read it, and the oracle driven by it must reproduce what the reference's own impute_mask made of the reloaded file."""
    import os
    from conftest import GOLDEN
    from diffputer_amd import formats
    g = load_golden("model_ref_H64_io")
    sd = formats.load_checkpoint(os.path.join(GOLDEN, "model_ref_H64.pt"))
    assert sorted(sd) == [str(k) for k in g["keys"]] and len(sd) == 14
    assert all(v.dtype.is_floating_point and v.element_size() == 4 for v in sd.values())
    sd = {k: v.numpy() for k, v in sd.items()}
    noise = golden_noise(g)
    trace = []
    out = orc.impute_mask(sd, g["x_obs"], g["mask"], noise, int(g["S"]), int(g["R"]), trace)
    close(out, g["out"], atol=1e-5)
    t = orc.sigma_schedule(int(g["S"]))
    for k, ref in enumerate(g["trace"]):
        close(trace[k], ref, atol=1e-5 * max(1.0, float(t[min(k // int(g["R"]), int(g["S"]) - 1)])))


def test_mstep_oracle_small_fixture_all_gradients():
    """oracle/diffputer_oracle_mstep.py (numpy forward + hand-written backward of Model.forward / EDMLoss / Precond /
    MLPDiffusion with per-row sigma) against the reference's loss, per-cell loss and all 14 gradients
    (tests/golden/mstep_D12_H64.npz, tools/make_mstep_golden.py)."""
    from oracle import diffputer_oracle_mstep as orm
    g = load_golden("mstep_D12_H64")
    sd = synth.make_weights(int(g["d_in"]), int(g["hid"]), int(g["wseed"]), float(g["gain"]))
    loss, percell, grads = orm.loss_and_grads(sd, g["x"], g["z_sigma"], g["z_noise"])
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * float(g["loss"])
    close(percell, g["percell"], atol=1e-5 * float(np.abs(g["percell"]).mean()), rtol=1e-5)
    for name in orm.LAYERS:
        for kind in ("weight", "bias"):
            ref = g[f"grad__{name}.{kind}"]
            assert float(np.abs(grads[f"{name}.{kind}"] - ref).max()) <= 1e-5 * float(np.abs(ref).max()), (name, kind)


def test_mstep_oracle_reference_width_and_batch():
    """The same at the reference's real width and batch size: H = 1024, D = 34, B = 4096 (tests/golden/mstep_D34_H1024.npz:
    loss, per-row loss, bias gradients in full, weight gradients as [::37, ::41] samples + Frobenius norms)."""
    from oracle import diffputer_oracle_mstep as orm
    g = load_golden("mstep_D34_H1024")
    d, hid, B = int(g["d_in"]), int(g["hid"]), int(g["B"])
    sd = synth.make_weights(d, hid, int(g["wseed"]), 1.0)
    X, mask = synth.make_table(B, d, seed=int(g["tseed"]))
    x = (X * (1.0 - mask)).astype(np.float32)
    rng = np.random.Generator(np.random.Philox(int(g["zseed"])))
    zs = rng.standard_normal(B, dtype=np.float32)
    zn = rng.standard_normal((B, d), dtype=np.float32)
    assert float(x.astype(np.float64).sum()) == float(g["x_sum"])
    loss, percell, grads = orm.loss_and_grads(sd, x, zs, zn)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * float(g["loss"])
    close(percell.mean(1), g["rowloss"], atol=1e-5 * float(g["rowloss"].mean()), rtol=2e-5)
    s0, s1 = (int(v) for v in g["stride"])
    for name in orm.LAYERS:
        ref = g[f"grad__{name}.bias"]
        assert float(np.abs(grads[f"{name}.bias"] - ref).max()) <= 2e-5 * float(np.abs(ref).max()), name
        w = grads[f"{name}.weight"]
        assert float(np.abs(w[::s0, ::s1] - g[f"gsample__{name}.weight"]).max()) <= 2e-5 * float(g[f"gmax__{name}.weight"]), name
        norm = float(np.sqrt((w.astype(np.float64) ** 2).sum()))
        assert abs(norm - float(g[f"gnorm__{name}.weight"])) <= 1e-5 * float(g[f"gnorm__{name}.weight"]), name
