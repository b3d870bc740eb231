import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden
from diffputer_amd import synth, estep, _lib
from diffputer_amd.sampler import SamplerPlan, n_draws
from diffputer_amd.evalmetrics import mae_rmse
g = load_golden("e2e_california_H256")
sd = {k[4:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}
lib = _lib.load()
n, d = int(g["replay_rows"]), int(g["d_in"])
noise = synth.reference_noise(int(g["nseed"]), n_draws(50, 20), (n, d))
X, m = g["X_test"][:n], g["mask_test"][:n]
x_obs = (X * (1.0 - m)).astype(np.float32)
ref = g["replay_out"]
for dtype, fold in (("f32", 1), ("bf16", 1), ("bf16", 0)):
    lib.dp_tune(b"fold_proj", fold)
    p = SamplerPlan(sd, num_steps=50, n_resample=20, dtype=dtype, device="cuda:0")
    out = p.impute_sum(x_obs, m, 1, noise=noise)[0].cpu().numpy(); p.close()
    e = np.abs(out - ref)[m]
    mae, rmse = mae_rmse(out * 2, X * 2, m)
    print(f"replay {dtype} fold={fold}: |d| max {e.max():.2e} mean {e.mean():.2e} p99 {np.quantile(e, .99):.2e} p999 {np.quantile(e, .999):.2e} n>5e-3 {(e > 5e-3).sum()}/{e.size}"
          f" | MAE {mae:.5f} vs {g['replay_mae_rmse'][0]:.5f} ({(mae / g['replay_mae_rmse'][0] - 1) * 100:+.3f}%) RMSE {rmse:.5f} vs {g['replay_mae_rmse'][1]:.5f} ({(rmse / g['replay_mae_rmse'][1] - 1) * 100:+.3f}%)")
lib.dp_tune(b"fold_proj", 1)
X, m, T = g["X_test"], g["mask_test"], int(g["free_trials"])
def stats(name, rec):
    err = np.abs((rec - X) * 2)[m]
    print(f"{name:34s} MAE {err.mean():.4f} RMSE {np.sqrt((err**2).mean()):.4f} median {np.median(err):.4f} p90 {np.quantile(err,0.9):.4f} max {err.max():.1f} n>3 {(err>3).sum()} trimmedMAE {np.minimum(err,3).mean():.4f} trimmedRMSE {np.sqrt((np.minimum(err,3)**2).mean()):.4f}")
for i in range(2): stats(f"reference free seed {i}", g["free_out"][i])
for dtype in ("f32", "bf16"):
    p = SamplerPlan(sd, num_steps=50, n_resample=20, dtype=dtype, device="cuda:0")
    for seed in (2024, 7, 8, 9):
        stats(f"gpu {dtype} philox seed {seed} T={T}", estep.impute_trials(p, np.asfortranarray(X), m, T, seed=seed).cpu().numpy())
    p.close()
