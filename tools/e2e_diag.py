"""Diagnostic: error distribution of free (Philox) E-step runs vs the reference's free runs on the trained fixture."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden
from diffputer_amd import synth, estep
from diffputer_amd.sampler import SamplerPlan, n_draws
from diffputer_amd.evalmetrics import mae_rmse
g = load_golden("e2e_california_H256")
sd = {k[4:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}
X, m, T = g["X_test"], g["mask_test"], int(g["free_trials"])
def stats(name, rec):
    err = np.abs((rec - X) * 2)[m]
    print(f"{name:34s} MAE {err.mean():.4f} RMSE {np.sqrt((err**2).mean()):.4f} median {np.median(err):.4f} p90 {np.quantile(err,0.9):.4f} "
          f"p99 {np.quantile(err,0.99):.4f} max {err.max():.1f} n>3 {(err>3).sum()} trimmedMAE(clip 3) {np.minimum(err,3).mean():.4f}")
if "free_out" in g:
    for i in range(2): stats(f"reference free seed {i}", g["free_out"][i])
for dtype in ("f32", "bf16"):
    p = SamplerPlan(sd, num_steps=50, n_resample=20, dtype=dtype, device="cuda:0")
    for seed in (2024, 7):
        rec = estep.impute_trials(p, X, m, T, seed=seed).cpu().numpy()
        stats(f"gpu {dtype} philox seed {seed} T={T}", rec)
    rec = estep.impute_trials(p, X, m, 1, seed=3).cpu().numpy()
    stats(f"gpu {dtype} philox T=1", rec)
    # torch-generated Gaussian noise injected through the replay interface, T=1
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    nz = torch.randn((1, n_draws(50, 20), X.shape[0], X.shape[1]), device="cuda:0", generator=gen)
    x_obs = (X * (1.0 - m)).astype(np.float32)
    out, _ = p.impute_sum(x_obs, m, 1, noise=nz)
    stats(f"gpu {dtype} torch-noise T=1", out.cpu().numpy())
    p.close()
