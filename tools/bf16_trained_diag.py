import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_golden
from diffputer_amd import synth, _lib
from diffputer_amd.sampler import SamplerPlan
from diffputer_amd.evalmetrics import mae_rmse
g = load_golden("e2e_california_H1024")
if len(sys.argv) > 1 and sys.argv[1] == "256":           # the wider replay of tools/make_replay256.py
    g = {**g, **load_golden("e2e_california_H1024_replay256")}
sd = {k[4:].replace("__", "."): torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}
n, d, S, R = int(g["replay_rows"]), int(g["d_in"]), int(g["S"]), int(g["R"])
noise = synth.reference_noise(int(g["nseed"]), synth.n_draws(S, R), (n, d))
X, m = g["X_test"][:n], g["mask_test"][:n]
x_obs = (X * (1.0 - m)).astype(np.float32)
lib = _lib.load()
for fold in (1, 0):
    lib.dp_tune(b"fold_proj", fold)
    p = SamplerPlan(sd, num_steps=S, n_resample=R, dtype="bf16", device="cuda:0")
    out, tr = p.impute_sum(x_obs, m, 1, noise=noise, trace=True)
    t = p.schedule(); out = out.cpu().numpy(); tr = tr[:, 0].cpu().numpy(); p.close()
    rows = []
    for k, ref in zip(g["trace_idx"], g["trace"]):
        lvl = min(int(k) // R, S - 1)
        sig = float(t[lvl + 1]) if (int(k) % R == R - 1 or lvl == S - 1) else float(t[lvl])
        e = np.abs(tr[int(k)] - ref)
        rows.append((int(k), sig, float(e.max()), float(e.mean()), float((e / np.maximum(np.abs(ref), 1e-2*max(sig,0.1))).max()), float(np.abs(ref).max()),
                     float(np.quantile(e, 0.999)), int((e.max(1) > 5e-3).sum())))
    print("fold", fold)
    for r in rows: print("  k %4d sigma %8.4f  max|d| %.3e mean|d| %.3e maxrel %.3e  max|ref| %.2f  p99.9 %.3e  rows with a cell > 5e-3: %d" % r)
    e = np.abs(out - g["replay_out"])[m]
    print("  final max %.3e mean %.3e p99 %.3e p99.9 %.3e frac>5e-3 %.4f" % (e.max(), e.mean(), np.quantile(e, .99), np.quantile(e, .999), (e > 5e-3).mean()), mae_rmse(out*2, X*2, m), g["replay_mae_rmse"])
    rowmax = np.abs(out - g["replay_out"]).max(1)
    print("  rows by final max|d|:", np.round(np.sort(rowmax)[-6:], 4).tolist(), "of", len(rowmax))
