#!/usr/bin/env python3
"""Production-mode (Philox) E-step on the trained H = 1024 california checkpoint, several seeds: the five statistics
tools/make_free_seeds.py records for the reference.  python tools/free_run_stats.py [--seeds 5 --dtype bf16]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffputer_amd import estep                      # noqa: E402
from diffputer_amd.evalmetrics import mae_rmse       # noqa: E402
from diffputer_amd.sampler import SamplerPlan        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "e2e_california_H1024.npz"))
sd = {k[4:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
X, m, T = g["X_test"], g["mask_test"], int(g["free_trials"])
p = SamplerPlan(sd, num_steps=int(g["S"]), n_resample=int(g["R"]), dtype=a.dtype, device="cuda:0")
rows = []
for seed in range(a.seeds):
    rec = estep.impute_trials(p, X, m, T, seed=2025 + seed).cpu().numpy()
    mae, rmse = mae_rmse(rec * 2, X * 2, m)
    err = np.abs((rec - X) * 2)[m]
    c = np.minimum(err, 3.0)
    rows.append([mae, rmse, c.mean(), np.sqrt((c ** 2).mean()), np.median(err)])
    print(seed, np.round(rows[-1], 5).tolist(), flush=True)
r = np.asarray(rows)
print("mean", np.round(r.mean(0), 5).tolist(), "sd", np.round(r.std(0, ddof=1), 5).tolist())
