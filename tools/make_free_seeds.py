#!/usr/bin/env python3
"""More free-run seeds of the REFERENCE's out-of-sample E-step on the trained H = 1024 california checkpoint.

Build-container only (imports /root/reference/{model,diffusion_utils}.py; nothing of the reference is copied or shipped).
tests/golden/e2e_california_H1024.npz (tools/make_e2e_golden.py) holds the checkpoint, the normalised held-out table, its
mask and TWO free runs (torch.manual_seed 0 and 1).  Two seeds say little about the sampling noise of MAE / RMSE, so this
script runs main.py:212-246 again under torch.manual_seed(2 .. 1 + --seeds) -- all 2 044 rows, 5 trials, torch's generator
-- and stores, per seed (seeds 0 and 1 recomputed from the stored imputations), only a few floats:
    MAE, RMSE (get_eval, dataset.py:199-201), MAE / RMSE of errors clipped at 3, median error
-> tests/golden/e2e_california_H1024_seeds.npz.  No new weights, no imputations.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_free_seeds.py [--seeds 8 --threads 6]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DIFFPUTER_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=8, help="new seeds, numbered from 2")
ap.add_argument("--threads", type=int, default=6)
ap.add_argument("--partial", default="/tmp/dp_free_seeds.json", help="per-seed results are appended here as they finish")
a = ap.parse_args()
sys.argv = sys.argv[:1]
torch.set_num_threads(a.threads)

import diffusion_utils as ref_du              # noqa: E402  (the reference)
from model import MLPDiffusion, Model         # noqa: E402

from diffputer_amd.evalmetrics import mae_rmse  # noqa: E402

g = np.load(os.path.join(REPO, "tests", "golden", "e2e_california_H1024.npz"))
sd = {k[4:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("sd__")}
hid, d_in, S, T = int(g["hid"]), int(g["d_in"]), int(g["S"]), int(g["free_trials"])
X_test, mask_test = torch.from_numpy(g["X_test"]), torch.from_numpy(g["mask_test"])
model = Model(denoise_fn=MLPDiffusion(d_in, hid), hid_dim=d_in)
model.load_state_dict(sd)
model.eval()
net = model.denoise_fn_D


def e_step(x_full, m_full, trials):           # main.py:212-246
    recs = []
    for _ in range(trials):
        impute_X = (1.0 - m_full.float()) * x_full
        with torch.no_grad():
            rec = ref_du.impute_mask(net, impute_X, m_full, impute_X.shape[0], impute_X.shape[1], S, "cpu")
        mi = m_full.to(torch.float)
        recs.append(rec * mi + impute_X * (1 - mi))
    return torch.stack(recs, dim=0).mean(0)


def stats(rec):
    """[MAE, RMSE, clipped MAE, clipped RMSE, median] of the masked-cell errors in the x2 scale (main.py:240-241)."""
    X, m = g["X_test"], g["mask_test"]
    mae, rmse = mae_rmse(rec * 2, X * 2, m)
    err = np.abs((rec - X) * 2)[m]
    c = np.minimum(err, 3.0)
    return [float(mae), float(rmse), float(c.mean()), float(np.sqrt((c ** 2).mean())), float(np.median(err))]


rows = {0: stats(g["free_out"][0]), 1: stats(g["free_out"][1])}
assert np.allclose(rows[0][:2], g["free_mae_rmse"][0]) and np.allclose(rows[1][:2], g["free_mae_rmse"][1])
if os.path.exists(a.partial):
    rows.update({int(k): v for k, v in json.load(open(a.partial)).items()})
for seed in range(2, 2 + a.seeds):
    if seed in rows:
        continue
    torch.manual_seed(seed)
    t0 = time.time()
    rows[seed] = stats(e_step(X_test, mask_test, T).numpy())
    print(f"seed {seed}: {time.time() - t0:.0f}s  {rows[seed]}", flush=True)
    json.dump({str(k): v for k, v in rows.items()}, open(a.partial, "w"))

seeds = np.asarray(sorted(rows))
table = np.asarray([rows[int(s)] for s in seeds])
path = os.path.join(REPO, "tests", "golden", "e2e_california_H1024_seeds.npz")
np.savez(path, seeds=seeds, stats=table, columns=np.asarray(["mae", "rmse", "mae_clip3", "rmse_clip3", "median"]), trials=T)
print(f"wrote {path}: {len(seeds)} seeds; mean {table.mean(0)}, sd {table.std(0, ddof=1)}")
