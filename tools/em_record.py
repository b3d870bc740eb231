#!/usr/bin/env python3
"""BASELINE.json configs[1] for the record: the EM loop (PyTorch M-step, HIP E-step, device scoring) on an adult-SHAPED mixed
table at its stated size -- 22 792 rows, 6 numeric + 8 categorical columns of 9/16/7/15/6/5/2/42 categories = 28 binary
digits, D = 34 (the UCI file is not in the container, so the table is synthetic: categorical columns
are functions of the numeric latent), hid_dim 1024, num_steps 50, num_trials 20 (main.py:25-28).  The reference trains up to
10 001 epochs per iteration; here --epochs (default 150) so that the record takes about a minute.  Prints one JSON line.

    python tools/em_record.py [--iters 2 --epochs 150] > profiles/<name>.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffputer_amd import em, estep, layout, prep       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=22792)
ap.add_argument("--iters", type=int, default=2)
ap.add_argument("--epochs", type=int, default=150)
ap.add_argument("--trials", type=int, default=20)
ap.add_argument("--hid", type=int, default=1024)
ap.add_argument("--mstep", default="native", choices=["native", "autograd"], help="loss + gradients from dp_mstep_fwd_bwd or from PyTorch autograd")
a = ap.parse_args()
dev = "cuda:0"
rng = np.random.default_rng(1994)
N, n_cat = a.rows, [9, 16, 7, 15, 6, 5, 2, 42]
z = rng.standard_normal((N, 3)).astype(np.float32)
num = (z @ rng.standard_normal((3, 6)).astype(np.float32) + 0.2 * rng.standard_normal((N, 6))).astype(np.float32)
score = z @ rng.standard_normal((3, len(n_cat))).astype(np.float32)
cat = np.stack([np.clip(((score[:, j] + 3.0) / 6.0 * n).astype(np.int64), 0, n - 1) for j, n in enumerate(n_cat)], axis=1)
X, bins = layout.encode_table(num, cat, n_cat)
mask0 = rng.random((N, 6 + len(n_cat))) < 0.3
mask = layout.extend_mask(mask0, 6, bins)
Xd, md = torch.from_numpy(X).to(dev), torch.from_numpy(mask).to(dev)
mean, std = prep.masked_mean_std(Xd, md)
Xn = prep.normalize(Xd, mean, std)

times = {"m_step_s": [], "e_step_s": []}
orig_m, orig_e = em.m_step, estep.e_step


def timed(fn, key):
    def wrapper(*args, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*args, **kw)
        torch.cuda.synchronize()
        times[key].append(time.perf_counter() - t0)
        return out
    return wrapper


em.m_step = timed(orig_m, "m_step_s")
estep.e_step = timed(orig_e, "e_step_s")
res = em.run_em(Xn, md, hid_dim=a.hid, max_iter=a.iters, num_trials=a.trials, num_steps=50, device=dev, max_epochs=a.epochs, seed=7,
                num_num=6, log=lambda *_: None, native_mstep=a.mstep == "native")
zero_mae, zero_rmse = prep.masked_errors(torch.zeros_like(Xn), Xn * 2, md, num_num=6)
rec = res.imputed / 2 * 2 * std + mean
got = layout.decode_bits(rec[:, 6:].cpu().numpy(), bins)
miss = mask0[:, 6:]
print(json.dumps({
    "workload": f"BASELINE configs[1]: adult-shaped mixed table {N} x {X.shape[1]} (6 numeric + 28 binary digits, synthetic), EM loop: "
                f"{a.iters} iterations x (M-step {a.epochs} epochs, batch 4096, " + ("HIP training step (bf16 hi/lo split GEMMs) + PyTorch Adam" if a.mstep == "native" else "PyTorch float32 autograd") + f"; E-step S=50 x R=20 x Heun, T={a.trials}, bf16 HIP)",
    "hid": a.hid, "mstep": a.mstep, "m_step_s": times["m_step_s"], "e_step_s": times["e_step_s"],
    "e_step_rows_per_s": [N / t for t in times["e_step_s"]],
    "mae_in": res.mae_in, "rmse_in": res.rmse_in, "zero_fill_mae_rmse": [zero_mae, zero_rmse], "train_loss": res.train_loss,
    "categorical_accuracy_on_masked_cells": float((got[miss] == cat[miss]).mean()),
    "majority_class_baseline": float(np.mean([np.bincount(cat[:, j]).max() / N for j in range(len(n_cat))])),
}))
