#!/usr/bin/env python3
"""End-to-end golden fixture: a TRAINED california checkpoint and what the reference's E-step makes of it.

Build-container only (imports /root/reference/{dataset,generate_mask,model,diffusion_utils}.py; nothing of the
reference is copied into the repo or shipped to the GPU box).  Steps, all on CPU:
  1. scratch tree /tmp/dp_e2e/datasets/{california,Info}: the CSVs the reference ships under
     baselines/MissingDataOT/datasets/california and datasets/Info/california.json;
  2. `generate_mask.generate_mask('california', 'MCAR', 0.3, 1)` under np.random.seed(0) (the reference never
     seeds, generate_mask.py:513-514), `dataset.load_dataset`, `dataset.mean_std`, the /2 scaling of main.py:55-64;
  3. the reference's M-step loop (main.py:92-152: Adam 1e-4, batch 4096, ReduceLROnPlateau(0.9, 50), best-loss
     checkpoint) on the zero-filled training table for --epochs epochs with hid_dim = --hid;
  4. the reference's out-of-sample E-step (main.py:212-246) on the test table:
       a. REPLAY: the first --replay-rows rows, 1 trial, Gaussian draws replayed from a seeded stream
          (tools/make_goldens.NoiseReplay) -> per-cell output for the GPU path fed the same stream;
       b. FREE: all test rows, --trials trials, torch's own generator under two different manual seeds
          -> the imputations and MAE / RMSE of each.  The two differ a lot in RMSE: with a briefly trained net a few
          chains diverge in the REFERENCE itself, so the test compares robust statistics of the error distribution.
Writes tests/golden/e2e_california_H<hid>.npz (checkpoint, normalised test table, mask, outputs, metrics).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_e2e_golden.py [--hid 256 --epochs 400]
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DIFFPUTER_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

ap = argparse.ArgumentParser()
ap.add_argument("--hid", type=int, default=256)
ap.add_argument("--epochs", type=int, default=400)
ap.add_argument("--replay-rows", type=int, default=256)
ap.add_argument("--trials", type=int, default=5)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--trace-every", type=int, default=0, help="replay run: also store x_t after every k-th inner iteration (0 = off)")
a = ap.parse_args()
sys.argv = sys.argv[:1]                       # the modules below must not see our flags
torch.set_num_threads(a.threads)

SCRATCH = "/tmp/dp_e2e"
src = os.path.join(REF, "baselines", "MissingDataOT", "datasets", "california")
os.makedirs(os.path.join(SCRATCH, "datasets", "california"), exist_ok=True)
os.makedirs(os.path.join(SCRATCH, "datasets", "Info"), exist_ok=True)
for f in ("data.csv", "train.csv", "test.csv"):
    shutil.copy(os.path.join(src, f), os.path.join(SCRATCH, "datasets", "california", f))
shutil.copy(os.path.join(REF, "datasets", "Info", "california.json"), os.path.join(SCRATCH, "datasets", "Info", "california.json"))
os.chdir(SCRATCH)

import dataset as ref_ds                      # noqa: E402  (the reference)
import diffusion_utils as ref_du              # noqa: E402
import generate_mask as ref_gm                # noqa: E402
from model import MLPDiffusion, Model         # noqa: E402

from make_goldens import NoiseReplay, checksum  # noqa: E402
from diffputer_amd import synth               # noqa: E402
from diffputer_amd.evalmetrics import mae_rmse  # noqa: E402

# ---- 2. masks + tables, exactly as main.py builds them ------------------------------------------------------
np.random.seed(0)
ref_gm.generate_mask(dataname="california", mask_type="MCAR", p=0.3, mask_num=1)
(train_X, test_X, ori_train_mask, ori_test_mask, train_num, test_num, train_cat_idx, test_cat_idx, train_mask, test_mask,
 cat_bin_num) = ref_ds.load_dataset("california", 0, "MCAR", "30")
mean_X, std_X = ref_ds.mean_std(train_X, train_mask)
in_dim = train_X.shape[1]
X = torch.tensor((train_X - mean_X) / std_X / 2)
X_test = torch.tensor((test_X - mean_X) / std_X / 2)
mask_train, mask_test = torch.tensor(train_mask), torch.tensor(test_mask)
print(f"train {tuple(X.shape)} test {tuple(X_test.shape)} miss rate {float(mask_train.float().mean()):.4f}")

# ---- 3. M-step (main.py:84-152), iteration 0 ------------------------------------------------------------------
torch.manual_seed(1234)
train_data = ((1.0 - mask_train.float()) * X).numpy()
loader = torch.utils.data.DataLoader(train_data, batch_size=4096, shuffle=True, num_workers=0)
model = Model(denoise_fn=MLPDiffusion(in_dim, a.hid), hid_dim=in_dim)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0)
sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.9, patience=50)
model.train()
best, best_sd, t0 = float("inf"), None, time.time()
CKPT = os.path.join(SCRATCH, f"model_H{a.hid}_E{a.epochs}.pt")           # scratch cache of this script's own training
if os.path.exists(CKPT):
    best_sd = torch.load(CKPT, weights_only=True)
    best = float(best_sd.pop("__best_loss"))
    print(f"loaded {CKPT} (loss {best:.4f})")
for epoch in range(0 if best_sd is None else a.epochs, a.epochs):
    tot, cnt = 0.0, 0
    for batch in loader:
        inputs = batch.float()
        loss = model(inputs).mean()
        tot += loss.item() * len(inputs)
        cnt += len(inputs)
        opt.zero_grad()
        loss.backward()
        opt.step()
    cur = tot / cnt
    sched.step(cur)
    if cur < best:
        best, best_sd = cur, {k: v.detach().clone() for k, v in model.state_dict().items()}
    if epoch % 25 == 0 or epoch == a.epochs - 1:
        print(f"epoch {epoch}: loss {cur:.4f} (best {best:.4f}) {time.time() - t0:.0f}s", flush=True)

if not os.path.exists(CKPT):
    torch.save({**best_sd, "__best_loss": torch.tensor(best)}, CKPT)

# ---- 4. out-of-sample E-step (main.py:212-246) -------------------------------------------------------------------
model = Model(denoise_fn=MLPDiffusion(in_dim, a.hid), hid_dim=in_dim)
model.load_state_dict(best_sd)
model.eval()
net = model.denoise_fn_D
S, R = 50, 20


def e_step(x_full, m_full, trials, replay=None, starts=None):
    """main.py:212-246.  `starts` (a list) receives the x_t handed to every sample_step call (the state BEFORE each inner
    iteration), captured by wrapping the reference's sample_step for the duration of the call."""
    recs = []
    for trial in range(trials):
        impute_X = (1.0 - m_full.float()) * x_full
        with torch.no_grad():
            if replay is not None:
                saved = ref_du.sample_step

                def spy(net_, num_steps, i, t_cur, t_next, x_next):
                    if starts is not None:
                        starts.append(x_next.numpy().copy())
                    return saved(net_, num_steps, i, t_cur, t_next, x_next)

                ref_du.sample_step = spy
                try:
                    with NoiseReplay(replay[trial]) as nr:
                        rec = ref_du.impute_mask(net, impute_X, m_full, impute_X.shape[0], impute_X.shape[1], S, "cpu")
                finally:
                    ref_du.sample_step = saved
                assert nr.k == replay.shape[1]
            else:
                rec = ref_du.impute_mask(net, impute_X, m_full, impute_X.shape[0], impute_X.shape[1], S, "cpu")
        mi = m_full.to(torch.float)
        recs.append(rec * mi + impute_X * (1 - mi))
    return torch.stack(recs, dim=0).mean(0)


def metrics(rec, x_true, m):
    # get_eval (dataset.py:171-206) for a numeric-only table: pred and truth in the x2 scale of main.py:240-241
    return mae_rmse(rec.numpy() * 2, x_true.numpy() * 2, m.numpy())


nr_rows = a.replay_rows
nseed = 777
stream = synth.reference_noise(nseed, synth.n_draws(S, R), (nr_rows, in_dim))
t0 = time.time()
starts = [] if a.trace_every else None
rep = e_step(X_test[:nr_rows], mask_test[:nr_rows], 1, replay=stream[None], starts=starts)
trace_kw = {}
if a.trace_every:
    # x_t after inner iteration k = the x_t handed to iteration k+1; the last one is impute_mask's return value, which for
    # finite values equals the blended `rep` (main.py:182-183 is then a no-op)
    full = np.stack(starts[1:] + [rep.numpy()])
    idx = np.arange(full.shape[0])
    keep = (idx % a.trace_every == a.trace_every - 1) | (idx == full.shape[0] - 1)
    trace_kw = dict(trace=full[keep], trace_idx=idx[keep])
print(f"replay E-step: {time.time() - t0:.0f}s  MAE/RMSE {metrics(rep, X_test[:nr_rows], mask_test[:nr_rows])}")
free, free_out = [], []
for seed in (0, 1):
    torch.manual_seed(seed)
    t0 = time.time()
    r = e_step(X_test, mask_test, a.trials)
    free.append(metrics(r, X_test, mask_test))
    free_out.append(r.numpy())
    print(f"free E-step seed {seed}: {time.time() - t0:.0f}s  MAE/RMSE {free[-1]}", flush=True)
zero = metrics((1.0 - mask_test.float()) * X_test, X_test, mask_test)
print("zero-fill MAE/RMSE", zero)

out = {k.replace(".", "__"): v.numpy() for k, v in best_sd.items()}
path = os.path.join(REPO, "tests", "golden", f"e2e_california_H{a.hid}.npz")
np.savez_compressed(
    path, **{"sd__" + k: v for k, v in out.items()}, hid=a.hid, d_in=in_dim, S=S, R=R, epochs=a.epochs, train_loss=best,
    X_test=X_test.numpy(), mask_test=mask_test.numpy(), replay_rows=nr_rows, nseed=nseed, noise_checksum=checksum(stream),
    replay_out=rep.numpy(), replay_mae_rmse=np.asarray(metrics(rep, X_test[:nr_rows], mask_test[:nr_rows])),
    free_trials=a.trials, free_mae_rmse=np.asarray(free), free_out=np.asarray(free_out), zero_fill_mae_rmse=np.asarray(zero), **trace_kw)
print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
