#!/usr/bin/env python3
"""Golden fixture for the data-prep / scoring helpers: the reference's `mean_std`
(dataset.py:161-169) and `get_eval` (dataset.py:171-206) run on the first 2 048 california training rows.

Build-container only (imports /root/reference/dataset.py; run tools/make_e2e_golden.py first: it builds the scratch
tree /tmp/dp_e2e with the CSVs, Info json and masks).  Writes tests/golden/prep_california.npz.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DIFFPUTER_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
os.chdir("/tmp/dp_e2e")
import dataset as ref_ds  # noqa: E402

(train_X, test_X, ori_train_mask, ori_test_mask, train_num, test_num, train_cat_idx, test_cat_idx, train_mask, test_mask,
 cat_bin_num) = ref_ds.load_dataset("california", 0, "MCAR", "30")
n = 2048
X, m, om = np.ascontiguousarray(train_X[:n]), np.ascontiguousarray(train_mask[:n]), np.ascontiguousarray(ori_train_mask[:n])
mean, std = ref_ds.mean_std(X, m)
Xn = ((X - mean) / std / 2).astype(np.float32)                       # main.py:58
rng = np.random.Generator(np.random.Philox(5))
pred = (Xn + 0.3 * rng.standard_normal(Xn.shape).astype(np.float32)) * 2   # an "imputation" in the x2 scale of main.py:190
mae, rmse = ref_ds.get_eval("california", pred.copy(), Xn * 2, None, X.shape[1], None, om)
path = os.path.join(REPO, "tests", "golden", "prep_california.npz")
np.savez_compressed(path, X=X, mask=m, ori_mask=om, mean=mean, std=std, Xn=Xn, pred=pred, mae=mae, rmse=rmse)
print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB): mean[0]={mean[0]:.4f} std[0]={std[0]:.4f} mae={mae:.6f} rmse={rmse:.6f}")
