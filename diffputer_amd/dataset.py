"""Drop-in for the reference's `dataset` module: `load_dataset`, `mean_std`, `get_eval` (dataset.py:10-206) with the
reference's signatures and return values, so `main.py:9` (`from dataset import load_dataset, get_eval, mean_std`) can point
here.  Host code around the sampler: it produces the `[N, D]` table layout the E-step consumes and
scores its output.  The array work is in `layout` (mixed-table encoding) and `evalmetrics`; this module adds the file
reading of `load_dataset`: `<data_dir>/<name>/{data,train,test}.csv`, `<data_dir>/Info/<name>.json`, and the boolean masks
`<data_dir>/<name>/masks/rate<ratio>/<mask_type>/{train,test}_mask_<idx>.npy` over the ORIGINAL columns (True = missing).

Differences that do not change results: the category -> index / binary maps are rebuilt from `data.csv` on every call
instead of being cached in `<column>_map_{bin,idx}.json` files next to the data (dataset.py:72-91 writes them once; the
maps depend only on `data.csv`), and nothing is written to the data directory.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import layout
from .evalmetrics import mae_rmse
from .evalmetrics import mean_std as _mean_std

DATA_DIR = "datasets"        # dataset.py:8 (relative to the working directory, like the reference)


def load_dataset(dataname, idx=0, mask_type="MCAR", ratio="30", data_dir: str | None = None):
    """dataset.py:10-159.  Returns the reference's 11-tuple: train_X, test_X, train_mask, test_mask (ORIGINAL columns),
    train_num, test_num, train_cat_idx, test_cat_idx, extend_train_mask, extend_test_mask, cat_bin_num -- the four
    categorical entries are None for a numeric-only table (dataset.py:53-56, 61-66)."""
    import pandas as pd
    root = DATA_DIR if data_dir is None else data_dir
    base = os.path.join(root, dataname)
    with open(os.path.join(root, "Info", f"{dataname}.json")) as f:
        info = json.load(f)
    num_idx, cat_idx = list(info["num_col_idx"]), list(info["cat_col_idx"])
    data_df, train_df, test_df = (pd.read_csv(os.path.join(base, f"{s}.csv")) for s in ("data", "train", "test"))
    mdir = os.path.join(base, "masks", f"rate{ratio}", mask_type)
    train_mask = np.load(os.path.join(mdir, f"train_mask_{idx}.npy"))
    test_mask = np.load(os.path.join(mdir, f"test_mask_{idx}.npy"))
    cols = train_df.columns                                  # dataset.py:36: every frame is indexed by train.csv's header

    def column_dict(df):
        return {j: df[cols[j]].values for j in num_idx + cat_idx}

    train_num = train_df[cols[num_idx]].values.astype(np.float32)
    test_num = test_df[cols[num_idx]].values.astype(np.float32)
    if not cat_idx:                                          # dataset.py:61-66
        return (train_num, test_num, train_mask, test_mask, train_num, test_num, None, None,
                train_mask[:, num_idx], test_mask[:, num_idx], None)
    data_cols = column_dict(data_df)
    train_X, ext_train, train_cat_idx, bins = layout.load_mixed(data_cols, column_dict(train_df), train_mask, num_idx, cat_idx)
    test_X, ext_test, test_cat_idx, _ = layout.load_mixed(data_cols, column_dict(test_df), test_mask, num_idx, cat_idx)
    return (train_X, test_X, train_mask, test_mask, train_num, test_num, train_cat_idx, test_cat_idx, ext_train, ext_test, bins)


def mean_std(data, mask):
    """dataset.py:161-169: per-column mean / std over the observed cells (mask False)."""
    return _mean_std(data, mask)


def get_eval(dataname, X_recon, X_true, truth_cat_idx, num_num, cat_bin_num, mask, oos=False, data_dir: str | None = None):
    """dataset.py:171-206: (MAE, RMSE) over the masked cells of the numeric columns; `mask` is over the ORIGINAL columns and
    is sliced by the Info file's num_col_idx (:183).  news out-of-sample drops row 6265 (:194-197).  The categorical
    arguments are accepted and, like in the reference, not scored."""
    root = DATA_DIR if data_dir is None else data_dir
    with open(os.path.join(root, "Info", f"{dataname}.json")) as f:
        info = json.load(f)
    num_mask = np.asarray(mask)[:, info["num_col_idx"]].astype(bool)
    drop = [6265] if (dataname == "news" and oos) else None
    return mae_rmse(np.asarray(X_recon)[:, :num_num], np.asarray(X_true)[:, :num_num], num_mask, num_num=num_num, drop_rows=drop)
