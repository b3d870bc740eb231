#!/usr/bin/env python3
"""Golden fixture for the MIXED-table layout contract: what the reference's `dataset.load_dataset`
and `dataset.mean_std` (dataset.py:10-169) make of a small synthetic mixed table.

Build-container only (imports /root/reference/dataset.py; nothing of the reference is copied).  Steps:
  1. write a scratch tree /tmp/dp_layout/datasets/{mixed,Info}: data/train/test CSVs of OUR OWN synthetic table whose
     numeric, categorical and target columns are interleaved (like adult's, datasets/Info/adult.json), an Info JSON in the
     reference's format, and seeded MCAR masks over the ORIGINAL columns (what generate_mask.py:513-514 writes);
     categorical columns: 9 string categories (4 digits), 5 integer-valued categories (3 digits; the reference maps them
     through str()), 2 categories (1 digit), 1 category (the '0{n}b' format of dataset.py:82 gives it ONE digit, not zero),
     categories in first-appearance order of data.csv, which differs from sorted order and from train.csv's order;
  2. run load_dataset('mixed', 0, 'MCAR', '30') and mean_std on its output;
  3. store the raw columns (inputs) and every array it returned (expected outputs) in tests/golden/layout_mixed.npz.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/make_layout_golden.py
"""
import json
import os
import shutil
import sys

import numpy as np
import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DIFFPUTER_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

SCRATCH = "/tmp/dp_layout"
shutil.rmtree(SCRATCH, ignore_errors=True)
os.makedirs(os.path.join(SCRATCH, "datasets", "mixed", "masks", "rate30", "MCAR"))
os.makedirs(os.path.join(SCRATCH, "datasets", "Info"))
os.chdir(SCRATCH)

rng = np.random.default_rng(20251111)
N, n_train = 300, 200
names9 = ["Private", "Self-emp", "Gov", "Never", "Without-pay", "State", "Local", "Fed", "?"]
df = pd.DataFrame({
    "age": rng.normal(40, 12, N).round(1),                                   # 0 numeric
    "workclass": rng.choice(names9, N, p=[.3, .1, .1, .05, .05, .1, .1, .1, .1]),   # 1 categorical, 9 values
    "fnlwgt": rng.lognormal(10, 1, N).round(0),                              # 2 numeric
    "rooms": rng.choice([3, 1, 7, 2, 11], N),                                # 3 categorical, 5 integer values
    "sex": rng.choice(["Male", "Female"], N),                                # 4 categorical, 2 values
    "gain": (rng.standard_normal(N) * 1e3).round(3),                         # 5 numeric
    "country": np.full(N, "US"),                                             # 6 categorical, 1 value
    "hours": rng.integers(1, 99, N),                                         # 7 numeric (integers in the CSV)
    "income": rng.choice(["<=50K", ">50K"], N),                              # 8 target (excluded)
})
info = {"name": "mixed", "task_type": "binclass", "header": None, "num_col_idx": [0, 2, 5, 7], "cat_col_idx": [1, 3, 4, 6],
        "target_col_idx": [8]}
perm = rng.permutation(N)
train_df, test_df = df.iloc[perm[:n_train]], df.iloc[perm[n_train:]]
df.to_csv("datasets/mixed/data.csv", index=False)
train_df.to_csv("datasets/mixed/train.csv", index=False)
test_df.to_csv("datasets/mixed/test.csv", index=False)
json.dump(info, open("datasets/Info/mixed.json", "w"))
train_mask0 = rng.random((n_train, 9)) < 0.3
test_mask0 = rng.random((N - n_train, 9)) < 0.3
np.save("datasets/mixed/masks/rate30/MCAR/train_mask_0.npy", train_mask0)
np.save("datasets/mixed/masks/rate30/MCAR/test_mask_0.npy", test_mask0)

import dataset as ref_ds                                                     # noqa: E402  (the reference)

(train_X, test_X, ori_train_mask, ori_test_mask, train_num, test_num, train_cat_idx, test_cat_idx, train_mask, test_mask,
 cat_bin_num) = ref_ds.load_dataset("mixed", 0, "MCAR", "30")
mean_X, std_X = ref_ds.mean_std(train_X, train_mask)
print("train_X", train_X.shape, train_X.dtype, "cat_bin_num", cat_bin_num, "mask", train_mask.shape, train_mask.mean())

# inputs as the CSV reader hands them over (the files are re-read so that float formatting round trips are included)
data_r, train_r, test_r = (pd.read_csv(f"datasets/mixed/{f}.csv") for f in ("data", "train", "test"))
cols = list(train_r.columns)
raw = {}
for tag, d in (("data", data_r), ("train", train_r), ("test", test_r)):
    for j in info["num_col_idx"]:
        raw[f"{tag}_num_{j}"] = d[cols[j]].values.astype(np.float64)
    for j in info["cat_col_idx"]:
        raw[f"{tag}_cat_{j}"] = d[cols[j]].astype(str).values.astype("U")
np.savez_compressed(
    os.path.join(REPO, "tests", "golden", "layout_mixed.npz"),
    num_col_idx=np.asarray(info["num_col_idx"]), cat_col_idx=np.asarray(info["cat_col_idx"]),
    train_mask0=train_mask0, test_mask0=test_mask0, **raw,
    train_X=train_X, test_X=test_X, train_mask=train_mask, test_mask=test_mask, train_num=train_num, test_num=test_num,
    train_cat_idx=train_cat_idx, test_cat_idx=test_cat_idx, cat_bin_num=cat_bin_num, mean=mean_X, std=std_X)
print("wrote tests/golden/layout_mixed.npz", os.path.getsize(os.path.join(REPO, "tests", "golden", "layout_mixed.npz")) // 1024, "KiB")
