"""The table layout the sampler consumes for MIXED tables, as array functions.

The reference builds it inside `load_dataset` (dataset.py:70-157) from CSV files: columns = [numeric columns, then per
categorical column the binary digits of its category index, MSB first, as 0.0 / 1.0 floats] with
`num_bits = (num_categories - 1).bit_length()` (dataset.py:80-82), and the missing-mask of a categorical column repeated
over its digits (`extend_mask`, :140-157).  A caller that already holds category indices gets the same arrays here
without the CSV / JSON round trip; numeric-only tables need none of this (dataset.py:61-66).  Host code, numpy.
"""
from __future__ import annotations

import numpy as np


def bits_for(num_categories: int) -> int:
    """Digits of a categorical column with `num_categories` values: `(n - 1).bit_length()` (dataset.py:80) -- except that
    a single-valued column still gets ONE digit (always 0): the reference formats index 0 with '00b', which prints '0'
    (dataset.py:82), and counts the digits it printed (:122).  Pinned by tests/golden/layout_mixed.npz."""
    return max(1, (int(num_categories) - 1).bit_length())


def category_index(data_values, values) -> np.ndarray:
    """Category indices as `load_dataset` assigns them (dataset.py:39,77,83,110): every value goes through str(), and a
    category's index is its rank of FIRST APPEARANCE in the full table's column (`data.csv`), not its sorted rank.
    data_values: the column of the full table; values: the column of the split to encode.  -> [N] int64."""
    order: dict = {}
    for v in np.asarray(data_values).astype(str):
        order.setdefault(v, len(order))
    try:
        return np.asarray([order[v] for v in np.asarray(values).astype(str)], dtype=np.int64)
    except KeyError as e:
        raise ValueError(f"category {e.args[0]!r} does not occur in the full table") from None


def load_mixed(data_cols, split_cols, mask, num_col_idx, cat_col_idx):
    """The mixed-table branch of `load_dataset` (dataset.py:36-49, 70-157) on columns already read from the CSVs.
    data_cols / split_cols: one array per ORIGINAL column of data.csv / of the split (train.csv or test.csv), indexable
    by original column index (a list, or a dict keyed by index); mask [N, n_original_columns] bool, True = missing.
    -> (X [N, D] float32, extended mask [N, D] bool, cat_idx [N, c] int64, cat_bin_num [c]); numeric columns first in
    num_col_idx order, then each categorical column's digits in cat_col_idx order; the target column is dropped."""
    num = np.stack([np.asarray(split_cols[j]).astype(np.float32) for j in num_col_idx], axis=1)
    cat_idx = np.stack([category_index(data_cols[j], split_cols[j]) for j in cat_col_idx], axis=1)
    n_cat = [len(set(np.asarray(data_cols[j]).astype(str))) for j in cat_col_idx]
    X, bins = encode_table(num, cat_idx, n_cat)
    mask = np.asarray(mask).astype(bool)
    m = extend_mask(np.concatenate([mask[:, list(num_col_idx)], mask[:, list(cat_col_idx)]], axis=1), len(num_col_idx), bins)
    return X, m, cat_idx, bins


def encode_bits(cat_idx: np.ndarray, num_bits: int) -> np.ndarray:
    """[N] category indices -> [N, num_bits] float32 digits, most significant first (the '0{n}b' format of dataset.py:82)."""
    idx = np.asarray(cat_idx, dtype=np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= (1 << num_bits)):
        raise ValueError(f"category index out of range for {num_bits} bits")
    shifts = np.arange(num_bits - 1, -1, -1, dtype=np.int64)
    return ((idx[:, None] >> shifts[None, :]) & 1).astype(np.float32)


def encode_table(num: np.ndarray, cat_idx: np.ndarray, num_categories):
    """num [N, k] float32, cat_idx [N, c] int, num_categories [c] -> (X [N, k + sum(bits)] float32, cat_bin_num [c] int):
    the `train_X` and `cat_bin_num` of dataset.py:124-133."""
    num = np.asarray(num, dtype=np.float32)
    cat_idx = np.asarray(cat_idx)
    bins = np.array([bits_for(n) for n in num_categories], dtype=np.int64)
    parts = [num] + [encode_bits(cat_idx[:, j], int(b)) for j, b in enumerate(bins)]
    return np.concatenate(parts, axis=1), bins


def extend_mask(mask: np.ndarray, num_numeric: int, cat_bin_num) -> np.ndarray:
    """mask [N, k + c] bool over the ORIGINAL columns (numeric first, True = missing) -> [N, k + sum(bits)]: each
    categorical column's bit repeated over its digits (dataset.py:135-157)."""
    mask = np.asarray(mask).astype(bool)
    bins = np.asarray(cat_bin_num, dtype=np.int64)
    if mask.shape[1] != num_numeric + len(bins):
        raise ValueError("mask width must be num_numeric + number of categorical columns")
    return np.concatenate([mask[:, :num_numeric], np.repeat(mask[:, num_numeric:], bins, axis=1)], axis=1)


def decode_bits(x_cat: np.ndarray, cat_bin_num) -> np.ndarray:
    """Imputed digit columns [N, sum(bits)] (de-normalised, any real values) -> category indices [N, c]: a digit is 1
    where the value is above 0.5.  (The reference de-normalises this slice, main.py:198-199, but never scores it.)"""
    bins = np.asarray(cat_bin_num, dtype=np.int64)
    digits = (np.asarray(x_cat) > 0.5).astype(np.int64)
    out = np.zeros((digits.shape[0], len(bins)), dtype=np.int64)
    o = 0
    for j, b in enumerate(bins):
        for k in range(int(b)):
            out[:, j] = (out[:, j] << 1) | digits[:, o + k]
        o += int(b)
    return out
