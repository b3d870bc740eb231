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
    """Digits of a categorical column with `num_categories` values (dataset.py:80); 0 for a single-valued column."""
    return (int(num_categories) - 1).bit_length()


def encode_bits(cat_idx: np.ndarray, num_bits: int) -> np.ndarray:
    """[N] category indices -> [N, num_bits] float32 digits, most significant first (the '0{n}b' format of dataset.py:82)."""
    idx = np.asarray(cat_idx, dtype=np.int64)
    if idx.size and (idx.min() < 0 or idx.max() >= (1 << num_bits) and num_bits > 0 or (num_bits == 0 and idx.max() > 0)):
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
