"""MAE / RMSE on the masked numeric cells, as the reference scores an imputation.

Counterpart of `get_eval` (dataset.py:171-206): `div = pred[mask] - true[mask]` over the numeric
columns, `mae = |div|.mean()`, `rmse = sqrt((div**2).mean())`, both tables in the x2 scale of
main.py:190-191,240-241 (standardised units).  The categorical slice is de-normalised by the caller
(main.py:198-199) but never scored (get_eval returns only mae, rmse).  Host code, O(N*D) once per
E-step; the news out-of-sample quirk (row 6265 dropped, dataset.py:194-197) is the caller's `drop_rows`.
"""
from __future__ import annotations

import numpy as np


def mae_rmse(pred, true, mask, num_num: int | None = None, drop_rows=None):
    """pred, true: [N, D] arrays (first `num_num` columns numeric; default all); mask: bool [N, >=num_num], True = missing."""
    pred = np.asarray(pred)
    true = np.asarray(true)
    mask = np.asarray(mask).astype(bool)
    k = pred.shape[1] if num_num is None else int(num_num)
    p, t, m = pred[:, :k], true[:, :k], mask[:, :k]
    if drop_rows is not None:
        p, t, m = (np.delete(x, drop_rows, axis=0) for x in (p, t, m))
    div = p[m] - t[m]
    if div.size == 0:
        raise ValueError("no masked numeric cell to score")
    return float(np.abs(div).mean()), float(np.sqrt((div ** 2).mean()))


def mean_std(data, mask):
    """Per-column mean / std over the OBSERVED cells (mask False), float32 like the reference (dataset.py:161-169)."""
    obs = (~np.asarray(mask).astype(bool)).astype(np.float32)
    data = np.asarray(data, dtype=np.float32)
    cnt = obs.sum(0)
    cnt[cnt == 0] = 1
    mean = (data * obs).sum(0) / cnt
    var = ((data - mean) ** 2 * obs).sum(0) / cnt
    return mean, np.sqrt(var)
