"""Normalisation and scoring around the E-step, on the device.

Counterparts of `mean_std` (dataset.py:161-169), the scaling of main.py:58,61 and `get_eval`'s MAE / RMSE
(dataset.py:199-201).  The numpy forms live in `evalmetrics`; these keep a large table on the GPU between the
M-step and the E-step.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def masked_mean_std(X: torch.Tensor, mask: torch.Tensor):
    """Per-column mean / std over observed cells (mask False).  X [N, D] float32 CUDA, mask bool/uint8.  -> (mean[D], std[D])."""
    lib = _lib.load()
    X = X.to(torch.float32).contiguous()
    m = mask.to(X.device).to(torch.uint8).contiguous()
    if not X.is_cuda or X.ndim != 2 or m.shape != X.shape:
        raise ValueError("X must be a 2-D CUDA tensor and mask must have its shape")
    n, d = X.shape
    mean = torch.empty(d, dtype=torch.float32, device=X.device)
    std = torch.empty(d, dtype=torch.float32, device=X.device)
    ws = torch.empty(32 * 2 * d, dtype=torch.float64, device=X.device)        # 32 copies of the partial sums (include/diffputer.h)
    with torch.cuda.device(X.device):
        rc = lib.dp_masked_mean_std(X.data_ptr(), m.data_ptr(), n, d, mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                    _stream(X.device))
    _lib.check(rc, "dp_masked_mean_std")
    return mean, std


def normalize(X: torch.Tensor, mean: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """(X - mean) / std / 2: data std 0.5 = the EDM sigma_data (main.py:58,61; model.py:137)."""
    return (X - mean) / std / 2


def masked_errors(pred: torch.Tensor, truth: torch.Tensor, mask: torch.Tensor, num_num: int | None = None, drop_rows=None):
    """MAE, RMSE over the missing cells of the first `num_num` columns (get_eval, dataset.py:199-201); pred/truth in the
    x2 scale of main.py:190-191.  One device reduction, one 24-byte read-back.  `drop_rows`: rows left out of the score
    (the reference drops row 6265 of news' out-of-sample table, dataset.py:194-197)."""
    lib = _lib.load()
    pred = pred.to(torch.float32).contiguous()
    truth = truth.to(pred.device, torch.float32).contiguous()
    m = mask.to(pred.device).to(torch.uint8).contiguous()
    if drop_rows is not None:
        m = m.clone()
        m[torch.as_tensor(drop_rows, device=m.device, dtype=torch.long)] = 0
    n, d = pred.shape
    k = d if num_num is None else int(num_num)
    sums = torch.empty(3, dtype=torch.float64, device=pred.device)
    with torch.cuda.device(pred.device):
        rc = lib.dp_masked_errors(pred.data_ptr(), truth.data_ptr(), m.data_ptr(), n, d, m.shape[1], k, sums.data_ptr(), _stream(pred.device))
    _lib.check(rc, "dp_masked_errors")
    a, s, c = sums.tolist()
    if c == 0:
        raise ValueError("no masked numeric cell to score")
    return a / c, math.sqrt(s / c)
