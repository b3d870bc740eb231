"""The two files the reference's EM loop exchanges between its M-step and E-step.

* `model.pt`: `torch.save(model.state_dict())` of `Model` (main.py:142); keys
  `denoise_fn_D.denoise_fn_F.{proj,mlp.0,mlp.2,mlp.4,mlp.6,time_embed.0,time_embed.2}.{weight,bias}`, float32.
  Loaded with `weights_only=True`: a checkpoint is data, never code.
* `iter_{k+1}.npy`: the trial-mean imputation of the training table in the x2 scale (main.py:190-193), float32
  `[N, D]`; the next M-step reads it back and divides by 2 (main.py:88-89).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .sampler import _LAYERS


def save_checkpoint(path: str, model) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)


def load_checkpoint(path: str, map_location="cpu") -> dict:
    """State dict of a reference- or diffputer_amd-trained `model.pt`; checks the 14 tensors are there."""
    sd = torch.load(path, map_location=map_location, weights_only=True)
    for name in _LAYERS:
        for kind in ("weight", "bias"):
            if not any(k == f"{name}.{kind}" or k.endswith(f".{name}.{kind}") for k in sd):
                raise KeyError(f"{path}: no '{name}.{kind}' tensor; not a DiffPuter checkpoint")
    return sd


def save_iteration(path: str, rec_x2) -> None:
    """rec_x2: the E-step's trial mean already multiplied by 2 (main.py:190)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    a = rec_x2.detach().cpu().numpy() if isinstance(rec_x2, torch.Tensor) else np.asarray(rec_x2)
    np.save(path, np.ascontiguousarray(a, dtype=np.float32))


def load_iteration(path: str) -> np.ndarray:
    """The next M-step's training table: iter_k.npy / 2 (main.py:88-89)."""
    return np.load(path, allow_pickle=False).astype(np.float32) / 2
