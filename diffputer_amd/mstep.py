"""The M-step's training step on the HIP kernels (`dp_mstep_fwd_bwd`, include/diffputer.h).

Counterpart of `loss = model(inputs); loss.backward()` (main.py:124-131): one call computes the EDM loss of a batch
(model.py:171-181 -> diffusion_utils.py:200-215 -> model.py:148-165 -> :121-128) and writes its gradient into the `.grad` of
the model's 14 parameters.  PyTorch keeps what the north star leaves it: the parameters, the random draws (`torch.randn`,
`torch.randn_like`, in the reference's order), the optimiser and the scheduler.

Arithmetic: float32 state; GEMM operands split into bf16 hi + lo, three bf16 MFMA products per float32 product, float32
accumulation (diffputer_amd/csrc/dp_mstep.h).  Gradients agree with the reference's float32 autograd to ~1e-5 of their
scale (tests/test_gpu_mstep.py).  There is no CPU fallback: on a CPU model use autograd (`Model.forward`).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .sampler import _LAYERS


class NativeTrainStep:
    """Binds a `model.Model` (or anything whose state dict holds the 14 DiffPuter tensors) to `dp_mstep_fwd_bwd`."""

    def __init__(self, model):
        self.lib = _lib.load()
        named = dict(model.named_parameters())
        self.params = []
        for name in _LAYERS:
            for kind in ("weight", "bias"):
                hits = [k for k in named if k == f"{name}.{kind}" or k.endswith(f".{name}.{kind}")]
                if len(hits) != 1:
                    raise KeyError(f"expected exactly one '{name}.{kind}' parameter, found {hits}")
                p = named[hits[0]]
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.DiffputerError("NativeTrainStep needs contiguous float32 parameters on a CUDA (ROCm) device; there is no CPU fallback")
                self.params.append(p)
        self.device = self.params[0].device
        self.hid, self.d_in = self.params[0].shape                  # proj.weight [hid, d_in]
        self._ws = None
        self._loss = torch.zeros((), dtype=torch.float32, device=self.device)

    def _workspace(self, batch: int) -> torch.Tensor:
        need = self.lib.dp_mstep_workspace_bytes(int(self.d_in), int(self.hid), int(batch))
        if need == 0:
            _lib.check(-1, "dp_mstep_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def loss_and_grads(self, x: torch.Tensor, z_sigma: torch.Tensor | None = None, z_noise: torch.Tensor | None = None,
                       percell: bool = False):
        """`loss = model(x); loss.backward()` after a zero_grad: returns the loss (0-dim tensor on the device; a view of a
        buffer the next call overwrites) and OVERWRITES every parameter's `.grad`.  z_sigma [B] / z_noise [B, D]: the two
        standard-normal draws of EDMLoss.__call__ (diffusion_utils.py:202,208); drawn here, in that order, when not given."""
        x = x.to(self.device, torch.float32).contiguous()
        if x.ndim != 2 or x.shape[1] != self.d_in or x.shape[0] < 1:
            raise ValueError(f"x {tuple(x.shape)} does not match d_in={self.d_in}")
        B = x.shape[0]
        if z_sigma is None:
            z_sigma = torch.randn(B, device=self.device)
        if z_noise is None:
            z_noise = torch.randn_like(x)
        z_sigma = z_sigma.to(self.device, torch.float32).contiguous()
        z_noise = z_noise.to(self.device, torch.float32).contiguous()
        if tuple(z_sigma.shape) != (B,) or tuple(z_noise.shape) != tuple(x.shape):
            raise ValueError("z_sigma must be [B] and z_noise [B, D]")
        for p in self.params:
            if p.grad is None or not p.grad.is_contiguous():
                p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
        w = (C.c_void_p * 7)(*[self.params[2 * i].data_ptr() for i in range(7)])
        b = (C.c_void_p * 7)(*[self.params[2 * i + 1].data_ptr() for i in range(7)])
        gw = (C.c_void_p * 7)(*[self.params[2 * i].grad.data_ptr() for i in range(7)])
        gb = (C.c_void_p * 7)(*[self.params[2 * i + 1].grad.data_ptr() for i in range(7)])
        pc = torch.empty_like(x) if percell else None
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            rc = self.lib.dp_mstep_fwd_bwd(int(self.d_in), int(self.hid), B, x.data_ptr(), z_sigma.data_ptr(), z_noise.data_ptr(), w, b, gw, gb,
                                           self._loss.data_ptr(), pc.data_ptr() if percell else None, ws.data_ptr(), ws.numel(),
                                           C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _lib.check(rc, "dp_mstep_fwd_bwd")
        return (self._loss, pc) if percell else self._loss
