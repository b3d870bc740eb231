"""SamplerPlan: the Python handle of a `dp_plan` (include/diffputer.h).

Owns nothing but the plan handle and a cached workspace tensor; every table, mask, output and
workspace is a torch CUDA tensor whose raw pointer is lent to the library for the duration of a
call, on torch's current stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_LAYERS = ["proj", "mlp.0", "mlp.2", "mlp.4", "mlp.6", "time_embed.0", "time_embed.2"]
_DTYPES = {"f32": _lib.DP_DTYPE_F32, "fp32": _lib.DP_DTYPE_F32, "float32": _lib.DP_DTYPE_F32,
           "bf16": _lib.DP_DTYPE_BF16, "bfloat16": _lib.DP_DTYPE_BF16}


def _state_dict_of(net_or_sd) -> dict:
    """Accept a Precond (`net`), a Model, an MLPDiffusion or a plain state dict (any key prefix)."""
    sd = net_or_sd if isinstance(net_or_sd, dict) else net_or_sd.state_dict()
    out = {}
    for name in _LAYERS:
        for kind in ("weight", "bias"):
            hits = [k for k in sd if k == f"{name}.{kind}" or k.endswith(f".{name}.{kind}")]
            if len(hits) != 1:
                raise KeyError(f"expected exactly one '{name}.{kind}' tensor in the state dict, found {hits}")
            out[f"{name}.{kind}"] = sd[hits[0]]
    return out


def n_draws(num_steps: int, n_resample: int, final_level: bool = True) -> int:
    return 1 + 4 * n_resample * (num_steps - 1) + (3 if final_level else 0)


def n_iter(num_steps: int, n_resample: int, final_level: bool = True) -> int:
    return n_resample * (num_steps - 1) + (1 if final_level else 0)


class SamplerPlan:
    """Schedule + packed weights + per-level embedding bias for one checkpoint on one GPU."""

    def __init__(self, net_or_state_dict, num_steps: int = 50, n_resample: int = 20, dtype: str = "bf16",
                 device="cuda:0"):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DiffputerError("SamplerPlan needs a CUDA (ROCm) device; there is no CPU fallback")
        if dtype not in _DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_DTYPES)}")
        sd = _state_dict_of(net_or_state_dict)
        self.hid, self.d_in = sd["proj.weight"].shape
        self.num_steps, self.n_resample, self.dtype = int(num_steps), int(n_resample), dtype
        self._h = C.c_void_p()
        self._ws = None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dp_plan_create(C.byref(self._h), int(self.d_in), int(self.hid), self.num_steps,
                                               self.n_resample, _DTYPES[dtype]), "dp_plan_create")
            self.set_weights(sd)

    # -- plumbing ------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_weights(self, net_or_state_dict):
        sd = _state_dict_of(net_or_state_dict)
        keep = []
        w = (C.c_void_p * 7)()
        b = (C.c_void_p * 7)()
        for i, name in enumerate(_LAYERS):
            for arr, kind in ((w, "weight"), (b, "bias")):
                t = torch.as_tensor(sd[f"{name}.{kind}"]).detach().to(self.device, torch.float32).contiguous()
                keep.append(t)
                arr[i] = t.data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dp_plan_set_weights(self._h, w, b, self._stream()), "dp_plan_set_weights")
        del keep

    def workspace_bytes(self, n_chains: int) -> int:
        n = self.lib.dp_workspace_bytes(self._h, int(n_chains))
        if n == 0:
            _lib.check(-1, "dp_workspace_bytes")
        return n

    def _workspace(self, n_chains: int) -> torch.Tensor:
        need = self.workspace_bytes(n_chains)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.dp_plan_destroy(self._h)
            self._h = C.c_void_p()
        self._ws = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inspection ----------------------------------------------------------------------------------------
    def schedule(self) -> np.ndarray:
        out = np.empty(self.num_steps + 1, dtype=np.float32)
        _lib.check(self.lib.dp_plan_schedule(self._h, out.ctypes.data_as(C.POINTER(C.c_float))), "dp_plan_schedule")
        return out

    def levels(self) -> np.ndarray:
        out = np.empty(2 * self.num_steps - 1, dtype=np.float32)
        _lib.check(self.lib.dp_plan_levels(self._h, out.ctypes.data_as(C.POINTER(C.c_float))), "dp_plan_levels")
        return out

    def embedding(self) -> np.ndarray:
        out = np.empty((2 * self.num_steps - 1, self.hid), dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dp_plan_embedding(self._h, out.ctypes.data_as(C.POINTER(C.c_float))), "dp_plan_embedding")
        return out

    def profile(self, stride: int):
        """Bracket every `stride`-th denoiser evaluation of later impute calls with HIP events (0 = off)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dp_plan_profile(self._h, int(stride)), "dp_plan_profile")

    def profile_read(self):
        """(ms_sum[8], count[8]) per kernel slot: 0..4 GEMM layers, 5 step_mid, 6 step_end, 7 step_final."""
        ms = (C.c_double * 8)()
        cnt = (C.c_int * 8)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dp_plan_profile_read(self._h, ms, cnt), "dp_plan_profile_read")
        return list(ms), list(cnt)

    def profile_clock(self):
        """(shader cycles, wall microseconds, samples) summed over the profiled mlp.2 launches: the in-kernel clock."""
        cyc, us, n = C.c_double(), C.c_double(), C.c_int()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dp_plan_profile_clock(self._h, C.byref(cyc), C.byref(us), C.byref(n)), "dp_plan_profile_clock")
        return cyc.value, us.value, n.value

    def gemm_rows(self, n_chains: int) -> int:
        """Rows per launch of the wide-layer GEMMs for a call of `n_chains` chains (dp_tune "m_chunk")."""
        return int(self.lib.dp_plan_gemm_rows(self._h, int(n_chains)))

    def gemm_kernel(self, layer: int) -> str:
        """Kernel that runs profile slot `layer` (0 proj, 1 mlp.0 / folded proj+mlp.0, 2 mlp.2, 3 mlp.4, 4 mlp.6) under the current
        dp_tune settings -- the library's own dispatch predicate (dp_plan_gemm_kernel); '' for a slot that launches nothing."""
        return (self.lib.dp_plan_gemm_kernel(self._h, int(layer)) or b"").decode()

    # -- the sampler ---------------------------------------------------------------------------------------
    def _prep(self, x_obs, mask):
        x = torch.as_tensor(x_obs).to(self.device, torch.float32).contiguous()
        m = torch.as_tensor(mask).to(self.device).to(torch.uint8).contiguous()
        if x.ndim != 2 or x.shape[1] != self.d_in or m.shape != x.shape:
            raise ValueError(f"x_obs {tuple(x.shape)} / mask {tuple(m.shape)} do not match d_in={self.d_in}")
        return x, m

    def impute_sum(self, x_obs, mask, num_trials: int = 1, *, seed: int = 0, row_offset: int = 0, trial_offset: int = 0,
                   noise=None, trace: bool = False, out=None, accumulate: bool = False, final_level: bool = True,
                   static_scratch: bool = False):
        """One `dp_impute` call: all `num_trials` trials of all rows as one batch of chains.
        Returns (sum over trials [n, D], trace or None).  final_level=False: DP_NO_FINAL_LEVEL (the MissDiff variant).
        static_scratch=True: `dp_impute_static` -- the library's own scratch array instead of a torch workspace (small
        calls only; the C call then allocates nothing, pass `out` to keep this wrapper from allocating too)."""
        x, m = self._prep(x_obs, mask)
        n = x.shape[0]
        if n == 0:
            raise ValueError("empty table: impute needs at least one row")
        if out is None:
            out = torch.empty_like(x)
            accumulate = False
        elif (not isinstance(out, torch.Tensor) or out.device != x.device or out.dtype != torch.float32
              or tuple(out.shape) != tuple(x.shape) or not out.is_contiguous()):
            raise ValueError("out must be a contiguous (row-major) float32 tensor of x_obs's shape on the plan's device")
        nz = None
        if noise is not None:
            nz = torch.as_tensor(noise).to(self.device, torch.float32).contiguous()
            want = (num_trials, n_draws(self.num_steps, self.n_resample, final_level), n, self.d_in)
            if nz.ndim == 3:
                nz = nz.unsqueeze(0)
            if tuple(nz.shape) != want:
                raise ValueError(f"noise has shape {tuple(nz.shape)}, expected {want}")
        tr = None
        if trace:
            tr = torch.empty((n_iter(self.num_steps, self.n_resample, final_level), num_trials, n, self.d_in), dtype=torch.float32,
                             device=self.device)
        flags = (_lib.DP_ACCUMULATE if accumulate else 0) | (0 if final_level else _lib.DP_NO_FINAL_LEVEL)
        if static_scratch:
            with torch.cuda.device(self.device):
                rc = self.lib.dp_impute_static(self._h, x.data_ptr(), m.data_ptr(), out.data_ptr(), n, int(num_trials), int(seed),
                                               int(row_offset), int(trial_offset), nz.data_ptr() if nz is not None else None,
                                               tr.data_ptr() if tr is not None else None, flags, self._stream())
            _lib.check(rc, "dp_impute_static")
            return out, tr
        ws = self._workspace(n * num_trials)
        with torch.cuda.device(self.device):
            rc = self.lib.dp_impute(self._h, x.data_ptr(), m.data_ptr(), out.data_ptr(), n, int(num_trials), int(seed),
                                    int(row_offset), int(trial_offset), nz.data_ptr() if nz is not None else None,
                                    tr.data_ptr() if tr is not None else None, flags,
                                    ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "dp_impute")
        return out, tr

    def denoise(self, x, sigma: float, want_F: bool = False):
        """D(x, sigma) for a scalar sigma: the `net(x, sigma)` of diffusion_utils.py:99,105."""
        xx = torch.as_tensor(x).to(self.device, torch.float32).contiguous()
        n = xx.shape[0]
        D = torch.empty_like(xx)
        F = torch.empty_like(xx) if want_F else None
        ws = self._workspace(n)
        with torch.cuda.device(self.device):
            rc = self.lib.dp_denoise(self._h, xx.data_ptr(), float(sigma), D.data_ptr(), F.data_ptr() if want_F else None, n,
                                     ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "dp_denoise")
        return (D, F) if want_F else D

    def sample_step(self, i: int, t_cur: float, t_next: float, x_cur, eps=None, *, seed: int = 0, row_offset: int = 0,
                    block: int = 0):
        xx = torch.as_tensor(x_cur).to(self.device, torch.float32).contiguous()
        n = xx.shape[0]
        e = None if eps is None else torch.as_tensor(eps).to(self.device, torch.float32).contiguous()
        out = torch.empty_like(xx)
        ws = self._workspace(n)
        with torch.cuda.device(self.device):
            rc = self.lib.dp_sample_step(self._h, int(i), float(t_cur), float(t_next), xx.data_ptr(),
                                         e.data_ptr() if e is not None else None, out.data_ptr(), n, int(seed), int(row_offset),
                                         int(block), ws.data_ptr(), ws.numel(), self._stream())
        _lib.check(rc, "dp_sample_step")
        return out


def linear(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, act: bool = False, out_f32: bool = False) -> torch.Tensor:
    """Single GEMM kernel entry (`dp_linear`): act(A @ W^T + bias).  A, W both bf16 or both f32."""
    lib = _lib.load()
    assert A.is_cuda and W.is_cuda and A.dtype == W.dtype and A.is_contiguous() and W.is_contiguous()
    M, K = A.shape
    N = W.shape[0]
    if A.dtype == torch.bfloat16:
        dt, odt = _lib.DP_DTYPE_BF16, (torch.float32 if out_f32 else torch.bfloat16)
    elif A.dtype == torch.float32:
        dt, odt = _lib.DP_DTYPE_F32, torch.float32
    else:
        raise ValueError("dp_linear takes bf16 or f32 operands")
    out = torch.empty((M, N), dtype=odt, device=A.device)
    b = bias.to(A.device, torch.float32).contiguous()
    with torch.cuda.device(A.device):
        rc = lib.dp_linear(dt, A.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, K, int(act), int(out_f32),
                           C.c_void_p(torch.cuda.current_stream(A.device).cuda_stream))
    _lib.check(rc, "dp_linear")
    return out


def noise_field(n_rows: int, d_in: int, seed: int, row_offset: int, trial: int, block: int, comp: int, device="cuda:0"):
    lib = _lib.load()
    dev = torch.device(device)
    out = torch.empty((n_rows, d_in), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dp_noise_field(out.data_ptr(), n_rows, d_in, int(seed), int(row_offset), int(trial), int(block) & 0xFFFFFFFF,
                                int(comp), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "dp_noise_field")
    return out
