"""The EM loop around the E-step, as a function (counterpart of main.py:74-262; no CLI, no dataset loading).

M-step = the reference's training loop (main.py:92-152) on `model.Model`, data resident on the device: PyTorch owns the
parameters, the draws, Adam and the scheduler (BASELINE.json north_star: "PyTorch-ROCm owns tensors and the M-step"); on a
CUDA device the loss and gradients of a batch come from the HIP training step (mstep.py), on CPU from
autograd.  E-step = `estep.e_step` on the HIP sampler.  Scoring =
`prep.masked_errors` on the device.  Files written are the reference's (formats.py): `<ckpt_dir>/<k>/model.pt`,
`<ckpt_dir>/iter_<k+1>.npy`.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import torch

from . import estep, formats, prep
from .model import MLPDiffusion, Model


def m_step(train_data: torch.Tensor, hid_dim: int, *, max_epochs: int = 10001, batch_size: int = 4096, lr: float = 1e-4,
           patience: int = 500, generator: torch.Generator | None = None, log=None, native: bool | None = None):
    """main.py:92-152: a FRESH MLPDiffusion each EM iteration, Adam(lr 1e-4), ReduceLROnPlateau(0.9, 50), shuffled batches of
    4096, best-epoch-loss checkpoint, early stop after `patience` epochs without improvement.  Returns (model with the best
    weights loaded, best loss).  `train_data` [N, D] float32 on the training device.

    native: loss and gradients of a batch from the HIP training step (`mstep.NativeTrainStep`, dp_mstep_fwd_bwd: bf16 hi/lo
    split GEMMs, ~1e-5 of float32 autograd, 1.7-1.8x its speed at H = 1024 / batch 4096) instead of PyTorch autograd; the draws,
    Adam and the scheduler stay PyTorch's.  Default (None): native on a CUDA device, autograd on CPU."""
    dev = train_data.device
    n, d = train_data.shape
    model = Model(denoise_fn=MLPDiffusion(d, hid_dim), hid_dim=d).to(dev)
    if native is None:
        native = dev.type == "cuda" and hid_dim % 64 == 0
    step = None
    if native:
        from .mstep import NativeTrainStep
        step = NativeTrainStep(model)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=0, **({"fused": True} if native else {}))
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.9, patience=50)
    model.train()
    best, best_sd, stale = float("inf"), None, 0
    for epoch in range(max_epochs):
        perm = torch.randperm(n, device=dev, generator=generator)
        tot = torch.zeros((), device=dev)
        for b0 in range(0, n, batch_size):
            x = train_data[perm[b0:b0 + batch_size]]
            if step is not None:
                loss = step.loss_and_grads(x)              # overwrites every .grad: zero_grad + backward in one call
            else:
                loss = model(x).mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
            opt.step()
            tot += loss.detach() * x.shape[0]              # accumulated on the device: one host sync per epoch, not per batch
        cur = float(tot) / n
        sched.step(cur)
        if cur < best:
            best, stale = cur, 0
            best_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        else:
            stale += 1
            if stale == patience:
                break
        if log and epoch % 100 == 0:
            log(f"  epoch {epoch}: loss {cur:.4f} (best {best:.4f})")
    if best_sd is None:                                     # every epoch's loss was NaN / inf (e.g. a constant column: std 0 in main.py:58)
        raise FloatingPointError("M-step: the training loss was never finite; check the normalised table for NaN / inf")
    model.load_state_dict(best_sd)
    model.eval()
    return model, best


_WAIT_GROUPS: dict = {}


def trainer_wait_group(days: float = 7.0):
    """A gloo (host-side) process group over all ranks with a timeout of `days`, created once per process (collective: every
    rank must call it).  The M-step runs on ONE rank for up to 10 001 epochs (main.py:100) -- hours on the large tables --
    while the default timeout of a device collective is 10 minutes (RCCL) / 30 (gloo): ranks that simply entered the weight
    broadcast would be aborted by the watchdog long before the trainer arrives.  They wait here instead, on the host."""
    import datetime

    import torch.distributed as dist
    key = dist.get_world_size()
    if key not in _WAIT_GROUPS:
        _WAIT_GROUPS[key] = dist.new_group(backend="gloo", timeout=datetime.timedelta(days=days))
    return _WAIT_GROUPS[key]


def wait_for_trainer(loss: float, src: int = 0, group=None) -> float:
    """Host-side rendezvous after the M-step: returns rank `src`'s loss on every rank once `src` has finished training.
    No device collective is pending while the others wait."""
    import torch.distributed as dist
    box = torch.tensor([loss], dtype=torch.float64)                         # CPU tensor on the gloo group
    dist.broadcast(box, src=src, group=group if group is not None else trainer_wait_group())
    return float(box.item())


def broadcast_model(model, loss: float, src: int = 0):
    """Every rank leaves with rank `src`'s parameters (in place, tensor by tensor in state-dict order) and its loss.
    Call it when every rank is ready (after `wait_for_trainer`): these are device collectives with the default timeout."""
    import torch.distributed as dist
    for t in model.state_dict().values():
        dist.broadcast(t, src=src)
    box = torch.tensor([loss], dtype=torch.float64, device=next(model.parameters()).device)
    dist.broadcast(box, src=src)
    return model, float(box.item())


@dataclass
class EMResult:
    mae_in: list = field(default_factory=list)
    rmse_in: list = field(default_factory=list)
    mae_out: list = field(default_factory=list)
    rmse_out: list = field(default_factory=list)
    train_loss: list = field(default_factory=list)
    imputed: torch.Tensor | None = None          # last in-sample trial mean, x2 scale (what iter_k.npy holds)


def run_em(X, mask_train, X_test=None, mask_test=None, *, hid_dim: int = 1024, max_iter: int = 10, num_trials: int = 20,
           num_steps: int = 50, device="cuda:0", ckpt_dir: str | None = None, max_epochs: int = 10001, dtype: str | None = None,
           seed: int = 0, num_num: int | None = None, oos_drop_rows=None, log=print, native_mstep: bool | None = None) -> EMResult:
    """X, X_test: normalised tables ((x - mean)/std/2, main.py:58,61) holding the TRUE values (used for scoring only, like the
    reference); masks True = missing.
    `oos_drop_rows`: rows of the test table left out of its score (news: [6265], dataset.py:194-197).
    `native_mstep`: see `m_step(native=...)`; None = the HIP training step on a CUDA device, PyTorch autograd on CPU.

    Under torch.distributed (every rank calls this with the same tables): the M-step runs on rank 0 only -- the reference
    has ONE trainer and GPU training is not bit-reproducible across ranks -- and its best weights are broadcast, so every
    rank's row shard of the E-step is sampled from the same model; rank 0 alone writes model.pt / iter_k.npy (one writer,
    as in the reference), followed by a barrier so that no rank runs ahead of the files."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if distributed else 0
    dev = torch.device(device)
    X = torch.as_tensor(X).to(dev, torch.float32).contiguous()
    m_tr = torch.as_tensor(mask_train).to(dev).contiguous()
    gen = torch.Generator(device=dev).manual_seed(seed)
    wait_group = trainer_wait_group() if distributed else None
    res = EMResult()
    train_data = (1.0 - m_tr.float()) * X                                     # iteration 0: zero-filled (main.py:84-86)
    for it in range(max_iter):
        log(f"EM iteration {it}: M-step on {tuple(train_data.shape)}")
        if rank == 0:
            model, loss = m_step(train_data, hid_dim, max_epochs=max_epochs, generator=gen, log=log, native=native_mstep)
        else:
            model, loss = Model(denoise_fn=MLPDiffusion(X.shape[1], hid_dim), hid_dim=X.shape[1]).to(dev).eval(), 0.0
        if distributed:
            loss = wait_for_trainer(loss, src=0, group=wait_group)           # host-side: may take hours
            model, loss = broadcast_model(model, loss, src=0)
        res.train_loss.append(loss)
        if ckpt_dir and rank == 0:
            formats.save_checkpoint(os.path.join(ckpt_dir, str(it), "model.pt"), model)
        net = model.denoise_fn_D
        rec = estep.e_step(net, X, m_tr, num_trials, num_steps, dev, seed=seed + 1000 * it, dtype=dtype)       # main.py:156-188
        rec2 = rec * 2                                                        # main.py:190
        if ckpt_dir and rank == 0:
            formats.save_iteration(os.path.join(ckpt_dir, f"iter_{it + 1}.npy"), rec2)
        if distributed:
            dist.barrier()
        mae, rmse = prep.masked_errors(rec2, X * 2, m_tr, num_num)             # main.py:201
        res.mae_in.append(mae); res.rmse_in.append(rmse)
        log(f"  in-sample      MAE {mae:.4f} RMSE {rmse:.4f}")
        if X_test is not None:
            Xt = torch.as_tensor(X_test).to(dev, torch.float32).contiguous()
            mt = torch.as_tensor(mask_test).to(dev).contiguous()
            rec_t = estep.e_step(net, Xt, mt, num_trials, num_steps, dev, seed=seed + 1000 * it + 500, dtype=dtype)   # main.py:208-238
            mae, rmse = prep.masked_errors(rec_t * 2, Xt * 2, mt, num_num, drop_rows=oos_drop_rows)
            res.mae_out.append(mae); res.rmse_out.append(rmse)
            log(f"  out-of-sample  MAE {mae:.4f} RMSE {rmse:.4f}")
        train_data = rec2 / 2                                                 # next M-step trains on the imputation (main.py:88-90)
        res.imputed = rec2
    return res
