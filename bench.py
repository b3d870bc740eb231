#!/usr/bin/env python3
"""bench.py -- E-step imputation throughput on MI355X (BASELINE.json metric: imputed rows/sec,
E-step, 50 steps x 10 trials).

    python bench.py --gpus N --steps K --warmup W

A "step" is one full E-step pass (S=50 noise levels, R=20 resamples, Heun, T=10 trials -> 19 610
denoiser evaluations per row) over one row shard of the synthetic 10M x 256 table of BASELINE.json
configs[4].  The whole 10M-row table is ~1 h of MFMA work per GPU, so each step processes
`--rows` rows of it per GPU (rows are independent chains; throughput in rows/s does not depend on
the shard length once the GPU is full).  Inputs are resident in HBM before the timed region.
Weak scaling: every rank takes its own `--rows` rows (global row offsets keep the Philox noise
shard-invariant) and the per-step all-gather of the trial means is inside the timed region.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own ranks: before anything touches
the GPU it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same
flags>` as a child process, passes rank 0's JSON line through and exits with the child's code (a process that has
initialised the GPU is never re-executed).  `--config adult|news|synth1m|synth10m` selects the BASELINE.json shape.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the 256x256-tile bf16 MFMA
GEMM that runs mlp.0 / mlp.2 / mlp.4): algorithmic FLOPs per launch / in-situ average launch
duration from HIP events recorded on the launch stream during the timed steps (dp_plan_profile).
`cpu_baseline` is the numpy oracle (a port of the reference algorithm) timed on this box's host
cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from diffputer_amd import estep, synth                                   # noqa: E402
from diffputer_amd.model import flops_per_row_eval, nfe_per_chain        # noqa: E402
from diffputer_amd.sampler import SamplerPlan                            # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3        # f32-input MFMA peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec peak


def cpu_baseline(d, hid, S, R, T, rows, budget_note):
    """The CPU port on the host cores: `rows` rows x 1 trial of the same workload, float32, torch CPU operators
    (what the reference's sampler runs on: `addmm` on torch's BLAS, torch's SiLU), oracle/diffputer_oracle_torch.py."""
    from oracle import diffputer_oracle_torch as orct                     # the only place bench.py touches oracle/
    cores = torch.get_num_threads()
    try:                                                                  # physical cores of the host, beside the threads used
        import psutil
        phys, logical = psutil.cpu_count(logical=False), psutil.cpu_count(logical=True)
    except Exception:                                                     # noqa: BLE001
        phys, logical = None, os.cpu_count()
    sd = synth.make_weights(d, hid, seed=11, gain=1.0)
    X, mask = synth.make_table(rows, d, seed=12)
    x_obs = (X * (1.0 - mask)).astype(np.float32)

    class LazyNoise:                                                      # 3 924 x rows x d floats would not fit; draw on demand
        def __init__(self):
            self.gen = torch.Generator().manual_seed(13)

        def __getitem__(self, k):
            return torch.randn((rows, d), generator=self.gen)

    t0 = time.perf_counter()
    orct.impute_mask(sd, x_obs, mask, LazyNoise(), S, R)
    dt = time.perf_counter() - t0
    row_trials_per_s = rows / dt
    return {"value": row_trials_per_s / T, "unit": "rows/s", "cores": cores, "host_physical_cores": phys, "host_logical_cpus": logical,
            "kind": "port",
            "sample": f"{rows} rows x 1 trial, full S={S}/R={R} schedule, D={d}, H={hid}, float32 port on torch CPU operators "
                      f"({dt:.1f} s on {cores} threads; rows/s = row-trials/s / T={T}){budget_note}",
            "row_trials_per_s": row_trials_per_s}


def mstep_line(d, hid, dev, batch=4096, steps=40):
    """SURVEY.md §8 f1 beside the E-step: one M-step training step (main.py:119-133: loss, backward, Adam) at the reference's
    batch size, on the HIP training step (dp_mstep_fwd_bwd + PyTorch's fused Adam) and on PyTorch eager autograd."""
    from diffputer_amd.model import MLPDiffusion, Model
    from diffputer_amd.mstep import NativeTrainStep
    torch.manual_seed(0)
    x = torch.randn(batch, d, device=dev) * 0.5

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3

    model = Model(denoise_fn=MLPDiffusion(d, hid), hid_dim=d).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0, fused=True)
    step = NativeTrainStep(model)

    def native():
        step.loss_and_grads(x)
        opt.step()

    native_ms = timed(native)
    model = Model(denoise_fn=MLPDiffusion(d, hid), hid_dim=d).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0)

    def eager():
        loss = model(x)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    eager_ms = timed(eager)
    flops = 3 * batch * (4.0 * d * hid + 16.0 * hid * hid + 4.0 * hid * hid)      # fwd + 2 x bwd, incl. the per-row time embedding
    return {"batch": batch, "native_ms_per_step": native_ms, "pytorch_eager_ms_per_step": eager_ms, "speedup": eager_ms / native_ms,
            "float32_equivalent_tflops": {"native": flops / native_ms / 1e9, "pytorch_eager": flops / eager_ms / 1e9},
            "note": "native = dp_mstep_fwd_bwd (float32 operands split into bf16 hi + lo, three bf16 MFMA products per float32 product) "
                    "+ torch.optim.Adam(fused=True); gradients within 1e-4 of the reference's float32 autograd (tests/test_gpu_mstep.py)"}


def csrc_sha() -> str:
    """sha256 over the kernel sources: PMC_TRAFFIC.json is stamped with it (tools/pmc_summary.py --stamp) so counters
    measured on other kernels are never reported as this build's."""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(REPO, "diffputer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            text = open(os.path.join(d, f), encoding="utf-8").read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # code only: comments and layout do not change counters
            text = re.sub(r"//[^\n]*", "", text)
            h.update(f.encode())
            h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str, rows_per_launch: int):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
    WRITE_SIZE passes of their own, corrected as MI355X_MICROARCH.md §HBM prescribes; tools/pmc_summary.py).  Counters
    cannot be read inside a timed run, so this is the last measurement on file, valid for the same rows per launch."""
    path = os.path.join(REPO, "PMC_TRAFFIC.json")     # = the newest profiles/r*_pmc_summary.json (profiles/ does not travel to the GPU box)
    try:
        with open(path) as f:
            d = json.load(f)
        e = d.get(kernel) or d[kernel.replace("<true, 0>", "<true, false>")]
        if d.get("_csrc_sha") != csrc_sha():
            return None, f"PMC_TRAFFIC.json was collected on other kernel sources (sha {d.get('_csrc_sha')}, this build {csrc_sha()}): re-run the --pmc passes"
        if int(d.get("_rows_per_launch", 0)) != int(rows_per_launch):
            return None, f"PMC_TRAFFIC.json is for {d.get('_rows_per_launch')} rows per launch"
        return e["traffic_bytes"], "PMC_TRAFFIC.json: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, L2 fabric side (Infinity-Cache hits included)"
    except (OSError, KeyError, ValueError):
        return None, "no PMC summary on file"


CONFIGS = {   # BASELINE.json configs -> (rows per GPU per step, cols, trials, what it stands for)
    "synth10m": (32768, 256, 10, "BASELINE configs[4]: synthetic 10M x 256 table"),
    "synth1m": (125000, 64, 10, "BASELINE configs[3]: synthetic 1M x 64 table, one of 8 row shards"),
    "news": (27750, 51, 50, "BASELINE configs[2]: news-shaped table (27 750 train rows, 45 numeric + 6 binary-digit columns, synthetic stand-in)"),
    "adult": (22792, 34, 20, "BASELINE configs[1]: adult-shaped table (22 792 train rows, 6 numeric + 28 binary-digit columns, synthetic stand-in), E-step of one EM iteration"),
}


class StandInPlan:
    """--standin (tests only, never a measurement): a plan that does no sampling -- trial t of global row r "imputes"
    r + t/64 -- so the launcher, the rank wiring and the collectives can run on CPU ranks over gloo."""

    def __init__(self, device):
        self.device = device

    def impute_sum(self, x_obs, mask, num_trials, *, seed=0, row_offset=0, trial_offset=0, out=None, accumulate=False):
        rows = torch.arange(row_offset, row_offset + x_obs.shape[0], dtype=torch.float32, device=self.device)[:, None]
        add = sum(rows + (trial_offset + t) / 64.0 for t in range(num_trials)).expand_as(out)
        out.copy_(out + add if accumulate else add)
        return out, None

    def gemm_rows(self, n):
        return n


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """Start `n` ranks of this script under torch.distributed.run as a CHILD process (this process has not touched the GPU
    and never will), pass the child's stdout through, return its exit code (non-zero if any rank failed)."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("[bench] starting ranks: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="synth10m", choices=sorted(CONFIGS), help="BASELINE.json shape (rows/cols/trials defaults)")
    ap.add_argument("--rows", type=int, default=None, help="rows of the table per GPU per step")
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--hid", type=int, default=1024)
    ap.add_argument("--num-steps", type=int, default=50)
    ap.add_argument("--resample", type=int, default=20)
    ap.add_argument("--trials", type=int, default=None)
    ap.add_argument("--standin", action="store_true", help="tests only: CPU ranks over gloo with a plan that does no sampling")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--max-chains", type=int, default=1 << 16)
    ap.add_argument("--cpu-rows", type=int, default=1024, help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-mstep", action="store_true", help="skip the M-step training-step timing (SURVEY.md section 8 f1) beside the E-step line")
    ap.add_argument("--no-fold", action="store_true", help="run proj and mlp.0 as two GEMMs (dp_tune fold_proj=0)")
    ap.add_argument("--tune", action="append", default=[], help="dp_tune key=value (A/B runs), e.g. --tune ws_gemm=0")
    ap.add_argument("--m-chunk", type=int, default=None, help="rows per wide-layer GEMM launch (dp_tune m_chunk; 0 = whole batch)")
    a = ap.parse_args()
    preset = CONFIGS[a.config]
    a.rows = preset[0] if a.rows is None else a.rows
    a.cols = preset[1] if a.cols is None else a.cols
    a.trials = preset[2] if a.trials is None else a.trials

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus))                # nothing above this line initialises the GPU

    if min(a.rows, a.cols, a.trials, a.steps) < 1 or a.warmup < 0:
        raise SystemExit("--rows, --cols, --trials and --steps must be >= 1, --warmup >= 0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    if a.standin:
        dev, backend = torch.device("cpu"), "gloo"
    else:
        torch.cuda.set_device(local_rank)
        dev, backend = torch.device("cuda", local_rank), "nccl"
    dist = None
    use_dist = "WORLD_SIZE" in os.environ                    # under torch.distributed.run, also with one rank: the RCCL
    if use_dist:                                              # branch (init, all-gather, all-reduce) then runs on a 1-GPU box too
        import torch.distributed as dist
        dist.init_process_group(backend, **({} if a.standin else {"device_id": dev}))

    d, hid, S, R, T = a.cols, a.hid, a.num_steps, a.resample, a.trials
    folded = a.dtype == "bf16" and not a.no_fold
    if a.standin:
        plan = StandInPlan(dev)
        a.no_profile, a.cpu_rows = True, 0
    else:
        from diffputer_amd import _lib
        _lib.check(_lib.load().dp_tune(b"fold_proj", 0 if a.no_fold else 1), "dp_tune")
        if a.m_chunk is not None:
            _lib.check(_lib.load().dp_tune(b"m_chunk", a.m_chunk), "dp_tune")
        for kv in a.tune:
            k, v = kv.split("=")
            _lib.check(_lib.load().dp_tune(k.encode(), int(v)), "dp_tune")
        # random-init weights of the reference architecture (throughput is weight-independent), synthetic table shard
        sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(d, hid, seed=1, gain=1.0).items()}
        plan = SamplerPlan(sd, num_steps=S, n_resample=R, dtype=a.dtype, device=dev)
    X, mask = synth.make_table(a.rows, d, seed=1234 + rank)
    Xd = torch.from_numpy(X).to(dev)
    md = torch.from_numpy(mask).to(dev)
    n_total = a.rows * world
    row_offset = rank * a.rows

    def one_step(step_idx):
        local = estep.impute_trials(plan, Xd, md, T, seed=1000 + step_idx, row_offset=row_offset, max_chains=a.max_chains)
        if use_dist:
            return estep.gather_rows(local, n_total)
        return local

    def fence():
        if use_dist:
            dist.barrier()
        if not a.standin:
            torch.cuda.synchronize(dev)

    for w in range(a.warmup):
        one_step(-1 - w)
    if not a.no_profile:
        evals_per_step = nfe_per_chain(S, R) * len(list(estep.plan_calls(a.rows, T, a.max_chains)))
        plan.profile(max(1, (evals_per_step * a.steps) // 200))           # ~200 samples over the timed region
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_step(k)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    assert torch.isfinite(out).all(), "non-finite imputations"
    obs = ~md
    if world == 1 and not a.standin:
        assert torch.allclose(out[obs], Xd[obs], rtol=1e-6, atol=0), "observed cells must pass through"
    assert tuple(out.shape) == (n_total, d), "the (gathered) table must hold every rank's rows"
    ranks_seen = world
    if use_dist:                                              # what the process group itself reports, counted by a collective
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        assert ranks_seen == dist.get_world_size() == world

    rows_per_s = n_total * a.steps / dt
    nfe = nfe_per_chain(S, R)
    fl_row_eval = flops_per_row_eval(d, hid)
    whole_tflops = rows_per_s * T * nfe * fl_row_eval / 1e12 / world      # per GPU, the reference's FLOP model
    peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
    # FLOPs the GEMM kernels actually execute per row-evaluation (padded widths; proj folded into mlp.0 in bf16)
    H, K1 = hid, -(-d // 64) * 64
    exec_row_eval = (2.0 * 2 * H * K1 if folded else 2.0 * H * K1 + 2.0 * 2 * H * H) + 2.0 * 2 * H * 2 * H + 2.0 * H * 2 * H + 2.0 * K1 * H
    exec_tflops = rows_per_s * T * nfe * exec_row_eval / 1e12 / world

    result = {
        "metric": "imputed rows/sec (E-step, 50 steps x 10 trials)",
        "value": rows_per_s, "unit": "rows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.dtype, "data": "standin (no sampling: launcher/collective test only, NOT a measurement)" if a.standin else "synthetic",
        "ranks": ranks_seen, "backend": (("rccl (torch 'nccl')" if backend == "nccl" else backend) if use_dist else "none (single process)"),
        "config": {"workload": f"{preset[3]}, {d} cols, S={S} x R={R} x Heun, T={T} trials, "
                               f"hid={hid}; each step = full E-step of a {a.rows}-row shard per GPU (MCAR 30%, random-init weights)",
                   "name": a.config,
                   "rows_per_step_per_gpu": a.rows, "cols": d, "num_steps": S, "n_resample": R, "num_trials": T, "hid": hid,
                   "chains_per_call": max((r1 - r0) * (t1 - t0) for r0, r1, t0, t1 in estep.plan_calls(a.rows, T, a.max_chains)),
                   "wide_gemm_rows_per_launch": plan.gemm_rows(max((r1 - r0) * (t1 - t0) for r0, r1, t0, t1 in estep.plan_calls(a.rows, T, a.max_chains))),
                   "parallelism": f"row-shard x{world}, all-gather of trial means"},
        "row_trials_per_s": rows_per_s * T, "denoiser_row_evals_per_s": rows_per_s * T * nfe,
        # primary figures = FLOPs the GEMM kernels execute; the reference's FLOP model (5 layers, 4DH + 16H^2 per
        # row-evaluation) is the secondary "model_equivalent" pair: with proj folded into mlp.0 the same function costs fewer FLOPs
        "whole_path": {"tflops_per_gpu": exec_tflops, "frac_of_mfma_peak": exec_tflops / peak,
                       "executed_flops_per_row": T * nfe * exec_row_eval,
                       "model_equivalent_tflops_per_gpu": whole_tflops, "model_equivalent_frac_of_mfma_peak": whole_tflops / peak,
                       "model_flops_per_row": T * nfe * fl_row_eval,
                       "note": "tflops_per_gpu counts the GEMMs as launched (padded widths; proj folded into mlp.0 = 4 GEMMs per "
                               "evaluation); model_equivalent_* prices the same rows/s with the reference's 5-layer FLOP model",
                       "proj_folded_into_mlp0": folded},
    }

    if rank == 0 and not a.no_profile:
        ms, cnt = plan.profile_read()
        # rows per profiled launch: the calls of one step may differ in size (plan_calls), every call runs the same number of
        # evaluations and the samples are spread evenly over them, so time and FLOPs are both averaged over the calls
        call_chains = [(r1 - r0) * (t1 - t0) for r0, r1, t0, t1 in estep.plan_calls(a.rows, T, a.max_chains)]
        chains = sum(call_chains) / len(call_chains)
        Mp = sum(-(-c // 256) * 256 for c in call_chains) / len(call_chains)
        Mc = sum(plan.gemm_rows(c) for c in call_chains) / len(call_chains)   # wide layers: rows of the profiled (last) chunk
        N5 = K1
        layer_flops = [2.0 * Mc * H * K1, 2.0 * Mc * 2 * H * (K1 if folded else H), 2.0 * Mc * 2 * H * 2 * H, 2.0 * Mc * H * 2 * H,
                       2.0 * Mp * N5 * H]
        per = {}
        names = ["gemm_proj", "gemm_mlp0", "gemm_mlp2", "gemm_mlp4", "gemm_mlp6", "step_mid", "step_end", "step_final"]
        for i, nm in enumerate(names):
            if cnt[i]:
                per[nm] = {"avg_ms": ms[i] / cnt[i], "samples": cnt[i]}
                if i < 5:
                    per[nm]["tflops"] = layer_flops[i] / (ms[i] / cnt[i] * 1e-3) / 1e12
        # dominant kernel: the SiLU / bf16-out GEMM of the deep layers = the mlp.2 and mlp.4 launches (one of each per
        # evaluation; K = 2 hid) on the four-wave kernel, plus mlp.0 when it is not folded (K = hid); the folded proj+mlp.0
        # (K = padded table width) runs on the weight-stationary or the ping-pong kernel and is listed under "kernels".
        # achieved = executed FLOPs of those launches / their in-situ duration.  --tune gemm_variant=2: all three on the
        # ping-pong kernel, as before round 2's four-wave kernel.
        # The kernel of each slot is asked of the library (its own dispatch predicate under the current dp_tune settings).
        slot_kernel = [plan.gemm_kernel(i) for i in range(5)]
        pmc_names = {"k_gemm_bf16_w4": "dp::k_gemm_bf16_w4<true>", "k_gemm_bf16_pp": "dp::k_gemm_bf16_pp<true, 0>",
                     "k_gemm_bf16_ws": None, "k_gemm_f32": "dp::k_gemm_f32<true>"}
        dom_kernel = slot_kernel[2]                                        # mlp.2: half of the FLOPs
        dom = tuple(i for i in (1, 2, 3) if slot_kernel[i] == dom_kernel)
        dom_name = dom_kernel + "<SiLU>"
        pmc_name = pmc_names.get(dom_kernel)
        if all(cnt[i] for i in dom):
            avg_ms = sum(ms[i] / cnt[i] for i in dom) / len(dom)
            avg_flops = sum(layer_flops[i] for i in dom) / len(dom)
            ach = avg_flops / (avg_ms * 1e-3) / 1e12
            traffic, traffic_src = pmc_traffic(pmc_name, int(round(Mc))) if pmc_name else (None, "no PMC summary for this kernel")
            clock = {}
            if a.dtype == "bf16":
                cyc, us, n = plan.profile_clock()
                if n:
                    cus = torch.cuda.get_device_properties(dev).multi_processor_count
                    fpc = layer_flops[2] / (cyc / n) / cus                # mlp.2 launch FLOPs per shader cycle per CU
                    clock = {"clock_mhz_in_kernel": cyc / us, "clock_samples": n, "mlp2_kernel_cycles": cyc / n,
                             "mlp2_flop_per_cycle_per_cu": fpc, "mfma_peak_flop_per_cycle_per_cu": 4096,
                             "mlp2_frac_of_mfma_peak_at_clock_held": fpc / 4096.0,
                             "clock_note": "s_memtime / s_memrealtime of workgroup 0 of the sampled mlp.2 launches; 4096 = 4 SIMDs x "
                                           "1024 FLOP per cycle (v_mfma_f32_16x16x32_bf16: 16 384 FLOP per 16 cycles); the 2.5 PFLOP/s peak assumes 2.4 GHz"}
            result["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, **clock,
                                  "traffic_source": traffic_src,
                                  "kernel": dom_name + " (" + ", ".join({1: "proj+mlp.0" if folded else "mlp.0", 2: "mlp.2", 3: "mlp.4"}[i] for i in dom) + ")",
                                  "slot_kernels": dict(zip(["proj", "mlp0", "mlp2", "mlp4", "mlp6"], slot_kernel)),
                                  "avg_launch_ms": avg_ms, "alg_flops_per_launch": avg_flops, "rows_per_launch": Mc}
        # fused step kernel (after the 2nd evaluation): algorithmic bytes per cell = x_hat r/w 8 + d_cur 4 + F 4 + x_obs 4 + mask 1 + u 2
        if cnt[6]:
            cells = chains * d
            step_bytes = cells * (8 + 4 + 4 + 4 + 1 + (2 if a.dtype == "bf16" else 4))
            gbs = step_bytes / (ms[6] / cnt[6] * 1e-3) / 1e9
            result["fused_step"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                    "avg_launch_ms": ms[6] / cnt[6], "alg_bytes_per_launch": step_bytes}
        result["kernels"] = per

    if a.standin:                                             # never a throughput claim
        result["value"] = result["row_trials_per_s"] = result["denoiser_row_evals_per_s"] = 0.0
        del result["whole_path"]
    if rank == 0 and a.gpus == 1 and a.cpu_rows > 0:
        result["cpu_baseline"] = cpu_baseline(d, hid, S, R, T, a.cpu_rows, "")
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0 and a.gpus == 1 and not a.standin and not a.no_mstep and hid % 64 == 0:
        result["mstep"] = mstep_line(d, hid, dev)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
