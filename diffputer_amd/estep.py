"""The E-step driver: trial loop, final blend, trial mean -- and the row sharding across GPUs.

Counterpart of main.py:156-193 (in-sample) and :208-246 (out-of-sample).  Differences that do not
change results: the model is not rebuilt / model.pt not re-read per trial (main.py:170-173 is a
no-op semantically); trials are folded into the batch (chains = rows x trials) and summed on the
device in trial order instead of `torch.stack(...).mean(0)` (main.py:188).

Multi-GPU: rows are independent Markov chains given the weights, so each rank takes
a contiguous row shard and all trials of it; the only collective is one all-gather of the
trial-mean shards at the end (RCCL over xGMI when the process group is `nccl`).  Noise is
counter-based (global row, global trial), so the result is the same for any number of ranks.
"""
from __future__ import annotations

import torch

from .diffusion_utils import N_RESAMPLE, plan_for
from .sampler import SamplerPlan

DEFAULT_MAX_CHAINS = 1 << 16     # chains per dp_impute call: bounds the workspace (about 20 KB/chain in bf16)


def shard_bounds(n_rows: int, world_size: int, rank: int):
    """Contiguous, balanced row shard [r0, r1) of rank `rank` (first n % W ranks get one extra row)."""
    base, extra = divmod(n_rows, world_size)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


ROUND_CHAINS = 8192   # chains one round of the persistent GEMM covers at N = 2048: 256 CUs x 256-row tiles / 8 column tiles


def _call_cost(chains: int) -> int:
    """Relative cost of a dp_impute call: whole GEMM rounds, plus a fixed per-call part (launch-bound tail of every kernel)."""
    return -(-chains // ROUND_CHAINS) * ROUND_CHAINS + 1024


def plan_calls(n_rows: int, num_trials: int, max_chains: int):
    """Split rows x trials into rectangular dp_impute calls (r0, r1, t0, t1) of at most `max_chains` chains each.

    Chains are independent, so any partition gives the same numbers; what differs is how well each call fills the
    GPU: the GEMMs walk 256-row tiles in rounds of one tile per CU, so 32 770 chains cost five rounds where 32 768 cost
    four.  Among the splits into k equal row chunks x equal trial batches this picks the cheapest by `_call_cost`."""
    best = None
    for k in range(1, 65):
        rows = -(-n_rows // k)
        if rows > max_chains:
            continue
        per_max = max(1, min(num_trials, max_chains // rows))
        for per in {per_max, -(-num_trials // -(-num_trials // per_max))}:   # fullest batches / equal batches
            calls = []
            for r0 in range(0, n_rows, rows):
                r1 = min(n_rows, r0 + rows)
                for t0 in range(0, num_trials, per):
                    calls.append((r0, r1, t0, min(num_trials, t0 + per)))
            cost = sum(_call_cost((c[1] - c[0]) * (c[3] - c[2])) for c in calls)
            if best is None or cost < best[0]:
                best = (cost, calls)
        if rows * num_trials <= ROUND_CHAINS:
            break
    if best is None:                                              # more rows than chains per call even at k = 64
        calls = [(r0, min(n_rows, r0 + max_chains), t, t + 1) for r0 in range(0, n_rows, max_chains) for t in range(num_trials)]
        return iter(calls)
    return iter(best[1])


def impute_trials(plan: SamplerPlan, X, mask, num_trials: int, *, seed: int = 0, row_offset: int = 0,
                  max_chains: int = DEFAULT_MAX_CHAINS, prefilled: bool = False, trial_range=None, out=None) -> torch.Tensor:
    """Mean over `num_trials` trials of rec*mask + x_obs*(1-mask) for the rows given (main.py:160-188).
    X [n, D] float32 in the sampler's scale ((x-mean)/std/2, main.py:58); mask True = missing.
    trial_range=(t0, t1): only the global trials t0..t1-1 are run and their SUM is returned (trial sharding).
    out: optional contiguous float32 [n, D] tensor on the plan's device that receives the result (no allocation here)."""
    dev = plan.device
    # pandas-derived tables are often column-major: everything handed to the library must be row-major
    X = torch.as_tensor(X).to(dev, torch.float32).contiguous()
    m = torch.as_tensor(mask).to(dev).contiguous()
    x_obs = X if prefilled else ((1.0 - m.to(torch.float32)) * X)            # main.py:164
    m8 = m.to(torch.uint8).contiguous()
    if out is None:
        out = torch.zeros(x_obs.shape, dtype=torch.float32, device=dev)
    elif tuple(out.shape) != tuple(x_obs.shape) or out.dtype != torch.float32 or out.device != x_obs.device or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor of X's shape on the plan's device")
    else:
        out.zero_()
    n = x_obs.shape[0]
    if n == 0:
        return out
    tb, te = (0, num_trials) if trial_range is None else trial_range
    if te <= tb:
        return out
    for r0, r1, t0, t1 in plan_calls(n, te - tb, max_chains):
        plan.impute_sum(x_obs[r0:r1], m8[r0:r1], t1 - t0, seed=seed, row_offset=row_offset + r0, trial_offset=tb + t0,
                        out=out[r0:r1], accumulate=t0 > 0)
    if trial_range is None:
        out /= num_trials                                                    # the .mean(0) of main.py:188
    return out


def e_step(net, X, mask, num_trials: int = 20, num_steps: int = 50, device="cuda:0", *, seed: int = 0, dtype=None,
           n_resample: int = N_RESAMPLE, max_chains: int = DEFAULT_MAX_CHAINS, group=None, gather=True,
           shard: str = "rows", local_rows=None, n_rows: int | None = None) -> torch.Tensor:
    """Full E-step for one table: returns the trial-mean imputation [N, D] in the sampler's scale
    (the caller multiplies by 2 and saves iter_{k+1}.npy, main.py:190-193).

    With torch.distributed initialised, every rank samples its own contiguous row shard (`shard_bounds`).
      * X / mask are the FULL table on every rank (small tables, as main.py holds them), or -- `local_rows=(r0, r1)`,
        `n_rows=N` -- only this rank's rows [r0, r1) of an N-row table (the 10 M-row config: no rank ever holds the whole
        input; (r0, r1) must be this rank's `shard_bounds(N, world, rank)` unless gather=False);
      * gather=True: every rank receives the full [N, D] result (one all-gather); gather="rank0": only rank 0 does
        (the M-step trains on one rank; the others get None); gather=False: the local shard only.
    shard="trials" (few rows, many trials -- news with T = 50): every rank runs all rows for its own
    contiguous block of trials and the per-rank sums are all-reduced; the trial sum is then grouped per rank, so the
    result equals the row-sharded one to float32 rounding of that regrouping, not bit for bit."""
    import torch.distributed as dist

    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    plan = plan_for(net, num_steps, device, dtype, n_resample)
    if shard == "trials":
        if local_rows is not None:
            raise ValueError("shard='trials' runs every row on every rank: local_rows does not apply")
        t0, t1 = shard_bounds(num_trials, world, rank)
        total = impute_trials(plan, X, mask, num_trials, seed=seed, max_chains=max_chains, trial_range=(t0, t1))
        if world > 1:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        return total / num_trials
    if shard != "rows":
        raise ValueError(f"shard must be 'rows' or 'trials', not {shard!r}")
    if gather not in (True, False, "rank0"):
        raise ValueError("gather must be True, False or 'rank0'")
    if local_rows is None:
        n = X.shape[0]
        r0, r1 = shard_bounds(n, world, rank)
        X, mask = X[r0:r1], mask[r0:r1]
    else:
        r0, r1 = (int(v) for v in local_rows)
        if n_rows is None or not 0 <= r0 <= r1 <= n_rows or X.shape[0] != r1 - r0:
            raise ValueError(f"local_rows={local_rows} needs n_rows and an X of exactly r1 - r0 rows (got {X.shape[0]}, n_rows={n_rows})")
        n = int(n_rows)
        if gather and world > 1 and (r0, r1) != shard_bounds(n, world, rank):
            raise ValueError(f"rank {rank}: local_rows={local_rows} is not shard_bounds({n}, {world}, {rank}) = "
                             f"{shard_bounds(n, world, rank)}; the gather assumes that layout")
    if world == 1 or not gather:
        return impute_trials(plan, X, mask, num_trials, seed=seed, row_offset=r0, max_chains=max_chains)
    # the shard's result is written straight into the block the collective sends (shards differ by at most one row)
    d = X.shape[1]
    send = torch.zeros((-(-n // world), d), dtype=torch.float32, device=plan.device)
    impute_trials(plan, X, mask, num_trials, seed=seed, row_offset=r0, max_chains=max_chains, out=send[: r1 - r0])
    return gather_rows(send, n, group, padded=True, dst=0 if gather == "rank0" else None)


def gather_rows(local: torch.Tensor, n_rows: int, group=None, *, padded: bool = False, dst: int | None = None):
    """Gather of contiguous row shards (shard_bounds layout) into the full [n_rows, D] table: on every rank (dst=None, one
    all-gather) or on rank `dst` only (the others return None).

    Shards differ by at most one row, so every rank sends a block of ceil(n_rows / world) rows (`padded=True`: `local` is
    already such a block; else it is copied into one).  The collective writes into ONE buffer of world x that many rows,
    which is compacted in place (a shard moves down by at most `rank` rows) and returned as its first n_rows rows: beyond
    the result itself a rank holds its send block and, while compacting, one shard -- no second copy of the table."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    d = local.shape[1]
    biggest = -(-n_rows // world)
    if padded:
        if local.shape[0] != biggest:
            raise ValueError(f"padded send block must have {biggest} rows, not {local.shape[0]}")
        send = local.contiguous()
    else:
        send = torch.zeros((biggest, d), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    if dst is None:
        buf = torch.empty((world * biggest, d), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, send, group=group)
    else:
        buf = torch.empty((world * biggest, d), dtype=local.dtype, device=local.device) if rank == dst else None
        dist.gather(send, [buf[r * biggest:(r + 1) * biggest] for r in range(world)] if rank == dst else None,
                    dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
    if n_rows % world:
        for r in range(1, world):
            a, b = shard_bounds(n_rows, world, r)
            if a != r * biggest:
                buf[a:b] = buf[r * biggest: r * biggest + (b - a)].clone()   # ranges overlap: through a one-shard temporary
    return buf[:n_rows]
