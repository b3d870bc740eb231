"""CPU tests: the C-ABI library loads and exports what include/diffputer.h declares (no compute
calls without a GPU), host-side E-step logic, and the world_size-2 gather over gloo."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch

from conftest import REPO
from diffputer_amd import _lib, estep, synth
from diffputer_amd import model as dmodel


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    header = open(os.path.join(REPO, "include", "diffputer.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(dp_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.dp_version.restype = ctypes.c_int
    assert lib.dp_version() == 1


def test_argument_errors_without_gpu():
    """Argument validation that happens before any HIP call works on a CPU-only box."""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.dp_last_error.restype = ctypes.c_char_p
    h = ctypes.c_void_p()
    assert lib.dp_plan_create(ctypes.byref(h), 9, 100, 50, 20, 1) == -1            # hid % 64 != 0
    assert b"multiple of 64" in lib.dp_last_error()
    assert lib.dp_plan_create(ctypes.byref(h), 0, 64, 50, 20, 1) == -1
    assert lib.dp_plan_create(ctypes.byref(h), 9, 64, 1, 20, 1) == -1
    assert lib.dp_plan_create(ctypes.byref(h), 9, 64, 50, 20, 7) == -1
    assert h.value is None


def test_no_cpu_fallback():
    from diffputer_amd.sampler import SamplerPlan
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(9, 64, 0).items()}
    with pytest.raises(_lib.DiffputerError):
        SamplerPlan(sd, device="cpu")


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "diffputer_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("# checker only", ""), f


def test_shard_bounds_cover_rows_exactly():
    for n in (0, 1, 7, 8, 18389, 10_000_000):
        for w in (1, 2, 4, 8):
            spans = [estep.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_plan_calls_cover_rows_times_trials_once():
    # the last two have more rows than 64 calls of `mc` chains can hold: the planner's one-trial-per-call fallback
    for n, T, mc in ((18389, 20, 65536), (100, 7, 65536), (200000, 3, 65536), (65536, 2, 65536), (5, 50, 8),
                     (5_000_000, 2, 65536), (1000, 3, 8)):
        seen = np.zeros((n, T), dtype=np.int32)
        for r0, r1, t0, t1 in estep.plan_calls(n, T, mc):
            assert (r1 - r0) * (t1 - t0) <= mc
            seen[r0:r1, t0:t1] += 1
        assert (seen == 1).all()


def test_work_model_numbers():
    assert dmodel.num_parameters(9, 1024) == 10_512_393                 # SURVEY.md §5 [probed]
    assert dmodel.nfe_per_chain(50, 20) == 1961
    assert dmodel.flops_per_row_eval(256, 1024) == 16_777_216 + 4096 * 256
    assert abs(dmodel.nfe_per_chain(50, 20) * dmodel.flops_per_row_eval(256, 1024) / 3.4956e10 - 1) < 1e-4


def test_mirrored_modules_keep_checkpoint_keys():
    m = dmodel.Model(dmodel.MLPDiffusion(9, 64), hid_dim=9)
    keys = list(m.state_dict().keys())
    want = [synth.CKPT_PREFIX + f"{n}.{k}" for n, _, _ in synth.layer_shapes(9, 64) for k in ("weight", "bias")]
    assert keys == want
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(9, 64, 3).items()})
    loss = m(torch.randn(16, 9))
    assert loss.ndim == 0 and torch.isfinite(loss)


def test_mirrored_precond_matches_oracle():
    from oracle import diffputer_oracle as orc
    sd = synth.make_weights(12, 64, 4, 2.0)
    m = dmodel.Model(dmodel.MLPDiffusion(12, 64), hid_dim=12)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    x = np.random.default_rng(0).standard_normal((10, 12)).astype(np.float32)
    for s in (80.0, 1.3, 0.002):
        with torch.no_grad():
            got = m.denoise_fn_D(torch.from_numpy(x), torch.tensor(s)).numpy()
        assert np.allclose(got, orc.precond(sd, x, np.float32(s)), rtol=1e-5, atol=1e-5)


def test_synth_is_deterministic():
    a, ma = synth.make_table(1000, 17, seed=5)
    b, mb = synth.make_table(1000, 17, seed=5)
    assert np.array_equal(a, b) and np.array_equal(ma, mb)
    assert abs(float(ma.mean()) - 0.3) < 0.02 and abs(float(a.std()) - 0.5) < 0.02


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gather_worker(rank, world, port, n, d, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    full = torch.arange(n * d, dtype=torch.float32).reshape(n, d)
    r0, r1 = estep.shard_bounds(n, world, rank)
    got = estep.gather_rows(full[r0:r1].clone(), n)
    q.put((rank, bool(torch.equal(got, full))))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [11, 64])
def test_gather_rows_world_size_2_gloo(n):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, n, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


class _FakePlan:
    """Stands in for SamplerPlan on CPU: trial t of global row r "imputes" r + t/64 in every column, so the E-step
    driver's sharding, call planning, accumulation and collectives can be checked exactly without a GPU."""
    device = torch.device("cpu")

    def impute_sum(self, x_obs, mask, num_trials, *, seed=0, row_offset=0, trial_offset=0, out=None, accumulate=False):
        rows = torch.arange(row_offset, row_offset + x_obs.shape[0], dtype=torch.float32)[:, None]
        add = sum(rows + (trial_offset + t) / 64.0 for t in range(num_trials)).expand_as(out)
        if accumulate:
            out += add
        else:
            out.copy_(add)
        return out, None


def _estep_worker(rank, world, port, n, d, T, shard, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    estep.plan_for = lambda *a, **k: _FakePlan()
    X = torch.zeros(n, d)
    mask = torch.zeros(n, d, dtype=torch.bool)
    got = estep.e_step(None, X, mask, num_trials=T, device="cpu", max_chains=16, shard=shard)
    want = (torch.arange(n, dtype=torch.float32)[:, None] + sum(t / 64.0 for t in range(T)) / T).expand(n, d)
    q.put((rank, bool(torch.allclose(got, want, rtol=1e-6, atol=0)), tuple(got.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("shard", ["rows", "trials"])
def test_e_step_world_size_2_gloo(shard):
    """The N > 1 E-step path on CPU: two gloo ranks, row sharding (all-gather) and trial sharding (all-reduce)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n, d, T = 13, 3, 5
    procs = [ctx.Process(target=_estep_worker, args=(r, 2, port, n, d, T, shard, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, (n, d)), (1, True, (n, d))]


def _local_estep_worker(rank, world, port, n, d, T, gather, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    estep.plan_for = lambda *a, **k: _FakePlan()
    r0, r1 = estep.shard_bounds(n, world, rank)
    X = torch.zeros(r1 - r0, d)                          # this rank holds ONLY its own rows of the table
    mask = torch.zeros(r1 - r0, d, dtype=torch.bool)
    got = estep.e_step(None, X, mask, num_trials=T, device="cpu", max_chains=16, local_rows=(r0, r1), n_rows=n, gather=gather)
    want = (torch.arange(n, dtype=torch.float32)[:, None] + sum(t / 64.0 for t in range(T)) / T).expand(n, d)
    if gather == "rank0" and rank != 0:
        ok, shape = got is None, None
    else:
        ok, shape = bool(torch.allclose(got, want, rtol=1e-6, atol=0)), tuple(got.shape)
    bad = False
    if rank == 1:                                         # a shard that is not this rank's shard_bounds block is refused
        try:
            estep.e_step(None, X, mask, num_trials=T, device="cpu", local_rows=(r0 - 1, r1 - 1), n_rows=n)
        except ValueError:
            bad = True
    q.put((rank, ok, shape, bad))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,gather", [(13, True), (13, "rank0"), (12, True), (3, True)])
def test_e_step_shard_local_inputs_world_size_2_gloo(n, gather):
    """every rank passes only its own rows (uneven shards: 13 = 7 + 6); the shard's result is written into the
    block the collective sends and the gathered table is compacted in place -- on every rank, or on rank 0 only."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    d, T = 3, 5
    procs = [ctx.Process(target=_local_estep_worker, args=(r, 2, port, n, d, T, gather, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    other = None if gather == "rank0" else (n, d)
    assert res == [(0, True, (n, d), False), (1, True, other, True)]


def _slow_trainer_worker(rank, world, port, q):
    import datetime
    import time
    import torch.distributed as dist
    from diffputer_amd import em, prep
    from diffputer_amd.evalmetrics import mae_rmse
    # a process group whose collectives give up after 3 s, and an M-step that takes 7 s on rank 0
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=3))
    estep.plan_for = lambda *a, **k: _FakePlan()
    prep.masked_errors = lambda pred, truth, m, num_num=None, drop_rows=None: mae_rmse(pred.numpy(), truth.numpy(), m.numpy())
    real = em.m_step

    def slow(*a, **k):
        time.sleep(7.0)
        return real(*a, **k)

    em.m_step = slow
    X, mask = synth.make_table(32, 4, seed=1)
    t0 = time.time()
    try:
        res = em.run_em(X, mask, hid_dim=64, max_iter=1, num_trials=1, num_steps=5, device="cpu", max_epochs=1, log=lambda *_: None)
        q.put((rank, "ok", len(res.train_loss), time.time() - t0 >= 6.5))
    except Exception as e:                                # noqa: BLE001 -- reported to the parent
        q.put((rank, f"{type(e).__name__}: {e}"[:200], 0, False))
    dist.destroy_process_group()


def test_run_em_trainer_outlasts_the_process_group_timeout():
    """rank 0's M-step takes longer than the process group's collective timeout.  The other rank must
    not sit in a collective of that group meanwhile: it waits on em.trainer_wait_group (host side, week-long timeout)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slow_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "ok", 1, True), (1, "ok", 1, True)], res


def _em_worker(rank, world, port, ckpt_dir, q):
    import torch.distributed as dist
    from diffputer_amd import em, prep
    from diffputer_amd.evalmetrics import mae_rmse
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different initialisations: only a broadcast can make the ranks agree
    sums = []

    def fake_plan_for(net, *a, **k):
        sums.append(float(sum(p.double().sum() for p in net.parameters())))
        return _FakePlan()

    estep.plan_for = fake_plan_for
    prep.masked_errors = lambda pred, truth, m, num_num=None, drop_rows=None: mae_rmse(pred.numpy(), truth.numpy(), m.numpy())
    X, mask = synth.make_table(64, 4, seed=1)
    res = em.run_em(X, mask, hid_dim=64, max_iter=2, num_trials=2, num_steps=5, device="cpu", ckpt_dir=ckpt_dir, max_epochs=2,
                    log=lambda *_: None)
    q.put((rank, sums, res.train_loss))
    dist.destroy_process_group()


def test_run_em_world_size_2_gloo_one_trainer_one_writer(tmp_path):
    """em.run_em under torch.distributed: rank 0 trains, its weights and loss are broadcast (every rank samples its row shard
    from the SAME model), and only rank 0 writes model.pt / iter_k.npy."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_em_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    (_, sums0, loss0), (_, sums1, loss1) = res
    assert len(sums0) == 2 and sums0 == sums1 and loss0 == loss1, res
    assert sorted(os.listdir(tmp_path)) == ["0", "1", "iter_1.npy", "iter_2.npy"]
    assert os.listdir(tmp_path / "0") == ["model.pt"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's command form) launches two ranks itself
    under torch.distributed.run, and rank 0's JSON line reports what the process group saw.  CPU ranks over gloo with
    bench.py's stand-in plan (no sampling): this covers the launcher, the rank wiring, the row-offset arithmetic and the
    all-gather inside the timed step -- the RCCL run itself is the driver's."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--standin", "--steps", "2", "--warmup", "1",
                        "--rows", "37", "--cols", "5", "--trials", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["backend"] == "gloo" and j["steps"] == 2 and j["warmup"] == 1
    assert j["value"] == 0.0 and "standin" in j["data"] and j["scaling"] == "weak"
    # a failing rank makes the launcher exit non-zero
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--standin", "--rows", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_mixed_table_layout_matches_reference_load_dataset():
    """a9 pinned to the reference: tests/golden/layout_mixed.npz holds what `dataset.load_dataset` (dataset.py:10-159) and
    `mean_std` made of a small synthetic mixed table with interleaved numeric / categorical / target columns
    (tools/make_layout_golden.py).  layout.load_mixed must reproduce train_X / test_X, the extended masks, the category
    indices and cat_bin_num exactly -- including first-appearance category order, integer-valued categories mapped through
    str(), and the single-valued column that still gets one digit."""
    from conftest import load_golden
    from diffputer_amd import layout
    from diffputer_amd.evalmetrics import mean_std
    g = load_golden("layout_mixed")
    ni, ci = list(g["num_col_idx"]), list(g["cat_col_idx"])
    data_cols = {j: g[f"data_num_{j}"] for j in ni} | {j: g[f"data_cat_{j}"] for j in ci}
    for tag in ("train", "test"):
        cols = {j: g[f"{tag}_num_{j}"] for j in ni} | {j: g[f"{tag}_cat_{j}"] for j in ci}
        X, m, cat_idx, bins = layout.load_mixed(data_cols, cols, g[f"{tag}_mask0"], ni, ci)
        assert X.dtype == np.float32 and np.array_equal(X, g[f"{tag}_X"])
        assert m.dtype == bool and np.array_equal(m, g[f"{tag}_mask"])
        assert np.array_equal(cat_idx, g[f"{tag}_cat_idx"]) and np.array_equal(bins, g["cat_bin_num"])
        assert np.array_equal(layout.decode_bits(X[:, len(ni):], bins), cat_idx)
    assert list(g["cat_bin_num"]) == [4, 3, 1, 1]
    X, m, _, _ = layout.load_mixed(data_cols, {j: g[f"train_num_{j}"] for j in ni} | {j: g[f"train_cat_{j}"] for j in ci},
                                   g["train_mask0"], ni, ci)
    mean, std = mean_std(X, m)
    assert np.allclose(mean, g["mean"], rtol=1e-6, atol=0) and np.allclose(std, g["std"], rtol=1e-6, atol=1e-12)
    with pytest.raises(ValueError):
        layout.category_index(["a", "b"], ["c"])


def mstep_loss_and_grads(g, device):
    """diffputer_amd.model.Model on `device` with the fixture's weights and its two Gaussian draws replayed:
    (loss, per-cell loss, gradients by short name).  Shared with the GPU test."""
    sd = synth.make_weights(int(g["d_in"]), int(g["hid"]), int(g["wseed"]), float(g["gain"]))
    m = dmodel.Model(dmodel.MLPDiffusion(int(g["d_in"]), int(g["hid"])), hid_dim=int(g["d_in"])).to(device)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.train()
    x = torch.from_numpy(g["x"]).to(device)
    zs, zn = torch.from_numpy(g["z_sigma"]).to(device), torch.from_numpy(g["z_noise"]).to(device)
    saved = (torch.randn, torch.randn_like)
    torch.randn = lambda *a, **k: zs.clone()
    torch.randn_like = lambda t, **k: zn.clone()
    try:
        percell = m.loss_fn(m.denoise_fn_D, x).detach()
        loss = m(x)
    finally:
        torch.randn, torch.randn_like = saved
    loss.backward()
    grads = {k[len(synth.CKPT_PREFIX):]: v.grad for k, v in m.named_parameters()}
    return loss.detach(), percell, grads


def check_mstep(g, device, rtol):
    loss, percell, grads = mstep_loss_and_grads(g, device)
    assert abs(float(loss) - float(g["loss"])) <= rtol * float(g["loss"]), (float(loss), float(g["loss"]))
    pc = percell.cpu().numpy()
    assert np.all(np.abs(pc - g["percell"]) <= rtol * np.maximum(np.abs(g["percell"]), float(np.abs(g["percell"]).mean())))
    for name in ("proj.weight", "mlp.2.weight", "time_embed.0.weight", "mlp.6.bias"):
        ref = g["grad__" + name]
        got = grads[name].cpu().numpy()
        scale = float(np.abs(ref).max())
        assert float(np.abs(got - ref).max()) <= rtol * scale, (name, float(np.abs(got - ref).max()), scale)


def test_mstep_loss_and_gradient_match_reference():
    """f1 pinned: tests/golden/mstep_D12_H64.npz holds the reference's Model.forward (model.py:171-181 -> EDMLoss,
    diffusion_utils.py:200-215 -> Precond.forward with per-row sigma) and its gradients on one batch with replayed draws
    (tools/make_mstep_golden.py).  The mirrored modules reproduce loss, per-cell loss and gradients to 1e-5 relative."""
    from conftest import load_golden
    check_mstep(load_golden("mstep_D12_H64"), "cpu", 1e-5)


def test_pingpong_gemm_counted_waits_match_the_emitted_code():
    """tools/check_pp_asm.py: the vmcnt(8 + NST) / vmcnt(10 + NST) waits of k_gemm_bf16_pp assume NST epilogue stores sit
    behind the next tile's prologue DMA; compile the kernel and count them in the gfx950 assembly (also: no scratch).
    The same for k_gemm_bf16_w4: 4 DMA instructions per phase, 32 epilogue stores, vmcnt(24) / vmcnt(56), every M0 write an
    MFMA ahead of its DMA, no scratch access, no vmcnt(0) inside the tile loop."""
    import subprocess
    import sys
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_pp_asm.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_dataset_module_reads_files_like_the_reference(tmp_path):
    """diffputer_amd.dataset.load_dataset / get_eval with the reference's signatures, on CSV / JSON / .npy files written from
    the columns tests/golden/layout_mixed.npz stores (the same table the reference's load_dataset read when the fixture
    was made): the 11-tuple must equal the reference's arrays; the numeric-only branch is exercised on a second table."""
    import json
    import pandas as pd
    from conftest import load_golden
    from diffputer_amd import dataset as ds
    g = load_golden("layout_mixed")
    ni, ci = [int(j) for j in g["num_col_idx"]], [int(j) for j in g["cat_col_idx"]]
    ncol = max(ni + ci) + 2                                            # + the target column
    names = [f"c{j}" for j in range(ncol)]
    root = tmp_path / "datasets"
    (root / "mixed" / "masks" / "rate30" / "MCAR").mkdir(parents=True)
    (root / "Info").mkdir()
    for tag in ("data", "train", "test"):
        n = len(g[f"{tag}_num_{ni[0]}"])
        df = {names[j]: g[f"{tag}_num_{j}"] for j in ni} | {names[j]: g[f"{tag}_cat_{j}"] for j in ci}
        df[names[ncol - 1]] = np.zeros(n)
        pd.DataFrame({c: df[c] for c in names if c in df}, columns=names).to_csv(root / "mixed" / f"{tag}.csv", index=False)
    json.dump({"name": "mixed", "num_col_idx": ni, "cat_col_idx": ci, "target_col_idx": [ncol - 1]}, open(root / "Info" / "mixed.json", "w"))
    np.save(root / "mixed" / "masks" / "rate30" / "MCAR" / "train_mask_0.npy", g["train_mask0"])
    np.save(root / "mixed" / "masks" / "rate30" / "MCAR" / "test_mask_0.npy", g["test_mask0"])
    out = ds.load_dataset("mixed", 0, "MCAR", "30", data_dir=str(root))
    want = [g[k] for k in ("train_X", "test_X", "train_mask0", "test_mask0", "train_num", "test_num", "train_cat_idx", "test_cat_idx",
                           "train_mask", "test_mask", "cat_bin_num")]
    assert len(out) == 11
    for got, ref in zip(out, want):
        assert np.array_equal(got, ref) and got.dtype == ref.dtype
    mean, std = ds.mean_std(out[0], out[8])
    assert np.allclose(mean, g["mean"], rtol=1e-6, atol=0) and np.allclose(std, g["std"], rtol=1e-6, atol=1e-12)
    # get_eval: the mask over the ORIGINAL columns is sliced by num_col_idx; a zero prediction scores the numeric truth itself
    Xn = np.where(std > 0, (out[0] - mean) / np.where(std > 0, std, 1), 0).astype(np.float32)
    mae, rmse = ds.get_eval("mixed", np.zeros_like(Xn), Xn, out[6], len(ni), out[10], out[2], data_dir=str(root))
    nm = g["train_mask0"][:, ni]
    assert abs(mae - float(np.abs(Xn[:, :len(ni)][nm]).mean())) < 1e-6
    # numeric-only table: categorical entries are None, the extended mask is the numeric slice (dataset.py:61-66)
    json.dump({"name": "mixed", "num_col_idx": ni, "cat_col_idx": [], "target_col_idx": [ncol - 1]}, open(root / "Info" / "mixed.json", "w"))
    o2 = ds.load_dataset("mixed", 0, "MCAR", "30", data_dir=str(root))
    assert o2[6] is None and o2[7] is None and o2[10] is None
    assert np.array_equal(o2[0], g["train_num"]) and np.array_equal(o2[8], g["train_mask0"][:, ni])


def test_prep_helpers_match_reference():
    """evalmetrics.mean_std / mae_rmse against the reference's mean_std (dataset.py:161-169) and get_eval (:171-206)
    run on 2 048 california rows (tools/make_prep_golden.py)."""
    from conftest import load_golden
    from diffputer_amd.evalmetrics import mae_rmse, mean_std
    g = load_golden("prep_california")
    mean, std = mean_std(g["X"], g["mask"])
    assert np.allclose(mean, g["mean"], rtol=1e-6, atol=0) and np.allclose(std, g["std"], rtol=1e-6, atol=0)
    mae, rmse = mae_rmse(g["pred"], g["Xn"] * 2, g["ori_mask"], num_num=g["X"].shape[1])
    assert abs(mae - float(g["mae"])) < 1e-6 and abs(rmse - float(g["rmse"])) < 1e-6


def test_committed_bench_line_keeps_the_contract():
    """The last bench line committed under profiles/ (written by `python bench.py` on the GPU box) carries every key the
    driver and the judge read, with the roofline arithmetic consistent."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = sorted(p for p in glob.glob(os.path.join(root, "profiles", "r*_bench.json")) if "rocprof" not in p)
    if not os.path.isdir(os.path.join(root, "profiles")):
        pytest.skip("profiles/ does not travel to the GPU box (.gpurunignore)")
    assert paths, "no bench line under profiles/"
    r = json.loads(open(paths[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "ranks", "backend", "whole_path"):
        assert k in r, k
    assert r["ranks"] == r["n_gpus"]
    wp = r["whole_path"]                                   # executed FLOPs first; the reference's 5-layer model second
    assert wp["frac_of_mfma_peak"] <= wp["model_equivalent_frac_of_mfma_peak"] and "executed_flops_per_row" in wp
    assert r["unit"] == "rows/s" and r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "bf16" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert abs(rf["achieved"] - rf["alg_flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * rf["achieved"]
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "rows/s" and cb["sample"]
    rows = r["config"]["rows_per_step_per_gpu"] * r["n_gpus"]
    assert abs(r["value"] - rows / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]


def test_mixed_table_layout():
    """layout.py against the rules of dataset.py:80-82 (digits = '0{n}b' of the category index, MSB first) and :140-157
    (a categorical column's mask bit tiled over its digits), spelled out the slow way."""
    from diffputer_amd import layout
    rng = np.random.default_rng(5)
    n_cat = [9, 16, 7, 2, 42, 1]                                     # adult's column sizes among them
    assert [layout.bits_for(n) for n in n_cat] == [4, 4, 3, 1, 6, 1]  # a single-valued column still prints one digit: '00b' -> '0'
    N, k = 50, 3
    num = rng.standard_normal((N, k)).astype(np.float32)
    cat = np.stack([rng.integers(0, n, N) for n in n_cat], axis=1)
    X, bins = layout.encode_table(num, cat, n_cat)
    assert X.dtype == np.float32 and X.shape == (N, k + 19) and list(bins) == [4, 4, 3, 1, 6, 1]
    for r in range(N):
        want = "".join(format(int(cat[r, j]), "0" + str((n - 1).bit_length()) + "b") for j, n in enumerate(n_cat))   # dataset.py:80-82
        assert "".join(str(int(v)) for v in X[r, k:]) == want
    assert np.array_equal(X[:, :k], num)
    mask = rng.random((N, k + len(n_cat))) < 0.3
    ext = layout.extend_mask(mask, k, bins)
    assert ext.shape == X.shape and np.array_equal(ext[:, :k], mask[:, :k])
    o = k
    for j, b in enumerate(bins):
        for d in range(b):
            assert np.array_equal(ext[:, o + d], mask[:, k + j])
        o += b
    noisy = X[:, k:] + rng.uniform(-0.4, 0.4, X[:, k:].shape)         # an imputation near the digits decodes to the category
    assert np.array_equal(layout.decode_bits(noisy, bins), cat)
    with pytest.raises(ValueError):
        layout.encode_bits(np.array([16]), 4)


def test_native_training_step_host_checks():
    """The M-step's native entry points without a GPU: dp_mstep_workspace_bytes is host arithmetic (it must grow with the
    batch, be 256-byte granular and refuse shapes the kernels cannot take), and NativeTrainStep refuses a CPU model loudly --
    there is no CPU fallback behind it (autograd on `model.Model` is the CPU path)."""
    from diffputer_amd import _lib
    from diffputer_amd.mstep import NativeTrainStep
    lib = _lib.load()
    a = lib.dp_mstep_workspace_bytes(34, 1024, 4096)
    b = lib.dp_mstep_workspace_bytes(34, 1024, 8192)
    assert 0 < a < b and a % 256 == 0
    assert lib.dp_mstep_workspace_bytes(34, 1024, 4095) == a                      # rows are padded to 256
    assert lib.dp_mstep_workspace_bytes(34, 100, 4096) == 0 and b"bad shape" in lib.dp_last_error()
    assert lib.dp_mstep_workspace_bytes(0, 1024, 4096) == 0
    m = dmodel.Model(dmodel.MLPDiffusion(12, 64), hid_dim=12)
    with pytest.raises(_lib.DiffputerError):
        NativeTrainStep(m)
