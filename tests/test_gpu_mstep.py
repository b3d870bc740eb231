"""SURVEY.md §8 f1 on the GPU: the native training step (`dp_mstep_fwd_bwd`, through the C ABI) against the REFERENCE's
`Model.forward` / `loss.backward()` run on CPU with the same two Gaussian draws replayed (tools/make_mstep_golden.py).

Tolerance, stated from the arithmetic (dp_mstep.h): GEMM operands are split into bf16 hi + lo and multiplied as three bf16
products with float32 accumulation -- ~1e-5 relative per product -- so loss, per-cell loss and every one of the 14 gradients
are held to 1e-4 of their scale (max |reference gradient| of that tensor); the CPU oracle (oracle/diffputer_oracle_mstep.py,
pinned to the same fixtures at 1e-5) and float32 autograd on the same device are the checkers at sizes the fixtures do not cover (a batch that is not a multiple of 256, D = 9).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from diffputer_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAYERS = ["proj", "mlp.0", "mlp.2", "mlp.4", "mlp.6", "time_embed.0", "time_embed.2"]


def _model(d, hid, wseed):
    from diffputer_amd.model import MLPDiffusion, Model
    sd = synth.make_weights(d, hid, wseed, 1.0)
    m = Model(denoise_fn=MLPDiffusion(d, hid), hid_dim=d).to(DEV)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.train()


def _grads(model):
    return {k[len(synth.CKPT_PREFIX):]: v.grad.detach().clone() for k, v in model.named_parameters()}


def _autograd(model, x, zs, zn):
    """float32 autograd of diffputer_amd.model.Model with the two draws replayed (the checker of tests/test_host_logic.py)."""
    model.zero_grad(set_to_none=True)
    saved = (torch.randn, torch.randn_like)
    torch.randn = lambda *a, **k: zs.clone()
    torch.randn_like = lambda t, **k: zn.clone()
    try:
        loss = model(x)
    finally:
        torch.randn, torch.randn_like = saved
    loss.backward()
    return loss.detach(), _grads(model)


def test_small_fixture_all_14_gradients():
    """tests/golden/mstep_D12_H64.npz: D = 12, H = 64, B = 48 (every dimension far from a multiple of 256: all padding paths),
    loss, per-cell loss and all 14 gradients of the reference."""
    from diffputer_amd.mstep import NativeTrainStep
    g = load_golden("mstep_D12_H64")
    model = _model(int(g["d_in"]), int(g["hid"]), int(g["wseed"]))
    step = NativeTrainStep(model)
    x, zs, zn = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "z_sigma", "z_noise"))
    loss, pc = step.loss_and_grads(x, zs, zn, percell=True)
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * float(g["loss"]), (float(loss), float(g["loss"]))
    ref = g["percell"]
    assert np.all(np.abs(pc.cpu().numpy() - ref) <= 1e-4 * np.maximum(np.abs(ref), float(np.abs(ref).mean())))
    got = _grads(model)
    worst = 0.0
    for name in LAYERS:
        for kind in ("weight", "bias"):
            r = g[f"grad__{name}.{kind}"]
            e = float(np.abs(got[f"{name}.{kind}"].cpu().numpy() - r).max()) / float(np.abs(r).max())
            worst = max(worst, e)
            assert e <= 1e-4, (name, kind, e)
    print(f"native M-step vs reference, D=12 H=64 B=48: worst gradient error {worst:.2e} of the tensor's scale")
    loss2 = step.loss_and_grads(x, zs, zn).clone()                      # deterministic: same inputs, same bits
    again = _grads(model)
    assert float(loss2) == float(loss) and all(torch.equal(again[k], got[k]) for k in got)


def test_wide_fixture_reference_width_and_batch():
    """tests/golden/mstep_D34_H1024.npz: the reference's real width and batch (H = 1024, B = 4096, main.py:25,92) at adult's
    D = 34.  Inputs are regenerated from the fixture's seeds (checksums verified); the fixture holds the reference's loss,
    per-row loss, the 7 bias gradients, and for every weight gradient its [::37, ::41] sample, Frobenius norm and maximum."""
    from diffputer_amd.mstep import NativeTrainStep
    g = load_golden("mstep_D34_H1024")
    d, hid, B = int(g["d_in"]), int(g["hid"]), int(g["B"])
    X, mask = synth.make_table(B, d, seed=int(g["tseed"]))
    x = (X * (1.0 - mask)).astype(np.float32)
    rng = np.random.Generator(np.random.Philox(int(g["zseed"])))
    zs = rng.standard_normal(B, dtype=np.float32)
    zn = rng.standard_normal((B, d), dtype=np.float32)
    assert float(x.astype(np.float64).sum()) == float(g["x_sum"])
    assert float(zn.astype(np.float64).sum() + zs.astype(np.float64).sum()) == float(g["z_sum"])
    model = _model(d, hid, int(g["wseed"]))
    step = NativeTrainStep(model)
    loss, pc = step.loss_and_grads(torch.from_numpy(x).to(DEV), torch.from_numpy(zs).to(DEV), torch.from_numpy(zn).to(DEV), percell=True)
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * float(g["loss"]), (float(loss), float(g["loss"]))
    rl = pc.mean(1).cpu().numpy()
    assert np.all(np.abs(rl - g["rowloss"]) <= 1e-4 * np.maximum(np.abs(g["rowloss"]), float(g["rowloss"].mean())))
    got = _grads(model)
    s0, s1 = (int(v) for v in g["stride"])
    worst = 0.0
    for name in LAYERS:
        r = g[f"grad__{name}.bias"]
        e = float(np.abs(got[f"{name}.bias"].cpu().numpy() - r).max()) / float(np.abs(r).max())
        w = got[f"{name}.weight"]
        scale = float(g[f"gmax__{name}.weight"])
        es = float(np.abs(w[::s0, ::s1].cpu().numpy() - g[f"gsample__{name}.weight"]).max()) / scale
        en = abs(float(w.double().pow(2).sum().sqrt()) - float(g[f"gnorm__{name}.weight"])) / float(g[f"gnorm__{name}.weight"])
        assert abs(float(w.abs().max()) - scale) <= 1e-4 * scale
        worst = max(worst, e, es, en)
        assert e <= 1e-4 and es <= 1e-4 and en <= 1e-4, (name, e, es, en)
    print(f"native M-step vs reference, D=34 H=1024 B=4096: worst gradient error {worst:.2e} of the tensor's scale")


@pytest.mark.parametrize("d,hid,B", [(9, 256, 2312), (51, 128, 77)])
def test_against_float32_autograd_on_ragged_shapes(d, hid, B):
    """Shapes no fixture covers -- the last batch of an epoch (22 792 mod 4 096 = 2 312 rows), california's and news' widths --
    against float32 autograd of the mirrored modules on the same device, same draws."""
    from diffputer_amd.mstep import NativeTrainStep
    model = _model(d, hid, 55)
    gen = torch.Generator(device=DEV).manual_seed(d * 1000 + B)
    x = torch.randn(B, d, device=DEV, generator=gen) * 0.5
    zs, zn = torch.randn(B, device=DEV, generator=gen), torch.randn(B, d, device=DEV, generator=gen)
    ref_loss, ref = _autograd(model, x, zs, zn)
    loss = NativeTrainStep(model).loss_and_grads(x, zs, zn).clone()
    got = _grads(model)
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * float(ref_loss)
    for k in ref:
        scale = float(ref[k].abs().max())
        assert float((got[k] - ref[k]).abs().max()) <= 1e-4 * scale, (k, float((got[k] - ref[k]).abs().max()), scale)
    # ... and against the CPU oracle (numpy, hand-written backward; pinned to the reference in tests/test_oracle_golden.py)
    from oracle import diffputer_oracle_mstep as orm
    sd = synth.make_weights(d, hid, 55, 1.0)
    o_loss, _, o_grads = orm.loss_and_grads(sd, x.cpu().numpy(), zs.cpu().numpy(), zn.cpu().numpy())
    assert abs(float(loss) - float(o_loss)) <= 1e-4 * float(o_loss)
    for k, r in o_grads.items():
        assert float(np.abs(got[k].cpu().numpy() - r).max()) <= 1e-4 * float(np.abs(r).max()), k


def test_training_with_the_native_step_follows_autograd():
    """main.py:119-133 for 40 Adam steps from the same initialisation with the same draws replayed every step: the loss curve
    of the native step stays within 1e-3 of float32 autograd's (the two differ by ~1e-5 per gradient; Adam normalises
    gradients, so tiny differences may flip the sign of near-zero components -- the curves, not the bits, must agree)."""
    from diffputer_amd.mstep import NativeTrainStep
    d, hid, B = 9, 256, 1024
    X, mask = synth.make_table(B, d, seed=5)
    x = torch.from_numpy((X * (1.0 - mask)).astype(np.float32)).to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(11)
    zs, zn = torch.randn(B, device=DEV, generator=gen), torch.randn(B, d, device=DEV, generator=gen)
    curves = []
    for native in (False, True):
        model = _model(d, hid, 77)
        opt = torch.optim.Adam(model.parameters(), lr=3e-4, weight_decay=0)
        step = NativeTrainStep(model) if native else None
        losses = []
        for _ in range(40):
            losses.append(float(step.loss_and_grads(x, zs, zn)) if native else float(_autograd(model, x, zs, zn)[0]))
            opt.step()
        curves.append(np.asarray(losses))
    assert curves[0][-1] < 0.95 * curves[0][0], curves[0][[0, -1]]         # it trains
    assert np.all(np.abs(curves[1] - curves[0]) <= 1e-3 * curves[0]), np.abs(curves[1] / curves[0] - 1).max()


def test_m_step_loop_native_and_autograd_reach_the_same_loss():
    """em.m_step (main.py:92-152: shuffled batches, Adam, ReduceLROnPlateau, best-epoch checkpoint) with the HIP training step
    and with PyTorch autograd, same seeds: 25 epochs over 9 000 x 8 rows in batches of 4 096 (the last batch of an epoch
    has 808 rows).  Both must train and end within 1 % of each other's best loss."""
    from diffputer_amd import em
    X, mask = synth.make_table(9000, 8, seed=31, latent=2)
    data = torch.from_numpy((X * (1.0 - mask)).astype(np.float32)).to(DEV)
    best = {}
    for native in (False, True):
        torch.manual_seed(123)                                            # initialisation and the loss's draws
        gen = torch.Generator(device=DEV).manual_seed(9)                  # the shuffles
        model, b = em.m_step(data, 256, max_epochs=25, generator=gen, native=native)
        assert all(torch.isfinite(p).all() for p in model.parameters())
        best[native] = b
    assert best[True] < 1.0 and abs(best[True] - best[False]) <= 1e-2 * best[False], best
