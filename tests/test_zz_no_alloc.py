"""dp_impute_static (include/diffputer.h): the sampler on the library's own scratch array, for callers that cannot
allocate.  It has no counterpart in the reference; its contract is "the same bits as dp_impute", checked here through the
C ABI on two calls in a row, the second with a smaller input (the scratch keeps the first call's contents), in Philox and
in replay mode, float32 and bf16, plus the size limit.  Collected last (test_zz_*): the round's end runs `pytest -x`."""
import ctypes as C

import numpy as np
import pytest
import torch

from diffputer_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _plan(d, hid, S, R, dtype, seed=3):
    from diffputer_amd.sampler import SamplerPlan
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(d, hid, seed=seed, gain=1.0).items()}
    return SamplerPlan(sd, num_steps=S, n_resample=R, dtype=dtype, device=DEV)


@pytest.mark.parametrize("dtype,d,hid", [("bf16", 256, 256), ("f32", 9, 64), ("bf16", 12, 1024)])
def test_static_scratch_equals_workspace_call(dtype, d, hid):
    S, R = 6, 3
    p = _plan(d, hid, S, R, dtype)
    for n, T, seed in ((300, 4, 11), (37, 2, 12)):                       # second call smaller: stale scratch beyond it
        X, mask = synth.make_table(n, d, seed=seed)
        x_obs = torch.from_numpy((X * (1.0 - mask)).astype(np.float32)).to(DEV)
        m8 = torch.from_numpy(mask).to(DEV).to(torch.uint8)
        want, _ = p.impute_sum(x_obs, m8, T, seed=seed, row_offset=5, trial_offset=1)
        got = torch.full_like(want, float("nan"))
        p.impute_sum(x_obs, m8, T, seed=seed, row_offset=5, trial_offset=1, out=got, static_scratch=True)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all()
        assert torch.equal(got, want), (dtype, n, float((got - want).abs().max()))
    # replay mode with a trace, DP_ACCUMULATE and DP_NO_FINAL_LEVEL through the same entry point
    n, T = 9, 2
    X, mask = synth.make_table(n, d, seed=21)
    x_obs = (X * (1.0 - mask)).astype(np.float32)
    for final in (True, False):
        nd = 1 + 4 * R * (S - 1) + (3 if final else 0)
        noise = np.stack([synth.reference_noise(30 + t, nd, (n, d)) for t in range(T)])
        base = torch.ones(n, d, device=DEV)
        want, tr_w = p.impute_sum(x_obs, mask, T, noise=noise, trace=True, out=base.clone(), accumulate=True, final_level=final)
        got, tr_g = p.impute_sum(x_obs, mask, T, noise=noise, trace=True, out=base.clone(), accumulate=True, final_level=final,
                                 static_scratch=True)
        assert torch.equal(got, want) and torch.equal(tr_g, tr_w)
    p.close()


def test_static_scratch_rejects_oversize_and_allocates_nothing():
    p = _plan(256, 1024, 4, 2, "bf16")
    lib = _lib.load()
    fits = 0
    while p.workspace_bytes(fits + 256) <= _lib.DP_STATIC_SCRATCH_BYTES:
        fits += 256
    assert fits == 2816                                                  # the figure include/diffputer.h quotes
    n = fits + 1
    x = torch.zeros(n, 256, device=DEV)
    m = torch.ones(n, 256, device=DEV, dtype=torch.uint8)
    out = torch.empty_like(x)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.dp_impute_static(p._h, x.data_ptr(), m.data_ptr(), out.data_ptr(), n, 1, 0, 0, 0, None, None, 0, st)
    assert rc == -4 and b"static" in lib.dp_last_error()
    # the largest call that fits runs, and device memory in use does not move across calls (no allocation on the path)
    x, m, out = x[:fits], m[:fits], out[:fits]
    rc = lib.dp_impute_static(p._h, x.data_ptr(), m.data_ptr(), out.data_ptr(), fits, 1, 0, 0, 0, None, None, 0, st)
    assert rc == 0, lib.dp_last_error()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        rc = lib.dp_impute_static(p._h, x.data_ptr(), m.data_ptr(), out.data_ptr(), fits, 1, 0, 0, 0, None, None, 0, st)
        assert rc == 0
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0
    want, _ = p.impute_sum(x, m, 1)
    assert torch.equal(out, want)
    p.close()
