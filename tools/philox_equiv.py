"""Philox mode of dp_impute == replay mode fed the same numbers (dp_noise_field laid out in the reference's draw order)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from diffputer_amd import synth
from diffputer_amd.sampler import SamplerPlan, n_draws, noise_field
d, hid, S, R, n, seed, ro, T = 9, 64, 5, 3, 6, 11, 100, 2
sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(d, hid, 3, 1.5).items()}
X, mask = synth.make_table(n, d, seed=4)
x_obs = (X * (1.0 - mask)).astype(np.float32)
p = SamplerPlan(sd, num_steps=S, n_resample=R, dtype="f32", device="cuda:0")
nd = n_draws(S, R)
nz = torch.zeros((T, nd, n, d), device="cuda:0")
INIT = 0xFFFFFFFF
for t in range(T):
    f = lambda block, comp: noise_field(n, d, seed, ro, t, block, comp)
    nz[t, 0] = f(INIT, 0)
    n_it = R * (S - 1) + 1
    for it in range(n_it):
        base = 1 + 4 * it
        nz[t, base + 1] = f(it, 0)
        nz[t, base + 2] = f(INIT, 2) if it == 0 else f(it - 1, 2)
        if base + 3 < nd:
            nz[t, base + 3] = f(it, 1)
a, tra = p.impute_sum(x_obs, mask, T, seed=seed, row_offset=ro, trace=True)
b, trb = p.impute_sum(x_obs, mask, T, noise=nz, trace=True)
print("final max diff", float((a - b).abs().max()))
dd = (tra - trb).abs().amax(dim=(1, 2, 3)).cpu().numpy()
print("per-iteration max diff", np.round(dd, 5))
