import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import golden_noise, load_golden
from diffputer_amd import synth, _lib
from diffputer_amd.sampler import SamplerPlan
lib = _lib.load()
for name in ["impute_D9_H64_S50_N8", "impute_D9_H1024_S50_N16", "impute_D64_H1024_S10_N32", "impute_D9_H64_S5_N8", "impute_D34_H256_S10_N16"]:
    g = load_golden(name)
    S, R = int(g["S"]), int(g["R"])
    noise = g["noise"] if "noise" in g else golden_noise(g)
    sd = synth.make_weights(int(g["d_in"]), int(g["hid"]), int(g["wseed"]), float(g["gain"]))
    for fold in (0, 1):
        lib.dp_tune(b"fold_proj", fold)
        p = SamplerPlan({k: torch.from_numpy(v) for k, v in sd.items()}, num_steps=S, n_resample=R, dtype="bf16", device="cuda:0")
        out, tr = p.impute_sum(g["x_obs"], g["mask"], 1, noise=noise, trace=True)
        t = p.schedule(); tr = tr[:, 0].cpu().numpy(); out = out.cpu().numpy(); p.close()
        worst = 0.0; wabs = 0.0; wrms = 0.0
        for k, ref in zip(g["trace_idx"], g["trace"]):
            lvl = min(int(k) // R, S - 1)
            lim = np.maximum(5e-2 * max(1.0, float(t[lvl])), 2e-2 * np.abs(ref))
            worst = max(worst, float((np.abs(tr[k] - ref) / lim).max()))
            wrms = max(wrms, float(np.sqrt(np.mean((tr[k] - ref) ** 2)) / max(1.0, float(t[lvl]))))
            wabs = max(wabs, float(np.abs(tr[k] - ref).max() / max(1.0, float(t[lvl]))))
        fin = float((np.abs(out - g["out"]) / np.maximum(5e-2, 2e-2 * np.abs(g["out"]))).max())
        print(f"{name:28s} fold={fold} gain={float(g['gain'])} worst traj ratio {worst:.3f} (abs/scale {wabs:.4f}) final ratio {fin:.3f} final maxabs {np.abs(out-g['out']).max():.4f} worst rms/scale {wrms:.4f} |ref|max {np.abs(g['out']).max():.1f}")
