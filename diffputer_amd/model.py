"""Host-side mirror of the reference's denoiser call surface (model.py:17-30, 97-181).

PyTorch owns these modules, their parameters and the M-step (training); the E-step sampler reads
their weights through `SamplerPlan` and runs on the HIP kernels.  Class names, constructor
signatures, attribute names and therefore state-dict keys
(`denoise_fn_D.denoise_fn_F.{proj,mlp.0,mlp.2,mlp.4,mlp.6,time_embed.0,time_embed.2}.{weight,bias}`)
match the reference so `model.pt` checkpoints interchange (main.py:142,173).
"""
from __future__ import annotations

import torch
from torch import nn


class PositionalEmbedding(nn.Module):
    """Sinusoidal embedding of the noise label, `[cos | sin]` order (model.py:17-30).
    `MLPDiffusion.forward` swaps the halves afterwards (model.py:123)."""

    def __init__(self, num_channels: int, max_positions: int = 10000, endpoint: bool = False):
        super().__init__()
        self.num_channels = num_channels
        self.max_positions = max_positions
        self.endpoint = endpoint

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        half = self.num_channels // 2
        k = torch.arange(half, dtype=torch.float32, device=x.device)
        freqs = (1 / self.max_positions) ** (k / (half - (1 if self.endpoint else 0)))
        ang = torch.outer(x, freqs.to(x.dtype))
        return torch.cat([ang.cos(), ang.sin()], dim=1)


class MLPDiffusion(nn.Module):
    """proj(x) + time_embed(posemb(noise)) -> Linear/SiLU x3 -> Linear (model.py:97-128)."""

    def __init__(self, d_in: int, dim_t: int = 512):
        super().__init__()
        self.dim_t = dim_t
        self.proj = nn.Linear(d_in, dim_t)
        self.mlp = nn.Sequential(
            nn.Linear(dim_t, 2 * dim_t), nn.SiLU(),
            nn.Linear(2 * dim_t, 2 * dim_t), nn.SiLU(),
            nn.Linear(2 * dim_t, dim_t), nn.SiLU(),
            nn.Linear(dim_t, d_in),
        )
        self.map_noise = PositionalEmbedding(num_channels=dim_t)
        self.time_embed = nn.Sequential(nn.Linear(dim_t, dim_t), nn.SiLU(), nn.Linear(dim_t, dim_t))

    def forward(self, x, noise_labels, class_labels=None):
        pe = self.map_noise(noise_labels)
        half = pe.shape[1] // 2
        pe = torch.cat([pe[:, half:], pe[:, :half]], dim=1)         # [sin | cos]  (the flip at model.py:123)
        return self.mlp(self.proj(x) + self.time_embed(pe))


class Precond(nn.Module):
    """EDM preconditioning D(x, sigma) = c_skip x + c_out F(c_in x, ln(sigma)/4) (model.py:131-168).

    This torch forward is what the M-step differentiates through (per-row sigma).  The E-step does
    not call it: `diffusion_utils.impute_mask` hands `denoise_fn_F`'s weights to the HIP sampler.
    """

    def __init__(self, denoise_fn, hid_dim, sigma_min=0, sigma_max=float("inf"), sigma_data=0.5):
        super().__init__()
        self.hid_dim = hid_dim
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.sigma_data = sigma_data
        self.denoise_fn_F = denoise_fn

    def forward(self, x, sigma):
        x = x.to(torch.float32)
        sigma = sigma.to(torch.float32).reshape(-1, 1)
        sd = self.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
        c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
        c_noise = sigma.log() / 4
        F_x = self.denoise_fn_F(c_in * x, c_noise.flatten())
        return c_skip * x + c_out * F_x.to(torch.float32)

    def round_sigma(self, sigma):
        return torch.as_tensor(sigma)


class EDMLoss:
    """EDM training loss (diffusion_utils.py:190-215): ln(sigma) ~ N(P_mean, P_std^2),
    weight (sigma^2 + sd^2) / (sigma sd)^2.  Stays in PyTorch (the M-step is out of the HIP scope)."""

    def __init__(self, P_mean=-1.2, P_std=1.2, sigma_data=0.5, hid_dim=100, gamma=5, opts=None):
        self.P_mean, self.P_std, self.sigma_data = P_mean, P_std, sigma_data
        self.hid_dim, self.gamma, self.opts = hid_dim, gamma, opts

    def __call__(self, denoise_fn, data):
        sigma = (torch.randn(data.shape[0], device=data.device) * self.P_std + self.P_mean).exp()
        weight = (sigma ** 2 + self.sigma_data ** 2) / (sigma * self.sigma_data) ** 2
        noisy = data + torch.randn_like(data) * sigma.unsqueeze(1)
        return weight.unsqueeze(1) * (denoise_fn(noisy, sigma) - data) ** 2


class Model(nn.Module):
    """Training wrapper: `model(x)` is the scalar EDM loss; `model.denoise_fn_D` is the `net`
    the sampler takes (model.py:171-181, main.py:107,177)."""

    def __init__(self, denoise_fn, hid_dim, P_mean=-1.2, P_std=1.2, sigma_data=0.5, gamma=5, opts=None, pfgmpp=False):
        super().__init__()
        self.denoise_fn_D = Precond(denoise_fn, hid_dim)
        self.loss_fn = EDMLoss(P_mean, P_std, sigma_data, hid_dim=hid_dim, gamma=gamma, opts=opts)

    def forward(self, x):
        return self.loss_fn(self.denoise_fn_D, x).mean(-1).mean()


def num_parameters(d_in: int, hid: int) -> int:
    """10 512 393 at d_in = 9, hid = 1024."""
    h = hid
    return (d_in * h + h) + (h * 2 * h + 2 * h) + (4 * h * h + 2 * h) + (2 * h * h + h) + (h * d_in + d_in) + 2 * (h * h + h)


def flops_per_row_eval(d_in: int, hid: int) -> int:
    """GEMM FLOPs of one denoiser evaluation of one row: 4 D H + 16 H^2."""
    return 4 * d_in * hid + 16 * hid * hid


def nfe_per_chain(num_steps: int, n_resample: int) -> int:
    """Denoiser evaluations per row per trial: 2 R (S-1) + 1 = 1961 at S=50, R=20."""
    return 2 * n_resample * (num_steps - 1) + 1


__all__ = ["PositionalEmbedding", "MLPDiffusion", "Precond", "EDMLoss", "Model",
           "num_parameters", "flops_per_row_eval", "nfe_per_chain"]
