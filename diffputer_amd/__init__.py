"""diffputer_amd: DiffPuter's E-step imputation sampler on MI355X (gfx950).

`diffusion_utils` mirrors the reference's sampler API (impute_mask / sample_step / sample),
`model` mirrors its denoiser modules (state-dict compatible), `estep` is the trial loop + row
sharding, `sampler.SamplerPlan` is the handle of the C-ABI plan in csrc/libdiffputer_hip.so.
Importing the package does not load the HIP library; the first sampler call does, and raises if
it has not been built (there is no CPU fallback).
"""
__version__ = "0.1.0"
