// Counter-based Gaussian noise for the sampler: Philox4x32-10 + Box-Muller.
//
// The reference draws from torch's global, unseeded generator (diffusion_utils.py:25,44,45,52,96).  A
// row-sharded multi-GPU run needs draws that do not depend on the sharding, so every draw is addressed:
//   key     = seed (64 bit)
//   counter = (global_row lo, global_row hi, (global_trial << 16) | column, draw block)
// One Philox block yields the three Gaussians a cell needs for one inner iteration (see dp_step.h):
//   block b = inner-iteration index (0 .. R*(S-1)):   [0] n_prev of iteration b   (the draw of :45)
//                                                      [1] re-noise n of iteration b (the draw of :52)
//                                                      [2] churn eps of iteration b+1 (the draw of :96)
//   block 0xFFFFFFFF:                                  [0] the initial x_T draw (:25), [2] churn eps of iteration 0
// The dead draw n_curr (:44) is never generated.
#pragma once
#include "dp_util.h"

namespace dp {

#define DP_PHILOX_INIT_BLOCK 0xFFFFFFFFu

__host__ __device__ inline uint4 philox4x32_10(uint4 c, uint2 k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
        c = make_uint4((uint32_t)(p1 >> 32) ^ c.y ^ k.x, (uint32_t)p1, (uint32_t)(p0 >> 32) ^ c.w ^ k.y, (uint32_t)p0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}

// 24-bit uniform in (0, 1): never 0, so log() is finite; |z| <= sqrt(-2 ln 2^-25) = 5.89
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// Gaussians of one block.  v_sin/v_cos take their argument in revolutions, so cos(2*pi*u) = cosf_rev(u).
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t grow, uint32_t trial_col, uint32_t block,
                                               float& z0, float& z1, float& z2, float& z3) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)grow, (uint32_t)(grow >> 32), trial_col, block),
                                  make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    // v_sqrt_f32 / v_log_f32 directly (1 ulp): the argument is in [6e-8, 34.7], no denormal or range fix-up is needed
    const float ra = __builtin_amdgcn_sqrtf(-2.0f * __logf(u01(r.x)));
    const float rb = __builtin_amdgcn_sqrtf(-2.0f * __logf(u01(r.z)));
    const float ua = u01(r.y), ub = u01(r.w);
    z0 = ra * __builtin_amdgcn_cosf(ua);
    z1 = ra * __builtin_amdgcn_sinf(ua);
    z2 = rb * __builtin_amdgcn_cosf(ub);
    z3 = rb * __builtin_amdgcn_sinf(ub);
}

__device__ __forceinline__ uint32_t trial_col(int gtrial, int col) { return ((uint32_t)gtrial << 16) | (uint32_t)col; }

}  // namespace dp
