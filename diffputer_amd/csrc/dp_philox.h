// Counter-based Gaussian noise for the sampler: Philox4x32-10 + Box-Muller.
//
// The reference draws from torch's global, unseeded generator (diffusion_utils.py:25,44,45,52,96).  A
// row-sharded multi-GPU run needs draws that do not depend on the sharding, so every draw is addressed.  One Philox
// block yields four Gaussians; they serve the four cells of a COLUMN GROUP (columns 4g .. 4g+3 of one row and trial),
// so a cell costs a quarter of a block per draw:
//   key     = seed (64 bit)
//   counter = (global_row lo, global_row hi, (global_trial << 16) | column group g, block id)
//   block id of (inner iteration it, kind) = 4 it + kind;  kind 0 = n_prev of iteration it      (the draw of :45)
//                                                                1 = re-noise n of iteration it  (the draw of :52)
//                                                                2 = churn eps of iteration it+1 (the draw of :96)
//   block id of (DP_PHILOX_INIT_BLOCK, kind) = 0xFFFFFFFC + kind;  kind 0 = the initial x_T draw (:25), 2 = churn of iteration 0
//   cell (.., column c) takes component c & 3 of its group's block.
// The dead draw n_curr (:44) is never generated.  A thread that owns a whole group (V = 4 step kernels) evaluates the
// block once; a thread that owns one cell evaluates its group's block and keeps its own component -- the same bits.
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

__device__ __forceinline__ uint32_t noise_block(uint32_t it, int kind) {
    return it == DP_PHILOX_INIT_BLOCK ? 0xFFFFFFFCu + (uint32_t)kind : it * 4u + (uint32_t)kind;
}

// the four Gaussians of kind `kind` at iteration `it` for the column group that holds column `col`
__device__ __forceinline__ void noise_group4(uint64_t seed, uint64_t grow, int gtrial, int col, uint32_t it, int kind, float z[4]) {
    philox_normal4(seed, grow, trial_col(gtrial, col >> 2), noise_block(it, kind), z[0], z[1], z[2], z[3]);
}

// the Gaussian of one cell: component col & 3 of its group's block (only the Box-Muller pair that holds it is evaluated)
__device__ __forceinline__ float noise_cell(uint64_t seed, uint64_t grow, int gtrial, int col, uint32_t it, int kind) {
    const uint4 r = philox4x32_10(make_uint4((uint32_t)grow, (uint32_t)(grow >> 32), trial_col(gtrial, col >> 2), noise_block(it, kind)),
                                  make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const bool hi = (col & 2) != 0;
    const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(u01(hi ? r.z : r.x)));
    const float ang = u01(hi ? r.w : r.y);
    return rad * ((col & 1) ? __builtin_amdgcn_sinf(ang) : __builtin_amdgcn_cosf(ang));
}

}  // namespace dp
