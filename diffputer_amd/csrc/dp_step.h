// Fused elementwise kernels of the masked EDM sampler: one launch per denoiser evaluation.
//
// Every kernel works on chains: chain m = trial_local * n_rows + row carries one row of the table through one
// trial.  State per chain (float32 [M, D]): x_hat (the churned state fed to the first evaluation of an inner
// iteration) and d_cur (the Euler slope).  x_t itself is never stored: the kernel that finishes inner
// iteration `it` immediately applies the churn of iteration it+1 and writes x_hat and the network input
// u = c_in * x_hat for the next evaluation.  Reference code being restated (diffusion_utils.py):
//   init : x_t = randn * t0 (:25,:35); churn x_hat = x_t + sqrt(t_hat^2 - t_cur^2) * eps (:96)
//   mid  : D = c_skip x_hat + c_out F (model.py:164); d_cur = (x_hat - D)/t_hat (:100); x_e = x_hat + (t_next - t_hat) d_cur (:101)
//   end  : d' = (x_e - D(x_e))/t_next (:106); Heun (:107); x_known = x + randn * t_next (:45,:47);
//          blend x_known (1-mask) + x_unknown mask (:50); re-noise + randn * sqrt(t_cur^2 - t_next^2) iff j < R-1 (:52-57);
//          then the churn (:96) of the next inner iteration
//   final: Euler-only last level (t_next = 0, :58-66) + the caller's blend and trial sum (main.py:182-188)
// Arithmetic is float32 in the reference's operation order; contraction into fma is disabled so the float32
// path rounds like the reference's separate torch ops.
#pragma once
#include "dp_philox.h"

namespace dp {

struct StepCommon {
    float* x_hat;          // [M, D]
    float* d_cur;          // [M, D]
    const float* F;        // [Mp, ldF] raw network output of the evaluation that just ran
    void* u;               // [Mp, ldu] network input for the next evaluation (bf16 or f32)
    const float* x_obs;    // [n_rows, D]
    const uint8_t* mask;   // [n_rows, D], 1 = missing
    const float* noise;    // NULL or [T, n_draws, n_rows, D]
    float* trace;          // NULL or [n_iter, T, n_rows, D]
    int64_t n_cells;       // M * D
    int64_t n_rows;
    int64_t draw_stride;   // n_rows * D
    int64_t trial_stride;  // n_draws * n_rows * D
    uint64_t seed, row_offset;
    int D, ldF, ldu, u_bf16, trial_offset, num_trials;
};

// What the mlp.6 GEMM needs to finish the first half of a Heun iteration in its epilogue (EPI_MID): D = 256 columns
struct MidStep {
    const float* x_hat;    // [m_valid, 256]
    float* d_cur;          // [m_valid, 256]
    bf16_t* u;             // [Mp, ldu]
    int64_t m_valid;       // chains; GEMM rows beyond it are padding
    int ldu;
    float c_skip, c_out, t_hat, h, c_in_next;
    unsigned long long* clk;   // NULL, or 4 words that workgroup 0 of the launch fills: shader-clock ticks (s_memtime) at its
                               // start and end, 100 MHz wall ticks (s_memrealtime) at its start and end -> the clock the
                               // chip held under this kernel (bench.py's in-situ profiling samples only)
};

struct Cell { int64_t m, row; int col, tl; };
__device__ __forceinline__ Cell cell_of(const StepCommon& c, int64_t e) {
    Cell k;
    k.m = e / c.D;
    k.col = (int)(e - k.m * c.D);
    k.tl = (int)(k.m / c.n_rows);
    k.row = k.m - (int64_t)k.tl * c.n_rows;
    return k;
}
// reference draw k of trial tl at (row, col) in injected-noise mode
__device__ __forceinline__ float draw(const StepCommon& c, const Cell& k, int64_t idx) {
    return c.noise[(int64_t)k.tl * c.trial_stride + idx * c.draw_stride + k.row * c.D + k.col];
}

#pragma clang fp contract(off)

// The four sampler kernels are templated on V = cells per thread (consecutive columns of one row): V = 4 when D % 4 == 0
// and the caller's tables are 16-byte aligned (16-byte loads and stores, a quarter of the memory instructions), else 1.
// The per-cell arithmetic is the same sequence of float32 operations for either V.
template <int V> struct VecF { float v[V]; };
template <int V> __device__ __forceinline__ VecF<V> ldf(const float* p) {
    VecF<V> r;
    if (V == 4) { const float4 t = *(const float4*)p; r.v[0] = t.x; r.v[1 % V] = t.y; r.v[2 % V] = t.z; r.v[3 % V] = t.w; }
    else r.v[0] = *p;
    return r;
}
template <int V> __device__ __forceinline__ void stf(float* p, const VecF<V>& r) {
    if (V == 4) *(float4*)p = make_float4(r.v[0], r.v[1 % V], r.v[2 % V], r.v[3 % V]);
    else *p = r.v[0];
}
template <int V> __device__ __forceinline__ void store_u(const StepCommon& c, int64_t m, int col, const VecF<V>& r) {
    if (c.u_bf16) {
        bf16_t* p = (bf16_t*)c.u + m * c.ldu + col;
        if (V == 4) *(uint2*)p = make_uint2(pack_bf16x2(r.v[0], r.v[1 % V]), pack_bf16x2(r.v[2 % V], r.v[3 % V]));
        else *p = f2bf(r.v[0]);
    } else {
        stf<V>((float*)c.u + m * c.ldu + col, r);
    }
}
template <int V> __device__ __forceinline__ VecF<V> ldmask(const uint8_t* p) {
    VecF<V> r;
    if (V == 4) { const uint32_t t = *(const uint32_t*)p; r.v[0] = (float)(t & 255u); r.v[1 % V] = (float)((t >> 8) & 255u); r.v[2 % V] = (float)((t >> 16) & 255u); r.v[3 % V] = (float)(t >> 24); }
    else r.v[0] = (float)*p;
    return r;
}

// x_t = z0 * t0, then the churn of inner iteration 0.
template <int V>
__global__ __launch_bounds__(256) void k_step_init(StepCommon c, float t0, float churn0, float c_in0) {
    const int64_t nv = c.n_cells / V;
    for (int64_t ev = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ev < nv; ev += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = ev * V;
        Cell k = cell_of(c, e);
        VecF<V> xh, uu;
        float g0[4], gc[4];                                  // V = 4: the group's x_T and churn draws, one block each
        if (V == 4 && !c.noise) {
            noise_group4(c.seed, c.row_offset + k.row, c.trial_offset + k.tl, k.col, DP_PHILOX_INIT_BLOCK, 0, g0);
            noise_group4(c.seed, c.row_offset + k.row, c.trial_offset + k.tl, k.col, DP_PHILOX_INIT_BLOCK, 2, gc);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float z0, zc;
            if (c.noise) {
                z0 = draw(c, k, 0);
                zc = draw(c, k, 3);                      // iteration 0: draws 1 n_curr, 2 n_prev, 3 churn, 4 n
            } else if (V == 4) {
                z0 = g0[v];
                zc = gc[v];
            } else {
                z0 = noise_cell(c.seed, c.row_offset + k.row, c.trial_offset + k.tl, k.col, DP_PHILOX_INIT_BLOCK, 0);
                zc = noise_cell(c.seed, c.row_offset + k.row, c.trial_offset + k.tl, k.col, DP_PHILOX_INIT_BLOCK, 2);
            }
            const float x_t = z0 * t0;
            xh.v[v] = x_t + churn0 * zc;
            uu.v[v] = c_in0 * xh.v[v];
            ++k.col;
        }
        stf<V>(c.x_hat + e, xh);
        store_u<V>(c, k.m, k.col - V, uu);
    }
}

// Per-cell arithmetic after the first evaluation of an inner iteration at a Heun level (model.py:164,
// diffusion_utils.py:100-101): shared by k_step_mid and by the GEMM epilogue that fuses it (dp_gemm_bf16_pp.h, EPI_MID).
// The slopes divide by the noise level, the same number for every cell of a launch: callers form 1/t once per thread
// (an IEEE division) and the cells multiply -- within 1 ulp of the reference's per-cell division (:100, :106), at a
// tenth of its instruction count.
__device__ __forceinline__ void mid_step_cell(float xh, float Fx, float c_skip, float c_out, float inv_t_hat, float h, float c_in_next,
                                              float& d, float& u) {
    const float den = c_skip * xh + c_out * Fx;
    d = (xh - den) * inv_t_hat;
    const float xe = xh + h * d;
    u = c_in_next * xe;
}

// After the first evaluation of an inner iteration at a Heun level.
template <int V>
__global__ __launch_bounds__(256) void k_step_mid(StepCommon c, float c_skip, float c_out, float t_hat, float h, float c_in_next) {
    const int64_t nv = c.n_cells / V;
    const float inv_t_hat = 1.0f / t_hat;
    for (int64_t ev = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ev < nv; ev += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = ev * V;
        const int64_t m = e / c.D;
        const int col = (int)(e - m * c.D);
        const VecF<V> xh = ldf<V>(c.x_hat + e), Fx = ldf<V>(c.F + m * c.ldF + col);
        VecF<V> dd, uu;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            mid_step_cell(xh.v[v], Fx.v[v], c_skip, c_out, inv_t_hat, h, c_in_next, dd.v[v], uu.v[v]);
        }
        stf<V>(c.d_cur + e, dd);
        store_u<V>(c, m, col, uu);
    }
}

// After the second evaluation: Heun, observed-value blend, re-noise, next churn.
//   it = global inner-iteration index; renoise > 0 iff j < R-1; churn_next/c_in_next belong to iteration it+1.
//   has_next = 0 on the last iteration of a run without the final level (DP_NO_FINAL_LEVEL): no iteration it+1 exists, so
//   its churn draw is not taken (a replay buffer holds no such draw) and x_hat / u are left as they are.
template <int V>
__global__ __launch_bounds__(256) void k_step_end(StepCommon c, int it, float c_skip, float c_out, float t_next, float h,
                                                  float renoise, int add_renoise, float churn_next, float c_in_next,
                                                  float* __restrict__ xt_out, int has_next) {
    const int64_t nv = c.n_cells / V;
    const float inv_t_next = 1.0f / t_next;
    for (int64_t ev = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ev < nv; ev += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = ev * V;
        Cell k = cell_of(c, e);
        const int64_t oc = k.row * c.D + k.col;
        const VecF<V> xhv = ldf<V>(c.x_hat + e), dv = ldf<V>(c.d_cur + e), Fv = ldf<V>(c.F + k.m * c.ldF + k.col);
        const VecF<V> xov = ldf<V>(c.x_obs + oc), mkv = ldmask<V>(c.mask + oc);
        VecF<V> xt, xh2, uu;
        float gp[4], gr[4], gc[4];                           // V = 4: one block per kind for the thread's column group
        if (V == 4 && !c.noise) {
            const uint64_t grow = c.row_offset + k.row;
            const int gt = c.trial_offset + k.tl;
            noise_group4(c.seed, grow, gt, k.col, (uint32_t)it, 0, gp);
            if (add_renoise) noise_group4(c.seed, grow, gt, k.col, (uint32_t)it, 1, gr);
            if (has_next) noise_group4(c.seed, grow, gt, k.col, (uint32_t)it, 2, gc);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float z_prev, z_re = 0.f, z_churn = 0.f;
            if (c.noise) {
                const int64_t base = 1 + 4 * (int64_t)it;
                z_prev = draw(c, k, base + 1);
                z_re = add_renoise ? draw(c, k, base + 3) : 0.f;
                if (has_next) z_churn = draw(c, k, base + 4 + 2);      // churn draw of iteration it+1
            } else if (V == 4) {
                z_prev = gp[v];
                if (add_renoise) z_re = gr[v];
                if (has_next) z_churn = gc[v];
            } else {
                const uint64_t grow = c.row_offset + k.row;
                const int gt = c.trial_offset + k.tl;
                z_prev = noise_cell(c.seed, grow, gt, k.col, (uint32_t)it, 0);
                if (add_renoise) z_re = noise_cell(c.seed, grow, gt, k.col, (uint32_t)it, 1);
                if (has_next) z_churn = noise_cell(c.seed, grow, gt, k.col, (uint32_t)it, 2);
            }
            const float xh = xhv.v[v], d = dv.v[v], mk = mkv.v[v];
            const float xe = xh + h * d;
            const float den = c_skip * xe + c_out * Fv.v[v];
            const float dp = (xe - den) * inv_t_next;
            const float xn = xh + h * (0.5f * d + 0.5f * dp);
            const float xk = xov.v[v] + z_prev * t_next;
            float x_t = xk * (1.0f - mk) + xn * mk;
            if (add_renoise) x_t = x_t + z_re * renoise;
            xt.v[v] = x_t;
            xh2.v[v] = x_t + churn_next * z_churn;
            uu.v[v] = c_in_next * xh2.v[v];
            ++k.col;
        }
        if (c.trace) stf<V>(c.trace + (int64_t)it * c.n_cells + e, xt);
        if (xt_out) stf<V>(xt_out + e, xt);                       // DP_NO_FINAL_LEVEL: x_t itself is the result
        if (has_next) {
            stf<V>(c.x_hat + e, xh2);
            store_u<V>(c, k.m, k.col - V, uu);
        }
    }
}

// Step kernels for tables whose rows cannot be walked 16 bytes at a time (D % 4 != 0: california 9, adult 34, news 51; or
// unaligned caller tables): a thread owns one COLUMN GROUP of one chain (columns 4g .. 4g+3, clipped to D).  In Philox
// mode the group's blocks are then evaluated once instead of once per cell (the per-cell kernels spend 4x the generator
// work), and a whole group moves with one 16-byte access that is only 4-byte aligned (gfx950 takes unaligned dwordx4
// global accesses; F and u are padded to 64 columns, so theirs are aligned).  Same draws and the same float32 operations
// per cell as k_step_mid / k_step_end: bit-identical results.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct GroupOf { int64_t m, row; int col0, nv, tl; };
__device__ __forceinline__ GroupOf group_of(const StepCommon& c, int64_t idx, int G) {
    GroupOf k;
    k.m = idx / G;
    k.col0 = (int)(idx - k.m * G) * 4;
    k.nv = min(4, c.D - k.col0);
    k.tl = (int)(k.m / c.n_rows);
    k.row = k.m - (int64_t)k.tl * c.n_rows;
    return k;
}
// 4 (or, for a row's last group, nv < 4) consecutive floats at p
__device__ __forceinline__ void ld_group(const float* p, int nv, float v[4]) {
    if (nv == 4) { const f32x4u t = *(const f32x4u*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    else { v[0] = p[0]; v[1] = nv > 1 ? p[1] : 0.f; v[2] = nv > 2 ? p[2] : 0.f; v[3] = 0.f; }
}
__device__ __forceinline__ void st_group(float* p, int nv, const float v[4]) {
    if (nv == 4) { f32x4u t = {v[0], v[1], v[2], v[3]}; *(f32x4u*)p = t; }
    else { p[0] = v[0]; if (nv > 1) p[1] = v[1]; if (nv > 2) p[2] = v[2]; }
}
__device__ __forceinline__ void st_group_u(const StepCommon& c, int64_t m, int col0, int nv, const float v[4]) {
    if (c.u_bf16) {
        bf16_t* p = (bf16_t*)c.u + m * c.ldu + col0;             // 8-byte aligned: ldu % 64 == 0, col0 % 4 == 0
        if (nv == 4) *(uint2*)p = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        else { p[0] = f2bf(v[0]); if (nv > 1) p[1] = f2bf(v[1]); if (nv > 2) p[2] = f2bf(v[2]); }
    } else {
        float* p = (float*)c.u + m * c.ldu + col0;
        if (nv == 4) *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        else { p[0] = v[0]; if (nv > 1) p[1] = v[1]; if (nv > 2) p[2] = v[2]; }
    }
}

__global__ __launch_bounds__(256) void k_step_mid_grp(StepCommon c, float c_skip, float c_out, float t_hat, float h, float c_in_next) {
    const int G = (c.D + 3) >> 2;
    const int64_t n_groups = (c.n_cells / c.D) * G;
    const float inv_t_hat = 1.0f / t_hat;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_groups; idx += (int64_t)gridDim.x * blockDim.x) {
        const GroupOf k = group_of(c, idx, G);
        const int64_t e0 = k.m * c.D + k.col0;
        float xh[4], Fx[4], dd[4], uu[4];
        ld_group(c.x_hat + e0, k.nv, xh);
        const float4 Fv = *(const float4*)(c.F + k.m * c.ldF + k.col0);       // padded to 64 columns: always 4 readable floats
        Fx[0] = Fv.x; Fx[1] = Fv.y; Fx[2] = Fv.z; Fx[3] = Fv.w;
#pragma unroll
        for (int v = 0; v < 4; ++v) mid_step_cell(xh[v], Fx[v], c_skip, c_out, inv_t_hat, h, c_in_next, dd[v], uu[v]);
        st_group(c.d_cur + e0, k.nv, dd);
        st_group_u(c, k.m, k.col0, k.nv, uu);
    }
}

__global__ __launch_bounds__(256) void k_step_end_grp(StepCommon c, int it, float c_skip, float c_out, float t_next, float h,
                                                      float renoise, int add_renoise, float churn_next, float c_in_next,
                                                      float* __restrict__ xt_out, int has_next) {
    const int G = (c.D + 3) >> 2;
    const int64_t n_groups = (c.n_cells / c.D) * G;
    const float inv_t_next = 1.0f / t_next;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_groups; idx += (int64_t)gridDim.x * blockDim.x) {
        const GroupOf k = group_of(c, idx, G);
        const uint64_t grow = c.row_offset + k.row;
        const int gt = c.trial_offset + k.tl;
        float gp[4], gr[4] = {0.f, 0.f, 0.f, 0.f}, gc[4] = {0.f, 0.f, 0.f, 0.f};
        noise_group4(c.seed, grow, gt, k.col0, (uint32_t)it, 0, gp);
        if (add_renoise) noise_group4(c.seed, grow, gt, k.col0, (uint32_t)it, 1, gr);
        if (has_next) noise_group4(c.seed, grow, gt, k.col0, (uint32_t)it, 2, gc);
        const int64_t e0 = k.m * c.D + k.col0, oc0 = k.row * c.D + k.col0;
        float xh[4], dv[4], Fx[4], xo[4], mk[4], xt[4], xh2[4], uu[4];
        ld_group(c.x_hat + e0, k.nv, xh);
        ld_group(c.d_cur + e0, k.nv, dv);
        ld_group(c.x_obs + oc0, k.nv, xo);
        const float4 Fv = *(const float4*)(c.F + k.m * c.ldF + k.col0);
        Fx[0] = Fv.x; Fx[1] = Fv.y; Fx[2] = Fv.z; Fx[3] = Fv.w;
#pragma unroll
        for (int v = 0; v < 4; ++v) mk[v] = v < k.nv ? (float)c.mask[oc0 + v] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float d = dv[v];
            const float xe = xh[v] + h * d;
            const float den = c_skip * xe + c_out * Fx[v];
            const float dp = (xe - den) * inv_t_next;
            const float xn = xh[v] + h * (0.5f * d + 0.5f * dp);
            const float xk = xo[v] + gp[v] * t_next;
            float x_t = xk * (1.0f - mk[v]) + xn * mk[v];
            if (add_renoise) x_t = x_t + gr[v] * renoise;
            xt[v] = x_t;
            xh2[v] = x_t + churn_next * gc[v];
            uu[v] = c_in_next * xh2[v];
        }
        if (c.trace) st_group(c.trace + (int64_t)it * c.n_cells + e0, k.nv, xt);
        if (xt_out) st_group(xt_out + e0, k.nv, xt);
        if (has_next) {
            st_group(c.x_hat + e0, k.nv, xh2);
            st_group_u(c, k.m, k.col0, k.nv, uu);
        }
    }
}

// Last level (t_next = 0): Euler step = the denoised sample; blend; caller's blend; sum over this call's trials.
// One thread per V cells of (row, col); trials are summed in order, so the result is deterministic.
template <int V>
__global__ __launch_bounds__(256) void k_step_final(StepCommon c, int it, float c_skip, float c_out, float t_hat, float h,
                                                    float* __restrict__ out, int accumulate) {
    const int64_t per_trial = c.n_rows * c.D;
    const float inv_t_hat = 1.0f / t_hat;
    for (int64_t ov = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ov < per_trial / V; ov += (int64_t)gridDim.x * blockDim.x) {
        const int64_t oc = ov * V;
        const int64_t row = oc / c.D;
        const int col = (int)(oc - row * c.D);
        const VecF<V> mkv = ldmask<V>(c.mask + oc), xov = ldf<V>(c.x_obs + oc);
        VecF<V> sum;
        if (accumulate) sum = ldf<V>(out + oc);
        else for (int v = 0; v < V; ++v) sum.v[v] = 0.f;
        for (int tl = 0; tl < c.num_trials; ++tl) {
            const int64_t e = (int64_t)tl * per_trial + oc;
            const int64_t m = (int64_t)tl * c.n_rows + row;
            const VecF<V> xhv = ldf<V>(c.x_hat + e), Fv = ldf<V>(c.F + m * c.ldF + col);
            VecF<V> xt;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float xh = xhv.v[v], mk = mkv.v[v], xo = xov.v[v];
                const float den = c_skip * xh + c_out * Fv.v[v];
                const float d = (xh - den) * inv_t_hat;
                const float xe = xh + h * d;
                // x_known = x + randn * 0 = x (:60-62); blend (:65)
                const float x_t = xo * (1.0f - mk) + xe * mk;
                xt.v[v] = x_t;
                sum.v[v] = sum.v[v] + (x_t * mk + xo * (1.0f - mk));       // main.py:183
            }
            if (c.trace) stf<V>(c.trace + (int64_t)it * c.n_cells + e, xt);
        }
        stf<V>(out + oc, sum);
    }
}

// DP_NO_FINAL_LEVEL: the caller's blend and trial sum over x_t [M, D] (main.py:182-188), trials in order.
template <int V>
__global__ __launch_bounds__(256) void k_blend_sum(StepCommon c, const float* __restrict__ xt, float* __restrict__ out, int accumulate) {
    const int64_t per_trial = c.n_rows * c.D;
    for (int64_t ov = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ov < per_trial / V; ov += (int64_t)gridDim.x * blockDim.x) {
        const int64_t oc = ov * V;
        const VecF<V> mkv = ldmask<V>(c.mask + oc), xov = ldf<V>(c.x_obs + oc);
        VecF<V> sum;
        if (accumulate) sum = ldf<V>(out + oc);
        else for (int v = 0; v < V; ++v) sum.v[v] = 0.f;
        for (int tl = 0; tl < c.num_trials; ++tl) {
            const VecF<V> x = ldf<V>(xt + (int64_t)tl * per_trial + oc);
#pragma unroll
            for (int v = 0; v < V; ++v) sum.v[v] = sum.v[v] + (x.v[v] * mkv.v[v] + xov.v[v] * (1.0f - mkv.v[v]));
        }
        stf<V>(out + oc, sum);
    }
}

// --- stand-alone pieces used by dp_denoise / dp_sample_step (the reference's net(x, sigma) and sample_step) ---

// u = c_in * x  (model.py:160)
__global__ __launch_bounds__(256) void k_scale_in(const float* __restrict__ x, void* u, int ldu, int u_bf16, int64_t n_cells, int D, float c_in) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_cells; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / D;
        const int col = (int)(e - m * D);
        const float v = c_in * x[e];
        if (u_bf16) ((bf16_t*)u)[m * ldu + col] = f2bf(v);
        else ((float*)u)[m * ldu + col] = v;
    }
}

// D = c_skip x + c_out F (model.py:164); optionally also copy F out densely.
__global__ __launch_bounds__(256) void k_precond_out(const float* __restrict__ x, const float* __restrict__ F, int ldF, float* D_out,
                                                     float* F_out, int64_t n_cells, int D, float c_skip, float c_out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_cells; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / D;
        const int col = (int)(e - m * D);
        const float Fx = F[m * ldF + col];
        if (F_out) F_out[e] = Fx;
        if (D_out) D_out[e] = c_skip * x[e] + c_out * Fx;
    }
}

// churn for dp_sample_step: x_hat = x_cur + churn * eps
__global__ __launch_bounds__(256) void k_churn(const float* __restrict__ x_cur, const float* __restrict__ eps, float* x_hat, void* u, int ldu,
                                               int u_bf16, int64_t n_cells, int D, float churn, float c_in, uint64_t seed,
                                               uint64_t row_offset, uint32_t block) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_cells; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / D;
        const int col = (int)(e - m * D);
        float z;
        if (eps) z = eps[e];
        else z = noise_cell(seed, row_offset + m, 0, col, block, 2);
        const float xh = x_cur[e] + churn * z;
        x_hat[e] = xh;
        const float v = c_in * xh;
        if (u_bf16) ((bf16_t*)u)[m * ldu + col] = f2bf(v);
        else ((float*)u)[m * ldu + col] = v;
    }
}

// second half of sample_step at a Heun level / result of the Euler-only last level
__global__ __launch_bounds__(256) void k_heun_out(const float* __restrict__ x_hat, const float* __restrict__ d_cur, const float* __restrict__ F,
                                                  int ldF, float* x_next, int64_t n_cells, int D, float c_skip, float c_out, float t_div,
                                                  float h, int heun) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_cells; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / D;
        const int col = (int)(e - m * D);
        const float xh = x_hat[e];
        const float Fx = F[m * ldF + col];
        if (heun) {
            const float d = d_cur[e];
            const float xe = xh + h * d;
            const float den = c_skip * xe + c_out * Fx;
            const float dp = (xe - den) / t_div;
            x_next[e] = xh + h * (0.5f * d + 0.5f * dp);
        } else {
            const float den = c_skip * xh + c_out * Fx;
            const float d = (xh - den) / t_div;
            x_next[e] = xh + h * d;
        }
    }
}

__global__ __launch_bounds__(256) void k_noise_field(float* out, int64_t n_rows, int D, uint64_t seed, uint64_t row_offset, int trial,
                                                     uint32_t block, int comp) {
    const int64_t n = n_rows * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / D;
        const int col = (int)(e - row * D);
        out[e] = noise_cell(seed, row_offset + row, trial, col, block, comp);
    }
}

#pragma clang fp contract(fast)

}  // namespace dp
