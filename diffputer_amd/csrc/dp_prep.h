// Data-prep and scoring reductions around the sampler, on the device so that a 10M-row table
// never round-trips through the host between the E-step and the M-step:
//   masked column mean / std over OBSERVED cells          dataset.py:161-169 `mean_std`
//   sum |d|, sum d^2, count over MISSING numeric cells     dataset.py:199-201 (`get_eval`'s MAE / RMSE)
// HBM-bound streaming reductions: 5 B/cell (stats, per pass) and 9 B/cell (errors).  Accumulation is float64 (the reference
// sums float32 in numpy's pairwise order; float64 partial sums make the order irrelevant at float32 output precision).
#pragma once
#include "dp_util.h"

namespace dp {

// Column mapping shared by both reductions.  A "unit" is VEC consecutive columns (one 16-byte load of values and one 4-byte
// load of mask bytes when VEC = 4; dispatch requires D % 4 == 0 and aligned pointers for that).  blockIdx.y picks a chunk of
// up to 256 units; inside it a thread keeps ONE unit for its whole life -- its column never changes, so partial sums live in
// registers -- and the block's 256 threads cover 256 / units rows per iteration: consecutive lanes read consecutive
// addresses (whole rows back to back when the chunk is the whole row).  Four iterations' loads are issued before the first
// is consumed: ~160 KB in flight per CU at full occupancy.
struct ColMap {
    int units, rpi, ri, c0;      // units in this chunk, rows per iteration, this thread's row within an iteration, first column
    bool active;
    __device__ ColMap(int D, int VEC) {
        const int U = D / VEC, u0 = blockIdx.y * 256;
        units = U - u0 < 256 ? U - u0 : 256;
        rpi = 256 / units;
        ri = threadIdx.x / units;
        c0 = (u0 + (int)threadIdx.x - ri * units) * VEC;
        active = ri < rpi;
    }
};

template <int VEC> struct PrepVec;
template <> struct PrepVec<1> { typedef float F; typedef uint8_t M; };
template <> struct PrepVec<4> { typedef float4 F; typedef uchar4 M; };
__device__ __forceinline__ float prep_get(const float& v, int) { return v; }
__device__ __forceinline__ float prep_get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ unsigned prep_get(const uint8_t& v, int) { return v; }
__device__ __forceinline__ unsigned prep_get(const uchar4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }

// pass 0: cnt[c] += observed, acc[c] += x        pass 1: acc[c] += (x - mean[c])^2 over observed cells
// A block owns a slab of rows; per-thread float64 partials, one LDS reduction over the rows of an iteration, then one
// float64 atomic per column and block.
template <int VEC>
__global__ __launch_bounds__(256) void k_col_stats(const float* __restrict__ X, const uint8_t* __restrict__ mask, int64_t n_rows, int D,
                                                   int pass, const float* __restrict__ mean, double* __restrict__ acc,
                                                   double* __restrict__ cnt, int rows_per_block, int replicas) {
    typedef typename PrepVec<VEC>::F F;
    typedef typename PrepVec<VEC>::M M;
    __shared__ double s_a[256 * VEC];
    acc += (size_t)(blockIdx.x % replicas) * 2 * D;               // replica r: acc at ws + r * 2D, cnt D further (spreads the atomics)
    cnt += (size_t)(blockIdx.x % replicas) * 2 * D;
    __shared__ unsigned s_n[256 * VEC];
    const ColMap cm(D, VEC);
    double a[VEC];
    unsigned n[VEC];
    float mu[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { a[k] = 0.0; n[k] = 0u; mu[k] = (pass == 1 && cm.active) ? mean[cm.c0 + k] : 0.f; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    auto take = [&](const F& x, const M& m) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (prep_get(m, k)) continue;                         // 1 = missing: not part of the statistics
            const float v = prep_get(x, k);
            if (pass == 0) { a[k] += (double)v; n[k] += 1u; }
            else { const double d = (double)(v - mu[k]); a[k] += d * d; }   // the reference subtracts in float32 (dataset.py:167)
        }
    };
    if (cm.active) {
        int64_t r = r0 + cm.ri;
        const int64_t step = cm.rpi;
        for (; r + 3 * step < r1; r += 4 * step) {
            F x[4];
            M m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t e = (r + j * step) * D + cm.c0;
                x[j] = *(const F*)(X + e);
                m[j] = *(const M*)(mask + e);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) take(x[j], m[j]);
        }
        for (; r < r1; r += step) {
            const int64_t e = r * D + cm.c0;
            take(*(const F*)(X + e), *(const M*)(mask + e));
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) { s_a[threadIdx.x * VEC + k] = a[k]; s_n[threadIdx.x * VEC + k] = n[k]; }
    __syncthreads();
    if (cm.active && cm.ri == 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            double sa = a[k];
            unsigned sn = n[k];
            for (int j = 1; j < cm.rpi; ++j) { sa += s_a[(threadIdx.x + j * cm.units) * VEC + k]; sn += s_n[(threadIdx.x + j * cm.units) * VEC + k]; }
            if (sa != 0.0) atomicAdd(&acc[cm.c0 + k], sa);
            if (pass == 0 && sn != 0u) atomicAdd(&cnt[cm.c0 + k], (double)sn);
        }
    }
}

// pass 0 -> mean[c] = acc / max(cnt, 1) (dataset.py:164-165), acc cleared for pass 1;   pass 1 -> std[c] = sqrt(acc / max(cnt, 1))
// acc / cnt are summed over the replicas; the total count is kept in replica 0 for pass 1.
__global__ void k_col_finish(int D, int pass, double* ws, int replicas, float* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    double a = 0.0, n = 0.0;
    for (int r = 0; r < replicas; ++r) {
        a += ws[(size_t)r * 2 * D + c];
        if (pass == 0) n += ws[(size_t)r * 2 * D + D + c];
        ws[(size_t)r * 2 * D + c] = 0.0;
    }
    if (pass == 0) ws[D + c] = n;
    else n = ws[D + c];
    if (n == 0.0) n = 1.0;
    out[c] = pass == 0 ? (float)(a / n) : (float)sqrt(a / n);
}

// out[0] += sum |pred - truth|, out[1] += sum (pred - truth)^2, out[2] += count over mask == 1 in the first num_num columns.
// Same column mapping; a unit lies inside the first num_num columns or outside (dispatch: num_num % VEC == 0).
template <int VEC>
__global__ __launch_bounds__(256) void k_masked_errors(const float* __restrict__ pred, const float* __restrict__ truth,
                                                       const uint8_t* __restrict__ mask, int64_t n_rows, int D, int ld_mask, int num_num,
                                                       double* __restrict__ out, int rows_per_block) {
    typedef typename PrepVec<VEC>::F F;
    typedef typename PrepVec<VEC>::M M;
    __shared__ double s_r[3 * 4];
    const ColMap cm(D, VEC);
    double a = 0.0, s = 0.0;
    unsigned n = 0u;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    auto take = [&](const F& p, const F& t, const M& m) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            if (!prep_get(m, k)) continue;
            const double d = (double)(prep_get(p, k) - prep_get(t, k));   // float32 difference, as numpy computes it (dataset.py:199)
            a += fabs(d); s += d * d; n += 1u;
        }
    };
    if (cm.active && cm.c0 < num_num) {
        int64_t r = r0 + cm.ri;
        const int64_t step = cm.rpi;
        for (; r + 3 * step < r1; r += 4 * step) {
            F p[4], t[4];
            M m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t e = (r + j * step) * D + cm.c0;
                p[j] = *(const F*)(pred + e);
                t[j] = *(const F*)(truth + e);
                m[j] = *(const M*)(mask + (r + j * step) * ld_mask + cm.c0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) take(p[j], t[j], m[j]);
        }
        for (; r < r1; r += step) {
            const int64_t e = r * D + cm.c0;
            take(*(const F*)(pred + e), *(const F*)(truth + e), *(const M*)(mask + r * ld_mask + cm.c0));
        }
    }
    double cn = (double)n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); s += __shfl_down(s, off, 64); cn += __shfl_down(cn, off, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_r[wave * 3] = a; s_r[wave * 3 + 1] = s; s_r[wave * 3 + 2] = cn; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double v = s_r[threadIdx.x] + s_r[3 + threadIdx.x] + s_r[6 + threadIdx.x] + s_r[9 + threadIdx.x];
        if (v != 0.0) atomicAdd(&out[threadIdx.x], v);
    }
}

}  // namespace dp
