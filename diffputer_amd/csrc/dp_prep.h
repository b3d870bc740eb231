// Data-prep and scoring reductions around the sampler, on the device so that a 10M-row table
// never round-trips through the host between the E-step and the M-step:
//   masked column mean / std over OBSERVED cells          dataset.py:161-169 `mean_std`
//   sum |d|, sum d^2, count over MISSING numeric cells     dataset.py:199-201 (`get_eval`'s MAE / RMSE)
// HBM-bound streaming reductions: 5 B/cell (stats, per pass) and 9 B/cell (errors).  Accumulation is float64 (the reference
// sums float32 in numpy's pairwise order; float64 atomics make the order irrelevant at float32 output precision).
#pragma once
#include "dp_util.h"

namespace dp {

// pass 0: cnt[c] += observed, acc[c] += x        pass 1: acc[c] += (x - mean[c])^2 over observed cells
// One block owns a slab of rows and keeps per-column partials in LDS (float64 LDS atomics), then flushes them.
__global__ __launch_bounds__(256) void k_col_stats(const float* __restrict__ X, const uint8_t* __restrict__ mask, int64_t n_rows, int D,
                                                   int pass, const float* __restrict__ mean, double* __restrict__ acc,
                                                   double* __restrict__ cnt, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_acc = (double*)smem;
    double* s_cnt = s_acc + D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) { s_acc[c] = 0.0; s_cnt[c] = 0.0; }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n_rows ? r0 + rows_per_block : n_rows;
    const int64_t e1 = r1 * D;
    for (int64_t e = r0 * D + threadIdx.x; e < e1; e += blockDim.x) {
        if (mask[e]) continue;                                    // 1 = missing: not part of the statistics
        const int c = (int)(e % D);
        const double x = (double)X[e];
        if (pass == 0) {
            atomicAdd(&s_acc[c], x);
            atomicAdd(&s_cnt[c], 1.0);
        } else {
            const double d = (double)(X[e] - mean[c]);            // the reference subtracts in float32 (dataset.py:167)
            atomicAdd(&s_acc[c], d * d);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        if (s_acc[c] != 0.0) atomicAdd(&acc[c], s_acc[c]);
        if (pass == 0 && s_cnt[c] != 0.0) atomicAdd(&cnt[c], s_cnt[c]);
    }
}

// pass 0 -> mean[c] = acc / max(cnt, 1) (dataset.py:164-165), acc cleared for pass 1;   pass 1 -> std[c] = sqrt(acc / max(cnt, 1))
__global__ void k_col_finish(int D, int pass, double* acc, const double* cnt, float* out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    const double n = cnt[c] == 0.0 ? 1.0 : cnt[c];
    if (pass == 0) { out[c] = (float)(acc[c] / n); acc[c] = 0.0; }
    else out[c] = (float)sqrt(acc[c] / n);
}

// out[0] += sum |pred - truth|, out[1] += sum (pred - truth)^2, out[2] += count over mask == 1 in the first num_num columns
__global__ __launch_bounds__(256) void k_masked_errors(const float* __restrict__ pred, const float* __restrict__ truth,
                                                       const uint8_t* __restrict__ mask, int64_t n_cells, int D, int ld_mask, int num_num,
                                                       double* __restrict__ out) {
    double a = 0.0, s = 0.0, n = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_cells; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / D;
        const int c = (int)(e - r * D);
        if (c < num_num && mask[r * ld_mask + c]) {
            const double d = (double)(pred[e] - truth[e]);        // float32 difference, as numpy computes it (dataset.py:199)
            a += fabs(d); s += d * d; n += 1.0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); s += __shfl_down(s, off, 64); n += __shfl_down(n, off, 64); }
    if ((threadIdx.x & 63) == 0 && n != 0.0) { atomicAdd(&out[0], a); atomicAdd(&out[1], s); atomicAdd(&out[2], n); }
}

}  // namespace dp
