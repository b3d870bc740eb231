// float32 MFMA GEMM (v_mfma_f32_32x32x2_f32) with fused bias / SiLU epilogue: the parity path.
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] ),  all float32.
//
// The f32-input MFMA is bit-for-bit a k-ordered fmaf chain (no reduced-precision path exists on gfx950), so
// this path differs from the reference's fp32 torch GEMMs only by summation order.  Tile 128(m) x 64(n) x 16,
// 4 waves as 2x2, each wave 64 x 32 = two 32x32 accumulators.  Operands are staged through registers into
// LDS rows padded to 17 floats (column reads with ds_read_b32 are then conflict-free), two LDS stages.
#pragma once
#include "dp_util.h"

namespace dp {

template <bool ACT>
__global__ __launch_bounds__(256) void k_gemm_f32(const float* __restrict__ A, const float* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                  int K, int nbn) {
    constexpr int BM = 128, BN = 64, BK = 16, LD = BK + 1;
    __shared__ float As[2][BM * LD];
    __shared__ float Ws[2][BN * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = (blockIdx.x / nbn) * BM, n0 = (blockIdx.x % nbn) * BN;

    // staging map: a float4 = 4 consecutive k of one row; A tile = 128 rows x 4 float4, W tile = 64 rows x 4 float4
    const int ar0 = tid >> 2, ac = (tid & 3) * 4;                 // A rows ar0 and ar0 + 64
    const float* a_src0 = A + (size_t)(m0 + ar0) * K + ac;
    const float* a_src1 = a_src0 + (size_t)64 * K;
    const float* w_src = W + (size_t)(n0 + ar0) * K + ac;          // valid for all 256 threads: 64 rows x 4

    float4 ra0, ra1, rw;
    auto gload = [&](int kt) {
        ra0 = *(const float4*)(a_src0 + kt * BK);
        ra1 = *(const float4*)(a_src1 + kt * BK);
        rw = *(const float4*)(w_src + kt * BK);
    };
    auto lstore = [&](int s) {
        float* a = &As[s][ar0 * LD + ac];
        a[0] = ra0.x; a[1] = ra0.y; a[2] = ra0.z; a[3] = ra0.w;
        float* b = &As[s][(ar0 + 64) * LD + ac];
        b[0] = ra1.x; b[1] = ra1.y; b[2] = ra1.z; b[3] = ra1.w;
        float* w = &Ws[s][ar0 * LD + ac];
        w[0] = rw.x; w[1] = rw.y; w[2] = rw.z; w[3] = rw.w;
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    // A operand: lane l holds A[i = l & 31][k = l >> 5]; B operand: B[k = l >> 5][j = l & 31]
    const int fa = (wm * 64 + (lane & 31)) * LD + (lane >> 5);
    const int fw = (wn * 32 + (lane & 31)) * LD + (lane >> 5);

    const int nk = K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* as = As[cur];
        const float* ws = Ws[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float b = ws[fw + kk * 2];
            const float a0 = as[fa + kk * 2];
            const float a1 = as[fa + 32 * LD + kk * 2];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);       // the other stage was last read before the previous barrier
        __syncthreads();
    }

    // C/D map: col = lane & 31 (n), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (m)
    const int n = n0 + wn * 32 + (lane & 31);
    const float bv = bias[n];
    const int mbase = m0 + wm * 64 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = (r & 3) + 8 * (r >> 2);
        float v0 = acc0[r] + bv, v1 = acc1[r] + bv;
        if (ACT) { v0 = silu_precise(v0); v1 = silu_precise(v1); }
        out[(size_t)(mbase + mr) * ldo + n] = v0;
        out[(size_t)(mbase + 32 + mr) * ldo + n] = v1;
    }
}

}  // namespace dp
