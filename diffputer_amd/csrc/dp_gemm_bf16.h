// bf16 MFMA GEMM with fused bias / SiLU epilogue for the denoiser's Linear layers (model.py:102-111).
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] )        A: activations [M, K], W: torch Linear weight [N, K]
//
// Both operands are K-contiguous, which is exactly what the gfx950 16x16x32 bf16 MFMA wants: a lane's
// fragment is 8 consecutive k of one row.  The product is computed TRANSPOSED (MFMA A-operand = W rows,
// B-operand = activation rows), so that a lane's 4 accumulator registers are 4 consecutive n of one output
// row m: the epilogue packs them into one 8-byte bf16 store instead of four 2-byte ones.
//
// Tile BM x BN x 64, WM x WN waves, each wave (BM/WM) x (BN/WN).  Operand tiles are staged global -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), two LDS stages, one barrier per K-tile.
// LDS image of a tile: [rows][64 bf16] = 128-B rows, 16-B chunk c of row r stored at chunk position c ^ (r & 7).
// The LDS-DMA destination is lane-linear (wave base + lane*16), so the XOR is applied to the per-lane global
// SOURCE address and again on the ds_read_b128 (cdna_hip_programming.md §5.4 rule 21); with it the 16 lanes
// of every ds_read_b128 lane group hit 16 distinct 16-B bank slots (conflict-free).
#pragma once
#include "dp_util.h"

namespace dp {

#define DP_GLOBAL_AS __attribute__((address_space(1)))
#define DP_LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const DP_GLOBAL_AS void*)gsrc, (DP_LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous run of tiles; neighbouring tiles then share operand panels in that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int BM, int BN, int WM, int WN, bool ACT, bool OUT_F32>
__global__ __launch_bounds__(WM* WN * 64) void k_gemm_bf16(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                            const float* __restrict__ bias, void* __restrict__ out,
                                                            int ldo, int K, int nbn, int nwg) {
    constexpr int NW = WM * WN, BK = 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;          // 16x16 tiles per wave along m / n
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int GA = BM / 8, GB = BN / 8;                      // 8-row (1 KiB) DMA pieces per tile
    static_assert(GA % NW == 0 && GB % NW == 0, "pieces must divide over the waves");
    constexpr int PA = GA / NW, PB = GB / NW;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int t = xcd_remap(blockIdx.x, nwg);
    const int m0 = (t / nbn) * BM, n0 = (t % nbn) * BN;

    // ---- staging: per-lane source pointers (row = piece*8 + lane/8, chunk = (lane%8) ^ (row%8)) ----
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const bf16_t* a_src = A + (size_t)(m0 + wave * 8 + srow) * K + schunk * 8;
    const bf16_t* w_src = W + (size_t)(n0 + wave * 8 + srow) * K + schunk * 8;
    const size_t piece_stride = (size_t)NW * 8 * K;               // elements between a wave's consecutive pieces

    auto stage = [&](int s, int kt) {
        char* base = smem + s * STAGE;
#pragma unroll
        for (int p = 0; p < PA; ++p)
            glds16(a_src + p * piece_stride + kt * BK, base + (wave + p * NW) * 1024);
#pragma unroll
        for (int p = 0; p < PB; ++p)
            glds16(w_src + p * piece_stride + kt * BK, base + A_BYTES + (wave + p * NW) * 1024);
    };

    // ---- fragment read offsets: row (lane&15) of a 16-row tile, chunk ks*4 + lane/16, swizzled ----
    const int foff0 = (lane & 15) * 128 + ((((lane >> 4)) ^ (lane & 7)) << 4);   // ks = 0
    const int foff1 = foff0 ^ 64;                                                // ks = 1 (chunk ^ 4)
    const int a_tile0 = (wm * TM * 16) * 128;                      // byte offset of this wave's first m-tile
    const int w_tile0 = A_BYTES + (wn * TN * 16) * 128;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    __syncthreads();                                               // hipcc drains vmcnt(0) before the barrier

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sb = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? foff1 : foff0;
            bf16x8 af[TM], wf[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *(const bf16x8*)(sb + w_tile0 + j * 2048 + fo);
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(sb + a_tile0 + i * 2048 + fo);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m = lane&15][n = 4*(lane>>4) + 0..3] of each 16x16 tile ----
    const int em = m0 + wm * TM * 16 + (lane & 15);
    const int en = n0 + wn * TN * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float4 bv = *(const float4*)(bias + en + j * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
            if (ACT) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
            const size_t o = (size_t)(em + i * 16) * ldo + en + j * 16;
            if (OUT_F32) {
                *(float4*)((float*)out + o) = make_float4(v0, v1, v2, v3);
            } else {
                *(uint2*)((bf16_t*)out + o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
        }
    }
}

template <int BM, int BN>
constexpr int gemm_bf16_lds_bytes() { return 2 * (BM + BN) * 64 * 2; }

}  // namespace dp
