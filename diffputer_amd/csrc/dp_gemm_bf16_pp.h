// "Ping-pong" bf16 MFMA GEMM for the denoiser's wide Linear layers (model.py:105-110): 256x256x64 tile, 8 waves,
// LDS-DMA staging that stays in flight across barriers.
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] )        A: activations [M, K], W: torch Linear weight [N, K]
//
// Why a second kernel: k_gemm_bf16 (dp_gemm_bf16.h) drains its LDS-DMA at every K-tile barrier and tops out near
// 1.0-1.1 PFLOP/s; software pipelining of that loop changes nothing (measured).  This kernel follows the
// structure cdna_hip_programming.md §5 describes for the 256^2 tile:
//   * 8 waves = 2 (m) x 4 (n), each owning 128 x 64 of the tile as 2 x 2 "quadrants" of 64 x 32;
//   * a K-tile (64 deep) is cut into four 16-KiB HALF-TILES: A0/A1 = the first/second 64 rows of both wave rows,
//     B0/B1 = the first/second 32 columns of all four wave columns; two K-tile buffers = 128 KiB of LDS;
//   * a K-tile is four PHASES, one quadrant (16 MFMAs per wave) each:
//        c0: read B0 (4 ds_read_b128) + A0 (8)   -> q(A0,B0)        c2: read A1 (8) -> q(A1,B1)
//        c1: read B1 (4)                          -> q(A0,B1)        c3: (nothing)   -> q(A1,B0)   (B0 stays in registers)
//     and every phase issues the LDS-DMA of ONE half-tile (2 x global_load_lds_dwordx4 per wave) 5-6 phases ahead
//     of its first read:   c0: B1(t+1)   c1: A1(t+1)   c2: A0(t+2)   c3: B0(t+2)
//   * phase = { ds_reads ; stage ; s_waitcnt vmcnt(8) ; s_barrier ; lgkmcnt(0) ; 16 MFMA ; s_barrier }.  The four
//     waves of wave row 1 run one barrier behind those of wave row 0, so on every SIMD one wave is in its MFMA
//     segment while its partner reads LDS and issues DMA.
// Hazards (barrier instances b1, b2, ...; row 0 runs [R_P S_P W_P] b(2P+1) [M_P] b(2P+2), row 1 one barrier later):
//   RAW  a half-tile read in phase Q was staged in phase <= Q-5; the vmcnt(8) of phase Q-1 (issued after that
//        phase's stage) leaves only the four newest half-tiles in flight, and every wave has passed that wait
//        before b(2Q), which precedes every read of phase Q.
//   WAR  a slot is re-staged >= 2 phases after its last ds_read: the lagging row's reads of phase P are complete
//        (its lgkmcnt(0)) before it arrives at b(2P+3), and the leading row stages phase P+2 after b(2P+4).
// The LDS image of a half-tile is the one of k_gemm_bf16: 8-row x 128-B pieces, 16-B chunk c of row r at c ^ (r&7)
// (swizzle on the DMA's per-lane SOURCE address and on the ds_read; conflict-free ds_read_b128).
#pragma once
#include "dp_gemm_bf16.h"

namespace dp {

#define DP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define DP_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64, PP_THREADS = 512;
constexpr int PP_HALF = 16384;                                  // bytes of one half-tile
constexpr int PP_LDS_BYTES = 8 * PP_HALF;                       // 2 K-tile buffers x {A0, A1, B0, B1}

template <bool ACT, bool OUT_F32>
__global__ __launch_bounds__(PP_THREADS) void k_gemm_bf16_pp(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ out,
                                                              int ldo, int K, int nbn, int nwg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int m0 = (t / nbn) * PP_BM, n0 = (t % nbn) * PP_BN;
    const int nk = K / PP_BK;                                   // host guarantees nk >= 2

    // ---- staging: a wave moves pieces `wave` and `wave + 8` of every half-tile (a piece = 8 rows x 128 B) ----
    //   A half h, piece p: half-local rows 8p.. = wave row p/8, rows h*64 + (p%8)*8 ..  of that wave row's 128
    //   B half h, piece p: half-local rows 8p.. = wave col p/4, cols h*32 + (p%4)*8 ..  of that wave col's 64
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const bf16_t* a_src = A + (size_t)(m0 + wave * 8 + srow) * K + schunk * 8;                       // h = 0, piece `wave`
    const bf16_t* w_src = W + (size_t)(n0 + (wave >> 2) * 64 + (wave & 3) * 8 + srow) * K + schunk * 8;
    const size_t a_p2 = (size_t)128 * K, a_h = (size_t)64 * K;   // second piece = other wave row; half 1 = +64 rows
    const size_t w_p2 = (size_t)128 * K, w_h = (size_t)32 * K;   // second piece = wave col + 2;   half 1 = +32 cols
    char* const lds_piece = smem + wave * 1024;

    // slot of (buffer b, operand o, half h) = ((b * 2 + o) * 2 + h) * 16 KiB
    auto stage_a = [&](int kt, int h) {
        char* dst = lds_piece + ((kt & 1) * 4 + h) * PP_HALF;
        const bf16_t* src = a_src + h * a_h + (size_t)kt * PP_BK;
        glds16(src, dst);
        glds16(src + a_p2, dst + 8192);
    };
    auto stage_b = [&](int kt, int h) {
        char* dst = lds_piece + ((kt & 1) * 4 + 2 + h) * PP_HALF;
        const bf16_t* src = w_src + h * w_h + (size_t)kt * PP_BK;
        glds16(src, dst);
        glds16(src + w_p2, dst + 8192);
    };

    // ---- fragment reads: row (lane&15) of a 16-row tile, chunk ks*4 + lane/16, swizzled ----
    const int foff = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
    const int a_rd = wm * 8192 + foff;                           // + i * 2048 (m-tile i of the half), ^ 64 for ks = 1
    const int b_rd = 2 * PP_HALF + wn * 4096 + foff;             // + j * 2048 (n-tile j of the half)

    f32x4 acc[2][4][2][2];                                       // [A half][m-tile][B half][n-tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 af[2][4], bf0[2][2], bf1[2][2];                       // [ks][tile]

#define PP_READ_A(buf, h)                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)              \
        af[ks][i] = *(const bf16x8*)((buf) + (h) * PP_HALF + ((a_rd + i * 2048) ^ (ks * 64)))
#define PP_READ_B(dst, buf, h)                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j)              \
        dst[ks][j] = *(const bf16x8*)((buf) + (h) * PP_HALF + ((b_rd + j * 2048) ^ (ks * 64)))
#define PP_MFMA(ha, hb, bfr)                                                                                    \
    do {                                                                                                        \
        __builtin_amdgcn_s_barrier();                                                                           \
        DP_LGKMCNT0();                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
                acc[ha][i][hb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[ha][i][hb][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        __builtin_amdgcn_s_barrier();                                                                           \
    } while (0)

    // ---- prologue: half-tiles in staging order A0(0) B0(0) B1(0) A1(0) A0(1) B0(1) ----
    stage_a(0, 0); stage_b(0, 0); stage_b(0, 1); stage_a(0, 1); stage_a(1, 0); stage_b(1, 0);
    DP_VMCNT(8);                                                 // A0(0), B0(0) have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                   // wave row 1 runs one barrier behind

    // One K-tile.  ST1: tile kt+1 exists (stage B1, A1 of it); ST2: tile kt+2 exists (stage A0, B0 of it).
    // V0..V3: vmcnt after the stage of phase c (8 in steady state; smaller in the tail where stages are skipped).
#define PP_TILE(kt, ST1, ST2, V0, V1, V2, V3)                                                                   \
    do {                                                                                                        \
        const char* buf = smem + ((kt) & 1) * 4 * PP_HALF;                                                      \
        /* c0 */                                                                                                \
        PP_READ_B(bf0, buf, 0);                                                                                 \
        PP_READ_A(buf, 0);                                                                                      \
        if (ST1) stage_b((kt) + 1, 1);                                                                          \
        DP_VMCNT(V0);                                                                                           \
        PP_MFMA(0, 0, bf0);                                                                                     \
        /* c1 */                                                                                                \
        PP_READ_B(bf1, buf, 1);                                                                                 \
        if (ST1) stage_a((kt) + 1, 1);                                                                          \
        DP_VMCNT(V1);                                                                                           \
        PP_MFMA(0, 1, bf1);                                                                                     \
        /* c2 */                                                                                                \
        PP_READ_A(buf, 1);                                                                                      \
        if (ST2) stage_a((kt) + 2, 0);                                                                          \
        DP_VMCNT(V2);                                                                                           \
        PP_MFMA(1, 1, bf1);                                                                                     \
        /* c3 */                                                                                                \
        if (ST2) stage_b((kt) + 2, 0);                                                                          \
        DP_VMCNT(V3);                                                                                           \
        PP_MFMA(1, 0, bf0);                                                                                     \
    } while (0)

    int kt = 0;
    for (; kt < nk - 2; ++kt) PP_TILE(kt, true, true, 8, 8, 8, 8);
    // tile nk-2: its c2/c3 stages do not exist; in flight after c1 are A0,B0,B1,A1 of tile nk-1
    PP_TILE(kt, true, false, 8, 8, 8, 4);                        // c3: A0(nk-1), B0(nk-1) must land -> 2 newer half-tiles
    ++kt;
    PP_TILE(kt, false, false, 2, 0, 0, 0);                       // c0: B1(nk-1) must land -> 1 newer; c1: A1(nk-1)
    if (wm == 0) __builtin_amdgcn_s_barrier();                   // balance wave row 1's extra barrier

#undef PP_TILE
#undef PP_MFMA
#undef PP_READ_A
#undef PP_READ_B

    // ---- epilogue: lane holds out[m = lane&15][n = 4*(lane>>4) + 0..3] of each 16x16 tile ----
    const int em = m0 + wm * 128 + (lane & 15);
    const int en = n0 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = en + hb * 32 + j * 16;
            const float4 bv = *(const float4*)(bias + n);
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 c = acc[ha][i][hb][j];
                    float v0 = c[0] + bv.x, v1 = c[1] + bv.y, v2 = c[2] + bv.z, v3 = c[3] + bv.w;
                    if (ACT) { v0 = silu_fast(v0); v1 = silu_fast(v1); v2 = silu_fast(v2); v3 = silu_fast(v3); }
                    const size_t o = (size_t)(em + ha * 64 + i * 16) * ldo + n;
                    if (OUT_F32) *(float4*)((float*)out + o) = make_float4(v0, v1, v2, v3);
                    else *(uint2*)((bf16_t*)out + o) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                }
        }
}

}  // namespace dp
