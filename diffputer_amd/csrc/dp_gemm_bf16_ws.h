// Weight-stationary bf16 MFMA GEMM for the denoiser's SHALLOW Linear layers (K <= 256): the folded proj+mlp.0 layer
// (model.py:102,105,126-128; K = d_in padded to 64) and, unfolded, proj itself.
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] )        A: activations [M, K], W: torch Linear weight [N, K]
//
// Why a third kernel: with K = 256 a 256x256 output tile is only four K-tiles deep, and k_gemm_bf16_pp spends as long in
// its SiLU / store epilogue (exp + rcp per element, quarter rate) as in its K loop, with the matrix pipes idle meanwhile
// (57 us per 32 768 x 2048 launch = 600 TFLOP/s).  Its two wave rows cannot drift apart to overlap one's epilogue with
// the other's MFMAs: every phase ends in a workgroup barrier, because each wave stages operand pieces the others read.
// Here nothing is shared after set-up:
//   * the workgroup's whole W tile (256 columns x K <= 256: at most 128 KiB) is loaded into LDS ONCE (LDS-DMA, the XOR-
//     swizzled image of k_gemm_bf16) and stays; a workgroup keeps its column tile for the whole launch (blocks b and b+8
//     share an XCD, and with N = 2048 an XCD's 32 workgroups share one column tile, so W is read from HBM/L2 once per XCD);
//   * a WAVE owns 32 rows x 256 columns (128 accumulator registers, as in the other kernels) of a "row group" and loads
//     its activation fragments straight from global memory into registers (16 B per lane, each byte loaded once per
//     workgroup; the next row group's fragments are requested as soon as the current ones have fed their last MFMA);
//   * so after the single set-up barrier no wave waits for another: the eight waves run free, and on each SIMD one
//     wave's epilogue (VALU) runs under its partner's MFMAs.  Waves 4-7 start half a row group late to set that up.
// LDS reads: one ds_read_b128 of W per two MFMAs (512 B per MFMA, 50 % of the LDS read rate with all four SIMDs in their
// K loops).  The accumulation order over k is that of k_gemm_bf16 (k-steps of 32, ascending), so results are bit-identical.
#pragma once
#include "dp_gemm_bf16.h"

namespace dp {

// Diagnostic build only (-DDP_PP_STAMP -DWS_STAMP, tools/ws_stamps.py): s_memtime stamps per wave and row group into the
// debug buffer of dp_gemm_bf16_pp.h ([block][wave][16][4] u64); no stamp executes in the shipped library.
#ifdef WS_STAMP
extern __device__ unsigned long long* g_pp_stamp_buf;
#define WS_STAMP_AT(var)                                                                                        \
    do {                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)
#else
#define WS_STAMP_AT(var) do { } while (0)
#endif

constexpr int WS_BN = 256, WS_ROWS = 32, WS_THREADS = 512;
constexpr int WS_STAGE_WAVE = 2048;                              // epilogue transposer per wave (16 rows x 128 B)
constexpr int ws_lds_bytes(int kt) { return kt * WS_BN * 128 + 8 * WS_STAGE_WAVE + WS_BN * 4 + 16; }   // W tile | transposers | bias tile | ticket

// KT = K / 64 (1..4).  Grid: a multiple of nbn = N / 256 workgroups, at most one per CU; workgroup b works on column tile
// b % nbn and on the 32-row groups slot, slot + slots, ... (slot = b / nbn, slots = gridDim / nbn), which its eight waves
// draw one at a time from a ticket counter in LDS.
template <int KT, bool ACT, bool OUT_F32>
__global__ __launch_bounds__(WS_THREADS) void k_gemm_bf16_ws(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ out,
                                                              int ldo, int nbn, int n_rg) {
    constexpr int K = KT * 64, KS = KT * 2;                      // k-steps of 32
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WS_STAMP
    unsigned long long st_k0 = 0, st_a = 0, st_b = 0, st_c = 0;
    int st_i = 0;
    WS_STAMP_AT(st_k0);
#endif
    const int n0 = (blockIdx.x % nbn) * WS_BN;
    const int slot = blockIdx.x / nbn, slots = gridDim.x / nbn;

    // ---- set-up: the W tile, K-tile by K-tile: [256 cols][64 k] = 32 pieces of 8 rows x 128 B, chunk c of row r at c ^ (r&7) ----
    {
        const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int piece = wave + p * 8;
                glds16(W + (size_t)(n0 + piece * 8 + srow) * K + kt * 64 + schunk * 8, smem + kt * (WS_BN * 128) + piece * 1024);
            }
    }

    // the column tile's bias: 256 floats behind the transposers.  The epilogue reads it from LDS (lgkmcnt), not from global
    // memory: a global load there would sit behind the row group's stores in the in-order vmcnt queue, and waiting for
    // it would wait for every store to complete.
    float* const bias_lds = (float*)(smem + KT * (WS_BN * 128) + 8 * WS_STAGE_WAVE);
    if (tid < WS_BN) bias_lds[tid] = bias[n0 + tid];
    // row groups are handed out by a ticket counter in LDS: on each SIMD the older wave wins the issue arbitration and
    // gets through its row groups faster, so a static split leaves the younger waves a tail with nobody to overlap with
    int* const ticket = (int*)(bias_lds + WS_BN);
    if (tid == 0) *ticket = 8;                                   // tickets 0..7 are the waves' first row groups

    // ---- per-lane addresses ----
    // W fragment (col tile j, k-step s): row (lane&15) of the tile, chunk (s&1)*4 + lane/16, swizzled; K-tile s/2
    const int foff = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
    const char* const wrd[2] = {smem + foff, smem + (foff ^ 64)};                          // + (s/2) * 32 KiB + j * 2 KiB
    // activation fragment (row tile t, k-step s): row t*16 + (lane&15), k = s*32 + (lane/16)*8
    // (addresses are a wave-uniform 64-bit base in SGPRs + a 32-bit per-lane byte offset: two address VGPRs in all)
    const unsigned a_lane0 = (unsigned)(((lane & 15) * K + (lane >> 4) * 8) * 2);
    const unsigned a_lane1 = a_lane0 + 16 * K * 2;
    // epilogue transposer (the one of k_gemm_bf16_pp, 16 rows at a time): 128-B rows, 16-B chunk c of row r at c ^ (r&7).
    //   write side: lane holds the 8-byte unit u = jj*4 + lane/16 of row lane&15: chunk u/2, half u&1
    //   read side:  lane reads chunk lane&7 of rows lane/8 + 8 q
    char* const stg = smem + KT * (WS_BN * 128) + wave * WS_STAGE_WAVE;
    const int wr = lane & 15, wq = lane >> 4;
    const int wa = wr * 128 + (((wq >> 1) ^ (wr & 7)) << 4) + (wq & 1) * 8;                // ^ (jj * 32)
    const int rr = lane >> 3, rc = lane & 7;
    const int ra = rr * 128 + ((rc ^ rr) << 4);                                            // + q * 1024
    const unsigned o_lane = (unsigned)((rr * ldo + rc * 8) * 2);                           // bf16 out: row rr, chunk rc
    const unsigned o_lane_f32 = (unsigned)((wr * ldo + wq * 4) * 4);                       // f32 out: row wr, 4 columns at wq*4

    // this workgroup's row groups: slot + i * slots, i = 0, 1, ...  (ticket i -> row group)
    const int n_mine = n_rg > slot ? (n_rg - slot + slots - 1) / slots : 0;
    auto rg_of = [&](int i) { return slot + i * slots; };
    int ti = wave;                                               // my current ticket
    int rg = rg_of(ti);

    // activation fragments: a ring of RD k-steps; the slot of k-step s is refilled with k-step s + RD of the stream
    // (this row group's, then the next one's) as soon as its MFMAs are issued -- three k-steps (>= 1 500 cycles) ahead
    constexpr int RD = KS == 8 ? 4 : KS;
    bf16x8 af[RD][2];
    auto load_a = [&](int g, int s, int slot) {
        const char* ab = (const char*)A + (size_t)g * (WS_ROWS * K * 2) + s * 64;     // wave-uniform
        af[slot][0] = *(const bf16x8*)(ab + a_lane0);
        af[slot][1] = *(const bf16x8*)(ab + a_lane1);
    };
    if (ti < n_mine) {
#pragma unroll
        for (int s = 0; s < RD; ++s) load_a(rg, s, s);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's W pieces (and its first fragments) have landed
    __syncthreads();                                             // ... and everyone else's: the only barrier of the kernel
#ifdef WS_DIAG_HALF                                              /* timing-only diagnostic: one wave per SIMD (row groups of waves 4-7 are skipped) */
    if (wave >= 4) return;
#endif
#ifndef WS_DIAG_NOSLEEP
    if (wave >= 4) __builtin_amdgcn_s_sleep(KT * 8);             // start the second wave of each SIMD ~half a K loop late
#endif

    while (ti < n_mine) {
        // next ticket (wave-uniform: one lane draws it)
        int tn = 0;
        if (lane == 0) tn = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tn = __builtin_amdgcn_readfirstlane(tn);
        WS_STAMP_AT(st_a);
        f32x4 acc[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the row group whose fragments are requested next: this wave's next one, or (at the end) the current one again --
        // always a valid address, so the loads need no branch and hipcc can count them exactly in its vmcnt waits
        const int next = tn < n_mine ? rg_of(tn) : rg;
        // W fragments: a ring of eight, each read seven MFMA pairs (~224 MFMA cycles) ahead of the pair that consumes it -- a
        // group of four read one group ahead (128 cycles) does not cover the LDS latency: K loop 6.2 k cycles for 4.1 k of
        // MFMAs, measured with one wave per SIMD.  The scheduling barriers keep hipcc from hoisting more reads (it would spill).
        constexpr int NF = KS * 16, WD = 8;                      // fragments per row group (k-step major, column tile minor)
        bf16x8 wf[WD];
        auto read_w = [&](int f) {
            const int s = f >> 4, j = f & 15;
            wf[f % WD] = *(const bf16x8*)(wrd[s & 1] + (s >> 1) * (WS_BN * 128) + j * 2048);
        };
#pragma unroll
        for (int f = 0; f < WD - 1; ++f) read_w(f);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int s = f >> 4, j = f & 15;
            if (f + WD - 1 < NF) read_w(f + WD - 1);
#ifdef WS_DIAG_NOMFMA                                            /* timing-only diagnostic: fragments are read, products skipped */
            asm volatile("" :: "v"(wf[f % WD]), "v"(af[s % RD][0]), "v"(af[s % RD][1]));
#else
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f % WD], af[s % RD][0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[f % WD], af[s % RD][1], acc[1][j], 0, 0, 0);
#endif
            if (j == 15) {                                       // this ring slot has fed its last MFMA
                if (s + RD < KS) load_a(rg, s + RD, s % RD);
                else load_a(next, s + RD - KS, s % RD);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        WS_STAMP_AT(st_b);
        // ---- epilogue: lane holds out[row t*16 + (lane&15)][col j*16 + 4*(lane>>4) + 0..3]; four chunks of 64 columns ----
        const size_t row0 = (size_t)rg * WS_ROWS;
        // Eight steps of 16 rows x 64 columns: SiLU the step's four fragments, write them to the transposer, read them back
        // as whole 128-B rows, store.  The LDS round trip of step i (writes, reads, ~200 cycles) runs under the VALU work of
        // step i + 1: its stores are placed AFTER that work.  Same-wave LDS operations execute in issue order, so step
        // i + 1's writes queue behind step i's reads and one staging area is enough.
        // The bias of a 64-column chunk is read one chunk ahead for the same reason: LDS reads return in order, so a bias
        // read issued behind a step's transposer traffic would make the next step's first add wait for all of it.
        uint4 rd0 = make_uint4(0, 0, 0, 0), rd1 = rd0;
        int pc = 0, pt = 0;                                      // the step whose rows are in rd0 / rd1
        float4 bv[4], bn[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bn[jj] = *(const float4*)(bias_lds + jj * 16 + wq * 4);
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int c = step >> 1, t = step & 1;
            if (t == 0) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bv[jj] = bn[jj];
                if (c < 3) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) bn[jj] = *(const float4*)(bias_lds + (c + 1) * 64 + jj * 16 + wq * 4);
                }
            }
            uint2 pk[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 v = acc[t][c * 4 + jj];
                float v0 = v[0] + bv[jj].x, v1 = v[1] + bv[jj].y, v2 = v[2] + bv[jj].z, v3 = v[3] + bv[jj].w;
                if (ACT) { silu_fast2(v0, v1); silu_fast2(v2, v3); }
                if (OUT_F32)
                    *(float4*)((char*)out + ((row0 + t * 16) * ldo + n0 + c * 64 + jj * 16) * 4 + o_lane_f32) = make_float4(v0, v1, v2, v3);
                else
                    pk[jj] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
            if (!OUT_F32) {
                __builtin_amdgcn_sched_barrier(0);               // keep the previous step's stores behind this step's arithmetic
                if (step > 0) {
#ifdef WS_DIAG_NOSTORE                                           /* timing-only diagnostic: transposed rows are read, not stored */
                    if (n_rg < 0)
#endif
                    {
                        *(uint4*)((char*)out + ((row0 + pt * 16) * ldo + n0 + pc * 64) * 2 + o_lane) = rd0;
                        *(uint4*)((char*)out + ((row0 + pt * 16 + 8) * ldo + n0 + pc * 64) * 2 + o_lane) = rd1;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) *(uint2*)(stg + (wa ^ (jj * 32))) = pk[jj];
                rd0 = *(const uint4*)(stg + ra);
                rd1 = *(const uint4*)(stg + ra + 1024);
                pc = c; pt = t;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!OUT_F32) {
#ifdef WS_DIAG_NOSTORE
            if (n_rg < 0)
#endif
            {
                *(uint4*)((char*)out + ((row0 + pt * 16) * ldo + n0 + pc * 64) * 2 + o_lane) = rd0;
                *(uint4*)((char*)out + ((row0 + pt * 16 + 8) * ldo + n0 + pc * 64) * 2 + o_lane) = rd1;
            }
        }
#ifdef WS_STAMP
        WS_STAMP_AT(st_c);
        if (g_pp_stamp_buf && st_i < 16 && lane == 0) {
            unsigned long long* d = g_pp_stamp_buf + (((size_t)blockIdx.x * 8 + wave) * 16 + st_i) * 4;
            d[0] = st_k0; d[1] = st_a; d[2] = st_b; d[3] = st_c;
        }
        ++st_i;
#endif
        ti = tn;
        rg = next;
    }
}

}  // namespace dp
