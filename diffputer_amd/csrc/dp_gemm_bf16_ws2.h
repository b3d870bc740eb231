// Weight-stationary bf16 MFMA GEMM for K = 256 (the folded proj+mlp.0 layer of tables wider than 128 columns), with the K
// loop, every LDS access and every wait written as inline asm in the style of k_gemm_bf16_w4.
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] )        A: activations [M, 256], W: torch Linear weight [N, 256]
//
// Structure = k_gemm_bf16_ws (dp_gemm_bf16_ws.h): the workgroup's W tile (256 columns x 256 k = 128 KiB) is loaded into LDS
// once and stays; a WAVE owns 32 rows x 256 columns of a row group (128 accumulator registers, here in AGPRs), streams its
// activation fragments from global memory into a four-k-step register ring, and after the set-up barrier no wave waits for
// another: two waves per SIMD, one's SiLU epilogue (VALU) under the other's MFMAs.
// What hipcc's schedule of that loop costs (K loops of 6.5-12 k cycles for 4.1 k cycles of MFMAs, a
// drain of every load in flight before compiler-visible LDS traffic) is what this file removes:
//   * W fragments: one ds_read_b128 per MFMA pair into a ring of eight, issued SEVEN pairs ahead, `s_waitcnt lgkmcnt(7)`
//     before the pair (tools/issue_probe.hip mode 8: 18 cycles per MFMA on one wave).  The ring runs dry at a row group's
//     end and is refilled at the next one's start: its 32 registers are the epilogue's temporaries (kept across the epilogue
//     the kernel spilled 38 registers, and a spill reload is a vmcnt(0)).
//   * activation fragments: two global_load_dwordx4 per k-step, issued when the ring slot's last MFMA has been issued (four
//     k-steps = half a row group ahead); counted vmcnt before each k-step's first MFMA.  The in-order vmcnt queue also holds
//     the previous row group's WS2_NST stores: vmcnt(6 + NST) for the first four k-steps, vmcnt(6) afterwards.
//   * epilogue: accumulators read from AGPRs where they are used, transposer and bias traffic as asm ds_* (nothing
//     compiler-visible touches LDS after the barrier, so hipcc inserts no wait of its own).
// Per output element the MFMA sequence is that of k_gemm_bf16 (k-steps of 32, ascending): bit-identical results.
#pragma once
#include <utility>

#include "dp_gemm_bf16_pp.h"
#include "dp_gemm_bf16_ws.h"

namespace dp {

constexpr int WS2_THREADS = 512, WS2_BN = 256, WS2_ROWS = 32, WS2_K = 256;
constexpr int WS2_W_BYTES = 4 * WS2_BN * 128;                    // 128 KiB: four K-tiles of 256 columns x 64 k
constexpr int WS2_STAGE_WAVE = 2048;                             // epilogue transposer per wave (16 rows x 128 B)
constexpr int WS2_BIAS_OFF = WS2_W_BYTES + 8 * WS2_STAGE_WAVE;
constexpr int WS2_TICKET_OFF = WS2_BIAS_OFF + WS2_BN * 4;
constexpr int WS2_LDS_TOTAL = WS2_TICKET_OFF + 16;
#if defined(WS2_DIAG_NOSTORE) || defined(WS2_DIAG_NOEPI)         /* timing-only diagnostics (tools/build_diag.sh): results are not stored */
constexpr int WS2_NST = 0;
#else
constexpr int WS2_NST = 16;                                      // global stores one wave's epilogue issues per row group
#endif

struct Ws2Regs {
    unsigned wr[2][2];                                           // W fragment bases: [k-step parity][k-step >= 4]
    unsigned a_lane[2];                                          // activation fragment lane offsets (row tile 0 / 1)
};

// MFMA pair F (0..127) of a row group: k-step F / 16, column tile F % 16, both row tiles.  Behind it in the instruction
// stream: the read of the fragment seven pairs ahead.  (A wave's FIRST row group has no stores in the queue, but its first
// four k-steps were awaited with vmcnt(0) before the barrier, so the lenient count is harmless there -- and one K-loop
// variant keeps hipcc from preserving one variant's live-in registers across the other: that cost 38 spilled registers.)
template <int F>
__device__ __forceinline__ void ws2_pair(f32x4 (&acc)[2][16], bf16x8 (&wf)[8], bf16x8 (&af)[4][2], const Ws2Regs& r,
                                         const char* a_cur, const char* a_next) {
    constexpr int s = F >> 4, j = F & 15;
    if constexpr (j == 0) {                                      // this k-step's activation fragments have landed
        if constexpr (s >= 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + WS2_NST) : "memory");
    }
    if constexpr (F + 7 < 128) {
        constexpr int FN = F + 7, sn = FN >> 4, jn = FN & 15;
        constexpr int off = ((sn >> 1) & 1) * 32768 + jn * 2048;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[FN & 7]) : "v"(r.wr[sn & 1][sn >> 2]), "n"(off));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
    } else {                                                     // the ring runs dry at the row group's end (its registers are the
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(127 - F) : "memory");   // epilogue's): fragments F .. 127 are all that is in flight
    }
    if constexpr (s == 0) {                                      // first touch of the accumulators: C = 0
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[0][j]) : "v"(wf[F & 7]), "v"(af[s & 3][0]));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[1][j]) : "v"(wf[F & 7]), "v"(af[s & 3][1]));
    } else {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[0][j]) : "v"(wf[F & 7]), "v"(af[s & 3][0]));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[1][j]) : "v"(wf[F & 7]), "v"(af[s & 3][1]));
    }
    if constexpr (j == 15) {                                     // ring slot s % 4 has fed its last MFMA: refill it
        constexpr int sl = (s + 4) & 7;                          // k-step s + 4 of this row group, or s - 4 of the next one
        const char* const base = s < 4 ? a_cur : a_next;
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(af[s & 3][0]) : "v"(r.a_lane[0]), "s"(base), "n"(sl * 64) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(af[s & 3][1]) : "v"(r.a_lane[1]), "s"(base), "n"(sl * 64) : "memory");
    }
}
template <int... F>
__device__ __forceinline__ void ws2_kloop(std::integer_sequence<int, F...>, f32x4 (&acc)[2][16], bf16x8 (&wf)[8], bf16x8 (&af)[4][2],
                                          const Ws2Regs& r, const char* a_cur, const char* a_next) {
    (ws2_pair<F>(acc, wf, af, r, a_cur, a_next), ...);
}

// Grid: a multiple of nbn = N / 256 workgroups, at most one per CU; workgroup b works on column tile b % nbn and on the
// 32-row groups slot, slot + slots, ... (slot = b / nbn, slots = gridDim / nbn), drawn one at a time from an LDS ticket.
template <bool ACT>
__global__ __launch_bounds__(WS2_THREADS) void k_gemm_bf16_ws2(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                                const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                int ldo, int nbn, int n_rg) {
    constexpr int K = WS2_K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef WS_STAMP                                                  /* diagnostic build only (tools/ws_stamps.py --ws 3) */
    unsigned long long st_k0 = 0, st_a = 0, st_b = 0, st_c = 0;
    int st_i = 0;
    WS_STAMP_AT(st_k0);
#endif
    const int n0 = (blockIdx.x % nbn) * WS2_BN;
    const int slot = blockIdx.x / nbn, slots = gridDim.x / nbn;
    const unsigned lds0 = lds_u32(smem);

    // ---- set-up (compiler-visible): the W tile, K-tile by K-tile, as in k_gemm_bf16_ws; bias; ticket ----
    {
        const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int piece = wave + p * 8;
                glds16(W + (size_t)(n0 + piece * 8 + srow) * K + kt * 64 + schunk * 8, smem + kt * (WS2_BN * 128) + piece * 1024);
            }
    }
    if (tid < WS2_BN) ((float*)(smem + WS2_BIAS_OFF))[tid] = bias[n0 + tid];
    if (tid == 0) *(int*)(smem + WS2_TICKET_OFF) = 8;            // tickets 0..7 are the waves' first row groups

    Ws2Regs r;
    const unsigned foff = (unsigned)((lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4));
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) r.wr[par][hi] = lds0 + (foff ^ (par * 64)) + hi * 65536;
    r.a_lane[0] = (unsigned)(((lane & 15) * K + (lane >> 4) * 8) * 2);
    r.a_lane[1] = r.a_lane[0] + 16 * K * 2;

    const int n_mine = n_rg > slot ? (n_rg - slot + slots - 1) / slots : 0;
    int ti = wave;
    if (ti >= n_mine) {                                          // a wave without work still takes part in the one barrier
        DP_VMCNT(0);
        __syncthreads();
        return;
    }
    int rg = slot + ti * slots;
    const char* a_cur = (const char*)A + (size_t)rg * (WS2_ROWS * K * 2);

    f32x4 acc[2][16];
    bf16x8 wf[8], af[4][2];
    // the first row group's first four k-steps
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        af[s][0] = *(const bf16x8*)(a_cur + s * 64 + r.a_lane[0]);
        af[s][1] = *(const bf16x8*)(a_cur + s * 64 + r.a_lane[1]);
    }
    DP_VMCNT(0);                                                 // this wave's W pieces and first fragments have landed
    __syncthreads();                                             // ... and everyone else's: the only barrier of the kernel
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(af[s][0]), "+v"(af[s][1]));
#ifndef WS_DIAG_NOSLEEP
    if (wave >= 4) __builtin_amdgcn_s_sleep(32);                 // start the second wave of each SIMD about half a K loop late
#endif

    // epilogue addresses (the transposer of k_gemm_bf16_ws: 16 rows x 128 B per wave, 16-B chunk c of row r at c ^ (r & 7))
    const unsigned stg = lds0 + WS2_W_BYTES + wave * WS2_STAGE_WAVE;
    const unsigned ticket_ad = lds0 + WS2_TICKET_OFF;

    for (;;) {
        // next ticket (one lane draws it; the wait also covers the ring reads in flight, which is harmless)
        int tn = 0;
        if (lane == 0) asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(tn) : "v"(ticket_ad), "v"(1) : "memory");
        tn = __builtin_amdgcn_readfirstlane(tn);
        // the row group whose fragments are requested next: this wave's next one, or (at the end) the current one again --
        // always a valid address, so the loads need no branch and the counted waits stay exact
        // W fragment ring: fragments 0..6 (the K loop's first wait, lgkmcnt(7) behind the read of fragment 7, is for fragment 0)
#define WS2_RING0(f) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[f]) : "v"(r.wr[0][0]), "n"((f) * 2048))
        WS2_RING0(0); WS2_RING0(1); WS2_RING0(2); WS2_RING0(3); WS2_RING0(4); WS2_RING0(5); WS2_RING0(6);
#undef WS2_RING0
        const int next = tn < n_mine ? slot + tn * slots : rg;
        const char* const a_next = (const char*)A + (size_t)next * (WS2_ROWS * K * 2);
        WS_STAMP_AT(st_a);
        ws2_kloop(std::make_integer_sequence<int, 128>{}, acc, wf, af, r, a_cur, a_next);
        WS_STAMP_AT(st_b);

        // ---- epilogue: lane holds out[row t*16 + (lane&15)][col j*16 + 4*(lane>>4) + 0..3] ----
        int elane;                                               // recomputed here (volatile: not hoisted above the K loop)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
        const unsigned bias_ad = lds0 + WS2_BIAS_OFF + (elane >> 4) * 16;
        const int wr_ = elane & 15, wq = elane >> 4, rr = elane >> 3, rc = elane & 7;
        const unsigned wa = stg + wr_ * 128 + (((wq >> 1) ^ (wr_ & 7)) << 4) + (wq & 1) * 8;      // ^ (jj * 32)
        const unsigned ra = stg + rr * 128 + ((rc ^ rr) << 4);                                    // rows rr and rr + 8
        bf16_t* const orow = out + ((size_t)rg * WS2_ROWS + rr) * ldo + n0 + rc * 8;
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        float4 bv[4];
        uint2 pk[4];
        u32x4_t rd0, rd1;
        // step = (64-column chunk c = step / 2, row tile t = step % 2): 16 rows x 64 columns
#define WS2_EPI_BIAS(c)                                                                                         \
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t" \
                 "ds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"                                        \
                 : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3])                                       \
                 : "v"(bias_ad), "n"((c) * 256), "n"((c) * 256 + 64), "n"((c) * 256 + 128), "n"((c) * 256 + 192) : "memory")
#define WS2_EPI_COMPUTE(c, t)                                                                                   \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                          \
        f32x4 v;                                                                                                \
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\t"                               \
                     "v_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"                                   \
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])                                       \
                     : "a"(acc[t][(c) * 4 + jj][0]), "a"(acc[t][(c) * 4 + jj][1]), "a"(acc[t][(c) * 4 + jj][2]),  \
                       "a"(acc[t][(c) * 4 + jj][3]));                                                           \
        const f32x2_t s01 = (f32x2_t){v[0], v[1]} + (f32x2_t){bv[jj].x, bv[jj].y};                              \
        const f32x2_t s23 = (f32x2_t){v[2], v[3]} + (f32x2_t){bv[jj].z, bv[jj].w};                              \
        float v0 = s01.x, v1 = s01.y, v2 = s23.x, v3 = s23.y;                                                   \
        if (ACT) { silu_fast2(v0, v1); silu_fast2(v2, v3); }                                                    \
        pk[jj] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));                                          \
    }
#define WS2_EPI_WRITE()                                                                                         \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) asm volatile("ds_write_b64 %0, %1" ::"v"(wa ^ (unsigned)(jj << 5)), "v"(pk[jj]) : "memory")
        // one step: the transposed rows of step STEP are in flight while step STEP + 1 is computed into registers
#define WS2_EPI_STEP(STEP)                                                                                      \
    do {                                                                                                        \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(rd0), "=&v"(rd1) : "v"(ra) : "memory"); \
        if ((STEP) < 7) {                                                                                       \
            if (((STEP) & 1) == 1) { WS2_EPI_BIAS((((STEP) >> 1) + 1) & 3); }                                   \
            if (((STEP) & 1) == 0) { WS2_EPI_COMPUTE((STEP) >> 1, 1); }                                         \
            else { WS2_EPI_COMPUTE((((STEP) >> 1) + 1) & 3, 0); }                                               \
        }                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rd0), "+v"(rd1) :: "memory");                                \
        bf16_t* const o = orow + (size_t)(((STEP) & 1) * 16) * ldo + ((STEP) >> 1) * 64;                        \
        if (WS2_NST) {                                                                                          \
            *(u32x4_t*)(o) = rd0;                                                                               \
            *(u32x4_t*)(o + (size_t)8 * ldo) = rd1;                                                             \
        } else asm volatile("" ::"v"(rd0), "v"(rd1), "v"(o));                                                   \
        if ((STEP) < 7) { WS2_EPI_WRITE(); }                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)
#ifdef WS2_DIAG_NOEPI                                            /* timing-only diagnostic: K loops only */
        if (false) {
#else
        {
#endif
        WS2_EPI_BIAS(0);
        WS2_EPI_COMPUTE(0, 0);
        WS2_EPI_WRITE();
        WS2_EPI_STEP(0); WS2_EPI_STEP(1); WS2_EPI_STEP(2); WS2_EPI_STEP(3);
        WS2_EPI_STEP(4); WS2_EPI_STEP(5); WS2_EPI_STEP(6); WS2_EPI_STEP(7);
        }
#undef WS2_EPI_STEP
#undef WS2_EPI_WRITE
#undef WS2_EPI_COMPUTE
#undef WS2_EPI_BIAS
#ifdef WS_STAMP
        WS_STAMP_AT(st_c);
        if (g_pp_stamp_buf && st_i < 16 && lane == 0) {
            unsigned long long* d = g_pp_stamp_buf + (((size_t)blockIdx.x * 8 + wave) * 16 + st_i) * 4;
            d[0] = st_k0; d[1] = st_a; d[2] = st_b; d[3] = st_c;
        }
        ++st_i;
#endif
        if (tn >= n_mine) break;
        ti = tn;
        rg = next;
        a_cur = a_next;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the fragments requested past the last row group
}

}  // namespace dp
