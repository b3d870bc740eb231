// bf16 MFMA GEMM for the deep layers (K >= 256), FOUR waves per workgroup: one wave per SIMD, 128 x 128 of the 256 x 256
// tile each, accumulators in the 256 AGPRs of a 512-register wave.
//
//   out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] )        A: activations [M, K], W: torch Linear weight [N, K]
//
// Why beside k_gemm_bf16_pp: with eight 128 x 64 wave tiles every K-tile costs 8 x 24 = 192 KiB of LDS fragment reads;
// four 128 x 128 wave tiles read 128 KiB, and four waves need a quarter of the barrier traffic.  There is no partner wave
// to hide behind, so the wave pipelines itself, and the budget is the instruction issue: a wave issues one instruction
// per 4 cycles, a 16x16x32 MFMA holds the matrix pipe for 16, so TWO other instructions fit behind every MFMA and a
// third delays the next one (measured: a 6-instruction DMA sequence per MFMA pair cost 19 cycles each).
//   * K-tile (64 deep) = four 16-KiB half-tiles A0/A1 (first/second 64 rows of both wave rows), B0/B1 (first/second 64
//     columns of both wave columns); two K-tile buffers = 128 KiB of LDS; image of a half-tile as in k_gemm_bf16.
//   * K-tile g = four phases; phase = { 32 MFMA with, behind them, 8 fragment reads for the NEXT phase and the LDS-DMA
//     of ONE half-tile, 7 phases ahead of its read ; counted vmcnt + lgkmcnt(0) ; s_barrier }:
//       phase   reads (into)          LDS-DMA      MFMA quadrant
//       p0      B1(g)   -> by         B0(g+2)      q(A0, B0)      B0 in bx
//       p1      A1(g)   -> a1         B1(g+2)      q(A0, B1)
//       p2      A0(g+1) -> a0         A1(g+2)      q(A1, B1)
//       p3      B0(g+1) -> by         A0(g+3)      q(A1, B0)      bx / by swap roles every K-tile
//     RAW: a half-tile read in phase P was staged in phase P-7; the vmcnt(24) that ends phase P-1 leaves only the six
//     newer half-tiles (4 DMA instructions per wave each) in flight, and the barrier after it covers the other waves'
//     pieces.  WAR: a slot is re-staged in the phase after its last read, whose lgkmcnt(0) + barrier every wave passed.
//   * the K-tile stream does not stop at an output tile's end: the stage cursor walks on into the workgroup's next tile
//     (persistent, tile = round * gridDim + xcd_remap), so the epilogue runs with seven half-tiles of the next tile in
//     flight and the first fragments of the next tile already in registers.  Past the workgroup's last tile the cursor
//     parks on the first K-tile of its first one: those loads land in slots nobody reads again and keep every wait a
//     counted one (they re-read one 32-KiB pair of half-tiles instead of streaming on).  For six
//     phases after an epilogue its W4_NST stores sit between the half-tiles in the in-order vmcnt queue: vmcnt(24 + NST).
//   * the phase body is inline asm throughout (MFMA with its accumulator pinned to AGPRs, ds_read_b128, the DMA with its
//     M0 write one MFMA earlier, the waits): hipcc's own schedule of the same loop spills accumulators, and its waits for
//     compiler-visible LDS reads drain the DMA.  Source order is issue order.
//   * bias vector in LDS (a global load at the epilogue would wait behind the DMA in the in-order vmcnt queue).
//   * per output element the same MFMA sequence as k_gemm_bf16 / k_gemm_bf16_pp: bit-identical results.
#pragma once
#include "dp_gemm_bf16_pp.h"

namespace dp {

constexpr int W4_THREADS = 256;
constexpr int W4_HALF = 16384;
constexpr int W4_LDS_TILES = 8 * W4_HALF;
constexpr int W4_STAGE_WAVE = 4096;                              // epilogue transposer per wave: 16 rows x 256 B
constexpr int W4_BIAS_OFF = W4_LDS_TILES + 4 * W4_STAGE_WAVE;
constexpr int W4_MAX_N = 4096;                                   // bias vector kept in LDS
constexpr int W4_LDS_TOTAL = W4_BIAS_OFF + W4_MAX_N * 4;         // 160 KiB
constexpr int W4_NST = 32;                                       // global stores one wave's epilogue issues

#ifdef W4_DIAG_NOBAR                                             /* timing-only diagnostics (tools/build_diag.sh): wrong results */
#define W4_BARRIER_ASM ""
#else
#define W4_BARRIER_ASM "\n\ts_barrier"
#endif
#ifdef W4_DIAG_NOREAD
#define W4_DIAG_RD 0
#else
#define W4_DIAG_RD 1
#endif
#ifdef W4_DIAG_NODMA
#define W4_DIAG_ST 0
#else
#define W4_DIAG_ST 1
#endif

template <bool ACT>
__global__ __launch_bounds__(W4_THREADS) void k_gemm_bf16_w4(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                              int ldo, int K, int N, int nwg, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x;
    const int nk = K / 64;                                       // host guarantees nk >= 4 and even
    const int nbn = N / 256;
    const unsigned lds0 = lds_u32(smem);
    // in-situ clock sample (k_gemm_bf16_pp has the same hook): workgroup 0's lifetime, only when the host asked for it
    if (clk != nullptr && blockIdx.x == 0 && tid == 0) {
        clk[0] = __builtin_amdgcn_s_memtime();
        clk[2] = __builtin_amdgcn_s_memrealtime();
    }

    // ---- staging: a wave moves pieces wave, wave+4, wave+8, wave+12 of a half-tile (piece = 8 rows x 128 B) ----
    //   piece p = half-local rows 8p..: wave row/col p/8, rows h*64 + (p%8)*8.. of its 128.  Source address = the half's
    //   first byte (SGPR pair) + a 32-bit lane offset that carries the piece's row offset (vq[q]).
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const unsigned voff = (unsigned)(((wave * 8 + srow) * K + schunk * 8) * 2);
    const unsigned vq[4] = {voff, voff + 64u * K, voff + 256u * K, voff + 320u * K};     // +0, +32, +128, +160 rows
    const size_t hstep = (size_t)128 * K;                        // bytes: half 1 = +64 rows
    const unsigned piece0 = lds0 + wave * 1024;                  // + slot * 16 KiB + q * 4 KiB
    auto stage_half = [&](const char* src_kt, int slot, int h) { // compiler-visible form (start of the stream only)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned vo = vq[q];
            asm volatile("" : "+v"(vo));
            glds16(src_kt + h * hstep + vo, smem + wave * 1024 + slot * W4_HALF + q * 4096);
        }
    };
    // stage cursor: K-tile `kt` of output tile `tile` (wave-uniform)
    struct Cur { const char* a; const char* w; int kt, tile; };
    const int tile0 = xcd_remap(blockIdx.x, G);                  // host guarantees G <= nwg
    auto cur_tile = [&](Cur& c, int tile) {
        // past the end: the cursor parks on K-tile 0 of the workgroup's first tile (kt = nk keeps advance() from moving
        // it): the seven half-tiles still to be issued re-load the same 2 x 16 KiB instead of walking 112 KiB further
        const bool dead = tile >= nwg;
        if (dead) tile = tile0;
        c.tile = dead ? nwg : tile; c.kt = dead ? nk : 0;
        const int m0 = __builtin_amdgcn_readfirstlane((tile / nbn) * 256), n0 = __builtin_amdgcn_readfirstlane((tile % nbn) * 256);
        c.a = (const char*)A + (size_t)m0 * K * 2;
        c.w = (const char*)W + (size_t)n0 * K * 2;
    };
    auto advance = [&](Cur& c) {
        if (c.kt + 1 < nk) { ++c.kt; c.a += 128; c.w += 128; }
        else if (c.kt < nk) cur_tile(c, c.tile + G);             // kt == nk: parked past the end
    };

    // ---- fragment reads: address registers per (buffer, half, k-step); + i * 2048 as the instruction's offset ----
    const unsigned foff = (unsigned)((lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4));
    unsigned a_ad[2][2][2], b_ad[2][2][2];                       // [par][h][ks]
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                a_ad[par][h][ks] = lds0 + ((par * 2 + 0) * 2 + h) * W4_HALF + ((wm * 8192 + foff) ^ (ks * 64));
                b_ad[par][h][ks] = lds0 + ((par * 2 + 1) * 2 + h) * W4_HALF + ((wn * 8192 + foff) ^ (ks * 64));
            }
#define W4_RD1(dst, ad, i)                                                                                      \
    do {                                                                                                        \
        if ((i) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(ad));                                \
        else if ((i) == 1) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(dst) : "v"(ad));               \
        else if ((i) == 2) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(dst) : "v"(ad));               \
        else asm volatile("ds_read_b128 %0, %1 offset:6144" : "=v"(dst) : "v"(ad));                             \
    } while (0)
    // read r (0..7) of a fragment set: k-step r / 4, tile r % 4
#define W4_READ1(set, ad, r) W4_RD1(set[(r) >> 2][(r) & 3], ad[(r) >> 2], (r) & 3)
    // MFMA n (0..31) of quadrant (ha, hb): k-step n / 16, m-tile (n / 4) % 4, n-tile n % 4; ZERO: first touch of the tile
#define W4_MFMA1(ZERO, ha, hb, as, bs, n)                                                                       \
    do {                                                                                                        \
        if ((ZERO) && (n) < 16)                                                                                 \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0"                                               \
                         : "=a"(acc[ha][((n) >> 2) & 3][hb][(n) & 3])                                           \
                         : "v"(bs[(n) >> 4][(n) & 3]), "v"(as[(n) >> 4][((n) >> 2) & 3]));                      \
        else                                                                                                    \
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                              \
                         : "+a"(acc[ha][((n) >> 2) & 3][hb][(n) & 3])                                           \
                         : "v"(bs[(n) >> 4][(n) & 3]), "v"(as[(n) >> 4][((n) >> 2) & 3]));                      \
    } while (0)
    // M0 = LDS address of piece q (written one MFMA ahead of the load that uses it: the wait state the hardware wants)
#define W4_M0(sl, q)                                                                                            \
    do {                                                                                                        \
        if ((q) == 0) asm volatile("s_mov_b32 m0, %0" ::"s"(sl));                                                \
        else if ((q) == 1) asm volatile("s_add_u32 m0, %0, 0x1000" ::"s"(sl) : "scc");                    \
        else if ((q) == 2) asm volatile("s_add_u32 m0, %0, 0x2000" ::"s"(sl) : "scc");                    \
        else asm volatile("s_add_u32 m0, %0, 0x3000" ::"s"(sl) : "scc");                                  \
    } while (0)
#define W4_DMA(sb, q) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(vq[q]), "s"(sb) : "memory")
#define W4_END(VMW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" W4_BARRIER_ASM ::"n"(VMW) : "memory")
    // one phase: 32 MFMAs of quadrant (ha, hb); read r of (RSET <- RAD) behind MFMA 2r+1; DMA piece q of the half-tile
    // at SSRC into slot SLOT behind MFMA 17 + 2q (its M0 write behind MFMA 16 + 2q)
#define W4_PHASE(ZERO, ha, hb, as, bs, RSET, RAD, SSRC, SLOT, VMW)                                              \
    do {                                                                                                        \
        const char* const sb_ = (SSRC);                                                                         \
        const unsigned sl_ = piece0 + (SLOT) * W4_HALF;                                                         \
        _Pragma("unroll") for (int n = 0; n < 32; ++n) {                                                        \
            W4_MFMA1(ZERO, ha, hb, as, bs, n);                                                                  \
            if (W4_DIAG_RD && (n & 1) && n < 16) W4_READ1(RSET, RAD, n >> 1);                                   \
            if (W4_DIAG_ST && !(n & 1) && n >= 16 && n < 24) W4_M0(sl_, (n - 16) >> 1);                         \
            if (W4_DIAG_ST && (n & 1) && n > 16 && n < 24) W4_DMA(sb_, (n - 17) >> 1);                          \
        }                                                                                                       \
        W4_END(VMW);                                                                                            \
    } while (0)
    // One K-tile of stream parity PAR (B0 in BX, B1 in BY on entry; B0 of the next K-tile in BY on exit).
    // slot = (par * 2 + operand) * 2 + half; ZERO: the tile's first K-tile; V0..V3: the vmcnt that ends each phase
#define W4_TILE(ZERO, PAR, BX, BY, V0, V1, V2, V3)                                                              \
    do {                                                                                                        \
        W4_PHASE(ZERO, 0, 0, a0, BX, BY, b_ad[PAR][1], c2.w, ((PAR) * 2 + 1) * 2 + 0, V0);                      \
        W4_PHASE(ZERO, 0, 1, a0, BY, a1, a_ad[PAR][1], c2.w + hstep, ((PAR) * 2 + 1) * 2 + 1, V1);              \
        W4_PHASE(ZERO, 1, 1, a1, BY, a0, a_ad[1 - (PAR)][0], c2.a + hstep, ((PAR) * 2 + 0) * 2 + 1, V2);        \
        W4_PHASE(ZERO, 1, 0, a1, BX, BY, b_ad[1 - (PAR)][0], c3.a, ((1 - (PAR)) * 2 + 0) * 2 + 0, V3);          \
        c2 = c3;                                                                                                \
        advance(c3);                                                                                            \
    } while (0)

    f32x4 acc[2][4][2][4];                                       // [A half][m-tile][B half][n-tile]
    bf16x8 a0[2][4], a1[2][4], bx[2][4], by[2][4];               // [ks][tile]

    // ---- start of the stream: seven half-tiles in the order of their first read ----
    int tile = tile0;
    Cur c2, c3;
    cur_tile(c2, tile);                                          // K-tile 0
    c3 = c2;
    advance(c3);                                                 // K-tile 1
    stage_half(c2.a, 0, 0); stage_half(c2.w, 2, 0); stage_half(c2.w, 3, 1); stage_half(c2.a, 1, 1);
    stage_half(c3.a, 4, 0); stage_half(c3.w, 6, 0); stage_half(c3.w, 7, 1);
    c2 = c3;                                                     // K-tile 1
    advance(c3);                                                 // K-tile 2
    // bias -> LDS, behind the first seven half-tiles: the compiler's wait for these loads (and its drain of the DMA before
    // a visible LDS store) is the wait for the first half-tiles the stream needs anyway -- one memory latency, not two
    for (int c = tid * 4; c < N; c += W4_THREADS * 4)
        *(float4*)(smem + W4_BIAS_OFF + c * 4) = *(const float4*)(bias + c);
    DP_VMCNT(0);
    __syncthreads();
    // pseudo-phases p2(-1), p3(-1)
#pragma unroll
    for (int r = 0; r < 8; ++r) W4_READ1(a0, a_ad[0][0], r);
    stage_half(c2.a, 5, 1);                                      // A1(1)
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int r = 0; r < 8; ++r) W4_READ1(bx, b_ad[0][0], r);
    stage_half(c3.a, 0, 0);                                      // A0(2)
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    c2 = c3;                                                     // K-tile 2
    advance(c3);                                                 // K-tile 3

    bool first = true;
#ifdef DP_PP_STAMP
    unsigned long long stamp[6];
    int round = 0;
#endif
    for (;;) {
        PP_STAMP(0);
#ifdef DP_PP_STAMP
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[5]) :: "memory");
        stamp[1] = stamp[0];
#endif
        const int m0 = __builtin_amdgcn_readfirstlane((tile / nbn) * 256), n0 = __builtin_amdgcn_readfirstlane((tile % nbn) * 256);
        if (first) {
            W4_TILE(true, 0, bx, by, 24, 24, 24, 24);
            W4_TILE(false, 1, by, bx, 24, 24, 24, 24);
        } else {                                                 // the last epilogue's stores are still in the queue
            W4_TILE(true, 0, bx, by, 24 + W4_NST, 24 + W4_NST, 24 + W4_NST, 24 + W4_NST);
            W4_TILE(false, 1, by, bx, 24 + W4_NST, 24 + W4_NST, 24, 24);
        }
#pragma clang loop unroll(disable)
        for (int kt = 2; kt < nk; kt += 2) {
            W4_TILE(false, 0, bx, by, 24, 24, 24, 24);
            W4_TILE(false, 1, by, bx, 24, 24, 24, 24);
        }
        PP_STAMP(2);

#ifndef W4_DIAG_NOEPI
        // ---- epilogue: lane holds out[m = lane&15][n = 4*(lane>>4) + 0..3] of each 16x16 tile ----
        // lane id recomputed (mbcnt) and the accumulators read where they are used (explicit v_accvgpr_read): nothing
        // of the epilogue stays live in VGPRs across the K loop, and no spill reload (a scratch load: vmcnt(0)) drains the DMA
        int elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(elane));
        const int wr = elane & 15, wq = elane >> 4;               // write side: row of the 16-row tile, 4-column group
        const int lr = elane >> 4, rc = elane & 15;               // read side: row within a 4-row group, 16-B chunk
        const int om = m0 + wm * 128, on = n0 + wn * 128;
        float4 bv[2][4];
        {
            const unsigned bad = lds0 + W4_BIAS_OFF + (on + wq * 4) * 4;
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\t"
                         "ds_read_b128 %3, %8 offset:192\n\tds_read_b128 %4, %8 offset:256\n\tds_read_b128 %5, %8 offset:320\n\t"
                         "ds_read_b128 %6, %8 offset:384\n\tds_read_b128 %7, %8 offset:448\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(bv[0][0]), "=&v"(bv[0][1]), "=&v"(bv[0][2]), "=&v"(bv[0][3]), "=&v"(bv[1][0]), "=&v"(bv[1][1]),
                           "=&v"(bv[1][2]), "=&v"(bv[1][3])
                         : "v"(bad) : "memory");
        }
        // transposer: 16 rows x 256 B per step, 16-B chunk c of row r at c ^ r
        const unsigned stg = lds0 + W4_LDS_TILES + wave * W4_STAGE_WAVE;
        const unsigned wa = stg + wr * 256 + ((((wq >> 1) ^ wr)) << 4) + (wq & 1) * 8;
        const unsigned ra = stg + lr * 256 + ((rc ^ lr) << 4);
        bf16_t* const orow = out + (size_t)(om + lr) * ldo + on + rc * 8;
        // step s = 16 rows (A half s / 4, m-tile s % 4) x 128 columns.  Software pipeline: the transposer reads of step s
        // are in flight while step s + 1 is computed into registers; its ds_writes follow the wait that ends the reads.
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        uint2 pk[8];
        u32x4_t r0, r1, r2, r3;
#define W4_EPI_COMPUTE(s)                                                                                       \
    _Pragma("unroll") for (int hb = 0; hb < 2; ++hb) _Pragma("unroll") for (int j = 0; j < 4; ++j) {            \
        f32x4 c;                                                                                                \
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\t"                             \
                     "v_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"                                   \
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])                                       \
                     : "a"(acc[(s) >> 2][(s) & 3][hb][j][0]), "a"(acc[(s) >> 2][(s) & 3][hb][j][1]),            \
                       "a"(acc[(s) >> 2][(s) & 3][hb][j][2]), "a"(acc[(s) >> 2][(s) & 3][hb][j][3]));           \
        const f32x2_t s01 = (f32x2_t){c[0], c[1]} + (f32x2_t){bv[hb][j].x, bv[hb][j].y};  /* packed adds: same bits */ \
        const f32x2_t s23 = (f32x2_t){c[2], c[3]} + (f32x2_t){bv[hb][j].z, bv[hb][j].w};                         \
        float v0 = s01.x, v1 = s01.y, v2 = s23.x, v3 = s23.y;                                                   \
        if (ACT) { silu_fast2(v0, v1); silu_fast2(v2, v3); }                                                    \
        pk[hb * 4 + j] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));                                  \
    }
#define W4_EPI_WRITE()                                                                                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) asm volatile("ds_write_b64 %0, %1" ::"v"(wa ^ (unsigned)(t << 5)), "v"(pk[t]) : "memory")
        W4_EPI_COMPUTE(0);
        W4_EPI_WRITE();
#pragma unroll
        for (int s = 0; s < 8; ++s) {                            // rows 4 q + lr, q = 0..3: chunk ^ (4 q + lr)
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5 offset:1024\n\tds_read_b128 %2, %6 offset:2048\n\t"
                         "ds_read_b128 %3, %7 offset:3072"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                         : "v"(ra), "v"(ra ^ 64u), "v"(ra ^ 128u), "v"(ra ^ 192u) : "memory");
            if (s < 7) { W4_EPI_COMPUTE(s + 1); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory");
            bf16_t* const o = orow + (size_t)(s * 16) * ldo;
            *(u32x4_t*)(o) = r0;
            *(u32x4_t*)(o + (size_t)4 * ldo) = r1;
            *(u32x4_t*)(o + (size_t)8 * ldo) = r2;
            *(u32x4_t*)(o + (size_t)12 * ldo) = r3;
            if (s < 7) { W4_EPI_WRITE(); }
            __builtin_amdgcn_sched_barrier(0);                   // keep the accumulator reads of later steps down here (VGPRs)
        }
#undef W4_EPI_WRITE
#undef W4_EPI_COMPUTE
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef DP_PP_STAMP
        PP_STAMP(3);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[4]) :: "memory");
        if (g_pp_stamp_buf && round < 16 && (tid & 127) == 0) {                     // waves 0 and 2 (one per wave row)
            unsigned long long* d = g_pp_stamp_buf + (((size_t)blockIdx.x * 16 + round) * 2 + (tid >> 7)) * 8;
            for (int k = 0; k < 6; ++k) d[k] = stamp[k];
        }
        ++round;
#endif
        first = false;
        tile += G;
        if (tile >= nwg) break;
    }
    DP_VMCNT(0);                                                 // nothing of this workgroup's DMA may outlive it
    if (clk != nullptr && blockIdx.x == 0 && tid == 0) {
        clk[1] = __builtin_amdgcn_s_memtime();
        clk[3] = __builtin_amdgcn_s_memrealtime();
    }

#undef W4_TILE
#undef W4_PHASE
#undef W4_END
#undef W4_DMA
#undef W4_M0
#undef W4_MFMA1
#undef W4_READ1
#undef W4_RD1
}

}  // namespace dp
