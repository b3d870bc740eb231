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
//   * a K-tile is four PHASES, one quadrant (16 MFMAs per wave) each; B0 stays in registers for q(A1,B0) and B1 of
//     the NEXT K-tile is read during that phase into the registers it does not use; every half-tile's LDS-DMA
//     (2 x global_load_lds_dwordx4 per wave) is issued 5-6 phases ahead of its first read (table at PP_TILE);
//   * phase = { ds_reads ; stage ; counted s_waitcnt vmcnt ; s_barrier ; lgkmcnt(0) ; 16 MFMA ; s_barrier }.  The four
//     waves of wave row 1 run one barrier behind those of wave row 0, so on every SIMD one wave is in its MFMA
//     segment while its partner reads LDS and issues DMA.
//   * the workgroup is PERSISTENT (one per CU: it owns the whole 160 KiB of LDS) and walks output tiles
//     tile = round * gridDim + xcd_remap(blockIdx); the prologue DMA of the next tile is issued BEFORE the epilogue
//     of the current one, so the epilogue's SiLU, its stores and their drain hide under the next tile's first
//     loads instead of leaving the chip idle while every CU stores at once.
// Hazards (barrier instances b1, b2, ...; row 0 runs [R_P S_P W_P] b(2P+1) [M_P] b(2P+2), row 1 one barrier later):
//   RAW  a half-tile read in phase Q was staged in phase <= Q-5; the counted vmcnt of phase Q-1 (issued after that
//        phase's stage) leaves only newer half-tiles in flight, and every wave has passed that wait before b(2Q),
//        which precedes every read of phase Q.
//   WAR  a slot is re-staged >= 2 phases after its last ds_read: the lagging row's reads of phase P are complete
//        (its lgkmcnt(0)) before it arrives at b(2P+3), and the leading row stages phase P+2 after b(2P+4).
// The LDS image of a half-tile is the one of k_gemm_bf16: 8-row x 128-B pieces, 16-B chunk c of row r at c ^ (r&7)
// (swizzle on the DMA's per-lane SOURCE address and on the ds_read; conflict-free ds_read_b128).
#pragma once
#include "dp_gemm_bf16.h"
#include "dp_step.h"

namespace dp {

#define DP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define DP_VMCNT_N(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define DP_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// 32-bit LDS byte address of a pointer into __shared__ memory (operand of inline-asm ds_* instructions)
__device__ __forceinline__ unsigned lds_u32(const void* p) { return (unsigned)(size_t)(DP_LDS_AS const char*)p; }

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64, PP_THREADS = 512;
constexpr int PP_HALF = 16384;                                  // bytes of one half-tile
constexpr int PP_LDS_BYTES = 8 * PP_HALF;                       // 2 K-tile buffers x {A0, A1, B0, B1}
constexpr int PP_STAGE_WAVE = 4096;                             // epilogue staging per wave (32 rows x 128 B)
constexpr int PP_LDS_TOTAL = PP_LDS_BYTES + 8 * PP_STAGE_WAVE;  // 160 KiB: the whole LDS of a CU

// Diagnostic build only (-DDP_PP_STAMP, tools/pp_stamps.py): s_memtime / s_memrealtime stamps per workgroup and round,
// written to a buffer of their own; no stamp executes in the shipped library.
#ifdef DP_PP_STAMP
__device__ unsigned long long* g_pp_stamp_buf = nullptr;        // [gridDim][16 rounds][2 wave rows][8]
#define PP_STAMP(k)                                                                                             \
    do {                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[k]) :: "memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)
#else
#define PP_STAMP(k) do { } while (0)
#endif

// EPI: what the epilogue does with act(acc + bias)
//   EPI_BF16  store as bf16 (whole 128-B rows through the LDS transposer)
//   EPI_F32   store as float32
//   EPI_MID   N = 256 = the table's width: the value is the raw network output F of the FIRST evaluation of a Heun
//             iteration; apply k_step_mid's arithmetic in place (reads x_hat, writes d_cur and the next evaluation's
//             input u) instead of storing F.  Its epilogue waits on ordinary loads, so the next tile's prologue is
//             issued after it, not before.
//   EPI_F32X3 as EPI_F32, for the M-step's three-product GEMMs (dp_mstep.h): the operands hold, per block of 256 logical
//             columns, [hi (256) | lo (256)]; the K loop runs over THREE 256-wide segments per block and maps them to the
//             stored parts -- A operand (hi, lo, hi), W operand (hi, hi, lo) -- so hi and lo are stored once.
enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_MID = 2, EPI_F32X3 = 3 };

// Batched launches (the M-step's split-K products, dp_mstep_host.h): tile index = slice * tps + tile within the slice; slice s
// reads A + s * a_bs and W + s * w_bs (bytes) with row stride `ld` elements (a slice is a column range of longer rows) and
// writes out + s * o_bs.  Plain launches: ld = K, tps = nwg, strides 0.
struct PpBatch {
    int ld, tps;
    long long a_bs, w_bs, o_bs;
};

template <bool ACT, int EPI>
__global__ __launch_bounds__(PP_THREADS) void k_gemm_bf16_pp(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                              const float* __restrict__ bias, void* __restrict__ out,
                                                              int ldo, int K, int nbn, int nwg, MidStep ms, PpBatch pb) {
    constexpr bool OUT_F32 = EPI == EPI_F32 || EPI == EPI_F32X3;
    constexpr bool X3 = EPI == EPI_F32X3;
    constexpr bool PREFETCH = EPI != EPI_MID;                   // next tile's prologue DMA ahead of this tile's epilogue
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // vector-memory instructions one wave's epilogue issues (they sit between the next tile's prologue DMA and its
    // phase DMA in the in-order vmcnt queue)
    constexpr int NST = OUT_F32 ? 32 : 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int G = gridDim.x;
    // in-situ clock sample (one pair of scalar reads per LAUNCH, only when the host asked for it): workgroup 0's lifetime
    // (written out at once: nothing of it stays live across the K loops)
    if (ms.clk != nullptr && blockIdx.x == 0 && tid == 0) {
        ms.clk[0] = __builtin_amdgcn_s_memtime();
        ms.clk[2] = __builtin_amdgcn_s_memrealtime();
    }
    const int nk = K / PP_BK;                                   // host guarantees nk >= 3

    // ---- staging: a wave moves pieces `wave` and `wave + 8` of every half-tile (a piece = 8 rows x 128 B) ----
    //   A half h, piece p: half-local rows 8p.. = wave row p/8, rows h*64 + (p%8)*8 ..  of that wave row's 128
    //   B half h, piece p: half-local rows 8p.. = wave col p/4, cols h*32 + (p%4)*8 ..  of that wave col's 64
    // Source addresses are a wave-uniform base (SGPRs) + a 32-bit per-lane byte offset, so the eight DMA streams share
    // two address VGPRs.
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    const int ld = pb.ld;                                        // row stride of both operands in elements (= K unless batched)
    const unsigned a_voff = (unsigned)(((wave * 8 + srow) * ld + schunk * 8) * 2);                    // h = 0, piece `wave`
    const unsigned w_voff = (unsigned)((((wave >> 2) * 64 + (wave & 3) * 8 + srow) * ld + schunk * 8) * 2);
    const size_t a_p2 = (size_t)256 * ld, a_h = (size_t)128 * ld;  // bytes: second piece = other wave row (+128 rows); half 1 = +64 rows
    const size_t w_p2 = (size_t)256 * ld, w_h = (size_t)64 * ld;   // bytes: second piece = wave col + 2 (+128 cols); half 1 = +32 cols
    char* const lds_piece = smem + wave * 1024;
    const char *a_base, *w_base;                                 // wave-uniform: first byte of the tile's A rows / W rows

    // byte offset of K-tile kt within a row.  X3: K-tile kt = segment (kt % 12) / 4 of block kt / 12; the segment's part
    // (0 = hi, 1 = lo) is `part_a` / `part_w` of it; a block is 512 stored columns.
    auto koff = [&](int kt, bool is_w) -> size_t {
        if (!X3) return (size_t)kt * (PP_BK * 2);
        const int f = kt / 12, r = kt - f * 12, seg = r >> 2, q = r & 3;
        const int part = is_w ? (seg == 2) : (seg == 1);
        return (size_t)(f * 512 + part * 256 + q * 64) * 2;
    };
    // slot of (buffer b, operand o, half h) = ((b * 2 + o) * 2 + h) * 16 KiB
    auto stage_a = [&](int kt, int h) {
        char* dst = lds_piece + ((kt & 1) * 4 + h) * PP_HALF;
#ifdef PP_DIAG_SAMEK
        kt = kt & 1;                                             // diagnostic: every K-tile re-reads tiles 0/1 (L2 hits)
#endif
#ifdef PP_DIAG_NOSTAGE
        if (kt >= 2) return;                                     // diagnostic: no DMA after the prologue (the waits pass at once)
#endif
        const char* src = a_base + h * a_h + koff(kt, false);
        unsigned vo = a_voff;
        asm volatile("" : "+v"(vo));                             // keep base + offset un-hoisted: selects the saddr + voffset form
        glds16(src + vo, dst);
        glds16(src + a_p2 + vo, dst + 8192);
    };
    auto stage_b = [&](int kt, int h) {
        char* dst = lds_piece + ((kt & 1) * 4 + 2 + h) * PP_HALF;
#ifdef PP_DIAG_SAMEK
        kt = kt & 1;
#endif
#ifdef PP_DIAG_NOSTAGE
        if (kt >= 2) return;
#endif
        const char* src = w_base + h * w_h + koff(kt, true);
        unsigned vo = w_voff;
        asm volatile("" : "+v"(vo));
        glds16(src + vo, dst);
        glds16(src + w_p2 + vo, dst + 8192);
    };
    // prologue DMA of a tile: half-tiles in staging order A0(0) B0(0) B1(0) A1(0) A0(1) B0(1)
    auto prologue = [&](int m0, int n0, int sl) {
        a_base = (const char*)A + (size_t)sl * pb.a_bs + (size_t)m0 * ld * 2;
        w_base = (const char*)W + (size_t)sl * pb.w_bs + (size_t)n0 * ld * 2;
        stage_b(0, 1); stage_a(0, 0); stage_b(0, 0); stage_a(0, 1); stage_b(1, 1); stage_a(1, 0); stage_b(1, 0);
    };

    // ---- fragment reads: row (lane&15) of a 16-row tile, chunk ks*4 + lane/16, swizzled ----
    const int foff = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) << 4);
    // k-step 1 = chunk ^ 4 = byte offset ^ 64: kept as a second base so that tile offsets stay immediates
    const int a_rd[2] = {wm * 8192 + foff, (wm * 8192 + foff) ^ 64};                      // + i * 2048 (m-tile i of the half)
    const int b_rd[2] = {2 * PP_HALF + wn * 4096 + foff, (2 * PP_HALF + wn * 4096 + foff) ^ 64};   // + j * 2048 (n-tile j)

#ifdef PP_DIAG_NOREAD                                            /* timing-only diagnostic: fragments are read during K-tile 0 of a tile and then kept */
#define PP_READ_A(buf, h)                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i) {            \
        if (diag_read) af[ks][i] = *(const bf16x8*)((buf) + (h) * PP_HALF + a_rd[ks] + i * 2048);               \
        else asm volatile("" : "+v"(af[ks][i]));                                                                \
    }
#define PP_READ_B(dst, buf, h)                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j) {            \
        if (diag_read) dst[ks][j] = *(const bf16x8*)((buf) + (h) * PP_HALF + b_rd[ks] + j * 2048);              \
        else asm volatile("" : "+v"(dst[ks][j]));                                                               \
    }
#else
#define PP_READ_A(buf, h)                                                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)              \
        af[ks][i] = *(const bf16x8*)((buf) + (h) * PP_HALF + a_rd[ks] + i * 2048)
#define PP_READ_B(dst, buf, h)                                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j)              \
        dst[ks][j] = *(const bf16x8*)((buf) + (h) * PP_HALF + b_rd[ks] + j * 2048)
#endif
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
#ifdef DP_PP_PHASES                                              /* stamp build: cycles per phase type */
    unsigned long long ph_acc[4] = {0, 0, 0, 0}, ph_last = 0;
#define PP_PHASE_MARK(c)                                                                                        \
    do {                                                                                                        \
        unsigned long long t_;                                                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                            \
        ph_acc[c] += t_ - ph_last;                                                                              \
        ph_last = t_;                                                                                           \
    } while (0)
#else
#define PP_PHASE_MARK(c) do { } while (0)
#endif

    // One K-tile (DMA issue kept out of the phase with the 12 fragment reads; B1 read one phase early into
    // the registers q(A1,B0) does not use):
    //   phase   ds_read_b128                 LDS-DMA (half-tiles)         MFMA quadrant    vmcnt after the stage
    //   c0      B0(t) 4 + A0(t) 8            -                            q(A0,B0)         -
    //   c1      -                            A1(t+1), B1(t+2)             q(A0,B1)         10  (A1(t) landed)
    //   c2      A1(t) 8                      A0(t+2)                      q(A1,B1)         10  (B1(t+1) landed)
    //   c3      B1(t+1) 4                    B0(t+2)                      q(A1,B0)          8  (A0(t+1), B0(t+1) landed)
    // Issue order of the DMA stream: ... A1(t+1) B1(t+2) A0(t+2) B0(t+2) A1(t+2) B1(t+3) ...; every half-tile is read
    // >= 5 phases after its DMA and re-staged >= 2 phases after its last read (A0, B0: +2/+3; A1: +3; B1: +2).
    // ST1: tile kt+1 exists; ST2: tile kt+2 exists.  WAIT(c): the vmcnt after the stage of phase c.
#define PP_TILE(kt, ST1, ST2, WAIT)                                                                             \
    do {                                                                                                        \
        const char* buf = smem + ((kt) & 1) * 4 * PP_HALF;                                                      \
        const char* nbuf = smem + (((kt) + 1) & 1) * 4 * PP_HALF;                                               \
        /* c0 */                                                                                                \
        PP_READ_B(bf0, buf, 0);                                                                                 \
        PP_READ_A(buf, 0);                                                                                      \
        PP_MFMA(0, 0, bf0);                                                                                     \
        PP_PHASE_MARK(0);                                                                                     \
        /* c1 */                                                                                                \
        if (ST1) stage_a((kt) + 1, 1);                                                                          \
        if (ST2) stage_b((kt) + 2, 1);                                                                          \
        WAIT(1);                                                                                                \
        PP_MFMA(0, 1, bf1);                                                                                     \
        PP_PHASE_MARK(1);                                                                                     \
        /* c2 */                                                                                                \
        PP_READ_A(buf, 1);                                                                                      \
        if (ST2) stage_a((kt) + 2, 0);                                                                          \
        WAIT(2);                                                                                                \
        PP_MFMA(1, 1, bf1);                                                                                     \
        PP_PHASE_MARK(2);                                                                                     \
        /* c3 */                                                                                                \
        if (ST1) PP_READ_B(bf1, nbuf, 1);                                                                       \
        if (ST2) stage_b((kt) + 2, 0);                                                                          \
        WAIT(3);                                                                                                \
        PP_MFMA(1, 0, bf0);                                                                                     \
        PP_PHASE_MARK(3);                                                                                     \
    } while (0)
    // During K-tile 0 of a tile whose prologue was issued ahead of the previous epilogue, that epilogue's NST stores sit
    // behind the prologue DMA in the in-order vmcnt queue.
#define PP_WAIT_MAIN(c) do { if ((c) == 3) { DP_VMCNT(8); } else { DP_VMCNT(10); } } while (0)
#define PP_WAIT_T0(c) do { if ((c) == 3) { DP_VMCNT_N(8 + NST); } else { DP_VMCNT_N(10 + NST); } } while (0)
#define PP_WAIT_T2(c) do { if ((c) == 1) { DP_VMCNT(8); } else if ((c) == 2) { DP_VMCNT(6); } else { DP_VMCNT(2); } } while (0)  /* tile nk-2 */
#define PP_WAIT_T1(c) DP_VMCNT(0)                                                                                /* tile nk-1 */

    // tile coordinates live in SGPRs (integer division runs on the VALU: read the result back)
    auto tile_sl = [&](int tile) { return __builtin_amdgcn_readfirstlane(tile / pb.tps); };
    auto tile_m0 = [&](int tile) { return __builtin_amdgcn_readfirstlane(((tile % pb.tps) / nbn) * PP_BM); };
    auto tile_n0 = [&](int tile) { return __builtin_amdgcn_readfirstlane(((tile % pb.tps) % nbn) * PP_BN); };
    int tile = xcd_remap(blockIdx.x, G);                         // host guarantees G <= nwg
    int m0 = tile_m0(tile), n0 = tile_n0(tile), sl = tile_sl(tile);
    prologue(m0, n0, sl);
    bool first = true;

#ifdef DP_PP_STAMP
    unsigned long long stamp[6];
    int round = 0;
#endif
    for (;;) {
        PP_STAMP(0);
#ifdef DP_PP_STAMP
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[5]) :: "memory");   // wall clock (100 MHz) at tile start
#endif
        f32x4 acc[2][4][2][2];                                   // [A half][m-tile][B half][n-tile]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 af[2][4], bf0[2][2], bf1[2][2];                   // [ks][tile]
#ifdef PP_DIAG_NOREAD
        bool diag_read = true;
#endif

        // A0(0), B0(0) have landed (this wave's pieces): all but the 4 newer half-tiles (and the epilogue's stores)
        if (first) { DP_VMCNT(8); } else { DP_VMCNT_N(8 + NST); }
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();               // wave row 1 runs one barrier behind
        PP_READ_B(bf1, smem, 1);                                 // "phase -1": B1(0), a phase of its own so that its slot
        __builtin_amdgcn_s_barrier();                            // may be re-staged in phase 1 like every later B1
        DP_LGKMCNT0();
        __builtin_amdgcn_s_barrier();
        PP_STAMP(1);
#ifdef DP_PP_PHASES
        ph_last = stamp[1];
        ph_acc[0] = ph_acc[1] = ph_acc[2] = ph_acc[3] = 0;
#endif

        if (first) PP_TILE(0, true, true, PP_WAIT_MAIN);
        else PP_TILE(0, true, true, PP_WAIT_T0);
#ifdef PP_DIAG_NOREAD
        diag_read = false;
#endif
        int kt = 1;
#pragma clang loop unroll(disable)
        for (; kt < nk - 2; ++kt) PP_TILE(kt, true, true, PP_WAIT_MAIN);
        PP_TILE(kt, true, false, PP_WAIT_T2);                    // tile nk-2: only the stages of tile nk-1 are left
        ++kt;
        PP_TILE(kt, false, false, PP_WAIT_T1);                   // tile nk-1: nothing left to stage
        if (wm == 0) __builtin_amdgcn_s_barrier();               // balance wave row 1's extra barrier: LDS is quiescent
        PP_STAMP(2);

        // ---- epilogue: lane holds out[m = lane&15][n = 4*(lane>>4) + 0..3] of each 16x16 tile ----
        int elane = lane;                                        // opaque copy: keeps the epilogue's lane-dependent address
        asm volatile("" : "+v"(elane));                          // arithmetic from being hoisted across the K loop (VGPRs)
        const int en = n0 + wn * 64 + (elane >> 4) * 4;
        float4 bv[2][2];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[hb][j] = *(const float4*)(bias + en + hb * 32 + j * 16);
        // materialise the bias NOW (one wait for the four loads): a later compiler-placed wait for them would be a
        // vmcnt(0) that drains the next tile's prologue DMA
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                asm volatile("" : "+v"(bv[hb][j].x), "+v"(bv[hb][j].y), "+v"(bv[hb][j].z), "+v"(bv[hb][j].w));
        const int om = m0 + wm * 128, on = n0 + wn * 64;
        char* const out_sl = (char*)out + (size_t)sl * pb.o_bs;   // this tile's slice of the output

        tile += G;
        const bool more = tile < nwg;
        if (more) {
            m0 = tile_m0(tile);
            n0 = tile_n0(tile);
            sl = tile_sl(tile);
            if (PREFETCH) prologue(m0, n0, sl);
        }
        // The counted waits of the next tile's K-tile 0 (PP_WAIT_T0, the vmcnt(8 + NST) above) assume the in-order vmcnt
        // queue holds [prologue DMA][NST epilogue stores]: pin that order -- nothing of the epilogue may be scheduled above
        // this point, nothing of the prologue below it.
        __builtin_amdgcn_sched_barrier(0);

        if (EPI == EPI_MID) {
            const int64_t row0 = om + (elane & 15);
            const int col0 = on + (elane >> 4) * 4;
            const float inv_t_hat = 1.0f / ms.t_hat;
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int64_t m = row0 + ha * 64 + i * 16;
                    if (m < ms.m_valid) {
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int col = col0 + hb * 32 + j * 16;
                                const f32x4 c = acc[ha][i][hb][j];
                                const float4 xh = *(const float4*)(ms.x_hat + m * 256 + col);
                                float4 d, u;
                                mid_step_cell(xh.x, c[0] + bv[hb][j].x, ms.c_skip, ms.c_out, inv_t_hat, ms.h, ms.c_in_next, d.x, u.x);
                                mid_step_cell(xh.y, c[1] + bv[hb][j].y, ms.c_skip, ms.c_out, inv_t_hat, ms.h, ms.c_in_next, d.y, u.y);
                                mid_step_cell(xh.z, c[2] + bv[hb][j].z, ms.c_skip, ms.c_out, inv_t_hat, ms.h, ms.c_in_next, d.z, u.z);
                                mid_step_cell(xh.w, c[3] + bv[hb][j].w, ms.c_skip, ms.c_out, inv_t_hat, ms.h, ms.c_in_next, d.w, u.w);
                                *(float4*)(ms.d_cur + m * 256 + col) = d;
                                *(uint2*)(ms.u + m * ms.ldu + col) = make_uint2(pack_bf16x2(u.x, u.y), pack_bf16x2(u.z, u.w));
                            }
                    }
                }
            if (more) prologue(m0, n0, sl);
        } else if (OUT_F32) {
            float* const orow = (float*)out_sl + (size_t)(om + (elane & 15)) * ldo + on + (elane >> 4) * 4;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x4 c = acc[ha][i][hb][j];
                            float v0 = c[0] + bv[hb][j].x, v1 = c[1] + bv[hb][j].y, v2 = c[2] + bv[hb][j].z, v3 = c[3] + bv[hb][j].w;
                            if (ACT) { silu_fast2(v0, v1); silu_fast2(v2, v3); }
                            *(float4*)(orow + (size_t)(ha * 64 + i * 16) * ldo + hb * 32 + j * 16) = make_float4(v0, v1, v2, v3);
                        }
        } else {
            // bf16: a lane's 8-byte pieces would make 32-B store segments; transpose 32 rows at a time through this
            // wave's private 4-KiB staging area (128-B rows, 16-B chunk c of row r at c ^ (r&7)) and store whole
            // 128-B rows, 16 B per lane.  The LDS accesses are inline asm: for a compiler-visible ds_write hipcc
            // first drains every LDS-DMA in flight (vmcnt(0)), i.e. the next tile's prologue.  Same-wave LDS
            // operations execute in issue order, so the only wait is the one that ends the read statement.
            const unsigned stg = lds_u32(smem) + PP_LDS_BYTES + wave * PP_STAGE_WAVE;
            const int wr = elane & 15, wq = elane >> 4;           // write side: row within the 16-row tile, 4-column group
            const int rr = elane >> 3, rc = elane & 7;            // read side: row within an 8-row group, 16-B chunk
            // write address of 8-byte unit u = hb*8 + j*4 + wq of row wr: chunk (u>>1) ^ (wr&7), half u&1
            const unsigned wa = stg + wr * 128 + ((((wq >> 1) ^ (wr & 7))) << 4) + (wq & 1) * 8;
            const unsigned ra = stg + rr * 128 + ((rc ^ rr) << 4);                         // rows rr + 8s: + 1024 s
            bf16_t* const orow = (bf16_t*)out_sl + (size_t)(om + rr) * ldo + on + rc * 8;
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const f32x4 c = acc[ha][ip * 2 + ii][hb][j];
                                float v0 = c[0] + bv[hb][j].x, v1 = c[1] + bv[hb][j].y, v2 = c[2] + bv[hb][j].z, v3 = c[3] + bv[hb][j].w;
                                if (ACT) { silu_fast2(v0, v1); silu_fast2(v2, v3); }
                                const uint2 pk = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                                const unsigned addr = wa ^ ((hb * 4 + j * 2) << 4);
                                if (ii == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(pk) : "memory");
                                else asm volatile("ds_write_b64 %0, %1 offset:2048" ::"v"(addr), "v"(pk) : "memory");
                            }
                    uint4 r0, r1, r2, r3;
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(ra) : "memory");
                    bf16_t* const o = orow + (size_t)(ha * 64 + ip * 32) * ldo;
                    *(uint4*)(o) = r0;
                    *(uint4*)(o + (size_t)8 * ldo) = r1;
                    *(uint4*)(o + (size_t)16 * ldo) = r2;
                    *(uint4*)(o + (size_t)24 * ldo) = r3;
                }
        }
        __builtin_amdgcn_sched_barrier(0);                       // ... and no epilogue store below this point (next tile's DMA)
#ifdef DP_PP_STAMP
        PP_STAMP(3);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[4]) :: "memory");
        if (g_pp_stamp_buf && round < 16 && (tid & 255) == 0) {                     // wave 0 and wave 4 (one per wave row)
            unsigned long long* d = g_pp_stamp_buf + (((size_t)blockIdx.x * 16 + round) * 2 + (tid >> 8)) * 8;
            for (int k = 0; k < 5; ++k) d[k] = stamp[k];
#ifndef DP_PP_PHASES
            d[5] = stamp[5];
#endif
#ifdef DP_PP_PHASES
            for (int k = 0; k < 3; ++k) d[5 + k] = ph_acc[k];     // c3 = K loop - (c0 + c1 + c2)
#endif
        }
        ++round;
#endif
        if (!more) break;
        first = !PREFETCH;                                       // without the prefetch no stores sit behind the prologue DMA
    }

    if (ms.clk != nullptr && blockIdx.x == 0 && tid == 0) {
        ms.clk[1] = __builtin_amdgcn_s_memtime();
        ms.clk[3] = __builtin_amdgcn_s_memrealtime();
    }

#undef PP_WAIT_MAIN
#undef PP_WAIT_T0
#undef PP_WAIT_T2
#undef PP_WAIT_T1
#undef PP_TILE
#undef PP_MFMA
#undef PP_READ_A
#undef PP_READ_B
}

}  // namespace dp
