// Host side of the native M-step training step (kernels and arithmetic: dp_mstep.h).  Included by diffputer_hip.hip after
// gemm(): every product below goes through the E-step's GEMM dispatcher (bf16 operands, float32 output, no activation).
#pragma once
#include "dp_mstep.h"

namespace dp {

// Every logical dimension is padded to 256 (a GEMM's M must be a multiple of 256 and each of B, H, 2H, D appears as some
// product's M); padding is zero-filled and contributes nothing.
struct MsDims {
    int D, H, B;
    int Dp, Hp, H2p, Bp;
};
static MsDims ms_dims(int d_in, int hid, int64_t batch) {
    MsDims d;
    d.D = d_in; d.H = hid; d.B = (int)batch;
    d.Dp = round_up(d_in, 256); d.Hp = round_up(hid, 256); d.H2p = round_up(2 * hid, 256); d.Bp = round_up((int)batch, 256);
    return d;
}

// layers in the order of the C ABI's w[] / b[]: 0 proj, 1 mlp.0, 2 mlp.2, 3 mlp.4, 4 mlp.6, 5 time_embed.0, 6 time_embed.2
struct MsLayer { int N, K, Np, Kp; };
static void ms_layers(const MsDims& d, MsLayer L[7]) {
    L[0] = {d.H, d.D, d.Hp, d.Dp};
    L[1] = {2 * d.H, d.H, d.H2p, d.Hp};
    L[2] = {2 * d.H, 2 * d.H, d.H2p, d.H2p};
    L[3] = {d.H, 2 * d.H, d.Hp, d.H2p};
    L[4] = {d.D, d.H, d.Dp, d.Hp};
    L[5] = {d.H, d.H, d.Hp, d.Hp};
    L[6] = {d.H, d.H, d.Hp, d.Hp};
}

struct MsWork {
    MsRow* rc;
    float *noisy, *emb, *gout, *pre[4], *colpart[6], *zero;     // pre: at0, a1, a2, a3; colpart: one per bias-gradient reduction
    double* losspart;
    bf16_t *uN, *uT, *peN, *peT, *t1N, *t1T, *h0N, *h0T, *h1N, *h1T, *h2N, *h2T, *h3N, *h3T;
    bf16_t *dFN, *dFT, *da3N, *da3T, *da2N, *da2T, *da1N, *da1T, *dh0N, *dh0T, *dat0T;
    bf16_t *WN[7], *WT[7];
    size_t bytes;
};
static MsWork ms_carve(const MsDims& d, void* base) {
    MsWork w;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += round_up<size_t>(n, 256); return base ? (void*)((char*)base + o) : (void*)nullptr; };
    auto expd = [&](int rows, int cols) { return (bf16_t*)take((size_t)rows * cols * 2 * sizeof(bf16_t)); };
    const int maxf = std::max(d.H2p, d.Dp);
    w.rc = (MsRow*)take((size_t)d.Bp * sizeof(MsRow));
    w.noisy = (float*)take((size_t)d.Bp * d.Dp * 4);
    w.emb = (float*)take((size_t)d.Bp * d.Hp * 4);
    // the current product's float32 partial sums: at most 16 slices, and never more than ~256 tiles' worth in all
    w.gout = (float*)take(std::max((size_t)std::max(d.Bp, maxf) * maxf, (size_t)512 * 65536) * 4);
    w.pre[0] = (float*)take((size_t)d.Bp * d.Hp * 4);
    w.pre[1] = (float*)take((size_t)d.Bp * d.H2p * 4);
    w.pre[2] = (float*)take((size_t)d.Bp * d.H2p * 4);
    w.pre[3] = (float*)take((size_t)d.Bp * d.Hp * 4);
    for (int j = 0; j < 6; ++j) w.colpart[j] = (float*)take((size_t)(d.Bp / 64) * maxf * 4);
    w.zero = (float*)take((size_t)maxf * 4);
    w.losspart = (double*)take((size_t)(d.Bp / 64) * (d.Dp / 128) * 8);
    w.uN = expd(d.Bp, d.Dp); w.uT = expd(d.Dp, d.Bp);
    w.peN = expd(d.Bp, d.Hp); w.peT = expd(d.Hp, d.Bp);
    w.t1N = expd(d.Bp, d.Hp); w.t1T = expd(d.Hp, d.Bp);
    w.h0N = expd(d.Bp, d.Hp); w.h0T = expd(d.Hp, d.Bp);
    w.h1N = expd(d.Bp, d.H2p); w.h1T = expd(d.H2p, d.Bp);
    w.h2N = expd(d.Bp, d.H2p); w.h2T = expd(d.H2p, d.Bp);
    w.h3N = expd(d.Bp, d.Hp); w.h3T = expd(d.Hp, d.Bp);
    w.dFN = expd(d.Bp, d.Dp); w.dFT = expd(d.Dp, d.Bp);
    w.da3N = expd(d.Bp, d.Hp); w.da3T = expd(d.Hp, d.Bp);
    w.da2N = expd(d.Bp, d.H2p); w.da2T = expd(d.H2p, d.Bp);
    w.da1N = expd(d.Bp, d.H2p); w.da1T = expd(d.H2p, d.Bp);
    w.dh0N = expd(d.Bp, d.Hp); w.dh0T = expd(d.Hp, d.Bp);
    w.dat0T = expd(d.Hp, d.Bp);
    MsLayer L[7];
    ms_layers(d, L);
    for (int l = 0; l < 7; ++l) { w.WN[l] = expd(L[l].Np, L[l].Kp); w.WT[l] = expd(L[l].Kp, L[l].Np); }
    w.bytes = off;
    return w;
}

template <int OP>
static int ms_emit(const MsEmit& e, int Mp, int Np, hipStream_t st) {
    hipLaunchKernelGGL(k_ms_emit<OP>, dim3(Np / 128, Mp / 64), dim3(256), 0, st, e);
    DP_LAUNCH_CHECK("k_ms_emit");
    return DP_OK;
}

// Split factor of a product with `tiles` 256 x 256 output tiles over a reduction of K columns: enough slices to give every
// CU a tile (the batch is only 4 096 rows: a [4096, 1024] output is 64 tiles for 256 CUs), each slice a whole number of
// 256-column blocks.  Partial sums cost memory traffic, so no more slices than that.
#ifndef MS_SPLIT_THRESH
#define MS_SPLIT_THRESH 192
#endif
static int ms_split(int tiles, int K) {
    int S = 1;
    while (tiles * S < MS_SPLIT_THRESH && K % (512 * S) == 0 && S < 16) S *= 2;
    return S;
}

// gout[s][M][N] (float32) = partial products of A[M][K] W[N][K]^T (each as three bf16 products) over S column slices: ONE
// batched launch of the ping-pong kernel.  Returns S through *S_out; the consumer sums the partials in slice order.
static int ms_gemm(const bf16_t* A, const bf16_t* W, float* out, const float* zero_bias, int M, int N, int K, hipStream_t st, int* S_out) {
    const int S = ms_split((M / 256) * (N / 256), K);
    *S_out = S;
    const int Ks = K / S;                                        // logical columns per slice: 3 Ks of K loop over 2 Ks stored columns
    return launch_bf16_pp(A, W, zero_bias, out, N, M, N, 3 * Ks, 0, 1, st, nullptr, nullptr, S, 2 * K, (long long)Ks * 4, (long long)Ks * 4,
                          (long long)M * N * 4, true);
}

static int mstep_fwd_bwd(int d_in, int hid, int64_t batch, const float* x, const float* z_sigma, const float* z_noise,
                         const float* const w[7], const float* const b[7], float* const gw[7], float* const gb[7], float* loss,
                         float* percell, void* workspace, size_t workspace_bytes, hipStream_t st) {
    const MsDims d = ms_dims(d_in, hid, batch);
    MsLayer L[7];
    ms_layers(d, L);
    MsWork k = ms_carve(d, workspace);
    if (k.bytes > workspace_bytes) return fail(DP_EWORKSPACE, "M-step workspace of %zu bytes is too small (need %zu)", workspace_bytes, k.bytes);
    int rc;
    const int maxf = std::max(d.H2p, d.Dp);
    DP_HIP(hipMemsetAsync(k.zero, 0, (size_t)maxf * 4, st));
    hipLaunchKernelGGL(k_ms_rows, dim3(ceil_div(d.Bp, 256)), dim3(256), 0, st, z_sigma, k.rc, d.B, d.Bp);
    DP_LAUNCH_CHECK("k_ms_rows");

    auto blank = [&]() { MsEmit e; memset(&e, 0, sizeof e); e.S = 1; return e; };
    // ---- operands that do not depend on the network: weights (as stored: forward; transposed: dX), input, embedding ----
    {
        MsEmitGroup g;
        int nb = 0;
        for (int l = 0; l < 7; ++l) {
            MsEmit e = blank();
            e.src = w[l]; e.ld_src = L[l].K; e.m_valid = L[l].N; e.n_valid = L[l].K;
            e.outN = {k.WN[l], L[l].Np, L[l].Kp};
            if (l != 0 && l != 5) e.outT = {k.WT[l], L[l].Kp, L[l].Np};       // proj and time_embed.0 have no input gradient
            g.e[l] = e;
            g.first[l] = nb;
            g.nbx[l] = L[l].Kp / 128;
            nb += (L[l].Kp / 128) * (L[l].Np / 64);
        }
        g.first[7] = nb;
        hipLaunchKernelGGL(k_ms_emit_weights, dim3(nb), dim3(256), 0, st, g);
        DP_LAUNCH_CHECK("k_ms_emit_weights");
    }
    {
        MsEmit e = blank();
        e.src = x; e.x = z_noise; e.rc = k.rc; e.m_valid = d.B; e.n_valid = d.D; e.d_valid = d.D;
        e.f32_out = k.noisy; e.ld_f32 = d.Dp;
        e.outN = {k.uN, d.Bp, d.Dp}; e.outT = {k.uT, d.Dp, d.Bp};
        if ((rc = ms_emit<MS_U>(e, d.Bp, d.Dp, st))) return rc;
    }
    {
        MsEmit e = blank();
        e.rc = k.rc; e.m_valid = d.Bp; e.n_valid = d.H;
        e.outN = {k.peN, d.Bp, d.Hp}; e.outT = {k.peT, d.Hp, d.Bp};
        if ((rc = ms_emit<MS_PE>(e, d.Bp, d.Hp, st))) return rc;
    }
    // forward layer: gout = A W^T, then bias (+ addend), SiLU, split
    auto fwd = [&](int l, const bf16_t* A, bool act, float* pre, const float* addend, float* f32_out, bf16_t* outN, bf16_t* outT) -> int {
        int S = 1;
        int r = ms_gemm(A, k.WN[l], k.gout, k.zero, d.Bp, L[l].Np, L[l].Kp, st, &S);
        if (r) return r;
        MsEmit e = blank();
        e.S = S; e.src_stride = (size_t)d.Bp * L[l].Np;
        e.src = k.gout; e.ld_src = L[l].Np; e.bias = b[l]; e.m_valid = d.Bp; e.n_valid = L[l].N;
        e.addend = addend; e.ld_add = L[l].Np; e.pre_out = pre; e.ld_pre = L[l].Np; e.f32_out = f32_out; e.ld_f32 = L[l].Np;
        if (outN) e.outN = {outN, d.Bp, L[l].Np};
        if (outT) e.outT = {outT, L[l].Np, d.Bp};
        return act ? ms_emit<MS_FWD_ACT>(e, d.Bp, L[l].Np, st) : ms_emit<MS_FWD_LIN>(e, d.Bp, L[l].Np, st);
    };
    if ((rc = fwd(5, k.peN, true, k.pre[0], nullptr, nullptr, k.t1N, k.t1T))) return rc;          // time_embed.0 + SiLU
    if ((rc = fwd(6, k.t1N, false, nullptr, nullptr, k.emb, nullptr, nullptr))) return rc;        // time_embed.2 -> emb
    if ((rc = fwd(0, k.uN, false, nullptr, k.emb, nullptr, k.h0N, k.h0T))) return rc;             // proj(x) + emb
    if ((rc = fwd(1, k.h0N, true, k.pre[1], nullptr, nullptr, k.h1N, k.h1T))) return rc;          // mlp.0 + SiLU
    if ((rc = fwd(2, k.h1N, true, k.pre[2], nullptr, nullptr, k.h2N, k.h2T))) return rc;          // mlp.2 + SiLU
    if ((rc = fwd(3, k.h2N, true, k.pre[3], nullptr, nullptr, k.h3N, k.h3T))) return rc;          // mlp.4 + SiLU
    // mlp.6, preconditioning, loss and its gradient with respect to F
    const int nlb = (d.Bp / 64) * (d.Dp / 128);
    {
        int S = 1;
        if ((rc = ms_gemm(k.h3N, k.WN[4], k.gout, k.zero, d.Bp, d.Dp, d.Hp, st, &S))) return rc;
        MsEmit e = blank();
        e.S = S; e.src_stride = (size_t)d.Bp * d.Dp;
        e.src = k.gout; e.ld_src = d.Dp; e.bias = b[4]; e.rc = k.rc; e.x = x; e.noisy_in = k.noisy; e.ld_noisy = d.Dp;
        e.m_valid = d.B; e.n_valid = d.D; e.d_valid = d.D; e.scale = (float)(2.0 / ((double)d.B * d.D));
        e.percell = percell; e.losspart = k.losspart; e.colpart = k.colpart[0]; e.ld_col = maxf;
        e.outN = {k.dFN, d.Bp, d.Dp}; e.outT = {k.dFT, d.Dp, d.Bp};
        if ((rc = ms_emit<MS_LOSS>(e, d.Bp, d.Dp, st))) return rc;
        hipLaunchKernelGGL(k_ms_loss, dim3(1), dim3(256), 0, st, (const double*)k.losspart, nlb, 1.0 / ((double)d.B * d.D), loss);
        DP_LAUNCH_CHECK("k_ms_loss");
    }
    // bias gradients = column sums of the dY each emit leaves as per-64-row partials; reduced together at the end
    MsColGroup cg;
    memset(&cg, 0, sizeof cg);
    cg.blocks = d.Bp / 64; cg.ld = maxf;
    int ncol = 0, colblocks = 0;
    auto bias_grad = [&](int l, float* second) {                 // the reduction of the most recently written partials
        cg.colpart[ncol] = k.colpart[ncol]; cg.out[ncol] = gb[l]; cg.out2[ncol] = second; cg.N[ncol] = L[l].N;
        cg.first[ncol] = colblocks;
        colblocks += ceil_div(L[l].N, 64);
        ++ncol;
    };
    // weight gradient of layer l: dW[n][k] = sum_b dYt[n][b] Xt[k][b].  (Tried: these products on a second stream, so that they
    // fill the ramp-up and tail of the input-gradient chain -- no gain: a persistent GEMM holds every CU's LDS until its tiles
    // are done, so the chain's next kernel waits for it just the same; tools/mstep_bench.py 1.487 vs 1.498 ms.  Also tried: only the
    // small reduction below on a side stream, beside the next product, with a partials buffer of its own: 1.34 -> 1.42 ms -- the
    // fork / join events cost more than the 12 us kernel they hide.)
    auto wgrad = [&](int l, const bf16_t* dYT, const bf16_t* XT) -> int {
        int S = 1;
        int r = ms_gemm(dYT, XT, k.gout, k.zero, L[l].Np, L[l].Kp, d.Bp, st, &S);
        if (r) return r;
        const int64_t total = (int64_t)L[l].N * L[l].K;
        hipLaunchKernelGGL(k_ms_reduce_grad, dim3((unsigned)std::min<int64_t>(ceil_div<int64_t>(total, 256), 4096)), dim3(256), 0, st,
                           (const float*)k.gout, S, (size_t)L[l].Np * L[l].Kp, L[l].Kp, gw[l], L[l].N, L[l].K);
        DP_LAUNCH_CHECK("k_ms_reduce_grad");
        return DP_OK;
    };
    // input gradient through layer l, then through the SiLU in front of it (pre != nullptr) : gout = dY W, emit
    auto xgrad = [&](int l, const bf16_t* dYN, const float* pre, bf16_t* outN, bf16_t* outT) -> int {
        int S = 1;
        int r = ms_gemm(dYN, k.WT[l], k.gout, k.zero, d.Bp, L[l].Kp, L[l].Np, st, &S);
        if (r) return r;
        MsEmit e = blank();
        e.S = S; e.src_stride = (size_t)d.Bp * L[l].Kp;
        e.src = k.gout; e.ld_src = L[l].Kp; e.m_valid = d.Bp; e.n_valid = L[l].K; e.pre_in = pre; e.ld_pre = L[l].Kp;
        e.colpart = k.colpart[ncol]; e.ld_col = maxf;
        if (outN) e.outN = {outN, d.Bp, L[l].Kp};
        if (outT) e.outT = {outT, L[l].Kp, d.Bp};
        return pre ? ms_emit<MS_BWD_ACT>(e, d.Bp, L[l].Kp, st) : ms_emit<MS_BWD_LIN>(e, d.Bp, L[l].Kp, st);
    };
    // ---- backward (what loss.backward() computes, main.py:131) ----
    bias_grad(4, nullptr);                                                   // mlp.6.bias: column sums of dF
    if ((rc = wgrad(4, k.dFT, k.h3T))) return rc;
    if ((rc = xgrad(4, k.dFN, k.pre[3], k.da3N, k.da3T))) return rc;                               // d a3
    bias_grad(3, nullptr);
    if ((rc = wgrad(3, k.da3T, k.h2T))) return rc;
    if ((rc = xgrad(3, k.da3N, k.pre[2], k.da2N, k.da2T))) return rc;                              // d a2
    bias_grad(2, nullptr);
    if ((rc = wgrad(2, k.da2T, k.h1T))) return rc;
    if ((rc = xgrad(2, k.da2N, k.pre[1], k.da1N, k.da1T))) return rc;                              // d a1
    bias_grad(1, nullptr);
    if ((rc = wgrad(1, k.da1T, k.h0T))) return rc;
    if ((rc = xgrad(1, k.da1N, nullptr, k.dh0N, k.dh0T))) return rc;                               // d h0 = d emb
    bias_grad(0, gb[6]);                                                    // proj.bias and time_embed.2.bias
    if ((rc = wgrad(0, k.dh0T, k.uT))) return rc;
    if ((rc = wgrad(6, k.dh0T, k.t1T))) return rc;
    if ((rc = xgrad(6, k.dh0N, k.pre[0], nullptr, k.dat0T))) return rc;                            // d at0
    bias_grad(5, nullptr);
    if ((rc = wgrad(5, k.dat0T, k.peT))) return rc;
    for (int j = ncol; j <= 6; ++j) cg.first[j] = colblocks;
    hipLaunchKernelGGL(k_ms_colsum_group, dim3(colblocks), dim3(256), 0, st, cg);
    DP_LAUNCH_CHECK("k_ms_colsum_group");
    return DP_OK;
}

}  // namespace dp
