// The M-step's training step as HIP kernels: loss and gradients of the EDM denoising objective for one batch.
//
//   loss = mean_{b,d} w(sigma_b) (D(y_b + sigma_b n_b, sigma_b) - y_b)_d^2        Model.forward, model.py:171-181
//          sigma_b = exp(1.2 z_b - 1.2), w = (sigma^2 + 0.25) / (0.5 sigma)^2     EDMLoss.__call__, diffusion_utils.py:200-215
//          D(x, sigma) = c_skip x + c_out F(c_in x, ln(sigma)/4)                  Precond.forward, model.py:148-165
//          F = MLPDiffusion.forward with a PER-ROW noise label                    model.py:121-128
// and d loss / d theta for the 14 parameter tensors (what loss.backward() leaves in .grad, main.py:124-133).
//
// Why it is not the E-step's GEMM chain: sigma differs per row, so the timestep embedding is a real [B, H] computation
// (positional embedding + the two time_embed layers) instead of a per-level bias row, and every layer needs its two
// backward products.  All of it is GEMM work of three shapes, each an "NT" product of two K-contiguous operands, which is
// what the E-step's bf16 MFMA kernels compute:
//     forward   Y[b, n]  = sum_k X[b, k]  W[n, k]          operands: X, W as stored
//     dX        dX[b, k] = sum_n dY[b, n] Wt[k, n]         operands: dY, W transposed
//     dW        dW[n, k] = sum_b dYt[n, b] Xt[k, b]        operands: dY transposed, X transposed (reduction over the batch)
//
// Arithmetic: the reference trains in float32.  The float32-input MFMA peaks at 157 TFLOP/s and PyTorch's own M-step already
// runs at 107, so a float32 kernel cannot buy much.  Here every float32 operand x is split into two bf16 numbers,
// x = hi + lo + O(2^-17 |x|) (hi = bf16(x), lo = bf16(x - hi)), and a product is evaluated as
//     a w  ~=  a_hi w_hi + a_lo w_hi + a_hi w_lo            (the dropped term a_lo w_lo is O(2^-18 |a w|))
// with float32 accumulation in the bf16 MFMA: three bf16 products for one float32 one, on a unit whose peak is 16 x the
// float32 MFMA's.  Relative error per product ~1e-5 (between float32's 6e-8 and TF32's 5e-4); measured against the
// reference's gradients in tests/test_gpu_mstep.py.  The three products are ONE GEMM over a three times longer K:
//     A' = [A_hi | A_lo | A_hi],  W' = [W_hi | W_hi | W_lo]          (per block of 256 columns)
// Operands are written once, by the kernel that produces them, as [hi | lo] blocks -- both as stored and transposed (for the
// dW products) -- and the E-step's ping-pong GEMM maps its K-tiles to the parts (EPI_F32X3: float32 output, no activation).
// Everything between two GEMMs (bias, SiLU and its derivative, the loss, the split, the transposes, the column sums that
// are the bias gradients) is ONE tile kernel, k_ms_emit<OP>.
#pragma once
#include "dp_util.h"

namespace dp {

// An expanded bf16 operand: logical matrix [rows][cols] (cols = the consuming GEMM's reduction dimension, a multiple of
// 256), stored as [rows][2 * cols] in blocks of 256 columns: block f of a row = [hi (256) | lo (256)] of columns
// 256 f .. 256 f + 255, i.e. element (r, c) has hi at r * 2 cols + (c / 256) * 512 + c % 256 and lo 256 further.
// The GEMM (k_gemm_bf16_pp<false, EPI_F32X3>) walks THREE 256-wide segments per block and maps them to the parts: as its A
// operand (hi, lo, hi), as its W operand (hi, hi, lo).  Any range of whole blocks is itself a valid operand with row stride
// 2 * cols: that is what a split-K slice of a product reads.  p == nullptr: not wanted.
struct MsExp {
    bf16_t* p;
    int rows, cols;
};
// elements (r, c) and (r, c + 1), c even: two 4-byte stores
__device__ __forceinline__ void ms_exp_store2(const MsExp& e, int r, int c, float v0, float v1) {
    const bf16_t h0 = f2bf(v0), h1 = f2bf(v1);
    uint32_t* q = (uint32_t*)(e.p + (size_t)r * 2 * e.cols + (c >> 8) * 512 + (c & 255));
    q[0] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    q[128] = pack_bf16x2(v0 - bf2f(h0), v1 - bf2f(h1));
}

__device__ __forceinline__ float ms_sigmoid(float a) { return 1.0f / (1.0f + expf(-a)); }

// per-row constants of a batch: EDMLoss's sigma and weight, Precond's coefficients (float32, the reference's operation order)
struct MsRow { float c_in, c_skip, c_out, weight, c_noise, sigma, pad0, pad1; };

// rows [0, B) from z_sigma; rows [B, Bp) are padding: sigma = exp(-1.2), weight 0 (they contribute nothing to loss or gradients)
__global__ __launch_bounds__(256) void k_ms_rows(const float* __restrict__ z_sigma, MsRow* __restrict__ rc, int B, int Bp) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bp) return;
    const float z = b < B ? z_sigma[b] : 0.f;
    const float s = expf(z * 1.2f + -1.2f);                     // (rnd_normal * P_std + P_mean).exp(), diffusion_utils.py:202-203
    const float sd = 0.5f, sd2 = sd * sd;
    MsRow r;
    r.sigma = s;
    r.weight = b < B ? (s * s + sd2) / ((s * sd) * (s * sd)) : 0.f;          // diffusion_utils.py:205
    r.c_skip = sd2 / (s * s + sd2);                             // model.py:155-158
    r.c_out = s * sd / sqrtf(s * s + sd2);
    r.c_in = 1.0f / sqrtf(sd2 + s * s);
    r.c_noise = logf(s) / 4.0f;
    r.pad0 = r.pad1 = 0.f;
    rc[b] = r;
}

enum { MS_U = 0, MS_PE, MS_W, MS_FWD_ACT, MS_FWD_LIN, MS_BWD_ACT, MS_BWD_LIN, MS_LOSS };

struct MsEmit {
    // sources
    const float* src;       // GEMM output [S][Mp][ld_src] (partials summed in order s = 0, 1, ...) | MS_U: x [B][D] | MS_W: W [N][K]
    int S;                  // number of partials
    size_t src_stride;      // floats between partials
    int ld_src;
    const float* bias;      // [n_valid] or nullptr
    const float* addend;    // [Mp][ld_add] float32 or nullptr (MS_FWD_LIN: the time embedding added to proj's output)
    int ld_add;
    const float* pre_in;    // MS_BWD_ACT: the layer's pre-activation [Mp][ld_pre]
    const MsRow* rc;        // per-row constants (MS_U, MS_PE, MS_LOSS)
    const float* x;         // MS_LOSS: the clean batch [B][D];  MS_U: z_noise [B][D]
    const float* noisy_in;  // MS_LOSS: y + n [Mp][ld_noisy]
    int ld_noisy;
    // valid extent of the logical matrix (outside it the emitted value is 0)
    int m_valid, n_valid;
    int d_valid;            // MS_LOSS / MS_U: the table's width D (row stride of x)
    float scale;            // MS_LOSS: 2 / (B D)
    // sinks
    float* pre_out;         // MS_FWD_ACT: pre-activation [Mp][ld_pre]
    int ld_pre;
    float* f32_out;         // the emitted value as float32 [Mp][ld_f32] (MS_FWD_LIN's embedding, MS_U's noisy table)
    int ld_f32;
    float* percell;         // MS_LOSS: optional [B][D] per-cell loss
    MsExp outN, outT;       // expanded operand as stored [Mp][Np] / transposed [Np][Mp]
    float* colpart;         // optional [gridDim.y][Np] column sums of the emitted value over each 64-row block
    int ld_col;
    double* losspart;       // MS_LOSS: [gridDim.y * gridDim.x] per-block sums of the per-cell loss
};

// value of one cell for the operations that read unpadded caller tables (odd row strides: scalar accesses)
template <int OP>
__device__ __forceinline__ float ms_cell_in(const MsEmit& e, const MsRow& r, int m, int n, bool valid, float a) {
    if (OP == MS_U) {
        // noisy = y + randn_like(y) * sigma (diffusion_utils.py:207-209); network input c_in * noisy (model.py:160)
        return valid ? e.src[(size_t)m * e.d_valid + n] + e.x[(size_t)m * e.d_valid + n] * r.sigma : 0.f;
    }
    if (OP == MS_W) return valid ? e.src[(size_t)m * e.ld_src + n] : 0.f;
    return a;
}

// One 64 (m) x 128 (n) tile of an emitted matrix per workgroup: blockIdx.x = column block, blockIdx.y = row block.  A thread
// owns two adjacent columns (8-byte loads, 4-byte bf16-pair stores) of 16 rows; the transposed copy goes through LDS and is
// written as pairs of adjacent batch rows.
template <int OP>
__device__ __forceinline__ void ms_emit_tile(const MsEmit& e, const int bx, const int by, const int nbx) {
    __shared__ __attribute__((aligned(8))) float tile[2][64][66];      // [column parity][column pair][row]
    __shared__ float colred[4][128];
    __shared__ double lossred[4];
    const int tid = threadIdx.x, nl = tid & 63, mg = tid >> 6;
    const int m0 = by * 64, n0 = bx * 128;
    const int n = n0 + 2 * nl;
    float cs0 = 0.f, cs1 = 0.f;
    double lsum = 0.0;
    const float b0 = (e.bias && n < e.n_valid) ? e.bias[n] : 0.f;
    const float b1 = (e.bias && n + 1 < e.n_valid) ? e.bias[n + 1] : 0.f;
    float f0 = 0.f, f1 = 0.f;
    bool cos0 = false, cos1 = false;
    if (OP == MS_PE) {                                          // model.py:25-29 then the swap at :123 -> [sin | cos]
        const int half = e.n_valid / 2;
        cos0 = n >= half; cos1 = n + 1 >= half;
        f0 = powf(1.0f / 10000.0f, (float)(cos0 ? n - half : n) / (float)half);
        f1 = powf(1.0f / 10000.0f, (float)(cos1 ? n + 1 - half : n + 1) / (float)half);
    }
    const bool nv0 = n < e.n_valid, nv1 = n + 1 < e.n_valid;
    // Rows in groups of four: every load of a group is issued before the first store of the group (the stores may alias the
    // loads as far as the compiler knows, so a row-at-a-time loop pays one memory latency per row: 37 us for the loss
    // kernel's 128 workgroups).
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        float A0[4], A1[4], Y0[4], Y1[4], X0[4], X1[4];
        float2 P[4], NZ[4], AD[4];
        MsRow R[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + mg + 4 * (4 * g + j);
            const bool mv = m < e.m_valid;
            A0[j] = A1[j] = Y0[j] = Y1[j] = X0[j] = X1[j] = 0.f;
            P[j] = NZ[j] = AD[j] = make_float2(0.f, 0.f);
            R[j].sigma = 0.f; R[j].c_in = 1.f; R[j].c_skip = R[j].c_out = R[j].weight = R[j].c_noise = 0.f;
            if (OP == MS_U || OP == MS_PE || OP == MS_LOSS) R[j] = e.rc[m];
            if (OP == MS_U || OP == MS_W) {
                Y0[j] = ms_cell_in<OP>(e, R[j], m, n, mv && nv0, 0.f);
                Y1[j] = ms_cell_in<OP>(e, R[j], m, n + 1, mv && nv1, 0.f);
            } else if (OP != MS_PE) {
                for (int sl = 0; sl < e.S; ++sl) {
                    const float2 t = *(const float2*)(e.src + sl * e.src_stride + (size_t)m * e.ld_src + n);
                    A0[j] += t.x; A1[j] += t.y;
                }
                if (OP == MS_FWD_LIN && e.addend) AD[j] = *(const float2*)(e.addend + (size_t)m * e.ld_add + n);
                if (OP == MS_BWD_ACT) P[j] = *(const float2*)(e.pre_in + (size_t)m * e.ld_pre + n);
                if (OP == MS_LOSS) {
                    NZ[j] = *(const float2*)(e.noisy_in + (size_t)m * e.ld_noisy + n);
                    if (mv && nv0) X0[j] = e.x[(size_t)m * e.d_valid + n];
                    if (mv && nv1) X1[j] = e.x[(size_t)m * e.d_valid + n + 1];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ml = mg + 4 * (4 * g + j), m = m0 + ml;
            const bool mv = m < e.m_valid;
            float v0 = 0.f, v1 = 0.f;
            if (OP == MS_U) {
                if (e.f32_out) *(float2*)(e.f32_out + (size_t)m * e.ld_f32 + n) = make_float2(Y0[j], Y1[j]);     // the noisy table
                v0 = R[j].c_in * Y0[j]; v1 = R[j].c_in * Y1[j];
            } else if (OP == MS_W) {
                v0 = Y0[j]; v1 = Y1[j];
            } else if (OP == MS_PE) {
                const float cn = R[j].c_noise;
                const float a0 = cn * f0, a1 = cn * f1;
                v0 = nv0 ? (cos0 ? cosf(a0) : sinf(a0)) : 0.f;
                v1 = nv1 ? (cos1 ? cosf(a1) : sinf(a1)) : 0.f;
            } else {
                float a0 = A0[j], a1 = A1[j];
                if (OP == MS_FWD_ACT) {
                    a0 += b0; a1 += b1;
                    *(float2*)(e.pre_out + (size_t)m * e.ld_pre + n) = make_float2(a0, a1);
                    v0 = nv0 ? a0 * ms_sigmoid(a0) : 0.f;                                      // nn.SiLU, model.py:106,108,110,117
                    v1 = nv1 ? a1 * ms_sigmoid(a1) : 0.f;
                } else if (OP == MS_FWD_LIN) {
                    a0 += b0; a1 += b1;
                    if (e.addend) { a0 += AD[j].x; a1 += AD[j].y; }
                    v0 = nv0 ? a0 : 0.f; v1 = nv1 ? a1 : 0.f;
                    if (e.f32_out) *(float2*)(e.f32_out + (size_t)m * e.ld_f32 + n) = make_float2(v0, v1);
                } else if (OP == MS_BWD_ACT) {
                    const float2 p = P[j];
                    const float s0 = ms_sigmoid(p.x), s1 = ms_sigmoid(p.y);
                    v0 = nv0 ? a0 * (s0 + p.x * s0 * (1.0f - s0)) : 0.f;                       // d/dp [p sigmoid(p)]
                    v1 = nv1 ? a1 * (s1 + p.y * s1 * (1.0f - s1)) : 0.f;
                } else if (OP == MS_BWD_LIN) {
                    v0 = nv0 ? a0 : 0.f; v1 = nv1 ? a1 : 0.f;
                } else if (OP == MS_LOSS) {
                    // D = c_skip x + c_out F (model.py:164); loss = w (D - y)^2 (diffusion_utils.py:213); mean over all B x D cells
                    // (model.py:180-181) -> dF = (2 / (B D)) w (D - y) c_out
                    const MsRow r = R[j];
                    const float2 nz = NZ[j];
                    if (mv && nv0) {
                        const float diff = (r.c_skip * nz.x + r.c_out * (a0 + b0)) - X0[j];
                        const float pc = r.weight * (diff * diff);
                        if (e.percell) e.percell[(size_t)m * e.d_valid + n] = pc;
                        lsum += (double)pc;
                        v0 = e.scale * r.weight * diff * r.c_out;
                    }
                    if (mv && nv1) {
                        const float diff = (r.c_skip * nz.y + r.c_out * (a1 + b1)) - X1[j];
                        const float pc = r.weight * (diff * diff);
                        if (e.percell) e.percell[(size_t)m * e.d_valid + n + 1] = pc;
                        lsum += (double)pc;
                        v1 = e.scale * r.weight * diff * r.c_out;
                    }
                }
            }
            cs0 += v0; cs1 += v1;
            if (e.outN.p) ms_exp_store2(e.outN, m, n, v0, v1);
            if (e.outT.p) { tile[0][nl][ml] = v0; tile[1][nl][ml] = v1; }
        }
    }
    if (e.colpart) {
        colred[mg][2 * nl] = cs0;
        colred[mg][2 * nl + 1] = cs1;
        __syncthreads();
        if (mg < 2) {
            const int c = mg * 64 + nl;
            e.colpart[(size_t)by * e.ld_col + n0 + c] = (colred[0][c] + colred[1][c]) + (colred[2][c] + colred[3][c]);
        }
    }
    if (OP == MS_LOSS) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
        if (nl == 0) lossred[mg] = lsum;
        __syncthreads();
        if (tid == 0) e.losspart[by * nbx + bx] = (lossred[0] + lossred[1]) + (lossred[2] + lossred[3]);
    }
    if (e.outT.p) {
        __syncthreads();
        // transposed: row n0 + nn of the transposed matrix; a lane writes the pair of batch rows m0 + 2 l, + 1 (32 lanes = 128 B)
        const int ml2 = tid & 31, ng = tid >> 5;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int nn = ng + 8 * i;
            const float2 t = *(const float2*)&tile[nn & 1][nn >> 1][2 * ml2];
            ms_exp_store2(e.outT, n0 + nn, m0 + 2 * ml2, t.x, t.y);
        }
    }
}

template <int OP>
__global__ __launch_bounds__(256) void k_ms_emit(const MsEmit e) { ms_emit_tile<OP>(e, blockIdx.x, blockIdx.y, gridDim.x); }

// The seven weight tensors' operands (as stored and transposed) in ONE launch: they are re-split every step (the optimiser
// has just changed them) and seven launches of 32 ... 512 tiles each cost ~11 us apiece, mostly ramp-up and tail.
struct MsEmitGroup {
    MsEmit e[7];
    int first[8];           // first[l] = first workgroup of tensor l; first[7] = grid size
    int nbx[7];             // column blocks of tensor l
};
__global__ __launch_bounds__(256) void k_ms_emit_weights(const MsEmitGroup g) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < 7; ++i) l += (int)blockIdx.x >= g.first[i];
    const int t = blockIdx.x - g.first[l];
    ms_emit_tile<MS_W>(g.e[l], t % g.nbx[l], t / g.nbx[l], g.nbx[l]);
}

// grad[n][k] = sum_s part[s][n][k] over the valid extent (the GEMM output is padded to [Np][ld])
__global__ __launch_bounds__(256) void k_ms_reduce_grad(const float* __restrict__ part, int S, size_t stride, int ld, float* __restrict__ grad,
                                                        int N, int K) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / K), c = (int)(i - (int64_t)r * K);
        float a = 0.f;
        for (int s = 0; s < S; ++s) a += part[s * stride + (size_t)r * ld + c];
        grad[i] = a;
    }
}

// out[n] = sum over the row blocks of colpart[block][n] (the bias gradients); out2 optionally receives a copy.  64 columns
// per workgroup, four threads per column (each sums a quarter of the blocks in order), combined in a fixed order.
__global__ __launch_bounds__(256) void k_ms_colsum(const float* __restrict__ colpart, int blocks, int ld, float* __restrict__ out,
                                                   float* __restrict__ out2, int N) {
    __shared__ float red[4][64];
    const int nl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + nl;
    const int per = (blocks + 3) / 4;
    float a = 0.f;
    if (n < N) {
#pragma unroll 8
        for (int b = q * per; b < min(blocks, (q + 1) * per); ++b) a += colpart[(size_t)b * ld + n];
    }
    red[q][nl] = a;
    __syncthreads();
    if (q == 0 && n < N) {
        const float v = (red[0][nl] + red[1][nl]) + (red[2][nl] + red[3][nl]);
        out[n] = v;
        if (out2) out2[n] = v;
    }
}

// The six bias-gradient reductions of a step in one launch (each emit keeps its own column partials until the end).
struct MsColGroup {
    const float* colpart[6];
    float* out[6];
    float* out2[6];
    int N[6], first[7];     // first[j] = first workgroup of reduction j; first[6] = grid size
    int blocks, ld;
};
__global__ __launch_bounds__(256) void k_ms_colsum_group(const MsColGroup g) {
    __shared__ float red[4][64];
    int j = 0;
#pragma unroll
    for (int i = 1; i < 6; ++i) j += (int)blockIdx.x >= g.first[i];
    const int nl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n = (blockIdx.x - g.first[j]) * 64 + nl;
    const int per = (g.blocks + 3) / 4;
    const float* __restrict__ cp = g.colpart[j];
    float a = 0.f;
    if (n < g.N[j]) {
#pragma unroll 8
        for (int b = q * per; b < min(g.blocks, (q + 1) * per); ++b) a += cp[(size_t)b * g.ld + n];
    }
    red[q][nl] = a;
    __syncthreads();
    if (q == 0 && n < g.N[j]) {
        const float v = (red[0][nl] + red[1][nl]) + (red[2][nl] + red[3][nl]);
        g.out[j][n] = v;
        if (g.out2[j]) g.out2[j][n] = v;
    }
}

// loss = (sum of the per-block partials, fixed tree) / (B D)
__global__ __launch_bounds__(256) void k_ms_loss(const double* __restrict__ part, int n, double inv_cells, float* __restrict__ loss) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * inv_cells);
}

}  // namespace dp
