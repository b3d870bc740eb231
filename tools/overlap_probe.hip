// Which epilogue instructions of one wave make progress under the MFMA stream of the OTHER wave of its SIMD?
// (tools/diag/overlap_probe, gfx950; follows tools/issue_probe.hip, whose two-wave modes timed each wave to the end of its
// own loop -- after the faster wave has left, the slower one runs alone, which hid how little it got while both ran.)
// Waves 0-3 (one per SIMD) issue v_mfma_f32_16x16x32_bf16 back to back; waves 4-7 (their SIMD partners) run one of:
//   kind 0  the shipped SiLU mix per value: v_pk_mul, v_exp_f32, v_add, v_rcp_f32, v_pk_mul, v_add
//   kind 1  six full-rate VALU instructions (v_pk_mul / v_add), no transcendental
//   kind 2  v_exp_f32 + v_rcp_f32 only
//   kind 3  four v_accvgpr_read_b32
//   kind 4  LDS round trip: ds_write_b64 x2, ds_read_b128, s_waitcnt lgkmcnt(0)
//   kind 5  the no-transcendental SiLU of tools/silu_probe.hip variant 3 (magic-number exp2 + Newton reciprocal), packed, per 2 values
//   kind 6  v_pk_fma_f32 x6
// For each kind three launches: the partner alone (A), the partner with a FIXED amount of work while the MFMA waves run ten
// times longer (B: time of that work under MFMAs), and the MFMA waves with a fixed amount of work while the partner runs
// ten times longer (C: MFMA time under the partner).  Prints cycles per partner iteration alone / under MFMAs, and cycles per
// 8 MFMAs alone / under the partner.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void partner_iter(float& v0, float& v1, f32x2& p, f32x2& q, f32x4& ag, unsigned lds_ad) {
    if (KIND == 0) {
        asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(v1));
        asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(q));
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
    } else if (KIND == 1) {
        asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1));
        asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(q));
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
        asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1));
    } else if (KIND == 2) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(v1));
    } else if (KIND == 3) {
        float a, b, c, d;
        asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                     : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "a"(ag[0]), "a"(ag[1]), "a"(ag[2]), "a"(ag[3]));
        asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d));
    } else if (KIND == 4) {
        f32x4 r;
        asm volatile("ds_write_b64 %1, %2\n\tds_write_b64 %1, %3 offset:8\n\tds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r) : "v"(lds_ad), "v"(p), "v"(q) : "memory");
        asm volatile("" ::"v"(r));
    } else if (KIND == 5) {
        // two values: t = clamp(x * -log2e); n = round(t); f = t - n; 2^f cubic; exponent add; d = e + 1; r0 = magic - d; 3 Newton steps; y = x r
        const f32x2 x = p;
        f32x2 t = x * (f32x2){-1.44269504f, -1.44269504f};
        t.x = __builtin_amdgcn_fmed3f(t.x, -126.0f, 126.0f);
        t.y = __builtin_amdgcn_fmed3f(t.y, -126.0f, 126.0f);
        const f32x2 magic = {12582912.0f, 12582912.0f};
        const f32x2 nn = t + magic;
        const f32x2 f = t - (nn - magic);
        f32x2 e = f * (f32x2){0.0555041f, 0.0555041f} + (f32x2){0.2402265f, 0.2402265f};
        e = e * f + (f32x2){0.6931472f, 0.6931472f};
        e = e * f + (f32x2){1.0f, 1.0f};
        e.x = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e.x) + (__builtin_bit_cast(unsigned, nn.x) << 23));
        e.y = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, e.y) + (__builtin_bit_cast(unsigned, nn.y) << 23));
        const f32x2 d = e + (f32x2){1.0f, 1.0f};
        f32x2 r;
        r.x = __builtin_bit_cast(float, 0x7EF311C7u - __builtin_bit_cast(unsigned, d.x));
        r.y = __builtin_bit_cast(float, 0x7EF311C7u - __builtin_bit_cast(unsigned, d.y));
        const f32x2 two = {2.0f, 2.0f};
        r = r * (two - d * r);
        r = r * (two - d * r);
        r = r * (two - d * r);
        p = x * r + (f32x2){1e-3f, 1e-3f};
        asm volatile("" : "+v"(p));
    } else if (KIND == 8) {
        // as kind 7 with the accumulator reads in the gap, ahead of the VALU burst: [16 accread][40 VALU][16 exp][16 rcp]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a, b, c, d;
            asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "a"(ag[0]), "a"(ag[1]), "a"(ag[2]), "a"(ag[3]));
            asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d));
        }
        f32x2 w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = p + q * (float)k;
#pragma unroll
        for (int rep = 0; rep < 5; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[k]) : "v"(q));
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = w[k & 7][k >> 3];
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(t[k]));
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(t[k]));
        float acc2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc2 += t[k];
        asm volatile("" : "+v"(acc2));
        p.x = acc2 * 1e-30f + 1.0f; p.y = 1.0f;
    } else if (KIND == 7) {
        // one slot of a stage-major epilogue: a burst of 40 independent-enough VALU instructions, then 32 transcendentals
        // and 16 accumulator reads (called 4x per iteration by the harness: the harness' "4 units" = 4 slots)
        f32x2 w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = p + q * (float)k;
#pragma unroll
        for (int rep = 0; rep < 5; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[k]) : "v"(q));
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = w[k & 7][k >> 3];
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(t[k]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a, b, c, d;
            asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "a"(ag[0]), "a"(ag[1]), "a"(ag[2]), "a"(ag[3]));
            asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d));
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(t[k]));
        float acc2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc2 += t[k];
        asm volatile("" : "+v"(acc2));
        p.x = acc2 * 1e-30f + 1.0f; p.y = 1.0f;
    } else if (KIND == 6) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p) : "v"(q));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q) : "v"(p));
        }
    }
}

template <int KIND, int MV>
__global__ __launch_bounds__(512) void probe(int mfma_iters, int partner_iters, unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[8192];
    const int wave = threadIdx.x >> 6;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 big[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
    f32x4 ag = {1.f, 2.f, 3.f, 4.f};
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[0]) : "v"(1.0f));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[1]) : "v"(1.0f));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[2]) : "v"(1.0f));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[3]) : "v"(1.0f));
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
    float v0 = (float)threadIdx.x * 1e-3f, v1 = 1.0f + v0;
    f32x2 p = {v0, v1}, q = {v1, v0};
    const unsigned lds_ad = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + threadIdx.x * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (MV == 0 || MV == 10) {                                // 16x16x32, A / B from VGPRs (the kernels' form)
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
            }
        } else if (MV == 1) {                                     // 32x32x16: the same FLOP in four instructions of 8 passes
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(big[k]) : "v"(a), "v"(b));
            }
        } else if (MV == 2) {                                     // 16x16x32, A / B from AGPRs
            f32x4 aa, ab;
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(aa[i]) : "v"(1.0f));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ab[i]) : "v"(1.0f));
            }
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "a"(aa), "a"(ab));
            }
        } else if (MV >= 20 && MV < 30) {                         // 16 MFMAs, then a gap of (MV - 20) x 16 idle cycles (s_nop 15)
            for (int it = 0; it < mfma_iters; it += 2) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k & 7]) : "v"(a), "v"(b));
#pragma unroll
                for (int g = 0; g < MV - 20; ++g) asm volatile("s_nop 15");
            }
        } else if (MV >= 30 && MV < 40) {                         // fine gaps: s_nop (MV - 30) behind every PAIR of MFMAs
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a), "v"(b));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k + 1]) : "v"(a), "v"(b));
                    asm volatile("s_nop %0" ::"n"(MV - 30));
                }
            }
        } else if (MV >= 40 && MV < 56) {                         // s_nop (MV - 40) behind every FOURTH MFMA
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; k += 4) {
#pragma unroll
                    for (int q2 = 0; q2 < 4; ++q2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k + q2]) : "v"(a), "v"(b));
                    asm volatile("s_nop %0" ::"n"(MV - 40));
                }
            }
        } else if (MV >= 60 && MV < 70) {                         // 32 MFMAs, then (MV - 60) x s_nop 15
            for (int it = 0; it < mfma_iters; it += 4) {
#pragma unroll
                for (int k = 0; k < 32; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k & 7]) : "v"(a), "v"(b));
#pragma unroll
                for (int g = 0; g < MV - 60; ++g) asm volatile("s_nop 15");
            }
        } else if (MV == 3) {                                     // 16x16x32 with one idle issue slot behind every MFMA
            for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 3" : "+a"(acc[k]) : "v"(a), "v"(b));
            }
        }
    } else {
        if (MV >= 10) __builtin_amdgcn_s_setprio(3);              // MFMA stream 10: as 0, the PARTNER at raised priority
        for (int it = 0; it < partner_iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) partner_iter<KIND>(v0, v1, p, q, ag, lds_ad);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = v0 + v1 + p[0] + p[1] + q[0] + q[1];
    for (int i = 0; i < 8; ++i) { float x; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i][0])); s += x; }
    if (MV == 1) for (int i = 0; i < 4; ++i) { float x; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(big[i][0])); s += x; }
    if (s == 12345.678f) sink[0] = s;
}

// ONE wave per SIMD: per MFMA pair, behind the MFMAs, a slice of epilogue work of the SAME wave (a software-pipelined kernel).
//   SL 0 nothing   1 two transcendentals   2 two v_accvgpr_read   3 two full-rate VALU   4 one value's whole slice: 1 accread,
//   1 exp, 1 rcp, 2.5 VALU (alternating 2 / 3)   5 as 4 plus one ds_read_b128 into a ring (the W fragment) with lgkmcnt(7)
template <int SL>
__global__ __launch_bounds__(256) void probe_same(int iters, unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[16384];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ag;
    for (int i = 0; i < 4; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[i]) : "v"(1.0f));
    bf16x8 a, b, ring[8];
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
    for (int i = 0; i < 8; ++i) ring[i] = a;
    for (int i = threadIdx.x; i < 16384 / 16; i += 256) ((uint4*)lds)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16;
    float v0 = (float)threadIdx.x * 1e-3f, v1 = 1.0f + v0, x0, x1;
    f32x2 p = {v0, v1}, q = {v1, v0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {                             // 8 pairs per iteration
            if (SL == 5) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(k + 7) & 7]) : "v"(ad), "n"(1024 * (k + 1)));
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
            }
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(SL == 5 ? ring[k] : a), "v"(b));
            if (SL == 1 || SL >= 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
            if (SL == 2 || SL >= 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(ag[k & 3]));
            if (SL == 3 || SL >= 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(k + 4) & 7]) : "v"(SL == 5 ? ring[k] : a), "v"(b));
            if (SL == 1 || SL >= 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(v1));
            if (SL == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(ag[(k + 1) & 3]));
            if (SL == 3 || SL >= 4) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(q));
            if (SL >= 4 && (k & 1)) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(q));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    float sum = v0 + v1 + p[0] + p[1] + q[0] + q[1];
    if (SL == 2 || SL >= 4) sum += x0;
    if (SL == 2) sum += x1;
    for (int i = 0; i < 8; ++i) { float x; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[i][0])); sum += x; }
    if (SL == 5) for (int i = 0; i < 8; ++i) sum += (float)ring[i][0];
    if (sum == 12345.678f) sink[0] = sum;
}

template <int SL>
static void run_same(int cus, unsigned long long* cyc, float* sink, const char* what) {
    std::vector<unsigned long long> h(cus * 8);
    const int N = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, cus * 8 * sizeof(unsigned long long));
        hipLaunchKernelGGL(probe_same<SL>, dim3(cus), dim3(256), 0, 0, N, cyc, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    }
    hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    printf("same wave, slice %d (%s): %.1f cycles per MFMA pair\n", SL, what, (double)h[0] / N / 8);
}

template <int KIND, int MV = 0>
static void run(int cus, unsigned long long* cyc, float* sink, const char* what) {
    std::vector<unsigned long long> h(cus * 8);
    const int N = 4000;
    double res[4] = {0, 0, 0, 0};
    for (int cfg = 0; cfg < 4; ++cfg) {
        // cfg 0: partner alone   1: partner (N) under MFMAs (long)   2: MFMAs alone   3: MFMAs (N) under the partner (long)
        const int mi = cfg == 0 ? 0 : cfg == 1 ? 40 * N : N;
        const int pi = cfg == 0 ? N : cfg == 1 ? N : cfg == 2 ? 0 : 40 * N;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(cyc, 0, cus * 8 * sizeof(unsigned long long));
            hipLaunchKernelGGL((probe<KIND, MV>), dim3(cus), dim3(512), 0, 0, mi, pi, cyc, sink);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
        }
        hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        res[cfg] = cfg < 2 ? (double)h[4] / N : (double)h[0] / N;
    }
    printf("MFMA stream %d, kind %d (%s): partner iteration (4 units) %.1f cycles alone, %.1f under MFMAs (%.0f %% of its speed); 8 MFMAs %.1f cycles alone, %.1f under the partner\n",
           MV, KIND, what, res[0], res[1], 100.0 * res[0] / res[1], res[2], res[3]);
}

int main() {
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    unsigned long long* cyc;
    float* sink;
    (void)hipMalloc(&cyc, cus * 8 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 4);
    run_same<0>(cus, cyc, sink, "MFMAs only");
    run_same<1>(cus, cyc, sink, "+ exp, rcp");
    run_same<2>(cus, cyc, sink, "+ 2 v_accvgpr_read");
    run_same<3>(cus, cyc, sink, "+ 2 full-rate VALU");
    run_same<4>(cus, cyc, sink, "+ one value's epilogue: accread, exp, rcp, 2.5 VALU");
    run_same<5>(cus, cyc, sink, "as 4 + W fragment ring");
    run<0>(cus, cyc, sink, "SiLU mix: pk_mul, exp, add, rcp, pk_mul, add");
    run<1>(cus, cyc, sink, "six full-rate VALU");
    run<2>(cus, cyc, sink, "exp + rcp");
    run<3>(cus, cyc, sink, "four v_accvgpr_read");
    run<4>(cus, cyc, sink, "LDS round trip");
    run<5>(cus, cyc, sink, "SiLU of two values without transcendentals");
    run<6>(cus, cyc, sink, "six v_pk_fma_f32");
    printf("MFMA stream 1 = v_mfma_f32_32x32x16_bf16 (4 per iteration), 2 = 16x16x32 with A and B in AGPRs, 3 = 16x16x32 + s_nop 3\n");
    run<1, 1>(cus, cyc, sink, "six full-rate VALU");
    run<0, 1>(cus, cyc, sink, "SiLU mix");
    run<1, 2>(cus, cyc, sink, "six full-rate VALU");
    run<0, 2>(cus, cyc, sink, "SiLU mix");
    run<1, 3>(cus, cyc, sink, "six full-rate VALU");
    run<0, 3>(cus, cyc, sink, "SiLU mix");
    printf("MFMA streams 22 / 24 / 26: 16 MFMAs then 32 / 64 / 96 idle cycles\n");
    run<1, 22>(cus, cyc, sink, "six full-rate VALU");
    run<1, 24>(cus, cyc, sink, "six full-rate VALU");
    run<1, 26>(cus, cyc, sink, "six full-rate VALU");
    run<0, 24>(cus, cyc, sink, "SiLU mix");
    run<6, 24>(cus, cyc, sink, "six v_pk_fma_f32");
    printf("MFMA streams 3x: s_nop x behind every pair; 4x/5x: s_nop (MV - 40) behind every fourth MFMA\n");
    run<1, 31>(cus, cyc, sink, "six full-rate VALU");
    run<1, 33>(cus, cyc, sink, "six full-rate VALU");
    run<1, 35>(cus, cyc, sink, "six full-rate VALU");
    run<1, 37>(cus, cyc, sink, "six full-rate VALU");
    run<1, 43>(cus, cyc, sink, "six full-rate VALU");
    run<1, 47>(cus, cyc, sink, "six full-rate VALU");
    run<1, 51>(cus, cyc, sink, "six full-rate VALU");
    run<1, 55>(cus, cyc, sink, "six full-rate VALU");
    run<0, 47>(cus, cyc, sink, "SiLU mix");
    printf("stage-major epilogue slots (kind 7: 40 VALU, then 32 transcendentals + 16 accumulator reads; 4 slots per iteration) under 32 MFMAs + g x s_nop 15 (streams 60 + g; 8 MFMAs = a quarter of a segment)\n");
    run<7, 0>(cus, cyc, sink, "stage-major slot");
    run<7, 62>(cus, cyc, sink, "stage-major slot");
    run<7, 63>(cus, cyc, sink, "stage-major slot");
    run<7, 64>(cus, cyc, sink, "stage-major slot");
    run<7, 65>(cus, cyc, sink, "stage-major slot");
    run<7, 66>(cus, cyc, sink, "stage-major slot");
    run<8, 64>(cus, cyc, sink, "slot, accumulator reads first");
    run<8, 65>(cus, cyc, sink, "slot, accumulator reads first");
    run<8, 66>(cus, cyc, sink, "slot, accumulator reads first");
    run<8, 67>(cus, cyc, sink, "slot, accumulator reads first");
    printf("MFMA stream 10 = stream 0 with the partner at s_setprio 3\n");
    run<1, 10>(cus, cyc, sink, "six full-rate VALU");
    run<0, 10>(cus, cyc, sink, "SiLU mix");
    run<2, 10>(cus, cyc, sink, "exp + rcp");
    run<6, 10>(cus, cyc, sink, "six v_pk_fma_f32");
    return 0;
}
