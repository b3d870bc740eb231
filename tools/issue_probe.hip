// Does VALU work of one wave run under the MFMAs of ANOTHER wave on the same SIMD?  (tools/diag/issue_probe, gfx950)
//   mode 0: one wave per SIMD issues MFMAs only            mode 1: one wave per SIMD issues the SiLU VALU mix only
//   mode 2: two waves per SIMD, one of each                mode 3: one wave per SIMD issues both, interleaved
//   mode 4: two waves per SIMD, both MFMA                  mode 5: two waves per SIMD, both VALU
//   mode 8: one wave per SIMD, the weight-stationary kernel's K loop: per MFMA pair one ds_read_b128 into a ring of 8 that the
//           pair 7 ahead consumes, s_waitcnt lgkmcnt(7) before each pair     mode 9: as 8 without the waits (fragments not consumed)
//   mode 10: as 8 with the MFMA wave's partner (second wave of the SIMD) running the VALU mix
//   modes 11, 12: as 8, 10 with the accumulators in AGPRs
//   mode 6: as 2 with the roles swapped (the OLDER wave does the VALU work)   mode 7: as 6, MFMA wave at s_setprio 3
// Prints shader cycles per iteration (s_memtime of wave 0) for each mode.  One workgroup per CU, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(int mode, int iters, unsigned long long* cyc, float* sink) {
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;                                // waves 4..7 share the SIMDs of waves 0..3
    bool do_mfma = false, do_valu = false;
    switch (mode) {
        case 0: do_mfma = !second; break;
        case 1: do_valu = !second; break;
        case 2: do_mfma = !second; do_valu = second; break;
        case 3: do_mfma = do_valu = !second; break;
        case 4: do_mfma = true; break;
        case 5: do_valu = true; break;
        case 6: case 7: do_valu = !second; do_mfma = second; break;
    }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
    float v0 = (float)threadIdx.x * 1e-3f, v1 = 1.0f + v0;
    f32x2 p = {v0, v1};
    if (mode == 7 && do_mfma) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {                        // 8 MFMAs + one value's VALU mix (2 transcendentals, 4 full-rate)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
                if (k == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
                if (k == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(v1));
                if (k == 0 || k == 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
                if (k == 3 || k == 7) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                        // four values' worth per iteration
                asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
                asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(v1));
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p));
                asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = v0 + v1 + p[0] + p[1];
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
}

// the weight-stationary kernel's K loop on one wave per SIMD (waves 0-3); waves 4-7 run the VALU mix when VALU_PARTNER
template <bool WAIT, bool VALU_PARTNER, bool AGPR>
__global__ __launch_bounds__(512) void probe_ws(int iters, unsigned long long* cyc, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) ((uint4*)lds)[i] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b, ring[8];
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 7); b[i] = (__bf16)1.0f; }
    for (int i = 0; i < 8; ++i) ring[i] = a;
    float v0 = (float)threadIdx.x * 1e-3f, v1 = 1.0f + v0;
    f32x2 p2 = {v0, v1};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        for (int it = 0; it < iters; it += 2) {                   // eight pairs (two iterations) per trip: ring indices are constants
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[(p + 7) & 7]) : "v"(ad), "n"(1024 * (p + 1)));
                if (WAIT) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                if (AGPR) {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(2 * p) & 7]) : "v"(WAIT ? ring[p] : a), "v"(b));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(2 * p + 1) & 7]) : "v"(WAIT ? ring[p] : a), "v"(b));
                } else {
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(2 * p) & 7]) : "v"(WAIT ? ring[p] : a), "v"(b));
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(2 * p + 1) & 7]) : "v"(WAIT ? ring[p] : a), "v"(b));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < 8; ++i) v0 += (float)ring[i][0];
    } else if (VALU_PARTNER) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p2));
                asm volatile("v_add_f32 %0, %0, %0" : "+v"(v1));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(v1));
                asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p2));
                asm volatile("v_add_f32 %0, %0, %0" : "+v"(v0));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float s = v0 + v1 + p2[0] + p2[1];
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) sink[0] = s;
}

int main() {
    int dev = 0, cus = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    unsigned long long* cyc;
    float* sink;
    hipMalloc(&cyc, cus * 8 * sizeof(unsigned long long));
    hipMalloc(&sink, 4);
    const int iters = 20000;
    std::vector<unsigned long long> h(cus * 8);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 13; ++mode) {
            if (mode >= 8) {
                hipMemset(cyc, 0, cus * 8 * sizeof(unsigned long long));
                if (mode == 8) hipLaunchKernelGGL((probe_ws<true, false, false>), dim3(cus), dim3(512), 0, 0, iters, cyc, sink);
                if (mode == 9) hipLaunchKernelGGL((probe_ws<false, false, false>), dim3(cus), dim3(512), 0, 0, iters, cyc, sink);
                if (mode == 11) hipLaunchKernelGGL((probe_ws<true, false, true>), dim3(cus), dim3(512), 0, 0, iters, cyc, sink);
                if (mode == 12) hipLaunchKernelGGL((probe_ws<true, true, true>), dim3(cus), dim3(512), 0, 0, iters, cyc, sink);
                if (mode == 10) hipLaunchKernelGGL((probe_ws<true, true, false>), dim3(cus), dim3(512), 0, 0, iters, cyc, sink);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                if (rep) printf("mode %d: wave 0 (first of its SIMD) %.1f cycles / iteration, wave 4 (second) %.1f\n", mode, (double)h[0] / iters, (double)h[4] / iters);
                continue;
            }
            hipMemset(cyc, 0, cus * 8 * sizeof(unsigned long long));
            hipLaunchKernelGGL(probe, dim3(cus), dim3(512), 0, 0, mode, iters, cyc, sink);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            if (rep == 0) continue;
            printf("mode %d: wave 0 (first of its SIMD) %.1f cycles / iteration, wave 4 (second) %.1f\n", mode, (double)h[0] / iters, (double)h[4] / iters);
        }
    printf("modes 8-10: iteration = 8 MFMA in 4 pairs, each behind a ds_read_b128\n");
    printf("iteration = 8 MFMA 16x16x32 (128 pipe cycles) and / or VALU: modes 1, 2, 5 run 4 values' mix per iteration, mode 3 one value's\n");
    return 0;
}
