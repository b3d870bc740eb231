// Shared helpers for libdiffputer_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <algorithm>
#include <string>

#include "../../include/diffputer.h"

namespace dp {

typedef uint16_t bf16_t;                                           // bf16 bit pattern
typedef __attribute__((ext_vector_type(8))) short bf16x8;          // MFMA A/B fragment, 16x16x32 / 32x32x16 bf16
typedef __attribute__((ext_vector_type(4))) float f32x4;           // 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(16))) float f32x16;         // 32x32 accumulator fragment

extern thread_local std::string g_err;

inline int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define DP_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return dp::fail(DP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),      \
                            __FILE__, __LINE__);                                                  \
    } while (0)

#define DP_LAUNCH_CHECK(name)                                                                     \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess)                                                                    \
            return dp::fail(DP_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e__));    \
    } while (0)

template <typename T> __host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }
template <typename T> __host__ __device__ inline T round_up(T a, T b) { return ceil_div(a, b) * b; }

// fp32 -> bf16, round to nearest even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs.
__device__ __forceinline__ bf16_t f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// SiLU for the GEMM epilogues: x * sigmoid(x) = x / (1 + e^-x)   (model.py:106,108,110)
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Two at a time: the scale, the +1 and the final product as packed-f32 instructions (same operations, same bits as
// silu_fast; the transcendentals stay one per element).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void silu_fast2(float& a, float& b) {
    const f32x2_t x = {a, b};
    const f32x2_t t = x * (f32x2_t){-1.44269504088896340736f, -1.44269504088896340736f};
    const f32x2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    const f32x2_t d = e + (f32x2_t){1.0f, 1.0f};
    const f32x2_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2_t y = x * r;
    a = y.x;
    b = y.y;
}
__device__ __forceinline__ float silu_precise(float x) { return x / (1.0f + expf(-x)); }

}  // namespace dp
