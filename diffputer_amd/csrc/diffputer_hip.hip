// libdiffputer_hip.so -- plan object, host-side sampler loop and the C ABI of include/diffputer.h.
// gfx950 (MI355X) only.  Build: see __graft_entry__.build().
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "dp_embed.h"
#include "dp_gemm_bf16.h"
#include "dp_gemm_bf16_pp.h"
#include "dp_gemm_bf16_ws.h"
#include "dp_gemm_bf16_ws2.h"
#include "dp_gemm_bf16_w4.h"
#include "dp_gemm_f32.h"
#include "dp_philox.h"
#include "dp_prep.h"
#include "dp_step.h"
#include "dp_util.h"

namespace dp {
thread_local std::string g_err;

// ---- host-side float32 scalar algebra, in the reference's operation order -------------------------------------
// (host code targets baseline x86-64, which has no fused multiply-add: nothing here can be contracted)
struct Coef { float c_skip, c_out, c_in, c_noise; };
static Coef precond_coef(float s) {                       // model.py:155-158 with sigma_data = 0.5 (model.py:137)
    const float sd2 = 0.25f;
    Coef c;
    c.c_skip = sd2 / (s * s + sd2);
    c.c_out = s * 0.5f / sqrtf(s * s + sd2);
    c.c_in = 1.0f / sqrtf(sd2 + s * s);
    c.c_noise = logf(s) / 4.0f;
    return c;
}
static float churn_gamma(int num_steps) {                 // diffusion_utils.py:94 with S_churn=1, S_min=0, S_max=inf
    const double g = std::min(1.0 / num_steps, sqrt(2.0) - 1.0);
    return (float)g;
}

struct Level { float sigma; Coef c; };

}  // namespace dp

using namespace dp;

struct dp_plan {
    int d_in = 0, hid = 0, S = 0, R = 0, dtype = 0, device = 0;
    int K1 = 0, N5 = 0;                  // padded widths of GEMM-1's K and GEMM-5's N
    int n_levels = 0;                    // 2S-1
    bool weights_set = false;
    // host tables
    std::vector<float> t, th, churn, renoise;
    std::vector<Level> lv;               // [2i] = t_hat_i, [2i+1] = t_{i+1}
    // device
    void* W[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};     // packed weights (bf16 or f32)
    float* B[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};    // biases (B[4] padded to N5; B[0] = proj.bias)
    float *te_w0 = nullptr, *te_b0 = nullptr, *te_w2 = nullptr, *te_b2 = nullptr;
    // proj folded into mlp.0 (no activation separates them, model.py:126-128):
    //   mlp.0(proj(u) + emb) = (W0 Wp) u + (W0 b_eff + b0)
    void* W01 = nullptr;                 // packed [2 hid, K1] = W_mlp.0 W_proj
    float* w0_f32 = nullptr;             // [2 hid, hid] float32 copy of mlp.0.weight (for the scratch levels' bias rows)
    float* b01 = nullptr;                // [(n_levels + 2), 2 hid] = W0 b_eff[e] + b_mlp.0
    float *b_eff = nullptr;              // [(n_levels + 2), hid]; the last two rows are scratch for dp_denoise/dp_sample_step
    float *c_noise_dev = nullptr, *pe = nullptr, *hidden = nullptr;   // [(n_levels + 2)] / [.., hid] scratch
    // optional in-situ kernel timing (dp_plan_profile): event pairs around sampled evaluations
    int prof_stride = 0;                 // 0 = off; else every prof_stride-th evaluation is sampled
    long prof_evals = 0;                 // evaluations seen since profiling was enabled
    int prof_used = 0;                   // samples recorded
    std::vector<hipEvent_t> prof_ev;     // DP_PROF_SAMPLES * 7 events: before GEMM-1, after GEMM-1..5, after the step kernel
    std::vector<int> prof_tag;           // which step kernel followed: 0 mid, 1 end, 2 final
    unsigned long long* prof_clk = nullptr;   // device [DP_PROF_SAMPLES][4]: in-kernel clock sample of the mlp.2 launch of each profiled evaluation
    void* static_ws = nullptr;           // this device's copy of the file-scope scratch array (dp_impute_static), resolved at create
    size_t esize() const { return dtype == DP_DTYPE_BF16 ? 2 : 4; }
};
#define DP_PROF_SAMPLES 256

namespace dp {

static int grid_for(int64_t n, int block = 256, int cap = 2048) {
    int64_t g = ceil_div<int64_t>(n, block);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

// tuning knobs (dp_tune): "gemm_variant" 3 = four-wave kernel for the deep layers, ping-pong kernel for the other wide ones
// (default), 2 = ping-pong kernel for all wide layers, 1 = k_gemm_bf16 everywhere;
// "fold_proj" 1 = evaluate proj and mlp.0 as one Linear (default), 0 = as two GEMMs like the reference's module list.
// Process-wide and read once per evaluation; relaxed atomics, so a dp_tune on one thread never tears a read on another
// (include/diffputer.h, "Threading").
static std::atomic<int> g_gemm_variant{3};
static std::atomic<int> g_fold_proj{1};
static std::atomic<int> g_m_chunk{32768};  // "m_chunk": rows per launch of the wide layers in bf16 plans (0 = whole batch), see eval_F
static std::atomic<int> g_ws_gemm{1};      // "ws_gemm": weight-stationary kernel for shallow layers with N % 256 == 0 (bf16 plans):
                                           // 1 = where it wins, K <= 128 (default); 2 = wherever it runs, K <= 256; 0 = never
static std::atomic<int> g_w4_min_k{1024};  // "w4_min_k": smallest K the four-wave kernel takes under gemm_variant 3 (A/B knob; >= 256)
static std::atomic<int> g_narrow_bm{128};   // "narrow_bm": rows per tile of the simple kernel on layers with N % 256 != 0 (mlp.6 of narrow tables): 128
                                           // (default: 512 tiles of 48 KiB LDS keep three workgroups per CU loading; the layer is bound by reading h3) or 256
static std::atomic<int> g_fuse_mid{1};     // "fuse_mid": 1 = k_step_mid runs inside the mlp.6 GEMM when d_in = 256 (bf16 plans)
static inline int tune(const std::atomic<int>& v) { return v.load(std::memory_order_relaxed); }

// dp_impute_static's scratch: device memory beyond the caller's tables, held by every GPU the library runs on until the
// process ends.  On the host this name is only a handle for hipGetSymbolAddress; the address kernels and memsets use is
// DeviceState::scratch (one per device: each GPU has its own copy of the array at its own address).
static __device__ __align__(256) unsigned char scratch[DP_STATIC_SCRATCH_BYTES];

// Per-device launch state (a process normally drives one GPU, but nothing here may assume it): CU count for the
// persistent kernel's grid, the dynamic-LDS attribute of every GEMM variant, the address of this device's scratch array.
// Filled once per device under a mutex, by dp_plan_create (or by the first dp_linear on a device that has no plan), with
// that device current; afterwards read-only, so launches take no lock and make no set-up call.
struct DeviceState {
    std::atomic<int> ready{0};
    int num_cus = 0;
    void* scratch = nullptr;
    std::string lds_err;     // non-empty: a GEMM variant could not get its dynamic-LDS window on this device (raised when a bf16 GEMM is launched)
};
static DeviceState g_dev[64];
static std::mutex g_dev_mutex;
static int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 ? dev : -1;
}
static int ensure_device_ready(int dev);     // defined after the launchers (it names every kernel variant)
static const DeviceState* device_state() {
    const int dev = current_device();
    if (dev < 0) { fail(DP_EHIP, "hipGetDevice failed or device index >= 64"); return nullptr; }
    if (!g_dev[dev].ready.load(std::memory_order_acquire) && ensure_device_ready(dev) != DP_OK) return nullptr;
    return &g_dev[dev];
}
// for the bf16 GEMM launchers: the device state, or NULL (error set) when the LDS attributes could not be set on this device
static const DeviceState* device_state_bf16() {
    const DeviceState* ds = device_state();
    if (ds && !ds->lds_err.empty()) { fail(DP_EHIP, "%s", ds->lds_err.c_str()); return nullptr; }
    return ds;
}

// `slices` > 1: a batched launch (split-K products of the M-step): slice s multiplies A + s * a_bs by W + s * w_bs (byte
// strides; rows `ld` elements apart) into out + s * o_bs
static int launch_bf16_pp(const bf16_t* A, const bf16_t* W, const float* bias, void* out, int ldo, int64_t M, int N, int K,
                          int act, int out_f32, hipStream_t st, const MidStep* mid = nullptr, unsigned long long* clk = nullptr,
                          int slices = 1, int ld = 0, long long a_bs = 0, long long w_bs = 0, long long o_bs = 0, bool x3 = false) {
    const int nbn = N / PP_BN;
    const int64_t nwg64 = (M / PP_BM) * nbn * slices;
    if (nwg64 > 0x7fffffff) return fail(DP_EINVAL, "GEMM grid too large");
    const int nwg = (int)nwg64;
    const DeviceState* ds = device_state_bf16();
    if (!ds) return DP_EHIP;
    const int num_cus = ds->num_cus;
    MidStep ms = mid ? *mid : MidStep{};
    ms.clk = clk;
    const PpBatch pb = {ld ? ld : K, nwg / slices, a_bs, w_bs, o_bs};
#define DP_GO(ACT, EP)                                                                                              \
    hipLaunchKernelGGL((k_gemm_bf16_pp<ACT, EP>), dim3(std::min(nwg, num_cus)), dim3(PP_THREADS), PP_LDS_TOTAL, st, A, W, bias, out, ldo, K, nbn, nwg, ms, pb)
    if (x3) {
        if (act || !out_f32 || mid || K % 768) return fail(DP_EINVAL, "the three-product GEMM needs float32 output, no activation and K %% 768 == 0");
        DP_GO(false, EPI_F32X3);
    } else if (mid) {
        if (act || N != 256) return fail(DP_EINVAL, "the fused mid-step epilogue needs N = 256 and no activation");
        DP_GO(false, EPI_MID);
    } else if (act && !out_f32) DP_GO(true, EPI_BF16);
    else if (!act && !out_f32) DP_GO(false, EPI_BF16);
    else if (!act && out_f32) DP_GO(false, EPI_F32);
    else DP_GO(true, EPI_F32);
#undef DP_GO
    DP_LAUNCH_CHECK("k_gemm_bf16_pp");
    return DP_OK;
}

// "gemm_variant" 3 (default): the four-wave kernel (dp_gemm_bf16_w4.h) for the deep bf16-out layers (K >= 1024, K % 128 == 0,
// N <= 4096); the ping-pong kernel keeps the rest (float32 output, the fused mid step, K < 1024)
static int launch_bf16_w4(const bf16_t* A, const bf16_t* W, const float* bias, void* out, int ldo, int64_t M, int N, int K,
                          int act, hipStream_t st, unsigned long long* clk = nullptr) {
    const int nbn = N / 256;
    const int64_t nwg64 = (M / 256) * nbn;
    if (nwg64 > 0x7fffffff) return fail(DP_EINVAL, "GEMM grid too large");
    const int nwg = (int)nwg64;
    const DeviceState* ds = device_state_bf16();
    if (!ds) return DP_EHIP;
    const int grid = std::min(nwg, ds->num_cus);
    if (act) hipLaunchKernelGGL((k_gemm_bf16_w4<true>), dim3(grid), dim3(W4_THREADS), W4_LDS_TOTAL, st, A, W, bias, (bf16_t*)out, ldo, K, N, nwg, clk);
    else hipLaunchKernelGGL((k_gemm_bf16_w4<false>), dim3(grid), dim3(W4_THREADS), W4_LDS_TOTAL, st, A, W, bias, (bf16_t*)out, ldo, K, N, nwg, clk);
    DP_LAUNCH_CHECK("k_gemm_bf16_w4");
    return DP_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_bf16_cfg(const bf16_t* A, const bf16_t* W, const float* bias, void* out, int ldo, int64_t M, int N, int K,
                           int act, int out_f32, hipStream_t st) {
    constexpr int LDS = gemm_bf16_lds_bytes<BM, BN>();
    const int nbn = N / BN;
    const int64_t nwg64 = (M / BM) * nbn;
    if (nwg64 > 0x7fffffff) return fail(DP_EINVAL, "GEMM grid too large");
    const int nwg = (int)nwg64;
    if (!device_state_bf16()) return DP_EHIP;
#define DP_GO(ACT, OF)                                                                                              \
    hipLaunchKernelGGL((k_gemm_bf16<BM, BN, WM, WN, ACT, OF>), dim3(nwg), dim3(WM* WN * 64), LDS, st, A, W, bias, out, ldo, K, nbn, nwg)
    if (act && !out_f32) DP_GO(true, false);
    else if (!act && !out_f32) DP_GO(false, false);
    else if (!act && out_f32) DP_GO(false, true);
    else DP_GO(true, true);
#undef DP_GO
    DP_LAUNCH_CHECK("k_gemm_bf16");
    return DP_OK;
}

// K <= 256: the weight-stationary kernel (dp_gemm_bf16_ws.h)
static int launch_bf16_ws(const bf16_t* A, const bf16_t* W, const float* bias, void* out, int ldo, int64_t M, int N, int K,
                          int act, int out_f32, hipStream_t st) {
    const DeviceState* ds = device_state_bf16();
    if (!ds) return DP_EHIP;
    const int nbn = N / WS_BN;
    const int64_t n_rg64 = M / WS_ROWS;
    if (n_rg64 > 0x7fffffff) return fail(DP_EINVAL, "GEMM too tall");
    const int n_rg = (int)n_rg64;
    // one workgroup per CU at most, a whole number of workgroups per column tile, none without work
    const int slots = std::max(1, std::min(ds->num_cus / nbn, ceil_div(n_rg, 8)));
    const int grid = slots * nbn;
#define DP_GO(KT)                                                                                                   \
    do {                                                                                                            \
        const int lds = ws_lds_bytes(KT);                                                                           \
        if (act && !out_f32) hipLaunchKernelGGL((k_gemm_bf16_ws<KT, true, false>), dim3(grid), dim3(WS_THREADS), lds, st, A, W, bias, out, ldo, nbn, n_rg); \
        else if (!act && !out_f32) hipLaunchKernelGGL((k_gemm_bf16_ws<KT, false, false>), dim3(grid), dim3(WS_THREADS), lds, st, A, W, bias, out, ldo, nbn, n_rg); \
        else if (!act && out_f32) hipLaunchKernelGGL((k_gemm_bf16_ws<KT, false, true>), dim3(grid), dim3(WS_THREADS), lds, st, A, W, bias, out, ldo, nbn, n_rg); \
        else hipLaunchKernelGGL((k_gemm_bf16_ws<KT, true, true>), dim3(grid), dim3(WS_THREADS), lds, st, A, W, bias, out, ldo, nbn, n_rg); \
    } while (0)
    switch (K / 64) {
        case 1: DP_GO(1); break;
        case 2: DP_GO(2); break;
        case 3: DP_GO(3); break;
        case 4: DP_GO(4); break;
        default: return fail(DP_EINVAL, "weight-stationary GEMM needs K <= 256");
    }
#undef DP_GO
    DP_LAUNCH_CHECK("k_gemm_bf16_ws");
    return DP_OK;
}

// K = 256, bf16 output: the weight-stationary kernel with its K loop in asm (dp_gemm_bf16_ws2.h)
static int launch_bf16_ws2(const bf16_t* A, const bf16_t* W, const float* bias, void* out, int ldo, int64_t M, int N, int act, hipStream_t st) {
    const DeviceState* ds = device_state_bf16();
    if (!ds) return DP_EHIP;
    const int nbn = N / WS2_BN;
    const int64_t n_rg64 = M / WS2_ROWS;
    if (n_rg64 > 0x7fffffff) return fail(DP_EINVAL, "GEMM too tall");
    const int n_rg = (int)n_rg64;
    const int slots = std::max(1, std::min(ds->num_cus / nbn, ceil_div(n_rg, 8)));
    const int grid = slots * nbn;
    if (act) hipLaunchKernelGGL((k_gemm_bf16_ws2<true>), dim3(grid), dim3(WS2_THREADS), WS2_LDS_TOTAL, st, A, W, bias, (bf16_t*)out, ldo, nbn, n_rg);
    else hipLaunchKernelGGL((k_gemm_bf16_ws2<false>), dim3(grid), dim3(WS2_THREADS), WS2_LDS_TOTAL, st, A, W, bias, (bf16_t*)out, ldo, nbn, n_rg);
    DP_LAUNCH_CHECK("k_gemm_bf16_ws2");
    return DP_OK;
}

// One-time set-up of device `dev` (current on the calling thread): see DeviceState.
static int ensure_device_ready(int dev) {
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    DeviceState& d = g_dev[dev];
    if (d.ready.load(std::memory_order_relaxed)) return DP_OK;
    hipDeviceProp_t prop;
    DP_HIP(hipGetDeviceProperties(&prop, dev));
    d.num_cus = prop.multiProcessorCount;
    // a failure here (a device with a smaller LDS) must not take float32 plans down with it: it is recorded and raised by
    // the first bf16 GEMM launch on this device
#define DP_ATTR(kern, bytes)                                                                                              \
    do {                                                                                                                  \
        hipError_t e_ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));    \
        if (e_ != hipSuccess && d.lds_err.empty()) {                                                                      \
            (void)hipGetLastError();                                                                                      \
            d.lds_err = std::string("hipFuncSetAttribute(" #kern ", ") + std::to_string((int)(bytes)) + " bytes of dynamic LDS) failed: " + hipGetErrorString(e_); \
        }                                                                                                                 \
    } while (0)
    DP_ATTR((k_gemm_bf16_pp<true, EPI_BF16>), PP_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_pp<false, EPI_BF16>), PP_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_pp<true, EPI_F32>), PP_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_pp<false, EPI_F32>), PP_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_pp<false, EPI_MID>), PP_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_pp<false, EPI_F32X3>), PP_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_w4<true>), W4_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_w4<false>), W4_LDS_TOTAL);
#define DP_ATTR4(BM, BN, WM, WN)                                                                  \
    DP_ATTR((k_gemm_bf16<BM, BN, WM, WN, true, false>), (gemm_bf16_lds_bytes<BM, BN>()));        \
    DP_ATTR((k_gemm_bf16<BM, BN, WM, WN, false, false>), (gemm_bf16_lds_bytes<BM, BN>()));       \
    DP_ATTR((k_gemm_bf16<BM, BN, WM, WN, true, true>), (gemm_bf16_lds_bytes<BM, BN>()));         \
    DP_ATTR((k_gemm_bf16<BM, BN, WM, WN, false, true>), (gemm_bf16_lds_bytes<BM, BN>()))
    DP_ATTR4(256, 256, 2, 4);
    DP_ATTR4(256, 64, 4, 1);
    DP_ATTR4(128, 64, 4, 1);
#undef DP_ATTR4
#define DP_ATTR_WS(KT)                                                                            \
    DP_ATTR((k_gemm_bf16_ws<KT, true, false>), ws_lds_bytes(KT));                                 \
    DP_ATTR((k_gemm_bf16_ws<KT, false, false>), ws_lds_bytes(KT));                                \
    DP_ATTR((k_gemm_bf16_ws<KT, true, true>), ws_lds_bytes(KT));                                  \
    DP_ATTR((k_gemm_bf16_ws<KT, false, true>), ws_lds_bytes(KT))
    DP_ATTR_WS(1); DP_ATTR_WS(2); DP_ATTR_WS(3); DP_ATTR_WS(4);
    DP_ATTR((k_gemm_bf16_ws2<true>), WS2_LDS_TOTAL);
    DP_ATTR((k_gemm_bf16_ws2<false>), WS2_LDS_TOTAL);
#undef DP_ATTR_WS
#undef DP_ATTR
    // this device's copy of the scratch array: from here on host code uses d.scratch, never the array's name
    void* sp = nullptr;
    DP_HIP(hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)));
    d.scratch = sp;
    d.ready.store(1, std::memory_order_release);
    return DP_OK;
}

// Which kernel runs act(A W^T + b) of shape [*, K] -> [*, N]: ONE predicate, used by gemm() and by dp_plan_gemm_kernel
// (bench.py labels its roofline with what actually ran).
enum GemmKernel { GK_F32, GK_WS, GK_WS2, GK_W4, GK_PP, GK_SIMPLE, GK_NARROW128, GK_NARROW256, GK_NONE };
static GemmKernel pick_gemm(int dtype, int N, int K, int out_f32) {
    if (dtype == DP_DTYPE_F32) return GK_F32;
    if (dtype != DP_DTYPE_BF16) return GK_NONE;
    const int ws = tune(g_ws_gemm), variant = tune(g_gemm_variant);
    if (N % 256 == 0 && K == 256 && !out_f32 && ws == 3 && variant >= 2) return GK_WS2;
    if (N % 256 == 0 && (K <= 128 ? ws >= 1 : K <= 256 && ws == 2) && variant >= 2) return GK_WS;
    if (N % 256 == 0 && N <= W4_MAX_N && K >= tune(g_w4_min_k) && K % 128 == 0 && !out_f32 && variant == 3) return GK_W4;
    if (N % 256 == 0 && K >= 3 * PP_BK && variant >= 2) return GK_PP;
    if (N % 256 == 0) return GK_SIMPLE;
    return tune(g_narrow_bm) == 128 ? GK_NARROW128 : GK_NARROW256;
}
static const char* gemm_kernel_name(GemmKernel k) {
    switch (k) {
        case GK_F32: return "k_gemm_f32";
        case GK_WS: return "k_gemm_bf16_ws";
        case GK_WS2: return "k_gemm_bf16_ws2";
        case GK_W4: return "k_gemm_bf16_w4";
        case GK_PP: return "k_gemm_bf16_pp";
        case GK_SIMPLE: return "k_gemm_bf16<256,256>";
        case GK_NARROW128: return "k_gemm_bf16<128,64>";
        case GK_NARROW256: return "k_gemm_bf16<256,64>";
        default: return "";
    }
}

static int gemm(int dtype, const void* A, const void* W, const float* bias, void* out, int ldo, int64_t M, int N, int K, int act,
                int out_f32, hipStream_t st, unsigned long long* clk = nullptr) {
    if (M <= 0 || M % 256 || N <= 0 || N % 64 || K <= 0 || K % 64)
        return fail(DP_EINVAL, "gemm shape M=%lld N=%d K=%d: need M%%256==0, N%%64==0, K%%64==0", (long long)M, N, K);
    if (dtype == DP_DTYPE_BF16) {
        const bf16_t *a = (const bf16_t*)A, *w = (const bf16_t*)W;
        switch (pick_gemm(dtype, N, K, out_f32)) {
            case GK_WS: return launch_bf16_ws(a, w, bias, out, ldo, M, N, K, act, out_f32, st);
            case GK_WS2: return launch_bf16_ws2(a, w, bias, out, ldo, M, N, act, st);
            case GK_W4: return launch_bf16_w4(a, w, bias, out, ldo, M, N, K, act, st, clk);
            case GK_PP: return launch_bf16_pp(a, w, bias, out, ldo, M, N, K, act, out_f32, st, nullptr, clk);
            case GK_SIMPLE: return launch_bf16_cfg<256, 256, 2, 4>(a, w, bias, out, ldo, M, N, K, act, out_f32, st);
            case GK_NARROW128: return launch_bf16_cfg<128, 64, 4, 1>(a, w, bias, out, ldo, M, N, K, act, out_f32, st);
            default: return launch_bf16_cfg<256, 64, 4, 1>(a, w, bias, out, ldo, M, N, K, act, out_f32, st);
        }
    }
    if (dtype != DP_DTYPE_F32) return fail(DP_EINVAL, "unknown dtype %d", dtype);
    const int nbn = N / 64;
    const int64_t nwg = (M / 128) * nbn;
    if (nwg > 0x7fffffff) return fail(DP_EINVAL, "GEMM grid too large");
    if (act) hipLaunchKernelGGL(k_gemm_f32<true>, dim3((unsigned)nwg), dim3(256), 0, st, (const float*)A, (const float*)W, bias, (float*)out, ldo, K, nbn);
    else hipLaunchKernelGGL(k_gemm_f32<false>, dim3((unsigned)nwg), dim3(256), 0, st, (const float*)A, (const float*)W, bias, (float*)out, ldo, K, nbn);
    DP_LAUNCH_CHECK("k_gemm_f32");
    return DP_OK;
}

}  // namespace dp
#include "dp_mstep_host.h"
namespace dp {

// ---- workspace -------------------------------------------------------------------------------------------------
struct Workspace {
    float *x_hat, *d_cur, *F;
    void *u, *actA, *actB;
    int64_t M, Mp;
    size_t bytes;
};
static Workspace carve(const dp_plan* p, int64_t M, void* base) {
    Workspace w;
    w.M = M;
    w.Mp = round_up<int64_t>(M, 256);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += round_up<size_t>(n, 256); return base ? (void*)((char*)base + o) : (void*)nullptr; };
    w.x_hat = (float*)take((size_t)M * p->d_in * 4);
    w.d_cur = (float*)take((size_t)M * p->d_in * 4);
    w.F = (float*)take((size_t)w.Mp * p->N5 * 4);
    w.u = take((size_t)w.Mp * p->K1 * p->esize());
    w.actA = take((size_t)w.Mp * 2 * p->hid * p->esize());
    w.actB = take((size_t)w.Mp * 2 * p->hid * p->esize());
    w.bytes = off;
    return w;
}

// proj folded into mlp.0: bf16 plans only -- the float32 path keeps the reference's operation order, so its parity
// bound (1e-3 relative per step) is tested against the same sequence of roundings
static bool folded(const dp_plan* p) { return tune(g_fold_proj) && p->dtype == DP_DTYPE_BF16; }
// the step after a Heun iteration's first evaluation runs in mlp.6's epilogue when that GEMM's tile spans the whole row
static bool mid_fused(const dp_plan* p) {
    return tune(g_fuse_mid) && tune(g_gemm_variant) >= 2 && p->dtype == DP_DTYPE_BF16 && p->d_in == 256 && p->hid >= 3 * PP_BK;
}

// ---- one denoiser evaluation F(u, level): the Linear layers as GEMMs with fused epilogues (model.py:121-128) ------
// `level` = row of the b_eff / b01 tables (0 .. n_levels-1 the schedule's levels, n_levels / n_levels+1 scratch).
// `ev`: NULL, or 7 events of a profiling sample; [0..5] are recorded here, [6] by the caller after its step kernel.
// Rows per launch of the wide layers for a batch of Mp rows: at most g_m_chunk and at most Mp/2 (eval_F's buffer layout
// needs that), a multiple of 256; a remainder, if any, forms a smaller first chunk.  bf16 plans only (the float32 path
// keeps one launch per layer).
static int64_t chunk_rows(const dp_plan* p, int64_t Mp) {
    const int mc = tune(g_m_chunk);
    if (p->dtype != DP_DTYPE_BF16 || mc <= 0 || Mp <= mc) return Mp;
    const int64_t n = ceil_div<int64_t>(Mp, mc);                  // >= 2, and Mp / n > mc / 2 >= 128
    return std::max<int64_t>(256, Mp / n / 256 * 256);
}

// The wide layers run chunk by chunk (proj+mlp.0, mlp.2, mlp.4 of rows [r0, r1), then the next chunk) and mlp.6 once
// over the whole batch: a chunk's activations (2 x 128 MiB at 32 768 rows, H = 1024) are then still in the 256 MiB
// Infinity Cache when the next layer reads them, where a 65 536-row layer streams them through HBM; measured 6-8 %
// off mlp.0 / mlp.2 per row.  Rows are independent, so results do not depend on the chunking.  Every chunk writes its
// h1 / h2 to the START of actB / actA (the same, cache-resident addresses each time; chunk <= Mp/2 rows of width 2H fit
// the first half); h3 = mlp.4's output [Mp, H] is the second half of actB (with a single chunk h1 spans all of actB,
// but it is dead when mlp.4 writes h3).
static int eval_F(const dp_plan* p, const Workspace& w, int level, hipStream_t st, hipEvent_t* ev = nullptr, const MidStep* mid = nullptr) {
    const int H = p->hid, dt = p->dtype;
    const size_t es = p->esize();
    int rc;
#define DP_EV(i) do { if (ev && last) DP_HIP(hipEventRecord(ev[i], st)); } while (0)
    const int64_t chunk = chunk_rows(p, w.Mp);
    char* const h3 = (char*)w.actB + (size_t)w.Mp * H * es;       // second half of actB
    // ascending rows (the h3 layout below relies on it); the FIRST chunk takes the remainder, so the last one, which a
    // profiled evaluation times back to back with mlp.6, is always a full chunk
    const int64_t first_rows = w.Mp - (ceil_div<int64_t>(w.Mp, chunk) - 1) * chunk;
    for (int64_t r0 = 0, m = first_rows; r0 < w.Mp; r0 += m, m = chunk) {
        const bool last = r0 + m == w.Mp;
        const char* u = (const char*)w.u + (size_t)r0 * p->K1 * es;
        char* a2 = (char*)w.actA;                                 // h2 [m, 2H] (and proj's output [m, H] before it): the SAME
        char* b2 = (char*)w.actB;                                 // addresses for every chunk, h1 [m, 2H] likewise
        char* a1 = (char*)w.actA;
        char* b1 = h3 + (size_t)r0 * H * es;                      // this chunk's rows of h3 [Mp, H]
        DP_EV(0);
        if (folded(p)) {
            // (W0 Wp) u + (W0 b_eff + b0), SiLU: proj + emb + mlp.0 + mlp.1 in one GEMM; slot 0 of the profile stays empty
            DP_EV(1);
            if ((rc = gemm(dt, u, p->W01, p->b01 + (size_t)level * 2 * H, b2, 2 * H, m, 2 * H, p->K1, 1, 0, st))) return rc;
        } else {
            if ((rc = gemm(dt, u, p->W[0], p->b_eff + (size_t)level * H, a1, H, m, H, p->K1, 0, 0, st))) return rc;   // proj + emb (no act)
            DP_EV(1);
            if ((rc = gemm(dt, a1, p->W[1], p->B[1], b2, 2 * H, m, 2 * H, H, 1, 0, st))) return rc;  // mlp.0 + SiLU
        }
        DP_EV(2);
        // profiled evaluation: the mlp.2 launch of the last chunk also samples the clock the chip holds under it
        unsigned long long* clk = ev && last && p->prof_clk ? p->prof_clk + 4 * ((ev - p->prof_ev.data()) / 7) : nullptr;
        if ((rc = gemm(dt, b2, p->W[2], p->B[2], a2, 2 * H, m, 2 * H, 2 * H, 1, 0, st, clk))) return rc;  // mlp.2 + SiLU
        DP_EV(3);
        if ((rc = gemm(dt, a2, p->W[3], p->B[3], b1, H, m, H, 2 * H, 1, 0, st))) return rc;          // mlp.4 + SiLU
        DP_EV(4);
    }
    if (mid) {                                                                                              // mlp.6 + the mid step
        if ((rc = launch_bf16_pp((const bf16_t*)h3, (const bf16_t*)p->W[4], p->B[4], w.F, p->N5, w.Mp, p->N5, H, 0, 1, st, mid))) return rc;
    } else if ((rc = gemm(dt, h3, p->W[4], p->B[4], w.F, p->N5, w.Mp, p->N5, H, 0, 1, st))) return rc;   // mlp.6 -> f32
    if (ev) DP_HIP(hipEventRecord(ev[5], st));
#undef DP_EV
    return DP_OK;
}

// next profiling sample slot (7 events) if this evaluation is to be sampled, else NULL
static hipEvent_t* prof_slot(dp_plan* p, int tag) {
    if (!p->prof_stride) return nullptr;
    const long k = p->prof_evals++;
    if (k % p->prof_stride != 0 || p->prof_used >= DP_PROF_SAMPLES) return nullptr;
    p->prof_tag[p->prof_used] = tag;
    return &p->prof_ev[(size_t)7 * p->prof_used++];
}

// b_eff rows [first, first+count) from c_noise_dev[first..] (model.py:122-127)
static int compute_embedding(dp_plan* p, int first, int count, hipStream_t st) {
    const int H = p->hid;
    hipLaunchKernelGGL(k_posemb, dim3(grid_for((int64_t)count * H / 2)), dim3(256), 0, st, p->c_noise_dev + first, p->pe + (size_t)first * H, count, H);
    DP_LAUNCH_CHECK("k_posemb");
    const int waves = count * H, blocks = ceil_div(waves, 4);
    hipLaunchKernelGGL(k_small_linear<true>, dim3(blocks), dim3(256), 0, st, p->pe + (size_t)first * H, p->te_w0, p->te_b0, (const float*)nullptr,
                       p->hidden + (size_t)first * H, count, H, H);
    DP_LAUNCH_CHECK("k_small_linear<silu>");
    hipLaunchKernelGGL(k_small_linear<false>, dim3(blocks), dim3(256), 0, st, p->hidden + (size_t)first * H, p->te_w2, p->te_b2, (const float*)p->B[0],
                       p->b_eff + (size_t)first * H, count, H, H);
    DP_LAUNCH_CHECK("k_small_linear");
    // folded bias rows: b01[e] = W0 b_eff[e] + b_mlp.0
    const int waves01 = count * 2 * H;
    hipLaunchKernelGGL(k_small_linear<false>, dim3(ceil_div(waves01, 4)), dim3(256), 0, st, p->b_eff + (size_t)first * H, p->w0_f32, p->B[1],
                       (const float*)nullptr, p->b01 + (size_t)first * 2 * H, count, 2 * H, H);
    DP_LAUNCH_CHECK("k_small_linear(b01)");
    return DP_OK;
}

// W01[o, d] = sum_k W0[o, k] Wp[k, d]  (float32, k ascending), written zero-padded to [rows, ld] in the plan's dtype
__global__ __launch_bounds__(256) void k_fold_proj(const float* __restrict__ W0, const float* __restrict__ Wp, void* dst, int rows, int H, int D,
                                                   int ld, int to_bf16) {
    const int64_t n = (int64_t)rows * ld;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx / ld), d = (int)(idx - (int64_t)o * ld);
        float s = 0.f;
        if (d < D)
            for (int k = 0; k < H; ++k) s = fmaf(W0[(size_t)o * H + k], Wp[(size_t)k * D + d], s);
        if (to_bf16) ((bf16_t*)dst)[idx] = f2bf(s);
        else ((float*)dst)[idx] = s;
    }
}

__global__ void k_set2(float* dst, float a, float b) { dst[0] = a; dst[1] = b; }

static int check_plan(const dp_plan* p, bool need_weights) {
    if (!p) return fail(DP_EINVAL, "plan is NULL");
    if (need_weights && !p->weights_set) return fail(DP_ESTATE, "dp_plan_set_weights has not been called on this plan");
    int dev = -1;
    DP_HIP(hipGetDevice(&dev));
    if (dev != p->device) return fail(DP_ESTATE, "plan belongs to device %d but device %d is current", p->device, dev);
    return DP_OK;
}

static void free_plan_events(dp_plan* p) {
    for (auto e : p->prof_ev) (void)hipEventDestroy(e);
    p->prof_ev.clear();
    if (p->prof_clk) (void)hipFree(p->prof_clk);
    p->prof_clk = nullptr;
    p->prof_stride = 0;
}

static void free_plan_buffers(dp_plan* p) {
    for (int i = 0; i < 5; ++i) { if (p->W[i]) (void)hipFree(p->W[i]); if (p->B[i]) (void)hipFree(p->B[i]); p->W[i] = nullptr; p->B[i] = nullptr; }
    if (p->W01) (void)hipFree(p->W01);
    p->W01 = nullptr;
    float** f[] = {&p->te_w0, &p->te_b0, &p->te_w2, &p->te_b2, &p->b_eff, &p->c_noise_dev, &p->pe, &p->hidden, &p->w0_f32, &p->b01};
    for (auto q : f) { if (*q) (void)hipFree(*q); *q = nullptr; }
}

// chains that fit the static scratch array for this plan: carve() grows in steps of 256 chains (the GEMM row padding)
static int64_t static_max_chains(const dp_plan* p) {
    int64_t n = 0;
    while (carve(p, n + 256, nullptr).bytes <= DP_STATIC_SCRATCH_BYTES) n += 256;
    return n;
}

}  // namespace dp

// ================================================================================================================
extern "C" {

int dp_version(void) { return 1; }

const char* dp_last_error(void) { return g_err.c_str(); }

int dp_plan_create(dp_plan** out, int d_in, int hid, int num_steps, int n_resample, int dtype) {
    if (!out) return fail(DP_EINVAL, "out is NULL");
    *out = nullptr;
    if (d_in < 1 || d_in > 4096) return fail(DP_EINVAL, "d_in=%d out of range [1, 4096]", d_in);
    if (hid < 64 || hid % 64) return fail(DP_EINVAL, "hid=%d must be a positive multiple of 64", hid);
    if (num_steps < 2 || num_steps > 100000) return fail(DP_EINVAL, "num_steps=%d must be >= 2", num_steps);
    if (n_resample < 1) return fail(DP_EINVAL, "n_resample=%d must be >= 1", n_resample);
    if (dtype != DP_DTYPE_F32 && dtype != DP_DTYPE_BF16) return fail(DP_EINVAL, "dtype=%d unknown", dtype);
    dp_plan* p = new dp_plan();
    p->d_in = d_in; p->hid = hid; p->S = num_steps; p->R = n_resample; p->dtype = dtype;
    p->K1 = round_up(d_in, 64);
    p->N5 = round_up(d_in, 64);
    p->n_levels = 2 * num_steps - 1;
    hipError_t e = hipGetDevice(&p->device);
    if (e != hipSuccess) { delete p; return fail(DP_EHIP, "hipGetDevice failed: %s", hipGetErrorString(e)); }
    // one-time set-up of this device (CU count, dynamic-LDS attributes, address of its scratch array): after this no
    // sampler call makes a set-up or allocating HIP call
    const DeviceState* ds = device_state();
    if (!ds) { delete p; return DP_EHIP; }
    p->static_ws = ds->scratch;

    // sigma schedule (diffusion_utils.py:24,27-32), float32 op by op; sigma_min = max(0.002, 0), sigma_max = min(80, inf)
    const int S = num_steps;
    const double rho = 7.0;
    const float a = (float)pow(80.0, 1.0 / rho);
    const float b = (float)(pow(0.002, 1.0 / rho) - pow(80.0, 1.0 / rho));
    p->t.resize(S + 1);
    for (int i = 0; i < S; ++i) {
        const float q = (float)i / (float)(S - 1);
        const float base = a + q * b;
        p->t[i] = powf(base, 7.0f);
    }
    p->t[S] = 0.f;
    // The reference's `** rho` is torch's vectorised float32 pow, which is not correctly rounded: for its default
    // num_steps = 50 (main.py:28) one level (i = 30) comes out 1 ulp away from powf.  For that S the levels are the
    // reference's own bits (tests/golden/schedule.npz, written by running diffusion_utils.py:27-32 on torch CPU); any
    // other S is powf, within 1 ulp per level of what torch computes.
    if (S == 50) {
        static const uint32_t kRefS50[50] = {
            0x429ffffe, 0x428f0087, 0x427f26f4, 0x426332a8, 0x4249e937, 0x4233136b, 0x421e7db8, 0x420bf800, 0x41f6aad8, 0x41d8d878,
            0x41be2b51, 0x41a65b4f, 0x419125a6, 0x417c98fc, 0x415b2d56, 0x413d9eeb, 0x41238c7c, 0x410c9c9c, 0x40f0fa7d, 0x40cdc647,
            0x40af12f5, 0x409462e8, 0x407a8769, 0x40529ab9, 0x4030432c, 0x4012d631, 0x3ff373d0, 0x3fc8c7c1, 0x3fa4afbd, 0x3f864dbf,
            0x3f59b743, 0x3f2f52bf, 0x3f0c3658, 0x3ede9e61, 0x3eaf57ba, 0x3e88f281, 0x3e53fe11, 0x3e227dc5, 0x3df67ec7, 0x3db8d9b7,
            0x3d88ebf1, 0x3d48217d, 0x3d101fdb, 0x3ccc4173, 0x3c8e2def, 0x3c420fcf, 0x3c018c48, 0x3ba8bb2e, 0x3b55b26b, 0x3b03126d};
        memcpy(p->t.data(), kRefS50, sizeof kRefS50);
    }
    // per-level constants of sample_step (diffusion_utils.py:94-96) and of the re-noise (:52)
    const float gamma = churn_gamma(S);
    p->th.resize(S); p->churn.resize(S); p->renoise.resize(S);
    p->lv.resize(p->n_levels);
    for (int i = 0; i < S; ++i) {
        const float tc = p->t[i], tn = p->t[i + 1];
        const float th = tc + gamma * tc;
        p->th[i] = th;
        p->churn[i] = sqrtf(th * th - tc * tc);
        p->renoise[i] = sqrtf(tc * tc - tn * tn);
        p->lv[2 * i] = {th, precond_coef(th)};
        if (i < S - 1) p->lv[2 * i + 1] = {tn, precond_coef(tn)};
    }
    *out = p;
    return DP_OK;
}

void dp_plan_destroy(dp_plan* p) {
    if (!p) return;
    free_plan_buffers(p);
    free_plan_events(p);
    delete p;
}

int dp_plan_set_weights(dp_plan* p, const float* const w[7], const float* const b[7], void* stream) {
    int rc = check_plan(p, false);
    if (rc) return rc;
    if (!w || !b) return fail(DP_EINVAL, "w or b is NULL");
    for (int i = 0; i < 7; ++i)
        if (!w[i] || !b[i]) return fail(DP_EINVAL, "weight/bias %d is NULL", i);
    hipStream_t st = (hipStream_t)stream;
    const int H = p->hid, D = p->d_in;
    const size_t es = p->esize();
    const int bf = p->dtype == DP_DTYPE_BF16;
    free_plan_buffers(p);
    p->weights_set = false;
    // [out, in] of the five GEMM layers and their padded shapes
    const int rows[5] = {H, 2 * H, 2 * H, H, D}, cols[5] = {D, H, 2 * H, 2 * H, H};
    const int rows_p[5] = {H, 2 * H, 2 * H, H, p->N5}, cols_p[5] = {p->K1, H, 2 * H, 2 * H, H};
    for (int i = 0; i < 5; ++i) {
        DP_HIP(hipMalloc(&p->W[i], (size_t)rows_p[i] * cols_p[i] * es));
        DP_HIP(hipMalloc((void**)&p->B[i], (size_t)rows_p[i] * 4));
        hipLaunchKernelGGL(k_pack_weight, dim3(grid_for((int64_t)rows_p[i] * cols_p[i])), dim3(256), 0, st, w[i], p->W[i], rows[i], cols[i],
                           rows_p[i], cols_p[i], bf);
        DP_LAUNCH_CHECK("k_pack_weight");
        hipLaunchKernelGGL(k_pad_bias, dim3(grid_for(rows_p[i])), dim3(256), 0, st, b[i], p->B[i], rows[i], rows_p[i]);
        DP_LAUNCH_CHECK("k_pad_bias");
    }
    const int E = p->n_levels + 2;
    DP_HIP(hipMalloc((void**)&p->te_w0, (size_t)H * H * 4));
    DP_HIP(hipMalloc((void**)&p->te_w2, (size_t)H * H * 4));
    DP_HIP(hipMalloc((void**)&p->te_b0, (size_t)H * 4));
    DP_HIP(hipMalloc((void**)&p->te_b2, (size_t)H * 4));
    DP_HIP(hipMalloc((void**)&p->b_eff, (size_t)E * H * 4));
    DP_HIP(hipMalloc((void**)&p->pe, (size_t)E * H * 4));
    DP_HIP(hipMalloc((void**)&p->hidden, (size_t)E * H * 4));
    DP_HIP(hipMalloc((void**)&p->c_noise_dev, (size_t)E * 4));
    DP_HIP(hipMalloc(&p->W01, (size_t)2 * H * p->K1 * es));
    DP_HIP(hipMalloc((void**)&p->w0_f32, (size_t)2 * H * H * 4));
    DP_HIP(hipMalloc((void**)&p->b01, (size_t)E * 2 * H * 4));
    DP_HIP(hipMemcpyAsync(p->w0_f32, w[1], (size_t)2 * H * H * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_fold_proj, dim3(grid_for((int64_t)2 * H * p->K1)), dim3(256), 0, st, w[1], w[0], p->W01, 2 * H, H, D, p->K1, bf);
    DP_LAUNCH_CHECK("k_fold_proj");
    DP_HIP(hipMemcpyAsync(p->te_w0, w[5], (size_t)H * H * 4, hipMemcpyDeviceToDevice, st));
    DP_HIP(hipMemcpyAsync(p->te_w2, w[6], (size_t)H * H * 4, hipMemcpyDeviceToDevice, st));
    DP_HIP(hipMemcpyAsync(p->te_b0, b[5], (size_t)H * 4, hipMemcpyDeviceToDevice, st));
    DP_HIP(hipMemcpyAsync(p->te_b2, b[6], (size_t)H * 4, hipMemcpyDeviceToDevice, st));
    std::vector<float> cn(E, 0.f);
    for (int e = 0; e < p->n_levels; ++e) cn[e] = p->lv[e].c.c_noise;
    DP_HIP(hipMemcpyAsync(p->c_noise_dev, cn.data(), (size_t)E * 4, hipMemcpyHostToDevice, st));
    DP_HIP(hipStreamSynchronize(st));                                 // cn is a stack-lifetime host buffer
    if ((rc = compute_embedding(p, 0, p->n_levels, st))) return rc;
    DP_HIP(hipStreamSynchronize(st));
    p->weights_set = true;
    return DP_OK;
}

size_t dp_workspace_bytes(const dp_plan* p, int64_t n_chains) {
    if (!p || n_chains <= 0) { fail(DP_EINVAL, "dp_workspace_bytes: bad plan or n_chains"); return 0; }
    return carve(p, n_chains, nullptr).bytes;
}

static int impute_impl(dp_plan* p, const float* x_obs, const uint8_t* mask, float* out, int64_t n_rows, int num_trials, uint64_t seed,
                       uint64_t row_offset, int trial_offset, const float* noise, float* trace, int flags, void* workspace,
                       size_t workspace_bytes, void* stream) {
    int rc;
    if (!x_obs || !mask || !out || !workspace) return fail(DP_EINVAL, "x_obs, mask, out and workspace must be non-NULL");
    if (n_rows <= 0 || num_trials <= 0) return fail(DP_EINVAL, "n_rows=%lld and num_trials=%d must be positive", (long long)n_rows, num_trials);
    if (trial_offset < 0 || (int64_t)trial_offset + num_trials > 65536) return fail(DP_EINVAL, "global trial index must stay below 65536");
    if (((uintptr_t)workspace & 255) != 0) return fail(DP_EINVAL, "workspace must be 256-byte aligned");
    const int64_t M = n_rows * num_trials;
    Workspace w = carve(p, M, workspace);
    if (w.bytes > workspace_bytes)
        return fail(DP_EWORKSPACE, "workspace of %zu bytes is too small for %lld chains (need %zu)", workspace_bytes, (long long)M, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    const int S = p->S, R = p->R, D = p->d_in;
    const bool no_final = (flags & DP_NO_FINAL_LEVEL) != 0;
    const int n_iter = R * (S - 1) + (no_final ? 0 : 1);

    StepCommon c;
    c.x_hat = w.x_hat; c.d_cur = w.d_cur; c.F = w.F; c.u = w.u;
    c.x_obs = x_obs; c.mask = mask; c.noise = noise; c.trace = trace;
    c.n_cells = M * D; c.n_rows = n_rows;
    c.draw_stride = n_rows * D;
    c.trial_stride = (int64_t)(1 + 4 * (int64_t)R * (S - 1) + (no_final ? 0 : 3)) * c.draw_stride;
    c.seed = seed; c.row_offset = row_offset;
    c.D = D; c.ldF = p->N5; c.ldu = p->K1; c.u_bf16 = p->dtype == DP_DTYPE_BF16;
    c.trial_offset = trial_offset; c.num_trials = num_trials;

    // pad rows / pad columns of u stay zero for the whole call; the step kernels only write [0,M) x [0,D)
    DP_HIP(hipMemsetAsync(w.u, 0, (size_t)w.Mp * p->K1 * p->esize(), st));
    // 4 cells per thread when every row of every table starts 16-byte aligned (mask: 4-byte)
    const bool vec = D % 4 == 0 && ((uintptr_t)x_obs & 15) == 0 && ((uintptr_t)mask & 3) == 0 && ((uintptr_t)out & 15) == 0 &&
                     (!trace || ((uintptr_t)trace & 15) == 0);
    const int g = grid_for(c.n_cells / (vec ? 4 : 1));
#define DP_STEP(K, G, ...)                                                                       \
    do {                                                                                         \
        if (vec) hipLaunchKernelGGL(K<4>, dim3(G), dim3(256), 0, st, __VA_ARGS__);               \
        else hipLaunchKernelGGL(K<1>, dim3(G), dim3(256), 0, st, __VA_ARGS__);                   \
        DP_LAUNCH_CHECK(#K);                                                                     \
    } while (0)
    DP_STEP(k_step_init, g, c, p->t[0], p->churn[0], p->lv[0].c.c_in);

    for (int it = 0; it < n_iter; ++it) {
        const int i = std::min(it / R, S - 1);
        const int j = it - i * R;
        const Level& l0 = p->lv[2 * i];
        const float h = p->t[i + 1] - p->th[i];
        hipEvent_t* ev = prof_slot(p, i < S - 1 ? 0 : 2);
        MidStep ms;
        const bool fuse = i < S - 1 && mid_fused(p);
        if (fuse) {
            ms.x_hat = w.x_hat; ms.d_cur = w.d_cur; ms.u = (bf16_t*)w.u; ms.m_valid = M; ms.ldu = p->K1;
            ms.c_skip = l0.c.c_skip; ms.c_out = l0.c.c_out; ms.t_hat = p->th[i]; ms.h = h; ms.c_in_next = p->lv[2 * i + 1].c.c_in;
        }
        if ((rc = eval_F(p, w, 2 * i, st, ev, fuse ? &ms : nullptr))) return rc;
        if (i < S - 1) {
            const Level& l1 = p->lv[2 * i + 1];
            if (!fuse && !vec) {                                  // rows that cannot be walked 16 bytes at a time: one column group per thread
                hipLaunchKernelGGL(k_step_mid_grp, dim3(grid_for(M * ((D + 3) / 4))), dim3(256), 0, st, c, l0.c.c_skip, l0.c.c_out, p->th[i], h, l1.c.c_in);
                DP_LAUNCH_CHECK("k_step_mid_grp");
            } else if (!fuse) DP_STEP(k_step_mid, g, c, l0.c.c_skip, l0.c.c_out, p->th[i], h, l1.c.c_in);
            if (ev) DP_HIP(hipEventRecord(ev[6], st));
            ev = prof_slot(p, 1);
            if ((rc = eval_F(p, w, 2 * i + 1, st, ev))) return rc;
            const int add_re = j < R - 1;
            const int i_next = std::min((it + 1) / R, S - 1);
            const bool last = no_final && it == n_iter - 1;       // then d_cur, read by this launch cell by cell, receives x_t
            if (!vec && !noise) {                                 // Philox mode on rows that cannot be walked 16 bytes at a time
                hipLaunchKernelGGL(k_step_end_grp, dim3(grid_for(M * ((D + 3) / 4))), dim3(256), 0, st, c, it, l1.c.c_skip, l1.c.c_out, p->t[i + 1], h,
                                   p->renoise[i], add_re, p->churn[i_next], p->lv[2 * i_next].c.c_in, last ? w.d_cur : (float*)nullptr, last ? 0 : 1);
                DP_LAUNCH_CHECK("k_step_end_grp");
            } else {
                DP_STEP(k_step_end, g, c, it, l1.c.c_skip, l1.c.c_out, p->t[i + 1], h, p->renoise[i], add_re, p->churn[i_next],
                        p->lv[2 * i_next].c.c_in, last ? w.d_cur : (float*)nullptr, last ? 0 : 1);
            }
            if (last)
                DP_STEP(k_blend_sum, grid_for(n_rows * D / (vec ? 4 : 1)), c, (const float*)w.d_cur, out, (flags & DP_ACCUMULATE) ? 1 : 0);
            if (ev) DP_HIP(hipEventRecord(ev[6], st));
        } else {
            DP_STEP(k_step_final, grid_for(n_rows * D / (vec ? 4 : 1)), c, it, l0.c.c_skip, l0.c.c_out, p->th[i], h, out,
                    (flags & DP_ACCUMULATE) ? 1 : 0);
            if (ev) DP_HIP(hipEventRecord(ev[6], st));
        }
    }
#undef DP_STEP
    return DP_OK;
}

int dp_impute(dp_plan* p, const float* x_obs, const uint8_t* mask, float* out, int64_t n_rows, int num_trials, uint64_t seed,
              uint64_t row_offset, int trial_offset, const float* noise, float* trace, int flags, void* workspace,
              size_t workspace_bytes, void* stream) {
    int rc = check_plan(p, true);
    if (rc) return rc;
    return impute_impl(p, x_obs, mask, out, n_rows, num_trials, seed, row_offset, trial_offset, noise, trace, flags, workspace,
                       workspace_bytes, stream);
}

// dp_impute on the library's own scratch array instead of a caller workspace: no hipMalloc / hipFree / allocating call
// on this path (device set-up happened in dp_plan_create; the kernels take the array's per-device address as an ordinary
// pointer argument, exactly as they take a caller's workspace).
int dp_impute_static(dp_plan* p, const float* x_obs, const uint8_t* mask, float* out, int64_t n_rows, int num_trials, uint64_t seed,
                     uint64_t row_offset, int trial_offset, const float* noise, float* trace, int flags, void* stream) {
    int rc = check_plan(p, true);
    if (rc) return rc;
    if (n_rows <= 0 || num_trials <= 0) return fail(DP_EINVAL, "n_rows=%lld and num_trials=%d must be positive", (long long)n_rows, num_trials);
    if (!p->static_ws) return fail(DP_ESTATE, "the plan has no static scratch address");
    const int64_t M = n_rows * num_trials;
    const size_t need = carve(p, M, nullptr).bytes;
    if (need > DP_STATIC_SCRATCH_BYTES)
        return fail(DP_EWORKSPACE, "dp_impute_static: %lld chains need %zu bytes of scratch, the static array holds %zu (%lld chains for this plan); "
                    "use dp_impute with a caller workspace", (long long)M, need, (size_t)DP_STATIC_SCRATCH_BYTES, (long long)static_max_chains(p));
    return impute_impl(p, x_obs, mask, out, n_rows, num_trials, seed, row_offset, trial_offset, noise, trace, flags, p->static_ws,
                       DP_STATIC_SCRATCH_BYTES, stream);
}

int dp_plan_profile(dp_plan* p, int stride) {
    int rc = check_plan(p, false);
    if (rc) return rc;
    if (stride < 0) return fail(DP_EINVAL, "stride must be >= 0");
    if (stride && (p->prof_ev.empty() || !p->prof_clk)) {
        free_plan_events(p);                                       // a half-built set from an earlier failure
        p->prof_tag.assign(DP_PROF_SAMPLES, 0);
        p->prof_ev.reserve((size_t)7 * DP_PROF_SAMPLES);
        for (int i = 0; i < 7 * DP_PROF_SAMPLES; ++i) {
            hipEvent_t e;
            hipError_t err = hipEventCreate(&e);
            if (err != hipSuccess) { free_plan_events(p); return fail(DP_EHIP, "hipEventCreate failed: %s", hipGetErrorString(err)); }
            p->prof_ev.push_back(e);
        }
        hipError_t err = hipMalloc((void**)&p->prof_clk, (size_t)DP_PROF_SAMPLES * 4 * sizeof(unsigned long long));
        if (err != hipSuccess) { p->prof_clk = nullptr; free_plan_events(p); return fail(DP_EHIP, "hipMalloc(prof_clk) failed: %s", hipGetErrorString(err)); }
    }
    if (stride) {
        // a profiled launch of an earlier call may still be running on the caller's stream and would write its clock words
        // after a null-stream memset: wait for the device, then clear
        DP_HIP(hipDeviceSynchronize());
        DP_HIP(hipMemset(p->prof_clk, 0, (size_t)DP_PROF_SAMPLES * 4 * sizeof(unsigned long long)));
    }
    p->prof_stride = stride;
    p->prof_evals = 0;
    p->prof_used = 0;
    return DP_OK;
}

int dp_plan_profile_read(dp_plan* p, double ms_sum[8], int count[8]) {
    int rc = check_plan(p, false);
    if (rc) return rc;
    if (!ms_sum || !count) return fail(DP_EINVAL, "NULL argument");
    for (int k = 0; k < 8; ++k) { ms_sum[k] = 0.0; count[k] = 0; }
    for (int s = 0; s < p->prof_used; ++s) {
        hipEvent_t* ev = &p->prof_ev[(size_t)7 * s];
        DP_HIP(hipEventSynchronize(ev[6]));
        for (int k = folded(p) ? 1 : 0; k < 6; ++k) {           // folded: no separate proj launch between ev[0] and ev[1]
            float ms = 0.f;
            DP_HIP(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            if (k == 5 && p->prof_tag[s] == 0 && mid_fused(p)) continue;   // the mid step ran inside mlp.6: no launch to time
            const int slot = k < 5 ? k : 5 + p->prof_tag[s];        // 0..4 GEMM-1..5, 5 step_mid, 6 step_end, 7 step_final
            ms_sum[slot] += ms;
            count[slot] += 1;
        }
    }
    return DP_OK;
}

int dp_plan_profile_clock(dp_plan* p, double* cycles_sum, double* wall_us_sum, int* count) {
    int rc = check_plan(p, false);
    if (rc) return rc;
    if (!cycles_sum || !wall_us_sum || !count) return fail(DP_EINVAL, "NULL argument");
    *cycles_sum = *wall_us_sum = 0.0;
    *count = 0;
    if (!p->prof_clk || !p->prof_used) return DP_OK;
    for (int s = 0; s < p->prof_used; ++s) DP_HIP(hipEventSynchronize(p->prof_ev[(size_t)7 * s + 6]));
    std::vector<unsigned long long> h((size_t)p->prof_used * 4);
    DP_HIP(hipMemcpy(h.data(), p->prof_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int s = 0; s < p->prof_used; ++s) {
        const unsigned long long c0 = h[4 * s], c1 = h[4 * s + 1], r0 = h[4 * s + 2], r1 = h[4 * s + 3];
        if (c1 <= c0 || r1 <= r0) continue;                        // mlp.2 of this sample ran on a kernel without the clock hook (narrow plan, f32)
        *cycles_sum += (double)(c1 - c0);
        *wall_us_sum += (double)(r1 - r0) * 0.01;                   // s_memrealtime ticks at 100 MHz
        *count += 1;
    }
    return DP_OK;
}

// Embedding rows for up to two arbitrary sigmas into the plan's scratch rows; returns their row indices.
static int scratch_embedding(dp_plan* p, float s0, float s1, hipStream_t st) {
    const int first = p->n_levels;
    hipLaunchKernelGGL(k_set2, dim3(1), dim3(1), 0, st, p->c_noise_dev + first, precond_coef(s0).c_noise, precond_coef(s1).c_noise);
    DP_LAUNCH_CHECK("k_set2");
    return compute_embedding(p, first, 2, st);
}

int dp_denoise(dp_plan* p, const float* x, float sigma, float* D_out, float* F_out, int64_t n, void* workspace,
               size_t workspace_bytes, void* stream) {
    int rc = check_plan(p, true);
    if (rc) return rc;
    if (!x || !workspace || n <= 0) return fail(DP_EINVAL, "x, workspace must be non-NULL and n positive");
    if (!(sigma > 0.f)) return fail(DP_EINVAL, "sigma must be positive");
    if (((uintptr_t)workspace & 255) != 0) return fail(DP_EINVAL, "workspace must be 256-byte aligned");
    Workspace w = carve(p, n, workspace);
    if (w.bytes > workspace_bytes) return fail(DP_EWORKSPACE, "workspace of %zu bytes is too small for %lld rows (need %zu)", workspace_bytes, (long long)n, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    const Coef k = precond_coef(sigma);
    if ((rc = scratch_embedding(p, sigma, sigma, st))) return rc;
    DP_HIP(hipMemsetAsync(w.u, 0, (size_t)w.Mp * p->K1 * p->esize(), st));
    const int64_t cells = n * p->d_in;
    hipLaunchKernelGGL(k_scale_in, dim3(grid_for(cells)), dim3(256), 0, st, x, w.u, p->K1, (int)(p->dtype == DP_DTYPE_BF16), cells, p->d_in, k.c_in);
    DP_LAUNCH_CHECK("k_scale_in");
    if ((rc = eval_F(p, w, p->n_levels, st))) return rc;
    hipLaunchKernelGGL(k_precond_out, dim3(grid_for(cells)), dim3(256), 0, st, x, (const float*)w.F, p->N5, D_out, F_out, cells, p->d_in, k.c_skip, k.c_out);
    DP_LAUNCH_CHECK("k_precond_out");
    return DP_OK;
}

int dp_sample_step(dp_plan* p, int i, float t_cur, float t_next, const float* x_cur, const float* eps, float* x_next, int64_t n,
                   uint64_t seed, uint64_t row_offset, uint32_t block, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_plan(p, true);
    if (rc) return rc;
    if (!x_cur || !x_next || !workspace || n <= 0) return fail(DP_EINVAL, "x_cur, x_next, workspace must be non-NULL and n positive");
    if (!(t_cur > 0.f) || t_next < 0.f) return fail(DP_EINVAL, "need t_cur > 0 and t_next >= 0");
    const bool heun = i < p->S - 1;                                   // diffusion_utils.py:104
    if (heun && !(t_next > 0.f)) return fail(DP_EINVAL, "t_next must be positive at a Heun level (i < num_steps-1)");
    if (((uintptr_t)workspace & 255) != 0) return fail(DP_EINVAL, "workspace must be 256-byte aligned");
    Workspace w = carve(p, n, workspace);
    if (w.bytes > workspace_bytes) return fail(DP_EWORKSPACE, "workspace of %zu bytes is too small for %lld rows (need %zu)", workspace_bytes, (long long)n, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    const float gamma = churn_gamma(p->S);
    const float th = t_cur + gamma * t_cur;
    const float churn = sqrtf(th * th - t_cur * t_cur);
    const float h = t_next - th;
    const Coef k0 = precond_coef(th);
    const Coef k1 = heun ? precond_coef(t_next) : k0;
    if ((rc = scratch_embedding(p, th, heun ? t_next : th, st))) return rc;
    const int bf = p->dtype == DP_DTYPE_BF16, D = p->d_in;
    const int64_t cells = n * D;
    const int g = grid_for(cells);
    DP_HIP(hipMemsetAsync(w.u, 0, (size_t)w.Mp * p->K1 * p->esize(), st));
    hipLaunchKernelGGL(k_churn, dim3(g), dim3(256), 0, st, x_cur, eps, w.x_hat, w.u, p->K1, bf, cells, D, churn, k0.c_in, seed, row_offset, block);
    DP_LAUNCH_CHECK("k_churn");
    if ((rc = eval_F(p, w, p->n_levels, st))) return rc;
    if (heun) {
        StepCommon c;
        memset(&c, 0, sizeof c);
        c.x_hat = w.x_hat; c.d_cur = w.d_cur; c.F = w.F; c.u = w.u; c.n_cells = cells; c.n_rows = n;
        c.D = D; c.ldF = p->N5; c.ldu = p->K1; c.u_bf16 = bf;
        hipLaunchKernelGGL(k_step_mid<1>, dim3(g), dim3(256), 0, st, c, k0.c_skip, k0.c_out, th, h, k1.c_in);
        DP_LAUNCH_CHECK("k_step_mid");
        if ((rc = eval_F(p, w, p->n_levels + 1, st))) return rc;
        hipLaunchKernelGGL(k_heun_out, dim3(g), dim3(256), 0, st, (const float*)w.x_hat, (const float*)w.d_cur, (const float*)w.F, p->N5, x_next, cells, D,
                           k1.c_skip, k1.c_out, t_next, h, 1);
    } else {
        hipLaunchKernelGGL(k_heun_out, dim3(g), dim3(256), 0, st, (const float*)w.x_hat, (const float*)nullptr, (const float*)w.F, p->N5, x_next, cells, D,
                           k0.c_skip, k0.c_out, th, h, 0);
    }
    DP_LAUNCH_CHECK("k_heun_out");
    return DP_OK;
}

int dp_plan_schedule(const dp_plan* p, float* t_out) {
    if (!p || !t_out) return fail(DP_EINVAL, "NULL argument");
    memcpy(t_out, p->t.data(), p->t.size() * sizeof(float));
    return DP_OK;
}

int dp_plan_levels(const dp_plan* p, float* sigma_out) {
    if (!p || !sigma_out) return fail(DP_EINVAL, "NULL argument");
    for (int e = 0; e < p->n_levels; ++e) sigma_out[e] = p->lv[e].sigma;
    return DP_OK;
}

int dp_plan_embedding(const dp_plan* p, float* emb_out) {
    int rc = check_plan(p, true);
    if (rc) return rc;
    if (!emb_out) return fail(DP_EINVAL, "NULL argument");
    DP_HIP(hipDeviceSynchronize());
    DP_HIP(hipMemcpy(emb_out, p->b_eff, (size_t)p->n_levels * p->hid * 4, hipMemcpyDeviceToHost));
    return DP_OK;
}

int dp_linear(int dtype, const void* A, const void* W, const float* bias, void* out, int64_t M, int N, int K, int act, int out_f32,
              void* stream) {
    if (!A || !W || !bias || !out) return fail(DP_EINVAL, "NULL argument");
    if (dtype == DP_DTYPE_F32) out_f32 = 1;
    return gemm(dtype, A, W, bias, out, N, M, N, K, act ? 1 : 0, out_f32 ? 1 : 0, (hipStream_t)stream);
}

#ifdef DP_PP_STAMP
// diagnostic build only: where the ping-pong kernel writes its stamps (device pointer, [grid][16][2][8] u64) or NULL
int dp_debug_stamps(void* buf) {
    DP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(dp::g_pp_stamp_buf), &buf, sizeof buf));
    return DP_OK;
}
#endif

int64_t dp_plan_gemm_rows(const dp_plan* plan, int64_t n_chains) {
    if (!plan || n_chains <= 0) return 0;
    return chunk_rows(plan, round_up<int64_t>(n_chains, 256));
}

const char* dp_plan_gemm_kernel(const dp_plan* p, int layer) {
    if (!p || layer < 0 || layer > 4) { fail(DP_EINVAL, "dp_plan_gemm_kernel: bad plan or layer"); return ""; }
    const int H = p->hid;
    const bool fold = folded(p);
    switch (layer) {
        case 0: return fold ? "" : gemm_kernel_name(pick_gemm(p->dtype, H, p->K1, 0));
        case 1: return gemm_kernel_name(pick_gemm(p->dtype, 2 * H, fold ? p->K1 : H, 0));
        case 2: return gemm_kernel_name(pick_gemm(p->dtype, 2 * H, 2 * H, 0));
        case 3: return gemm_kernel_name(pick_gemm(p->dtype, H, 2 * H, 0));
        default: return mid_fused(p) ? "k_gemm_bf16_pp" : gemm_kernel_name(pick_gemm(p->dtype, p->N5, H, 1));
    }
}

int dp_tune(const char* key, int value) {
    if (!key) return fail(DP_EINVAL, "key is NULL");
    if (!strcmp(key, "gemm_variant")) {
        if (value < 1 || value > 3) return fail(DP_EINVAL, "gemm_variant must be 1, 2 or 3");
        g_gemm_variant = value;
        return DP_OK;
    }
    if (!strcmp(key, "narrow_bm")) {
        if (value != 128 && value != 256) return fail(DP_EINVAL, "narrow_bm must be 128 or 256");
        g_narrow_bm = value;
        return DP_OK;
    }
    if (!strcmp(key, "w4_min_k")) {
        if (value < 256) return fail(DP_EINVAL, "w4_min_k must be >= 256");
        g_w4_min_k = value;
        return DP_OK;
    }
    if (!strcmp(key, "ws_gemm")) {
        if (value < 0 || value > 3) return fail(DP_EINVAL, "ws_gemm must be 0, 1, 2 or 3");
        g_ws_gemm = value;
        return DP_OK;
    }
    if (!strcmp(key, "fuse_mid")) {
        if (value < 0 || value > 1) return fail(DP_EINVAL, "fuse_mid must be 0 or 1");
        g_fuse_mid = value;
        return DP_OK;
    }
    if (!strcmp(key, "m_chunk")) {
        if (value < 0 || value % 256) return fail(DP_EINVAL, "m_chunk must be 0 (off) or a positive multiple of 256");
        g_m_chunk = value;
        return DP_OK;
    }
    if (!strcmp(key, "fold_proj")) {
        if (value < 0 || value > 1) return fail(DP_EINVAL, "fold_proj must be 0 or 1");
        g_fold_proj = value;
        return DP_OK;
    }
    return fail(DP_EINVAL, "unknown tuning key '%s'", key);
}

// Launch shape of the two prep reductions (dp_prep.h): grid.y = chunks of 256 units, grid.x = row slabs, ~2048 blocks in all
// (8 per CU: full occupancy) but no slab shorter than 32 iterations of its block (every block ends in atomics).
struct PrepGrid { dim3 grid; int rows_per_block, vec; };
static PrepGrid prep_grid(int64_t n_rows, int d, bool vec4) {
    PrepGrid g;
    g.vec = vec4 ? 4 : 1;
    const int units = d / g.vec, by = ceil_div(units, 256);
    const int rpi = 256 / std::min(units, 256);
    const int64_t bx = std::max<int64_t>(1, std::min<int64_t>(ceil_div<int64_t>(n_rows, (int64_t)32 * rpi), 2048 / by));
    g.rows_per_block = (int)ceil_div<int64_t>(n_rows, bx);
    g.grid = dim3((unsigned)ceil_div<int64_t>(n_rows, g.rows_per_block), (unsigned)by);
    return g;
}

int dp_masked_mean_std(const float* X, const uint8_t* mask, int64_t n_rows, int d, float* mean, float* std_out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!X || !mask || !mean || !std_out || !workspace) return fail(DP_EINVAL, "NULL argument");
    if (n_rows <= 0 || d < 1 || d > 4096) return fail(DP_EINVAL, "need n_rows > 0 and 1 <= d <= 4096");
    if (workspace_bytes < (size_t)2 * d * sizeof(double) || ((uintptr_t)workspace & 7)) return fail(DP_EWORKSPACE, "workspace: need %zu bytes, 8-byte aligned", (size_t)2 * d * sizeof(double));
    hipStream_t st = (hipStream_t)stream;
    double* acc = (double*)workspace;
    double* cnt = acc + d;
    const int replicas = (int)std::min<size_t>(32, workspace_bytes / ((size_t)2 * d * sizeof(double)));
    DP_HIP(hipMemsetAsync(workspace, 0, (size_t)replicas * 2 * d * sizeof(double), st));
    const PrepGrid g = prep_grid(n_rows, d, ((uintptr_t)X & 15) == 0 && ((uintptr_t)mask & 3) == 0 && d % 4 == 0);
    for (int pass = 0; pass < 2; ++pass) {
        if (g.vec == 4) hipLaunchKernelGGL(k_col_stats<4>, g.grid, dim3(256), 0, st, X, mask, n_rows, d, pass, (const float*)mean, acc, cnt, g.rows_per_block, replicas);
        else hipLaunchKernelGGL(k_col_stats<1>, g.grid, dim3(256), 0, st, X, mask, n_rows, d, pass, (const float*)mean, acc, cnt, g.rows_per_block, replicas);
        DP_LAUNCH_CHECK("k_col_stats");
        hipLaunchKernelGGL(k_col_finish, dim3(ceil_div(d, 256)), dim3(256), 0, st, d, pass, acc, replicas, pass == 0 ? mean : std_out);
        DP_LAUNCH_CHECK("k_col_finish");
    }
    return DP_OK;
}

int dp_masked_errors(const float* pred, const float* truth, const uint8_t* mask, int64_t n_rows, int d, int ld_mask, int num_num,
                     double* sums, void* stream) {
    if (!pred || !truth || !mask || !sums) return fail(DP_EINVAL, "NULL argument");
    if (n_rows <= 0 || d < 1 || d > 4096 || num_num < 1 || num_num > d || ld_mask < num_num) return fail(DP_EINVAL, "need n_rows > 0, 1 <= num_num <= d <= 4096, ld_mask >= num_num");
    hipStream_t st = (hipStream_t)stream;
    DP_HIP(hipMemsetAsync(sums, 0, 3 * sizeof(double), st));
    const bool vec = (((uintptr_t)pred | (uintptr_t)truth) & 15) == 0 && ((uintptr_t)mask & 3) == 0 && d % 4 == 0 && ld_mask % 4 == 0 && num_num % 4 == 0;
    const PrepGrid g = prep_grid(n_rows, d, vec);
    if (g.vec == 4) hipLaunchKernelGGL(k_masked_errors<4>, g.grid, dim3(256), 0, st, pred, truth, mask, n_rows, d, ld_mask, num_num, sums, g.rows_per_block);
    else hipLaunchKernelGGL(k_masked_errors<1>, g.grid, dim3(256), 0, st, pred, truth, mask, n_rows, d, ld_mask, num_num, sums, g.rows_per_block);
    DP_LAUNCH_CHECK("k_masked_errors");
    return DP_OK;
}

size_t dp_mstep_workspace_bytes(int d_in, int hid, int64_t batch) {
    if (d_in < 1 || d_in > 4096 || hid < 64 || hid % 64 || batch < 1 || batch > (1 << 20)) { fail(DP_EINVAL, "dp_mstep_workspace_bytes: bad shape"); return 0; }
    return ms_carve(ms_dims(d_in, hid, batch), nullptr).bytes;
}

int dp_mstep_fwd_bwd(int d_in, int hid, int64_t batch, const float* x, const float* z_sigma, const float* z_noise,
                     const float* const w[7], const float* const b[7], float* const gw[7], float* const gb[7], float* loss,
                     float* percell, void* workspace, size_t workspace_bytes, void* stream) {
    if (d_in < 1 || d_in > 4096) return fail(DP_EINVAL, "d_in=%d out of range [1, 4096]", d_in);
    if (hid < 64 || hid % 64) return fail(DP_EINVAL, "hid=%d must be a positive multiple of 64", hid);
    if (batch < 1 || batch > (1 << 20)) return fail(DP_EINVAL, "batch=%lld out of range [1, 2^20]", (long long)batch);
    if (!x || !z_sigma || !z_noise || !w || !b || !gw || !gb || !loss || !workspace) return fail(DP_EINVAL, "NULL argument");
    for (int i = 0; i < 7; ++i)
        if (!w[i] || !b[i] || !gw[i] || !gb[i]) return fail(DP_EINVAL, "parameter / gradient pointer %d is NULL", i);
    if (((uintptr_t)workspace & 255) != 0) return fail(DP_EINVAL, "workspace must be 256-byte aligned");
    return mstep_fwd_bwd(d_in, hid, batch, x, z_sigma, z_noise, w, b, gw, gb, loss, percell, workspace, workspace_bytes, (hipStream_t)stream);
}

int dp_noise_field(float* out, int64_t n_rows, int d_in, uint64_t seed, uint64_t row_offset, int trial, uint32_t block, int comp,
                   void* stream) {
    if (!out || n_rows <= 0 || d_in <= 0 || d_in > 65535 || trial < 0 || trial > 65535 || comp < 0 || comp > 3)
        return fail(DP_EINVAL, "dp_noise_field: bad argument");
    hipLaunchKernelGGL(k_noise_field, dim3(grid_for(n_rows * d_in)), dim3(256), 0, (hipStream_t)stream, out, n_rows, d_in, seed, row_offset, trial,
                       block, comp);
    DP_LAUNCH_CHECK("k_noise_field");
    return DP_OK;
}

}  // extern "C"
