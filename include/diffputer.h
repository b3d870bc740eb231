/*
 * diffputer.h -- C ABI of libdiffputer_hip.so: DiffPuter's E-step imputation sampler on MI355X (gfx950).
 *
 * The reference (hengruizhang98/DiffPuter) is pure Python and has no FFI layer; the seam this library
 * sits behind is the Python import `from diffusion_utils import sample_step, impute_mask`
 * (main.py:14) and the two call sites main.py:180 and main.py:231.  Each entry point below names the
 * reference code it replaces (paths relative to the reference root).  The Python host side
 * (diffputer_amd/diffusion_utils.py) binds these with ctypes and keeps the reference's signatures;
 * INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - Every pointer named x, mask, out, noise, trace, workspace, w[], b[] is a DEVICE pointer owned by the
 *     caller (PyTorch).  The library borrows it for the duration of the call.  No torch types cross here.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  All work is enqueued on it; no entry
 *     point synchronises the device except dp_plan_set_weights (once per checkpoint) and the test getters.
 *   - Return value 0 = DP_OK, negative = error; dp_last_error() gives the message (thread-local).
 *   - Tables are row-major float32 [rows, d_in]; masks are uint8 [rows, d_in] with 1 = MISSING
 *     (generate_mask.py:513-514, main.py:85).
 *   - A plan is bound to the device that was current at dp_plan_create.
 *   - Threading: distinct plans may be driven from distinct threads at the same time (per-device launch state is set up
 *     once under a lock in dp_plan_create and read-only afterwards; dp_last_error is thread-local).  ONE plan must not be
 *     used from two threads at once (its profiling slots and scratch embedding rows are unsynchronised).  dp_tune values
 *     are process-wide atomics read at every evaluation: changing one while another thread is inside a sampler call gives
 *     that call a mix of old and new settings ("fold_proj" changes the rounding sequence, the others do not change bits).
 */
#ifndef DIFFPUTER_H
#define DIFFPUTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DP_OK            0
#define DP_EINVAL       -1   /* bad argument (shape, alignment, null pointer, unsupported size) */
#define DP_EHIP         -2   /* a HIP runtime call failed; message carries hipGetErrorString */
#define DP_ESTATE       -3   /* call order violated (e.g. impute before set_weights) */
#define DP_EWORKSPACE   -4   /* workspace too small for the requested number of chains */

#define DP_DTYPE_F32     0   /* GEMM operands float32, v_mfma_f32_32x32x2_f32 (exact f32 fma chain) */
#define DP_DTYPE_BF16    1   /* GEMM operands bfloat16, fp32 accumulate, v_mfma_f32_16x16x32_bf16 */

/* dp_impute flags */
#define DP_ACCUMULATE     1  /* out += sum over this call's trials (default: out = sum) */
#define DP_NO_FINAL_LEVEL 2  /* stop after the last resampled level and return x_t at sigma_min: the impute_mask of
                                baselines/Missdiff_SDE/diffusion_utils.py:24-60, whose last level has no else branch
                                (create the plan with n_resample = 10 for that baseline) */

typedef struct dp_plan dp_plan;

/* Library ABI version (this header: 1). */
int dp_version(void);

/* Message of the last error raised on the calling thread ("" if none). */
const char* dp_last_error(void);

/*
 * Create a sampler plan for a denoiser of input width d_in and hidden width hid.
 * Replaces: the schedule/constant set-up at the top of impute_mask (diffusion_utils.py:24-37, constants
 * :15-21, N = 20 at :37) and the shape part of MLPDiffusion.__init__ (model.py:97-119).
 *   num_steps   S  (main.py:28 --num_steps, default 50), >= 2
 *   n_resample  R  (the local `N = 20` of diffusion_utils.py:37; 10 in baselines/Missdiff_SDE), >= 1
 *   dtype       DP_DTYPE_F32 or DP_DTYPE_BF16
 * Constraints: hid % 64 == 0, 1 <= d_in <= 4096.
 */
int dp_plan_create(dp_plan** out, int d_in, int hid, int num_steps, int n_resample, int dtype);

/*
 * Load a checkpoint's 14 tensors (device pointers, float32, torch.nn.Linear layout [out, in] row-major).
 * Replaces: `model.load_state_dict(torch.load(.../model.pt))` per trial (main.py:170-173) and, for the
 * 2*S-1 noise levels a run visits, the whole timestep-embedding branch of MLPDiffusion.forward
 * (PositionalEmbedding model.py:24-30, sin/cos swap :123, time_embed :124) folded with proj.bias into one
 * bias row per level.
 * Order of w[]/b[]: 0 proj, 1 mlp.0, 2 mlp.2, 3 mlp.4, 4 mlp.6, 5 time_embed.0, 6 time_embed.2.
 * The plan keeps its own packed copies; the caller's tensors may be freed after the call returns
 * (the call synchronises `stream`).
 */
int dp_plan_set_weights(dp_plan* plan, const float* const w[7], const float* const b[7], void* stream);

void dp_plan_destroy(dp_plan* plan);

/* Bytes of device workspace dp_impute / dp_denoise / dp_sample_step need for `n_chains` chains
 * (a chain = one row x one trial).  0 on error. */
size_t dp_workspace_bytes(const dp_plan* plan, int64_t n_chains);

/*
 * The E-step sampler for `num_trials` trials of `n_rows` rows (n_rows * num_trials chains in one batch).
 * Replaces: impute_mask (diffusion_utils.py:23-68) including every sample_step (:90-109) and
 * Precond/MLPDiffusion forward (model.py:121-128,148-165) it makes, plus the caller's per-trial blend and
 * trial sum (main.py:182-188: rec*mask + x*(1-mask), stack().mean(0) -- the mean's division by T is left
 * to the caller).
 *   x_obs      [n_rows, d_in]  observed values, the `x` of impute_mask (zero at missing cells, main.py:164)
 *   mask       [n_rows, d_in]  uint8, 1 = missing
 *   out        [n_rows, d_in]  sum over the trials of rec*mask + x_obs*(1-mask)
 *   seed, row_offset, trial_offset
 *              counter-based Philox4x32-10 noise keyed by (seed; global row = row_offset + r, global trial =
 *              trial_offset + t, column group, draw block; one block serves the 4 columns of a group): results do not
 *              depend on how rows/trials are sharded, nor on the alignment of the tables.
 *   noise      NULL, or [num_trials, n_draws, n_rows, d_in] float32: the reference's Gaussian draws in
 *              program order (n_draws = 1 + 4*R*(S-1) + 3, without the + 3 under DP_NO_FINAL_LEVEL; draw 0 =
 *              torch.randn :25, then per inner iteration n_curr :44 (unused), n_prev :45, churn eps :96, n :52).
 *              Used instead of Philox: parity mode.
 *   trace      NULL, or [n_iter, num_trials, n_rows, d_in] float32 receiving x_t after every inner iteration
 *              (n_iter = R*(S-1) + 1; R*(S-1) under DP_NO_FINAL_LEVEL).
 *   flags      DP_ACCUMULATE | DP_NO_FINAL_LEVEL or 0
 */
int dp_impute(dp_plan* plan, const float* x_obs, const uint8_t* mask, float* out,
              int64_t n_rows, int num_trials, uint64_t seed, uint64_t row_offset, int trial_offset,
              const float* noise, float* trace, int flags,
              void* workspace, size_t workspace_bytes, void* stream);

/*
 * dp_impute for callers that cannot allocate (code under HIP graph capture, tight serving loops that impute a few rows per
 * request, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and same results, bit for
 * bit, as dp_impute -- minus the workspace: the sampler state and activations live in a file-scope __device__ array of
 * DP_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends.  The call makes no hipMalloc /
 * hipFree / other allocating call (device set-up, including the look-up of the array's per-device address, happens in
 * dp_plan_create).  Limit: n_rows * num_trials chains must satisfy dp_workspace_bytes(plan, chains) <=
 * DP_STATIC_SCRATCH_BYTES (2 816 chains at d_in = 256, hid = 1024, bf16; 3 840 at d_in = 9); larger calls are rejected with
 * DP_EWORKSPACE.  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams, two
 * threads, two plans) race; serialise them on one stream.  Nothing needs to start at zero except the padded network
 * input, which every call clears itself.
 */
#define DP_STATIC_SCRATCH_BYTES ((size_t)32 << 20)
int dp_impute_static(dp_plan* plan, const float* x_obs, const uint8_t* mask, float* out,
                     int64_t n_rows, int num_trials, uint64_t seed, uint64_t row_offset, int trial_offset,
                     const float* noise, float* trace, int flags, void* stream);

/*
 * One denoiser evaluation D(x, sigma) for a batch-wide scalar sigma.
 * Replaces: Precond.forward(x, sigma) with 0-dim sigma (model.py:148-165), i.e. the `net(x_hat, t_hat)`
 * calls at diffusion_utils.py:99,105.
 *   x [n, d_in] -> D_out [n, d_in] (may be NULL), F_out [n, d_in] = raw network output F_x (may be NULL)
 */
int dp_denoise(dp_plan* plan, const float* x, float sigma, float* D_out, float* F_out, int64_t n,
               void* workspace, size_t workspace_bytes, void* stream);

/*
 * One stochastic EDM step.  Replaces: sample_step(net, num_steps, i, t_cur, t_next, x_next)
 * (diffusion_utils.py:90-109): churn to t_hat = 1.02 t_cur (gamma = min(S_churn/S, sqrt2-1)), Euler step,
 * Heun correction iff i < num_steps-1.
 *   eps  [n, d_in] the churn draw of :96, or NULL to use Philox: the field dp_noise_field(.., trial 0, block, comp 2)
 */
int dp_sample_step(dp_plan* plan, int i, float t_cur, float t_next, const float* x_cur, const float* eps,
                   float* x_next, int64_t n, uint64_t seed, uint64_t row_offset, uint32_t block,
                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * Test/inspection getters (host output pointers; they synchronise the stream they use).
 *   dp_plan_schedule:  t_out[S+1] = sigma schedule with trailing 0 (diffusion_utils.py:24-32)
 *   dp_plan_levels:    sigma_out[2S-1] = the noise levels a run evaluates, [2i] = t_hat_i, [2i+1] = t_{i+1}
 *   dp_plan_embedding: emb_out[(2S-1) * hid] = proj.bias + time_embed(posemb(ln(sigma)/4)) per level
 */
int dp_plan_schedule(const dp_plan* plan, float* t_out);
int dp_plan_levels(const dp_plan* plan, float* sigma_out);
int dp_plan_embedding(const dp_plan* plan, float* emb_out);

/*
 * In-situ kernel timing for bench.py (no counterpart in the reference).  dp_plan_profile(plan, stride) with
 * stride > 0 makes dp_impute bracket every stride-th denoiser evaluation with HIP events on the caller's stream
 * (at most 256 samples; stride = 0 switches it off).  dp_plan_profile_read waits for the recorded events and
 * returns summed milliseconds and sample counts per kernel slot: 0..4 = the five GEMM layers (proj, mlp.0, mlp.2,
 * mlp.4, mlp.6; with "fold_proj" slot 0 stays empty and slot 1 is the folded proj+mlp.0 GEMM), 5 = step kernel after
 * the first evaluation, 6 = step kernel after the second evaluation, 7 = final-level step kernel.
 */
int dp_plan_profile(dp_plan* plan, int stride);
/* Rows per launch of the wide-layer GEMMs (proj/mlp.0, mlp.2, mlp.4) for a call of n_chains chains: the batch is cut
 * into equal chunks of at most "m_chunk" rows (dp_tune) in bf16 plans; mlp.6 and the step kernels take the whole batch.
 * With chunking on, profile slots 0..3 time the launches of ONE chunk.  0 on error. */
int64_t dp_plan_gemm_rows(const dp_plan* plan, int64_t n_chains);
int dp_plan_profile_read(dp_plan* plan, double ms_sum[8], int count[8]);
/* Name of the GEMM kernel that profile slot `layer` (0 proj, 1 mlp.0 or the folded proj+mlp.0, 2 mlp.2, 3 mlp.4, 4 mlp.6)
 * runs on under the current dp_tune settings: "k_gemm_bf16_w4", "k_gemm_bf16_pp", "k_gemm_bf16_ws", "k_gemm_bf16<..>",
 * "k_gemm_f32", or "" for a slot that launches nothing (slot 0 with "fold_proj").  The library's own dispatch predicate,
 * so that a benchmark labels its roofline with the kernel that ran.  No counterpart in the reference. */
const char* dp_plan_gemm_kernel(const dp_plan* plan, int layer);
/* The clock the chip held under the dominant kernel: every profiled evaluation's mlp.2 launch (on the four-wave or the ping-pong kernel, whichever runs that layer) has its
 * workgroup 0 read the shader clock (s_memtime) and the 100 MHz wall clock (s_memrealtime) at its start and end -- two
 * scalar reads per LAUNCH, none in unprofiled launches.  Returns the sums over the samples: shader cycles and wall
 * microseconds of that workgroup's lifetime (clock = cycles / us MHz; FLOP per cycle = launch FLOPs / cycles).  No
 * counterpart in the reference. */
int dp_plan_profile_clock(dp_plan* plan, double* cycles_sum, double* wall_us_sum, int* count);

/*
 * Single-kernel test entry: out[M, N] = act(A[M, K] @ W[N, K]^T + bias[N]) with the plan-independent GEMM
 * kernels (model.py:102-111 are five such layers).  dtype as above: for DP_DTYPE_BF16, A and W are bf16
 * (uint16 bit patterns) and out is bf16 unless out_f32 != 0; for DP_DTYPE_F32 everything is float32.
 * Constraints: M % 256 == 0, N % 64 == 0, K % 64 == 0.  act: 0 = none, 1 = SiLU.
 */
int dp_linear(int dtype, const void* A, const void* W, const float* bias, void* out,
              int64_t M, int N, int K, int act, int out_f32, void* stream);

/*
 * Process-wide tuning knobs for benchmarks and A/B tools (no counterpart in the reference).  Keys:
 *   "gemm_variant"  3 (default) = four-wave 256x256 kernel for the deep bf16-output layers (K >= 1024), ping-pong 256x256
 *                   kernel for the other wide ones; 2 = ping-pong kernel for all wide layers; 1 = the simple
 *                   double-buffered kernel everywhere; results are bit-identical.
 *   "narrow_bm"     rows per tile of the simple kernel on layers with N % 256 != 0 (mlp.6 of tables narrower than 256
 *                   columns): 128 (default) or 256; the layer is bound by reading its input, smaller tiles keep more loads in flight.
 *   "w4_min_k"      smallest K the four-wave kernel takes under "gemm_variant" 3 (default 1024, at least 256; it needs
 *                   K % 128 == 0, N % 256 == 0, N <= 4096 and bf16 output): below 1024 the ping-pong kernel is as fast or faster.
 *   "fold_proj"     1 (default) = in DP_DTYPE_BF16 plans proj and mlp.0 run as ONE Linear, (W0 Wp) u + (W0 (proj.bias + emb) + b0): no
 *                   activation separates them (model.py:126-128), so this is the same function with one rounding
 *                   of h0 fewer; 0 = two GEMMs as in the reference's module list.  Read at every evaluation.
 *   "m_chunk"       rows per launch of the wide layers in bf16 plans (default 32768; 0 = the whole batch in one launch):
 *                   proj+mlp.0, mlp.2, mlp.4 run chunk after chunk so that a chunk's activations stay in the Infinity
 *                   Cache between layers; rows are independent, results do not change.
 *   "ws_gemm"       which layers run on the weight-stationary kernel (dp_gemm_bf16_ws.h: the W tile stays in LDS, every wave
 *                   streams its own rows, no barrier after set-up, one wave's SiLU epilogue under its SIMD partner's MFMAs):
 *                   1 (default) = layers with K <= 128 (the folded proj+mlp.0 layer of tables up to 128 columns: 55 -> 31 us
 *                   per 32 768 x 2048 launch at K = 64), 2 = K <= 256 as well (on par with the ping-pong kernel there),
 *                   3 = as 2, with K = 256 on the variant whose K loop is inline assembly (dp_gemm_bf16_ws2.h; also on
 *                   par: DESIGN.md section 4), 0 = never.  Bit-identical results.
 *   "fuse_mid"      1 (default) = for bf16 plans with d_in = 256 the arithmetic after a Heun iteration's first evaluation
 *                   (diffusion_utils.py:100-101) runs in the epilogue of the mlp.6 GEMM instead of a kernel of its own;
 *                   bit-identical results.
 */
int dp_tune(const char* key, int value);

/*
 * Masked column statistics on the device.  Replaces: mean_std(data, mask) (dataset.py:161-169) as main.py:55 calls it:
 * per column, mean and (population) std over the OBSERVED cells (mask == 0); a column with no observed cell gets
 * count 1 like the reference (:164).  X [n_rows, d] float32 row-major, mask [n_rows, d] uint8, mean/std_out [d] float32
 * device pointers; workspace >= 2*d*8 bytes of device memory, 8-byte aligned (up to 32 times that is used when given: the
 * blocks' partial sums are then spread over as many copies, which shortens the chains of atomics).  Asynchronous on `stream`.
 */
int dp_masked_mean_std(const float* X, const uint8_t* mask, int64_t n_rows, int d, float* mean, float* std_out,
                       void* workspace, size_t workspace_bytes, void* stream);

/*
 * Scoring sums on the device.  Replaces: the arithmetic of get_eval (dataset.py:199-201): over the MISSING cells
 * (mask == 1) of the first num_num (numeric) columns, sums[0] = sum |pred - truth|, sums[1] = sum (pred - truth)^2,
 * sums[2] = count, float64 on the device; MAE = sums[0]/sums[2], RMSE = sqrt(sums[1]/sums[2]).
 * pred, truth [n_rows, d]; mask [n_rows, ld_mask] (the ORIGINAL-column mask of dataset.py:183 may be wider than d).
 */
int dp_masked_errors(const float* pred, const float* truth, const uint8_t* mask, int64_t n_rows, int d, int ld_mask,
                     int num_num, double* sums, void* stream);

/*
 * One M-step training step's arithmetic: the EDM loss of a batch and its gradient for the 14 parameter tensors.
 * Replaces: `loss = model(inputs)` ... `loss.backward()` (main.py:124-131), i.e. Model.forward (model.py:171-181) ->
 * EDMLoss.__call__ (diffusion_utils.py:200-215) -> Precond.forward with per-row sigma (model.py:148-165) ->
 * MLPDiffusion.forward (model.py:121-128), and autograd's backward through all of it.  The optimiser step stays the
 * caller's (torch.optim.Adam, main.py:109,132).
 *   x        [batch, d_in]  the training rows of this batch (main.py:122)
 *   z_sigma  [batch]        the standard-normal draw behind sigma = exp(1.2 z - 1.2)   (torch.randn, diffusion_utils.py:202)
 *   z_noise  [batch, d_in]  the standard-normal draw behind n = z sigma                (torch.randn_like, :208)
 *   w, b     the 14 parameters, float32, torch.nn.Linear layout, order as dp_plan_set_weights
 *   gw, gb   receive the gradients (same shapes; overwritten, not accumulated)
 *   loss     one float32: mean over the batch's cells                                  (model.py:180-181)
 *   percell  NULL, or [batch, d_in] receiving the per-cell loss                        (diffusion_utils.py:213)
 * Arithmetic: float32 everywhere except the GEMMs, whose float32 operands are split into bf16 hi + lo and multiplied as
 * three bf16 MFMA products with float32 accumulation (error per product ~1e-5 relative; diffputer_amd/csrc/dp_mstep.h).
 * Deterministic: same inputs, same bits (every reduction has a fixed order).  Asynchronous on `stream`; allocates nothing
 * (workspace >= dp_mstep_workspace_bytes(d_in, hid, batch), 256-byte aligned).
 */
size_t dp_mstep_workspace_bytes(int d_in, int hid, int64_t batch);
int dp_mstep_fwd_bwd(int d_in, int hid, int64_t batch, const float* x, const float* z_sigma, const float* z_noise,
                     const float* const w[7], const float* const b[7], float* const gw[7], float* const gb[7],
                     float* loss, float* percell, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Fill out[n_rows, d_in] with one of the Gaussian fields dp_impute draws when noise == NULL, for global rows
 * row_offset.. and global trial `trial`: block = inner-iteration index (or 0xFFFFFFFF = before the first iteration),
 * comp = kind: 0 = n_prev of that iteration (diffusion_utils.py:45; for 0xFFFFFFFF the initial x_T draw :25),
 * 1 = the re-noise n (:52), 2 = the churn eps (:96) of the NEXT iteration.  Laid out in the reference's draw order these
 * fields replay a Philox run exactly (tests/test_gpu_parity.py).  No counterpart in the reference (it uses torch's
 * global RNG, unseeded).
 */
int dp_noise_field(float* out, int64_t n_rows, int d_in, uint64_t seed, uint64_t row_offset, int trial,
                   uint32_t block, int comp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFPUTER_H */
