/* use_hip.h -- C ABI of libuse_hip.so: the MI355X (gfx950) implementation of the SGMSE reverse-SDE
 * sampling path of nanless/universal-speech-enhancement.
 *
 * Boundary (reference file:line, all Python -- the reference has no FFI for this path; the entry points below
 * are what a ctypes binding of the path binds):
 *   use_score        <- ScoreModel.forward / forward_score      src/models/components/sgmse/model_wrapper.py:135-145
 *                       (= -NCSNpp.forward(cat[x, Y], t)         .../backbones/ncsnpp.py:324-501)
 *   use_sample       <- sampling.get_pc_sampler()->pc_sampler()  .../sampling/__init__.py:23-73
 *   use_set_weight   <- LightningModule checkpoint load: state_dict keys "all_modules.<i>....", "output_layer.*"
 *                       (.../backbones/ncsnpp.py:116-316)
 *   use_sde_*        <- OUVESDE.prior_sampling (sdes.py:248-254), ReverseDiffusionPredictor / EulerMaruyamaPredictor
 *                       update_fn (sampling/predictors.py:40-68), LangevinCorrector / AnnealedLangevinDynamics
 *                       update_fn (sampling/correctors.py:37-98)
 *
 * Conventions: every function returns 0 on success or a negative USE_E_* code and never throws; the failing
 * call's message is available from use_last_error() (thread-local).  The caller owns every input / output
 * buffer and passes raw DEVICE pointers (complex64 spectrograms are [B][1][F][T'] == interleaved (re, im)
 * floats, T' a multiple of 64) plus the HIP stream to run on; the library owns weights-on-device, workspace and
 * captured graphs inside the handle.  One handle per (process, device); a handle is not re-entrant.  No call
 * synchronises the device except use_commit_weights / use_plan / use_destroy.
 */
#ifndef USE_HIP_H
#define USE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct use_handle use_handle;
typedef void* use_stream_t; /* hipStream_t */

enum { USE_OK = 0, USE_E_INVALID = -1, USE_E_HIP = -2, USE_E_STATE = -3, USE_E_NOMEM = -4 };
enum { USE_PREC_FP32 = 0, USE_PREC_BF16 = 1, USE_PREC_FP16 = 2 };   /* storage of activations / conv weights; accumulation is always fp32 */
enum { USE_PRED_REVERSE_DIFFUSION = 0, USE_PRED_EULER_MARUYAMA = 1, USE_PRED_NONE = 2 };
enum { USE_CORR_NONE = 0, USE_CORR_LANGEVIN = 1, USE_CORR_ALD = 2 };

typedef struct use_config {
    int nf;                /* base width (128)                                   ncsnpp.py:46            */
    int n_levels;          /* len(ch_mult)                                                             */
    int ch_mult[8];        /* (1,1,2,2,2,2,2) for NCSNppLarge                    ncsnpp.py:512-517       */
    int num_res_blocks;    /* 2                                                                        */
    int n_freq;            /* F = n_fft/2+1 = 512                                SGMSE_Large.yaml:11     */
    int precision;         /* USE_PREC_*: activation/weight storage; accumulation is always fp32        */
    float theta, sigma_min, sigma_max; /* OUVE SDE (1.5, 0.05, 0.5)              sdes.py:184             */
    /* NCSNpp(discriminative=True), the generator of the LSGAN refine stage (ncsnpp.py:86-92,
     * GAN/generator/ncsnpp/model_wrapper.py:54): 0 everywhere = the score network above */
    int input_channels;    /* real input channels: 0 or 4 = (x, y) re/im; 2 = (y) alone; 6 = (x, y, y2)   ncsnpp.py:63,92 */
    int unconditional;     /* 1: no time embedding (conditional=False)                                 ncsnpp.py:89   */
    int no_sigma_scale;    /* 1: output not divided by t (scale_by_sigma=False)                        ncsnpp.py:90   */
} use_config;

typedef struct use_sampler_config {
    int N;                 /* reverse steps                                     model_wrapper.py:266    */
    int predictor;         /* USE_PRED_*                                                               */
    int corrector;         /* USE_CORR_*                                                               */
    int corrector_steps;   /* ignored for USE_CORR_NONE                                                */
    float snr;             /* corrector snr                                                            */
    float t_eps;           /* smallest time (3e-2)                               model_wrapper.py:27     */
    int use_graph;         /* 1: capture the whole loop in a hipGraph and replay it                    */
} use_sampler_config;

/* Process-wide tuning knobs (no reference counterpart).  "subbatch" (default -1 = by batch size: three sub-batches for 6-11 items, else two; round 5: batch 8 as 3 + 3 + 2 is 1.6 % faster than 4 + 4): batches of >= 4 items are evaluated as
 * sub-batches (>= 2 items each) on separate streams, staggered so that the small-map kernels of one run beside the large convolutions of
 * the others; items never interact inside the network, so results are identical to subbatch = 0 (read at use_plan).  "conv_v4_min_blocks": smallest per-image grid (workgroups) the
 * wide-tile convolution kernel is selected for, default 80; results do not depend on it beyond rounding order.  "conv_v5" (default 1): the wide-tile kernel
 * on v_mfma_f32_16x16x32 in the 16-bit storage modes (conv_v5_kernel; 0: conv_v4_kernel, v_mfma_f32_32x32x16) - bit-identical results, less energy per FLOP.  "stats_part" (default 1): on maps
 * above "gn_inline" pixels (default 128 x 160) the convolutions write per-workgroup GroupNorm partial totals with plain stores and the
 * finalisation sums them, instead of 64-bit atomics on the item's totals - integer sums either way: bit-identical results (read at use_plan).
 * "fir_strip" (default 1): the res-block down-sampler walks 8- or 4-row strips (0: the 2 x 2 block form; 8 / 4: forced) - bit-identical.
 * "conv_in_wgs" (default 256): most workgroups per item of the input convolution (each walks tiles / conv_in_wgs tiles; stored values do not depend on
 * it; read at use_plan).
 * ALL options are read at use_plan (they size workspace buffers - "conv_in_wgs", "stats_part", "gn_inline", "subbatch*" - or choose kernels):
 * every use_set_option marks the plans built before it stale.  use_score / use_forward / use_sample* / use_set_sampler on a stale plan return
 * USE_E_STATE ("... call use_plan (and use_set_sampler) again" in use_last_error()); nothing is launched, the handle stays usable.  Should an
 * evaluation still not fit its workspace the entry point returns USE_E_STATE as well (output invalid) - never a write past the allocation,
 * never abort() (round 6; SURVEY 8b "never throw across the ABI").
 * Others: "stagger_level", "gn_inline", "plan_cache", "attn_fused", "pyr_ws", "conv_sk_max_px", "wgrad_mfma16", "wgrad_blocks" (INTEGRATION.md). */
int use_set_option(const char* name, long long value);
const char* use_last_error(void);
const char* use_version(void);

int use_create(const use_config* cfg, int device, use_handle** out);
int use_destroy(use_handle* h);

/* Weights.  name = reference state-dict key relative to the score net ("all_modules.4.Conv_0.weight", ...);
 * data = HOST float32, C-contiguous, shape as in the reference.  use_commit_weights packs (transposes to
 * [tap][cout][cin], converts to the compute dtype) and uploads one blob. */
int use_set_weight(use_handle* h, const char* name, const float* data, const int64_t* shape, int ndim);
int use_commit_weights(use_handle* h);
/* Multi-GPU: the packed device blob can be broadcast (RCCL) instead of re-packing on every rank:
 * rank 0 commits, the others call use_alloc_weight_blob, all ranks broadcast [ptr, ptr+bytes). */
int use_alloc_weight_blob(use_handle* h);
int use_weight_blob(use_handle* h, void** dev_ptr, size_t* bytes);
/* Packed weight file: a versioned header (architecture, precision, blob layout version, crc32) + the device blob, so that a
 * deployment starts from one read instead of a Lightning checkpoint (which replaces `torch.load` + `load_state_dict`,
 * predict.py:79).  use_save_weight_blob works after use_set_weight of every tensor (packs on the host: no GPU needed) or
 * after a commit / load; use_load_weight_blob replaces use_set_weight x N + use_commit_weights and refuses files packed
 * for another configuration, precision or layout version. */
int use_save_weight_blob(use_handle* h, const char* path);
int use_load_weight_blob(use_handle* h, const char* path);
int use_num_expected_weights(use_handle* h);
int use_expected_weight(use_handle* h, int index, const char** name, int64_t* shape4, int* ndim);

/* Workspace for batch B and T' padded frames (multiple of 64).  Re-planning is allowed. */
int use_plan(use_handle* h, int B, int Tpad);
int use_workspace_bytes(use_handle* h, size_t* bytes);

/* One score evaluation: out = -score_net(cat[x, y], t).  x, y, out: complex64 [B,1,F,T']; t: float32 [B] (device). */
int use_score(use_handle* h, const void* x, const void* y, const float* t, void* out, use_stream_t stream);
/* The same for a handle created with input_channels = 6 (ScoreModel(condition="both"), model_wrapper.py:43-46, 287-288):
 * out = -score_net(cat[x, y, y2], t) with the two conditioning spectrograms [Y, Y_denoised]. */
int use_score2(use_handle* h, const void* x, const void* y, const void* y2, const float* t, void* out, use_stream_t stream);

/* Whole PC sampler: prior sampling -> N x (corrector, predictor) -> x_mean of the last predictor step.
 * noise: complex64 [n_draws][B,1,F,T'] consumed in the reference's order (prior, then per step the corrector
 * draws followed by the predictor draw), or NULL to draw on-device (Philox4x32-10, `seed`). */
int use_set_sampler(use_handle* h, const use_sampler_config* sc);
int use_num_noise_draws(use_handle* h);
int use_get_timesteps(use_handle* h, float* out, int n);
int use_sample(use_handle* h, const void* y, const void* noise, uint64_t seed, void* out, use_stream_t stream);
/* The same loop with the score conditioning separated from the SDE's y: the network sees cat[x, cond] while the drift, the
 * prior mean and the result refer to y -- ScoreModel.sample with condition="denoised" (conditioning = the GAN-denoised
 * spectrogram) and sde_input "noisy" or "denoised" (model_wrapper.py:283-301).  cond == NULL is use_sample. */
int use_sample_cond(use_handle* h, const void* y, const void* cond, const void* noise, uint64_t seed, void* out, use_stream_t stream);
/* condition="both": the network sees cat[x, cond, cond2] (6-channel handle; cond == NULL: the SDE's y). */
int use_sample_cond2(use_handle* h, const void* y, const void* cond, const void* cond2, const void* noise, uint64_t seed, void* out,
                     use_stream_t stream);

/* Element-wise SDE pieces for callers that drive the loop themselves through the reference's
 * Predictor / Corrector registries (uniform t over the batch). n = number of complex elements. */
int use_sde_prior(use_handle* h, const void* y, const void* noise, uint64_t seed, void* x, int64_t n, use_stream_t s);
int use_sde_predictor(use_handle* h, int predictor, float t, int N, const void* x, const void* y, const void* score,
                      const void* noise, uint64_t seed, void* x_out, void* x_mean, int64_t n, use_stream_t s);
int use_sde_corrector(use_handle* h, int corrector, float t, float snr, int B, const void* x, const void* score,
                      const void* noise, uint64_t seed, void* x_out, void* x_mean, int64_t n, use_stream_t s);

/* The device noise stream of use_sample(noise = NULL, seed), draw by draw: out[i] (complex64, n elements = the whole batch tensor
 * [B,1,F,T']) = the z that draw `draw` uses at element i -- draw 0 is the prior's randn_like (sdes.py:254), then per reverse step the
 * corrector draws (sampling/correctors.py:54) followed by the predictor draw (sampling/predictors.py:63), the reference's order of
 * consumption.  use_sample(noise = [use_fill_noise(seed, d) for d in 0..use_num_noise_draws-1]) is bit-identical to
 * use_sample(noise = NULL, seed): this is how the timed (device-noise) branch is pinned to the oracle in tests/. */
int use_fill_noise(use_handle* h, uint64_t seed, int draw, void* out, int64_t n, use_stream_t s);

/* Raw backbone output NCSNpp.forward(cat[x, y], t) (ncsnpp.py:324-501), i.e. without the sign flip of use_score.
 * x, y: complex64 [B,1,F,T'] device; y must be null when input_channels == 2 (the input is x alone), t must be null when
 * the handle is unconditional and may be null when it is conditional only if no_sigma_scale... (not supported: give t). */
int use_forward(use_handle* h, const void* x, const void* y, const float* t, void* out, use_stream_t stream);

/* Spectrogram glue either side of the sampler, handle-free (ScoreModel.spec_fwd + pad_spec, model_wrapper.py:92-96,275-278,
 * util/other.py:128-135; ScoreModel.spec_back + crop, model_wrapper.py:98-103,320).  stft: complex64 [B,F,T] (torch.stft
 * output), Y / X: complex64 [B,1,F,Tpad].  fwd: Y = |S|^e e^{j arg S} * factor, zero for T <= t < Tpad; back: its inverse. */
int use_spec_fwd(const void* stft, void* Y, int B, int F, int T, int Tpad, float factor, float exponent, use_stream_t s);
int use_spec_back(const void* X, void* stft, int B, int F, int T, int Tpad, float factor, float exponent, use_stream_t s);

/* The transforms themselves on the device, fused with that glue: one kernel per direction, no intermediate
 * buffer.  use_stft_fwd = pad_spec(spec_fwd(torch.stft(wav, n_fft, hop, window, center=True, return_complex=True)))
 * (model_wrapper.py:116-118, 92-96, util/other.py:128-135); use_istft_back = torch.istft(spec_back(X), n_fft, hop, window,
 * center=True, length=L) over all Tpad frames (model_wrapper.py:98-103, 120-122, 320).  wav: float32 [B][L] device, window:
 * float32 [n_fft] device (periodic Hann in the reference's configs), Y / X: complex64 [B][1][n_fft/2+1][Tpad].  n_fft even
 * (1022 in SGMSE_Large.yaml - not a power of two: the transforms are table-driven direct sums, 10 GFLOP per batch of 8 x 4 s),
 * L > n_fft/2 (reflect padding), Tpad >= 1 + L / hop.  The first call per (device, n_fft) builds a twiddle table and
 * synchronises the stream once. */
int use_stft_fwd(const float* wav, void* Y, int B, int L, int n_fft, int hop, const float* window, int Tpad, float factor,
                 float exponent, use_stream_t s);
int use_istft_back(const void* X, float* wav, int B, int L, int n_fft, int hop, const float* window, int Tpad, float factor,
                   float exponent, use_stream_t s);

/* Counters of a handle: "graph_captures" (segments of the sampling loop captured so far), "plans_built", "plan_cache_hits",
 * "plans_parked", "plan_stale" (1: use_set_option was called since use_plan - the evaluation entry points will refuse the plan).  A handle keeps the plans - workspace, state, time-embedding tables, captured graphs - of the most recently used
 * (B, T') shapes (use_set_option("plan_cache", k), default 4 besides the current one): use_plan of a parked shape costs nothing and
 * its graphs replay as they are. */
int use_get_stat(use_handle* h, const char* name, long long* value);
/* Introspection for tests / profiling */
/* One eager score evaluation with a HIP-event pair around every launch of the dominant kernel (conv_v4_kernel, the
 * wide-tile implicit-GEMM 3x3 convolution of the large feature maps): summed kernel time, their algorithmic FLOPs and
 * algorithmic HBM bytes (every operand once), launch count, and the wall time of the whole evaluation on `stream`.
 * Synchronises the stream. */
int use_profile_score(use_handle* h, const void* x, const void* y, const float* t, void* out, use_stream_t stream,
                      double* conv_ms, double* conv_flops, double* conv_bytes, int* conv_launches, double* total_ms);
/* The HBM-bound kernels of the same evaluation (FIR x2 resampling of the activation maps, pyramid-head convolutions, the input
 * convolution), one record per launch in launch order: kernel class ("fir_up", "fir_down", "pyr_conv", "conv_in"), the map it read
 * (H x W per item of the sub-batch), its algorithmic HBM bytes (every operand once) and its duration between HIP events.  Returns 0,
 * 1 when `index` is past the last record of the most recent use_profile_score, negative on error. */
int use_profile_aux(use_handle* h, int index, char* name, int name_cap, int* H, int* W, double* bytes, double* ms);
/* ... the same list also carries the MFMA-bound kernels beside the dominant one ("conv_v2": 3x3 convolutions of the middle maps,
 * "conv_sk": the smallest maps): their algorithmic FLOPs (0 for the HBM-bound classes). */
int use_profile_aux_flops(use_handle* h, int index, double* flops);
/* Single-convolution harness for kernel bring-up and same-box A/B timing (no reference counterpart): builds one fused
 * implicit-GEMM 3x3 convolution (optional second channel-concat source C1, GroupNorm affine + SiLU on the input, bias +
 * time-embedding bias, fused 1x1 shortcut over XC0+XC1 channels, residual, GroupNorm partial sums) on deterministic
 * pseudo-random device data, runs it `iters` times through the chosen kernel (variant 0: the library's dispatcher, 2 / 4 / 5:
 * conv_v2 / conv_v4 / conv_v5) and returns the average launch time, the algorithmic FLOPs, the output as float32
 * [B][H][W][Cout] (host, may be null) and the per-(item, channel) (sum, sum of squares) totals [B][Cout][2] (host, may be null). */
typedef struct use_conv_case {
    int B, H, W, C0, C1, Cout, XC0, XC1;
    int act, gn, temb, res, stats;   /* flags */
    int dtype;                       /* 0 fp32, 1 bf16, 2 fp16 (storage) */
    int variant, iters;
} use_conv_case;
int use_conv_bench(const use_conv_case* c, float* out_host, float* stats_host, double* ms_avg, double* flops);
/* ---- single operators of the path on caller-owned device tensors (NHWC: [B][H][W][C], C contiguous) with host fp32 weights.
 * Test / bring-up surface: the per-operator golden vectors of the reference reach the HIP kernels through these.
 * use_op_conv: out = ((conv3x3|1x1(act(a x + b)) + conv1x1(x0|x1) + bias + temb[b]) + res) * out_scale, exactly the fused operator of the
 *   res-block (ResnetBlockBigGANpp, layerspp.py:282-314); channel counts are multiples of 32 (zero-pad smaller ones); `w` is the
 *   reference's [Cout][Cin][3][3] (ntaps 9) or [Cout][Cin] (ntaps 1) tensor, `w2` [Cout][XC]; coef = GroupNorm folded to (a, b) per
 *   (item, channel) or null; stats (zeroed by the caller) receives the fixed-point GroupNorm totals of the output.  variant: 0 = the
 *   library's dispatcher, 1 generic, 2 conv_v2, 4 conv_v4, 7 conv_sk.  Synchronises the stream.
 * use_op_fir: upsample_2d / downsample_2d with the [1,3,3,1] kernel (up_or_down_sampling.py:202-264); out_act = FIR(act(a x + b)),
 *   out_raw = FIR(x) (either may be null).
 * use_op_attention: softmax(q k^T / sqrt(C)) v per item (AttnBlockpp core, layerspp.py:84-88), q/k/v/out [B][N][C].
 * use_op_gn_finalize: GroupNorm totals (as accumulated in `stats`) of up to two concatenated sources -> coef[b][c] = (a, b). */
typedef struct use_conv_op {
    int B, H, W, C0, C1, Cout, XC0, XC1, ntaps, act, dtype, out_dtype, variant;
    const void *src0, *src1;
    const float* coef;
    const float* w;
    const float* bias;
    const float* temb;
    const void *x0, *x1;
    const float* w2;
    const void* res;
    float out_scale;
    void* out;
    long long* stats;
} use_conv_op;
int use_op_conv(const use_conv_op* c, use_stream_t stream);
/* use_op_conv_dev: the same operator for the training path (reference SGMSE_module.py:46-54 -> model_wrapper.py:147-208, where the
 *   parameters are nn.Parameters on the device and change every optimiser step): c->w / c->bias are DEVICE fp32 tensors, laid out for
 *   the kernels by a kernel on `stream` into the caller's workspace (use_op_conv_dev_workspace(c) bytes); no allocation, no
 *   synchronisation, no fused shortcut (XC0 = XC1 = 0).  w_mode: 0 = conv weight [Cout][Cin][taps]; 1 = data gradient of the conv whose
 *   weight is c->w [Cin][Cout][taps] (the kernel sees w'[co][ci][tap] = w[ci][co][taps-1-tap]); 2 = NIN matrix [Cin][Cout] (ntaps 1). */
size_t use_op_conv_dev_workspace(const use_conv_op* c);
int use_op_conv_dev(const use_conv_op* c, int w_mode, void* work, size_t work_bytes, use_stream_t stream);
int use_op_fir(const void* src, int dtype, const float* coef, int act, void* out_act, void* out_raw, int B, int H, int W, int C, int up,
               use_stream_t stream);
int use_op_attention(const void* q, const void* k, const void* v, void* out, int dtype, int B, int N, int C, use_stream_t stream);
int use_op_gn_finalize(const long long* st0, int C0, const long long* st1, int C1, const float* gamma, const float* beta, int groups,
                       int hw, float eps, float* coef, int B, use_stream_t stream);
/* ---- backward operators of one res-block (minimum slice of ScoreModel.train_step's gradients, reference
 * model_wrapper.py:147-208 / SGMSE_module.py:46-54).  NHWC device tensors; activations and their gradients in `dtype` (0 fp32 /
 * 1 bf16 / 2 fp16 storage: mixed-precision training keeps fp32 parameters, statistics and parameter gradients) where a dtype
 * argument is present, fp32 elsewhere.  The data gradient of a convolution is use_op_conv
 * itself on the flipped, transposed weights; these are the rest:
 * use_op_wgrad:      dW[co][ci][tap] = alpha * sum dY[b,p,co] X[b,p+tap,ci] (reference weight layout), db[co] = alpha * sum dY (or null).
 *                    work: use_op_wgrad_workspace(...) floats of scratch for the tiled kernel (per-slice partial tiles, summed without
 *                    atomics), or null for the small-tile kernel with atomic accumulation.
 * use_op_gn_act_bwd: gradient of act(GroupNorm(groups, eps)(x)) (act: 0 none, 1 SiLU) against dy: dx (+ add_scale * add when add is
 *                    given), dgamma, dbeta; `work`: use_op_gn_workspace(B, C, groups) floats of scratch, 8-byte aligned; have_stats != 0:
 *                    `work` is the workspace use_op_gn_act_fwd ran in for the same x, whose statistics are reused.
 * use_op_gn_act_fwd: y = act(GroupNorm(x)) (the operand of the following convolution's weight gradient, recomputed); same workspace.
 * use_op_colsum:     out[b][c] = scale * sum_p x[b,p,c] (fp32 out; the gradient reaching Dense_0's output); work: 128 B C floats of
 *                    scratch (8-byte aligned) for the pixel-sliced kernel, or null (fp32 input only: one workgroup per 64 channels).
 * use_op_dense_bwd:  Dense_0(SiLU(temb)): g [B][Cout] -> dW [Cout][K], db [Cout], dtemb [B][K].
 * use_op_attention_bwd: the AttnBlockpp core, out = softmax(q k^T / sqrt(C)) v: dq, dk, dv from dO ([B][N][C] each; work: 2 B N N floats). */
size_t use_op_wgrad_workspace(int B, int H, int W, int Cout, int Cin, int ntaps, int dtype);
int use_op_wgrad(const void* dy, const void* x, int dtype, float* dw, float* db, int B, int H, int W, int Cout, int Cin, int ntaps, float alpha,
                 float* work, size_t work_floats, use_stream_t stream);
size_t use_op_gn_workspace(int B, int C, int groups);
int use_op_gn_act_bwd(const void* x, const void* dy, int dtype, const float* gamma, const float* beta, int groups, float eps, int act, const void* add,
                      float add_scale, int B, int HW, int C, float* work, int have_stats, void* dx, float* dgamma, float* dbeta, use_stream_t stream);
int use_op_gn_act_fwd(const void* x, int dtype, const float* gamma, const float* beta, int groups, float eps, int act, int B, int HW, int C, float* work,
                      void* y, use_stream_t stream);
int use_op_colsum(const void* x, int dtype, int B, int HW, int C, float scale, float* out, float* work, use_stream_t stream);
int use_op_dense_bwd(const float* g, const float* temb, const float* Wd, int B, int K, int Cout, float* dW, float* db, float* dtemb,
                     use_stream_t stream);
int use_op_attention_bwd(const float* q, const float* k, const float* v, const float* dO, float* work, float* dq, float* dk, float* dv, int B, int N,
                         int C, use_stream_t stream);
/* ---- wire formats either side of the path, host functions: no device, no handle ----
 * use_wav_read: RIFF/WAVE (PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE) -> interleaved float64 frames scaled like
 *   libsndfile's sf.read (integer PCM / 2^(bits-1)); *samples is malloc'ed, release it with use_free.
 * use_resample_fft: scipy.signal.resample(x, num) == librosa.resample(res_type="fft") for real x (loadwav_dataset.py:95-98).
 * use_load_utterance: the reference's inference loader for one file (loadwav_dataset.py:90-120): first channel, FFT resampling
 *   to target_rate (0: keep), x / max|x| * 0.8 when normalize, float64 throughout, float32 out (malloc'ed; use_free).
 * use_wav_write: sf.write(path, x, rate) of SGMSE_module.py:80 (USE_WAV_PCM16 = soundfile's default WAV subtype) or 32-bit float. */
enum { USE_WAV_PCM16 = 0, USE_WAV_FLOAT32 = 1 };
int use_wav_read(const char* path, double** samples, int64_t* frames, int* channels, int* sample_rate);
int use_wav_write(const char* path, const float* samples, int64_t frames, int channels, int sample_rate, int subtype);
int use_resample_fft(const double* x, int64_t n, int64_t num, double* y);
int use_load_utterance(const char* path, int target_rate, int normalize, float** wav, int64_t* length, int* sample_rate);
void use_free(void* p);

/* timesteps of the sampler, torch.linspace(1, t_eps, N) float32 semantics (sampling/__init__.py:63); host only */
int use_timesteps(int N, float t_eps, float* out);
int use_debug_tensor(use_handle* h, const char* name, void** dev_ptr, int* dims4, int* dtype);
double use_flops_per_score(use_handle* h);   /* 2*MAC of the conv/linear layers for the current plan */

#ifdef __cplusplus
}
#endif
#endif
