// Launch wrappers for the gfx950 kernels of the SGMSE sampling path (internal C++ interface between
// use_engine.cpp and use_kernels.hip).  All tensors are NHWC: [B][H=freq][W=frame][C], C contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <thread>

namespace use {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a property of (kernel, DEVICE): one flag per device, not per process - a second
// handle on another device of the same process (use_create(cfg, device, ...)) would otherwise launch its > 64 KB-LDS kernels unprepared
constexpr int USE_MAX_DEVICES = 64;
struct LdsAttrOnce {
    std::atomic<int> st[USE_MAX_DEVICES] = {};               // per device: 0 not applied, 1 being applied, 2 applied
    // f() applies the attribute(s); it has RUN when once() returns, on whichever host thread got there first (the round-5 form
    // raised its flag before the attributes were set, so a second thread could launch a > 64 KB-LDS kernel unprepared)
    template <typename F> void once(F&& f) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= USE_MAX_DEVICES) { f(); return; }
        if (st[d].load(std::memory_order_acquire) == 2) return;
        int expected = 0;
        if (st[d].compare_exchange_strong(expected, 1, std::memory_order_acq_rel)) { f(); st[d].store(2, std::memory_order_release); }
        else while (st[d].load(std::memory_order_acquire) != 2) std::this_thread::yield();
    }
    template <typename K> void operator()(K kern, int bytes) {
        once([&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    }
};


enum DType { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }

constexpr int CONV_IN_SPLIT_K = 112;   // K of conv_in_split_kernel: 3 blocks (hi*hi, hi*lo, lo*hi) x 9 taps x 4 channels, padded to 7 x 16
constexpr int TILE_H = 8;    // conv output tile: 8 x 16 pixels = 128 GEMM rows
constexpr int TILE_W = 16;

inline int tiles_per_image(int H, int W) { return ((H + TILE_H - 1) / TILE_H) * ((W + TILE_W - 1) / TILE_W); }
extern int g_conv_in_wgs;                               // conv_in_split_kernel: most workgroups per item (use_set_option("conv_in_wgs"))
inline int conv_in_split_tpw(int H, int W) { const int nt = tiles_per_image(H, W); return (nt + g_conv_in_wgs - 1) / g_conv_in_wgs; }
inline int conv_in_split_wgs(int H, int W) { const int nt = tiles_per_image(H, W), tpw = conv_in_split_tpw(H, W); return (nt + tpw - 1) / tpw; }
inline int conv_v2_tiles(int H, int W) { return ((H + 15) / 16) * ((W + 15) / 16); }   // conv_v2_kernel: 16x16 tiles

// Implicit-GEMM convolution (3x3 pad 1, or 1x1) with fused prologue / epilogue.
//   in   = concat(src0[C0], src1[C1]) along channels, optional per-(b, channel) affine
//          (GroupNorm folded to a*x+b) followed by optional SiLU, zero outside the image
//   out  = ((conv(in) + conv1x1(x) + bias + temb[b]) + res) * out_scale + (w4 . pyr + b4)
//   stats[b][cout][2] += per-channel (sum, sum of squares) of this workgroup's stored values (fixed point, integer atomics)
struct ConvArgs {
    const void* src0; const void* src1; int C0; int C1; int in_dtype;
    const float* coef;      // [B][C0+C1][2] (a, b) or null
    // ... or the GroupNorm is finalised by the consumer itself (no separate launch): per-(item, channel) totals of the two
    // concatenated sources as accumulated by their producers (see `stats`), the layer's affine and group count
    const long long* gn_st0; const long long* gn_st1;       // [B][C0][2], [B][C1][2] fixed-point (sum, sum of squares); null: use coef
    const float* gn_gamma; const float* gn_beta; int gn_groups; float gn_inv_n; float gn_eps;   // gn_inv_n = 1 / (channels per group * H * W)
    int act;                // 0: none, 1: SiLU (after the affine)
    const void* w;          // packed [CoutPad][ntaps][C0+C1] in in_dtype
    const void* wb;         // optional slab-major copy [ntaps][(C0+C1)/ck][CoutPad][ck], ck = conv_v4_chunk(in_dtype)
                            // (one (tap, chunk) slab contiguous: conv_v4), or null.  Rows are 64 bytes; the 16-byte
                            // piece q of row n is stored at position q ^ ((n >> 2) & 3) (bank-conflict-free LDS image for
                            // a lane-linear copy: conv_v4 stores it as it is)
    int cout_pad;
    // optional second K segment: + conv1x1(concat(x0[XC0], x1[XC1])) with weights w2 [CoutPad][1][XC0+XC1]
    // (the res-block shortcut Conv_2 fused into Conv_1; raw input, no affine / activation)
    const void* x0; const void* x1; int XC0; int XC1; const void* w2;
    const void* w2b;        // slab-major copy of w2: [(XC0+XC1)/ck][CoutPad][ck], or null
    const float* bias;      // [Cout] or null
    const float* temb;      // [B or 1][temb_stride] slice start for this conv, or null
    int temb_bstride;       // elements between batch rows (0: shared by the batch)
    const void* res;        // residual, out_dtype, [B,H,W,Cout] or null
    float out_scale;
    const float* pyr;       // fp32 [B,H,W,4] or null  (Combine 'sum' fused: + conv1x1_{4->Cout}(pyr))
    const float* w4;        // [Cout][4]
    const float* b4;        // [Cout]
    void* out; int out_dtype;
    long long* stats_part;  // or null: [B][conv_v4_tiles][Cout][2] per-workgroup partial totals in the format of `stats`, written with plain stores
                            // INSTEAD of the atomics below (conv_v4 on the large maps: 640 workgroups per item finishing in step queue
                            // 64 deep on each of an item's 256 totals - measured 3.3 % of a launch); gn_finalize sums them (launch_gn_finalize)
    long long* stats;       // or null: [B][Cout][2] fixed-point totals (sum * 2^20, sum of squares * 2^20) of the stored values,
                            // accumulated with 64-bit integer atomics (order-independent, hence deterministic); zeroed by the caller
    int B, H, W, Cout, ntaps;
    int dbg;                // ablation bits for kernel bring-up (0 in production)
    unsigned long long* trace;   // optional: s_memtime stamps of workgroup 0 (kernel bring-up), else null
};
void launch_conv(const ConvArgs& a, hipStream_t s);
int conv_stats_parts(const ConvArgs& a);                // partial totals per (item, channel) this launch would write (ConvArgs::stats_part), or 0
// software-pipelined variant for large maps (use_conv_v2.hip); launch_conv dispatches to it when eligible
bool conv_v2_eligible(const ConvArgs& a);
void launch_conv_v2(const ConvArgs& a, hipStream_t s);
// wide-tile variant (use_conv_v4.hip): 16x32-pixel tiles, K chunks of conv_v4_chunk() channels, slab-major weights
inline int conv_v4_chunk(int dtype) { return dtype == DT_F32 ? 16 : 32; }
inline int conv_v4_tiles(int H, int W) { return ((H + 15) / 16) * ((W + 31) / 32); }
// split-K variant for the small maps (use_conv_sk.hip): 64-pixel x 32/64-channel tiles, the 8 waves of a workgroup split K
bool conv_sk_eligible(const ConvArgs& a);
void conv_sk_set_max_px(long n);                         // largest map (H*W) it is used for (default 16x20)
void launch_conv_sk(const ConvArgs& a, hipStream_t s);
void launch_conv_generic(const ConvArgs& a, hipStream_t s);   // conv_kernel / pyr_conv_kernel / conv_in_kernel only (no specialised schedule)
void pyr_conv_set_ws(int n);                            // wave-specialised form of the same layer (0: off, 1: on, n > 1: workgroups per launch)
bool conv_v4_eligible(const ConvArgs& a);
void conv_v4_set_min_blocks(long n);                     // smallest grid conv_v4 is used for (default 80 workgroups per image)
void launch_conv_v4(const ConvArgs& a, hipStream_t s);
// the same kernel on v_mfma_f32_16x16x32 (use_conv_v5.hip; 16-bit storage): less energy per FLOP than the 32x32x16 shape
void conv_v5_set(int on);                                // use_set_option("conv_v5", 0 / 1), default 1
bool conv_v5_enabled(const ConvArgs& a);                 // for a launch conv_v4_eligible() accepted
void launch_conv_v5(const ConvArgs& a, hipStream_t s);
// GroupNorm finalisation for the consumers that take a coefficient array (FIR resampling kernels): per-(b, group) mean / rstd
// from the per-channel totals of up to two concatenated sources, folded with gamma/beta into coef[b][c] = (a, b): y = a*x + b.
// Source i is given either as totals st_i [B][C_i][2] or (pt_i != null) as nt_i per-workgroup partial totals pt_i [B][nt_i][C_i][2]
// (ConvArgs::stats_part): integer sums, so both forms give bit-identical coefficients.
void launch_gn_finalize(const long long* st0, int C0, const long long* st1, int C1, const float* gamma, const float* beta,
                        int groups, int hw, float eps, float* coef, int B, hipStream_t s, const long long* pt0 = nullptr, int nt0 = 0,
                        const long long* pt1 = nullptr, int nt1 = 0);

// FIR x2 resampling with the separable [1,3,3,1] kernel (upfirdn2d semantics of the reference).
// out_act (nullable) = FIR(act(a*x+b)), out_raw (nullable) = FIR(x).
void launch_fir_up2(const void* src, int dtype, const float* coef, int act, void* out_act, void* out_raw, int B,
                    int H, int W, int C, hipStream_t s);
void fir_set_strip(int on);                             // down-sampler as a strip walk (default on)
void launch_fir_down2(const void* src, int dtype, const float* coef, int act, void* out_act, void* out_raw, int B,
                      int H, int W, int C, hipStream_t s);

// out[r][t] = z |pre z|^(power-1) pre post for t < Tin (z = in[r][t]), 0 for Tin <= t < Tout  (spectrogram compression glue)
void launch_spec_map(const float2* in, float2* out, long rows, int Tin, int Tin_stride, int Tout, float pre, float power,
                     float post, hipStream_t s);

// Device STFT / iSTFT fused with the compression glue (n_fft N even, F = N/2 + 1 bins, centred frames, reflect padding).
// tw = table of (cos, sin)(2 pi m / N), m < N (launch_twiddle_table); win = analysis / synthesis window [N] (device).
void launch_twiddle_table(float2* tw, int N, hipStream_t s);
void launch_stft_fwd(const float* wav, const float* win, const float2* tw, float2* Y, int B, int L, int N, int hop, int T,
                     int Tpad, float factor, float expo, hipStream_t s);
void launch_istft_back(const float2* X, const float* win, const float2* tw, float* wav, int B, int L, int N, int hop, int Tpad,
                       float factor, float expo, hipStream_t s);

// x4[b,f,t,:] = 2*(x.re, x.im, y.re, y.im) - 1   (fp32), x/y complex64 [B,F,T]; y2 != null: 8 channels per pixel,
// 2*(x, y, y2) - 1 and two zero channels (the 6-channel input of condition="both")
void launch_pack_input(const float2* x, const float2* y, const float2* y2, float* x4, long npix, hipStream_t s);
// Combine 'sum' of an 8-channel (6 + 2 padding) input pyramid as its own pass (reference layerspp.py:50-55): in place
// h[b,p,:] += b8 + W8 . pyr[b,p,:], and the GroupNorm totals of the stored values into stats (zeroed by the caller)
void launch_combine_add(void* h, int dtype, const float* pyr, const float* w8, const float* b8, long long* stats, int B, long pix_per_b,
                        int C, hipStream_t s);

// Time embedding: t[B] -> silu(Linear2(silu(Linear1(fourier(log t)))))  [B][4nf]
void launch_temb_mlp(const float* t, int t_stride, const float* gfp_w, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* out_silu, int B, int nf, hipStream_t s);
// All res-block Dense_0 layers at once: out[b][r] = bias[r] + dot(W[r], silu_temb[b])
void launch_temb_dense(const float* silu_temb, const float* W, const float* bias, float* out, int B, int rows,
                       int dim, hipStream_t s);

// Bottleneck self-attention core: h[b,i,:] = sum_j softmax_j(q_i.k_j / sqrt(C)) v[b,j,:]
void launch_attention(const void* q, const void* k, const void* v, void* out, int dtype, int B, int N, int C,
                      hipStream_t s);

// The whole bottleneck attention block (GroupNorm -> q, k, v NIN -> softmax(q k^T / sqrt(C)) v -> NIN_3 -> (x + h) / sqrt(2)) in one
// launch, every contraction on the matrix pipe (use_attn.hip): C = 256, N <= 96 tokens, 16-bit storage.  NIN weights [Cout][Cin].
struct AttnArgs {
    const void* x; void* out; int N;
    const long long* gn_st; const float* gn_gamma; const float* gn_beta; int gn_groups; float gn_eps;
    const void *wq, *wk, *wv, *wo; const float *bq, *bk, *bv, *bo;
    long long* stats;        // GroupNorm totals of `out` (zeroed by the caller) or null
};
bool attn_fused_eligible(int dtype, int N, int C);
void launch_attn_fused(const AttnArgs& a, int dtype, int B, hipStream_t s);

// helpers of the long-sequence attention path (two implicit GEMMs on conv_kernel, see Fwd::attention)
void launch_softmax_rows(void* x, int dtype, long rows, int cols, hipStream_t s);            // in place, over the last axis
void launch_transpose_nc(const void* in, void* out, int dtype, int B, int N, int C, hipStream_t s);   // [B][N][C] -> [B][C][N]

// net = conv1x1_{4->2}(pyr / t[b]) (t == null: no division); out = sign * net  (complex64 out; sign -1: the score)
void launch_score_out(const float* pyr, int pc, const float* t, int t_stride, const float* w, const float* bias,
                      float2* score, int B, long pix_per_b, float sign, hipStream_t s);     // pc = 4 or 8 pyramid channels

// ---- backward kernels of one res-block (use_bwd.hip; fp32 storage, NHWC): the gradient half of train_step, minimum slice ----
// work = nullptr: the 32 x 32-tile kernel with atomic slices; else wgrad_workspace_floats() floats of scratch for the tiled kernel
size_t wgrad_workspace_floats(int B, int H, int W, int Cout, int Cin, int ntaps, int dtype);
void wgrad_set_mfma16(int v);                 // 16-bit tensors: 1 (default) 16-bit MFMA kernel, 0 fp32 MFMA on converted operands
void wgrad_set_blocks(int n);                 // target workgroup count of the tiled weight-gradient kernel (tiles x pixel slices)
// dy / x in `dtype` (fp32 / bf16 / fp16 storage; converted while staging, exact-fp32 MFMA); false: case not served (16-bit without workspace)
bool launch_wgrad(const void* dy, const void* x, int dtype, float* dw, float* db, int B, int H, int W, int Cout, int Cin, int ntaps, float alpha,
                  float* work, hipStream_t s);                            // dW [Cout][Cin][ntaps] (reference layout), db [Cout] or null
// GroupNorm(+SiLU) with a tape: x / dy / y / dx / add in `dtype`, statistics and affine parameters fp32.
// part (fp64 scratch of gn_workspace_floats) = nullptr: the one-block-per-(item, group) forms (fp32 only); false: case not served
bool launch_gn_stats(const void* x, int dtype, int B, int HW, int C, int G, float eps, float* mean, float* rstd, double* part, hipStream_t s);
constexpr int GN_MAX_SLICES = 256;            // pixel slices per item of the sliced GroupNorm reductions (fp64 partials per slice)
size_t gn_workspace_floats(int B, int C, int G);
// dx = d/dx of act(GroupNorm(x)) against dy, + add_scale * add (or add == null); s1 / s2: [B][C] scratch; dgamma / dbeta [C]
bool launch_gn_act_bwd(const void* x, const void* dy, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                       const void* add, float add_scale, int B, int HW, int C, int G, float* s1, float* s2, float* m12, double* part, void* dx,
                       float* dgamma, float* dbeta, hipStream_t s);
bool launch_gn_act_fwd(const void* x, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int B, int HW,
                       int C, int G, void* y, hipStream_t s);
bool launch_colsum(const void* x, int dtype, int B, int HW, int C, float scale, float* out, double* part, hipStream_t s);       // out[b][c] = scale * sum_p x[b,p,c]
// attention core backward (q, k, v, dO, dq, dk, dv: [B][N][C] fp32; work: 2 B N N floats)
void launch_attention_bwd(const float* q, const float* k, const float* v, const float* dO, float* work, float* dq, float* dk, float* dv, int B,
                          int N, int C, hipStream_t s);
void launch_dense_bwd(const float* g, const float* temb, const float* Wd, int B, int K, int Cout, float* dW, float* db, float* dtemb, hipStream_t s);
// fp32 parameter tensor in HBM -> the kernels' weight layouts (plain, and slab-major when dstb != nullptr) + the zero-padded bias
void launch_pack_conv_dev(const float* src, int mode, int cout, int cin, int ntaps, int cout_pad, int dtype, int ck, void* dst, void* dstb,
                          const float* bias, float* bias_out, hipStream_t s);

// ---- SDE updates (complex64 as float2, fp32 arithmetic) ----
struct RngRef { const unsigned long long* state; unsigned draw; };  // state[0]=seed, state[1]=draw base
// z source: noise != null -> read noise[i]; else Philox(seed, draw)
void launch_prior(const float2* y, const float2* noise, RngRef rng, float std1, float2* x, long n, hipStream_t s);
// reverse-diffusion / Euler-Maruyama predictor: x_mean = x + c_drift*(y-x) + c_score*score ; x' = x_mean + c_noise*z
void launch_predictor(const float2* x, const float2* y, const float2* score, const float2* noise, RngRef rng,
                      float c_drift, float c_score, float c_noise, float2* x_out, float2* x_mean, long n,
                      hipStream_t s);
// Langevin norms: partial[b][blk] = (sum|g|^2, sum|z|^2)
void launch_langevin_norms(const float2* score, const float2* noise, RngRef rng, float* partial, int B,
                           long n_per_b, int blocks_per_b, hipStream_t s);
// step[0] = 2*(snr * mean_b||z_b|| / mean_b||g_b||)^2
void launch_langevin_step(const float* partial, int B, int blocks_per_b, float snr, float* step, hipStream_t s);
// x_mean = x + eps*g ; x' = x_mean + sqrt(2 eps) z ; eps = step_dev[0] if step_dev else step_host
void launch_corrector(const float2* x, const float2* score, const float2* noise, RngRef rng, const float* step_dev,
                      float step_host, float2* x_out, float2* x_mean, long n, hipStream_t s);
void launch_fill_noise(float2* out, RngRef rng, long n, hipStream_t s);

}  // namespace use
