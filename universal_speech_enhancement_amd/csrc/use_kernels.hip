// gfx950 (CDNA4 / MI355X) kernels for the SGMSE reverse-SDE sampling path.
//
// What each kernel replaces in the reference (paths under src/models/components/sgmse/):
//   conv_kernel        nn.Conv2d 3x3 / 1x1 (backbones/ncsnpp_utils/layers.py:113-162) with the GroupNorm-apply +
//                      SiLU of its input (layerspp.py:283,304), bias, Dense_0(temb) bias (layerspp.py:302-303),
//                      residual + 1/sqrt(2) (layerspp.py:311-314), Combine 'sum' (layerspp.py:50-55) and the
//                      GroupNorm statistics of its output all fused in
//   gn_finalize_kernel nn.GroupNorm statistics (min(C//4,32) groups, eps 1e-6), folded to a per-channel affine
//   fir_*_kernel       upsample_2d / downsample_2d (up_or_down_sampling.py:202-264; CUDA: op/upfirdn2d_kernel.cu)
//   attention_kernel   AttnBlockpp einsum / softmax / einsum (layerspp.py:84-88)
//   temb_*             GaussianFourierProjection + 2 Linear (layerspp.py:37-39, ncsnpp.py:352-368), Dense_0
//   score_out / prior / predictor / langevin_* / corrector
//                      ncsnpp.py:492-500, model_wrapper.py:137, sdes.py:248-254, sampling/predictors.py:40-68,
//                      sampling/correctors.py:45-98
//
// Layout: activations are NHWC ([B][freq][frame][C]); a conv is an implicit GEMM with M = pixels (8x16 tiles),
// N = output channels, K = taps x input channels, computed with 32x32 MFMA tiles (bf16: v_mfma_f32_32x32x16_bf16,
// fp32 parity mode: v_mfma_f32_32x32x2_f32), fp32 accumulation.  The input halo tile (10x18 pixels x CK channels)
// is staged in LDS once per channel chunk and re-used by the 9 taps; weights stream through a double-buffered
// LDS slab.  Wave = 64 lanes everywhere.
#include "use_kernels.h"
#include <type_traits>
#include <cstdlib>
#include <cstdio>
#include "use_device.h"

#include <math.h>

namespace use {

// Workgroup barrier that orders LDS only (round 5).  __syncthreads() carries a release fence, i.e. s_waitcnt vmcnt(0): loads in flight for
// the NEXT tile / unit and the acknowledgement of the stores just issued would be waited for at every barrier of a tile walk (pyr_conv_ws:
// 3 us per unit whatever else the unit did).  For barriers that only separate LDS writes from LDS reads of the same workgroup.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")


// ---------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution
// ---------------------------------------------------------------------------------------------------------
constexpr int BM = TILE_H * TILE_W;     // 128 pixels
constexpr int HALO_MAX = (TILE_H + 2) * (TILE_W + 2);

template <typename TIN, typename TOUT, int CK, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs p) {
    typedef Mfma<TIN> MF;
    constexpr int VEC = 16 / sizeof(TIN);
    constexpr int PARTS = CK / VEC;                       // 16-byte pieces per pixel row of a chunk
    constexpr int ROWB = CK * (int)sizeof(TIN) + 16;      // padded LDS row (bytes)
    constexpr int MW = BM / WM, NW = BN / WN;             // wave tile
    constexpr int TM = MW / 32, TN = NW / 32;
    constexpr int KSTEPS = CK / MF::KM;
    constexpr int WPT = (BN * PARTS + 255) / 256;         // 16-byte weight pieces per thread per (tap, chunk)
    constexpr bool WGUARD = (BN * PARTS) % 256 != 0;
    constexpr bool ACC = sizeof(TIN) == 4;                // fp32 parity mode: accurate SiLU
    static_assert(WM * WN == 4 && MW % 32 == 0 && NW % 32 == 0 && CK % MF::KM == 0 && PARTS >= 1, "tiling");

    // one LDS object, carved: [halo tile][2 weight slabs]; the epilogue re-uses it as [4 waves x staging][stats]
    constexpr int HALO_BYTES = HALO_MAX * ROWB, W_BYTES = BN * ROWB;
    constexpr int STG_LD = NW + 4;                                   // fp32 staging row (padded)
    constexpr int STG_WAVE = 32 * STG_LD * 4;                        // one 32-pixel x NW tile per wave
    constexpr int EPI_BYTES = 4 * STG_WAVE + WM * BN * 2 * 4;
    constexpr int MAIN_BYTES = HALO_BYTES + 2 * W_BYTES;
    constexpr int SMEM_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
    char* const s_halo = smem;
    char* const s_w0 = smem + HALO_BYTES;
    __shared__ float2 coef_s[1024];                          // GroupNorm affine of this item's input channels (gn_fill_table)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int b = blockIdx.z;
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
    const int ty0 = (blockIdx.x / tiles_x) * TILE_H, tx0 = (blockIdx.x % tiles_x) * TILE_W;
    const int n0 = blockIdx.y * BN;
    const int pad = (p.ntaps == 9) ? 1 : 0;
    const int HW_ = TILE_W + 2 * pad, HH_ = TILE_H + 2 * pad;
    const int Ctot = p.C0 + p.C1;
    const int nchunks = Ctot / CK;
    const int nit1 = nchunks * p.ntaps;                   // segment 0: taps x chunks of the (normalised) input
    const int XCtot = p.XC0 + p.XC1;
    const int nit = nit1 + XCtot / CK;                    // segment 1: 1x1 shortcut over the raw block input

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (p.coef || p.gn_st0) {                                // GroupNorm finalised here (no separate launch), or copied
        gn_fill_table(coef_s, p, b, Ctot, tid, 256);
        __syncthreads();
    }

    // per-lane fragment bases (bytes)
    int a_base[TM], b_base[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int m = wm * MW + i * 32 + (lane & 31);
        a_base[i] = ((m >> 4) * HW_ + (m & 15)) * ROWB + (lane >> 5) * MF::KPL * (int)sizeof(TIN);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
        b_base[j] = (wn * NW + j * 32 + (lane & 31)) * ROWB + (lane >> 5) * MF::KPL * (int)sizeof(TIN);

    // stage the halo tile of the channel chunk that iteration `it` starts
    auto stage_halo = [&](int it) {
        const bool seg1 = it >= nit1;
        const int c_glob = (seg1 ? it - nit1 : it / p.ntaps) * CK;
        const TIN* src; int Cs, c_loc;
        if (!seg1) {
            if (c_glob < p.C0) { src = (const TIN*)p.src0; Cs = p.C0; c_loc = c_glob; }
            else               { src = (const TIN*)p.src1; Cs = p.C1; c_loc = c_glob - p.C0; }
        } else {
            if (c_glob < p.XC0) { src = (const TIN*)p.x0; Cs = p.XC0; c_loc = c_glob; }
            else                { src = (const TIN*)p.x1; Cs = p.XC1; c_loc = c_glob - p.XC0; }
        }
        const bool use_coef = (p.coef || p.gn_st0) && !seg1;
        const bool use_act = p.act && !seg1;
        const int part = tid % PARTS;                      // constant per thread (256 % PARTS == 0)
        float ca[VEC], cb[VEC];
        if (use_coef) {
            const float2* cf = coef_s + c_glob + part * VEC;
#pragma unroll
            for (int k = 0; k < VEC; ++k) { const float2 v = cf[k]; ca[k] = v.x; cb[k] = v.y; }
        }
        const int npix = HW_ * HH_;
        for (int idx = tid; idx < npix * PARTS; idx += 256) {
            const int pix = idx / PARTS;
            const int hy = pix / HW_, hx = pix - hy * HW_;
            const int gy = ty0 + hy - pad, gx = tx0 + hx - pad;
            float v[VEC];
            if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                Vec16<TIN>::load(src + ((size_t)(b * p.H + gy) * p.W + gx) * Cs + c_loc + part * VEC, v);
                if (use_coef) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[k] = fmaf(v[k], ca[k], cb[k]);
                }
                if (use_act) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[k] = silu_f<ACC>(v[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[k] = 0.f;
            }
            *reinterpret_cast<uint4*>(s_halo + pix * ROWB + part * 16) = Vec16<TIN>::pack(v);
        }
    };

    static_assert(WPT <= 4, "weight staging registers");
    uint4 wr0 = make_uint4(0, 0, 0, 0), wr1 = wr0, wr2 = wr0, wr3 = wr0;   // named (an array gets demoted to LDS)
#define USE_LOAD_Q(Q, DST)                                                                                     \
    if (WPT > (Q)) {                                                                                           \
        const int idx = tid + (Q)*256;                                                                         \
        if (!WGUARD || idx < BN * PARTS)                                                                       \
            DST = *reinterpret_cast<const uint4*>(wb_ + (size_t)(idx / PARTS) * wrs_ + (idx % PARTS) * VEC);   \
    }
#define USE_LOAD_W(IT)                                                                                         \
    {                                                                                                          \
        const bool seg1_ = (IT) >= nit1;                                                                       \
        const int chunk_ = seg1_ ? (IT)-nit1 : (IT) / p.ntaps, tap_ = seg1_ ? 0 : (IT)-chunk_ * p.ntaps;       \
        const int wld_ = seg1_ ? XCtot : Ctot;                                                                 \
        const int wrs_ = seg1_ ? wld_ : p.ntaps * wld_;            /* row (cout) stride in elements */                  \
        const TIN* wb_ = (seg1_ ? (const TIN*)p.w2 : (const TIN*)p.w) + (size_t)n0 * wrs_ + (size_t)tap_ * wld_ + chunk_ * CK; \
        USE_LOAD_Q(0, wr0) USE_LOAD_Q(1, wr1) USE_LOAD_Q(2, wr2) USE_LOAD_Q(3, wr3)                            \
    }
#define USE_STORE_Q(Q, SRC, BUF)                                                                               \
    if (WPT > (Q)) {                                                                                           \
        const int idx = tid + (Q)*256;                                                                         \
        if (!WGUARD || idx < BN * PARTS)                                                                       \
            *reinterpret_cast<uint4*>(s_w0 + (BUF)*W_BYTES + (idx / PARTS) * ROWB + (idx % PARTS) * 16) = SRC;             \
    }
#define USE_STORE_W(BUF)                                                                                       \
    { USE_STORE_Q(0, wr0, BUF) USE_STORE_Q(1, wr1, BUF) USE_STORE_Q(2, wr2, BUF) USE_STORE_Q(3, wr3, BUF) }

    stage_halo(0);
    USE_LOAD_W(0);
    USE_STORE_W(0);
    __syncthreads();

    for (int it = 0; it < nit; ++it) {
        const bool seg1 = it >= nit1;
        const int tap = seg1 ? 4 : it % p.ntaps;          // the shortcut reads the centre tap of the halo tile
        const bool has_next = (it + 1 < nit);
        if (has_next) USE_LOAD_W(it + 1);
        const int dy = pad ? tap / 3 : 0, dx = pad ? tap - dy * 3 : 0;
        const char* ha = s_halo + (dy * HW_ + dx) * ROWB;
        const char* wb = s_w0 + (it & 1) * W_BYTES;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            typename MF::frag af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = MF::ld(ha + a_base[i] + kk * MF::KM * (int)sizeof(TIN));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = MF::ld(wb + b_base[j] + kk * MF::KM * (int)sizeof(TIN));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(af[i], bf[j], acc[i][j]);
        }
        if (has_next) {
            USE_STORE_W((it + 1) & 1);
            if (it + 1 >= nit1 || (it + 1) % p.ntaps == 0) {   // next iteration starts a new channel chunk
                __syncthreads();
                stage_halo(it + 1);
            }
        }
        __syncthreads();
    }

    // ------------------------------ epilogue ------------------------------
    // Each wave transposes its accumulators through LDS (fp32, one 32-pixel x NW slab at a time) so that every lane
    // then owns 16 contiguous output bytes of one pixel: residual loads and output stores are full 16-byte accesses
    // (8 lanes = one pixel's 128-byte row segment) instead of 2-byte accesses scattered over two pixels.
    constexpr int CH = 16 / (int)sizeof(TOUT);               // output channels per 16-byte chunk
    constexpr int CPR = NW / CH;                             // chunks per staging row
    constexpr int QN = 32 * CPR / 64;                        // chunks per lane per slab
    static_assert(64 % CPR == 0 && QN >= 1, "epilogue chunking");
    float* const stg = reinterpret_cast<float*>(smem + wave * STG_WAVE);
    float* const red = reinterpret_cast<float*>(smem + 4 * STG_WAVE);     // [WM][BN][2]
    TOUT* out = (TOUT*)p.out;
    const TOUT* res = (const TOUT*)p.res;
    float addv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn * NW + j * 32 + (lane & 31);
        float add = 0.f;
        if (co < p.Cout) {
            if (p.bias) add += p.bias[co];
            if (p.temb) add += p.temb[(size_t)b * p.temb_bstride + co];
        }
        addv[j] = add;
    }
    const int ch = lane % CPR;                               // this lane's chunk column (constant over q)
    const int co0 = n0 + wn * NW + ch * CH;
    const bool cok = co0 < p.Cout;                           // Cout is a multiple of CH
    float st_s[CH], st_q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { st_s[c] = 0.f; st_q[c] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                stg[row * STG_LD + j * 32 + (lane & 31)] = acc[i][j][r] + addv[j];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int row = (q * 64 + lane) / CPR;
            const int m = wm * MW + i * 32 + row;
            const int gy = ty0 + (m >> 4), gx = tx0 + (m & 15);
            float v[CH];
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) {
                const float4 t4 = *reinterpret_cast<const float4*>(stg + row * STG_LD + ch * CH + c4 * 4);
                v[c4 * 4] = t4.x; v[c4 * 4 + 1] = t4.y; v[c4 * 4 + 2] = t4.z; v[c4 * 4 + 3] = t4.w;
            }
            if (cok && gy < p.H && gx < p.W) {
                const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
                if (res) {
                    float rv[CH];
                    Vec16<TOUT>::load(res + pix * p.Cout + co0, rv);
#pragma unroll
                    for (int c = 0; c < CH; ++c) v[c] += rv[c];
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] *= p.out_scale;
                if (p.pyr) {
                    const float4 pq = *reinterpret_cast<const float4*>(p.pyr + pix * 4);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const float4 wq = *reinterpret_cast<const float4*>(p.w4 + (size_t)(co0 + c) * 4);
                        v[c] += p.b4[co0 + c] + wq.x * pq.x + wq.y * pq.y + wq.z * pq.z + wq.w * pq.w;
                    }
                }
                const uint4 packed = Vec16<TOUT>::pack(v);
                *reinterpret_cast<uint4*>(out + pix * p.Cout + co0) = packed;
                if (p.stats) {
                    float vr[CH];
                    Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&packed), vr);   // statistics of the stored values
#pragma unroll
                    for (int c = 0; c < CH; ++c) { st_s[c] += vr[c]; st_q[c] += vr[c] * vr[c]; }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (p.stats) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { st_s[c] = reduce_lanes_stride<CPR>(st_s[c]); st_q[c] = reduce_lanes_stride<CPR>(st_q[c]); }
        if (lane < CPR) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int cl = wn * NW + ch * CH + c;
                red[(wm * BN + cl) * 2] = st_s[c]; red[(wm * BN + cl) * 2 + 1] = st_q[c];
            }
        }
        __syncthreads();
        if (tid < BN) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < p.Cout) gn_accumulate(p.stats + ((size_t)b * p.Cout + co) * 2, s, q);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// pyr_conv_kernel: the output-pyramid convolutions (GroupNorm + SiLU -> 3x3 conv to 4 channels, + the up-sampled
// pyramid so far; reference ncsnpp.py:437-470).  N = 4 is too narrow for conv_kernel's tiling (18 barrier-separated
// steps of 4 MFMAs); the layer is bound by reading its input once.  One workgroup per 8x16-pixel tile: the normalised,
// activated halo of 128 input channels (51 KB) and the 4 weight rows of all 9 taps are staged once per 128-channel block,
// then each wave runs 72 MFMAs (32 pixels x [4 of 32] channels x K = 9 x 128) with no barrier in between.
// ---------------------------------------------------------------------------------------------------------
#ifndef USE_PYR_CB
#define USE_PYR_CB 128
#endif
constexpr int PYR_CB = USE_PYR_CB;                           // input channels per staged block
constexpr int PYR_PP = PYR_CB / 8;                           // 16-byte pieces per pixel of a block
constexpr int PYR_ROWB = PYR_CB * 2 + 16;                    // 272 (144): pixel row pitch in LDS
constexpr int PYR_HPITCH = ((TILE_W + 2) * PYR_ROWB + 255) / 256 * 256;      // halo row pitch: multiple of 256 B (see conv_v2)
constexpr int PYR_HALO = (TILE_H + 2) * PYR_HPITCH;
constexpr int PYR_WB = 9 * 4 * PYR_ROWB;
constexpr int PYR_SMEM = PYR_HALO + PYR_WB;
template <typename T16>
__global__ __launch_bounds__(256) void pyr_conv_kernel(ConvArgs p) {
    typedef Mfma<T16> MF;
    extern __shared__ __attribute__((aligned(16))) char psm[];
    char* const s_halo = psm;
    char* const s_w = psm + PYR_HALO;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
    const int ty0 = (blockIdx.x / tiles_x) * TILE_H, tx0 = (blockIdx.x % tiles_x) * TILE_W;
    const int Cin = p.C0;
    const T16* src = (const T16*)p.src0;
    const int part = tid % PYR_PP;                           // this thread's 8 channels of every block
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int m = lane & 31;
    const int a_base = (wave * 2 + (m >> 4)) * PYR_HPITCH + (m & 15) * PYR_ROWB + (lane >> 5) * 16;
    const int b_base = (lane & 3) * PYR_ROWB + (lane >> 5) * 16;           // output channels >= 4 are never stored
    constexpr int NP = ((TILE_H + 2) * (TILE_W + 2) * PYR_PP + 255) / 256;  // halo pieces per thread
    for (int c0 = 0; c0 < Cin; c0 += PYR_CB) {
        if (c0) __syncthreads();                             // every wave is done with the previous block
        float ca[8], cb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ca[k] = 1.f; cb[k] = 0.f; }
        if (p.gn_st0) {                                      // GroupNorm finalised here: this thread's 8 channels of the block
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2 v = gn_coef_of(p.gn_st0, p.C0, p.gn_st1, p.C1, p.gn_gamma, p.gn_beta, p.gn_groups, p.gn_inv_n, p.gn_eps, b, c0 + part * 8 + k);
                ca[k] = v.x; cb[k] = v.y;
            }
        } else if (p.coef) {
            const float* cf = p.coef + ((size_t)b * Cin + c0 + part * 8) * 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) { ca[k] = cf[2 * k]; cb[k] = cf[2 * k + 1]; }
        }
        uint4 raw[NP]; int dst[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int idx = j * 256 + tid, pix = idx / PYR_PP;
            const int hy = pix / (TILE_W + 2), hx = pix - hy * (TILE_W + 2);
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool have = pix < (TILE_H + 2) * (TILE_W + 2);
            const bool inb = have && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            dst[j] = have ? (inb ? 1 : 2) * 0x100000 + hy * PYR_HPITCH + hx * PYR_ROWB + part * 16 : 0;
            raw[j] = inb ? *reinterpret_cast<const uint4*>(src + ((size_t)(b * p.H + gy) * p.W + gx) * Cin + c0 + part * 8)
                         : make_uint4(0, 0, 0, 0);
        }
        for (int i = tid; i < 9 * 4 * PYR_PP; i += 256) {    // weight rows 0..3: packed [CoutPad][9][Cin]
            const int pc = i % PYR_PP, row = i / PYR_PP;     // row = tap * 4 + co
            const int tap = row >> 2, co = row & 3;
            *reinterpret_cast<uint4*>(s_w + row * PYR_ROWB + pc * 16) =
                *reinterpret_cast<const uint4*>((const T16*)p.w + ((size_t)co * 9 + tap) * Cin + c0 + pc * 8);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (!dst[j]) continue;
            uint4 o = make_uint4(0, 0, 0, 0);                // outside the image: the conv's zero padding
            if (dst[j] < 0x200000) {
                float v[8];
                Vec16<T16>::load(reinterpret_cast<const T16*>(&raw[j]), v);
#pragma unroll
                for (int k = 0; k < 8; ++k) { v[k] = fmaf(v[k], ca[k], cb[k]); if (p.act) v[k] = silu_f<false>(v[k]); }
                o = Vec16<T16>::pack(v);
            }
            *reinterpret_cast<uint4*>(s_halo + (dst[j] & 0xfffff)) = o;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const char* ha = s_halo + a_base + (tap / 3) * PYR_HPITCH + (tap % 3) * PYR_ROWB;
            const char* wb = s_w + b_base + tap * 4 * PYR_ROWB;
#pragma unroll
            for (int kk = 0; kk < PYR_CB / 16; ++kk) acc = MF::mma(MF::ld(ha + kk * 32), MF::ld(wb + kk * 32), acc);
        }
    }
    __syncthreads();
    float* const s_out = reinterpret_cast<float*>(psm);      // [128 pixels][4]
    if ((lane & 31) < 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            s_out[(wave * 32 + row) * 4 + (lane & 31)] = acc[r];
        }
    }
    __syncthreads();
    if (tid < TILE_H * TILE_W) {
        const int gy = ty0 + (tid >> 4), gx = tx0 + (tid & 15);
        if (gy < p.H && gx < p.W) {
            const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
            float4 v = *reinterpret_cast<const float4*>(s_out + tid * 4);
            float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < p.Cout && p.bias) o[c] += p.bias[c];
            if (p.res) {
                const float* rp = (const float*)p.res + pix * p.Cout;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < p.Cout) o[c] += rp[c];
            }
            float* op = (float*)p.out + pix * p.Cout;
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < p.Cout) op[c] = o[c] * p.out_scale;
        }
    }
}
// 16x16x32 MFMA for the head: the 32x32x16 form computes 32 output columns of which 4 exist and chains all 36 MFMAs of a (tile, block)
// unit through ONE accumulator (the matrix pipe's dependent-issue latency, not its rate, set the unit's duration); the 16x16x32 form has
// a quarter of the passes per instruction and gives each wave two independent accumulators (its two 16-pixel rows).
template <typename T> struct Mfma16;
template <> struct Mfma16<__bf16> {
    DEVI static f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<_Float16> {
    DEVI static f32x4 mma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
#ifndef USE_PYR_ABL
#define USE_PYR_ABL 0                /* timing-only ablations of pyr_conv_ws_kernel: 1 no SiLU, 2 one MFMA pair per unit */
#endif
// pyr_conv_ws_kernel (round 5): the same layer with its two halves on different waves.  The round-4 form (pyr_conv_pipe_kernel: a 4-wave
// workgroup walks tiles with the next halo's loads in flight, two workgroups per CU) had parts that ADD UP - loads + LDS + stores 81 us,
// SiLU 67 us (the VALU rate of v_exp + v_rcp on a 1.4x halo), MFMAs + fragment reads 44 us of 191 us: a workgroup transforms,
// synchronises, multiplies, and the two workgroups of a CU run in step.  Here ONE workgroup of eight waves per CU: waves 4-7 (one per
// SIMD) load, normalise, activate and stage the halo of unit u + 1 into the second of two LDS halo buffers while waves 0-3 (one per
// SIMD) run the MFMAs and the stores of unit u - VALU and matrix pipe of a SIMD busy at the same time, one LDS-only barrier per unit;
// the launch is one round of <= 256 workgroups.  With one 128-channel block per tile the walk runs DOWN a 16-pixel column strip and the
// halo rolls (rows 0, 1 of a tile = rows 8, 9 of the tile above, copied LDS -> LDS: 1.125x instead of 1.41x redundant loads and SiLUs);
// the consumers' residual is requested one tile ahead; every global access of the walk is an unconditional buffer access (counted
// vmcnt waits that leave stores and prefetches in flight).  163 -> 91 us per 3-item launch at 512x640 (profiles/r5_pyr_conv_ws.txt).
// Measured and left out: three register sets (loads two units ahead), fragments requested a tap row ahead with each activation
// fragment read once (bursts of 24 ds_reads: 107 us), sched_group_barrier patterns (degenerate to read - wait - MFMA).
// Same arithmetic, same order per output as pyr_conv_kernel: bit-identical results (test_pyramid_head_forms_agree_bit_for_bit_and_match_torch).
constexpr int PYRW_SMEM = 2 * PYR_HALO + 2 * PYR_WB + 4 * 32 * 4 * 4;
template <typename T16, bool ONEBLK, bool ACT>
__global__ __launch_bounds__(512) void pyr_conv_ws_kernel(ConvArgs p, int tiles_per_wg) {
    typedef Mfma<T16> MF;
    extern __shared__ __attribute__((aligned(16))) char psm[];
    char* const s_w = psm + 2 * PYR_HALO;                                        // [nblk <= 2][9 taps x 4 rows][PYR_ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W, ntiles = tiles_x * ((p.H + TILE_H - 1) / TILE_H);
    const int Cin = p.C0, nblk = ONEBLK ? 1 : Cin / PYR_CB;                      // 1 or 2
    for (int i = tid; i < nblk * 9 * 4 * PYR_PP; i += 512) {                     // weights of every block, once
        const int blk = i / (9 * 4 * PYR_PP), r0 = i - blk * (9 * 4 * PYR_PP);
        const int pc = r0 % PYR_PP, row = r0 / PYR_PP;                           // row = tap * 4 + co
        const int tap = row >> 2, co = row & 3;
        *reinterpret_cast<uint4*>(s_w + blk * PYR_WB + row * PYR_ROWB + pc * 16) =
            *reinterpret_cast<const uint4*>((const T16*)p.w + ((size_t)co * 9 + tap) * Cin + blk * PYR_CB + pc * 8);
    }
    const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(ntiles, t_begin + tiles_per_wg);
    const int nunits = (t_end - t_begin) * nblk;                                 // unit = (tile, 128-channel block)
    if (nunits <= 0) return;
    constexpr int PYRW_SETS = 2;                                                 // register sets of the producers: the loads of unit u + 1 fly behind the transform of unit u
    const int nunits_pad = (nunits + PYRW_SETS - 1) / PYRW_SETS * PYRW_SETS;      // barriers per walk, both sides
    // walk order: down the 16-pixel column strips when the halo rolls (ONEBLK), row by row otherwise
    const int tiles_y = (p.H + TILE_H - 1) / TILE_H;
    auto tile_ty = [&](int tile) -> int { return ONEBLK ? tile % tiles_y : tile / tiles_x; };
    auto tile_tx = [&](int tile) -> int { return ONEBLK ? tile / tiles_y : tile % tiles_x; };
    if (wave >= 4) {
        // ---------------- producers: halo of unit u -> LDS buffer u & 1 (branch-free staging as in the pipelined form) ----------------
        const int ptid = tid - 256;
        const int part = ptid % PYR_PP;
        const T16* src = (const T16*)p.src0;
        // ROLL (one 128-channel block per tile): tiles are walked DOWN a 16-pixel column strip and the halo rolls - rows 0, 1 of a tile's halo are
        // rows 8, 9 of the tile above, copied LDS -> LDS from the other buffer; only the 8 new rows are loaded, normalised and activated
        // (1.41x -> 1.125x redundant loads and SiLUs).  The first tile of a walk / of a strip loads its two top rows itself.
        constexpr bool ROLL = ONEBLK;
        constexpr int HY0 = ROLL ? 2 : 0;                                        // first halo row of the prefetched body
        constexpr int NPX = (TILE_H + 2 - HY0) * (TILE_W + 2);                   // body pixels: 144 (9 pieces per thread, none spare) / 180
        constexpr int NP = (NPX * PYR_PP + 255) / 256;
        float ca[2][8], cb[2][8];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ca[blk][k] = 1.f; cb[blk][k] = 0.f;
                if (blk < nblk) {
                    const int c = blk * PYR_CB + part * 8 + k;
                    if (p.gn_st0) {
                        const float2 v = gn_coef_of(p.gn_st0, p.C0, p.gn_st1, p.C1, p.gn_gamma, p.gn_beta, p.gn_groups, p.gn_inv_n, p.gn_eps, b, c);
                        ca[blk][k] = v.x; cb[blk][k] = v.y;
                    } else if (p.coef) {
                        ca[blk][k] = p.coef[((size_t)b * Cin + c) * 2]; cb[blk][k] = p.coef[((size_t)b * Cin + c) * 2 + 1];
                    }
                }
            }
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<T16*>(src) + (size_t)b * p.H * p.W * Cin, 0,
                                                                                 (unsigned)((size_t)p.H * p.W * Cin * 2), 0x00020000);
        int rel[NP], dbase[NP]; unsigned hyx[NP];
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) {
            const int idx = jj * 256 + ptid, pix = idx / PYR_PP;
            const int hy = HY0 + pix / (TILE_W + 2), hx = pix % (TILE_W + 2);
            const bool have = pix < NPX;
            rel[jj] = (((hy - 1) * p.W + (hx - 1)) * Cin + part * 8) * 2;
            dbase[jj] = have ? hy * PYR_HPITCH + hx * PYR_ROWB + part * 16 : (TILE_W + 2) * PYR_ROWB + (ptid & 7) * 16;   // (spare bytes behind a halo row)
            hyx[jj] = have ? (unsigned)(hy << 8 | hx) : 0xff00u;
        }
        auto prefetch = [&](int u, uint4 (&raw)[NP], unsigned& ok) {
            u = min(u, nunits - 1);                                              // (past the end: the last unit again - unconditional loads keep the counted waits)
            const int tile = t_begin + u / nblk, c0 = (u % nblk) * PYR_CB;
            const int ty0 = tile_ty(tile) * TILE_H, tx0 = tile_tx(tile) * TILE_W;
            const int base = ((ty0 * p.W + tx0) * Cin + c0) * 2;
            ok = 0u;
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                const int gy = ty0 + (int)(hyx[jj] >> 8) - 1, gx = tx0 + (int)(hyx[jj] & 255u) - 1;
                const bool inb = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                ok |= inb ? 1u << jj : 0u;
                raw[jj] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, inb ? (unsigned)(rel[jj] + base) : 0xfffffff0u, 0, 0));
            }
        };
        // GroupNorm affine + SiLU of one 16-byte piece, two channels per instruction: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are the same
        // IEEE operations as the scalar forms of silu_f (the producers are bound by the VALU rate: 5.5 full-rate + 2 quarter-rate
        // instructions per element before, 4 + 2 now); `mk` = 0 outside the image: the conv's zero padding
        auto xform = [&](const uint4 rawp, const int blk, const unsigned mk) -> uint4 {
            float v[8];
            Vec16<T16>::load(reinterpret_cast<const T16*>(&rawp), v);
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                f32x2 t = {v[k], v[k + 1]};
                const f32x2 a2 = {ONEBLK ? ca[0][k] : (blk ? ca[1][k] : ca[0][k]), ONEBLK ? ca[0][k + 1] : (blk ? ca[1][k + 1] : ca[0][k + 1])};
                const f32x2 b2 = {ONEBLK ? cb[0][k] : (blk ? cb[1][k] : cb[0][k]), ONEBLK ? cb[0][k + 1] : (blk ? cb[1][k + 1] : cb[0][k + 1])};
                t = __builtin_elementwise_fma(t, a2, b2);
                if (ACT && !(USE_PYR_ABL & 1)) {
                    f32x2 e = t * (f32x2){-1.44269504088896341f, -1.44269504088896341f};
                    e.x = __builtin_amdgcn_exp2f(e.x); e.y = __builtin_amdgcn_exp2f(e.y);
                    e = e + (f32x2){1.0f, 1.0f};
                    e.x = __builtin_amdgcn_rcpf(e.x); e.y = __builtin_amdgcn_rcpf(e.y);
                    t = t * e;
                }
                v[k] = t.x; v[k + 1] = t.y;
            }
            uint4 o = Vec16<T16>::pack(v);
            o.x &= mk; o.y &= mk; o.z &= mk; o.w &= mk;
            return o;
        };
        auto stage = [&](int u, const uint4 (&raw)[NP], const unsigned ok) {
            const int blk = ONEBLK ? 0 : u % nblk;
            char* const hb = psm + (u & 1) * PYR_HALO;
            if (ROLL) {                                                          // halo rows 0, 1: 36 px x 16 pieces, <= 3 per thread
                const int uc = min(u, nunits - 1), tile = t_begin + uc;
                const int ty = tile_ty(tile), tx = tile_tx(tile);
                const bool cont = uc > 0 && ty != 0;                             // directly below the previous unit's tile (uniform)
                const char* const hprev = psm + ((u & 1) ^ 1) * PYR_HALO;
#pragma unroll
                for (int k3 = 0; k3 < 3; ++k3) {
                    const int idx = k3 * 256 + ptid, pix = idx / PYR_PP;         // pix < 36 for the pieces that exist
                    const int hy = pix / (TILE_W + 2), hx = pix % (TILE_W + 2);
                    const int d = hy * PYR_HPITCH + hx * PYR_ROWB + part * 16;
                    if (pix < 2 * (TILE_W + 2)) {
                        uint4 o;
                        if (cont) o = *reinterpret_cast<const uint4*>(hprev + d + TILE_H * PYR_HPITCH);
                        else {
                            const int gy = ty * TILE_H + hy - 1, gx = tx * TILE_W + hx - 1;
                            const bool inb = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
                            const uint4 r = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                rs_in, inb ? (unsigned)(((gy * p.W + gx) * Cin + part * 8) * 2) : 0xfffffff0u, 0, 0));
                            o = xform(r, 0, inb ? 0xffffffffu : 0u);
                        }
                        *reinterpret_cast<uint4*>(hb + d) = o;
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < NP; ++jj)
                *reinterpret_cast<uint4*>(hb + dbase[jj]) = xform(raw[jj], blk, ((ok >> jj) & 1u) ? 0xffffffffu : 0u);
        };
        constexpr int NSETS = PYRW_SETS;
        uint4 raw[NSETS][NP]; unsigned ok[NSETS];
#pragma unroll
        for (int q = 0; q < NSETS - 1; ++q) prefetch(q, raw[q], ok[q]);
        // (the walk is padded to a multiple of NSETS units - the padding units re-stage the last unit into a buffer nobody reads any more -
        // so that the body has no conditional: with `if (u + q < nunits)` around a unit hipcc copies register sets between the branches and
        // waits for all loads before it issues the next ones)
        for (int u = 0; u < nunits_pad; u += NSETS) {
#pragma unroll
            for (int q = 0; q < NSETS; ++q) {
                prefetch(u + q + NSETS - 1, raw[(q + NSETS - 1) % NSETS], ok[(q + NSETS - 1) % NSETS]);
                __builtin_amdgcn_sched_barrier(0);                               // (the loads first: hipcc hoists the unpacking of the staged set above them, behind a vmcnt(0))
                stage(u + q, raw[q], ok[q]);
                LDS_BARRIER();                                                 // unit u + q staged; the consumers are done with the unit before it
            }
        }
    } else {
        // ---------------- consumers: 72 MFMAs per unit and wave, the tile's [128 px][4] leave through a per-wave LDS transposition ----------------
        float* const s_out = reinterpret_cast<float*>(psm + 2 * PYR_HALO + 2 * PYR_WB) + wave * 128;    // [32 px][4]
        // 16x16x32 fragments: lane -> (pixel column | output channel) lane & 15, 8-channel k group lane >> 4; row i of the wave: + i * PYR_HPITCH
        const int a_base = (wave * 2) * PYR_HPITCH + (lane & 15) * PYR_ROWB + (lane >> 4) * 16;
        const int b_base = (lane & 3) * PYR_ROWB + (lane >> 4) * 16;
        f32x4 acc0, acc1;                                                        // the wave's two 16-pixel rows
        // Output pixel of this lane in a tile (lanes 0-31: tile row 2 wave + (lane >> 4), column lane & 15).  Its residual (the incoming
        // pyramid, fp32 [px][Cout]) is requested ONE TILE AHEAD and every global access of the walk is an unconditional dword buffer
        // access (no pixel / no channel: an out-of-range offset), so the wait for it is a counted vmcnt that leaves the previous tile's
        // stores in flight - fetched inside its own unit, its memory latency was the unit's duration (3 us with everything else ablated).
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((const float*)p.res) + (size_t)b * p.H * p.W * p.Cout, 0,
                                                                                  p.res ? (unsigned)((size_t)p.H * p.W * p.Cout * 4) : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((float*)p.out + (size_t)b * p.H * p.W * p.Cout, 0,
                                                                                (unsigned)((size_t)p.H * p.W * p.Cout * 4), 0x00020000);
        auto px_off = [&](int tile) -> unsigned {                                // byte offset of channel 0 of this lane's pixel, or out of range
            const int ty0 = tile_ty(tile) * TILE_H, tx0 = tile_tx(tile) * TILE_W;
            const int gy = ty0 + wave * 2 + ((lane >> 4) & 1), gx = tx0 + (lane & 15);
            return lane < 32 && gy < p.H && gx < p.W ? (unsigned)((gy * p.W + gx) * p.Cout) * 4u : 0xfffffff0u;
        };
        auto res_ld = [&](int tile, float (&r)[4]) {
            const unsigned o = px_off(min(tile, t_end - 1));
#pragma unroll
            for (int c = 0; c < 4; ++c) r[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, c < p.Cout ? o + 4u * c : 0xfffffff0u, 0, 0));
        };
        float bias4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bias4[c] = (c < p.Cout && p.bias) ? p.bias[c] : 0.f;
        float rese[4], resn[4];
        res_ld(t_begin, resn);
        for (int u = 0; u < nunits_pad; ++u) {
            if (u >= nunits) { LDS_BARRIER(); continue; }                     // padding units of the producers' walk
            const int blk = ONEBLK ? 0 : u % nblk;
            if (blk == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            }
            const int tile = t_begin + u / nblk;
            if (blk == nblk - 1) {                                               // the tile completes in this unit: its residual landed a unit ago
#pragma unroll
                for (int c = 0; c < 4; ++c) rese[c] = resn[c];
                res_ld(tile + 1, resn);
            }
            LDS_BARRIER();                                                     // unit u is in buffer u & 1
            const char* const hb = psm + (u & 1) * PYR_HALO + a_base;
            const char* const wb = s_w + blk * PYR_WB + b_base;
#pragma unroll
            for (int tap = 0; tap < ((USE_PYR_ABL & 2) ? 1 : 9); ++tap) {
                const char* ha = hb + (tap / 3) * PYR_HPITCH + (tap % 3) * PYR_ROWB;
                const char* wt = wb + tap * 4 * PYR_ROWB;
#pragma unroll
                for (int kk = 0; kk < ((USE_PYR_ABL & 2) ? 1 : PYR_CB / 32); ++kk) {
                    const auto wf = MF::ld(wt + kk * 64);
                    acc0 = Mfma16<T16>::mma(MF::ld(ha + kk * 64), wf, acc0);
                    acc1 = Mfma16<T16>::mma(MF::ld(ha + PYR_HPITCH + kk * 64), wf, acc1);
                }
            }
            if (blk == nblk - 1) {
                if ((lane & 15) < 4) {                                           // D of 16x16x32: column lane & 15 (output channel), rows 4 (lane >> 4) + r
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = 4 * (lane >> 4) + r;
                        s_out[col * 4 + (lane & 15)] = acc0[r];
                        s_out[(16 + col) * 4 + (lane & 15)] = acc1[r];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                {
                    const float4 v = *reinterpret_cast<const float4*>(s_out + (lane & 31) * 4);
                    const float o[4] = {v.x, v.y, v.z, v.w};
                    const unsigned off = px_off(tile);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (o[c] + bias4[c] + rese[c]) * p.out_scale), rs_o,
                                                              c < p.Cout ? off + 4u * c : 0xfffffff0u, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}
static bool pyr_conv_eligible(const ConvArgs& a) {
    return a.in_dtype != DT_F32 && a.out_dtype == DT_F32 && a.ntaps == 9 && a.Cout <= 4 && a.C1 == 0 && a.C0 % PYR_CB == 0 &&
           !a.temb && !a.pyr && !a.stats && a.XC0 + a.XC1 == 0;
}

// ---------------------------------------------------------------------------------------------------------
// conv_in_kernel: the network's first convolution (4 fp32 input channels -> Cout, 3x3, reference ncsnpp.py:326-331).
// K = 36 is too shallow for conv_kernel's chunked pipeline (nine 4-deep chunks, a barrier each) and a direct fp32 convolution
// on the VALU is bound by its 6 GFMA per launch (175 us at 512x640, measured 298); the layer should be bound by writing its
// output.  So: fp32 MFMA (v_mfma_f32_32x32x2_f32, the input stays fp32 in every precision mode), one workgroup per 8x16-pixel
// tile x 128 output channels, the 10x18x4 halo and the 128x36 weights staged in LDS once, then each wave runs its 32 pixels x
// 128 channels as 4 x 18 MFMAs with no barrier, and the tile leaves through an LDS transpose as 16-byte stores (256 contiguous
// bytes per pixel).  Output rounding, GroupNorm statistics of the stored values and their layout are those of conv_kernel.
// ---------------------------------------------------------------------------------------------------------
constexpr int CIN_WP = 37;                                   // weight row pitch (floats): odd -> conflict-free over output channels
constexpr int CIN_SP = 136;                                  // staging row pitch (floats) of a 32-pixel x 128-channel wave tile
template <typename TOUT>
__global__ __launch_bounds__(256) void conv_in_kernel(ConvArgs p) {
    constexpr int HALO = (TILE_H + 2) * (TILE_W + 2);
    constexpr int MAIN_BYTES = (HALO * 4 + 128 * CIN_WP) * 4, STG_BYTES = 4 * 32 * CIN_SP * 4;
    __shared__ __attribute__((aligned(16))) char smem[MAIN_BYTES > STG_BYTES ? MAIN_BYTES : STG_BYTES];
    __shared__ float s_red[4 * 128 * 2];
    float* const s_in = reinterpret_cast<float*>(smem);      // [10 x 18][4]
    float* const s_w = s_in + HALO * 4;                      // [128 co][36 (+1)]: k = tap * 4 + ci
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W;
    const int ty0 = (blockIdx.x / tiles_x) * TILE_H, tx0 = (blockIdx.x % tiles_x) * TILE_W;
    const int n0 = blockIdx.y * 128;
    const float* src = (const float*)p.src0;
    for (int i = tid; i < HALO; i += 256) {
        const int hy = i / (TILE_W + 2), hx = i - hy * (TILE_W + 2);
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) v = *reinterpret_cast<const float4*>(src + ((size_t)(b * p.H + gy) * p.W + gx) * 4);
        *reinterpret_cast<float4*>(s_in + i * 4) = v;
    }
    const float* wsrc = (const float*)p.w;                   // [CoutPad][9][4]
    for (int i = tid; i < 128 * 9; i += 256) {
        const int co = i / 9, tap = i - co * 9;
        const float4 w4 = n0 + co < p.cout_pad ? *reinterpret_cast<const float4*>(wsrc + ((size_t)(n0 + co) * 9 + tap) * 4)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = s_w + co * CIN_WP + tap * 4;
        d[0] = w4.x; d[1] = w4.y; d[2] = w4.z; d[3] = w4.w;
    }
    __syncthreads();
    // wave w: tile rows 2w, 2w+1 (32 pixels) x 128 channels.  MFMA k-step s: k = 2s + (lane >> 5) -> tap = s >> 1, ci = 2 (s & 1) + (lane >> 5)
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int m = lane & 31;
    const float* const pa = s_in + ((wave * 2 + (m >> 4)) * (TILE_W + 2) + (m & 15)) * 4 + (lane >> 5);
    const float* const pb = s_w + m * CIN_WP + (lane >> 5);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int tap = s >> 1, dy = tap / 3, dx = tap % 3;
        const float a = pa[(dy * (TILE_W + 2) + dx) * 4 + 2 * (s & 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, pb[j * 32 * CIN_WP + 2 * s], acc[j], 0, 0, 0);
    }
    __syncthreads();                                         // every wave is done with the operands: the region becomes the staging
    float* const stg = reinterpret_cast<float*>(smem) + wave * (32 * CIN_SP);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = n0 + j * 32 + m;
        const float add = (co < p.Cout && p.bias) ? p.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            stg[row * CIN_SP + j * 32 + m] = (acc[j][r] + add) * p.out_scale;
        }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int CH = 16 / (int)sizeof(TOUT);               // output channels per 16-byte chunk
    constexpr int CPR = 128 / CH;                            // chunks per pixel: 16 or 32
    constexpr int PPQ = 64 / CPR;                            // pixels per pass of the wave
    const int ch = lane % CPR, co0 = n0 + ch * CH;
    const bool cok = co0 < p.Cout;                           // Cout is a multiple of 8
    float st_s[CH], st_q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { st_s[c] = 0.f; st_q[c] = 0.f; }
    TOUT* out = (TOUT*)p.out;
#pragma unroll
    for (int q = 0; q < 32 / PPQ; ++q) {
        const int row = q * PPQ + lane / CPR;
        const int gy = ty0 + wave * 2 + (row >> 4), gx = tx0 + (row & 15);
        if (!cok || gy >= p.H || gx >= p.W) continue;
        float v[CH];
#pragma unroll
        for (int c4 = 0; c4 < CH / 4; ++c4) {
            const float4 t4 = *reinterpret_cast<const float4*>(stg + row * CIN_SP + ch * CH + c4 * 4);
            v[c4 * 4] = t4.x; v[c4 * 4 + 1] = t4.y; v[c4 * 4 + 2] = t4.z; v[c4 * 4 + 3] = t4.w;
        }
        const uint4 packed = Vec16<TOUT>::pack(v);
        *reinterpret_cast<uint4*>(out + ((size_t)(b * p.H + gy) * p.W + gx) * p.Cout + co0) = packed;
        float vr[CH];
        Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&packed), vr);       // statistics of the stored values
#pragma unroll
        for (int c = 0; c < CH; ++c) { st_s[c] += vr[c]; st_q[c] += vr[c] * vr[c]; }
    }
    if (p.stats) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { st_s[c] = reduce_lanes_stride<CPR>(st_s[c]); st_q[c] = reduce_lanes_stride<CPR>(st_q[c]); }
        if (lane < CPR) {
#pragma unroll
            for (int c = 0; c < CH; ++c) { s_red[(wave * 128 + ch * CH + c) * 2] = st_s[c]; s_red[(wave * 128 + ch * CH + c) * 2 + 1] = st_q[c]; }
        }
        __syncthreads();
        if (tid < 128) {
            float sm = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { sm += s_red[(w * 128 + tid) * 2]; q += s_red[(w * 128 + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < p.Cout) gn_accumulate(p.stats + ((size_t)b * p.Cout + co) * 2, sm, q);
        }
    }
}
// conv_in_split_kernel: the same layer in the 16-bit storage modes.  The fp32 MFMA runs at the fp32 vector rate (157 TF/s: 77 us of
// arithmetic at 512x640 before anything else), 16x below the bf16 pipe -- and the layer's job is to write 335 MB.  So the fp32
// operands are split into bf16 pairs, x = xh + xl, w = wh + wl (hi = bf16(v), lo = bf16(v - hi)), and
//   x w = xh wh + xh wl + xl wh            (the dropped xl wl term is 2^-16 relative: 250x below the bf16 rounding of the output)
// runs as ONE K = 3 x 36 (+4 zero) = 112 contraction on v_mfma_f32_32x32x16_bf16: 7 MFMAs per 32x32 output tile instead of 18
// fp32 ones at a quarter of the cycles each.  fp32 accumulation; output rounding / GroupNorm totals as in conv_in_kernel.  The
// weight operand comes pre-split from the blob (ConvArgs::wb, pack_conv_in_split); the fp32 parity mode keeps conv_in_kernel.
constexpr int CINS_WP = CONV_IN_SPLIT_K * 2 + 16;            // weight row pitch in LDS (bytes): 240 = 15 x 16, conflict-free b128 reads
constexpr int CINS_XB = (TILE_H + 2) * (TILE_W + 2) * 8;     // one halo array: [180 px][4] bf16
// A workgroup walks `tiles_per_wg` consecutive tiles of one item: the weights are staged once, and the GroupNorm totals leave as ONE
// pair of atomics per channel per workgroup -- with a workgroup per tile the 2560 tiles of a 512x640 map queue 2560 deep on each of
// the item's 256 totals, and that queue, not the arithmetic or the stores, set the kernel's duration (326 us).
// Round 6: the epilogue follows conv_v4's (use_conv_v4.hip): a wave's 32 pixels x 128 channels leave in two halves (its two tile rows), each staged
// by 32 ds_write_addtid_b32 (lane-linear rows, half the LDS-store cycles of ds_write_b32, v4_stage8), read back as eight conflict-free
// ds_read_b128 up front and finished in four straight-line passes (pack, 16-byte store of 4 pixels x 256 B, statistics on channel pairs);
// FULL (whole tiles, whole 128-channel block: every shipped shape) drops the masks.  Statistics of the stored values, as before.  77 -> see profiles/r6_conv_in_addtid.txt.
template <typename TOUT, bool FULL>
__global__ __launch_bounds__(256) void conv_in_split_kernel(ConvArgs p, int tiles_per_wg) {
    static_assert(sizeof(TOUT) == 2, "16-bit storage modes only");
    constexpr int HALO = (TILE_H + 2) * (TILE_W + 2);
    // one block, carved by hand: the staging regions first (ds_write_addtid_b32 takes its base from M0[15:0])
    constexpr int O_STG = 0, O_X = O_STG + 4 * V4_STG_ATID_BYTES, O_W = O_X + 3 * CINS_XB, O_RED = O_W + 128 * CINS_WP, SM_BYTES = O_RED + 4 * 128 * 2 * 4;
    static_assert(O_X % 16 == 0 && O_W % 16 == 0 && O_RED % 16 == 0 && SM_BYTES <= 80 * 1024, "LDS layout (two workgroups per CU)");
    __shared__ __attribute__((aligned(64))) char s_all[SM_BYTES];
    char* const s_x = s_all + O_X;                            // [xh | xl | zeros], each [180][4] bf16
    char* const s_w = s_all + O_W;                            // [128 co][112 (+8)] bf16
    float* const s_red = reinterpret_cast<float*>(s_all + O_RED);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    const int tiles_x = (p.W + TILE_W - 1) / TILE_W, ntiles = tiles_x * ((p.H + TILE_H - 1) / TILE_H);
    const int n0 = blockIdx.y * 128;
    const float* src = (const float*)p.src0;
    {
        const char* wsrc = (const char*)p.wb + (size_t)n0 * (CONV_IN_SPLIT_K * 2);      // [CoutPad][112] bf16, rows of 224 bytes
        for (int i = tid; i < 128 * 14; i += 256) {
            const int row = i / 14, pc = i - row * 14;
            *reinterpret_cast<uint4*>(s_w + row * CINS_WP + pc * 16) = *reinterpret_cast<const uint4*>(wsrc + (size_t)row * (CONV_IN_SPLIT_K * 2) + pc * 16);
        }
        for (int i = tid; i < HALO; i += 256) *reinterpret_cast<uint2*>(s_x + 2 * CINS_XB + i * 8) = make_uint2(0u, 0u);
    }
    const int m = lane & 31, hlf = lane >> 5;
    const char* const pa = s_x + ((wave * 2 + (m >> 4)) * (TILE_W + 2) + (m & 15)) * 8;
    const char* const pb = s_w + m * CINS_WP + hlf * 16;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned stg_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)s_all + (unsigned)(O_STG + wave_u * V4_STG_ATID_BYTES);
    constexpr int CH = 8;                                     // 16-byte chunks of 8 channels; lane -> (pixel lane / 16 of a pass, chunk lane % 16)
    const int ch = lane & 15, p4 = lane >> 4;
    // read-back address (use_device.h, v4_stage8): pass q of a half -> tile column x = 4 q + p4: register r' = p4 + 4 (q >> 1), lane half h = q & 1
    const float* const stg_rd = reinterpret_cast<const float*>(s_all + O_STG + wave * V4_STG_ATID_BYTES) + (ch >> 2) * 512 + p4 * 64 +
                                4 * (((ch >> 2) & 1) + 8 * (ch >> 3)) + (ch & 3) * 8;
    float bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int co = n0 + j * 32 + m; bias[j] = (co < p.Cout && p.bias) ? p.bias[co] : 0.f; }
    f32x2 st_s2[CH / 2], st_q2[CH / 2];
#pragma unroll
    for (int k = 0; k < CH / 2; ++k) { st_s2[k] = (f32x2){0.f, 0.f}; st_q2[k] = (f32x2){0.f, 0.f}; }
    TOUT* out = (TOUT*)p.out;
    const int t_end = min(ntiles, ((int)blockIdx.x + 1) * tiles_per_wg);
    // Round 5: the halo of tile t + 1 (one 16-byte pixel per thread, 180 of 256 threads) is requested right after tile t's halo is in LDS and
    // flies behind tile t's MFMAs and stores; every global access of the loop is an unconditional buffer access (outside the image: an
    // out-of-range offset - loads return 0, stores are dropped), so the wait for the halo is a counted vmcnt that leaves the stores in flight.
    static_assert(HALO <= 256, "one halo pixel per thread");
    const int hy_t = tid / (TILE_W + 2), hx_t = tid - hy_t * (TILE_W + 2);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + (size_t)b * p.H * p.W * 4, 0,
                                                                            (unsigned)((size_t)p.H * p.W * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)b * p.H * p.W * p.Cout, 0,
                                                                            (unsigned)((size_t)p.H * p.W * p.Cout * 2), 0x00020000);
    auto halo_ld = [&](int tile) -> float4 {
        const int ty0 = (tile / tiles_x) * TILE_H, tx0 = (tile % tiles_x) * TILE_W;
        const int gy = ty0 + hy_t - 1, gx = tx0 + hx_t - 1;
        const bool inb = tid < HALO && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, inb ? (unsigned)(gy * p.W + gx) * 16u : 0xfffffff0u, 0, 0));
    };
    const int co0 = n0 + ch * CH;
    const bool cok = FULL || co0 < p.Cout;
    const unsigned voff = cok ? (unsigned)((p4 * p.Cout + co0) * 2) : 0x80000000u;            // lane part of a store offset (beyond Cout: out of range)
    const unsigned pass_b = (unsigned)(4 * p.Cout * 2);                                      // 4 pixels per pass
    const bool scaled = p.out_scale != 1.f;                   // (1 for the network's input convolution)
    float4 hv = halo_ld(min((int)blockIdx.x * tiles_per_wg, ntiles - 1));
    for (int tile = blockIdx.x * tiles_per_wg; tile < t_end; ++tile) {
        const int ty0 = (tile / tiles_x) * TILE_H, tx0 = (tile % tiles_x) * TILE_W;
        if (tid < HALO) {
            const float xs[4] = {hv.x, hv.y, hv.z, hv.w};
            __bf16 hi[4], lo[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { hi[c] = (__bf16)xs[c]; lo[c] = (__bf16)(xs[c] - (float)hi[c]); }
            *reinterpret_cast<uint2*>(s_x + tid * 8) = *reinterpret_cast<const uint2*>(hi);
            *reinterpret_cast<uint2*>(s_x + CINS_XB + tid * 8) = *reinterpret_cast<const uint2*>(lo);
        }
        LDS_BARRIER();                                          // LDS-only barrier: the previous tile's stores stay in flight
        hv = halo_ld(min(tile + 1, ntiles - 1));             // (past the walk's end: a harmless re-load)
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = bias[j];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            // this lane's 8 k of the step = units 4 s + 2 hlf + {0, 1}; unit u -> (block u / 9: xh, xh, xl; u = 27: zeros), tap u % 9
            uint2 a2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int u0 = 4 * s + e, u1 = 4 * s + 2 + e;                      // hlf = 0 / 1 (both compile-time)
                const int o0 = (u0 < 18 ? 0 : u0 < 27 ? 1 : 2) * CINS_XB + (u0 < 27 ? (((u0 % 9) / 3) * (TILE_W + 2) + (u0 % 9) % 3) * 8 : 0);
                const int o1 = (u1 < 18 ? 0 : u1 < 27 ? 1 : 2) * CINS_XB + (u1 < 27 ? (((u1 % 9) / 3) * (TILE_W + 2) + (u1 % 9) % 3) * 8 : 0);
                a2[e] = *reinterpret_cast<const uint2*>(pa + (hlf ? o1 : o0));
            }
            const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(a2[0].x, a2[0].y, a2[1].x, a2[1].y));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<const bf16x8*>(pb + j * 32 * CINS_WP + s * 32), acc[j], 0, 0, 0);
        }
        if (scaled) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] *= p.out_scale;
        }
        // epilogue: the wave's two tile rows (16 pixels x 128 channels each) one after the other
        const unsigned tile_b = (unsigned)((((ty0 + wave_u * 2) * p.W + tx0) * p.Cout) * 2);     // (uniform)
        const unsigned rowp_b = (unsigned)(p.W * p.Cout * 2);
        f32x4 t[4][2];
        auto pin = [&]() {
            asm volatile("" : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0]), "+v"(t[3][1]) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto read_half = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c4 = 0; c4 < 2; ++c4) t[q][c4] = *reinterpret_cast<const f32x4*>(stg_rd + (q >> 1) * 256 + (q & 1) * 32 + c4 * 4);
        };
        auto finish_half = [&](int hb) {
            const int gy = ty0 + wave_u * 2 + hb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = FULL || (cok && gy < p.H && tx0 + q * 4 + p4 < p.W);
                float v[CH] = {t[q][0].x, t[q][0].y, t[q][0].z, t[q][0].w, t[q][1].x, t[q][1].y, t[q][1].z, t[q][1].w};
                const uint4 packed = Vec16<TOUT>::pack(v);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, packed), rs_o,
                                                       ok ? voff + tile_b + (unsigned)hb * rowp_b + (unsigned)q * pass_b : 0x80000000u, 0, 0);
                if (ok) {
                    // statistics of the STORED values (re-expanded): an all-zero item makes this layer's output a per-channel constant away from the
                    // borders, the variance is ~0 and the consumer's rstd ~1 / sqrt(eps) amplifies any difference between the mean and what it normalises
                    // (round 6 tried the fp32 values, as conv_v4 does on its non-degenerate maps: the silent-item test moved from inside to 1 % outside its bound)
                    float vr[CH];
                    Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&packed), vr);
#pragma unroll
                    for (int k = 0; k < CH / 2; ++k) {
                        const f32x2 x = {vr[2 * k], vr[2 * k + 1]};
                        st_s2[k] += x; st_q2[k] = __builtin_elementwise_fma(x, x, st_q2[k]);
                    }
                }
            }
        };
        v4_stage8<0, 0>(stg_lds, acc[0]); v4_stage8<1, 0>(stg_lds, acc[1]); v4_stage8<2, 0>(stg_lds, acc[2]); v4_stage8<3, 0>(stg_lds, acc[3]);
        read_half(); pin();
        v4_stage8<0, 8>(stg_lds, acc[0]); v4_stage8<1, 8>(stg_lds, acc[1]); v4_stage8<2, 8>(stg_lds, acc[2]); v4_stage8<3, 8>(stg_lds, acc[3]);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        finish_half(0);
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
        read_half(); pin();
        finish_half(1);
        LDS_BARRIER();                                          // every wave is done with this tile's halo (LDS-only: no wait for the stores' acknowledgement)
    }
    if (p.stats || p.stats_part) {
        // lanes holding the same channel chunk are 16 apart
#pragma unroll
        for (int k = 0; k < CH / 2; ++k) {
            const float s0 = reduce_lanes_stride<16>(st_s2[k].x), s1 = reduce_lanes_stride<16>(st_s2[k].y);
            const float q0 = reduce_lanes_stride<16>(st_q2[k].x), q1 = reduce_lanes_stride<16>(st_q2[k].y);
            if (lane < 16) {
                s_red[(wave * 128 + ch * CH + 2 * k) * 2] = s0; s_red[(wave * 128 + ch * CH + 2 * k) * 2 + 1] = q0;
                s_red[(wave * 128 + ch * CH + 2 * k + 1) * 2] = s1; s_red[(wave * 128 + ch * CH + 2 * k + 1) * 2 + 1] = q1;
            }
        }
        __syncthreads();
        if (tid < 128) {
            float sm = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { sm += s_red[(w * 128 + tid) * 2]; q += s_red[(w * 128 + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < p.Cout) {
                if (p.stats_part) {                          // this workgroup's partial totals (ConvArgs::stats_part): no queue on the item's 256 totals
                    long long* d = p.stats_part + (((size_t)b * gridDim.x + blockIdx.x) * p.Cout + co) * 2;
                    *reinterpret_cast<longlong2*>(d) = make_longlong2(__float2ll_rn(sm * GN_SUM_SCALE), __float2ll_rn(q * GN_SQ_SCALE));
                } else gn_accumulate(p.stats + ((size_t)b * p.Cout + co) * 2, sm, q);
            }
        }
    }
}
int g_conv_in_wgs = 256;
static bool conv_in_eligible(const ConvArgs& a) {
    return a.in_dtype == DT_F32 && a.C0 == 4 && a.C1 == 0 && a.ntaps == 9 && !a.coef && !a.act && !a.res && !a.pyr && !a.temb &&
           a.XC0 + a.XC1 == 0 && a.Cout % 8 == 0;
}

template <typename TIN, typename TOUT, int CK, int BN, int WM, int WN>
static void conv_launch_t(const ConvArgs& a, hipStream_t s) {
    dim3 grid(tiles_per_image(a.H, a.W), (a.Cout + BN - 1) / BN, a.B);
    hipLaunchKernelGGL((conv_kernel<TIN, TOUT, CK, BN, WM, WN>), grid, dim3(256), 0, s, a);
}

// Workgroups that would each write one partial total per output channel for this launch (ConvArgs::stats_part), 0 if the kernel the
// dispatch below picks only knows the atomic form.  Mirrors launch_conv's order.
int conv_stats_parts(const ConvArgs& a) {
    if (conv_sk_eligible(a)) return 0;
    if (conv_v4_eligible(a)) return conv_v4_tiles(a.H, a.W);
    if (conv_v2_eligible(a)) return 0;
    if (!pyr_conv_eligible(a) && conv_in_eligible(a) && a.wb && a.out_dtype != DT_F32) return conv_in_split_wgs(a.H, a.W);
    return 0;
}
void launch_conv(const ConvArgs& a, hipStream_t s) {
#ifdef USE_HIP_SKIPDBG   // timing-only bring-up build: drop the convolutions of maps with lo <= H*W <= hi pixels (USE_HIP_SKIP="lo:hi")
    {
        static long lo = -1, hi = -1;
        if (lo < 0) { lo = 0; hi = -1; if (const char* e = getenv("USE_HIP_SKIP")) sscanf(e, "%ld:%ld", &lo, &hi); }
        if ((long)a.H * a.W >= lo && (long)a.H * a.W <= hi) return;
    }
#endif
    if (conv_sk_eligible(a)) { launch_conv_sk(a, s); return; }
    if (conv_v4_eligible(a)) { if (conv_v5_enabled(a)) launch_conv_v5(a, s); else launch_conv_v4(a, s); return; }
    if (conv_v2_eligible(a)) { launch_conv_v2(a, s); return; }
    launch_conv_generic(a, s);
}

static int g_pyr_ws = 1;          // wave-specialised head: 0 the one-tile-per-workgroup form, 1 on (256 workgroups per launch), n > 1: n workgroups per launch
void pyr_conv_set_ws(int n) { g_pyr_ws = n; }
void launch_conv_generic(const ConvArgs& a, hipStream_t s) {
    if (pyr_conv_eligible(a)) {
        static LdsAttrOnce attr_set;
        attr_set.once([&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pyr_conv_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, PYR_SMEM);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pyr_conv_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, PYR_SMEM);
        });
        const int ntiles = tiles_per_image(a.H, a.W);
        if (a.C0 <= 2 * PYR_CB && g_pyr_ws) {                // wave-specialised form: one 8-wave workgroup per CU, the launch is one round of workgroups
            static LdsAttrOnce attr3;
            attr3.once([&] {
#define USE_PYRW_ATTR(T, O, A) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pyr_conv_ws_kernel<T, O, A>), hipFuncAttributeMaxDynamicSharedMemorySize, PYRW_SMEM);
                USE_PYRW_ATTR(__bf16, true, true) USE_PYRW_ATTR(__bf16, true, false) USE_PYRW_ATTR(__bf16, false, true) USE_PYRW_ATTR(__bf16, false, false)
                USE_PYRW_ATTR(_Float16, true, true) USE_PYRW_ATTR(_Float16, true, false) USE_PYRW_ATTR(_Float16, false, true) USE_PYRW_ATTR(_Float16, false, false)
#undef USE_PYRW_ATTR
            });
            const int per_item = std::max(1, (g_pyr_ws > 1 ? g_pyr_ws : 256) / std::max(1, a.B));
            const int tpw = (ntiles + per_item - 1) / per_item;
            const dim3 grid((ntiles + tpw - 1) / tpw, 1, a.B);
            const bool one = a.C0 <= PYR_CB, act = a.act != 0;
#define USE_PYRW_GO(T) { if (one && act) hipLaunchKernelGGL((pyr_conv_ws_kernel<T, true, true>), grid, dim3(512), PYRW_SMEM, s, a, tpw);      \
                         else if (one) hipLaunchKernelGGL((pyr_conv_ws_kernel<T, true, false>), grid, dim3(512), PYRW_SMEM, s, a, tpw);         \
                         else if (act) hipLaunchKernelGGL((pyr_conv_ws_kernel<T, false, true>), grid, dim3(512), PYRW_SMEM, s, a, tpw);         \
                         else hipLaunchKernelGGL((pyr_conv_ws_kernel<T, false, false>), grid, dim3(512), PYRW_SMEM, s, a, tpw); }
            if (a.in_dtype == DT_BF16) USE_PYRW_GO(__bf16) else USE_PYRW_GO(_Float16)
#undef USE_PYRW_GO
            return;
        }
        if (a.in_dtype == DT_BF16) hipLaunchKernelGGL(pyr_conv_kernel<__bf16>, dim3(ntiles, 1, a.B), dim3(256), PYR_SMEM, s, a);
        else                       hipLaunchKernelGGL(pyr_conv_kernel<_Float16>, dim3(ntiles, 1, a.B), dim3(256), PYR_SMEM, s, a);
        return;
    }
    if (conv_in_eligible(a)) {
        dim3 grid(tiles_per_image(a.H, a.W), (a.Cout + 127) / 128, a.B);
        const int tpw = conv_in_split_tpw(a.H, a.W);
        dim3 grid_s(conv_in_split_wgs(a.H, a.W), (a.Cout + 127) / 128, a.B);
        const bool full = a.H % TILE_H == 0 && a.W % TILE_W == 0 && a.Cout % 128 == 0;      // whole tiles, whole channel blocks: no masks in the epilogue
#define USE_CINS_GO(T) { if (full) hipLaunchKernelGGL((conv_in_split_kernel<T, true>), grid_s, dim3(256), 0, s, a, tpw); \
                         else hipLaunchKernelGGL((conv_in_split_kernel<T, false>), grid_s, dim3(256), 0, s, a, tpw); }
        if (a.out_dtype == DT_BF16)     { if (a.wb) USE_CINS_GO(__bf16)
                                          else      hipLaunchKernelGGL((conv_in_kernel<__bf16>), grid, dim3(256), 0, s, a); }
        else if (a.out_dtype == DT_F16) { if (a.wb) USE_CINS_GO(_Float16)
                                          else      hipLaunchKernelGGL((conv_in_kernel<_Float16>), grid, dim3(256), 0, s, a); }
#undef USE_CINS_GO
        else                            hipLaunchKernelGGL((conv_in_kernel<float>), grid, dim3(256), 0, s, a);
        return;
    }
    const int Ctot = a.C0 + a.C1;
    const bool small_n = a.Cout <= 32;
    // 16-bit activations: 64-channel chunks, or 32 (nf = 96 networks); a chunk never straddles the two concatenated sources
    const bool ck64 = Ctot % 64 == 0 && (a.XC0 + a.XC1) % 64 == 0 && (a.C1 == 0 || a.C0 % 64 == 0) && (a.XC1 == 0 || a.XC0 % 64 == 0);
    if (a.in_dtype == DT_BF16) {
        if (ck64) {
            if (a.out_dtype == DT_BF16) { small_n ? conv_launch_t<__bf16, __bf16, 64, 32, 4, 1>(a, s) : conv_launch_t<__bf16, __bf16, 64, 128, 2, 2>(a, s); }
            else                        { small_n ? conv_launch_t<__bf16, float, 64, 32, 4, 1>(a, s) : conv_launch_t<__bf16, float, 64, 128, 2, 2>(a, s); }
        } else {
            if (a.out_dtype == DT_BF16) { small_n ? conv_launch_t<__bf16, __bf16, 32, 32, 4, 1>(a, s) : conv_launch_t<__bf16, __bf16, 32, 128, 2, 2>(a, s); }
            else                        { small_n ? conv_launch_t<__bf16, float, 32, 32, 4, 1>(a, s) : conv_launch_t<__bf16, float, 32, 128, 2, 2>(a, s); }
        }
    } else if (a.in_dtype == DT_F16) {
        if (ck64) {
            if (a.out_dtype == DT_F16) { small_n ? conv_launch_t<_Float16, _Float16, 64, 32, 4, 1>(a, s) : conv_launch_t<_Float16, _Float16, 64, 128, 2, 2>(a, s); }
            else                       { small_n ? conv_launch_t<_Float16, float, 64, 32, 4, 1>(a, s) : conv_launch_t<_Float16, float, 64, 128, 2, 2>(a, s); }
        } else {
            if (a.out_dtype == DT_F16) { small_n ? conv_launch_t<_Float16, _Float16, 32, 32, 4, 1>(a, s) : conv_launch_t<_Float16, _Float16, 32, 128, 2, 2>(a, s); }
            else                       { small_n ? conv_launch_t<_Float16, float, 32, 32, 4, 1>(a, s) : conv_launch_t<_Float16, float, 32, 128, 2, 2>(a, s); }
        }
    } else {
        if (Ctot % 32 == 0) {
            if (a.out_dtype == DT_BF16)     { small_n ? conv_launch_t<float, __bf16, 32, 32, 4, 1>(a, s) : conv_launch_t<float, __bf16, 32, 128, 2, 2>(a, s); }
            else if (a.out_dtype == DT_F16) { small_n ? conv_launch_t<float, _Float16, 32, 32, 4, 1>(a, s) : conv_launch_t<float, _Float16, 32, 128, 2, 2>(a, s); }
            else                            { small_n ? conv_launch_t<float, float, 32, 32, 4, 1>(a, s) : conv_launch_t<float, float, 32, 128, 2, 2>(a, s); }
        } else {  // Cin = 4 (network input): one 4-channel chunk per tap
            if (a.out_dtype == DT_BF16)     conv_launch_t<float, __bf16, 4, 128, 2, 2>(a, s);
            else if (a.out_dtype == DT_F16) conv_launch_t<float, _Float16, 4, 128, 2, 2>(a, s);
            else                            conv_launch_t<float, float, 4, 128, 2, 2>(a, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm finalize for the consumers that take a coefficient array (the FIR resampling kernels): one block per batch item
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_finalize_kernel(const long long* __restrict__ st0, int C0, const long long* __restrict__ st1,
                                                          int C1, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int groups, float inv_n, float eps, float* __restrict__ coef) {
    const int b = blockIdx.x, C = C0 + C1;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float2 v = gn_coef_of(st0, C0, st1, C1, gamma, beta, groups, inv_n, eps, b, c);
        coef[((size_t)b * C + c) * 2] = v.x;
        coef[((size_t)b * C + c) * 2 + 1] = v.y;
    }
}

// The same from per-workgroup partial totals (ConvArgs::stats_part): one block per (item, group); the threads walk the group's
// (channel, workgroup) pairs, the integer sums are order-independent, the arithmetic behind them is gn_coef_of's.
__global__ __launch_bounds__(256) void gn_finalize_part_kernel(const long long* __restrict__ st0, const long long* __restrict__ pt0, int nt0, int C0,
                                                               const long long* __restrict__ st1, const long long* __restrict__ pt1, int nt1, int C1,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, int groups,
                                                               float inv_n, float eps, float* __restrict__ coef) {
    const int b = blockIdx.x, g = blockIdx.y, C = C0 + C1, cpg = C / groups, c0 = g * cpg;
    long long S = 0, Q = 0;
    for (int k = 0; k < cpg; ++k) {
        const int cc = c0 + k;
        const bool first = cc < C0;
        const long long* pt = first ? pt0 : pt1;
        const int Cs = first ? C0 : C1, cl = first ? cc : cc - C0, nt = first ? nt0 : nt1;
        if (pt) {
            for (int t = threadIdx.x; t < nt; t += 256) {
                const long long* q = pt + (((size_t)b * nt + t) * Cs + cl) * 2;
                S += q[0]; Q += q[1];
            }
        } else if (threadIdx.x == 0) {
            const long long* q = (first ? st0 : st1) + ((size_t)b * Cs + cl) * 2;
            S += q[0]; Q += q[1];
        }
    }
    __shared__ long long red[2][4];
    for (int o = 32; o > 0; o >>= 1) { S += __shfl_xor(S, o); Q += __shfl_xor(Q, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = S; red[1][threadIdx.x >> 6] = Q; }
    __syncthreads();
    if (threadIdx.x < cpg) {
        S = red[0][0] + red[0][1] + red[0][2] + red[0][3]; Q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const double mean = (double)S * (1.0 / 1048576.0) * (double)inv_n;                // (gn_coef_of, use_device.h)
        double var = (double)Q * (1.0 / 1048576.0) * (double)inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float v = (float)var + eps;
        float rstd = __builtin_amdgcn_rsqf(v);
        rstd = rstd * (1.5f - 0.5f * v * rstd * rstd);
        const int c = c0 + threadIdx.x;
        const float a = gamma[c] * rstd;
        coef[((size_t)b * C + c) * 2] = a;
        coef[((size_t)b * C + c) * 2 + 1] = beta[c] - (float)mean * a;
    }
}

void launch_gn_finalize(const long long* st0, int C0, const long long* st1, int C1, const float* gamma, const float* beta,
                        int groups, int hw, float eps, float* coef, int B, hipStream_t s, const long long* pt0, int nt0,
                        const long long* pt1, int nt1) {
    const float inv_n = 1.0f / ((float)((C0 + C1) / groups) * (float)hw);
    if (pt0 || pt1)
        hipLaunchKernelGGL(gn_finalize_part_kernel, dim3(B, groups), dim3(256), 0, s, st0, pt0, nt0, C0, st1, pt1, nt1, C1, gamma, beta, groups, inv_n, eps, coef);
    else
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, st0, C0, st1, C1, gamma, beta, groups, inv_n, eps, coef);
}

// ---------------------------------------------------------------------------------------------------------
// FIR x2 resampling, separable [1,3,3,1]: up -> polyphase (0.25, 0.75) pairs; down -> [1,3,3,1]/8 per axis
// ---------------------------------------------------------------------------------------------------------
// 16-byte store with the nontemporal cache policy: the resamplers' outputs are touched once and, on the large maps, larger than the L2
// (same-box end-to-end A/B, round 5: -0.25 % per score evaluation, profiles/r5_e2e_ab_cache_policy.txt; -DFIR_NT=0 for the plain store)
#ifndef FIR_NT
#define FIR_NT 1
#endif
template <typename T, int N>
DEVI void store16_nt(T* p, const float (&v)[N]) {
    const uint4 u = Vec16<T>::pack(v);
#if FIR_NT
    __builtin_nontemporal_store(u.x, reinterpret_cast<unsigned*>(p)); __builtin_nontemporal_store(u.y, reinterpret_cast<unsigned*>(p) + 1);
    __builtin_nontemporal_store(u.z, reinterpret_cast<unsigned*>(p) + 2); __builtin_nontemporal_store(u.w, reinterpret_cast<unsigned*>(p) + 3);
#else
    *reinterpret_cast<uint4*>(p) = u;
#endif
}
template <typename T, bool UP>
__global__ __launch_bounds__(256) void fir_kernel(const T* __restrict__ src, const float* __restrict__ coef, int act,
                                                  T* __restrict__ out_act, T* __restrict__ out_raw, int B, int H,
                                                  int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    constexpr bool ACC = sizeof(T) == 4;
    const int cv = C / VEC;
    const int OH = UP ? 2 * H : H / 2, OW = UP ? 2 * W : W / 2;
    const long total = (long)B * OH * OW * cv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * VEC;
        long pix = idx / cv;
        const int ox = (int)(pix % OW); pix /= OW;
        const int oy = (int)(pix % OH);
        const int b = (int)(pix / OH);
        float ca[VEC], cb[VEC];
        const bool want_act = out_act != nullptr;
        if (want_act && coef) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { ca[k] = coef[((size_t)b * C + c + k) * 2]; cb[k] = coef[((size_t)b * C + c + k) * 2 + 1]; }
        }
        float ar[VEC], aa[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { ar[k] = 0.f; aa[k] = 0.f; }
        constexpr int NT = UP ? 2 : 4;
        int ys[NT], xs[NT]; float wy[NT], wx[NT];
        if (UP) {
            const int my = oy >> 1, mx = ox >> 1;
            if (oy & 1) { ys[0] = my; wy[0] = 0.75f; ys[1] = my + 1; wy[1] = 0.25f; }
            else        { ys[0] = my - 1; wy[0] = 0.25f; ys[1] = my; wy[1] = 0.75f; }
            if (ox & 1) { xs[0] = mx; wx[0] = 0.75f; xs[1] = mx + 1; wx[1] = 0.25f; }
            else        { xs[0] = mx - 1; wx[0] = 0.25f; xs[1] = mx; wx[1] = 0.75f; }
        } else {
            const float k4[4] = {0.125f, 0.375f, 0.375f, 0.125f};
#pragma unroll
            for (int a = 0; a < NT; ++a) { ys[a] = 2 * oy - 1 + a; wy[a] = k4[a]; xs[a] = 2 * ox - 1 + a; wx[a] = k4[a]; }
        }
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            if (ys[a] < 0 || ys[a] >= H) continue;
#pragma unroll
            for (int e = 0; e < NT; ++e) {
                if (xs[e] < 0 || xs[e] >= W) continue;
                const float wgt = wy[a] * wx[e];
                float v[VEC];
                Vec16<T>::load(src + ((size_t)(b * H + ys[a]) * W + xs[e]) * C + c, v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    ar[k] = fmaf(wgt, v[k], ar[k]);
                    if (want_act) {
                        float u = coef ? fmaf(v[k], ca[k], cb[k]) : v[k];
                        if (act) u = silu_f<ACC>(u);
                        aa[k] = fmaf(wgt, u, aa[k]);
                    }
                }
            }
        }
        const size_t o = ((size_t)(b * OH + oy) * OW + ox) * C + c;
        if (out_raw) Vec16<T>::store(out_raw + o, ar);
        if (want_act) Vec16<T>::store(out_act + o, aa);
    }
}

// Up x2, one thread per 2x2 INPUT neighbourhood {m,m+1} x {n,n+1} (8 or 4 channels): it owns the 2x2 output block
// rows {2m+1, 2m+2} x cols {2n+1, 2n+2}, which depends on exactly those four inputs -> each input is normalised and
// activated 4x less often than in the output-stationary form (the kernel was VALU-bound on SiLU).
template <typename T>
__global__ __launch_bounds__(256) void fir_up_blk_kernel(const T* __restrict__ src, const float* __restrict__ coef,
                                                         int act, T* __restrict__ out_act, T* __restrict__ out_raw,
                                                         int B, int H, int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    constexpr bool ACC = sizeof(T) == 4;
    const int cv = C / VEC;
    const long total = (long)B * (H + 1) * (W + 1) * cv;
    const bool want_act = out_act != nullptr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * VEC;
        long r = idx / cv;
        const int n = (int)(r % (W + 1)) - 1; r /= (W + 1);
        const int m = (int)(r % (H + 1)) - 1;
        const int b = (int)(r / (H + 1));
        float ca[VEC], cb[VEC];
        if (want_act && coef) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { ca[k] = coef[((size_t)b * C + c + k) * 2]; cb[k] = coef[((size_t)b * C + c + k) * 2 + 1]; }
        }
        float xr[2][2][VEC], xa[2][2][VEC];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int y = m + dy, x = n + dx;
                if (y >= 0 && y < H && x >= 0 && x < W) {
                    Vec16<T>::load(src + ((size_t)(b * H + y) * W + x) * C + c, xr[dy][dx]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float u = (want_act && coef) ? fmaf(xr[dy][dx][k], ca[k], cb[k]) : xr[dy][dx][k];
                        xa[dy][dx][k] = (want_act && act) ? silu_f<ACC>(u) : u;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { xr[dy][dx][k] = 0.f; xa[dy][dx][k] = 0.f; }
                }
            }
        // output row 2m+1+py takes input rows (m, m+1) with weights (.75,.25) for py=0 and (.25,.75) for py=1
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int oy = 2 * m + 1 + py, ox = 2 * n + 1 + px;
                if (oy < 0 || oy >= 2 * H || ox < 0 || ox >= 2 * W) continue;
                const float wy0 = py ? 0.25f : 0.75f, wy1 = 1.f - wy0, wx0 = px ? 0.25f : 0.75f, wx1 = 1.f - wx0;
                float orr[VEC], oa[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    orr[k] = wy0 * (wx0 * xr[0][0][k] + wx1 * xr[0][1][k]) + wy1 * (wx0 * xr[1][0][k] + wx1 * xr[1][1][k]);
                    oa[k] = wy0 * (wx0 * xa[0][0][k] + wx1 * xa[0][1][k]) + wy1 * (wx0 * xa[1][0][k] + wx1 * xa[1][1][k]);
                }
                const size_t o = ((size_t)(b * 2 * H + oy) * (2 * W) + ox) * C + c;
                if (out_raw) store16_nt(out_raw + o, orr);
                if (want_act) store16_nt(out_act + o, oa);
            }
    }
}

// Down x2, one thread per 2x2 OUTPUT block (8 or 4 channels): the block reads a 6x6 input neighbourhood, each input is
// loaded and normalised + activated once per thread (36 per 4 outputs instead of 64), and the separable [1,3,3,1]/8
// kernel is applied horizontally per input row, then vertically.
// (round 4: FAST = the hot configuration - activated + raw output, affine given, SiLU - with the flags as compile-time constants and the
// column range check as a zero tap weight on a clamped address: with run-time flags hipcc computed both sides of `coef ? fma : v` and
// `act ? silu : u` per element and selected the result of EVERY accumulation by the pixel's validity - 1 110 v_cndmask of 4 200 VALU
// instructions per thread.  Same arithmetic in the same order: a tap weight of k4[t] * 1 is k4[t], and an excluded pixel adds 0.)
template <typename T, bool FAST>
__global__ __launch_bounds__(256) void fir_down_blk_kernel(const T* __restrict__ src, const float* __restrict__ coef_,
                                                           int act_, T* __restrict__ out_act, T* __restrict__ out_raw,
                                                           int B, int H, int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    constexpr bool ACC = sizeof(T) == 4;
    const int cv = C / VEC;
    const int OH = H / 2, OW = W / 2, BH = (OH + 1) / 2, BW = (OW + 1) / 2;
    const long total = (long)B * BH * BW * cv;
    const bool want_act = FAST ? true : out_act != nullptr;
    const float* const coef = coef_;
    const int act = FAST ? 1 : act_;
    const float k4[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * VEC;
        long q = idx / cv;
        const int bx = (int)(q % BW); q /= BW;
        const int by = (int)(q % BH);
        const int b = (int)(q / BH);
        float ca[VEC], cb[VEC];
        if (want_act && coef) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { ca[k] = coef[((size_t)b * C + c + k) * 2]; cb[k] = coef[((size_t)b * C + c + k) * 2 + 1]; }
        }
        float ar[2][2][VEC], aa[2][2][VEC];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) { ar[i][j][k] = 0.f; aa[i][j][k] = 0.f; }
        const int y0 = 4 * by - 1, x0 = 4 * bx - 1;          // first input row / column of the 6x6 neighbourhood
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int y = y0 + r;
            if (y < 0 || y >= H) continue;
            float hr[2][VEC], ha[2][VEC];                    // horizontal pass of this input row for the two output columns
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) { hr[j][k] = 0.f; ha[j][k] = 0.f; }
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int x = x0 + e;
                const bool xok = x >= 0 && x < W;
                if (!FAST && !xok) continue;
                float v[VEC], u[VEC];
                Vec16<T>::load(src + ((size_t)(b * H + y) * W + (FAST ? min(max(x, 0), W - 1) : x)) * C + c, v);
                if (want_act) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        u[k] = (FAST || coef) ? fmaf(v[k], ca[k], cb[k]) : v[k];
                        if (act) u[k] = silu_f<ACC>(u[k]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int t = e - 2 * j;                 // tap index of this input column for output column j
                    if (t < 0 || t > 3) continue;
                    const float wt = FAST ? (xok ? k4[t] : 0.f) : k4[t];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        hr[j][k] = fmaf(wt, v[k], hr[j][k]);
                        if (want_act) ha[j][k] = fmaf(wt, u[k], ha[j][k]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int t = r - 2 * i;
                if (t < 0 || t > 3) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        ar[i][j][k] = fmaf(k4[t], hr[j][k], ar[i][j][k]);
                        if (want_act) aa[i][j][k] = fmaf(k4[t], ha[j][k], aa[i][j][k]);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = 2 * by + i, ox = 2 * bx + j;
                if (oy >= OH || ox >= OW) continue;
                const size_t o = ((size_t)(b * OH + oy) * OW + ox) * C + c;
                if (out_raw) store16_nt(out_raw + o, ar[i][j]);
                if (want_act) store16_nt(out_act + o, aa[i][j]);
            }
    }
}

// Down x2 as a STRIP walk (round 5; the hot configuration only: activated + raw output, affine, SiLU, 16-bit storage): one thread owns two
// output columns x R output rows (8 channels) and walks down its 6-column input band row by row - horizontal pass of the row, then the row is
// added to the (at most two) output rows it belongs to, and an output row leaves as soon as its fourth input row has been added.  Every
// input is loaded, normalised and activated once per thread: 6 (2R + 2) per 2R outputs = 6.75 per output at R = 8 against 9 in the 2 x 2
// block form (the SiLU is what binds this kernel).  Per output the taps are accumulated in the block kernel's order (horizontal t = 0..3
// per row, rows in ascending order): bit-identical results.
template <typename T, int R>
__global__ __launch_bounds__(256) void fir_down_strip_kernel(const T* __restrict__ src, const float* __restrict__ coef, T* __restrict__ out_act,
                                                             T* __restrict__ out_raw, int B, int H, int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    constexpr bool ACC = sizeof(T) == 4;
    const int cv = C / VEC;
    const int OH = H / 2, OW = W / 2, BW = (OW + 1) / 2, BS = (OH + R - 1) / R;
    const long total = (long)B * BS * BW * cv;
    const float k4[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * VEC;
        long q = idx / cv;
        const int bx = (int)(q % BW); q /= BW;
        const int bs = (int)(q % BS);
        const int b = (int)(q / BS);
        float ca[VEC], cb[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { ca[k] = coef[((size_t)b * C + c + k) * 2]; cb[k] = coef[((size_t)b * C + c + k) * 2 + 1]; }
        float ar[2][2][VEC], aa[2][2][VEC];                  // [output row & 1][output column]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < VEC; ++k) { ar[i][j][k] = 0.f; aa[i][j][k] = 0.f; }
        const int oy0 = bs * R, x0 = 4 * bx - 1;
#pragma unroll
        for (int rr = 0; rr < 2 * R + 2; ++rr) {             // input row 2 oy0 - 1 + rr
            const int y = 2 * oy0 - 1 + rr;
            if (y >= 0 && y < H) {
                float hr[2][VEC], ha[2][VEC];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { hr[j][k] = 0.f; ha[j][k] = 0.f; }
#pragma unroll
                for (int e = 0; e < 6; ++e) {
                    const int x = x0 + e;
                    const bool xok = x >= 0 && x < W;
                    float v[VEC], u[VEC];
                    Vec16<T>::load(src + ((size_t)(b * H + y) * W + min(max(x, 0), W - 1)) * C + c, v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) u[k] = ACC ? silu_f<ACC>(fmaf(v[k], ca[k], cb[k])) : 0.f;
                    if (!ACC) affine_silu_pk<VEC>(v, ca, cb, u);             // 16-bit storage: packed fp32, the same IEEE operations
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int t = e - 2 * j;
                        if (t < 0 || t > 3) continue;
                        const float wt = xok ? k4[t] : 0.f;
#pragma unroll
                        for (int k = 0; k < VEC; ++k) { hr[j][k] = fmaf(wt, v[k], hr[j][k]); ha[j][k] = fmaf(wt, u[k], ha[j][k]); }
                    }
                }
                // this input row is tap (rr & 1) + 2 of local output row (rr >> 1) - 1 and tap rr & 1 of local output row rr >> 1
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int i = (rr >> 1) - 1 + w, t = (rr & 1) + 2 - 2 * w;
                    if (i < 0 || i >= R) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            ar[i & 1][j][k] = fmaf(k4[t], hr[j][k], ar[i & 1][j][k]);
                            aa[i & 1][j][k] = fmaf(k4[t], ha[j][k], aa[i & 1][j][k]);
                        }
                }
            }
            if ((rr & 1) && rr >= 3) {                       // local output row (rr >> 1) - 1 has seen its last input row
                const int i = (rr >> 1) - 1, oy = oy0 + i;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ox = 2 * bx + j;
                    if (oy < OH && ox < OW) {
                        const size_t o = ((size_t)(b * OH + oy) * OW + ox) * C + c;
                        store16_nt(out_raw + o, ar[i & 1][j]);
                        store16_nt(out_act + o, aa[i & 1][j]);
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { ar[i & 1][j][k] = 0.f; aa[i & 1][j][k] = 0.f; }
                }
            }
        }
    }
}

static int g_fir_strip = 1;            // use_set_option("fir_strip", 0): the 2 x 2 block form of the down-sampler everywhere; 1: strips of 8 or 4 rows by grid size; 8 / 4: forced
void fir_set_strip(int on) { g_fir_strip = on; }
template <bool UP>
static void fir_launch(const void* src, int dtype, const float* coef, int act, void* out_act, void* out_raw, int B,
                       int H, int W, int C, hipStream_t s) {
    const int vec = dtype == DT_F32 ? 4 : 8;
    if (UP) {
        const long total = (long)B * (H + 1) * (W + 1) * (C / vec);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 256 * 32) blocks = 256 * 32;
        if (blocks < 1) blocks = 1;
        if (dtype == DT_F32)
            hipLaunchKernelGGL((fir_up_blk_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)src, coef, act,
                               (float*)out_act, (float*)out_raw, B, H, W, C);
        else if (dtype == DT_F16)
            hipLaunchKernelGGL((fir_up_blk_kernel<_Float16>), dim3(blocks), dim3(256), 0, s, (const _Float16*)src, coef, act,
                               (_Float16*)out_act, (_Float16*)out_raw, B, H, W, C);
        else
            hipLaunchKernelGGL((fir_up_blk_kernel<__bf16>), dim3(blocks), dim3(256), 0, s, (const __bf16*)src, coef, act,
                               (__bf16*)out_act, (__bf16*)out_raw, B, H, W, C);
    } else {   // (an LDS-tiled variant that activates each input once measured slower: 64-byte input segments, 32 LDS reads/thread)
        const int OH = H / 2, OW = W / 2;
        const long total = (long)B * ((OH + 1) / 2) * ((OW + 1) / 2) * (C / vec);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 256 * 16) blocks = 256 * 16;
        if (blocks < 1) blocks = 1;
        const bool fast = out_act && out_raw && coef && act;    // the res-block down-sampler of the network
        if (fast && dtype != DT_F32 && g_fir_strip && OH % 8 == 0) {        // strip walk: each input activated once per thread (6.75 / 7.5 instead of 9 per output)
            const long per_row_strip = (long)B * ((OW + 1) / 2) * (C / vec);
            // 8-row strips while they still make >= 3 workgroups per CU, else 4-row strips (g_fir_strip = 8 / 4 forces one of them)
            const int R = g_fir_strip == 8 || g_fir_strip == 4 ? g_fir_strip : per_row_strip * (OH / 8) >= 256L * 256 * 3 ? 8 : 4;
            const long tot = per_row_strip * (OH / R);
            const int bl = (int)std::max<long>(1, std::min<long>((tot + 255) / 256, 256 * 16));
#define USE_FIR_STRIP_GO(T, RR) hipLaunchKernelGGL((fir_down_strip_kernel<T, RR>), dim3(bl), dim3(256), 0, s, (const T*)src, coef, (T*)out_act, (T*)out_raw, B, H, W, C)
            if (dtype == DT_F16) { if (R == 8) USE_FIR_STRIP_GO(_Float16, 8); else USE_FIR_STRIP_GO(_Float16, 4); }
            else                 { if (R == 8) USE_FIR_STRIP_GO(__bf16, 8); else USE_FIR_STRIP_GO(__bf16, 4); }
#undef USE_FIR_STRIP_GO
            return;
        }
        if (dtype == DT_F32)
            hipLaunchKernelGGL((fir_down_blk_kernel<float, false>), dim3(blocks), dim3(256), 0, s, (const float*)src, coef, act,
                               (float*)out_act, (float*)out_raw, B, H, W, C);
        else if (dtype == DT_F16) {
            if (fast) hipLaunchKernelGGL((fir_down_blk_kernel<_Float16, true>), dim3(blocks), dim3(256), 0, s, (const _Float16*)src, coef, act,
                                         (_Float16*)out_act, (_Float16*)out_raw, B, H, W, C);
            else hipLaunchKernelGGL((fir_down_blk_kernel<_Float16, false>), dim3(blocks), dim3(256), 0, s, (const _Float16*)src, coef, act,
                                    (_Float16*)out_act, (_Float16*)out_raw, B, H, W, C);
        } else {
            if (fast) hipLaunchKernelGGL((fir_down_blk_kernel<__bf16, true>), dim3(blocks), dim3(256), 0, s, (const __bf16*)src, coef, act,
                                         (__bf16*)out_act, (__bf16*)out_raw, B, H, W, C);
            else hipLaunchKernelGGL((fir_down_blk_kernel<__bf16, false>), dim3(blocks), dim3(256), 0, s, (const __bf16*)src, coef, act,
                                    (__bf16*)out_act, (__bf16*)out_raw, B, H, W, C);
        }
    }
}
void launch_fir_up2(const void* src, int dtype, const float* coef, int act, void* out_act, void* out_raw, int B,
                    int H, int W, int C, hipStream_t s) {
#ifdef USE_HIP_SKIPDBG
    if (getenv("USE_HIP_SKIP_FIR")) return;
#endif
    fir_launch<true>(src, dtype, coef, act, out_act, out_raw, B, H, W, C, s); }
void launch_fir_down2(const void* src, int dtype, const float* coef, int act, void* out_act, void* out_raw, int B,
                      int H, int W, int C, hipStream_t s) {
#ifdef USE_HIP_SKIPDBG
    if (getenv("USE_HIP_SKIP_FIR")) return;
#endif
    fir_launch<false>(src, dtype, coef, act, out_act, out_raw, B, H, W, C, s); }

static int ew_blocks(long n) { long b = (n + 255) / 256; if (b > 8192) b = 8192; if (b < 1) b = 1; return (int)b; }
// ---------------------------------------------------------------------------------------------------------
// Spectrogram glue either side of the sampler: magnitude compression + scaling + frame padding, and back
//   fwd:  Y[b,0,f,t] = |S|^e e^{j arg S} * factor for t < T, 0 for T <= t < Tpad   (model_wrapper.py:92-96, other.py:128-135)
//   back: S[b,f,t]   = (|X| / factor)^(1/e) e^{j arg X}, t < T                      (model_wrapper.py:98-103, 320)
// |z|^p e^{j arg z} = z |z|^(p-1) (0 at z = 0, like the reference's abs()**p * exp(1j*angle()))
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spec_map_kernel(const float2* __restrict__ in, float2* __restrict__ out, long rows,
                                                       int Tin, int Tin_stride, int Tout, float pre, float power, float post) {
    // rows = B*F; in row stride Tin_stride, Tin valid frames; out row stride Tout (frames >= Tin are zero)
    const long total = rows * Tout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / Tout; const int t = (int)(i - r * Tout);
        float2 o = make_float2(0.f, 0.f);
        if (t < Tin) {
            const float2 z = in[r * Tin_stride + t];
            const float a = hypotf(z.x, z.y) * pre;
            const float sc = a > 0.f ? powf(a, power - 1.f) * pre * post : 0.f;
            o = make_float2(z.x * sc, z.y * sc);
        }
        out[i] = o;
    }
}
void launch_spec_map(const float2* in, float2* out, long rows, int Tin, int Tin_stride, int Tout, float pre, float power,
                     float post, hipStream_t s) {
    hipLaunchKernelGGL(spec_map_kernel, dim3(ew_blocks(rows * Tout)), dim3(256), 0, s, in, out, rows, Tin, Tin_stride, Tout, pre,
                       power, post);
}

// ---------------------------------------------------------------------------------------------------------
// Device STFT / iSTFT fused with the spectrogram glue (reference model_wrapper.py:116-122 torch.stft / torch.istft
// with n_fft = 1022, hop 160, periodic Hann, center=True; 92-103 compression; util/other.py:128-135 padding).
// n_fft = 1022 = 2 * 7 * 73 is no radix-2 length (rocFFT runs it as Bluestein); at 0.6 k frames per utterance the transforms
// are 10 GFLOP per batch as plain DFTs, so each direction is ONE kernel of table-driven direct sums: exact twiddles
// cos/sin(2 pi m / N) from a table computed in double precision, indexed by (k n) mod N with an incremental index, fp32
// accumulation - and windowing, reflect padding, compression, frame padding / overlap-add, envelope normalisation and the
// [B][F][T'] layout all happen in the same pass, with no intermediate buffer.
// ---------------------------------------------------------------------------------------------------------
__global__ void twiddle_table_kernel(float2* tw, int N) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < N) {
        double sn, cs;
        sincospi(2.0 * (double)m / (double)N, &sn, &cs);
        tw[m] = make_float2((float)cs, (float)sn);
    }
}
void launch_twiddle_table(float2* tw, int N, hipStream_t s) {
    hipLaunchKernelGGL(twiddle_table_kernel, dim3((N + 255) / 256), dim3(256), 0, s, tw, N);
}

// Y[b][0][k][t] = compress(sum_n w[n] y_b[reflect(t hop + n - N/2)] e^{-2 pi i k n / N}), zero for T <= t < Tpad.
// One workgroup per (frame, item): the windowed frame and the twiddle table sit in LDS, every thread owns bins k, k + 256.
__global__ __launch_bounds__(256) void stft_fwd_kernel(const float* __restrict__ wav, const float* __restrict__ win,
                                                       const float2* __restrict__ tw, float2* __restrict__ Y, int L, int N,
                                                       int hop, int T, int Tpad, int F, float factor, float expo) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    float* xs = reinterpret_cast<float*>(fsm);                              // [N] windowed frame
    float2* tws = reinterpret_cast<float2*>(fsm + ((N * 4 + 15) & ~15));    // [N] twiddles
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (t >= T) {                                             // frame padding (pad_spec)
        for (int k = tid; k < F; k += 256) Y[((size_t)b * F + k) * Tpad + t] = make_float2(0.f, 0.f);
        return;
    }
    for (int n = tid; n < N; n += 256) {
        int q = t * hop + n - N / 2;                          // center=True: reflect padding of N/2 samples on both sides
        if (q < 0) q = -q;
        if (q >= L) q = 2 * (L - 1) - q;
        xs[n] = wav[(size_t)b * L + q] * win[n];
        tws[n] = tw[n];
    }
    __syncthreads();
    for (int k = tid; k < F; k += 256) {
        float re0 = 0.f, im0 = 0.f, re1 = 0.f, im1 = 0.f;     // two partial sums: shorter dependency chains, smaller error
        int idx = 0;
        int n = 0;
        for (; n + 1 < N; n += 2) {
            const float2 c0 = tws[idx]; idx += k; if (idx >= N) idx -= N;
            const float2 c1 = tws[idx]; idx += k; if (idx >= N) idx -= N;
            const float x0 = xs[n], x1 = xs[n + 1];
            re0 = fmaf(x0, c0.x, re0); im0 = fmaf(-x0, c0.y, im0);
            re1 = fmaf(x1, c1.x, re1); im1 = fmaf(-x1, c1.y, im1);
        }
        if (n < N) { const float2 c0 = tws[idx]; re0 = fmaf(xs[n], c0.x, re0); im0 = fmaf(-xs[n], c0.y, im0); }
        const float re = re0 + re1, im = im0 + im1;
        const float a = hypotf(re, im);
        const float sc = a > 0.f ? powf(a, expo - 1.f) * factor : 0.f;      // |S|^e e^{j arg S} * factor = S |S|^(e-1) factor
        Y[((size_t)b * F + k) * Tpad + t] = make_float2(re * sc, im * sc);
    }
}
void launch_stft_fwd(const float* wav, const float* win, const float2* tw, float2* Y, int B, int L, int N, int hop, int T,
                     int Tpad, float factor, float expo, hipStream_t s) {
    const size_t sh = ((size_t)N * 4 + 15 & ~(size_t)15) + (size_t)N * 8;
    hipLaunchKernelGGL(stft_fwd_kernel, dim3(Tpad, B), dim3(256), sh, s, wav, win, tw, Y, L, N, hop, T, Tpad, N / 2 + 1, factor, expo);
}

// y_b[m] = sum_t w[n] x_t[n] / sum_t w[n]^2, n = m + N/2 - t hop in [0, N), over the T' frames, with
// x_t[n] = (1/N) (Re S_0 + (-1)^n Re S_{F-1} + 2 sum_{k=1}^{F-2} (Re S_k cos(2 pi k n / N) - Im S_k sin(2 pi k n / N)))   (irfft),
// S = decompress(X).  One workgroup per (hop-sized block of output samples, item): every output sample of the block meets the
// same <= ceil(N / hop) frames, whose decompressed spectra are staged in LDS once.
__global__ __launch_bounds__(256) void istft_back_kernel(const float2* __restrict__ X, const float* __restrict__ win,
                                                         const float2* __restrict__ tw, float* __restrict__ wav, int L, int N,
                                                         int hop, int Tpad, int F, int nfr, float inv_factor, float inv_expo) {
    extern __shared__ __attribute__((aligned(16))) char ism[];
    float2* tws = reinterpret_cast<float2*>(ism);                           // [N]
    float2* sp = reinterpret_cast<float2*>(ism + (size_t)N * 8);            // [nfr][F] decompressed spectra of the frames in reach
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int t_hi = h;                                       // frames t_hi - nfr + 1 .. t_hi can reach samples [h hop, (h + 1) hop)
    for (int n = tid; n < N; n += 256) tws[n] = tw[n];
    for (int i = tid; i < nfr * F; i += 256) {
        const int j = i / F, k = i - j * F;
        const int t = t_hi - j;
        float2 o = make_float2(0.f, 0.f);
        if (t >= 0 && t < Tpad) {
            const float2 z = X[((size_t)b * F + k) * Tpad + t];
            const float a = hypotf(z.x, z.y) * inv_factor;
            const float sc = a > 0.f ? powf(a, inv_expo - 1.f) * inv_factor : 0.f;
            o = make_float2(z.x * sc, z.y * sc);
        }
        sp[i] = o;
    }
    __syncthreads();
    const float invN = 1.0f / (float)N;
    for (int r = tid; r < hop; r += 256) {
        const int g = h * hop + r;                            // sample index in the centre-padded signal
        const int m = g - N / 2;                              // output sample
        if (m < 0 || m >= L) continue;
        float acc = 0.f, env = 0.f;
        for (int j = 0; j < nfr; ++j) {
            const int t = t_hi - j;
            const int n = g - t * hop;                        // = j hop + r
            if (t < 0 || t >= Tpad || n >= N) continue;
            const float2* S = sp + j * F;
            float s0 = 0.f, s1 = 0.f;
            int idx = n; if (idx >= N) idx -= N;              // (k n) mod N for k = 1
            int k = 1;
            for (; k + 1 < F - 1; k += 2) {
                const float2 c0 = tws[idx]; idx += n; if (idx >= N) idx -= N;
                const float2 c1 = tws[idx]; idx += n; if (idx >= N) idx -= N;
                const float2 a0 = S[k], a1 = S[k + 1];
                s0 = fmaf(a0.x, c0.x, s0); s0 = fmaf(-a0.y, c0.y, s0);
                s1 = fmaf(a1.x, c1.x, s1); s1 = fmaf(-a1.y, c1.y, s1);
            }
            for (; k < F - 1; ++k) {
                const float2 c0 = tws[idx]; idx += n; if (idx >= N) idx -= N;
                const float2 a0 = S[k];
                s0 = fmaf(a0.x, c0.x, s0); s0 = fmaf(-a0.y, c0.y, s0);
            }
            const float x = (S[0].x + ((n & 1) ? -S[F - 1].x : S[F - 1].x) + 2.f * (s0 + s1)) * invN;
            const float w = win[n];
            acc = fmaf(w, x, acc); env = fmaf(w, w, env);
        }
        wav[(size_t)b * L + m] = env > 1e-11f ? acc / env : 0.f;
    }
}
void launch_istft_back(const float2* X, const float* win, const float2* tw, float* wav, int B, int L, int N, int hop, int Tpad,
                       float factor, float expo, hipStream_t s) {
    const int F = N / 2 + 1, nfr = (N + hop - 1) / hop;
    const int nblocks = (L + N / 2 + hop - 1) / hop;          // blocks of centre-padded samples that reach an output sample
    const size_t sh = (size_t)N * 8 + (size_t)nfr * F * 8;
    static LdsAttrOnce attr;
    attr(istft_back_kernel, 64 * 1024);
    hipLaunchKernelGGL(istft_back_kernel, dim3(nblocks, B), dim3(256), sh, s, X, win, tw, wav, L, N, hop, Tpad, F, nfr, 1.f / factor,
                       1.f / expo);
}

// ---------------------------------------------------------------------------------------------------------
// Input packing
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_input_kernel(const float2* __restrict__ x, const float2* __restrict__ y,
                                                         const float2* __restrict__ y2, float4* __restrict__ x4, long npix) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        const float2 a = x[i];
        // y == null: a 2-channel network (NCSNpp(discriminative=True)); the two padding channels meet zero weights
        const float2 c = y ? y[i] : make_float2(0.5f, 0.5f);
        const float4 v = make_float4(2.f * a.x - 1.f, 2.f * a.y - 1.f, 2.f * c.x - 1.f, 2.f * c.y - 1.f);
        if (y2) {                                            // 6 channels (+ 2 of padding that meet zero weights)
            const float2 d = y2[i];
            x4[2 * i] = v;
            x4[2 * i + 1] = make_float4(2.f * d.x - 1.f, 2.f * d.y - 1.f, 0.f, 0.f);
        } else {
            x4[i] = v;
        }
    }
}
void launch_pack_input(const float2* x, const float2* y, const float2* y2, float* x4, long npix, hipStream_t s) {
    hipLaunchKernelGGL(pack_input_kernel, dim3(ew_blocks(npix)), dim3(256), 0, s, x, y, y2, (float4*)x4, npix);
}

// Combine 'sum' with an 8-channel pyramid as its own pass (6-channel network input only; the 4-channel case is fused into the
// producing convolution's epilogue).  One block = 64 consecutive pixels of one item x all C channels, a thread owns one
// 16-byte channel chunk and walks the pixels; per-channel totals of the stored values through LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void combine_add_kernel(T* __restrict__ h, const float* __restrict__ pyr, const float* __restrict__ w8,
                                                          const float* __restrict__ b8, long long* __restrict__ stats, long pix_per_b, int C) {
    constexpr int CH = Vec16<T>::N;
    __shared__ float s_w[512 * 8];
    __shared__ float s_red[256 * CH * 2];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cpr = C / CH;                                  // chunks per pixel (C <= 512, C % CH == 0; cpr <= 256)
    for (int i = tid; i < C * 8; i += 256) s_w[i] = w8[i];
    __syncthreads();
    const int ch = tid % cpr, prow = tid / cpr, pstep = 256 / cpr;
    const bool active = prow < pstep;                        // 256 % cpr != 0: the last partial row of threads idles
    float st_s[CH], st_q[CH], bias[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { st_s[c] = 0.f; st_q[c] = 0.f; bias[c] = b8[ch * CH + c]; }
    const long p0 = (long)blockIdx.x * 64;
    if (active)
        for (long pi = p0 + prow; pi < p0 + 64 && pi < pix_per_b; pi += pstep) {
            const size_t pix = (size_t)b * pix_per_b + pi;
            const float4 q0 = *reinterpret_cast<const float4*>(pyr + pix * 8), q1 = *reinterpret_cast<const float4*>(pyr + pix * 8 + 4);
            float v[CH];
            Vec16<T>::load(h + pix * C + ch * CH, v);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float* wr = s_w + (ch * CH + c) * 8;
                v[c] += bias[c] + wr[0] * q0.x + wr[1] * q0.y + wr[2] * q0.z + wr[3] * q0.w + wr[4] * q1.x + wr[5] * q1.y + wr[6] * q1.z + wr[7] * q1.w;
            }
            const uint4 packed = Vec16<T>::pack(v);
            *reinterpret_cast<uint4*>(h + pix * C + ch * CH) = packed;
            float vr[CH];
            Vec16<T>::load(reinterpret_cast<const T*>(&packed), vr);
#pragma unroll
            for (int c = 0; c < CH; ++c) { st_s[c] += vr[c]; st_q[c] += vr[c] * vr[c]; }
        }
#pragma unroll
    for (int c = 0; c < CH; ++c) { s_red[(tid * CH + c) * 2] = st_s[c]; s_red[(tid * CH + c) * 2 + 1] = st_q[c]; }
    __syncthreads();
    for (int co = tid; co < C; co += 256) {
        const int cc = co / CH, c = co % CH;
        float sm = 0.f, q = 0.f;
        for (int r = 0; r < pstep; ++r) { sm += s_red[((r * cpr + cc) * CH + c) * 2]; q += s_red[((r * cpr + cc) * CH + c) * 2 + 1]; }
        gn_accumulate(stats + ((size_t)b * C + co) * 2, sm, q);
    }
}
void launch_combine_add(void* h, int dtype, const float* pyr, const float* w8, const float* b8, long long* stats, int B, long pix_per_b,
                        int C, hipStream_t s) {
    const dim3 grid((unsigned)((pix_per_b + 63) / 64), B);
    if (dtype == DT_F32)       hipLaunchKernelGGL(combine_add_kernel<float>, grid, dim3(256), 0, s, (float*)h, pyr, w8, b8, stats, pix_per_b, C);
    else if (dtype == DT_F16)  hipLaunchKernelGGL(combine_add_kernel<_Float16>, grid, dim3(256), 0, s, (_Float16*)h, pyr, w8, b8, stats, pix_per_b, C);
    else                       hipLaunchKernelGGL(combine_add_kernel<__bf16>, grid, dim3(256), 0, s, (__bf16*)h, pyr, w8, b8, stats, pix_per_b, C);
}

// ---------------------------------------------------------------------------------------------------------
// Time embedding
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void temb_mlp_kernel(const float* __restrict__ t, int t_stride,
                                                       const float* __restrict__ gfp_w, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ out, int nf) {
    extern __shared__ float sm[];           // [2nf] fourier features, then [4nf] hidden
    float* feat = sm; float* hid = sm + 2 * nf;
    const int b = blockIdx.x, D = 4 * nf;
    const float lt = logf(t[(size_t)b * t_stride]);
    for (int i = threadIdx.x; i < nf; i += blockDim.x) {
        const float xp = lt * gfp_w[i] * 2.f * 3.14159265358979323846f;
        feat[i] = sinf(xp); feat[nf + i] = cosf(xp);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < D; o += blockDim.x) {
        float a = b1[o];
        const float* wr = w1 + (size_t)o * 2 * nf;
        for (int k = 0; k < 2 * nf; ++k) a = fmaf(wr[k], feat[k], a);
        hid[o] = silu_f<true>(a);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < D; o += blockDim.x) {
        float a = b2[o];
        const float* wr = w2 + (size_t)o * D;
        for (int k = 0; k < D; ++k) a = fmaf(wr[k], hid[k], a);
        out[(size_t)b * D + o] = silu_f<true>(a);     // res-blocks consume act(temb) (layerspp.py:303)
    }
}
void launch_temb_mlp(const float* t, int t_stride, const float* gfp_w, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* out_silu, int B, int nf, hipStream_t s) {
    hipLaunchKernelGGL(temb_mlp_kernel, dim3(B), dim3(512), 6 * nf * sizeof(float), s, t, t_stride, gfp_w, w1, b1, w2,
                       b2, out_silu, nf);
}

__global__ __launch_bounds__(256) void temb_dense_kernel(const float* __restrict__ st, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out,
                                                         int rows, int dim) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* wr = W + (size_t)r * dim; const float* x = st + (size_t)b * dim;
    float a = 0.f;
    for (int k = lane; k < dim; k += 64) a = fmaf(wr[k], x[k], a);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) out[(size_t)b * rows + r] = a + bias[r];
}
void launch_temb_dense(const float* silu_temb, const float* W, const float* bias, float* out, int B, int rows,
                       int dim, hipStream_t s) {
    hipLaunchKernelGGL(temb_dense_kernel, dim3((rows + 3) / 4, B), dim3(256), 0, s, silu_temb, W, bias, out, rows, dim);
}

// ---------------------------------------------------------------------------------------------------------
// Attention core: one block per (query token, batch item); fp32 arithmetic
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ v, T* __restrict__ out, int N, int C) {
    extern __shared__ float sm[];            // [C] query, [N] scores
    float* sq = sm; float* sc = sm + C;
    __shared__ float red[4];
    const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const T* qb = q + ((size_t)b * N + i) * C;
    for (int c = tid; c < C; c += 256) sq[c] = to_f(qb[c]);
    __syncthreads();
    const float scale = rsqrtf((float)C);    // int(C) ** -0.5 (layerspp.py:84)
    float mx = -INFINITY;
    for (int j = tid; j < N; j += 256) {
        const T* kb = k + ((size_t)b * N + j) * C;
        float a = 0.f;
        for (int c = 0; c < C; ++c) a = fmaf(sq[c], to_f(kb[c]), a);
        a *= scale; sc[j] = a; mx = fmaxf(mx, a);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < N; j += 256) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = tid; c < C; c += 256) {
        float a = 0.f;
        for (int j = 0; j < N; ++j) a = fmaf(sc[j], to_f(v[((size_t)b * N + j) * C + c]), a);
        out[((size_t)b * N + i) * C + c] = from_f<T>(a * inv);
    }
}
void launch_attention(const void* q, const void* k, const void* v, void* out, int dtype, int B, int N, int C,
                      hipStream_t s) {
    const size_t sh = (size_t)(C + N) * sizeof(float);
    if (dtype == DT_F32)
        hipLaunchKernelGGL((attention_kernel<float>), dim3(N, B), dim3(256), sh, s, (const float*)q, (const float*)k,
                           (const float*)v, (float*)out, N, C);
    else if (dtype == DT_F16)
        hipLaunchKernelGGL((attention_kernel<_Float16>), dim3(N, B), dim3(256), sh, s, (const _Float16*)q,
                           (const _Float16*)k, (const _Float16*)v, (_Float16*)out, N, C);
    else
        hipLaunchKernelGGL((attention_kernel<__bf16>), dim3(N, B), dim3(256), sh, s, (const __bf16*)q,
                           (const __bf16*)k, (const __bf16*)v, (__bf16*)out, N, C);
}

// Long token sequences (the 64x80 bottleneck of the LSGAN refine generator: N = 5120) run the attention core as two
// implicit GEMMs on conv_kernel -- scores = conv1x1(q; weights = k), out = conv1x1(softmax(scores); weights = v^T), the
// [N][C] key tensor already being the packed [Cout][1][Cin] weight layout -- with these two helpers in between.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* __restrict__ x, int cols) {
    __shared__ float red[4];
    T* row = x + (size_t)blockIdx.x * cols;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int j = tid; j < cols; j += 256) mx = fmaxf(mx, to_f(row[j]));
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < cols; j += 256) sum += expf(to_f(row[j]) - mx);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    for (int j = tid; j < cols; j += 256) row[j] = from_f<T>(expf(to_f(row[j]) - mx) * inv);
}
void launch_softmax_rows(void* x, int dtype, long rows, int cols, hipStream_t s) {
    if (dtype == DT_F32) hipLaunchKernelGGL((softmax_rows_kernel<float>), dim3((unsigned)rows), dim3(256), 0, s, (float*)x, cols);
    else if (dtype == DT_F16) hipLaunchKernelGGL((softmax_rows_kernel<_Float16>), dim3((unsigned)rows), dim3(256), 0, s, (_Float16*)x, cols);
    else                 hipLaunchKernelGGL((softmax_rows_kernel<__bf16>), dim3((unsigned)rows), dim3(256), 0, s, (__bf16*)x, cols);
}
// out[b][c][n] = in[b][n][c]  (32x32 tiles through LDS)
template <typename T>
__global__ __launch_bounds__(256) void transpose_nc_kernel(const T* __restrict__ in, T* __restrict__ out, int N, int C) {
    __shared__ T tile[32][33];
    const int b = blockIdx.z, n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (n0 + r < N && c0 + tx < C) tile[r][tx] = in[((size_t)b * N + n0 + r) * C + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (c0 + r < C && n0 + tx < N) out[((size_t)b * C + c0 + r) * N + n0 + tx] = tile[tx][r];
}
void launch_transpose_nc(const void* in, void* out, int dtype, int B, int N, int C, hipStream_t s) {
    dim3 grid((N + 31) / 32, (C + 31) / 32, B);
    if (dtype == DT_F32) hipLaunchKernelGGL((transpose_nc_kernel<float>), grid, dim3(256), 0, s, (const float*)in, (float*)out, N, C);
    else if (dtype == DT_F16) hipLaunchKernelGGL((transpose_nc_kernel<_Float16>), grid, dim3(256), 0, s, (const _Float16*)in, (_Float16*)out, N, C);
    else                 hipLaunchKernelGGL((transpose_nc_kernel<__bf16>), grid, dim3(256), 0, s, (const __bf16*)in, (__bf16*)out, N, C);
}

// ---------------------------------------------------------------------------------------------------------
// Output layer + SDE updates
// ---------------------------------------------------------------------------------------------------------
template <int PC>
__global__ __launch_bounds__(256) void score_out_kernel(const float* __restrict__ pyr, const float* __restrict__ t,
                                                        int t_stride, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float2* __restrict__ score,
                                                        long pix_per_b, float sign) {
    const int b = blockIdx.y;
    const float tv = t ? t[(size_t)b * t_stride] : 1.0f;      // t == null: scale_by_sigma=False
    float wr[PC], wi[PC];
#pragma unroll
    for (int k = 0; k < PC; ++k) { wr[k] = w[k]; wi[k] = w[PC + k]; }
    const float b0 = bias[0], b1 = bias[1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pix_per_b; i += (long)gridDim.x * 256) {
        float hv[PC];
#pragma unroll
        for (int k4 = 0; k4 < PC / 4; ++k4) {
            const float4 q = *reinterpret_cast<const float4*>(pyr + ((size_t)b * pix_per_b + i) * PC + k4 * 4);
            hv[k4 * 4] = q.x; hv[k4 * 4 + 1] = q.y; hv[k4 * 4 + 2] = q.z; hv[k4 * 4 + 3] = q.w;
        }
        float re = b0, im = b1;
#pragma unroll
        for (int k = 0; k < PC; ++k) {
            const float hk = t ? hv[k] / tv : hv[k];           // h / used_sigmas (ncsnpp.py:492-494)
            re += wr[k] * hk; im += wi[k] * hk;
        }
        score[(size_t)b * pix_per_b + i] = make_float2(sign * re, sign * im);  // sign -1: score = -score_net(...) (model_wrapper.py:137)
    }
}
void launch_score_out(const float* pyr, int pc, const float* t, int t_stride, const float* w, const float* bias,
                      float2* score, int B, long pix_per_b, float sign, hipStream_t s) {
    int bx = (int)((pix_per_b + 255) / 256); if (bx > 2048) bx = 2048;
    if (pc == 8) hipLaunchKernelGGL(score_out_kernel<8>, dim3(bx, B), dim3(256), 0, s, pyr, t, t_stride, w, bias, score, pix_per_b, sign);
    else         hipLaunchKernelGGL(score_out_kernel<4>, dim3(bx, B), dim3(256), 0, s, pyr, t, t_stride, w, bias, score, pix_per_b, sign);
}

// Philox4x32-10 counter RNG -> complex normal with E|z|^2 = 1 (torch.randn_like(complex) law).
DEVI void philox_round(unsigned (&c)[4], unsigned (&k)[2]) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k[0], n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k[1], n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
DEVI float2 philox_cnormal(unsigned long long seed, unsigned long long draw, unsigned long long idx) {
    unsigned c[4] = {(unsigned)idx, (unsigned)(idx >> 32), (unsigned)draw, (unsigned)(draw >> 32)};
    unsigned k[2] = {(unsigned)seed, (unsigned)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-logf(u1));           // sqrt(-2 ln u) * sqrt(1/2)
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    return make_float2(r * cs, r * sn);
}
DEVI float2 get_noise(const float2* noise, RngRef rng, long i) {
    if (noise) return noise[i];
    return philox_cnormal(rng.state[0], rng.state[1] + rng.draw, (unsigned long long)i);
}

__global__ __launch_bounds__(256) void fill_noise_kernel(float2* out, RngRef rng, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = get_noise(nullptr, rng, i);
}
void launch_fill_noise(float2* out, RngRef rng, long n, hipStream_t s) {
    hipLaunchKernelGGL(fill_noise_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, out, rng, n);
}

__global__ __launch_bounds__(256) void prior_kernel(const float2* __restrict__ y, const float2* __restrict__ noise,
                                                    RngRef rng, float std1, float2* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float2 z = get_noise(noise, rng, i), yy = y[i];
        x[i] = make_float2(yy.x + z.x * std1, yy.y + z.y * std1);     // sdes.py:254
    }
}
void launch_prior(const float2* y, const float2* noise, RngRef rng, float std1, float2* x, long n, hipStream_t s) {
    hipLaunchKernelGGL(prior_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, y, noise, rng, std1, x, n);
}

__global__ __launch_bounds__(256) void predictor_kernel(const float2* __restrict__ x, const float2* __restrict__ y,
                                                        const float2* __restrict__ score,
                                                        const float2* __restrict__ noise, RngRef rng, float c_drift,
                                                        float c_score, float c_noise, float2* __restrict__ x_out,
                                                        float2* __restrict__ x_mean, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float2 xv = x[i], yv = y[i], sv = score[i], z = get_noise(noise, rng, i);
        // x_mean = x - [theta (y-x) dt - G^2 s]   (predictors.py:62-65, sdes.py:88-92,164-167)
        const float fx = c_drift * (yv.x - xv.x) - c_score * sv.x;
        const float fy = c_drift * (yv.y - xv.y) - c_score * sv.y;
        const float mx = xv.x - fx, my = xv.y - fy;
        if (x_mean) x_mean[i] = make_float2(mx, my);
        x_out[i] = make_float2(mx + c_noise * z.x, my + c_noise * z.y);
    }
}
void launch_predictor(const float2* x, const float2* y, const float2* score, const float2* noise, RngRef rng,
                      float c_drift, float c_score, float c_noise, float2* x_out, float2* x_mean, long n,
                      hipStream_t s) {
    hipLaunchKernelGGL(predictor_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, y, score, noise, rng, c_drift,
                       c_score, c_noise, x_out, x_mean, n);
}

__global__ __launch_bounds__(256) void langevin_norms_kernel(const float2* __restrict__ score,
                                                             const float2* __restrict__ noise, RngRef rng,
                                                             float* __restrict__ partial, long n_per_b) {
    const int b = blockIdx.y;
    float g2 = 0.f, z2 = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_per_b; i += (long)gridDim.x * 256) {
        const long gi = (long)b * n_per_b + i;
        const float2 g = score[gi], z = get_noise(noise, rng, gi);
        g2 += g.x * g.x + g.y * g.y; z2 += z.x * z.x + z.y * z.y;
    }
    __shared__ float rg[4], rz[4];
    for (int o = 32; o > 0; o >>= 1) { g2 += __shfl_xor(g2, o); z2 += __shfl_xor(z2, o); }
    if ((threadIdx.x & 63) == 0) { rg[threadIdx.x >> 6] = g2; rz[threadIdx.x >> 6] = z2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* dst = partial + ((size_t)b * gridDim.x + blockIdx.x) * 2;
        dst[0] = rg[0] + rg[1] + rg[2] + rg[3]; dst[1] = rz[0] + rz[1] + rz[2] + rz[3];
    }
}
void launch_langevin_norms(const float2* score, const float2* noise, RngRef rng, float* partial, int B,
                           long n_per_b, int blocks_per_b, hipStream_t s) {
    hipLaunchKernelGGL(langevin_norms_kernel, dim3(blocks_per_b, B), dim3(256), 0, s, score, noise, rng, partial,
                       n_per_b);
}

__global__ __launch_bounds__(64) void langevin_step_kernel(const float* __restrict__ partial, int B, int blocks_per_b,
                                                           float snr, float* __restrict__ step) {
    // mean over the batch of per-item L2 norms (correctors.py:55-56), one wave
    double gn = 0.0, zn = 0.0;
    for (int b = 0; b < B; ++b) {
        double g2 = 0.0, z2 = 0.0;
        for (int k = threadIdx.x; k < blocks_per_b; k += 64) {
            g2 += (double)partial[((size_t)b * blocks_per_b + k) * 2];
            z2 += (double)partial[((size_t)b * blocks_per_b + k) * 2 + 1];
        }
        for (int o = 32; o > 0; o >>= 1) { g2 += __shfl_xor(g2, o); z2 += __shfl_xor(z2, o); }
        gn += sqrt(g2); zn += sqrt(z2);
    }
    if (threadIdx.x == 0) {
        const float gm = (float)(gn / B), zm = (float)(zn / B);
        const float r = snr * zm / gm;
        step[0] = r * r * 2.f;                                   // correctors.py:57
    }
}
void launch_langevin_step(const float* partial, int B, int blocks_per_b, float snr, float* step, hipStream_t s) {
    hipLaunchKernelGGL(langevin_step_kernel, dim3(1), dim3(64), 0, s, partial, B, blocks_per_b, snr, step);
}

__global__ __launch_bounds__(256) void corrector_kernel(const float2* __restrict__ x, const float2* __restrict__ score,
                                                        const float2* __restrict__ noise, RngRef rng,
                                                        const float* __restrict__ step_dev, float step_host,
                                                        float2* __restrict__ x_out, float2* __restrict__ x_mean,
                                                        long n) {
    const float eps = step_dev ? step_dev[0] : step_host;
    const float sq = sqrtf(eps * 2.f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float2 xv = x[i], g = score[i], z = get_noise(noise, rng, i);
        const float mx = xv.x + eps * g.x, my = xv.y + eps * g.y;   // correctors.py:60-61
        if (x_mean) x_mean[i] = make_float2(mx, my);
        x_out[i] = make_float2(mx + z.x * sq, my + z.y * sq);
    }
}
void launch_corrector(const float2* x, const float2* score, const float2* noise, RngRef rng, const float* step_dev,
                      float step_host, float2* x_out, float2* x_mean, long n, hipStream_t s) {
    hipLaunchKernelGGL(corrector_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, x, score, noise, rng, step_dev,
                       step_host, x_out, x_mean, n);
}

}  // namespace use
