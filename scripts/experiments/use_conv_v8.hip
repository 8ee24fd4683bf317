// conv_v8_kernel: the 3x3 convolution of the large maps built the way the 16-bit weight-gradient kernel (wgrad16_kernel, use_bwd.hip)
// turned out to run best on this chip - 1.43 PFLOP/s there against 0.98 for conv_v4 on the same shape in the same process:
//
//   * TWO independent 4-wave workgroups per CU instead of one 8-wave workgroup whose wave groups trade phases through a barrier every
//     16 MFMAs: a workgroup's barrier stalls only its own four waves, the other workgroup's MFMAs fill the gap.
//   * K is walked in chunks of 16 input channels with ALL NINE taps of a chunk in LDS at once: 72 MFMAs per wave between two
//     barriers (conv_v4: 16), 8 chunks for 128 input channels.
//   * the next chunk (3 halo pieces + 9 weight pieces of 16 bytes per thread) is requested before the MFMAs of the current one and
//     written to LDS after them (register double buffering), the GroupNorm + SiLU transform of the halo pieces runs on the VALU at
//     that point.
//   * tile = 8 rows x 32 columns x 128 output channels; a wave owns 2 rows x 128 channels = 2 x 4 MFMA tiles (128 accumulators).
//     LDS: halo 10 x 34 pixels x 48 B + weights 9 x 128 rows x 48 B (16 channels = 32 B per row, padded to 48: the four 16-lane
//     groups of a ds_read_b128 then cover all 64 banks exactly once) = 71.6 KB, + coefficient table, bias, GroupNorm totals = 78 KB.
//   * operands swapped as in conv_v7 (A = weights, B = pixels): a lane ends up with 8 consecutive channels of one pixel after one
//     v_permlane32_swap - 16-byte stores, residual loads and the GroupNorm partial sums (one-hot MFMAs) without LDS.
//   * weights come from a chunk-major copy [cout / 128][cin / 16][tap][128][16] (36,864 contiguous bytes per chunk), built per layer
//     on first use from the plain layout (conv_v8_prepare).
// Same operator as conv_v4 without the fused 1x1 shortcut segment and the Combine term: in = concat(src0, src1) with the GroupNorm
// affine (+ SiLU) applied while staging, out = ((conv3x3(in) + bias + temb[b]) + res) * out_scale, fixed-point GroupNorm totals.
#include "use_kernels.h"
#include "use_device.h"

#include <cstdio>
#include <cstdlib>
#include <unordered_map>

namespace use {

constexpr int V8_TW = 32, V8_TH = 8, V8_HW = V8_TW + 2, V8_HH = V8_TH + 2, V8_BN = 128, V8_CK = 16;
constexpr int V8_PB = 48;                                     // LDS row pitch of one pixel / one weight row (32 B of data)
constexpr int V8_HALO = V8_HH * V8_HW * V8_PB;                // 16,320
constexpr int V8_WBYTES = 9 * V8_BN * V8_PB;                  // 55,296
constexpr int V8_OFF_W = V8_HALO;
constexpr int V8_OFF_COEF = V8_OFF_W + V8_WBYTES;             // 71,616: GroupNorm affine of the item's input channels (<= 512 x float2)
constexpr int V8_OFF_BIAS = V8_OFF_COEF + 4096;               // 75,712: bias + time embedding of the 128 channels
constexpr int V8_OFF_TOT = V8_OFF_BIAS + 512;                 // 76,224: [128][2] int64 GroupNorm totals of this workgroup
constexpr int V8_SMEM = V8_OFF_TOT + 2048;                    // 78,272
constexpr int V8_NHP = (V8_HH * V8_HW * 2 + 255) / 256;       // halo pieces per thread per chunk (3; 680 pieces)

template <typename T> struct OneHot8;
template <> struct OneHot8<__bf16> { static constexpr unsigned ONE = 0x3F80u; };
template <> struct OneHot8<_Float16> { static constexpr unsigned ONE = 0x3C00u; };

template <typename TIN, bool ACT>
__global__ __launch_bounds__(256, 2) void conv_v8_kernel(ConvArgs p, const TIN* __restrict__ w8) {
    typedef Mfma<TIN> MF;
    typedef typename MF::frag frag;
    constexpr int TM = 2, TN = 4;
    static_assert(sizeof(TIN) == 2, "16-bit storage only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    int tile = blockIdx.x;                                   // XCD-aware order: contiguous band of tiles per XCD
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tiles_x = p.W / V8_TW;
    const int ty0 = (tile / tiles_x) * V8_TH, tx0 = (tile % tiles_x) * V8_TW;
    const int nb = blockIdx.y, n0 = nb * V8_BN;
    const int Ctot = p.C0 + p.C1, nchunks = Ctot / V8_CK;
    const size_t img_px = (size_t)p.H * p.W;
    const int part = tid & 1;

    // ---- this thread's pieces: halo piece it (0..2) = pixel (it * 256 + tid) >> 1 of the 10 x 34 halo, half `part` of its 16 channels
    int ppix[V8_NHP], pdst[V8_NHP]; unsigned pmask[V8_NHP];
#pragma unroll
    for (int it = 0; it < V8_NHP; ++it) {
        const int idx = it * 256 + tid, pix = idx >> 1;
        const int hy = pix / V8_HW, hx = pix - hy * V8_HW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool have = pix < V8_HH * V8_HW;
        const bool inb = have && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        ppix[it] = inb ? gy * p.W + gx : 0;
        pmask[it] = inb ? 0xffffffffu : 0u;
        pdst[it] = have ? pix * V8_PB + part * 16 : -1;
    }
    const TIN* const s0 = (const TIN*)p.src0 + (size_t)b * img_px * p.C0;
    const TIN* const s1 = p.C1 ? (const TIN*)p.src1 + (size_t)b * img_px * p.C1 : nullptr;
    const TIN* const wblk = w8 + (size_t)nb * nchunks * (9 * V8_BN * V8_CK);

    uint4 rh0, rh1, rh2, rw0, rw1, rw2, rw3, rw4, rw5, rw6, rw7, rw8;   // (scalars, not arrays: hipcc keeps a uint4[9] in scratch here)
    static_assert(V8_NHP == 3, "three halo pieces per thread");
#define V8_TAPS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#define V8_HPS(X) X(0) X(1) X(2)
    float2* const coef_lds = reinterpret_cast<float2*>(smem + V8_OFF_COEF);
    const bool xform = ACT || p.coef || p.gn_st0;
    // (macros, not lambdas: with the arrays captured by reference hipcc keeps `rw` in scratch memory)
#define V8_FETCH(C)                                                                                                              \
    {                                                                                                                            \
        const int ch_ = (C) * V8_CK + part * 8;                                                                                  \
        const bool first_ = ch_ < p.C0;                                                                                          \
        const TIN* base_ = first_ ? s0 : s1;                                                                                     \
        const int Cs_ = first_ ? p.C0 : p.C1, cc_ = first_ ? ch_ : ch_ - p.C0;                                                   \
        V8_HPS(V8_LDH)                                                                                                           \
        const TIN* wc_ = wblk + (size_t)(C) * (9 * V8_BN * V8_CK) + tid * 8;                                                     \
        V8_TAPS(V8_LDW)                                                                                                          \
    }
#define V8_LDH(I) rh##I = make_uint4(0u, 0u, 0u, 0u); if (pmask[I]) rh##I = *reinterpret_cast<const uint4*>(base_ + (size_t)ppix[I] * Cs_ + cc_);
#define V8_LDW(T) rw##T = *reinterpret_cast<const uint4*>(wc_ + (T) * (V8_BN * V8_CK));
#define V8_STH(I) { uint4 v_ = rh##I; if (xform) v_ = stage_transform<TIN, ACT>(v_, pmask[I], ca_, cb_); if (pdst[I] >= 0) *reinterpret_cast<uint4*>(smem + pdst[I]) = v_; }
#define V8_STW(T) *reinterpret_cast<uint4*>(smem + V8_OFF_W + ((T) * V8_BN + (tid >> 1)) * V8_PB + part * 16) = rw##T;
#define V8_STASH(C)                                                                                                              \
    {                                                                                                                            \
        float ca_[8], cb_[8];                                                                                                    \
        if (xform) {                                                                                                             \
            const float2* cf_ = coef_lds + (C) * V8_CK + part * 8;                                                               \
            _Pragma("unroll") for (int k = 0; k < 8; ++k) { const float2 v_ = cf_[k]; ca_[k] = v_.x; cb_[k] = v_.y; }            \
        }                                                                                                                        \
        V8_HPS(V8_STH)                                                                                                           \
        V8_TAPS(V8_STW)                                                                                                          \
    }

    V8_FETCH(0)
    // ---- prologue behind the first loads: coefficient table, bias (+ time embedding) table, zeroed totals
    gn_fill_table(coef_lds, p, b, Ctot, tid, 256);
    if (tid < V8_BN) {
        const int co = n0 + tid;
        float add = 0.f;
        if (co < p.Cout) {
            if (p.bias) add += p.bias[co];
            if (p.temb) add += p.temb[(size_t)b * p.temb_bstride + co];
        }
        reinterpret_cast<float*>(smem + V8_OFF_BIAS)[tid] = add;
    }
    unsigned long long* const tot_lds = reinterpret_cast<unsigned long long*>(smem + V8_OFF_TOT);
    tot_lds[tid] = 0ull;                                      // 256 = 128 channels x 2
    __syncthreads();

    f32x16 acc[TM][TN];
    {
        const float* bt = reinterpret_cast<const float*>(smem + V8_OFF_BIAS);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4*>(bt + j * 32 + 8 * g + 4 * (lane >> 5));
#pragma unroll
                for (int i = 0; i < TM; ++i) { acc[i][j][4 * g] = bv.x; acc[i][j][4 * g + 1] = bv.y; acc[i][j][4 * g + 2] = bv.z; acc[i][j][4 * g + 3] = bv.w; }
            }
    }
    // fragment bases: pixels (B operand) of tile row i at tap (0, 0) = halo row 2 wave + i, column lane & 31; weights (A operand) row j * 32 + lane & 31
    const int pbase = ((wave * 2) * V8_HW + (lane & 31)) * V8_PB + (lane >> 5) * 16;
    const int wbase = V8_OFF_W + (lane & 31) * V8_PB + (lane >> 5) * 16;

    for (int c = 0; c < nchunks; ++c) {
        if (c) __syncthreads();                              // the previous chunk's MFMAs are done with the LDS
        V8_STASH(c)
        __syncthreads();
        if (c + 1 < nchunks) V8_FETCH(c + 1)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            frag pf[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) pf[i] = MF::ld(smem + pbase + ((i + tap / 3) * V8_HW + tap % 3) * V8_PB);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = MF::ld(smem + wbase + (tap * V8_BN + j * 32) * V8_PB);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(wf[j], pf[i], acc[i][j]);
        }
    }

    // ---- epilogue: no LDS transposition (operands swapped: lane = pixel, registers = channels) -------------------------------------
    auto one_hot = [&](int gp, unsigned one) -> uint4 {
        const int e = (lane & 31) - 16 * gp - 8 * (lane >> 5);
        const unsigned pat = (e & 1) ? one << 16 : one;
        return make_uint4((e >> 1) == 0 ? pat : 0u, (e >> 1) == 1 ? pat : 0u, (e >> 1) == 2 ? pat : 0u, (e >> 1) == 3 ? pat : 0u);
    };
    const bool has_stats = p.stats != nullptr;
    const uint4 selu = one_hot(0, OneHot8<TIN>::ONE), selq = one_hot(1, 0x3F80u);
    const float scale = p.out_scale;
    TIN* const out_b = (TIN*)p.out + (size_t)b * img_px * p.Cout;
    const TIN* const res_b = p.res ? (const TIN*)p.res + (size_t)b * img_px * p.Cout : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            f32x16 sT;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int gy = ty0 + wave * 2 + i;
                const size_t o = ((size_t)gy * p.W + tx0 + (lane & 31)) * p.Cout + n0 + j * 32 + gp * 16 + (lane >> 5) * 8;
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // X = register group 2gp (channels 16gp + 4h + k), Y = group 2gp + 1: after the half exchange a lane holds channels
                    // 16gp + 8h + (0..7) of its pixel  (element copies first: bit_cast of a vector-element lvalue reads element 0)
                    const float xk = acc[i][j][8 * gp + k], yk = acc[i][j][8 * gp + 4 + k];
                    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xk), __builtin_bit_cast(unsigned, yk), false, false);
                    v[k] = __builtin_bit_cast(float, (unsigned)r[0]);
                    v[4 + k] = __builtin_bit_cast(float, (unsigned)r[1]);
                }
                if (res_b) {
                    float rv[8];
                    Vec16<TIN>::load(res_b + o, rv);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += rv[k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= scale;
                const uint4 packed = Vec16<TIN>::pack(v);
                *reinterpret_cast<uint4*>(out_b + o) = packed;
                if (has_stats) {
                    // GroupNorm partial sums of 16 channels on the matrix pipe: (lane = pixel, 8 channels) x one-hot -> lane = channel,
                    // summed over the wave's 64 pixels; columns 0-15 the sums (of the stored, rounded values), 16-31 the sums of squares
                    bf16x8 sq;
#pragma unroll
                    for (int k = 0; k < 8; ++k) sq[k] = (__bf16)(v[k] * v[k]);
                    if (i == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        sT = MF::mma(__builtin_bit_cast(frag, packed), __builtin_bit_cast(frag, selu), z);
                    } else {
                        sT = MF::mma(__builtin_bit_cast(frag, packed), __builtin_bit_cast(frag, selu), sT);
                    }
                    sT = Mfma<__bf16>::mma(sq, __builtin_bit_cast(bf16x8, selq), sT);
                }
            }
            if (has_stats) {
                float s8[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) s8[r] = sT[r] + sT[r + 8];
#pragma unroll
                for (int r = 0; r < 4; ++r) s8[r] += s8[r + 4];
                float sv = (s8[0] + s8[2]) + (s8[1] + s8[3]);
                sv = reduce_lanes_stride<32>(sv);
                if (lane < 32) {                              // lanes 0-15: sum of channel 16gp + lane, lanes 16-31: its sum of squares
                    const int col = j * 32 + gp * 16 + (lane & 15);
                    const float fx = (lane & 16) ? GN_SQ_SCALE : GN_SUM_SCALE;
                    atomicAdd(tot_lds + col * 2 + ((lane >> 4) & 1), (unsigned long long)__float2ll_rn(sv * fx));
                }
            }
        }
    }
    if (has_stats) {
        __syncthreads();
        const int col = tid >> 1;
        if (n0 + col < p.Cout)
            atomicAdd(reinterpret_cast<unsigned long long*>(p.stats + ((size_t)b * p.Cout + n0 + col) * 2 + (tid & 1)), tot_lds[tid]);
    }
#undef V8_FETCH
#undef V8_STASH
#undef V8_LDH
#undef V8_LDW
#undef V8_STH
#undef V8_STW
#undef V8_TAPS
#undef V8_HPS
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
// chunk-major weight copy [cout / 128][cin / 16][tap][128][16] from the plain layout [cout_pad][tap][cin]
template <typename T>
__global__ __launch_bounds__(256) void v8_repack_kernel(const T* __restrict__ w, T* __restrict__ w8, int cout, int cin) {
    const long total = (long)cout * 9 * cin;
    const int nchunks = cin / V8_CK;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % V8_CK);
        long r = i / V8_CK;
        const int row = (int)(r % V8_BN); r /= V8_BN;
        const int tap = (int)(r % 9); r /= 9;
        const int c = (int)(r % nchunks);
        const int nb = (int)(r / nchunks);
        w8[i] = w[((size_t)(nb * V8_BN + row) * 9 + tap) * cin + c * V8_CK + k];
    }
}
static std::unordered_map<const void*, void*> g_v8_w;        // plain weight pointer (blob-resident: stable) -> chunk-major copy
static int g_v8_on = 0;                                      // 0: off (default until it carries conv_v4's whole operator)
static long g_v8_min_blocks = 512;
void conv_v8_set(int on) { g_v8_on = on; }
void conv_v8_set_min_blocks(long n) { g_v8_min_blocks = n; }
void conv_v8_clear() {
    for (auto& kv : g_v8_w) (void)hipFree(kv.second);
    g_v8_w.clear();
}
static const void* v8_weights(const ConvArgs& a, bool create, hipStream_t s) {
    auto it = g_v8_w.find(a.w);
    if (it != g_v8_w.end()) return it->second;
    if (!create) return nullptr;
    const int cin = a.C0 + a.C1;
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)a.Cout * 9 * cin * 2) != hipSuccess) return nullptr;
    const long total = (long)a.Cout * 9 * cin;
    const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 4096);
    if (a.in_dtype == DT_BF16) hipLaunchKernelGGL(v8_repack_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, (const __bf16*)a.w, (__bf16*)d, a.Cout, cin);
    else                       hipLaunchKernelGGL(v8_repack_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)a.w, (_Float16*)d, a.Cout, cin);
    g_v8_w[a.w] = d;
    return d;
}
bool conv_v8_supports(const ConvArgs& a) {
    const int Ctot = a.C0 + a.C1;
    return a.w != nullptr && a.ntaps == 9 && a.in_dtype != DT_F32 && a.in_dtype == a.out_dtype && a.XC0 + a.XC1 == 0 && a.pyr == nullptr &&
           Ctot % V8_CK == 0 && Ctot <= 512 && (a.C1 == 0 || a.C0 % V8_CK == 0) && a.Cout % V8_BN == 0 && a.cout_pad == a.Cout &&
           a.H % V8_TH == 0 && a.W % V8_TW == 0;
}
// called outside stream captures (plan time / first eager evaluation): builds the chunk-major copy of this layer's weights
void conv_v8_prepare(const ConvArgs& a, hipStream_t s) { if (conv_v8_supports(a)) (void)v8_weights(a, true, s); }
bool conv_v8_eligible(const ConvArgs& a) {
    if (!g_v8_on || !conv_v8_supports(a)) return false;
    const long blocks = (long)(a.H / V8_TH) * (a.W / V8_TW) * (a.Cout / V8_BN);
    return blocks >= g_v8_min_blocks && v8_weights(a, false, nullptr) != nullptr;
}
template <typename TIN, bool ACT>
static void v8_launch_t(const ConvArgs& a, const void* w8, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_v8_kernel<TIN, ACT>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, V8_SMEM);
        attr_set = true;
    }
    const dim3 grid((a.H / V8_TH) * (a.W / V8_TW), a.Cout / V8_BN, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), V8_SMEM, s, a, (const TIN*)w8);
}
void launch_conv_v8(const ConvArgs& a, hipStream_t s) {
    const void* w8 = v8_weights(a, true, s);
    if (!w8) { launch_conv_v4(a, s); return; }
    if (a.in_dtype == DT_BF16) { a.act ? v8_launch_t<__bf16, true>(a, w8, s) : v8_launch_t<__bf16, false>(a, w8, s); }
    else                       { a.act ? v8_launch_t<_Float16, true>(a, w8, s) : v8_launch_t<_Float16, false>(a, w8, s); }
}

}  // namespace use
