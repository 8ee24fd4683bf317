// Device-side helpers shared by the gfx950 kernels (16-byte vector <-> float conversion, SiLU, MFMA traits).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace use {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DEVI __device__ __forceinline__

template <bool ACCURATE>
DEVI float silu_f(float x) {
    if (ACCURATE) return x / (1.0f + expf(-x));
    // bf16 storage: v_exp_f32 / v_rcp_f32 approximations (1 ulp-class) are far below the bf16 rounding of the result;
    // __frcp_rn would expand to a ~10-instruction IEEE division sequence
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}

// ---------------------------------------------------------------------------------------------------------
// 16-byte vector <-> float helpers
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    DEVI static void load(const float* p, float (&v)[4]) {
        float4 u = *reinterpret_cast<const float4*>(p);
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    }
    DEVI static uint4 pack(const float (&v)[4]) {
        float4 u = make_float4(v[0], v[1], v[2], v[3]);
        return __builtin_bit_cast(uint4, u);
    }
    DEVI static void store(float* p, const float (&v)[4]) { *reinterpret_cast<uint4*>(p) = pack(v); }
};
template <> struct Vec16<__bf16> {
    static constexpr int N = 8;
    DEVI static void load(const __bf16* p, float (&v)[8]) {
        uint4 u = *reinterpret_cast<const uint4*>(p);
        bf16x8 b = __builtin_bit_cast(bf16x8, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)b[i];
    }
    DEVI static uint4 pack(const float (&v)[8]) {
        bf16x8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = (__bf16)v[i];
        return __builtin_bit_cast(uint4, b);
    }
    DEVI static void store(__bf16* p, const float (&v)[8]) { *reinterpret_cast<uint4*>(p) = pack(v); }
};

// GroupNorm affine (+ SiLU) of one 16-byte piece while it is staged into LDS: y = act(a*x + b) per channel, zeroed by
// `mask` outside the image (the zero padding of the convolution applies AFTER the activation).
template <typename TIN, bool ACT>
DEVI uint4 stage_transform(const uint4 raw, const unsigned mask, const float (&ca)[16 / sizeof(TIN)],
                           const float (&cb)[16 / sizeof(TIN)]) {
    constexpr int VEC = 16 / sizeof(TIN);
    float v[VEC];
    Vec16<TIN>::load(reinterpret_cast<const TIN*>(&raw), v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        v[k] = fmaf(v[k], ca[k], cb[k]);
        if (ACT) {
            if (sizeof(TIN) == 4) v[k] = v[k] / (1.0f + expf(-v[k]));        // fp32 parity mode: accurate
            else v[k] = v[k] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[k] * -1.44269504088896341f));
        }
    }
    uint4 o = Vec16<TIN>::pack(v);
    o.x &= mask; o.y &= mask; o.z &= mask; o.w &= mask;
    return o;
}

template <> struct Vec16<_Float16> {
    static constexpr int N = 8;
    DEVI static void load(const _Float16* p, float (&v)[8]) {
        uint4 u = *reinterpret_cast<const uint4*>(p);
        f16x8 b = __builtin_bit_cast(f16x8, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)b[i];
    }
    DEVI static uint4 pack(const float (&v)[8]) {
        f16x8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = (_Float16)v[i];
        return __builtin_bit_cast(uint4, b);
    }
    DEVI static void store(_Float16* p, const float (&v)[8]) { *reinterpret_cast<uint4*>(p) = pack(v); }
};

template <typename T> DEVI float to_f(T v) { return (float)v; }
template <typename T> DEVI T from_f(float v) { return (T)v; }

// ---------------------------------------------------------------------------------------------------------
// MFMA traits: one 32x32 output tile per instruction; lanes 0-31 carry the first half of the K slab and lanes
// 32-63 the second half, KPL contiguous k per lane (cdna_hip_programming.md section 3).
// ---------------------------------------------------------------------------------------------------------
template <typename T> struct Mfma;
template <> struct Mfma<__bf16> {
    static constexpr int KM = 16, KPL = 8;
    typedef bf16x8 frag;
    DEVI static frag ld(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
    DEVI static f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<_Float16> {
    static constexpr int KM = 16, KPL = 8;
    typedef f16x8 frag;
    DEVI static frag ld(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
    DEVI static f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<float> {
    static constexpr int KM = 2, KPL = 1;
    typedef float frag;
    DEVI static frag ld(const char* p) { return *reinterpret_cast<const float*>(p); }
    DEVI static f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};

// ---------------------------------------------------------------------------------------------------------
// GroupNorm statistics: producers add their per-channel partial sums into [B][C][2] 64-bit fixed-point totals (integer
// atomics commute, so the result does not depend on the arrival order: bit-reproducible); consumers turn the totals of a
// channel's group into the affine (a, b) of y = a x + b.  Scales: 2^-20 per count for the sums and the sums of squares - a 64-pixel
// partial sum of squares of activations around 1e-3 still holds ~70 counts (at round 2's 2^-12 it rounded to 0 and the variance of a
// near-silent item was off by about eps); range |sum|, sum of squares < 8.7e12 per (item, channel): an rms of 5000 over a 512 x 640 map.
// ---------------------------------------------------------------------------------------------------------
constexpr float GN_SUM_SCALE = 1048576.0f, GN_SQ_SCALE = 1048576.0f;
DEVI void gn_accumulate(long long* dst, float s, float q) {
    atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__float2ll_rn(s * GN_SUM_SCALE));
    atomicAdd(reinterpret_cast<unsigned long long*>(dst + 1), (unsigned long long)__float2ll_rn(q * GN_SQ_SCALE));
}
// (a, b) of channel c (of the concatenation [C0 | C1]) of item b
DEVI float2 gn_coef_of(const long long* __restrict__ st0, int C0, const long long* __restrict__ st1, int C1,
                       const float* __restrict__ gamma, const float* __restrict__ beta, int groups, float inv_n, float eps,
                       int b, int c) {
    const int cpg = (C0 + C1) / groups, g0 = (c / cpg) * cpg;
    long long S = 0, Q = 0;
    for (int k = 0; k < cpg; ++k) {
        const int cc = g0 + k;
        const long long* p = cc < C0 ? st0 + ((size_t)b * C0 + cc) * 2 : st1 + ((size_t)b * C1 + (cc - C0)) * 2;
        S += p[0]; Q += p[1];
    }
    // mean and E[x^2] - mean^2 in fp64 (the cancellation needs it: a handful of multiplies), the reciprocal square root in fp32:
    // v_rsq_f32 + one Newton step (relative error ~1e-7) instead of an fp64 sqrt and division (~100 instructions), so that every
    // workgroup of a consuming convolution can afford to finalise the GroupNorm of its own input channels
    const double mean = (double)S * (1.0 / 1048576.0) * (double)inv_n;
    double var = (double)Q * (1.0 / 1048576.0) * (double)inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float v = (float)var + eps;
    float rstd = __builtin_amdgcn_rsqf(v);
    rstd = rstd * (1.5f - 0.5f * v * rstd * rstd);
    const float a = gamma[c] * rstd;
    return make_float2(a, beta[c] - (float)mean * a);
}
// coefficient table of one item's input channels in LDS (every conv consumer builds it in its prologue)
template <typename ARGS>
DEVI void gn_fill_table(float2* tab, const ARGS& p, int b, int Ctot, int tid, int nthreads) {
    for (int c = tid; c < Ctot; c += nthreads)
        tab[c] = p.gn_st0 ? gn_coef_of(p.gn_st0, p.C0, p.gn_st1, p.C1, p.gn_gamma, p.gn_beta, p.gn_groups, p.gn_inv_n, p.gn_eps, b, c)
                 : p.coef ? *reinterpret_cast<const float2*>(p.coef + ((size_t)b * Ctot + c) * 2) : make_float2(1.f, 0.f);
}

// Sum over the lanes {l, l+S, l+2S, ...} of a wave (S = 4, 8, 16 or 32) with VALU cross-lane ops only (DPP row rotate,
// v_permlane16_swap, v_permlane32_swap) -- no LDS round trips (ds_bpermute) on the epilogue's critical path.
template <int S>
DEVI float reduce_lanes_stride(float v) {
    static_assert(S == 4 || S == 8 || S == 16 || S == 32, "stride");
    if (S <= 4)   // lanes i, i+4, i+8, i+12 of each 16-lane row: rotate by 4, then by 8 (sums are rotation-invariant)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    if (S <= 8)   // row_ror:8
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    if (S <= 16) {   // xor 16: after the swap the two results hold (row0,row0,row2,row2) and (row1,row1,row3,row3)
        const int vi = __builtin_bit_cast(int, v);
        const auto r = __builtin_amdgcn_permlane16_swap(vi, vi, false, false);
        v = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
    }
    {             // xor 32
        const int vi = __builtin_bit_cast(int, v);
        const auto r = __builtin_amdgcn_permlane32_swap(vi, vi, false, false);
        v = __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
    }
    return v;
}

}  // namespace use
