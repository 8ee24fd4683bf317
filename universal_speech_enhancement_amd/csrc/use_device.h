// Device-side helpers shared by the gfx950 kernels (16-byte vector <-> float conversion, SiLU, MFMA traits).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <utility>

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

// silu_f<false>(a x + b) on an even number of channels, two per instruction: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are the IEEE
// operations of the scalar form (bit-identical results); v_exp_f32 / v_rcp_f32 stay scalar.  For kernels bound by the VALU rate of the
// fused activation (round 5: the pyramid head's producer waves, the strip FIR down-sampler): 4 instead of 5.5 full-rate instructions per
// element beside the two quarter-rate ones.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N, bool ACT = true>
DEVI void affine_silu_pk(const float (&x)[N], const float (&ca)[N], const float (&cb)[N], float (&y)[N]) {
    static_assert(N % 2 == 0, "pairs of channels");
#pragma unroll
    for (int k = 0; k < N; k += 2) {
        f32x2 t = {x[k], x[k + 1]};
        t = __builtin_elementwise_fma(t, (f32x2){ca[k], ca[k + 1]}, (f32x2){cb[k], cb[k + 1]});
        if (ACT) {
            f32x2 e = t * (f32x2){-1.44269504088896341f, -1.44269504088896341f};
            e.x = __builtin_amdgcn_exp2f(e.x); e.y = __builtin_amdgcn_exp2f(e.y);
            e = e + (f32x2){1.0f, 1.0f};
            e.x = __builtin_amdgcn_rcpf(e.x); e.y = __builtin_amdgcn_rcpf(e.y);
            t = t * e;
        }
        y[k] = t.x; y[k + 1] = t.y;
    }
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
// 16 MFMAs carrying the GroupNorm + SiLU transform of one 16-bit halo piece, as FIVE asm statements: the only way to keep the
// transform inside the MFMAs' shadow.  Round 4 findings behind this form:
//  * hipcc emits a phase of "16 MFMAs + transform of one piece" as 8 MFMAs, the whole ~60-instruction VALU block (matrix pipe idle),
//    8 MFMAs - whatever sched_group_barrier / sched_barrier asks for: MFMAs and transform are side-effect-free and are placed before
//    the machine scheduler ever sees the barriers.  Such a phase took ~860 cycles against 512 of MFMA issue.  Volatile asm keeps its order.
//  * behind back-to-back v_mfma_f32_32x32x16 one wave issues 4 VALU instructions per MFMA for free (scripts/microbench/mfma_filler:
//    513 cycles bare, 529 with 60 instructions at <= 4 per gap, 573 at 5 per gap over the last 12 gaps, 653 at 7-8 over the last 8;
//    dependent neighbours cost: one element's chain per gap 545) - scalar f32 ops, two independent chains (the low and high half of
//    a dword) interleaved so that no instruction reads its predecessor's result.
//  * the halo piece arrives late: the first four MFMAs of the phase carry nothing, so the wait for the piece (which hipcc puts in front
//    of the first statement that reads it) sits 128+ cycles into the phase, and the transform runs 5 per gap behind the other twelve.
// Per dword (two elements, l / h):  x = unpack, u = a x + b, x = -log2(e) u, x = exp2(x), x = 1 + x, x = rcp(x), u = u x, pack(u_l, u_h):
// the operations of stage_transform in the same order per element: bit-identical results.
// Hazards the compiler cannot see inside asm: a transcendental's result is read two instructions later (one wait state needed); an
// MFMA's accumulator is next touched 8 MFMAs later as SrcC = vDst of the same shape (no wait states needed); the epilogue's VALU
// reads of the accumulators come after compiler-visible MFMAs (the last phase of a chunk carries no piece).
// ---------------------------------------------------------------------------------------------------------
template <typename TIN> struct XfAsm;
#define USE_XF_ASM_STRINGS(T, MFMA, LO, HI, PK)                                                                          \
    template <> struct XfAsm<T> {                                                                                        \
        typedef typename Mfma<T>::frag frag;                                                                             \
        DEVI static void bare4(f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3, const frag& a, const frag& b0, const frag& b1, const frag& b2, const frag& b3) { \
            asm volatile(MFMA " %0, %4, %5, %0\n\t" MFMA " %1, %4, %6, %1\n\t" MFMA " %2, %4, %7, %2\n\t" MFMA " %3, %4, %8, %3" \
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3));          \
        }                                                                                                                \
        /* three MFMAs (accumulators c0..c2 with operands (a0,b0) (a1,b1) (a2,b2)) + the transform of dword d (in place) */ \
        template <bool ACT>                                                                                              \
        DEVI static void dword3(f32x16& c0, f32x16& c1, f32x16& c2, const frag& a0, const frag& a1, const frag& a2, const frag& b0, const frag& b1, \
                                const frag& b2, unsigned& d, float al, float bl, float ah, float bh) {                   \
            float xl, xh, ul, uh;                                                                                        \
            if (ACT)                                                                                                     \
                asm volatile(MFMA " %0, %8, %11, %0\n\t"                                                                 \
                             LO("%4", "%3") "\n\t" HI("%5", "%3") "\n\tv_fma_f32 %6, %4, %14, %15\n\tv_fma_f32 %7, %5, %16, %17\n\tv_mul_f32 %4, 0xbfb8aa3b, %6\n\t" \
                             MFMA " %1, %9, %12, %1\n\t"                                                                 \
                             "v_mul_f32 %5, 0xbfb8aa3b, %7\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_add_f32 %4, 1.0, %4\n\tv_add_f32 %5, 1.0, %5\n\t" \
                             MFMA " %2, %10, %13, %2\n\t"                                                                \
                             "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_mul_f32 %6, %6, %4\n\tv_mul_f32 %7, %7, %5\n\t" PK("%3", "%6", "%7") \
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(d), "=&v"(xl), "=&v"(xh), "=&v"(ul), "=&v"(uh)                         \
                             : "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2), "v"(al), "v"(bl), "v"(ah), "v"(bh));               \
            else                                                                                                         \
                asm volatile(MFMA " %0, %8, %11, %0\n\t"                                                                 \
                             LO("%4", "%3") "\n\t" HI("%5", "%3") "\n\tv_fma_f32 %6, %4, %14, %15\n\tv_fma_f32 %7, %5, %16, %17\n\t" \
                             MFMA " %1, %9, %12, %1\n\t"                                                                 \
                             MFMA " %2, %10, %13, %2\n\t" PK("%3", "%6", "%7")                                           \
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(d), "=&v"(xl), "=&v"(xh), "=&v"(ul), "=&v"(uh)                         \
                             : "v"(a0), "v"(a1), "v"(a2), "v"(b0), "v"(b1), "v"(b2), "v"(al), "v"(bl), "v"(ah), "v"(bh));               \
        }                                                                                                                \
    };
#define USE_XF_BF16_LO(D, S) "v_lshlrev_b32 " D ", 16, " S
#define USE_XF_BF16_HI(D, S) "v_and_b32 " D ", 0xffff0000, " S
#define USE_XF_BF16_PK(D, A, B) "v_cvt_pk_bf16_f32 " D ", " A ", " B
#define USE_XF_F16_LO(D, S) "v_cvt_f32_f16 " D ", " S
#define USE_XF_F16_HI(D, S) "v_cvt_f32_f16_sdwa " D ", " S " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define USE_XF_F16_PK(D, A, B) "v_cvt_pk_f16_f32 " D ", " A ", " B
USE_XF_ASM_STRINGS(__bf16, "v_mfma_f32_32x32x16_bf16", USE_XF_BF16_LO, USE_XF_BF16_HI, USE_XF_BF16_PK)
USE_XF_ASM_STRINGS(_Float16, "v_mfma_f32_32x32x16_f16", USE_XF_F16_LO, USE_XF_F16_HI, USE_XF_F16_PK)
#undef USE_XF_ASM_STRINGS

// 16 MFMAs (acc[i][j] += a[kk][i] b[kk][j], MFMA g = (kk * 2 + i) * 4 + j: the 2 x 4 x 2 register tile of conv_v4 / conv_v2) carrying
// the transform of one 16-byte piece `raw` -> returned
template <typename TIN, bool ACT, typename ACC, typename AF, typename BF>
DEVI uint4 mfma16_with_transform(ACC& acc, const AF& af, const BF& bf, const uint4 raw, const float (&ca)[8], const float (&cb)[8]) {
    typedef XfAsm<TIN> X;
    unsigned d[4] = {raw.x, raw.y, raw.z, raw.w};
    X::bare4(acc[0][0], acc[0][1], acc[0][2], acc[0][3], af[0][0], bf[0][0], bf[0][1], bf[0][2], bf[0][3]);                          // g = 0 .. 3
    X::template dword3<ACT>(acc[1][0], acc[1][1], acc[1][2], af[0][1], af[0][1], af[0][1], bf[0][0], bf[0][1], bf[0][2], d[0], ca[0], cb[0], ca[1], cb[1]);   // 4 .. 6
    X::template dword3<ACT>(acc[1][3], acc[0][0], acc[0][1], af[0][1], af[1][0], af[1][0], bf[0][3], bf[1][0], bf[1][1], d[1], ca[2], cb[2], ca[3], cb[3]);   // 7 .. 9
    X::template dword3<ACT>(acc[0][2], acc[0][3], acc[1][0], af[1][0], af[1][0], af[1][1], bf[1][2], bf[1][3], bf[1][0], d[2], ca[4], cb[4], ca[5], cb[5]);   // 10 .. 12
    X::template dword3<ACT>(acc[1][1], acc[1][2], acc[1][3], af[1][1], af[1][1], af[1][1], bf[1][1], bf[1][2], bf[1][3], d[3], ca[6], cb[6], ca[7], cb[7]);   // 13 .. 15
    return make_uint4(d[0], d[1], d[2], d[3]);
}

// ---------------------------------------------------------------------------------------------------------
// Epilogue staging by ds_write_addtid_b32 (round 6; address = M0[15:0] + 16-bit offset + 4 * lane: no address register, 2 LDS-store cycles per
// wave-instruction against 4 for ds_write_b32 / 6 for ds_write2_b32 - MI355X_MICROARCH.md, LDS: a store's cost is the transfer of its
// address and data registers; the epilogue's 1 024 staging stores per tile were 4 k of its ~10 k cycles).  M0 holds 16 bits, so a wave
// stages HALF a round at a time (conv_v4; the input convolution: one of its two tile rows) - accumulator registers r = 8 hb ... 8 hb + 7 of its four 32-channel blocks = pixel columns 16 hb ...
// 16 hb + 15 of its tile row x 128 channels, 8 KB - and the eight regions end below 64 KB.  Register (j, r') is one lane-linear row of 64
// floats [pixel half h][channel c] at dword j * 512 + r' * 64 + 4 A(j), A(j) = (j & 1) + 8 (j >> 1): the shift by A(j) bank quads makes
// the read-back conflict-free (a 16-lane group of a ds_read_b128 - lanes {0-3, 12-15, 20-27} etc. - reads one pixel's channel chunks
// {0-3, 12-15} and its neighbour's {4-11}: bank quad = A(ch >> 2) + 2 (ch & 3) + 8 h + half: 16 distinct values).
constexpr int v4_stg_off(int r, int j) { return (j * 512 + r * 64 + 4 * ((j & 1) + 8 * (j >> 1))) * 4; }
constexpr int V4_STG_ATID_BYTES = 8384;                      // (3 * 512 + 7 * 64 + 36 + 64) dwords = 8 336 B, rounded up to 64 bytes
template <int J, int R0, typename ACC>
DEVI void v4_stage8(unsigned lds_base, const ACC& a) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[base]\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %[a0] offset:%[o0]\n\tds_write_addtid_b32 %[a1] offset:%[o1]\n\tds_write_addtid_b32 %[a2] offset:%[o2]\n\t"
                 "ds_write_addtid_b32 %[a3] offset:%[o3]\n\tds_write_addtid_b32 %[a4] offset:%[o4]\n\tds_write_addtid_b32 %[a5] offset:%[o5]\n\t"
                 "ds_write_addtid_b32 %[a6] offset:%[o6]\n\tds_write_addtid_b32 %[a7] offset:%[o7]\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [base] "s"(lds_base), [a0] "v"(a[R0]), [a1] "v"(a[R0 + 1]), [a2] "v"(a[R0 + 2]), [a3] "v"(a[R0 + 3]), [a4] "v"(a[R0 + 4]),
                   [a5] "v"(a[R0 + 5]), [a6] "v"(a[R0 + 6]), [a7] "v"(a[R0 + 7]),
                   [o0] "n"(v4_stg_off(0, J)), [o1] "n"(v4_stg_off(1, J)), [o2] "n"(v4_stg_off(2, J)), [o3] "n"(v4_stg_off(3, J)),
                   [o4] "n"(v4_stg_off(4, J)), [o5] "n"(v4_stg_off(5, J)), [o6] "n"(v4_stg_off(6, J)), [o7] "n"(v4_stg_off(7, J))
                 : "memory");
}

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
