// conv_v5_kernel (round 6): conv_v5_kernel - same tile, same pipeline, same epilogue, same arguments - on v_mfma_f32_16x16x32_{bf16,f16}
// instead of v_mfma_f32_32x32x16.  Why: under a convolution's matrix duty both shapes run at the same clock, and the 16x16x32 shape
// draws 0.16 pJ less per FLOP (scripts/microbench/mfma_dtype_power, profiles/r6_mfma_shape_power.txt: at ~37 % duty and 2.39 GHz
// 1 036 W vs 888 W for the same 0.9 PFLOP/s; back to back it holds 2.06 GHz where the 32x32x16 stream is held at 1.75 GHz) - it moves
// half the accumulator registers per FLOP.  The evaluation is limited by the package power limit, so joules are time.
//
// What changes against conv_v4 (16-bit storage only; fp32 stays on conv_v4):
//   * a (32 pixel x 32 channel, K = 32) product = four 16x16x32 MFMAs (pixel half a, channel half b), K = 32 in ONE instruction:
//     32 MFMAs per (tap, chunk) phase and wave instead of 16, the same 12 fragment reads (A: tile row i x pixel half a, B: block j x half b);
//   * lane -> (row m = lane & 15, 8-channel k group g = lane >> 4).  Row m of a fragment is physical row pi(m) of its group of 16 (pixels of
//     A, output channels of B), pi = (0-3, 8-15, 4-7): the 16-lane groups of a ds_read_b128 - lanes {0-3, 12-15, 20-27} and so on - then
//     read rows 0-7 with k group g and rows 8-15 with g ^ 1, and with the halo's slot swizzle s(P) = 2 ((P >> 2) & 1) (conv_v4: (P >> 2) & 3)
//     they hit 16 distinct bank quads for every column shift of a tap; the weight slab keeps the blob's swizzle (row >> 2) & 3, which
//     is conflict-free for the unshifted rows of B;
//   * accumulator (i, a, j, b) register r of lane l: pixel column 16 a + 4 Q(l >> 4) + r of tile row i, Q = (0, 2, 3, 1); channel
//     16 (2 j + b) + pi(l & 15).  The epilogue stages half rounds (i, a) exactly as conv_v4 does, with its own row layout (v5_stage8).
#include "use_kernels.h"
#include "use_device.h"

#include <cstdio>
#include <cstdlib>

namespace use {

constexpr int V4_TW = 32, V4_TH = 16;             // tile: 16 rows x 32 columns
constexpr int V4_HW = V4_TW + 2, V4_HH = V4_TH + 2;
constexpr int V4_BN = 128;
#ifndef V4_ABL                      /* timing-only ablation builds (results wrong): bit 1 transform off, 2 no halo loads / piece stores, */
#define V4_ABL 0                    /* 4 no weight loads / stores, 8 no fragment reads, 16 no epilogue, 32 no chunk-0 staging in the prologue; */
                                    /* epilogue parts: 1024 no output stores, 2048 no LDS transposition, 4096 no statistics, 8192 no residual loads */
#endif
// Cache-policy bits (buffer instruction aux: 1 sc0, 2 nt, 16 sc1) of the output stores and the residual loads: both streams are touched
// once per launch and are larger than the L2 (84-336 MB per launch on the conv_v4 maps).  Same-box end-to-end sweep, round 5
// (profiles/r5_e2e_ab_cache_policy.txt): stores nt + sc1 and residual loads nt: -0.9 % per score evaluation; nt alone -0.5 %, sc0 alone +0.1 %.
#ifndef V4_AUX_OUT
#define V4_AUX_OUT 18
#endif
#ifndef V4_AUX_IN
#define V4_AUX_IN 0
#endif
#ifndef V4_AUX_RES
#define V4_AUX_RES 2
#endif
#ifndef V4_DEAD_LOADS               /* 1: the loads of the last K chunk's staging pass (results unused) go through an empty buffer descriptor */
#define V4_DEAD_LOADS 1
#endif

// ---- conv_v5's own pieces ---------------------------------------------------------------------------------------------------------
DEVI int v5_pi(int m) { return m < 4 ? m : m < 12 ? m + 4 : m - 8; }   // logical MFMA row / column -> physical row of its group of 16
#define V5_SW(P) ((((P) >> 2) & 1) << 1)                               /* slot swizzle of halo pixel P: piece q at slot q ^ V5_SW(P) */
template <typename T> struct V5M;
template <> struct V5M<__bf16> { DEVI static f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); } };
template <> struct V5M<_Float16> { DEVI static f32x4 mma(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); } };

// Epilogue staging of one channel block J of a half round: accumulators (., ., J, 0) and (., ., J, 1), four registers each = rows (jb, r),
// jb = 2 J + b, at float (jb * 4 + r) * 64 + 8 jb of the wave's region (ds_write_addtid_b32: M0 + offset + 4 lane, see v4_stage8)
constexpr int v5_stg_off(int jb, int r) { return ((jb * 4 + r) * 64 + 8 * jb) * 4; }
constexpr int V5_STG_BYTES = 8448;                           // (31 * 64 + 56 + 64) floats = 8 416 B, rounded up to 64 bytes
template <int J>
DEVI void v5_stage8(unsigned lds_base, const f32x4& x0, const f32x4& x1) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[base]\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %[a0] offset:%[o0]\n\tds_write_addtid_b32 %[a1] offset:%[o1]\n\tds_write_addtid_b32 %[a2] offset:%[o2]\n\t"
                 "ds_write_addtid_b32 %[a3] offset:%[o3]\n\tds_write_addtid_b32 %[a4] offset:%[o4]\n\tds_write_addtid_b32 %[a5] offset:%[o5]\n\t"
                 "ds_write_addtid_b32 %[a6] offset:%[o6]\n\tds_write_addtid_b32 %[a7] offset:%[o7]\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [base] "s"(lds_base), [a0] "v"(x0[0]), [a1] "v"(x0[1]), [a2] "v"(x0[2]), [a3] "v"(x0[3]), [a4] "v"(x1[0]), [a5] "v"(x1[1]), [a6] "v"(x1[2]), [a7] "v"(x1[3]),
                   [o0] "n"(v5_stg_off(2 * J, 0)), [o1] "n"(v5_stg_off(2 * J, 1)), [o2] "n"(v5_stg_off(2 * J, 2)), [o3] "n"(v5_stg_off(2 * J, 3)),
                   [o4] "n"(v5_stg_off(2 * J + 1, 0)), [o5] "n"(v5_stg_off(2 * J + 1, 1)), [o6] "n"(v5_stg_off(2 * J + 1, 2)), [o7] "n"(v5_stg_off(2 * J + 1, 3))
                 : "memory");
}

// 32 MFMAs (acc[i][a][j][b] += A[i][a] B[j][b], g = (2 i + a) * 8 + 2 j + b) carrying the GroupNorm + SiLU transform of one 16-byte halo piece, as
// asm statements (conv_v4's XfAsm, use_device.h, re-cut for the 16-cycle instruction: the first eight MFMAs bare - the piece arrives late -,
// then per dword six MFMAs with the 15-instruction chain 3 2 3 2 3 2 in their gaps; the chain and its order per element are stage_transform's:
// bit-identical results).  Hazards inside the asm: a transcendental's result is read two instructions later; every accumulator is used once.
template <typename TIN> struct V5Xf;
#define USE_V5XF(T, MFMA, LO, HI, PK)                                                                                    \
    template <> struct V5Xf<T> {                                                                                         \
        typedef typename Mfma<T>::frag frag;                                                                             \
        DEVI static void bare8(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, f32x4& c4, f32x4& c5, f32x4& c6, f32x4& c7, const frag& a,       \
                               const frag& b0, const frag& b1, const frag& b2, const frag& b3, const frag& b4, const frag& b5, const frag& b6, const frag& b7) { \
            asm volatile(MFMA " %0, %8, %9, %0\n\t" MFMA " %1, %8, %10, %1\n\t" MFMA " %2, %8, %11, %2\n\t" MFMA " %3, %8, %12, %3\n\t" \
                         MFMA " %4, %8, %13, %4\n\t" MFMA " %5, %8, %14, %5\n\t" MFMA " %6, %8, %15, %6\n\t" MFMA " %7, %8, %16, %7"    \
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)                         \
                         : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7));                        \
        }                                                                                                                \
        /* six MFMAs (accumulator ck with operands (ak, bk)) + the transform of dword d (in place) */                     \
        template <bool ACT>                                                                                              \
        DEVI static void dword6(f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, f32x4& c4, f32x4& c5, const frag& a0, const frag& a1, const frag& a2,    \
                                const frag& a3, const frag& a4, const frag& a5, const frag& b0, const frag& b1, const frag& b2, const frag& b3,     \
                                const frag& b4, const frag& b5, unsigned& d, float al, float bl, float ah, float bh) {   \
            float xl, xh, ul, uh;                                                                                        \
            if (ACT)                                                                                                     \
                asm volatile(MFMA " %0, %11, %17, %0\n\t"                                                                \
                             LO("%7", "%6") "\n\t" HI("%8", "%6") "\n\tv_fma_f32 %9, %7, %23, %24\n\t"                    \
                             MFMA " %1, %12, %18, %1\n\t"                                                                \
                             "v_fma_f32 %10, %8, %25, %26\n\tv_mul_f32 %7, 0xbfb8aa3b, %9\n\t"                            \
                             MFMA " %2, %13, %19, %2\n\t"                                                                \
                             "v_mul_f32 %8, 0xbfb8aa3b, %10\n\tv_exp_f32 %7, %7\n\tv_exp_f32 %8, %8\n\t"                  \
                             MFMA " %3, %14, %20, %3\n\t"                                                                \
                             "v_add_f32 %7, 1.0, %7\n\tv_add_f32 %8, 1.0, %8\n\t"                                        \
                             MFMA " %4, %15, %21, %4\n\t"                                                                \
                             "v_rcp_f32 %7, %7\n\tv_rcp_f32 %8, %8\n\tv_mul_f32 %9, %9, %7\n\t"                           \
                             MFMA " %5, %16, %22, %5\n\t"                                                                \
                             "v_mul_f32 %10, %10, %8\n\t" PK("%6", "%9", "%10")                                          \
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(d), "=&v"(xl), "=&v"(xh), "=&v"(ul), "=&v"(uh) \
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5),   \
                               "v"(al), "v"(bl), "v"(ah), "v"(bh));                                                      \
            else                                                                                                         \
                asm volatile(MFMA " %0, %11, %17, %0\n\t"                                                                \
                             LO("%7", "%6") "\n\t" HI("%8", "%6") "\n\t"                                                 \
                             MFMA " %1, %12, %18, %1\n\t"                                                                \
                             "v_fma_f32 %9, %7, %23, %24\n\tv_fma_f32 %10, %8, %25, %26\n\t"                              \
                             MFMA " %2, %13, %19, %2\n\t"                                                                \
                             MFMA " %3, %14, %20, %3\n\t"                                                                \
                             PK("%6", "%9", "%10") "\n\t"                                                                \
                             MFMA " %4, %15, %21, %4\n\t"                                                                \
                             MFMA " %5, %16, %22, %5"                                                                    \
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(d), "=&v"(xl), "=&v"(xh), "=&v"(ul), "=&v"(uh) \
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5),   \
                               "v"(al), "v"(bl), "v"(ah), "v"(bh));                                                      \
        }                                                                                                                \
    };
USE_V5XF(__bf16, "v_mfma_f32_16x16x32_bf16", USE_XF_BF16_LO, USE_XF_BF16_HI, USE_XF_BF16_PK)
USE_V5XF(_Float16, "v_mfma_f32_16x16x32_f16", USE_XF_F16_LO, USE_XF_F16_HI, USE_XF_F16_PK)
#undef USE_V5XF

template <typename TIN, bool ACT, typename ACC, typename AF, typename BF>
DEVI uint4 v5_mfma32_with_transform(ACC& acc, const AF& af, const BF& bf, const uint4 raw, const float (&ca)[8], const float (&cb)[8]) {
    typedef V5Xf<TIN> X;
    unsigned d[4] = {raw.x, raw.y, raw.z, raw.w};
    // g = (2 i + a) * 8 + (2 j + b): accumulator acc[i][a][j][b], operands af[i][a], bf[j][b]
#define V5_C(G) acc[((G) >> 4) & 1][((G) >> 3) & 1][((G) >> 1) & 3][(G)&1]
#define V5_A(G) af[((G) >> 4) & 1][((G) >> 3) & 1]
#define V5_B(G) bf[((G) >> 1) & 3][(G)&1]
    X::bare8(V5_C(0), V5_C(1), V5_C(2), V5_C(3), V5_C(4), V5_C(5), V5_C(6), V5_C(7), V5_A(0), V5_B(0), V5_B(1), V5_B(2), V5_B(3), V5_B(4), V5_B(5), V5_B(6), V5_B(7));
#define V5_D6(G0, K)                                                                                                                          \
    X::template dword6<ACT>(V5_C(G0), V5_C(G0 + 1), V5_C(G0 + 2), V5_C(G0 + 3), V5_C(G0 + 4), V5_C(G0 + 5), V5_A(G0), V5_A(G0 + 1), V5_A(G0 + 2), V5_A(G0 + 3), \
                            V5_A(G0 + 4), V5_A(G0 + 5), V5_B(G0), V5_B(G0 + 1), V5_B(G0 + 2), V5_B(G0 + 3), V5_B(G0 + 4), V5_B(G0 + 5), d[K], ca[2 * K],      \
                            cb[2 * K], ca[2 * K + 1], cb[2 * K + 1]);
    V5_D6(8, 0) V5_D6(14, 1) V5_D6(20, 2) V5_D6(26, 3)
#undef V5_D6
#undef V5_C
#undef V5_A
#undef V5_B
    return make_uint4(d[0], d[1], d[2], d[3]);
}


template <typename TIN, typename TOUT, int CK, bool ACT, int EPI>
__global__ __launch_bounds__(512) void conv_v5_kernel(ConvArgs p) {
    typedef Mfma<TIN> MF;                                    // (fragment type and loads; the MFMA itself: V5M below)
    static_assert(sizeof(TIN) == 2 && sizeof(TOUT) == 2 && CK == 32 && EPI >= 0, "conv_v5: 16-bit storage, 32-channel chunks, specialised epilogues");
    constexpr int VEC = 16 / sizeof(TIN);
    constexpr int PARTS = CK / VEC;                          // 16-byte pieces per pixel row of a chunk (4)
    constexpr int PXB = CK * (int)sizeof(TIN);               // 64 bytes per pixel row of a chunk (and per weight row of a slab)
    constexpr int BN = V4_BN, TM = 2, TN = 4;
    // LDS layout (round 4): UNPADDED 64-byte rows with the 16-byte piece q of row P stored at slot q ^ ((P >> 2) & 3) - the layout the
    // blob's slab copy already has.  The round 1-3 layout padded rows to 80 bytes: conflict-free for the fragment reads, but every
    // 16-byte staging store (4 lanes per row, 8 lanes per LDS cycle) then straddled two rows 20 banks apart and hit 4 banks twice;
    // pricing the piece stores alone (lane-linear dummy addresses) gave 7 % of the kernel, the weight stores have the same pattern.
    // Here a staging store writes 128 contiguous bytes per 8 lanes (the slot permutation stays inside a row), the weight store is
    // lane-linear, and the 16-lane groups of a fragment read still see 16 distinct 16-byte bank groups: lanes of equal P mod 4 in
    // a group are 12, 20, 24 (or 4, 12, 24) rows apart, i.e. their (P >> 2) & 3 differ.  Halo rows are 48 pixels apart (34 used):
    // a multiple of 16, so that (P >> 2) & 3 of a lane's pixel depends on the tap's column shift only (3 address registers, one per
    // shift; the row shift and the k step are immediate offsets / one XOR); the 14 spare pixels of a row take the dummy stores.
    constexpr int HROW = 48;                                 // pixels between halo rows in LDS
    constexpr int HPITCH = HROW * PXB;                       // 3072
    constexpr int HALO_BYTES = V4_HH * HPITCH, W_BYTES = BN * PXB;   // 55,296 / 8,192
    constexpr int MAIN_BYTES = 2 * HALO_BYTES + 2 * W_BYTES;
    constexpr int NPIECE = V4_HH * V4_HW * PARTS;            // 2448 pieces per halo chunk
    constexpr int PIECE_ITERS = (NPIECE + 511) / 512;        // 5
    static_assert(PXB == 64, "v4 LDS layout: 64-byte rows");
    static_assert(PARTS == 4 && PIECE_ITERS == 5 && BN * PARTS == 512, "v4 staging layout");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [2][HALO_BYTES] halo tiles, [2][W_BYTES] weight slabs, [512] float2 GroupNorm affine, [2][5][512] int piece tables
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    int tile = blockIdx.x;                                   // XCD-aware order: contiguous band of tiles per XCD
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + V4_TW - 1) / V4_TW;
    const int ty0 = (tile / tiles_x) * V4_TH, tx0 = (tile % tiles_x) * V4_TW;
    const int n0 = blockIdx.y * BN;
    const int Ctot = p.C0 + p.C1, nchunks = Ctot / CK;
    const int XCtot = p.XC0 + p.XC1, nchunks2 = XCtot / CK;
    const int part = tid & (PARTS - 1);

    // (Round 6 measured a start-time spread of the first round's workgroups - all tiles of a launch take the same time, so the CUs of a lone
    // launch run in lock step and every tile's 128 KB of output stores meet the memory system in the same few microseconds: with up to 32 k
    // cycles of spread a launch repeated back to back is 3 % shorter, but in the three-stream evaluation, where other launches already
    // scramble the phases, every cycle of delay is lost: +2 % at 32 k.  profiles/r6_conv_v4_epilogue_ab.txt.  Not kept.)
    const int pim = v5_pi(lane & 15);                        // physical row (pixel of A / channel of B and D) of this lane inside a group of 16
    float addv[2 * TN];                                      // bias + time-embedding bias of this lane's channels: 16 (2 j + b) + pi(lane & 15)
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) {
        const int co = n0 + j * 16 + pim;
        float add = 0.f;
        if (co < p.Cout) {
            if (p.bias) add += p.bias[co];
            if (p.temb) add += p.temb[(size_t)b * p.temb_bstride + co];
        }
        addv[j] = add;
    }

#ifdef USE_HIP_TRACE_BUILD   /* bring-up: lane 0 of waves 0 and 4 of workgroup p.dbg of item 0 stamp the cycle counter at phase boundaries */
    // Stamps go to LDS (behind everything else, 2 x 124 x 16 B) and are copied out at the end: a global store per stamp sits in the
    // VMEM queue of the very waits it is meant to observe (round 4: that form inflated the LDS phases 2-3x).
    const bool tracing = p.trace != nullptr && (int)(blockIdx.x + gridDim.x * blockIdx.z) == p.dbg && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0;
    int trace_n = 0;
    unsigned long long* const trace_lds = reinterpret_cast<unsigned long long*>(smem + 152064) + (wave >> 2) * 248;
#define V4_STAMP(ID)                                                                                   \
    if (tracing && trace_n < 124) {                                                                    \
        trace_lds[2 * trace_n] = (unsigned long long)(ID);                                             \
        trace_lds[2 * trace_n + 1] = __builtin_readcyclecounter(); ++trace_n;                          \
    }
#ifdef USE_HIP_TRACE_PHASES   /* per-phase stamps cost ~100 cycles each: off for chunk-level cycle accounting (ids 50 + c) */
#define V4_TRACE_FORCE(R) asm volatile("" :: "v"((R).x), "v"((R).y), "v"((R).z), "v"((R).w));
#define V4_PSTAMP_C(C_, ID) if ((C_) >= 1 && (C_) <= 2) { V4_STAMP(ID) }
#define V4_PSTAMP(ID) if (c >= 1 && c <= 2) { V4_STAMP(ID) }
#else
#define V4_TRACE_FORCE(R)
#define V4_PSTAMP_C(C_, ID)
#define V4_PSTAMP(ID)
#endif   /* per-phase stamps of the second and third K chunk: 1xx = end of MFMA(T), 2xx = end of LDS(T) (before the barrier) */
#else
#define V4_STAMP(ID)
#define V4_PSTAMP(ID)
#define V4_PSTAMP_C(C_, ID)
#define V4_TRACE_FORCE(R)
#endif
    V4_STAMP(1)
    f32x4 acc[TM][2][TN][2];                                 // [tile row i][pixel half a][channel block j][half b]

    V4_STAMP(11)
    // LDS byte offsets of this lane's fragments: row pi(lane & 15) of the group, 16-byte piece g = lane >> 4 of its 64-byte row at slot
    // g ^ swizzle(row).  A: pixel column pim + dx of tile row 2 wave (+ 16 a columns = + 1024 B, + i / dy rows = + HPITCH: immediates);
    // B: weight row 16 (2 j + b) + pim (+ (2 j + b) * 1024 B).
    const int g_ = lane >> 4;
    int a_dx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int P = pim + dx;                              // (+ 16 a, + row * 48: do not change V5_SW(P))
        a_dx[dx] = wave * 2 * HPITCH + P * PXB + ((g_ ^ V5_SW(P)) << 4);
    }
    const int b_0 = 2 * HALO_BYTES + pim * PXB + ((g_ ^ ((pim >> 2) & 3)) << 4);

    // ---- segment-0 halo pieces: this thread's piece j (0..4) of every chunk --------------------------------------------
    // Per piece: the pixel offset of its global load and the byte offset of its LDS row - kept in two LDS tables (read back by the
    // owning thread only), not in registers: the register file is what limits the depth of the load pipeline below.  Pieces outside
    // the image (zero padding, applied AFTER the activation) are zeroed ONCE here in both halo buffers and from then on written into
    // one of the 14 spare pixels of a halo row: no per-piece mask in the main loop.
    constexpr int COEF_OFF = MAIN_BYTES;                     // [Ctot <= 512] float2
    constexpr int TAB_OFF = COEF_OFF + 512 * 8;              // [2][PIECE_ITERS][512] int
    int* const pix_tab = reinterpret_cast<int*>(smem + TAB_OFF);
    int* const dst_tab = pix_tab + PIECE_ITERS * 512;
    const int dummy_slot = ((tid / 56) * HROW + V4_HW) * PXB + (tid % 56) * 16;   // rows 0..9, pixels 34..47
    // GroupNorm affine (a, b) of every input channel of this item in LDS: finalised here from the producers' totals (or copied
    // from a coefficient array, or the identity) - no separate finalize launch, and the per-chunk reads are LDS reads
    float2* const coef_lds = reinterpret_cast<float2*>(smem + COEF_OFF);   // filled in the prologue, behind the first loads
    float ca[VEC], cb[VEC];                                  // GroupNorm affine of the chunk being staged
    auto load_coef = [&](int chunk) {
        const float2* cf = coef_lds + chunk * CK + part * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { const float2 v = cf[k]; ca[k] = v.x; cb[k] = v.y; }
    };
    // buffer loads: tensor descriptor + uniform SGPR offset + 32-bit lane offset
    // `live` (uniform): 0 makes the descriptor EMPTY (num_records 0) - every lane is out of range, the load returns zeros and fetches nothing.
    // That is how the staging pass of the LAST K chunk (nothing left to stage; its loads stay unconditional because a load on one
    // control-flow path costs hipcc's counted waits) is kept off the memory system: round 5, V4_DEAD_LOADS.
    auto buf_ld = [&](const void* base, unsigned voff, unsigned soff, int live = 1) -> uint4 {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, live ? 0x7fffffff : 0, 0x00020000);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    auto buf_ld_in = [&](const void* base, unsigned voff, unsigned soff, int live) -> uint4 {    // halo pieces (cache policy V4_AUX_IN)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, live ? 0x7fffffff : 0, 0x00020000);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, V4_AUX_IN));
    };
    auto src_ld0 = [&](int chunk, int pixoff, int live = 1) -> uint4 {
        const int c_glob = chunk * CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.C0) { src = (const TIN*)p.src0; Cs = p.C0; c_loc = c_glob; }
        else               { src = (const TIN*)p.src1; Cs = p.C1; c_loc = c_glob - p.C0; }
        const unsigned voff = (unsigned)pixoff * (unsigned)(Cs * (int)sizeof(TIN)) + (unsigned)(part * 16);
        // per-item buffer base: the 32-bit offsets only have to span one image (any batch size, < 2 GB per image tensor)
        return buf_ld_in(src + (size_t)b * p.H * p.W * Cs, voff, (unsigned)(c_loc * (int)sizeof(TIN)), live);
    };
    // ---- segment-1 (shortcut) pieces: the 16x32 centre pixels, 4 per thread, raw ----------------------------------------
    auto load_piece1 = [&](int chunk2, int q, uint4& raw) -> unsigned {
        const int pix = (q * 512 + tid) / PARTS;             // 0..511
        const int gy = ty0 + (pix >> 5), gx = tx0 + (pix & 31);
        const bool inb = gy < p.H && gx < p.W;
        const int c_glob = chunk2 * CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.XC0) { src = (const TIN*)p.x0; Cs = p.XC0; c_loc = c_glob; }
        else                { src = (const TIN*)p.x1; Cs = p.XC1; c_loc = c_glob - p.XC0; }
        const unsigned pixoff = inb ? (unsigned)(gy * p.W + gx) : 0u;
        raw = buf_ld(src + (size_t)b * p.H * p.W * Cs, pixoff * (unsigned)(Cs * (int)sizeof(TIN)) + (unsigned)(part * 16), (unsigned)(c_loc * (int)sizeof(TIN)));
        return inb ? 0xffffffffu : 0u;
    };
    auto piece1_dst = [&](int q, int hb) -> int {
        const int pix = (q * 512 + tid) / PARTS;
        const int P = ((pix >> 5) + 1) * HROW + (pix & 31) + 1;
        return hb * HALO_BYTES + P * PXB + ((part ^ V5_SW(P)) << 4);
    };

    // ---- weights: slab-major copy [tap][chunk][cout_pad][CK]; one 16-byte piece per thread per slab ----------------------
    const unsigned wvoff = (unsigned)tid * 16u;
    // the slab rows of the blob are piece-swizzled (use_kernels.h, ConvArgs::wb) exactly as the LDS image wants them: lane-linear store
    const int wdst = 2 * HALO_BYTES + tid * 16;
    const unsigned slab_b = (unsigned)(p.cout_pad * CK) * (unsigned)sizeof(TIN);     // bytes per (tap, chunk) slab
    const unsigned n0_b = (unsigned)(n0 * CK) * (unsigned)sizeof(TIN);
    // weights of iteration (chunk CC, tap TT) -> R ; TT may run past 8 (wraps into the next chunk)
#define V4_LOAD_W(CC, TT, R)                                                                                         \
    {                                                                                                                \
        const int cw0_ = (TT) > 8 ? (CC) + 1 : (CC);                                                                 \
        const int cw_ = cw0_ < nchunks ? cw0_ : nchunks - 1; /* past the end: a harmless re-load (NO branch: a load on one  */ \
        const int tw_ = (TT) > 8 ? (TT)-9 : (TT);            /* control-flow path only turns hipcc's next wait into vmcnt(0)) */ \
        R = buf_ld(p.wb, wvoff, (unsigned)(tw_ * nchunks + cw_) * slab_b + n0_b, V4_DEAD_LOADS ? (cw0_ < nchunks) : 1);       \
    }
#define V4_STORE_W(BUF, R) { *reinterpret_cast<uint4*>(smem + (BUF)*W_BYTES + wdst) = R; }

    // ---- prologue: chunk 0 halo (synchronous), weights of iterations 0 and 1 -------------------------------------------
    // Load pipeline.  Round 4 measurement: with the halo loads served from cache the kernel ran 7.5 % faster, i.e. it was waiting
    // for memory - a piece was loaded in LDS(k) and waited for ("parked") at the start of LDS(k+1), and the wait for the weight
    // slab issued behind it in LDS(k) forced it to have arrived by then anyway (VMEM returns in order).  Now: halo piece k of the
    // next chunk is issued in LDS(k) into register set k & 1 and first touched by the transform behind the MFMAs of MFMA(k+1) (the
    // wait sits in front of that phase's first MFMA: one and a half iterations of cover, no parking copy); the transformed piece is
    // stored at the end of the same phase (the other halo buffer: nobody reads it during this chunk).  The weight slab of iteration n
    // (L2-resident: every workgroup reads the same 295 KB) is issued in LDS(n-2) BEFORE that phase's halo load, stored in LDS(n-1) -
    // its wait (vmcnt(1)) leaves the younger, slower halo load in flight - and read in LDS(n).  Two full iterations of cover (three
    // halo sets, consumption in MFMA(k+2)) do not fit the 256 registers of a wave: 41 spills.
    uint4 wS, hL[3];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    wS = hL[0] = hL[1] = hL[2] = zero4;
    {
        // Order (round 4, cycle stamps of a steady-state workgroup: 15.1 k cycles from launch to the first MFMA of a 68 k tile, 5.5 k of
        // them the arrival of these 55 KB at the ~10 B / cycle a CU gets, and 4 k of address arithmetic in FRONT of their issue): the
        // pixel offsets of the five pieces first and their loads at once, weights and GroupNorm affine behind them, and everything that
        // needs no memory - LDS slots, tables, zeroing of the padding, accumulator start values - in the shadow of the latency; the
        // pieces are then transformed one by one as they arrive.
        V4_STAMP(12)
        uint4 w0 = zero4, raw[PIECE_ITERS];
        // (GroupNorm affine and weights FIRST: VMEM returns in order, and the table of the affine is needed - behind a rendezvous - before
        // the first piece can be transformed; the pieces are then transformed one by one as they arrive)
        float2 cfv = make_float2(1.f, 0.f);                  // GroupNorm affine of input channel tid of this item
        if (tid < Ctot)
            cfv = p.gn_st0 ? gn_coef_of(p.gn_st0, p.C0, p.gn_st1, p.C1, p.gn_gamma, p.gn_beta, p.gn_groups, p.gn_inv_n, p.gn_eps, b, tid)
                  : p.coef ? *reinterpret_cast<const float2*>(p.coef + ((size_t)b * Ctot + tid) * 2) : make_float2(1.f, 0.f);
        V4_LOAD_W(0, 0, w0);
        V4_LOAD_W(0, 1, wS);                                 // stored by LDS(0)
        int slot[PIECE_ITERS], ppv[PIECE_ITERS]; bool inbv[PIECE_ITERS];
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) {
            const int idx = j * 512 + tid;
            const int pix = idx < NPIECE ? idx / PARTS : 0;
            const int hy = pix / V4_HW, hx = pix - hy * V4_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool inb = idx < NPIECE && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            int pp = inb ? gy * p.W + gx : 0;                // pixel offset inside the item's image (the item offset sits in the buffer base)
#ifdef USE_HIP_ABLATE_HALO
            pp = idx & 127;                                  // (compile-time form: p.dbg is the traced workgroup in trace builds)
#endif
#ifdef USE_HIP_ABLATE
            if (p.dbg & 1) pp = idx & 127;                   // timing only: every halo load hits the same 32 KB (cache-resident): prices the exposed load latency
#endif
            raw[j] = (V4_ABL & 32) ? zero4 : src_ld0(0, pp);
            const int P = hy * HROW + hx;
            slot[j] = P * PXB + ((part ^ V5_SW(P)) << 4);
            ppv[j] = pp; inbv[j] = inb;
            if (idx >= NPIECE) slot[j] = -1;
        }
        __builtin_amdgcn_sched_barrier(0);                   // the loads above are issued before anything below
        V4_STAMP(13)
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) {
            const int idx = j * 512 + tid;
            if (slot[j] >= 0 && !inbv[j]) {
                *reinterpret_cast<uint4*>(smem + slot[j]) = zero4;
                *reinterpret_cast<uint4*>(smem + HALO_BYTES + slot[j]) = zero4;
            }
            pix_tab[idx] = ppv[j];
            slot[j] = inbv[j] ? slot[j] : dummy_slot;
            dst_tab[idx] = slot[j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][a][j][bb][r] = addv[2 * j + bb];   // bias + time-embedding bias: the sum starts there
        if (tid < Ctot) coef_lds[tid] = cfv;
        V4_STAMP(14)
        __syncthreads();                                     // coef_lds complete
        V4_STAMP(15)
        load_coef(0);
        V4_STORE_W(0, w0);
        V4_STAMP(16)
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j)
            *reinterpret_cast<uint4*>(smem + slot[j]) = (V4_ABL & 32) ? raw[j] : stage_transform<TIN, ACT>(raw[j], 0xffffffffu, ca, cb);
    }

    typename MF::frag af[TM][2], bf[TN][2];                  // A: [tile row i][pixel half a], B: [channel block j][half b]; K = 32 in one instruction
    int dst_ = 0;                                            // LDS address of the piece the next MFMA phase transforms
#ifdef USE_HIP_SETPRIO
#define V4_SETPRIO(N) __builtin_amdgcn_s_setprio(N);
#else
#define V4_SETPRIO(N)
#endif
    // V4_PEEL_LAST (round 5): the last K chunk of a segment has nothing to stage - its staging pass (five halo loads, their transforms and
    // stores: a quarter of the loop's staging work at Cin = 128) used to run anyway because a load on one control-flow path costs hipcc's
    // counted waits.  Peeled: the chunk loop runs nchunks - 1 times with staging and the last chunk is a second copy of the code without.
    // Pieces are then loaded in LDS(1..5) and transformed in MFMA(3..7) (V4_P0 = 1), so that LDS(c, 0) - which group 0 issues at the end of
    // chunk c - 1 - carries no load and needs no copy of its own.
#ifdef V4_PEEL_LAST
#define V4_P0 1
#else
#define V4_P0 0
#define V4_STG 1
#endif
#define V4_LD_PHASE(T) (V4_STG && (T) >= V4_P0 && (T) < PIECE_ITERS + V4_P0)            /* LDS(T) issues the load of piece T - V4_P0 */
#define V4_XF_PHASE(T) (V4_STG && (T) >= V4_P0 + 2 && (T) < PIECE_ITERS + V4_P0 + 2)   /* MFMA(T) carries the transform of piece T - V4_P0 - 2 */
#define V4_XF_IDX(T) (V4_XF_PHASE(T) ? (T) - V4_P0 - 2 : 0)
#define V4_LDS(CC, T)                                                                                                \
    {                                                                                                                \
        const int cc_ = (CC);                                                                                        \
        const int par_ = cc_ & 1;                            /* halo buffer this chunk reads; it parity = par_ ^ (T&1) */ \
        const int cn_ = cc_ + 1 < nchunks ? cc_ + 1 : cc_;   /* chunk being staged (last chunk: itself again, results unused - */ \
        int pix_ = 0;                                        /* everything below is unconditional, see V4_LOAD_W)             */ \
        /* table entries first: they return ahead of the fragments */                                                \
        if (V4_LD_PHASE(T)) pix_ = (V4_ABL & 128) ? tid * 3 : pix_tab[(V4_LD_PHASE(T) ? (T) - V4_P0 : 0) * 512 + tid]; \
        if (V4_XF_PHASE(T)) dst_ = (V4_ABL & 128) ? tid * 80 : dst_tab[V4_XF_IDX(T) * 512 + tid];    \
        {                                                                                                            \
            const char* ha_ = smem + par_ * HALO_BYTES + ((T) / 3) * HPITCH;                                         \
            const char* wbuf_ = smem + (par_ ^ ((T)&1)) * W_BYTES;                                                   \
            if (!(V4_ABL & 8) || ((CC) == 0 && (T) == 0)) {                                                          \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                       \
                    _Pragma("unroll") for (int a = 0; a < 2; ++a) af[i][a] = MF::ld(ha_ + a_dx[(T) % 3] + i * HPITCH + a * 16 * PXB); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                       \
                    _Pragma("unroll") for (int bb = 0; bb < 2; ++bb) bf[j][bb] = MF::ld(wbuf_ + b_0 + (2 * j + bb) * 16 * PXB); \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (V4_XF_PHASE(T)) dst_ += (par_ ^ 1) * HALO_BYTES; /* where MFMA(T) puts its transformed piece */          \
        if ((V4_ABL & 256) && V4_XF_PHASE(T)) dst_ = (par_ ^ 1) * HALO_BYTES + tid * 16;                             \
        if (!(V4_ABL & 4)) {                                                                                         \
        V4_STORE_W((par_ ^ ((T)&1)) ^ 1, wS);                                                                        \
        V4_LOAD_W(cc_, (T) + 2, wS);                                                                                 \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                   /* the slab load stays older than the halo load */      \
        if (V4_LD_PHASE(T)) {                                                                                        \
            if ((T) == V4_P0) load_coef(cn_);                                                                        \
            hL[((T) - V4_P0) % 3] = src_ld0(cn_, pix_, V4_DEAD_LOADS ? (cc_ + 1 < nchunks) : 1);                                                         \
        }                                                                                                            \
    }
    // (asm as well: through the builtin hipcc gives every 16x16x32 MFMA a fresh destination tuple - 79 spilled registers; the tied "+v" operands keep the accumulators in place)
#define V5_MMA_ALL()                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
            _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                            \
                V5Xf<TIN>::bare8(acc[i][a][0][0], acc[i][a][0][1], acc[i][a][1][0], acc[i][a][1][1], acc[i][a][2][0], acc[i][a][2][1], acc[i][a][3][0], acc[i][a][3][1], \
                                 af[i][a], bf[0][0], bf[0][1], bf[1][0], bf[1][1], bf[2][0], bf[2][1], bf[3][0], bf[3][1]);
#define V4_MFMA(CC, T)                                                                                               \
    {                                                                                                                \
        V4_SETPRIO(1)                                                                                                \
        if (V4_XF_PHASE(T)) {                                /* unconditional at run time: same basic block as the MFMAs */ \
            /* 32 MFMAs with the GroupNorm + SiLU transform of one halo piece in their gaps, as asm statements (V5Xf below): left to */ \
            /* itself hipcc runs the transform with the matrix pipe idle (conv_v4, round 4)                                          */ \
            V4_PSTAMP_C(CC, 300 + (T))                                                                               \
            V4_TRACE_FORCE(hL[V4_XF_IDX(T) % 3])                                                                     \
            V4_PSTAMP_C(CC, 400 + (T))                                                                               \
            if (V4_ABL & 1) {                                                                                        \
                V5_MMA_ALL()                                                                                         \
                if (!(V4_ABL & 64)) *reinterpret_cast<uint4*>(smem + dst_) = hL[V4_XF_IDX(T) % 3];                   \
            } else {                                                                                                 \
                const uint4 t0 = v5_mfma32_with_transform<TIN, ACT>(acc, af, bf, hL[V4_XF_IDX(T) % 3], ca, cb);      \
                if (!(V4_ABL & 64)) *reinterpret_cast<uint4*>(smem + dst_) = t0; /* the other halo buffer: nobody reads it during this chunk */ \
            }                                                                                                        \
        } else {                                                                                                     \
            V5_MMA_ALL()                                                                                             \
        }                                                                                                            \
        V4_SETPRIO(0)                                                                                                \
    }

    // Ping-pong over the 3x3 segment: the two waves that share a SIMD (w and w+4) are always in opposite phases.
    //   phase:   0        1        2        3        4       ...
    //   G0:    LDS(0)  MFMA(0)  LDS(1)  MFMA(1)  LDS(2)
    //   G1:     --     LDS(0)  MFMA(0)  LDS(1)  MFMA(1)
#define V4_BAR() { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
    // rendezvous at the end of an MFMA phase: the only LDS operation a wave may have in flight there is the store of the piece it has
    // just transformed, which nobody reads before the next chunk (several full barriers later) - no lgkmcnt wait in front of it
// (asm with a memory clobber, not __builtin_amdgcn_s_barrier(): the builtin is IntrNoMem, so nothing at IR level would keep LDS accesses on
// their side of it - ADVICE r4; the generated code is instruction-for-instruction the same, checked in round 5)
#define V4_BAR_M() { __builtin_amdgcn_sched_barrier(0); asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    V4_STAMP(2)
    V4_BAR();
    V4_STAMP(3)
#define V4_G0_CHUNK(LAST)                                                                                            \
            V4_STAMP(50 + c)                                                                                         \
            V4_G0_STEP(0) V4_G0_STEP(1) V4_G0_STEP(2) V4_G0_STEP(3) V4_G0_STEP(4) V4_G0_STEP(5) V4_G0_STEP(6) V4_G0_STEP(7) \
            V4_MFMA(c, 8)                                                                                            \
            V4_BAR_M();                                                                                              \
            if (!(LAST) && c + 1 < nchunks) V4_LDS(c + 1, 0)    /* (uniform; the last chunk has nothing left to read) */ \
            V4_BAR();
#define V4_G0_STEP(T) V4_MFMA(c, T) V4_PSTAMP(100 + (T)) V4_BAR_M(); V4_LDS(c, (T) + 1) V4_PSTAMP(200 + (T) + 1) V4_BAR();
#define V4_G1_STEP(T) V4_LDS(c, T) V4_PSTAMP(200 + (T)) V4_BAR(); V4_MFMA(c, T) V4_PSTAMP(100 + (T)) V4_BAR_M();
#define V4_G1_CHUNK                                                                                                  \
            V4_STAMP(50 + c)                                                                                         \
            V4_G1_STEP(0) V4_G1_STEP(1) V4_G1_STEP(2) V4_G1_STEP(3) V4_G1_STEP(4) V4_G1_STEP(5) V4_G1_STEP(6) V4_G1_STEP(7) V4_G1_STEP(8)
#ifdef V4_PEEL_LAST
#define V4_STG 1
    if (wave < 4) {
        V4_LDS(0, 0)
        V4_BAR();
        for (int c = 0; c < nchunks - 1; ++c) { V4_G0_CHUNK(false) }
#undef V4_STG
#define V4_STG 0
        { const int c = nchunks - 1; V4_G0_CHUNK(true) }   /* (its own constant, not the loop's counter: that one hipcc treats as divergent - waterfall loops) */
    } else {
        V4_BAR();
#undef V4_STG
#define V4_STG 1
        for (int c = 0; c < nchunks - 1; ++c) { V4_G1_CHUNK }
#undef V4_STG
#define V4_STG 0
        { const int c = nchunks - 1; V4_G1_CHUNK }
    }
#undef V4_STG
#define V4_STG 1
#else
    if (wave < 4) {
        V4_LDS(0, 0)
        V4_BAR();
        for (int c = 0; c < nchunks; ++c) { V4_G0_CHUNK(false) }
    } else {
        V4_BAR();
        for (int c = 0; c < nchunks; ++c) { V4_G1_CHUNK }
    }
#endif
#undef V4_G0_STEP
#undef V4_G1_STEP
#undef V4_G0_CHUNK
#undef V4_G1_CHUNK
#undef V4_BAR
#undef V4_BAR_M
#undef V4_LDS
#undef V4_LOAD_W

    V4_STAMP(4)
    if (V4_ABL & 16) {                                       // timing only: no epilogue (one value per lane keeps the accumulators alive)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][(r >> 3) & 1][j][(r >> 2) & 1][r & 3];
        if (sum == 1.2345f) ((float*)p.out)[tid] = sum;
        return;
    }
    // ---- segment 1: the fused 1x1 shortcut: raw centre pixels; double-buffered, one barrier per iteration ---------------
    if (nchunks2 > 0) {
        uint4 r0, r1, r2, r3, wa; unsigned m0, m1, m2, m3;
        const unsigned slab2_b = (unsigned)(p.cout_pad * CK) * (unsigned)sizeof(TIN);
#define V4_SC_LOAD(C2)                                                                                        \
        {                                                                                                     \
            m0 = load_piece1(C2, 0, r0); m1 = load_piece1(C2, 1, r1); m2 = load_piece1(C2, 2, r2); m3 = load_piece1(C2, 3, r3); \
            wa = buf_ld(p.w2b, wvoff, (unsigned)(C2)*slab2_b + n0_b);                                         \
        }
        V4_SC_LOAD(0)
        for (int c2 = 0; c2 < nchunks2; ++c2) {
            const int buf = c2 & 1;
            r0.x &= m0; r0.y &= m0; r0.z &= m0; r0.w &= m0; r1.x &= m1; r1.y &= m1; r1.z &= m1; r1.w &= m1;
            r2.x &= m2; r2.y &= m2; r2.z &= m2; r2.w &= m2; r3.x &= m3; r3.y &= m3; r3.z &= m3; r3.w &= m3;
            *reinterpret_cast<uint4*>(smem + piece1_dst(0, buf)) = r0; *reinterpret_cast<uint4*>(smem + piece1_dst(1, buf)) = r1;
            *reinterpret_cast<uint4*>(smem + piece1_dst(2, buf)) = r2; *reinterpret_cast<uint4*>(smem + piece1_dst(3, buf)) = r3;
            V4_STORE_W(buf, wa);
            __syncthreads();
            if (c2 + 1 < nchunks2) V4_SC_LOAD(c2 + 1)
            const char* ha_ = smem + buf * HALO_BYTES + HPITCH;               // centre tap: row shift 1, column shift 1 (a_dx[1])
            const char* wb2_ = smem + buf * W_BYTES;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) af[i][a] = MF::ld(ha_ + a_dx[1] + i * HPITCH + a * 16 * PXB);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) bf[j][bb] = MF::ld(wb2_ + b_0 + (2 * j + bb) * 16 * PXB);
            V4_MFMA(nchunks, 0)
        }
#undef V4_SC_LOAD
    }
    V4_STAMP(5)
    __syncthreads();                                         // the epilogue re-uses the LDS
    V4_STAMP(6)
#undef V4_MFMA
#undef V4_XF_PHASE
#undef V4_STORE_W

    // ------------------------------ epilogue: per-wave LDS transpose, 16-byte I/O ------------------------------------------
    // VALU diet (the epilogue used to be 47 % of the kernel's VALU instructions, all of them outside the MFMAs' shadow):
    // every global access is a buffer load / store with a 32-bit offset = lane-constant part + wave-uniform part (no
    // per-pass 64-bit address arithmetic; tiles are always full here: H % 16 == 0, W % 32 == 0), the bias already sits in
    // the accumulators, the multiply by out_scale is skipped when it is 1, and the GroupNorm partial sums are taken from the
    // fp32 values (of which the stored ones are the roundings) instead of re-expanding the packed result.
    constexpr bool ATID = true;                              // lane-linear staging of half rounds by ds_write_addtid_b32 (v5_stage8), every form
    constexpr int STG_WAVE = V5_STG_BYTES;                   // 8,448 B per wave and half round
    constexpr int CH = 16 / (int)sizeof(TOUT);
    constexpr int CPR = BN / CH;                             // 16-byte chunks per pixel row: 16 (bf16) / 32 (fp32)
    constexpr int QN = 32 * CPR / 64;                        // passes per round: 8 / 16
    constexpr int PPP = 64 / CPR;                            // pixels per pass: 4 / 2
    float* const stg = reinterpret_cast<float*>(smem + wave * STG_WAVE);
    float* const red = reinterpret_cast<float*>(smem + 8 * STG_WAVE);     // [8 waves][BN][2]
    const int ch = lane % CPR;
    const int co0 = n0 + ch * CH;
    const bool cok = co0 < p.Cout;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);  // provably uniform: the soffsets below must live in SGPRs
    const size_t img_elems = (size_t)p.H * p.W * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((TOUT*)p.out + (size_t)b * img_elems, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<TOUT*>((const TOUT*)p.res) + (size_t)b * img_elems, 0, 0x7fffffff, 0x00020000);
    const unsigned voff = cok ? (unsigned)(((lane / CPR) * p.Cout + co0) * (int)sizeof(TOUT)) : 0x80000000u;   // beyond Cout: out of the descriptor's range
    const unsigned pass_b = (unsigned)(PPP * p.Cout) * (unsigned)sizeof(TOUT);      // bytes between passes
    // Round 6: the epilogue is specialised at compile time (EPI >= 0: bit 0 residual, bit 1 out_scale, bit 2 Combine) and branch-free.
    // With run-time flags every pass of a round was its own chain of basic blocks - ds_read, wait, (branch) add, (branch) multiply,
    // (branch) Combine, (exec mask) pack + store - so each of the 16 passes of a wave paid the LDS latency and its own dependent chain
    // with nothing of the next pass behind it (disassembly: four branches per pass).  Now a round reads ALL its pieces back first, the
    // next round's staging stores and residual loads are issued behind those reads (DS operations of one wave execute in order), and the
    // passes are straight-line code; lanes whose channels lie beyond Cout carry an out-of-range buffer offset (stores dropped, loads 0)
    // instead of an exec mask.  EPI < 0 keeps the run-time flags (fp32 parity kernels: compile time).
    constexpr bool EPI_RT = EPI < 0;
    const bool has_res = EPI_RT ? p.res != nullptr : (EPI & 1) != 0;
    const bool has_scale = EPI_RT ? p.out_scale != 1.f : (EPI & 2) != 0;
    const bool has_pyr = EPI_RT ? p.pyr != nullptr : (EPI & 4) != 0;
    constexpr bool HOIST_W4 = sizeof(TOUT) == 2;             // fp32 parity kernels: no registers to spare, in-loop loads
    // Combine ('sum') weights of this lane's channels: loop-invariant, fetched once (they were re-read per pixel piece)
    float4 w4r[CH]; float b4r[CH];
    if (HOIST_W4 && has_pyr && cok) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { w4r[c] = *reinterpret_cast<const float4*>(p.w4 + (size_t)(co0 + c) * 4); b4r[c] = p.b4[co0 + c]; }
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) { w4r[c] = make_float4(0.f, 0.f, 0.f, 0.f); b4r[c] = 0.f; }
    }
    f32x2 st_s2[CH / 2], st_q2[CH / 2];                      // running (sum, sum of squares) of this lane's channels, as pairs
#pragma unroll
    for (int k = 0; k < CH / 2; ++k) { st_s2[k] = (f32x2){0.f, 0.f}; st_q2[k] = (f32x2){0.f, 0.f}; }
    // Read-back of pass q of a half round (pixel column 4 q + lane / 16 of the half, channels 8 ch ... 8 ch + 7): the half's register (jb, r) is the
    // lane-linear row [lane group g' = 16 floats][logical column n] at float (jb * 4 + r) * 64 + 8 jb (v5_stage8); lane group g' holds pixel
    // quad Q(g') = (0, 2, 3, 1), so pass q reads g' = (0, 3, 1, 2)[q], r = lane / 16; physical channels 0-3 / 4-7 / 8-11 / 12-15 of a group of 16
    // sit at logical columns 0 / 12 / 4 / 8 (pi): the two 16-byte halves of a lane's 8 channels are two separate pieces of the row.  The shift
    // by 2 jb bank quads makes a ds_read_b128's 16-lane groups hit 16 distinct quads.
    const float* const stg_rd0 = stg + ((ch >> 1) * 4 + (lane >> 4)) * 64 + 8 * (ch >> 1) + ((ch & 1) ? 4 : 0);
    const float* const stg_rd1 = stg + ((ch >> 1) * 4 + (lane >> 4)) * 64 + 8 * (ch >> 1) + ((ch & 1) ? 8 : 12);
    auto stg_rd_ptr = [&](int q, int c4) -> const float* { const int ql = q & 3; return (c4 ? stg_rd1 : stg_rd0) + 16 * (ql == 0 ? 0 : ql == 1 ? 3 : ql == 2 ? 1 : 2); };
    const unsigned stg_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(wave_u * STG_WAVE);
    auto row_bytes = [&](int i) -> unsigned {                // byte offset of this wave's tile row i inside the image (uniform)
        return (unsigned)(((ty0 + wave_u * 2 + i) * p.W + tx0) * p.Cout) * (unsigned)sizeof(TOUT);
    };
    auto res_load = [&](int i, uint4 (&rv)[QN]) {
        const unsigned row_b = row_bytes(i);
#pragma unroll
        for (int q = 0; q < QN; ++q)
            rv[q] = (V4_ABL & 8192) ? make_uint4(q, q, q, q) : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff + row_b + (unsigned)q * pass_b, 0, V4_AUX_RES));
    };
    // passes read back / finished together: all of a round (specialised forms), a quarter of a round with the Combine set (its weights take 40
    // registers), one at a time with run-time flags (fp32 parity kernels: 16 passes of a round would not fit the register file)
    constexpr int NQ = QN / 2;                               // the four passes of a half round
    auto stage_write_half = [&](int i, int hb) {             // pixel columns 16 hb ... 16 hb + 15 of tile row i = accumulators (i, a = hb, *, *)
        v5_stage8<0>(stg_lds, acc[i][hb][0][0], acc[i][hb][0][1]); v5_stage8<1>(stg_lds, acc[i][hb][1][0], acc[i][hb][1][1]);
        v5_stage8<2>(stg_lds, acc[i][hb][2][0], acc[i][hb][2][1]); v5_stage8<3>(stg_lds, acc[i][hb][3][0], acc[i][hb][3][1]);
    };
    auto stage_read = [&](int q0, f32x4 (&t)[NQ][CH / 4]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) t[q][c4] = *reinterpret_cast<const f32x4*>(stg_rd_ptr(q0 + q, c4));
    };
    auto finish = [&](int i, int q0, const f32x4 (&t)[NQ][CH / 4], const uint4 (&rv)[QN]) {
        const int gy = ty0 + wave_u * 2 + i;                 // this round's tile row (uniform)
        const unsigned row_b = row_bytes(i);
        float4 pq[NQ];
        if (has_pyr) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const size_t pix = (size_t)(b * p.H + gy) * p.W + tx0 + ((q0 + q) * 64 + lane) / CPR;
                pq[q] = *reinterpret_cast<const float4*>(p.pyr + pix * 4);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // arithmetic on channel PAIRS, spelled as 2-vectors: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 are the IEEE operations of the
            // scalar forms at half the instruction count (left to the SLP vectoriser the statistics came out scalar once the pieces
            // went through the asm pin above)
            f32x2 v2[CH / 2];
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) { v2[c4 * 2] = (f32x2){t[q][c4].x, t[q][c4].y}; v2[c4 * 2 + 1] = (f32x2){t[q][c4].z, t[q][c4].w}; }
            if (has_res) {
                float rvf[CH];
                Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&rv[q0 + q]), rvf);
#pragma unroll
                for (int k = 0; k < CH / 2; ++k) v2[k] += (f32x2){rvf[2 * k], rvf[2 * k + 1]};
            }
            if (has_scale) {
#pragma unroll
                for (int k = 0; k < CH / 2; ++k) v2[k] *= (f32x2){p.out_scale, p.out_scale};
            }
            float v[CH];
#pragma unroll
            for (int k = 0; k < CH / 2; ++k) { v[2 * k] = v2[k].x; v[2 * k + 1] = v2[k].y; }
            if (has_pyr) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float4 wq = HOIST_W4 ? w4r[c] : (cok ? *reinterpret_cast<const float4*>(p.w4 + (size_t)(co0 + c) * 4) : make_float4(0.f, 0.f, 0.f, 0.f));
                    v[c] += (HOIST_W4 ? b4r[c] : (cok ? p.b4[co0 + c] : 0.f)) + wq.x * pq[q].x + wq.y * pq[q].y + wq.z * pq[q].z + wq.w * pq[q].w;
                }
            }
            if (!EPI_RT || cok) {                            // (specialised forms: no exec mask - lanes beyond Cout store out of range)
            const uint4 packed = Vec16<TOUT>::pack(v);
            if (V4_ABL & 1024) asm volatile("" :: "v"(packed.x), "v"(packed.y), "v"(packed.z), "v"(packed.w));
            else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, packed), rs_out,
                                                   voff + row_b + (unsigned)(q0 + q) * pass_b, 0, V4_AUX_OUT);
            if (!(V4_ABL & 4096)) {
#pragma unroll
                for (int k = 0; k < CH / 2; ++k) {
                    const f32x2 x = {v[2 * k], v[2 * k + 1]};
                    st_s2[k] += x; st_q2[k] = __builtin_elementwise_fma(x, x, st_q2[k]);
                }
            }
            }
        }
    };
    {
        uint4 rv0[QN], rv1[QN];
        f32x4 t[NQ][CH / 4];
        if (has_res) res_load(0, rv0);
        if constexpr (ATID) {
            // Left alone, LLVM puts every read back in front of its pass and pulls the passes' arithmetic up between the reads (pure
            // arithmetic is ordered by nothing - not by sched_barrier, not by a memory clobber).  The pieces therefore pass through an
            // empty volatile asm as in/out operands: everything computed from them follows it, all reads precede it.  A half round's
            // staging stores are issued behind the previous half round's reads (DS operations of one wave execute in order).
            static_assert(NQ * (CH / 4) == 8 && QN == 8, "8 pieces per half round");
            auto pin = [&]() {
                asm volatile("" : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]), "+v"(t[3][0]), "+v"(t[3][1]) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            auto fence = [&]() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); };
            constexpr bool RV2 = !(EPI & 4);                 // with the Combine set (40 registers of weights) the second tile row's residual re-uses rv0
            stage_write_half(0, 0);
            stage_read(0, t); if (has_res && RV2) res_load(1, rv1); pin();
            stage_write_half(0, 1); fence();
            finish(0, 0, t, rv0); fence();
            stage_read(NQ, t); pin();
            stage_write_half(1, 0); fence();
            finish(0, NQ, t, rv0);
            V4_STAMP(7)
            fence();
            if (has_res && !RV2) res_load(1, rv0);
            stage_read(0, t); pin();
            stage_write_half(1, 1); fence();
            finish(1, 0, t, RV2 ? rv1 : rv0); fence();
            stage_read(NQ, t); pin();
            finish(1, NQ, t, RV2 ? rv1 : rv0);
        }
        V4_STAMP(7)
    }
    float st_s[CH], st_q[CH];
#pragma unroll
    for (int k = 0; k < CH / 2; ++k) { st_s[2 * k] = st_s2[k].x; st_s[2 * k + 1] = st_s2[k].y; st_q[2 * k] = st_q2[k].x; st_q[2 * k + 1] = st_q2[k].y; }
    if ((V4_ABL & 65536)) { float t = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) t += st_s[c] + st_q[c];
        asm volatile("" :: "v"(t)); }
    if ((p.stats || p.stats_part) && !(V4_ABL & (4096 | 65536))) {
        // lanes holding the same 16-byte channel chunk are CPR apart inside a wave
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (CPR == 16) { st_s[c] = reduce_lanes_stride<16>(st_s[c]); st_q[c] = reduce_lanes_stride<16>(st_q[c]); }
            else { st_s[c] += __shfl_xor(st_s[c], 32); st_q[c] += __shfl_xor(st_q[c], 32); }
        }
        if (lane < CPR) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                red[(wave * BN + ch * CH + c) * 2] = st_s[c]; red[(wave * BN + ch * CH + c) * 2 + 1] = st_q[c];
            }
        }
        // (not __syncthreads(): its release fence waits for the acknowledgement of this wave's 16 output stores - 2-3 k cycles in which
        // the reduction and the atomics below can already run; only the LDS writes above have to have landed)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < BN) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (V4_ABL & 32768) asm volatile("" :: "v"(s), "v"(q));
            else if (co < p.Cout) {
                if (p.stats_part) {                          // this workgroup's partial totals, plain 16-byte store (no queue on the item's totals)
                    long long* d = p.stats_part + (((size_t)b * gridDim.x + blockIdx.x) * p.Cout + co) * 2;
                    *reinterpret_cast<longlong2*>(d) = make_longlong2(__float2ll_rn(s * GN_SUM_SCALE), __float2ll_rn(q * GN_SQ_SCALE));
                } else gn_accumulate(p.stats + ((size_t)b * p.Cout + co) * 2, s, q);
            }
        }
    }
    V4_STAMP(8)
#ifdef USE_HIP_TRACE_BUILD
    if (tracing) {
        for (int i = 0; i < 2 * trace_n; ++i) p.trace[(wave >> 2) * 256 + i] = trace_lds[i];
    }
#endif
}

template <typename TIN, typename TOUT, int CK, bool ACT, int EPI>
static void v5_launch_e(const ConvArgs& a, hipStream_t s) {
    constexpr int MAIN = 2 * V4_HH * 48 * 64 + 2 * V4_BN * 64 + 512 * 8 + 2 * 5 * 512 * 4;   // halo + weight buffers, GroupNorm table, piece tables
    constexpr int EPIB = 8 * V5_STG_BYTES + 8 * V4_BN * 2 * 4;
#ifdef USE_HIP_TRACE_BUILD
    constexpr int SMEM = 152064 + 2 * 248 * 8;               // + the stamp buffers
#else
    constexpr int SMEM = MAIN > EPIB ? MAIN : EPIB;
#endif
    static_assert(MAIN <= 152064 && EPIB <= 152064 && SMEM <= 163840, "LDS budget");
    static LdsAttrOnce attr;                                 // per (instantiation, device)
    auto kern = conv_v5_kernel<TIN, TOUT, CK, ACT, EPI>;
    attr(kern, SMEM);
    dim3 grid(conv_v4_tiles(a.H, a.W), (a.Cout + V4_BN - 1) / V4_BN, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(512), SMEM, s, a);
}

// epilogue specialisation as in conv_v4 (EPI: bit 0 residual, bit 1 out_scale, bit 2 Combine)
template <typename T, bool ACT>
static void v5_launch_t(const ConvArgs& a, hipStream_t s) {
    const bool res = a.res != nullptr, scale = a.out_scale != 1.f, pyr = a.pyr != nullptr;
    if (pyr) { res ? v5_launch_e<T, T, 32, ACT, 7>(a, s) : v5_launch_e<T, T, 32, ACT, 6>(a, s); }
    else if (res) v5_launch_e<T, T, 32, ACT, 3>(a, s);
    else if (scale) v5_launch_e<T, T, 32, ACT, 2>(a, s);
    else v5_launch_e<T, T, 32, ACT, 0>(a, s);
}

static int g_conv_v5 = 1;              // use_set_option("conv_v5", 0): conv_v4 (32x32x16 MFMAs) for the 16-bit types as well
void conv_v5_set(int on) { g_conv_v5 = on; }
bool conv_v5_enabled(const ConvArgs& a) { return g_conv_v5 != 0 && a.in_dtype != DT_F32; }   // (a launch conv_v4_eligible has accepted)

void launch_conv_v5(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
#ifdef USE_HIP_TRACE_BUILD
    if (getenv("USE_HIP_TRACE")) {           // bring-up only: cycle stamps of one workgroup (see launch_conv_v4)
        static int printed = 0;
        static unsigned long long* trace_buf = nullptr;
        if (!trace_buf) (void)hipMalloc((void**)&trace_buf, 512 * 8);
        static int skip = getenv("USE_HIP_TRACE_SKIP") ? atoi(getenv("USE_HIP_TRACE_SKIP")) : 0;
        const bool match = a.H == 512 && a.C0 + a.C1 == atoi(getenv("USE_HIP_TRACE")) && a.in_dtype == DT_BF16 &&
                           (!getenv("USE_HIP_TRACE_RES") || (a.res != nullptr) == (atoi(getenv("USE_HIP_TRACE_RES")) != 0));
        if (!printed && match && skip-- <= 0) {
            (void)hipMemsetAsync(trace_buf, 0, 512 * 8, s);
            a.trace = trace_buf;
            if (getenv("USE_HIP_TRACE_WG")) a.dbg = atoi(getenv("USE_HIP_TRACE_WG"));
            a.act ? v5_launch_t<__bf16, true>(a, s) : v5_launch_t<__bf16, false>(a, s);
            (void)hipStreamSynchronize(s);
            unsigned long long hbuf[512];
            (void)hipMemcpy(hbuf, trace_buf, sizeof hbuf, hipMemcpyDeviceToHost);
            for (int g = 0; g < 2; ++g) {
                unsigned long long prev = hbuf[g * 256 + 1];
                for (int i = 0; i < 120 && hbuf[g * 256 + 2 * i]; ++i) {
                    fprintf(stderr, "[trace v4 G%d] id %3llu  +%6llu\n", g, hbuf[g * 256 + 2 * i], hbuf[g * 256 + 2 * i + 1] - prev);
                    prev = hbuf[g * 256 + 2 * i + 1];
                }
            }
            printed = 1;
            a.trace = nullptr; a.dbg = a0.dbg;
        }
    }
#endif
    if (a.in_dtype == DT_BF16) { a.act ? v5_launch_t<__bf16, true>(a, s) : v5_launch_t<__bf16, false>(a, s); }
    else                       { a.act ? v5_launch_t<_Float16, true>(a, s) : v5_launch_t<_Float16, false>(a, s); }
}

}  // namespace use
