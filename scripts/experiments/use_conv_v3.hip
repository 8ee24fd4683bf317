// conv_v3_kernel: wave-specialised implicit-GEMM 3x3 convolution (same arithmetic, tile and argument struct as
// conv_v2_kernel; see use_kernels.h).  One 512-thread workgroup per CU computes a 16x16-pixel x 128-channel tile:
//
//   waves 0-3  "MFMA waves"   one per SIMD; each owns 64 pixels x all 128 output channels (2x4 MFMA 32x32 tiles, 128
//                             accumulator VGPRs).  Per (tap, chunk) iteration: 24 ds_read_b128 + 32 MFMAs, fragments
//                             double-buffered per k-step, the first k-step of the next iteration prefetched across the
//                             barrier.  No VALU / global-memory work at all.
//   waves 4-7  "helper waves" one per SIMD; all staging: weight slabs global -> registers -> a 3-deep LDS ring (two
//                             iterations ahead of their use), halo pieces of the next 64-channel chunk global ->
//                             registers -> GroupNorm affine + SiLU -> the other halo buffer.
//
// One s_barrier per iteration.  The MFMA pipe of a SIMD is fed by a single wave that never waits on anything but its
// own LDS reads; the helper wave's ~80 instructions per iteration issue in the gaps.
#include "use_kernels.h"
#include "use_device.h"

#include <cstdio>
#include <cstdlib>

namespace use {

constexpr int V3_T = 16, V3_HE = 18, V3_BN = 128;


template <typename TIN, bool ACT>
DEVI uint4 v3_transform(const uint4 raw, const unsigned mask, const float (&ca)[16 / sizeof(TIN)],
                        const float (&cb)[16 / sizeof(TIN)]) {
    constexpr int VEC = 16 / sizeof(TIN);

    float v[VEC];
    Vec16<TIN>::load(reinterpret_cast<const TIN*>(&raw), v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        v[k] = fmaf(v[k], ca[k], cb[k]);
        if (ACT) {
            if (sizeof(TIN) == 4) v[k] = v[k] / (1.0f + expf(-v[k]));
            else v[k] = v[k] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[k] * -1.44269504088896341f));
        }
    }
    uint4 o = Vec16<TIN>::pack(v);
    o.x &= mask; o.y &= mask; o.z &= mask; o.w &= mask;
    return o;
}

template <typename TIN, typename TOUT, int CK, bool ACT>
__global__ __launch_bounds__(512) void conv_v3_kernel(ConvArgs p) {
    typedef Mfma<TIN> MF;
    constexpr int VEC = 16 / sizeof(TIN);
    constexpr int PARTS = CK / VEC;                          // 8
    constexpr int ROWB = CK * (int)sizeof(TIN) + 16;         // 144
    constexpr int BN = V3_BN, TM = 2, TN = 4;
    constexpr int KSTEPS = CK / MF::KM;
    constexpr int KB = MF::KM * (int)sizeof(TIN);            // bytes per k-step in a row
    constexpr int HPITCH = (V3_HE * ROWB / 16 + 15) / 16 * 16 * 16;
    constexpr int HALO_BYTES = V3_HE * HPITCH, W_BYTES = BN * ROWB;
    constexpr int RING0 = 2 * HALO_BYTES;                    // 3 weight slabs follow the 2 halo buffers
    constexpr int DUMMY0 = RING0 + 3 * W_BYTES;              // end of the staging buffers (trace builds keep their stamps here)
    (void)DUMMY0;
    constexpr int NPIECE = V3_HE * V3_HE * PARTS;            // 2592 16-byte pieces per halo chunk
    constexpr int HP = (NPIECE + 255) / 256;                 // 11 pieces per helper thread per chunk
    static_assert(PARTS == 8 && HP == 11 && KSTEPS % 2 == 0, "v3 staging layout");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool helper = wave >= 4;
    const int htid = tid & 255;
    const int b = blockIdx.z;
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous bands
    const int tiles_x = (p.W + V3_T - 1) / V3_T;
    const int ty0 = (tile / tiles_x) * V3_T, tx0 = (tile % tiles_x) * V3_T;
    const int n0 = blockIdx.y * BN;
    const int Ctot = p.C0 + p.C1, nchunks = Ctot / CK;
    const int XCtot = p.XC0 + p.XC1, nchunks2 = XCtot / CK;
    const int part = tid & (PARTS - 1);

    // per-lane epilogue constants of the MFMA waves (bias + time-embedding bias), fetched first
    float addv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + j * 32 + (lane & 31);
        float add = 0.f;
        if (co < p.Cout) {
            if (p.bias) add += p.bias[co];
            if (p.temb) add += p.temb[(size_t)b * p.temb_bstride + co];
        }
        addv[j] = add;
    }

    auto src_ptr0 = [&](int chunk, int pixoff) -> const uint4* {
        const int c_glob = chunk * CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.C0) { src = (const TIN*)p.src0; Cs = p.C0; c_loc = c_glob; }
        else               { src = (const TIN*)p.src1; Cs = p.C1; c_loc = c_glob - p.C0; }
        return reinterpret_cast<const uint4*>(src + (size_t)pixoff * Cs + c_loc + part * VEC);
    };
    float ca[VEC], cb[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { ca[k] = 1.f; cb[k] = 0.f; }
    auto load_coef = [&](int chunk) {
        if (p.coef) {
            const float* cf = p.coef + ((size_t)b * Ctot + chunk * CK + part * VEC) * 2;
#pragma unroll
            for (int k = 0; k < VEC; ++k) { ca[k] = cf[2 * k]; cb[k] = cf[2 * k + 1]; }
        }
    };
    // halo piece idx (0..2591) -> (image pixel index or 0, validity mask, LDS offset inside a halo buffer)
    auto piece_geom = [&](int idx, int& pix, unsigned& mask, int& dst) {
        const int hp = idx / PARTS;
        const int hy = hp / V3_HE, hx = hp - hy * V3_HE;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        pix = inb ? (b * p.H + gy) * p.W + gx : 0;
        mask = inb ? 0xffffffffu : 0u;
        dst = hy * HPITCH + hx * ROWB + part * 16;
    };
    const size_t wrstride = (size_t)9 * Ctot;                 // weight rows (cout) are 9*Ctot elements apart
    const TIN* const wseg0 = (const TIN*)p.w + (size_t)n0 * wrstride;

    // ---------------------------------------------- prologue (all 512 threads) --------------------------------------
    {
        load_coef(0);
        uint4 raw[6]; int pdst6[6]; unsigned pm6[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int idx = j * 512 + tid;
            int pix = 0; pm6[j] = 0; pdst6[j] = -1;
            if (idx < NPIECE) piece_geom(idx, pix, pm6[j], pdst6[j]);
            raw[j] = *src_ptr0(0, pix);
        }
        uint4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                         // slabs of iterations 0 (q<2) and 1 (q>=2): 2 pieces each
            const int it = q >> 1, row = (tid + (q & 1) * 512) / PARTS;
            const bool ok = it < nchunks * 9;
            w[q] = ok ? *reinterpret_cast<const uint4*>(wseg0 + (size_t)row * wrstride + (size_t)it * Ctot + part * VEC)
                      : make_uint4(0, 0, 0, 0);               // (chunk 0: tap == it)
        }
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (pdst6[j] >= 0) *reinterpret_cast<uint4*>(smem + pdst6[j]) = v3_transform<TIN, ACT>(raw[j], pm6[j], ca, cb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = q >> 1, row = (tid + (q & 1) * 512) / PARTS;
            *reinterpret_cast<uint4*>(smem + RING0 + it * W_BYTES + row * ROWB + part * 16) = w[q];
        }
    }

    // ---------------------------------------------- role state ---------------------------------------------------------
    // helper: this thread's 11 halo pieces of every chunk and its 4 weight pieces of every slab
    int ppix[HP], pdst[HP]; unsigned pmask[HP];
    const uint4 z4 = make_uint4(0, 0, 0, 0);
    uint4 hP0 = z4, hP1 = z4, hP2 = z4, hP3 = z4, hP4 = z4, hP5 = z4;      // halo pieces in flight (slot = piece % 6)
    uint4 wA0 = z4, wA1 = z4, wA2 = z4, wA3 = z4, wB0 = z4, wB1 = z4, wB2 = z4, wB3 = z4, wC0 = z4, wC1 = z4, wC2 = z4,
          wC3 = z4;                                                         // weight slabs in flight (set = tap % 3)
    unsigned wofs[4]; int wdst[4];
    if (helper) {
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const int idx = j * 256 + htid;
            ppix[j] = 0; pmask[j] = 0; pdst[j] = -1;
            if (idx < NPIECE) piece_geom(idx, ppix[j], pmask[j], pdst[j]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = (htid + q * 256) / PARTS;
            wofs[q] = (unsigned)(row * wrstride + part * VEC);
            wdst[q] = RING0 + row * ROWB + part * 16;
        }
    }
    // MFMA wave: accumulators (zeroed inside the role branches so that they are not live across the helper loop),
    // fragment registers (two k-step sets), LDS fragment bases
    f32x16 acc[TM][TN];
#define V3_ZERO_ACC()                                                                                                \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;                                       \
    }
    typename MF::frag fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    int a_base[TM], b_base[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = (wave & 3) * 64 + i * 32 + (lane & 31);
        a_base[i] = (m >> 4) * HPITCH + (m & 15) * ROWB + (lane >> 5) * MF::KPL * (int)sizeof(TIN);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
        b_base[j] = RING0 + (j * 32 + (lane & 31)) * ROWB + (lane >> 5) * MF::KPL * (int)sizeof(TIN);

#ifdef USE_HIP_TRACE_BUILD
    const bool tracing = p.trace != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave & 3) == 0;
    int trace_n = 0;
    // stamps go to the LDS tail (beyond every staging buffer) so that they stay off the vector-memory counter
    unsigned long long* const trace_lds = reinterpret_cast<unsigned long long*>(smem + DUMMY0) + (wave >> 2) * 256;
#define V3_STAMP(ID)                                                                                   \
    if (tracing && trace_n < 127) {                                                                    \
        trace_lds[2 * trace_n] = (unsigned long long)(ID);                                             \
        trace_lds[2 * trace_n + 1] = __builtin_readcyclecounter(); ++trace_n;                          \
    }
#else
#define V3_STAMP(ID)
#endif
    // ---- helper macros (T is a literal) ---------------------------------------------------------------------------------
    // weights of seg-0 iteration (chunk CC, tap TT >= 0, possibly > 8) -> register set S (A/B/C)
#define V3_LOAD_W(S, CC, TT)                                                                                         \
    {                                                                                                                \
        const int cw_ = (TT) > 8 ? (CC) + 1 : (CC);                                                                  \
        const int tw_ = (TT) > 8 ? (TT)-9 : (TT);                                                                    \
        if (cw_ < nchunks) {                                                                       \
            const TIN* wb_ = wseg0 + (size_t)tw_ * Ctot + cw_ * CK;                                                  \
            w##S##0 = *reinterpret_cast<const uint4*>(wb_ + wofs[0]); w##S##1 = *reinterpret_cast<const uint4*>(wb_ + wofs[1]); \
            w##S##2 = *reinterpret_cast<const uint4*>(wb_ + wofs[2]); w##S##3 = *reinterpret_cast<const uint4*>(wb_ + wofs[3]); \
        }                                                                                                            \
    }
#define V3_STORE_W(S, SLOT)                                                                                          \
    {                                                                                                                \
        *reinterpret_cast<uint4*>(smem + (SLOT)*W_BYTES + wdst[0]) = w##S##0;                                        \
        *reinterpret_cast<uint4*>(smem + (SLOT)*W_BYTES + wdst[1]) = w##S##1;                                        \
        *reinterpret_cast<uint4*>(smem + (SLOT)*W_BYTES + wdst[2]) = w##S##2;                                        \
        *reinterpret_cast<uint4*>(smem + (SLOT)*W_BYTES + wdst[3]) = w##S##3;                                        \
    }
#define V3_STORE_W3(T, SLOT) { if ((T) % 3 == 0) { V3_STORE_W(A, SLOT) } else if ((T) % 3 == 1) { V3_STORE_W(B, SLOT) } else { V3_STORE_W(C, SLOT) } }
    // Halo pieces of chunk n are written into halo buffer n&1 during chunk n-1.  Piece j (0..10) is consumed (affine +
    // SiLU + ds_write) at tap CT(j) = 1,1,2,3,4,5,5,6,7,8,8 of chunk n-1 and its global load is issued three taps earlier
    // (taps 7, 7, 8 of chunk n-2 for the first three), into register slot j % 6.  All 4 helper waves reach their memory
    // instructions at the same time after a barrier and the CU's one texture addresser takes a 1 KiB wave-load every 16
    // cycles, so every memory instruction is paired with a slice of VALU work ("slot") instead of issuing back to back.
    // Beyond the last chunk the loads are clamped to it and the stores land in buffers nobody reads any more.
    float tv[VEC], ev[VEC];
#define V3_PIECE_LOAD(J, CH)                                                                                         \
    {                                                                                                                \
        if ((J) % 6 == 0) hP0 = *src_ptr0(CH, ppix[J]); else if ((J) % 6 == 1) hP1 = *src_ptr0(CH, ppix[J]);         \
        else if ((J) % 6 == 2) hP2 = *src_ptr0(CH, ppix[J]); else if ((J) % 6 == 3) hP3 = *src_ptr0(CH, ppix[J]);    \
        else if ((J) % 6 == 4) hP4 = *src_ptr0(CH, ppix[J]); else hP5 = *src_ptr0(CH, ppix[J]);                      \
    }
#define V3_PIECE_REG(J) ((J) % 6 == 0 ? hP0 : (J) % 6 == 1 ? hP1 : (J) % 6 == 2 ? hP2 : (J) % 6 == 3 ? hP3 : (J) % 6 == 4 ? hP4 : hP5)
    // the transform of piece J in four slices
#define V3_STG1(J)                                                                                                   \
    if ((J) >= 0) {                                                                                                  \
        const uint4 raw_ = V3_PIECE_REG((J) >= 0 ? (J) : 0);                                                         \
        Vec16<TIN>::load(reinterpret_cast<const TIN*>(&raw_), tv);                                                   \
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) tv[k] = fmaf(tv[k], ca[k], cb[k]);                           \
    }
#define V3_STG2(J)                                                                                                   \
    if ((J) >= 0 && ACT) {                                                                                           \
        _Pragma("unroll") for (int k = 0; k < VEC; ++k)                                                              \
            ev[k] = sizeof(TIN) == 4 ? expf(-tv[k]) : __builtin_amdgcn_exp2f(tv[k] * -1.44269504088896341f);         \
    }
#define V3_STG3(J)                                                                                                   \
    if ((J) >= 0 && ACT) {                                                                                           \
        _Pragma("unroll") for (int k = 0; k < VEC; ++k)                                                              \
            ev[k] = sizeof(TIN) == 4 ? 1.0f / (1.0f + ev[k]) : __builtin_amdgcn_rcpf(1.0f + ev[k]);                  \
    }
#define V3_STG4(J)                                                                                                   \
    if ((J) >= 0) {                                                                                                  \
        constexpr int j_ = (J) >= 0 ? (J) : 0;                                                                       \
        if (ACT) { _Pragma("unroll") for (int k = 0; k < VEC; ++k) tv[k] *= ev[k]; }                                 \
        uint4 o_ = Vec16<TIN>::pack(tv);                                                                             \
        o_.x &= pmask[j_]; o_.y &= pmask[j_]; o_.z &= pmask[j_]; o_.w &= pmask[j_];                                  \
        if (j_ < HP - 1 || pdst[j_] >= 0) *reinterpret_cast<uint4*>(smem + hbw_ + pdst[j_]) = o_;                    \
    }
#define V3_LW(T, Q)                                                                                                  \
    {                                                                                                                \
        if ((T) % 3 == 0) wA##Q = *reinterpret_cast<const uint4*>(wb_ + wofs[Q]);                                    \
        else if ((T) % 3 == 1) wB##Q = *reinterpret_cast<const uint4*>(wb_ + wofs[Q]);                               \
        else wC##Q = *reinterpret_cast<const uint4*>(wb_ + wofs[Q]);                                                 \
    }
#define V3_PL(J, CH) if ((J) >= 0) V3_PIECE_LOAD((J) >= 0 ? (J) : 0, CH)
#define V3_FENCE() __builtin_amdgcn_sched_barrier(0);
    // tap T of chunk CC: consume pieces CA, CB (of chunk CC+1); load pieces LA, LB of chunk CC+LC (-1: none)
#define V3_HELPER(CC, T, CA, CB, LA, LB, LC)                                                                         \
    {                                                                                                                \
        const int cc_ = (CC);                                                                                        \
        const int hbw_ = ((cc_ + 1) & 1) * HALO_BYTES;                                                               \
        const int chl_ = min(cc_ + (LC), nchunks - 1);                                                               \
        const int cw_ = min((T) + 4 > 8 ? cc_ + 1 : cc_, nchunks - 1);                                               \
        const TIN* const wb_ = wseg0 + (size_t)((T) + 4 > 8 ? (T)-5 : (T) + 4) * Ctot + cw_ * CK;                    \
        /* weights of iteration it+2 (loaded two taps ago, set (T+1)%3) -> ring slot (T+2)%3; fetch those of it+4 */ \
        V3_STORE_W3((T) + 1, ((T) + 2) % 3)                                                                          \
        V3_FENCE() V3_STAMP(3)                                                                                       \
        V3_LW(T, 0) V3_STG1(CA) V3_FENCE()                                                                           \
        V3_LW(T, 1) V3_STG2(CA) V3_FENCE()                                                                           \
        V3_LW(T, 2) V3_STG3(CA) V3_FENCE()                                                                           \
        V3_LW(T, 3) V3_STG4(CA) V3_FENCE() V3_STAMP(4)                                                               \
        V3_PL(LA, chl_) V3_STG1(CB) V3_FENCE()                                                                       \
        V3_PL(LB, chl_) V3_STG2(CB) V3_FENCE()                                                                       \
        V3_STG3(CB) V3_FENCE()                                                                                       \
        V3_STG4(CB)                                                                                                  \
        if ((T) == 0) load_coef(min(cc_ + 1, nchunks - 1));                                                          \
        V3_FENCE() V3_STAMP(5)                                                                                       \
    }
    // fragment reads of k-step KK of (halo buffer HB bytes, tap offset TAPOFF, ring slot SLOT) into set S (0/1)
#define V3_READ(S, HBOFF, TAPOFF, SLOT, KK)                                                                          \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa##S[i] = MF::ld(smem + (HBOFF) + (TAPOFF) + a_base[i] + (KK)*KB); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb##S[j] = MF::ld(smem + (SLOT)*W_BYTES + b_base[j] + (KK)*KB);  \
    }
#define V3_MMA(S)                                                                                                    \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(fa##S[i], fb##S[j], acc[i][j]);       \
    }
#define V3_TAPOFF(T) (((T) / 3) * HPITCH + ((T) % 3) * ROWB)
    // MFMA-wave iteration (chunk CC, tap T): set 0 already holds k-step 0 unless T == 0; the next tap's k-step 0 is
    // prefetched before the barrier unless T == 8 (the next chunk's halo buffer is still being written).
#define V3_CONSUMER(CC, T)                                                                                           \
    {                                                                                                                \
        const int hb_ = ((CC)&1) * HALO_BYTES;                                                                       \
        if ((T) == 0) V3_READ(0, hb_, V3_TAPOFF(0), 0, 0)                                                            \
        _Pragma("unroll") for (int kk = 0; kk < KSTEPS; kk += 2) {                                                   \
            V3_READ(1, hb_, V3_TAPOFF(T), (T) % 3, kk + 1)                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            V3_MMA(0)                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            if (kk + 2 < KSTEPS) V3_READ(0, hb_, V3_TAPOFF(T), (T) % 3, kk + 2)                                      \
            else if ((T) < 8) V3_READ(0, hb_, V3_TAPOFF((T) < 8 ? (T) + 1 : 0), ((T) + 1) % 3, 0)                     \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
            V3_MMA(1)                                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
    }
#define V3_BAR() { V3_STAMP(1) __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); V3_STAMP(2) }

    if (helper) {
        __builtin_amdgcn_s_setprio(2);                       // second-dispatched waves lose VALU arbitration otherwise
        V3_LOAD_W(B, 0, 2)                                    // weights of iterations 2 and 3, stored at taps 0 and 1
        V3_LOAD_W(C, 0, 3)
        { const int ch1_ = min(1, nchunks - 1); V3_PIECE_LOAD(0, ch1_) V3_PIECE_LOAD(1, ch1_) V3_PIECE_LOAD(2, ch1_) }
    }
    V3_BAR();
    if (helper) {
        for (int c = 0; c < nchunks; ++c) {
            V3_HELPER(c, 0, -1, -1, 3, -1, 1) V3_BAR();
            V3_HELPER(c, 1, 0, 1, 4, -1, 1) V3_BAR();
            V3_HELPER(c, 2, 2, -1, 5, 6, 1) V3_BAR();
            V3_HELPER(c, 3, 3, -1, 7, -1, 1) V3_BAR();
            V3_HELPER(c, 4, 4, -1, 8, -1, 1) V3_BAR();
            V3_HELPER(c, 5, 5, 6, 9, 10, 1) V3_BAR();
            V3_HELPER(c, 6, 7, -1, -1, -1, 1) V3_BAR();
            V3_HELPER(c, 7, 8, -1, 0, 1, 2) V3_BAR();
            V3_HELPER(c, 8, 9, 10, 2, -1, 2) V3_BAR();
        }
        V3_ZERO_ACC()
    } else {
        V3_ZERO_ACC()
        for (int c = 0; c < nchunks; ++c) {
#define V3_C(T) V3_CONSUMER(c, T) V3_BAR();
            V3_C(0) V3_C(1) V3_C(2) V3_C(3) V3_C(4) V3_C(5) V3_C(6) V3_C(7) V3_C(8)
#undef V3_C
        }
    }
#undef V3_HELPER
#undef V3_CONSUMER
#undef V3_LOAD_W
#undef V3_PIECE_LOAD
#undef V3_PIECE_REG
#undef V3_STORE_W3

    __builtin_amdgcn_s_setprio(0);
    // ---- segment 1: the fused 1x1 shortcut (raw centre pixels): all threads stage, MFMA waves compute -------------------
    if (nchunks2 > 0) {
        uint4 r0, r1, r2, r3, w0, w1; unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
        auto sc_load = [&](int c2) {
            const int c_glob = c2 * CK;
            const TIN* src; int Cs, c_loc;
            if (c_glob < p.XC0) { src = (const TIN*)p.x0; Cs = p.XC0; c_loc = c_glob; }
            else                { src = (const TIN*)p.x1; Cs = p.XC1; c_loc = c_glob - p.XC0; }
            auto one = [&](int q, uint4& r, unsigned& m) {
                const int pix = (q * 512 + tid) / PARTS;
                const int gy = ty0 + (pix >> 4), gx = tx0 + (pix & 15);
                const bool inb = gy < p.H && gx < p.W;
                const size_t po = inb ? (size_t)(b * p.H + gy) * p.W + gx : 0;
                r = *reinterpret_cast<const uint4*>(src + po * Cs + c_loc + part * VEC);
                m = inb ? 0xffffffffu : 0u;
            };
            one(0, r0, m0); one(1, r1, m1); one(2, r2, m2); one(3, r3, m3);
            const TIN* wb_ = (const TIN*)p.w2 + (size_t)n0 * XCtot + c2 * CK;
            w0 = *reinterpret_cast<const uint4*>(wb_ + (size_t)(tid / PARTS) * XCtot + part * VEC);
            w1 = *reinterpret_cast<const uint4*>(wb_ + (size_t)((tid + 512) / PARTS) * XCtot + part * VEC);
        };
        auto sc_dst = [&](int q, int buf) -> int {
            const int pix = (q * 512 + tid) / PARTS;
            return buf * HALO_BYTES + ((pix >> 4) + 1) * HPITCH + ((pix & 15) + 1) * ROWB + part * 16;
        };
        sc_load(0);
        for (int c2 = 0; c2 < nchunks2; ++c2) {
            const int buf = c2 & 1;
            r0.x &= m0; r0.y &= m0; r0.z &= m0; r0.w &= m0; r1.x &= m1; r1.y &= m1; r1.z &= m1; r1.w &= m1;
            r2.x &= m2; r2.y &= m2; r2.z &= m2; r2.w &= m2; r3.x &= m3; r3.y &= m3; r3.z &= m3; r3.w &= m3;
            *reinterpret_cast<uint4*>(smem + sc_dst(0, buf)) = r0; *reinterpret_cast<uint4*>(smem + sc_dst(1, buf)) = r1;
            *reinterpret_cast<uint4*>(smem + sc_dst(2, buf)) = r2; *reinterpret_cast<uint4*>(smem + sc_dst(3, buf)) = r3;
            *reinterpret_cast<uint4*>(smem + RING0 + buf * W_BYTES + (tid / PARTS) * ROWB + part * 16) = w0;
            *reinterpret_cast<uint4*>(smem + RING0 + buf * W_BYTES + ((tid + 512) / PARTS) * ROWB + part * 16) = w1;
            __syncthreads();
            if (c2 + 1 < nchunks2) sc_load(c2 + 1);
            if (!helper) {
                const int hb_ = buf * HALO_BYTES;
                V3_READ(0, hb_, V3_TAPOFF(4), buf, 0)
#pragma unroll
                for (int kk = 0; kk < KSTEPS; kk += 2) {
                    V3_READ(1, hb_, V3_TAPOFF(4), buf, kk + 1)
                    V3_MMA(0)
                    if (kk + 2 < KSTEPS) V3_READ(0, hb_, V3_TAPOFF(4), buf, kk + 2)
                    V3_MMA(1)
                }
            }
        }
        __syncthreads();
    }
#undef V3_READ
#undef V3_MMA
#undef V3_TAPOFF
#undef V3_STORE_W

    // ------------------------------ epilogue -----------------------------------------------------------------------------
    // MFMA waves write one 32-pixel x 128-channel fp32 slab each to LDS; then ALL 512 threads read 16-byte output chunks
    // back row-wise (residual add, scale, Combine, conversion, store, GroupNorm partial sums).
    constexpr int STG_LD = BN + 4;
    constexpr int STG_WAVE = 32 * STG_LD * 4;                // 16,896 B per MFMA wave
    constexpr int CH = 16 / (int)sizeof(TOUT);               // channels per chunk
    constexpr int CPR = BN / CH;                             // chunks per row: 16 (bf16) / 32 (fp32)
    constexpr int RPP = 512 / CPR;                           // rows per pass
    constexpr int QN = 128 / RPP;                            // passes per round (128 rows = 4 waves x 32)
    float* const red = reinterpret_cast<float*>(smem + 4 * STG_WAVE);     // [8 waves][BN][2]
    TOUT* out = (TOUT*)p.out;
    const TOUT* res = (const TOUT*)p.res;
    const int ch = tid % CPR;
    const int co0 = n0 + ch * CH;
    const bool cok = co0 < p.Cout;
    float st_s[CH], st_q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { st_s[c] = 0.f; st_q[c] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // residual pieces of this round are fetched before the transposition so their latency overlaps it
        uint4 resv[QN];
        if (res) {
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                const int rg = tid / CPR + q * RPP;
                const int m = (rg >> 5) * 64 + i * 32 + (rg & 31);
                const int gy = ty0 + (m >> 4), gx = tx0 + (m & 15);
                const bool ok = cok && gy < p.H && gx < p.W;
                const size_t pix = ok ? (size_t)(b * p.H + gy) * p.W + gx : 0;
                resv[q] = *reinterpret_cast<const uint4*>(res + pix * p.Cout + (ok ? co0 : 0));
            }
        }
        if (!helper) {
            float* const stg = reinterpret_cast<float*>(smem + wave * STG_WAVE);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    stg[row * STG_LD + j * 32 + (lane & 31)] = acc[i][j][r] + addv[j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int rg = tid / CPR + q * RPP;               // 0..127: MFMA wave rg>>5, row rg&31
            const int m = (rg >> 5) * 64 + i * 32 + (rg & 31);
            const int gy = ty0 + (m >> 4), gx = tx0 + (m & 15);
            const float* srow = reinterpret_cast<const float*>(smem + (rg >> 5) * STG_WAVE) + (rg & 31) * STG_LD + ch * CH;
            float v[CH];
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) {
                const float4 t4 = *reinterpret_cast<const float4*>(srow + c4 * 4);
                v[c4 * 4] = t4.x; v[c4 * 4 + 1] = t4.y; v[c4 * 4 + 2] = t4.z; v[c4 * 4 + 3] = t4.w;
            }
            if (cok && gy < p.H && gx < p.W) {
                const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
                if (res) {
                    float rv[CH];
                    Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&resv[q]), rv);
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
                    Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&packed), vr);
#pragma unroll
                    for (int c = 0; c < CH; ++c) { st_s[c] += vr[c]; st_q[c] += vr[c] * vr[c]; }
                }
            }
        }
        __syncthreads();
    }
#ifdef USE_HIP_TRACE_BUILD
    if (tracing) for (int i = 0; i < 2 * trace_n; ++i) p.trace[(wave >> 2) * 256 + i] = trace_lds[i];
#endif
    if (p.stats) {
        // lanes holding the same chunk column are CPR apart inside a wave
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
        __syncthreads();
        if (tid < BN) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < p.Cout) {
                float* dst = p.stats + (((size_t)b * gridDim.x + tile) * p.Cout + co) * 2;
                dst[0] = s; dst[1] = q;
            }
        }
    }
}

template <typename TIN, typename TOUT, int CK, bool ACT>
static void v3_launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr int ROWB = CK * (int)sizeof(TIN) + 16;
    constexpr int HPITCH = (V3_HE * ROWB / 16 + 15) / 16 * 16 * 16;
    constexpr int MAIN = 2 * V3_HE * HPITCH + 3 * V3_BN * ROWB;
    constexpr int EPI = 4 * 32 * (V3_BN + 4) * 4 + 8 * V3_BN * 2 * 4;
#ifdef USE_HIP_TRACE_BUILD
    constexpr int SMEM = (MAIN > EPI ? MAIN : EPI) + 4096;   // + the stamp area
#else
    constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
#endif
    static bool attr_set = false;
    auto kern = conv_v3_kernel<TIN, TOUT, CK, ACT>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_set = true;
    }
    dim3 grid(conv_v2_tiles(a.H, a.W), (a.Cout + V3_BN - 1) / V3_BN, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(512), SMEM, s, a);
}

void launch_conv_v3(const ConvArgs& a, hipStream_t s) {
    if (a.in_dtype == DT_BF16) { a.act ? v3_launch_t<__bf16, __bf16, 64, true>(a, s) : v3_launch_t<__bf16, __bf16, 64, false>(a, s); }
    else                       { a.act ? v3_launch_t<float, float, 32, true>(a, s) : v3_launch_t<float, float, 32, false>(a, s); }
}

}  // namespace use
