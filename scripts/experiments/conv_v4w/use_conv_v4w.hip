// conv_v4w_kernel: conv_v4's schedule for the PLAIN and RESIDUAL 3x3 convolutions of the 16-bit modes (no fused shortcut, no Combine),
// as a short WALK: one workgroup computes tile (ty, tx) of `ipw` consecutive batch items (normally 2), and the seam between two items
// costs almost nothing.
//
// Why (round 5; profiles/r5_energy_ablation_conv_v4_L0_128.txt, profiles/r4_conv_v4_cycle_stamps_ablations.txt): a 128 -> 128 tile of conv_v4
// takes 68 k cycles, of which the prologue is 15 k (5.5 k waiting for the first halo chunk at the ~10 B / cycle a CU gets, 2.4 k of address
// arithmetic and table building in front of it, 1.9 k transforming it, 3 k starting the ping-pong) and the workgroup's retirement (its
// output stores have to be acknowledged before the CU takes the next workgroup) another 2-3 k: with one workgroup per CU nothing overlaps
// them, and under the package power limit (PPT active the whole time, profiles/r5_limit_reasons_eval.json) that idle time is paid in
// joules.  Here:
//   * the piece tables, LDS slots, fragment addresses and zero padding are built ONCE per workgroup - the next item's tile has the same
//     geometry, only the buffer bases (SGPRs), the GroupNorm affine and the time-embedding row change;
//   * conv_v4's last K chunk already runs a full staging pass whose result nobody reads (its loads are unconditional by design: a load on
//     one control-flow path only costs hipcc's counted waits); in the walk that pass stages chunk 0 of the NEXT item's tile into the
//     other halo buffer, and the weight pipeline wraps to slab (tap 0, chunk 0) - the next tile's first MFMA phase finds everything in LDS;
//   * the epilogue therefore must leave halo buffer 0 and the weight slabs alone: it transposes through the dead halo buffer 1 in eight
//     units of 16 pixels x 64 channels per wave (4.3 KB each) instead of two rounds of 32 x 128 (16.9 KB), same instruction counts;
//   * every barrier is LDS-only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads()'s release fence would wait for the previous tile's
//     output stores at the first barrier of the next tile.
// LDS: 2 x 55,296 halo + 2 x 8,192 weights + 2 x 4,096 GroupNorm tables (this item / next item) + 20,480 piece tables + 8,192 statistics
// = 163,840 B, the whole CU.  Results: the convolution sums run in conv_v4's order (outputs bit-identical to conv_v4); the GroupNorm
// partial sums are taken over other pixel subsets per lane (totals equal to fp32 rounding).  The result does not depend on `ipw`.
#include "use_kernels.h"
#include "use_device.h"

#include <cstdlib>
#include <type_traits>

namespace use {

namespace {
constexpr int WK_TW = 32, WK_TH = 16, WK_HW = WK_TW + 2, WK_HH = WK_TH + 2, WK_BN = 128;
constexpr int WK_HROW = 48, WK_PXB = 64, WK_HPITCH = WK_HROW * WK_PXB;       // 3,072 B between halo rows
constexpr int WK_HALO = WK_HH * WK_HPITCH;                                     // 55,296
constexpr int WK_WB = WK_BN * WK_PXB;                                          // 8,192
constexpr int WK_COEF_OFF = 2 * WK_HALO + 2 * WK_WB;                           // 126,976: [2][512] float2
constexpr int WK_TAB_OFF = WK_COEF_OFF + 2 * 512 * 8;                          // 135,168: [2][5][512] int
constexpr int WK_RED_OFF = WK_TAB_OFF + 2 * 5 * 512 * 4;                       // 155,648: [8][128][2] float
constexpr int WK_SMEM = WK_RED_OFF + 8 * WK_BN * 2 * 4;                        // 163,840
constexpr int WK_STG_LD = 64 + 4, WK_STG_WAVE = 16 * WK_STG_LD * 4;            // 4,352 B per wave: 16 pixels x 64 channels fp32 (add-TID image: 16 rows x 256 B + 16)
static_assert(WK_SMEM <= 163840 && 8 * WK_STG_WAVE <= WK_HALO, "LDS budget");
}  // namespace

template <typename T, bool ACT>
__global__ __launch_bounds__(512) void conv_v4w_kernel(ConvArgs p, int ipw) {
    typedef Mfma<T> MF;
    constexpr int CK = 32, VEC = 8, PARTS = 4, PXB = WK_PXB, BN = WK_BN, TM = 2, TN = 4, KSTEPS = 2;
    constexpr int HROW = WK_HROW, HPITCH = WK_HPITCH, HALO_BYTES = WK_HALO, W_BYTES = WK_WB;
    constexpr int NPIECE = WK_HH * WK_HW * PARTS;            // 2448 pieces per halo chunk
    constexpr int PIECE_ITERS = (NPIECE + 511) / 512;        // 5
    static_assert(PIECE_ITERS == 5 && CK / MF::KM == KSTEPS, "layout");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.z * ipw;
    const int nt = min(ipw, p.B - b0);                       // items this workgroup walks
    int tile = blockIdx.x;                                   // XCD-aware order: contiguous band of tiles per XCD
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tiles_x = p.W / WK_TW;
    const int ty0 = (tile / tiles_x) * WK_TH, tx0 = (tile % tiles_x) * WK_TW;
    const int n0 = blockIdx.y * BN;
    const int Ctot = p.C0 + p.C1, nchunks = Ctot / CK;       // even (eligibility)
    __builtin_assume(nchunks >= 2);                          // (the chunk loops run: nothing staged before them stays live across them)
    const int part = tid & (PARTS - 1);
    const unsigned img_px = (unsigned)(p.H * p.W);

    // bias + time-embedding bias of this lane's four output channels for item b: the accumulators' initial value
    auto load_addv = [&](int b, float (&addv)[TN]) {
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
    };

#ifdef USE_HIP_TRACE_BUILD   /* bring-up: lane 0 of waves 0 and 4 of workgroup p.dbg stamp the cycle counter (stores straight to global memory: coarse phases only) */
    const bool tracing = p.trace != nullptr && (int)(blockIdx.x + gridDim.x * blockIdx.z) == p.dbg && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0;
    int trace_n = 0;
#define WK_STAMP(ID) if (tracing && trace_n < 120) { p.trace[(wave >> 2) * 256 + 2 * trace_n] = (unsigned long long)(ID); p.trace[(wave >> 2) * 256 + 2 * trace_n + 1] = __builtin_readcyclecounter(); ++trace_n; }
#else
#define WK_STAMP(ID)
#endif
    WK_STAMP(1)
    float addv0[TN];                                         // (first: these loads must be older than the halo loads, or the accumulator start waits for all of them)
    load_addv(b0, addv0);
    f32x16 acc[TM][TN];
    // LDS byte offsets of this lane's fragments (see conv_v4: unpadded 64-byte rows, 16-byte piece q of row P at slot q ^ ((P >> 2) & 3))
    const int h_ = lane >> 5, col_ = lane & 31;
    int a_dx[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int P = col_ + dx;
        a_dx[dx] = wave * 2 * HPITCH + P * PXB + ((h_ ^ ((P >> 2) & 3)) << 4);
    }
    const int b_0 = 2 * HALO_BYTES + col_ * PXB + ((h_ ^ ((col_ >> 2) & 3)) << 4);

    int* const pix_tab = reinterpret_cast<int*>(smem + WK_TAB_OFF);
    int* const dst_tab = pix_tab + PIECE_ITERS * 512;
    const int dummy_slot = ((tid / 56) * HROW + WK_HW) * PXB + (tid % 56) * 16;   // rows 0..9, pixels 34..47
    // GroupNorm affine (a, b) of every input channel: table (it & 1) holds the item being computed, the other one the next item
    auto coef_tab = [&](int slot) -> float2* { return reinterpret_cast<float2*>(smem + WK_COEF_OFF + (slot & 1) * 4096); };
    auto coef_of = [&](int b, int c) -> float2 {
        return p.gn_st0 ? gn_coef_of(p.gn_st0, p.C0, p.gn_st1, p.C1, p.gn_gamma, p.gn_beta, p.gn_groups, p.gn_inv_n, p.gn_eps, b, c)
               : p.coef ? *reinterpret_cast<const float2*>(p.coef + ((size_t)b * Ctot + c) * 2) : make_float2(1.f, 0.f);
    };
    float ca[VEC], cb[VEC];                                  // GroupNorm affine of the chunk being staged
    auto load_coef = [&](int slot, int chunk) {
        const float2* cf = coef_tab(slot) + chunk * CK + part * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { const float2 v = cf[k]; ca[k] = v.x; cb[k] = v.y; }
    };
    // live == 0 (uniform): an EMPTY descriptor - the load returns zeros and fetches nothing (the staging pass of the last item's last chunk)
    auto buf_ld = [&](const void* base, unsigned voff, unsigned soff, int live = 1) -> uint4 {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, live ? 0x7fffffff : 0, 0x00020000);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    auto src_ld0 = [&](int b, int chunk, int pixoff, int live = 1) -> uint4 {
        const int c_glob = chunk * CK;
        const T* src; int Cs, c_loc;
        if (c_glob < p.C0) { src = (const T*)p.src0; Cs = p.C0; c_loc = c_glob; }
        else               { src = (const T*)p.src1; Cs = p.C1; c_loc = c_glob - p.C0; }
        const unsigned voff = (unsigned)pixoff * (unsigned)(Cs * 2) + (unsigned)(part * 16);
        return buf_ld(src + (size_t)b * img_px * Cs, voff, (unsigned)(c_loc * 2), live);
    };

    const unsigned wvoff = (unsigned)tid * 16u;
    const int wdst = 2 * HALO_BYTES + tid * 16;
    const unsigned slab_b = (unsigned)(p.cout_pad * CK) * 2u;     // bytes per (tap, chunk) slab
    const unsigned n0_b = (unsigned)(n0 * CK) * 2u;
    // weights of iteration (chunk CC, tap TT) -> R ; TT may run past 8: the next chunk, and past the last chunk the NEXT TILE's chunk 0
#define WK_LOAD_W(CC, TT, R)                                                                                         \
    {                                                                                                                \
        const int cw0_ = (TT) > 8 ? (CC) + 1 : (CC);                                                                 \
        const int cw_ = cw0_ < nchunks ? cw0_ : 0;                                                                   \
        const int tw_ = (TT) > 8 ? (TT)-9 : (TT);                                                                    \
        R = buf_ld(p.wb, wvoff, (unsigned)(tw_ * nchunks + cw_) * slab_b + n0_b, cw0_ < nchunks || more_m);             \
    }
#define WK_STORE_W(BUF, R) { *reinterpret_cast<uint4*>(smem + (BUF)*W_BYTES + wdst) = R; }

    uint4 wS, hL[3];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    wS = hL[0] = hL[1] = hL[2] = zero4;
    {   // ---- prologue of the first item: as conv_v4 (affine + weights first, the five halo loads at once, the rest in their shadow) ----
        uint4 w0 = zero4, raw[PIECE_ITERS];
        float2 cfv0 = make_float2(1.f, 0.f), cfv1 = make_float2(1.f, 0.f);
        if (tid < Ctot) {
            cfv0 = coef_of(b0, tid);
            if (nt > 1) cfv1 = coef_of(b0 + 1, tid);
        }
        const bool more_m = true;                            // (WK_LOAD_W's liveness term: not at a walk's end here)
        WK_LOAD_W(0, 0, w0);
        WK_LOAD_W(0, 1, wS);                                 // stored by LDS(0)
        int slot[PIECE_ITERS], ppv[PIECE_ITERS]; bool inbv[PIECE_ITERS];
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) {
            const int idx = j * 512 + tid;
            const int pix = idx < NPIECE ? idx / PARTS : 0;
            const int hy = pix / WK_HW, hx = pix - hy * WK_HW;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool inb = idx < NPIECE && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const int pp = inb ? gy * p.W + gx : 0;
            raw[j] = src_ld0(b0, 0, pp);
            const int P = hy * HROW + hx;
            slot[j] = P * PXB + ((part ^ ((P >> 2) & 3)) << 4);
            ppv[j] = pp; inbv[j] = inb;
            if (idx >= NPIECE) slot[j] = -1;
        }
        __builtin_amdgcn_sched_barrier(0);                   // the loads above are issued before anything below
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) {
            const int idx = j * 512 + tid;
            if (slot[j] >= 0 && !inbv[j]) {                  // zero padding (applied AFTER the activation): zeroed once per buffer use
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
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = addv0[j];
        if (tid < Ctot) { coef_tab(0)[tid] = cfv0; coef_tab(1)[tid] = cfv1; }
        __syncthreads();                                     // tables complete (no stores in flight yet: a plain barrier)
        load_coef(0, 0);
        WK_STORE_W(0, w0);
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j)
            *reinterpret_cast<uint4*>(smem + slot[j]) = stage_transform<T, ACT>(raw[j], 0xffffffffu, ca, cb);
    }
    WK_STAMP(2)
    __syncthreads();
    WK_STAMP(3)
    const bool wg_oob = ty0 == 0 || tx0 == 0 || ty0 + WK_TH >= p.H || tx0 + WK_TW >= p.W;   // (uniform; no __syncthreads_or: its static LDS word would exceed the 160 KB)

    typename MF::frag af[KSTEPS][TM], bf[KSTEPS][TN];
    int dst_ = 0;
    // item walked now (b), item whose chunk 0 the last K chunk stages (bn; the last item: itself again, unused), GroupNorm table slots
    int b = b0, bn = nt > 1 ? b0 + 1 : b0, it = 0;
#define WK_XF_PHASE(T) ((T) >= 2 && (T) < PIECE_ITERS + 2)   /* MFMA(T) carries the transform of piece T - 2 */
#define WK_KX(KK) (((KK) * 2) << 4)
#define WK_LDS(CC, T)                                                                                                \
    {                                                                                                                \
        const int cc_ = (CC);                                                                                        \
        const int par_ = cc_ & 1;                            /* halo buffer this chunk reads */                      \
        const bool wrap_ = cc_ + 1 >= nchunks;               /* staging the next item's chunk 0 */                   \
        const int cn_ = wrap_ ? 0 : cc_ + 1;                                                                         \
        const int bs_ = wrap_ ? bn_u : b_u;                                                                            \
        int pix_ = 0;                                                                                                \
        if ((T) < PIECE_ITERS) pix_ = pix_tab[((T) < PIECE_ITERS ? (T) : 0) * 512 + tid];                            \
        if (WK_XF_PHASE(T)) dst_ = dst_tab[(WK_XF_PHASE(T) ? (T)-2 : 0) * 512 + tid];                                \
        {                                                                                                            \
            const char* ha_ = smem + par_ * HALO_BYTES + ((T) / 3) * HPITCH;                                         \
            const char* wbuf_ = smem + (par_ ^ ((T)&1)) * W_BYTES;                                                   \
            _Pragma("unroll") for (int kk = 0; kk < KSTEPS; ++kk) {                                                  \
                const int ak_ = a_dx[(T) % 3] ^ WK_KX(kk), bk_ = b_0 ^ WK_KX(kk);                                    \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) af[kk][i] = MF::ld(ha_ + ak_ + i * HPITCH);           \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[kk][j] = MF::ld(wbuf_ + bk_ + j * 32 * PXB);       \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (WK_XF_PHASE(T)) dst_ += (par_ ^ 1) * HALO_BYTES; /* where MFMA(T) puts its transformed piece */          \
        WK_STORE_W((par_ ^ ((T)&1)) ^ 1, wS);                                                                        \
        WK_LOAD_W(cc_, (T) + 2, wS);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                   /* the slab load stays older than the halo load */      \
        if ((T) < PIECE_ITERS) {                                                                                     \
            if ((T) == 0) load_coef(wrap_ ? it_u + 1 : it_u, cn_);                                                       \
            hL[(T) % 3] = src_ld0(bs_, cn_, pix_, !wrap_ || more_m);                                                                \
        }                                                                                                            \
    }
#define WK_MFMA(T)                                                                                                   \
    {                                                                                                                \
        if (WK_XF_PHASE(T)) {                                                                                        \
            const uint4 t0 = mfma16_with_transform<T_, ACT>(acc, af, bf, hL[(WK_XF_PHASE(T) ? (T)-2 : 0) % 3], ca, cb); \
            *reinterpret_cast<uint4*>(smem + dst_) = t0;     /* the other halo buffer: nobody reads it during this chunk */ \
        } else {                                                                                                     \
            _Pragma("unroll") for (int kk = 0; kk < KSTEPS; ++kk)                                                    \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                       \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(af[kk][i], bf[kk][j], acc[i][j]);  \
        }                                                                                                            \
    }
    typedef T T_;
    // Rendezvous exactly as conv_v4's (round 5: the first form, asm "s_waitcnt lgkmcnt(0); s_barrier" with a memory clobber, hid the wait from
    // hipcc's waitcnt bookkeeping, which then put ~50 redundant counted waits per chunk between the MFMAs - +7 % per chunk by the cycle
    // stamps).  With WK_LDS_FENCE the barrier is spelled with workgroup-scope fences restricted to the LDS address space instead.
#ifdef WK_LDS_FENCE
#define WK_BAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                   __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); __builtin_amdgcn_sched_barrier(0); }
#else
#define WK_BAR() { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
#endif
    // end of an MFMA phase: the only LDS operation in flight is the store of the piece just transformed, which nobody reads before the next chunk
#define WK_BAR_M() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (;;) {
        // (the walk's loop-carried scalars, made provably uniform: hipcc otherwise treats them as divergent, keeps every buffer descriptor that
        // depends on the item index in VGPRs and wraps each halo load of the main loop in a waterfall loop - measured 3 % of an evaluation)
        const int b_u = __builtin_amdgcn_readfirstlane(b), bn_u = __builtin_amdgcn_readfirstlane(bn), it_u = __builtin_amdgcn_readfirstlane(it);
        const bool more_m = it_u + 1 < nt;                   // another item follows: the last chunk's staging pass is live
        // ---------------- the 3x3 segment: conv_v4's ping-pong (waves 0-3 / 4-7 alternate LDS and MFMA phases) ----------------
        if (wave < 4) {
            WK_LDS(0, 0)
            WK_BAR();
            for (int c = 0; c < nchunks; ++c) {
                WK_STAMP(50 + c)
#define WK_G0_STEP(T) WK_MFMA(T) WK_BAR_M(); WK_LDS(c, (T) + 1) WK_BAR();
                WK_G0_STEP(0) WK_G0_STEP(1) WK_G0_STEP(2) WK_G0_STEP(3) WK_G0_STEP(4) WK_G0_STEP(5) WK_G0_STEP(6) WK_G0_STEP(7)
#undef WK_G0_STEP
                WK_MFMA(8)
                WK_BAR_M();
                if (c + 1 < nchunks) WK_LDS(c + 1, 0)
                WK_BAR();
            }
        } else {
            WK_BAR();
            for (int c = 0; c < nchunks; ++c) {
                WK_STAMP(50 + c)
#define WK_G1_STEP(T) WK_LDS(c, T) WK_BAR(); WK_MFMA(T) WK_BAR_M();
                WK_G1_STEP(0) WK_G1_STEP(1) WK_G1_STEP(2) WK_G1_STEP(3) WK_G1_STEP(4) WK_G1_STEP(5) WK_G1_STEP(6) WK_G1_STEP(7) WK_G1_STEP(8)
#undef WK_G1_STEP
            }
        }
        // Every fragment read of the last chunk (halo buffer 1: nchunks is even) has been consumed by an MFMA in front of the last
        // rendezvous: the epilogue may overwrite buffer 1 at once.  Buffer 0 (the next item's chunk 0), the weight slabs (tap 0 in
        // buffer 0, tap 1 in wS), both GroupNorm tables and the piece tables stay as they are.
        const bool more = it_u + 1 < nt;
        WK_STAMP(4)
        float addv_n[TN] = {0.f, 0.f, 0.f, 0.f};             // the next item's accumulator start values: in flight behind the epilogue
        if (more) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));                 // (opaque: keeps the addresses out of the main loop's register budget)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = n0 + j * 32 + (lane_o & 31);
                float add = 0.f;
                if (co < p.Cout) {
                    if (p.bias) add += p.bias[co];
                    if (p.temb) add += p.temb[(size_t)bn_u * p.temb_bstride + co];
                }
                addv_n[j] = add;
            }
        }

        // ---------------- epilogue: eight units of 16 pixels x 64 channels per wave through the dead halo buffer ----------------
        // (one instantiation per has_res: with a run-time flag the residual registers of the other path are merged in as live values - spills)
        auto epilogue = [&](auto has_res_c) {
            constexpr int CH = 8;                            // channels per 16-byte piece
            // (lane-derived constants of the epilogue start from an opaque copy of the lane id: otherwise LICM hoists two dozen address
            // registers out of the item loop and across the main loop, whose 249 registers have no room for them - 102 spills)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            float* const stg = reinterpret_cast<float*>(smem + HALO_BYTES + wave_u * WK_STG_WAVE);
            float* const red = reinterpret_cast<float*>(smem + WK_RED_OFF);
            const int ch = lane_o & 7, pxl = lane_o >> 3;    // piece of the 64-channel half, pixel of the pass
            // LDS byte address of the staging area (M0 of the add-TID stores) and this lane's read-back offset in floats: row (jj = ch >> 2,
            // rr = pxl & 3), half-row pxl >> 2, channels (ch & 3) * 8 ..
            const unsigned stg_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(HALO_BYTES + wave_u * WK_STG_WAVE);
            const int rd_off = ((ch >> 2) * 8 + (pxl & 3)) * 64 + (ch >> 2) * 4 + (pxl >> 2) * 32 + (ch & 3) * 8;
#ifdef WK_PLAIN_STAGING
            const int lrow = 4 * (lane_o >> 5), lcol = lane_o & 31;
#endif
            const size_t img_elems = (size_t)img_px * p.Cout;
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((T*)p.out + (size_t)b_u * img_elems, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((const T*)p.res) + (size_t)b_u * img_elems, 0, 0x7fffffff, 0x00020000);
            constexpr bool has_res = decltype(has_res_c)::value;
            const bool has_scale = p.out_scale != 1.f;
            const unsigned pass_b = (unsigned)(8 * p.Cout) * 2u;                         // bytes between the two passes of a unit
            // unit u: channel half jp = u >> 2, tile row i = (u >> 1) & 1, pixel half h = u & 1
            auto unit_off = [&](int u) -> unsigned {         // byte offset of (pixel pxl of the unit's first pass, piece ch)
                const int jp = u >> 2, i = (u >> 1) & 1, h = u & 1;
                return (unsigned)((((ty0 + wave_u * 2 + i) * p.W + tx0 + h * 16 + pxl) * p.Cout + n0 + jp * 64 + ch * CH) * 2);
            };
            // residual pieces: all eight units' loads are issued before the first transposition (the main loop's fragment, halo and
            // GroupNorm registers are dead here: 64 registers of cover for the ~2 k cycles an HBM load takes under load; one unit ahead
            // (WK_RES_AHEAD = 1) left the +res convolutions 13 % slower than the plain ones against conv_v4's 5 %)
#ifndef WK_RES_AHEAD
#define WK_RES_AHEAD 4
#endif
            constexpr int RA = WK_RES_AHEAD;
            uint4 resv[RA < 8 ? RA + 1 : 8][2];
            constexpr int RN = RA < 8 ? RA + 1 : 8;
            if (has_res) {
#pragma unroll
                for (int u = 0; u < (RA < 8 ? RA : 8); ++u)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        resv[u][q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, unit_off(u) + (unsigned)q * pass_b, 0, 0));
            }
            float st_s[CH], st_q[CH];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jp = u >> 2, i = (u >> 1) & 1, h = u & 1;
                const bool cok = n0 + jp * 64 + ch * CH < p.Cout;
                const unsigned off = unit_off(u);
                if ((u & 3) == 0) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) { st_s[c] = 0.f; st_q[c] = 0.f; }
                }
                if (RA < 8 && has_res && u + RA < 8) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        resv[(u + RA) % RN][q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, unit_off(u + RA) + (unsigned)q * pass_b, 0, 0));
                }
#ifdef WK_PLAIN_STAGING           /* round-5 first form: ds_write_b32 into a [pixel][64 + 4] image (64 B / clock / CU: the epilogue's bottleneck) */
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int row = (rr & 3) + 8 * (rr >> 2);                        // (+ lrow): pixel of this 16-pixel half
                        stg[(row + lrow) * WK_STG_LD + jj * 32 + lcol] = acc[i][jp * 2 + jj][h * 8 + rr];
                    }
                __builtin_amdgcn_wave_barrier();
#else
                // Staging by ds_write_addtid_b32 (address = M0 + offset + 4 lane: no address register, 128 B / clock / CU - twice ds_write_b32):
                // register (jj, rr) of the unit is one lane-linear row of 64 floats = [pixel half-row hi][channel l31]; row (jj, rr) sits at
                // (jj * 8 + rr) * 256 + jj * 16 bytes (the shift spreads the read-back's 16-lane groups over more banks).  LDS operations of a
                // wave execute in order: no wait between these stores and the reads below, none between the reads and the next unit's stores.
                {
                    unsigned keep;
#define WK_AT(JJ, RR) "ds_write_addtid_b32 %[a" #JJ #RR "] offset:" WK_STR((JJ * 8 + RR) * 256 + JJ * 16) "\n\t"
#define WK_STR2(X) #X
#define WK_STR(X) WK_STR2(X)
                    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[base]\n\ts_nop 0\n\t"
                                 "ds_write_addtid_b32 %[a00] offset:0\n\tds_write_addtid_b32 %[a01] offset:256\n\tds_write_addtid_b32 %[a02] offset:512\n\t"
                                 "ds_write_addtid_b32 %[a03] offset:768\n\tds_write_addtid_b32 %[a04] offset:1024\n\tds_write_addtid_b32 %[a05] offset:1280\n\t"
                                 "ds_write_addtid_b32 %[a06] offset:1536\n\tds_write_addtid_b32 %[a07] offset:1792\n\t"
                                 "ds_write_addtid_b32 %[a10] offset:2064\n\tds_write_addtid_b32 %[a11] offset:2320\n\tds_write_addtid_b32 %[a12] offset:2576\n\t"
                                 "ds_write_addtid_b32 %[a13] offset:2832\n\tds_write_addtid_b32 %[a14] offset:3088\n\tds_write_addtid_b32 %[a15] offset:3344\n\t"
                                 "ds_write_addtid_b32 %[a16] offset:3600\n\tds_write_addtid_b32 %[a17] offset:3856\n\t"
                                 "s_mov_b32 m0, %[keep]"
                                 : [keep] "=&s"(keep)
                                 : [base] "s"(stg_lds),
                                   [a00] "v"(acc[i][jp * 2][h * 8 + 0]), [a01] "v"(acc[i][jp * 2][h * 8 + 1]), [a02] "v"(acc[i][jp * 2][h * 8 + 2]), [a03] "v"(acc[i][jp * 2][h * 8 + 3]),
                                   [a04] "v"(acc[i][jp * 2][h * 8 + 4]), [a05] "v"(acc[i][jp * 2][h * 8 + 5]), [a06] "v"(acc[i][jp * 2][h * 8 + 6]), [a07] "v"(acc[i][jp * 2][h * 8 + 7]),
                                   [a10] "v"(acc[i][jp * 2 + 1][h * 8 + 0]), [a11] "v"(acc[i][jp * 2 + 1][h * 8 + 1]), [a12] "v"(acc[i][jp * 2 + 1][h * 8 + 2]), [a13] "v"(acc[i][jp * 2 + 1][h * 8 + 3]),
                                   [a14] "v"(acc[i][jp * 2 + 1][h * 8 + 4]), [a15] "v"(acc[i][jp * 2 + 1][h * 8 + 5]), [a16] "v"(acc[i][jp * 2 + 1][h * 8 + 6]), [a17] "v"(acc[i][jp * 2 + 1][h * 8 + 7])
                                 : "memory");
#undef WK_AT
#undef WK_STR
#undef WK_STR2
                }
#endif
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float v[CH];
#ifdef WK_PLAIN_STAGING
                    const int row = q * 8 + pxl;
                    const float4 t0 = *reinterpret_cast<const float4*>(stg + row * WK_STG_LD + ch * CH);
                    const float4 t1 = *reinterpret_cast<const float4*>(stg + row * WK_STG_LD + ch * CH + 4);
#else
                    // pixel p = q * 8 + pxl of the unit lives in register rr = (p & 3) + 4 (p >> 3), half-row hi = (p >> 2) & 1
                    const float* src = stg + rd_off + q * (4 * 64);
                    const float4 t0 = *reinterpret_cast<const float4*>(src);
                    const float4 t1 = *reinterpret_cast<const float4*>(src + 4);
#endif
                    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
                    if (has_res) {
                        float rv[CH];
                        Vec16<T>::load(reinterpret_cast<const T*>(&resv[u % RN][q]), rv);
#pragma unroll
                        for (int c = 0; c < CH; ++c) v[c] += rv[c];
                    }
                    if (has_scale) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) v[c] *= p.out_scale;
                    }
                    if (cok) {
                        const uint4 packed = Vec16<T>::pack(v);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, packed), rs_out,
                                                               off + (unsigned)q * pass_b, 0, 0);
#pragma unroll
                        for (int c = 0; c < CH; ++c) { st_s[c] += v[c]; st_q[c] = fmaf(v[c], v[c], st_q[c]); }
                    }
                }
#ifdef WK_PLAIN_STAGING
                __builtin_amdgcn_wave_barrier();
#endif
                if ((u & 3) == 3 && (p.stats || p.stats_part)) {               // end of a channel half: lanes holding the same piece are 8 apart inside a wave
#pragma unroll
                    for (int c = 0; c < CH; ++c) { st_s[c] = reduce_lanes_stride<8>(st_s[c]); st_q[c] = reduce_lanes_stride<8>(st_q[c]); }
                    if (lane_o < 8) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            red[(wave_u * BN + jp * 64 + ch * CH + c) * 2] = st_s[c]; red[(wave_u * BN + jp * 64 + ch * CH + c) * 2 + 1] = st_q[c];
                        }
                    }
                }
            }
            WK_STAMP(5)
            // every wave is through with its staging area (the seam re-zeroes padding inside it) and `red` is complete
            if (p.stats || p.stats_part || more) WK_BAR();
            WK_STAMP(6)
            if (p.stats || p.stats_part) {
                int tid_o = tid;
                asm volatile("" : "+v"(tid_o));
                if (tid_o < BN) {
                    float s = 0.f, q = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) { s += red[(w * BN + tid_o) * 2]; q += red[(w * BN + tid_o) * 2 + 1]; }
                    const int co = n0 + tid_o;
                    if (co < p.Cout) {
                        if (p.stats_part) {                  // this workgroup's partial totals of item b_u, plain 16-byte store
                            long long* d = p.stats_part + (((size_t)b_u * gridDim.x + blockIdx.x) * p.Cout + co) * 2;
                            *reinterpret_cast<longlong2*>(d) = make_longlong2(__float2ll_rn(s * GN_SUM_SCALE), __float2ll_rn(q * GN_SQ_SCALE));
                        } else gn_accumulate(p.stats + ((size_t)b_u * p.Cout + co) * 2, s, q);
                    }
                }
            }
                };
        if (p.res != nullptr) epilogue(std::true_type{}); else epilogue(std::false_type{});
        WK_STAMP(7)
        if (!more) break;

        // ---------------- seam: the next item's tile ----------------
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = addv_n[j];
        if (wg_oob) {                                        // border tiles: the zero padding of halo buffer 1 (the epilogue has used it)
            int tid_o = tid;
            unsigned z_o = 0;
            asm volatile("" : "+v"(tid_o), "+v"(z_o));       // (a zero made here: a uint4 of zeros kept from the prologue was spilled and reloaded per piece)
            const uint4 zero_o = make_uint4(z_o, z_o, z_o, z_o);
#pragma unroll
            for (int j = 0; j < PIECE_ITERS; ++j) {
                const int idx = j * 512 + tid_o;
                const int pix = idx < NPIECE ? idx / PARTS : 0;
                const int hy = pix / WK_HW, hx = pix - hy * WK_HW;
                const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                const int P = hy * HROW + hx;
                if (idx < NPIECE && !inb) *reinterpret_cast<uint4*>(smem + HALO_BYTES + P * PXB + (((idx & 3) ^ ((P >> 2) & 3)) << 4)) = zero_o;
            }
        }
        ++it; b = bn; bn = it + 1 < nt ? b + 1 : b;
        if (it + 1 < nt) {                                   // walks of more than two items: the table of the item after next (exposed latency)
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));
            if (tid_o < Ctot) coef_tab(it + 1)[tid_o] = coef_of(bn, tid_o);
        }
        WK_STAMP(8)
        WK_BAR();                                            // padding and tables in place, every wave out of its epilogue
        WK_STAMP(9)
    }
#undef WK_BAR
#undef WK_BAR_M
#undef WK_LDS
#undef WK_MFMA
#undef WK_LOAD_W
#undef WK_STORE_W
#undef WK_XF_PHASE
#undef WK_KX
}

static int g_v4w_on = 1;               // use_set_option("conv_v4w", 0): the plain / residual convolutions go back to conv_v4
static int g_v4w_ipw = 0;              // use_set_option("conv_v4w_ipw", n): items per workgroup (0: 2); pure scheduling, results do not depend on it
static long g_v4w_min_blocks = 512;    // use_set_option("conv_v4w_min_blocks", n): smallest per-IMAGE grid the walk is used for
void conv_v4w_set_enable(int on) { g_v4w_on = on; }
void conv_v4w_set_ipw(int n) { g_v4w_ipw = n; }
void conv_v4w_set_min_blocks(long n) { g_v4w_min_blocks = n; }

bool conv_v4w_eligible(const ConvArgs& a) {
    // conv_v4's conditions + 16-bit storage, an even number of K chunks (the walk's halo-buffer parity), no second K segment, no Combine.
    // The grid threshold is per image: the kernel choice (and with it the summation order of the GroupNorm partial sums) must not depend
    // on the batch size.  Sustained A/B (profiles/r5_conv_v4w_sustained_ab.txt): walking two items is 1.5-2 % faster than conv_v4 on the
    // 512 x 640 maps (640 workgroups per image -> 1 280 two-item walks for a sub-batch of 4 = 5 rounds); a one-item walk is 1-2 % slower
    // than conv_v4 (its eight-unit epilogue), and on the smaller maps two-item walks leave fewer workgroups than CUs x 2.
    const long blocks = (long)conv_v4_tiles(a.H, a.W) * ((a.Cout + WK_BN - 1) / WK_BN);
    return g_v4w_on && conv_v4_eligible(a) && a.XC0 + a.XC1 == 0 && a.pyr == nullptr && (a.in_dtype == DT_BF16 || a.in_dtype == DT_F16) &&
           (a.C0 + a.C1) % 64 == 0 && blocks >= g_v4w_min_blocks
#ifndef USE_HIP_TRACE_BUILD
           && a.trace == nullptr && a.dbg == 0
#endif
        ;
}

template <typename T, bool ACT>
static void v4w_launch_t(const ConvArgs& a, int ipw, hipStream_t s) {
    static LdsAttrOnce attr;
    auto kern = conv_v4w_kernel<T, ACT>;
    attr(kern, WK_SMEM);
    dim3 grid(conv_v4_tiles(a.H, a.W), (a.Cout + WK_BN - 1) / WK_BN, (a.B + ipw - 1) / ipw);
    hipLaunchKernelGGL(kern, grid, dim3(512), WK_SMEM, s, a, ipw);
}

int conv_v4w_items_per_wg(const ConvArgs& a) {
    const int want = g_v4w_ipw > 0 ? g_v4w_ipw : 2;
    return want < a.B ? want : a.B;
}

void launch_conv_v4w(const ConvArgs& a, hipStream_t s) {
    const int ipw = conv_v4w_items_per_wg(a);
    if (a.in_dtype == DT_BF16) { a.act ? v4w_launch_t<__bf16, true>(a, ipw, s) : v4w_launch_t<__bf16, false>(a, ipw, s); }
    else                       { a.act ? v4w_launch_t<_Float16, true>(a, ipw, s) : v4w_launch_t<_Float16, false>(a, ipw, s); }
}

}  // namespace use
