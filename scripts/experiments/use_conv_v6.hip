// conv_v6_kernel: conv_v4's geometry (one 8-wave workgroup per CU, 16x32-pixel x 128-channel tile, every wave two tile rows
// x 128 channels = 2x4 MFMA 32x32x16 tiles, K in 32-channel chunks) with a different main loop, for 16-bit storage types.
//
// What the measurements of round 2 said about conv_v4 / conv_v5: a loop that is nothing but
// "barrier + 16 MFMAs" per step already loses half of the matrix pipe - the rendezvous of eight (or four) waves every 512
// MFMA cycles is the dominant cost, not the VALU work (an epilogue with a third fewer VALU instructions changed nothing) and
// not the memory path.  conv_v6 therefore synchronises once per TAP ROW (three taps = 48 MFMAs per wave) instead of once per
// tap, and hides the fragment reads inside each wave instead of ping-ponging two wave groups:
//
//   * weights of a whole tap row (3 slabs = 24 KB) are brought in by LDS-DMA, double-buffered, one tap row ahead - no VGPRs,
//     no ds_write; the slab rows are unpadded and piece-swizzled in the blob (ConvArgs::wb) so that the lane-linear DMA image
//     is bank-conflict-free for the fragment reads;
//   * inside a tap row a wave runs six half-steps (tap, 16-channel k-step) of 8 MFMAs; the 6 fragment reads of half-step
//     h+1 are issued in front of the MFMAs of half-step h into the other of two fragment register sets, so their LDS
//     latency sits under 256 cycles of MFMA work of the same wave (and the partner wave of the SIMD fills what is left);
//   * the GroupNorm+SiLU staging of the next chunk's halo (5 pieces per thread) is spread over the tap rows: loads at the
//     top of tap row 0 (3 pieces) and in the middle of tap row 1 (2 pieces), transforms on the VALU inside half-steps,
//     ds_writes into the idle halo buffer - each tap row's writes come BEFORE that tap row issues its DMA / loads, because
//     hipcc drains the vector-memory counter in front of any ds_write while an LDS-DMA is in flight.
//
// Arithmetic, operand order and summation order are conv_v4's: results are bit-identical to it.
#include "use_kernels.h"
#include "use_device.h"

#include <cstdio>
#include <cstdlib>

namespace use {

constexpr int V6_TW = 32, V6_TH = 16;
constexpr int V6_HW = V6_TW + 2, V6_HH = V6_TH + 2;
constexpr int V6_BN = 128;

template <typename TIN, bool ACT>
__global__ __launch_bounds__(512) void conv_v6_kernel(ConvArgs p) {
    typedef Mfma<TIN> MF;
    typedef TIN TOUT;
    static_assert(sizeof(TIN) == 2, "conv_v6 is a 16-bit storage kernel");
    constexpr int CK = 32, VEC = 8, PARTS = 4;
    constexpr int ROWB = 80;                                 // halo pixel pitch (padded: conflict-free 16-lane ds_read_b128 groups)
    constexpr int WROWB = 64;                                // weight row pitch (unpadded DMA image, swizzled pieces)
    constexpr int BN = V6_BN, TM = 2, TN = 4;
    constexpr int HPITCH = V6_HW * ROWB;
    constexpr int HALO_BYTES = V6_HH * HPITCH;               // 48,960
    constexpr int SLAB = BN * WROWB;                         // 8,192: one (tap, chunk) slab
    constexpr int WROW_BYTES = 3 * SLAB;                     // 24,576: one tap row
    constexpr int W_OFF = 2 * HALO_BYTES;
    constexpr int NPIECE = V6_HH * V6_HW * PARTS;            // 2448 pieces per halo chunk
    constexpr int PIECE_ITERS = (NPIECE + 511) / 512;        // 5
    static_assert(PIECE_ITERS == 5, "v6 staging schedule");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [2][HALO_BYTES] halo tiles, [2][WROW_BYTES] weight tap rows; the epilogue re-uses all of it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z;
    int tile = blockIdx.x;                                   // XCD-aware order: contiguous band of tiles per XCD
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + V6_TW - 1) / V6_TW;
    const int ty0 = (tile / tiles_x) * V6_TH, tx0 = (tile % tiles_x) * V6_TW;
    const int n0 = blockIdx.y * BN;
    const int Ctot = p.C0 + p.C1, nchunks = Ctot / CK;
    const int XCtot = p.XC0 + p.XC1, nchunks2 = XCtot / CK;
    const int part = tid & (PARTS - 1);

    float addv[TN];                                          // bias + time-embedding bias of this lane's channels
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
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = addv[j];   // the sum starts at the bias

    int a_base[TM], b_off[TN][2];                            // LDS byte offsets of this lane's fragments
#pragma unroll
    for (int i = 0; i < TM; ++i) a_base[i] = (wave * 2 + i) * HPITCH + (lane & 31) * ROWB + (lane >> 5) * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = j * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) b_off[j][kk] = W_OFF + n * WROWB + (((kk * 2 + (lane >> 5)) ^ ((n >> 2) & 3)) << 4);
    }

    // ---- halo pieces: this thread's piece j (0..4) of every chunk -------------------------------------------------------
    int ppix[PIECE_ITERS], pdst[PIECE_ITERS]; int pmbits = 0;   // bit j of pmbits: piece j exists and lies inside the image
    bool has_piece4;
#pragma unroll
    for (int j = 0; j < PIECE_ITERS; ++j) {
        const int idx = j * 512 + tid;
        const int pix = idx / PARTS;
        const int hy = pix / V6_HW, hx = pix - hy * V6_HW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inb = idx < NPIECE && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        ppix[j] = inb ? (b * p.H + gy) * p.W + gx : 0;
        pmbits |= inb ? 1 << j : 0;
        pdst[j] = hy * HPITCH + hx * ROWB + part * 16;
    }
    has_piece4 = 4 * 512 + tid < NPIECE;                     // only piece 4 can be missing (2448 = 4 * 512 + 400)
    // GroupNorm affine (a, b) of every input channel of this batch item, copied to LDS once: the per-piece reads are LDS
    // reads (short-lived registers, lgkmcnt) instead of 16 registers that live across the whole chunk
    constexpr int COEF_OFF = W_OFF + 2 * WROW_BYTES;         // [<= 512 channels][2] floats behind the weight buffers
    float2* const coef_lds = reinterpret_cast<float2*>(smem + COEF_OFF);
    gn_fill_table(coef_lds, p, b, Ctot, tid, 512);
    float ca[VEC], cb[VEC];                                  // GroupNorm affine of the piece being transformed
    auto load_coef = [&](int chunk) {
        const float4* cf = reinterpret_cast<const float4*>(coef_lds + chunk * CK + part * VEC);
#pragma unroll
        for (int k2 = 0; k2 < VEC / 2; ++k2) {
            const float4 v = cf[k2];
            ca[2 * k2] = v.x; cb[2 * k2] = v.y; ca[2 * k2 + 1] = v.z; cb[2 * k2 + 1] = v.w;
        }
    };
    auto make_rsrc = [&](const void* base) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    };
    auto buf_ld = [&](const void* base, unsigned voff, unsigned soff) -> uint4 {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(base), voff, soff, 0));
    };
    auto src_ld0 = [&](int chunk, int pixoff) -> uint4 {
        const int c_glob = chunk * CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.C0) { src = (const TIN*)p.src0; Cs = p.C0; c_loc = c_glob; }
        else               { src = (const TIN*)p.src1; Cs = p.C1; c_loc = c_glob - p.C0; }
        return buf_ld(src, (unsigned)pixoff * (unsigned)(Cs * 2) + (unsigned)(part * 16), (unsigned)(c_loc * 2));
    };
    auto load_piece1 = [&](int chunk2, int q, uint4& raw) -> unsigned {     // shortcut: raw centre pixels, 4 per thread
        const int pix = (q * 512 + tid) / PARTS;             // 0..511
        const int gy = ty0 + (pix >> 5), gx = tx0 + (pix & 31);
        const bool inb = gy < p.H && gx < p.W;
        const int c_glob = chunk2 * CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.XC0) { src = (const TIN*)p.x0; Cs = p.XC0; c_loc = c_glob; }
        else                { src = (const TIN*)p.x1; Cs = p.XC1; c_loc = c_glob - p.XC0; }
        const unsigned pixoff = inb ? (unsigned)((b * p.H + gy) * p.W + gx) : 0u;
        raw = buf_ld(src, pixoff * (unsigned)(Cs * 2) + (unsigned)(part * 16), (unsigned)(c_loc * 2));
        return inb ? 0xffffffffu : 0u;
    };
    auto piece1_dst = [&](int q, int hb) -> int {
        const int pix = (q * 512 + tid) / PARTS;
        return hb * HALO_BYTES + ((pix >> 5) + 1) * HPITCH + ((pix & 31) + 1) * ROWB + part * 16;
    };

    // ---- weights by LDS-DMA: slab (tap, chunk) = 8 KB contiguous in the blob copy; wave w copies bytes [1024 w, 1024 w + 1024) ----
    const unsigned slab_b = (unsigned)(p.cout_pad * CK) * 2u;            // bytes per (tap, chunk) slab over all output channels
    const unsigned n0_b = (unsigned)(n0 * CK) * 2u;
    const unsigned wvoff = (unsigned)tid * 16u;
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma_slab = [&](const void* wbase, unsigned soff, int lds_off) {
        // (named temporaries on purpose: with the descriptor / pointer expressions inline hipcc 7.2 drops the kernel's host stub)
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(wbase);
        char* dst = smem + lds_off + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wvoff, soff, 0, 0);
    };
    // tap row RR (0..2) of chunk CC -> weight buffer BUF; RR == 3 wraps into the next chunk / the shortcut segment
#define V6_DMA_TAPROW(CC, RR, BUF)                                                                                   \
    {                                                                                                                \
        const int cw_ = (RR) > 2 ? (CC) + 1 : (CC);                                                                  \
        const int rw_ = (RR) > 2 ? 0 : (RR);                                                                         \
        if (cw_ < nchunks) {                                                                                         \
            dma_slab(p.wb, (unsigned)((rw_ * 3 + 0) * nchunks + cw_) * slab_b + n0_b, W_OFF + (BUF)*WROW_BYTES);              \
            dma_slab(p.wb, (unsigned)((rw_ * 3 + 1) * nchunks + cw_) * slab_b + n0_b, W_OFF + (BUF)*WROW_BYTES + SLAB);       \
            dma_slab(p.wb, (unsigned)((rw_ * 3 + 2) * nchunks + cw_) * slab_b + n0_b, W_OFF + (BUF)*WROW_BYTES + 2 * SLAB);   \
        } else if (nchunks2 > 0) {                                                                                   \
            dma_slab(p.w2b, n0_b, W_OFF + (BUF)*WROW_BYTES);     /* slab of shortcut chunk 0 */                       \
        }                                                                                                            \
    }
#define V6_DRAIN_BARRIER()                                                                                           \
    {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                  \
        __syncthreads();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
#define V6_TOUCH(R) asm volatile("" : "+v"((R).x), "+v"((R).y), "+v"((R).z), "+v"((R).w));

    // ---- prologue: weights of tap row 0 (DMA), chunk 0 halo (synchronous) -------------------------------------------------
    V6_DMA_TAPROW(0, 0, 0)
    {
        uint4 raw[PIECE_ITERS];
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) raw[j] = src_ld0(0, ppix[j]);
        __syncthreads();                                     // coef_lds complete
        load_coef(0);
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) {
            const uint4 t = stage_transform<TIN, ACT>(raw[j], (unsigned)__builtin_amdgcn_sbfe(pmbits, j, 1), ca, cb);
            if (j < 4 || has_piece4) *reinterpret_cast<uint4*>(smem + pdst[j]) = t;
        }
    }

    typename MF::frag fa[2][TM], fb[2][TN];                  // two fragment sets: half-step h computes on set h & 1
    uint4 raw0 = make_uint4(0, 0, 0, 0), raw1 = raw0, raw2 = raw0;   // halo pieces between their load and their transform
    uint4 tq = raw0;                                         // the transformed piece on its way to LDS

    // fragment reads of half-step (tap T, k-step KK) into set SET
#define V6_READ(SET, T, KK, HALO, WBUF)                                                                              \
    {                                                                                                                \
        const char* ha_ = (HALO) + ((T) / 3) * HPITCH + ((T) % 3) * ROWB + (KK)*32;                                   \
        const char* wb_ = (WBUF) + ((T) % 3) * SLAB;                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[SET][i] = MF::ld(ha_ + a_base[i]);                         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[SET][j] = MF::ld(wb_ + b_off[j][KK]);                      \
    }
#define V6_MMA(SET)                                                                                                  \
    {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                               \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(fa[SET][i], fb[SET][j], acc[i][j]);   \
    }
    // 8 MFMAs with the GroupNorm+SiLU transform of piece K (held in RAW) on the VALU between them
#define V6_MMA_XF(SET, RAW, K)                                                                                       \
    {                                                                                                                \
        load_coef(next ? c + 1 : c);                                                                                 \
        V6_MMA(SET)                                                                                                  \
        tq = stage_transform<TIN, ACT>(RAW, (unsigned)__builtin_amdgcn_sbfe(pmbits, K, 1), ca, cb);                  \
        V6_TOUCH(tq)                                                                                                 \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           /* MFMA    */               \
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                           /* 6 VALU  */               \
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                           /* 2 TRANS */               \
        }                                                                                                            \
    }
#define V6_WRITE(K, HALO_NEXT) { if ((K) < 4 || has_piece4) *reinterpret_cast<uint4*>((HALO_NEXT) + pdst[K]) = tq; }
#define V6_HS_BAR() __builtin_amdgcn_sched_barrier(0);

    for (int c = 0; c < nchunks; ++c) {
        const bool next = c + 1 < nchunks;
        char* const halo = smem + (c & 1) * HALO_BYTES;          // this chunk reads it
        char* const halo_next = smem + ((c & 1) ^ 1) * HALO_BYTES;
        // tap row index in the stream: 3 c + r; its weight buffer: (3 c + r) & 1 = (c + r) & 1
        // ------------------------------------------------ tap row 0 --------------------------------------------------------
        {
            const int wb = c & 1;
            const char* wbuf = smem + wb * WROW_BYTES;
            V6_DRAIN_BARRIER()                               // this tap row's weights landed; halo of this chunk complete
            V6_DMA_TAPROW(c, 1, wb ^ 1)
            if (next) {
                raw0 = src_ld0(c + 1, ppix[0]); raw1 = src_ld0(c + 1, ppix[1]); raw2 = src_ld0(c + 1, ppix[2]);
            }
            V6_READ(0, 0, 0, halo, wbuf)
            V6_HS_BAR() V6_READ(1, 0, 1, halo, wbuf) V6_MMA(0)
            V6_HS_BAR() V6_READ(0, 1, 0, halo, wbuf) V6_MMA(1)
            V6_HS_BAR() V6_READ(1, 1, 1, halo, wbuf) V6_MMA(0)
            V6_HS_BAR() V6_READ(0, 2, 0, halo, wbuf) V6_MMA(1)
            V6_HS_BAR() V6_READ(1, 2, 1, halo, wbuf) V6_MMA(0)
            V6_HS_BAR() V6_MMA(1)
        }
        // ------------------------------------------------ tap row 1 --------------------------------------------------------
        // (one straight-line instruction sequence whether or not a next chunk exists: only the memory operations of the
        // staging are predicated, the transforms run on stale registers in the last chunk - as in conv_v4)
        {
            const int wb = (c + 1) & 1;
            const char* wbuf = smem + wb * WROW_BYTES;
            V6_DRAIN_BARRIER()
            V6_TOUCH(raw0) V6_TOUCH(raw1) V6_TOUCH(raw2)     // landed: no compiler wait on them later
            V6_READ(0, 3, 0, halo, wbuf)
            V6_HS_BAR() V6_READ(1, 3, 1, halo, wbuf) V6_MMA_XF(0, raw0, 0)
            V6_HS_BAR() if (next) V6_WRITE(0, halo_next) V6_READ(0, 4, 0, halo, wbuf) V6_MMA_XF(1, raw1, 1)
            V6_HS_BAR() if (next) V6_WRITE(1, halo_next) V6_READ(1, 4, 1, halo, wbuf) V6_MMA_XF(0, raw2, 2)
            V6_HS_BAR() if (next) V6_WRITE(2, halo_next)
            __builtin_amdgcn_sched_barrier(0);
            V6_DMA_TAPROW(c, 2, wb ^ 1)                      // after this tap row's ds_writes (see the header)
            if (next) { raw0 = src_ld0(c + 1, ppix[3]); raw1 = src_ld0(c + 1, ppix[4]); }
            V6_READ(0, 5, 0, halo, wbuf) V6_MMA(1)
            V6_HS_BAR() V6_READ(1, 5, 1, halo, wbuf) V6_MMA(0)
            V6_HS_BAR() V6_MMA(1)
        }
        // ------------------------------------------------ tap row 2 --------------------------------------------------------
        {
            const int wb = c & 1;
            const char* wbuf = smem + wb * WROW_BYTES;
            V6_DRAIN_BARRIER()
            V6_TOUCH(raw0) V6_TOUCH(raw1)
            V6_READ(0, 6, 0, halo, wbuf)
            V6_HS_BAR() V6_READ(1, 6, 1, halo, wbuf) V6_MMA_XF(0, raw0, 3)
            V6_HS_BAR() if (next) V6_WRITE(3, halo_next) V6_READ(0, 7, 0, halo, wbuf) V6_MMA_XF(1, raw1, 4)
            V6_HS_BAR() if (next) V6_WRITE(4, halo_next)
            __builtin_amdgcn_sched_barrier(0);
            V6_DMA_TAPROW(c, 3, wb ^ 1)
            V6_READ(1, 7, 1, halo, wbuf) V6_MMA(0)
            V6_HS_BAR() V6_READ(0, 8, 0, halo, wbuf) V6_MMA(1)
            V6_HS_BAR() V6_READ(1, 8, 1, halo, wbuf) V6_MMA(0)
            V6_HS_BAR() V6_MMA(1)
        }
    }

    // ---- the fused 1x1 shortcut: raw centre pixels through registers (double-buffered), weights by DMA ---------------------
    // slab of shortcut chunk c2 sits in weight buffer (3 nchunks + c2) & 1 = (nchunks + c2) & 1, first 8 KB
    if (nchunks2 > 0) {
        uint4 r0, r1, r2, r3; unsigned m0, m1, m2, m3;
#define V6_SC_LOAD(C2) { m0 = load_piece1(C2, 0, r0); m1 = load_piece1(C2, 1, r1); m2 = load_piece1(C2, 2, r2); m3 = load_piece1(C2, 3, r3); }
        V6_SC_LOAD(0)
        V6_DRAIN_BARRIER()                                   // all waves have left the 3x3 segment: both halo buffers are free
        for (int c2 = 0; c2 < nchunks2; ++c2) {
            const int hb = c2 & 1, wb = (nchunks + c2) & 1;
            r0.x &= m0; r0.y &= m0; r0.z &= m0; r0.w &= m0; r1.x &= m1; r1.y &= m1; r1.z &= m1; r1.w &= m1;
            r2.x &= m2; r2.y &= m2; r2.z &= m2; r2.w &= m2; r3.x &= m3; r3.y &= m3; r3.z &= m3; r3.w &= m3;
            *reinterpret_cast<uint4*>(smem + piece1_dst(0, hb)) = r0; *reinterpret_cast<uint4*>(smem + piece1_dst(1, hb)) = r1;
            *reinterpret_cast<uint4*>(smem + piece1_dst(2, hb)) = r2; *reinterpret_cast<uint4*>(smem + piece1_dst(3, hb)) = r3;
            V6_DRAIN_BARRIER()                               // pieces visible, slab of c2 landed, previous iteration's reads done
            if (c2 + 1 < nchunks2) {
                dma_slab(p.w2b, (unsigned)(c2 + 1) * slab_b + n0_b, W_OFF + (wb ^ 1) * WROW_BYTES);
                V6_SC_LOAD(c2 + 1)
            }
            const char* halo = smem + hb * HALO_BYTES;
            const char* wbuf = smem + wb * WROW_BYTES;
            V6_READ(0, 4, 0, halo, wbuf - SLAB)              // centre tap; (T % 3) * SLAB = SLAB is taken back: the slab is the first of the buffer
            V6_READ(1, 4, 1, halo, wbuf - SLAB)
            V6_MMA(0)
            V6_MMA(1)
        }
#undef V6_SC_LOAD
    }
    V6_DRAIN_BARRIER()                                       // the epilogue re-uses the LDS
#undef V6_READ
#undef V6_MMA
#undef V6_MMA_XF
#undef V6_WRITE
#undef V6_HS_BAR
#undef V6_TOUCH
#undef V6_DRAIN_BARRIER
#undef V6_DMA_TAPROW

    // ------------------------------ epilogue (conv_v4's): per-wave LDS transpose, 16-byte buffer I/O ---------------------------
    constexpr int STG_LD = BN + 4;
    constexpr int STG_WAVE = 32 * STG_LD * 4;                // 16,896 B per wave and round
    constexpr int CH = 8, CPR = BN / CH, QN = 32 * CPR / 64, PPP = 64 / CPR;
    float* const stg = reinterpret_cast<float*>(smem + wave * STG_WAVE);
    float* const red = reinterpret_cast<float*>(smem + 8 * STG_WAVE);     // [8 waves][BN][2]
    const int ch = lane % CPR;
    const int co0 = n0 + ch * CH;
    const bool cok = co0 < p.Cout;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const size_t img_elems = (size_t)p.H * p.W * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc((TOUT*)p.out + (size_t)b * img_elems);
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(const_cast<TOUT*>((const TOUT*)p.res) + (size_t)b * img_elems);
    const unsigned voff = cok ? (unsigned)(((lane / CPR) * p.Cout + co0) * 2) : 0u;
    const unsigned pass_b = (unsigned)(PPP * p.Cout) * 2u;   // bytes between passes
    const bool has_res = p.res != nullptr, has_scale = p.out_scale != 1.f;
    float4 w4r[CH]; float b4r[CH];                           // Combine ('sum') weights of this lane's channels
    if (p.pyr && cok) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { w4r[c] = *reinterpret_cast<const float4*>(p.w4 + (size_t)(co0 + c) * 4); b4r[c] = p.b4[co0 + c]; }
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) { w4r[c] = make_float4(0.f, 0.f, 0.f, 0.f); b4r[c] = 0.f; }
    }
    float st_s[CH], st_q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { st_s[c] = 0.f; st_q[c] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int gy = ty0 + wave_u * 2 + i;
        const unsigned row_b = (unsigned)((gy * p.W + tx0) * p.Cout) * 2u;
        uint4 resv[QN];
        if (has_res) {
#pragma unroll
            for (int q = 0; q < QN; ++q)
                resv[q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff + row_b + (unsigned)q * pass_b, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                stg[row * STG_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int row = (q * 64 + lane) / CPR;
            float v[CH];
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) {
                const float4 t4 = *reinterpret_cast<const float4*>(stg + row * STG_LD + ch * CH + c4 * 4);
                v[c4 * 4] = t4.x; v[c4 * 4 + 1] = t4.y; v[c4 * 4 + 2] = t4.z; v[c4 * 4 + 3] = t4.w;
            }
            if (has_res) {
                float rv[CH];
                Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&resv[q]), rv);
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] += rv[c];
            }
            if (has_scale) {
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] *= p.out_scale;
            }
            if (p.pyr) {
                const size_t pix = (size_t)(b * p.H + gy) * p.W + tx0 + row;
                const float4 pq = *reinterpret_cast<const float4*>(p.pyr + pix * 4);
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] += b4r[c] + w4r[c].x * pq.x + w4r[c].y * pq.y + w4r[c].z * pq.z + w4r[c].w * pq.w;
            }
            if (cok) {
                const uint4 packed = Vec16<TOUT>::pack(v);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, packed), rs_out,
                                                       voff + row_b + (unsigned)q * pass_b, 0, 0);
#pragma unroll
                for (int c = 0; c < CH; ++c) { st_s[c] += v[c]; st_q[c] = fmaf(v[c], v[c], st_q[c]); }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (p.stats) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { st_s[c] = reduce_lanes_stride<16>(st_s[c]); st_q[c] = reduce_lanes_stride<16>(st_q[c]); }
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
            if (co < p.Cout) gn_accumulate(p.stats + ((size_t)b * p.Cout + co) * 2, s, q);
        }
    }
}

template <typename TIN, bool ACT>
static void v6_launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr int MAIN = 2 * V6_HH * V6_HW * 80 + 2 * 3 * V6_BN * 64 + 512 * 8;
    constexpr int EPI = 8 * 32 * (V6_BN + 4) * 4 + 8 * V6_BN * 2 * 4;
    constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
    static_assert(SMEM <= 160 * 1024, "LDS");
    static bool attr_set = false;
    auto kern = conv_v6_kernel<TIN, ACT>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_set = true;
    }
    dim3 grid(conv_v4_tiles(a.H, a.W), (a.Cout + V6_BN - 1) / V6_BN, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(512), SMEM, s, a);
}

static bool g_v6_on = false;                             // opt-in: use_set_option("conv_v6", 1)
void conv_v6_enable(bool on) { g_v6_on = on; }

// same shapes as conv_v4 (so the per-tile statistics layout and the summation order do not depend on which of the two runs)
bool conv_v6_eligible(const ConvArgs& a) {
    static const bool off = getenv("USE_HIP_NO_V6") != nullptr && atoi(getenv("USE_HIP_NO_V6")) != 0;   // A/B switch
    return g_v6_on && !off && a.in_dtype != DT_F32 && conv_v4_eligible(a);
}

void launch_conv_v6(const ConvArgs& a, hipStream_t s) {
    if (a.in_dtype == DT_BF16)     { a.act ? v6_launch_t<__bf16, true>(a, s) : v6_launch_t<__bf16, false>(a, s); }
    else if (a.in_dtype == DT_F16) { a.act ? v6_launch_t<_Float16, true>(a, s) : v6_launch_t<_Float16, false>(a, s); }
}

}  // namespace use
