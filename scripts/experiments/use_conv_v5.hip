// conv_v5_kernel: the implicit-GEMM 3x3 convolution of the large feature maps as TWO independent 4-wave workgroups per
// CU (16-bit storage types).  Same arithmetic, operand order and argument struct as conv_v4_kernel - results are
// bit-identical to it - but a different occupancy model:
//
//   * one workgroup = 4 waves (one per SIMD) computes an 8-row x 32-column pixel tile x 128 output channels; every wave
//     owns two tile rows x all 128 channels = 2x4 MFMA 32x32x16 tiles (128 accumulator VGPRs), exactly as in conv_v4.
//     LDS <= 72 KB and <= 256 VGPRs, so two workgroups are resident per CU and every SIMD hosts one wave of each.
//   * the two co-resident workgroups are not synchronised with each other: while one sits in its prologue (first halo
//     chunk), at a barrier, in its fragment reads or in its epilogue (LDS transposition, residual, stores, GroupNorm
//     partial sums) the other one's MFMAs keep the matrix pipe busy.  conv_v4's single 8-wave workgroup per CU leaves
//     the pipe idle for the 25-35 % of a short-K tile that is prologue + epilogue; here that time belongs to the
//     neighbour.  Half of the first wave of workgroups is delayed by `stagger` so that the pairs start out of phase.
//   * weights: LDS-DMA (buffer_load ... lds, 16 B per lane) straight from the slab-major copy in the blob into a
//     double-buffered 8 KB slab - no VGPRs, no ds_write.  A DMA lands lane-linear, so the slab rows are unpadded
//     (64 B = 32 channels) and the bank-conflict-free image is pre-swizzled IN THE BLOB: 16-byte piece q of row n sits
//     at position q ^ ((n >> 2) & 3), the same involution the fragment reads apply.
//   * activations: halo tile (10 x 34 pixels x 32 channels, 80-byte padded rows) through registers, because GroupNorm +
//     SiLU is applied on the way (stage_transform), one 16-byte piece per thread per step in the shadow of the MFMAs.
//   * one __syncthreads per (chunk, tap) step; the loads issued at the top of step s (weights of step s+1, one halo
//     piece of the next chunk) have the whole step to land.
#include "use_kernels.h"
#include "use_device.h"

#include <cstdio>
#include <cstdlib>

namespace use {

constexpr int V5_TW = 32, V5_TH = 8;              // tile: 8 rows x 32 columns
constexpr int V5_HW = V5_TW + 2, V5_HH = V5_TH + 2;
constexpr int V5_BN = 128;

template <typename TIN, bool ACT>
__global__ __launch_bounds__(256, 2) void conv_v5_kernel(ConvArgs p) {
    typedef Mfma<TIN> MF;
    typedef TIN TOUT;
    static_assert(sizeof(TIN) == 2, "conv_v5 is the 16-bit storage kernel");
    constexpr int CK = 32, VEC = 8, PARTS = 4;
    constexpr int ROWB = 80;                                 // halo pixel pitch: conflict-free 16-lane ds_read_b128 groups
    constexpr int WROWB = 64;                                // weight row pitch (unpadded: DMA image), swizzled pieces
    constexpr int BN = V5_BN, TM = 2, TN = 4, KSTEPS = 2, KB = 32;
    constexpr int HPITCH = V5_HW * ROWB;
    constexpr int HALO_BYTES = V5_HH * HPITCH;               // 27,200
    constexpr int W_BYTES = BN * WROWB;                      // 8,192
    constexpr int MAIN_BYTES = 2 * HALO_BYTES + 2 * W_BYTES; // 70,784
    constexpr int COEF_OFF = MAIN_BYTES + 256 * 16;          // after the dummy slots: [<= 512 channels][2] floats
    constexpr int NPIECE = V5_HH * V5_HW * PARTS;            // 1360 pieces per halo chunk
    constexpr int PIECE_ITERS = (NPIECE + 255) / 256;        // 6
    static_assert(PIECE_ITERS == 6, "v5 staging layout");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [2][HALO_BYTES] halo tiles, [2][W_BYTES] weight slabs, [256][16] dummy slots (threads without a piece)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- workgroup -> (batch item, output-channel block, tile): contiguous band of the id space per XCD ----------------
    const int tiles_x = (p.W + V5_TW - 1) / V5_TW, tiles_y = (p.H + V5_TH - 1) / V5_TH;
    const int ntile = tiles_x * tiles_y, nblk = (p.Cout + BN - 1) / BN;
    int id;
    {
        const int L = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = L & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
        if (p.stagger > 0 && L >= p.stagger_lo && L < p.stagger_hi) {   // second resident workgroup of each CU: start out of phase
            for (int k = 0; k < p.stagger; ++k) __builtin_amdgcn_s_sleep(127);
        }
    }
    const int tile = id % ntile, rest = id / ntile;
    const int n0 = (rest % nblk) * BN, b = rest / nblk;
    const int ty0 = (tile / tiles_x) * V5_TH, tx0 = (tile % tiles_x) * V5_TW;
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

#ifdef USE_HIP_TRACE_BUILD   /* bring-up: lane 0 of wave 0 of workgroup p.dbg stamps the cycle counter at step boundaries */
    const bool tracing = p.trace != nullptr && (int)blockIdx.x == p.dbg && tid == 0;
    int trace_n = 0;
#define V5_STAMP(ID)                                                                                   \
    if (tracing && trace_n < 250) {                                                                    \
        p.trace[2 * trace_n] = (unsigned long long)(ID);                                               \
        p.trace[2 * trace_n + 1] = __builtin_readcyclecounter(); ++trace_n;                            \
    }
#else
#define V5_STAMP(ID)
#endif
    V5_STAMP(1)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_base[TM], b_base[TN][KSTEPS];                      // LDS byte offsets of this lane's fragments
#pragma unroll
    for (int i = 0; i < TM; ++i) a_base[i] = (wave * 2 + i) * HPITCH + (lane & 31) * ROWB + (lane >> 5) * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = j * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
            b_base[j][kk] = 2 * HALO_BYTES + n * WROWB + (((kk * 2 + (lane >> 5)) ^ ((n >> 2) & 3)) << 4);
    }

    // ---- halo pieces: this thread's piece j (0..5) of every chunk -------------------------------------------------------
    int ppix[PIECE_ITERS], pdst[PIECE_ITERS]; int pmbits = 0;   // bit j of pmbits: piece j lies inside the image
#pragma unroll
    for (int j = 0; j < PIECE_ITERS; ++j) {
        const int idx = j * 256 + tid;
        const int pix = idx / PARTS;
        const int hy = pix / V5_HW, hx = pix - hy * V5_HW;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inb = idx < NPIECE && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        ppix[j] = inb ? (b * p.H + gy) * p.W + gx : 0;
        pmbits |= inb ? 1 << j : 0;
        pdst[j] = idx < NPIECE ? hy * HPITCH + hx * ROWB + part * 16 : -1;
    }
    const int dummy_off = MAIN_BYTES + tid * 16;
    // GroupNorm affine (a, b) of every input channel of this batch item: copied to LDS once, so that the per-chunk reads are
    // LDS reads (lgkmcnt) and do not queue behind the weight DMA on the vector-memory counter
    float2* const coef_lds = reinterpret_cast<float2*>(smem + COEF_OFF);
    gn_fill_table(coef_lds, p, b, Ctot, tid, 256);
    float ca[VEC], cb[VEC];                                  // GroupNorm affine of the chunk being staged
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
        const unsigned voff = (unsigned)pixoff * (unsigned)(Cs * 2) + (unsigned)(part * 16);
        return buf_ld(src, voff, (unsigned)(c_loc * 2));
    };
    // shortcut (1x1, raw input) pieces: the 8x32 centre pixels, 4 per thread
    auto load_piece1 = [&](int chunk2, int q, uint4& raw) -> unsigned {
        const int pix = (q * 256 + tid) / PARTS;             // 0..255
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
        const int pix = (q * 256 + tid) / PARTS;
        return hb * HALO_BYTES + ((pix >> 5) + 1) * HPITCH + ((pix & 31) + 1) * ROWB + part * 16;
    };

    // ---- weights: LDS-DMA of one (tap, chunk) slab = 8 KB contiguous in the slab-major blob copy; every wave copies 2 KB ----
    const unsigned slab_b = (unsigned)(p.cout_pad * CK) * 2u;            // bytes per (tap, chunk) slab over all output channels
    const unsigned n0_b = (unsigned)(n0 * CK) * 2u;
    const unsigned wvoff = (unsigned)tid * 16u;                          // lane-linear inside the wave's 1 KB pieces
    typedef __attribute__((address_space(3))) void lds_void;
    auto dma_slab = [&](const void* wbase, unsigned soff, int buf) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(wbase);
        char* dst = smem + 2 * HALO_BYTES + buf * W_BYTES + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wvoff, soff, 0, 0);
        // (an instruction offset would be added to the LDS address as well as to the global one: keep it 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + 4096), 16, wvoff, soff + 4096u, 0, 0);
    };
    // weights of step (chunk CC, tap TT) -> buffer BUF; TT may be 9 (wraps into the next chunk / the shortcut segment)
#define V5_DMA_W(CC, TT, BUF)                                                                                        \
    {                                                                                                                \
        const int cw_ = (TT) > 8 ? (CC) + 1 : (CC);                                                                  \
        const int tw_ = (TT) > 8 ? 0 : (TT);                                                                         \
        if (cw_ < nchunks) dma_slab(p.wb, (unsigned)(tw_ * nchunks + cw_) * slab_b + n0_b, (BUF));                   \
        else if (nchunks2 > 0) dma_slab(p.w2b, n0_b, (BUF));                                                         \
    }

    // ---- prologue: weights of step 0 (DMA), chunk 0 halo (synchronous) ---------------------------------------------------
    V5_DMA_W(0, 0, 0)
    {
        uint4 raw[PIECE_ITERS];
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) raw[j] = src_ld0(0, ppix[j]);
        __syncthreads();                                     // coef_lds written by all threads
        load_coef(0);
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j)
            *reinterpret_cast<uint4*>(smem + (pdst[j] >= 0 ? pdst[j] : dummy_off)) = stage_transform<TIN, ACT>(raw[j], (unsigned)__builtin_amdgcn_sbfe(pmbits, j, 1), ca, cb);
    }

#ifdef USE_HIP_ABLATE        /* timing experiments only (results are wrong): p.dbg bits switch parts of the step off */
    const bool abl_xf = p.dbg & 1, abl_halo = p.dbg & 2, abl_dma = p.dbg & 4, abl_frag = p.dbg & 8;
#else
    constexpr bool abl_xf = false, abl_halo = false, abl_dma = false, abl_frag = false;
#endif
    typename MF::frag af[KSTEPS][TM], bf[KSTEPS][TN];
    uint4 hL = make_uint4(0, 0, 0, 0), hT = hL, t0 = hL;     // piece in flight, piece being transformed, transformed piece
    // One step = (chunk CC, tap T), T a literal.  Piece k (0..5) of chunk CC+1: global load issued in step k -> parked in
    // plain registers at the top of step k+1 -> GroupNorm+SiLU on the VALU behind the MFMAs of step k+1 -> written in
    // step k+2.  Weight slab of step s lives in buffer s & 1 = (CC ^ T) & 1 (nine steps per chunk).
#define V5_MFMA(T)                                                                                                   \
    {                                                                                                                \
        _Pragma("unroll") for (int kk = 0; kk < KSTEPS; ++kk)                                                        \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                           \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(af[kk][i], bf[kk][j], acc[i][j]);  \
        if ((T) >= 1 && (T) < PIECE_ITERS + 1) {             /* unconditional at run time: same basic block as the MFMAs */ \
            constexpr int k_ = (T) >= 1 && (T) < PIECE_ITERS + 1 ? (T)-1 : 0;                                        \
            t0 = abl_xf ? hT : stage_transform<TIN, ACT>(hT, (unsigned)__builtin_amdgcn_sbfe(pmbits, k_, 1), ca, cb);                                                 \
            asm volatile("" : "+v"(t0.x), "+v"(t0.y), "+v"(t0.z), "+v"(t0.w));   /* materialise here, not at the ds_write */ \
            _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                                         \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       /* MFMA  */                 \
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                       /* 3 VALU  */               \
                __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);                       /* 1 TRANS */               \
            }                                                                                                        \
        }                                                                                                            \
    }
// The compiler's own wait insertion is not relied on around the barriers: (a) it drops the vmcnt wait for the weight DMA on
// the loop back-edge, (b) it puts a vmcnt(0) in front of every ds_write while a DMA is in flight.  Hence the explicit drain
// in front of every barrier, and the staging ds_write sits between the barrier and the issue of this step's loads.
#define V5_SYNC()                                                                                                    \
    {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                  \
        __syncthreads();                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
#define V5_STEP(CC, T)                                                                                               \
    {                                                                                                                \
        const int cc_ = (CC);                                                                                        \
        const int par_ = cc_ & 1;                            /* halo buffer this chunk reads */                       \
        const int wb_ = par_ ^ ((T)&1);                      /* weight buffer of this step */                         \
        const bool next_ = cc_ + 1 < nchunks;                                                                        \
        V5_STAMP(100 + (T))                                                                                          \
        V5_SYNC()                                            /* slab of this step landed, previous step's reads done, */ \
        V5_STAMP(200 + (T))                                                                                          \
                                                             /* staged pieces visible, last step's piece load landed */ \
        if ((T) >= 2 && (T) < PIECE_ITERS + 2 && next_ && !abl_halo) {                                                          \
            constexpr int k_ = (T) >= 2 && (T) < PIECE_ITERS + 2 ? (T)-2 : 0;                                        \
            *reinterpret_cast<uint4*>(smem + (pdst[k_] >= 0 ? (par_ ^ 1) * HALO_BYTES + pdst[k_] : dummy_off)) = t0;  \
        }                                                                                                            \
        if ((T) >= 1 && (T) < PIECE_ITERS + 1) {                                                                     \
            hT = hL;                                                                                                 \
            asm volatile("" : "+v"(hT.x), "+v"(hT.y), "+v"(hT.z), "+v"(hT.w));                                       \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        if (!abl_dma) V5_DMA_W(cc_, (T) + 1, wb_ ^ 1)                                                                        \
        if ((T) < PIECE_ITERS && next_ && !abl_halo) {                                                                          \
            constexpr int k_ = (T) < PIECE_ITERS ? (T) : 0;                                                          \
            hL = src_ld0(cc_ + 1, ppix[k_]);                                                                         \
        }                                                                                                            \
        if ((T) == 0 && next_) load_coef(cc_ + 1);                                                                   \
        if (!abl_frag) {                                                                                             \
            const char* ha_ = smem + par_ * HALO_BYTES + ((T) / 3) * HPITCH + ((T) % 3) * ROWB;                      \
            const char* wbuf_ = smem + wb_ * W_BYTES;                                                                \
            _Pragma("unroll") for (int kk = 0; kk < KSTEPS; ++kk) {                                                  \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) af[kk][i] = MF::ld(ha_ + a_base[i] + kk * KB);        \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[kk][j] = MF::ld(wbuf_ + b_base[j][kk]);            \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        V5_STAMP(300 + (T))                                                                                          \
        V5_MFMA(T)                                                                                                   \
    }

    V5_STAMP(2)
    for (int c = 0; c < nchunks; ++c) {
        V5_STEP(c, 0) V5_STEP(c, 1) V5_STEP(c, 2) V5_STEP(c, 3) V5_STEP(c, 4) V5_STEP(c, 5) V5_STEP(c, 6) V5_STEP(c, 7) V5_STEP(c, 8)
    }
#undef V5_STEP
#undef V5_DMA_W

    // ---- the fused 1x1 shortcut: raw centre pixels; activations double-buffered through registers, weights by DMA ------------
    // weight slab of shortcut chunk c2 lives in buffer (9 * nchunks + c2) & 1 = (nchunks + c2) & 1 (the slab of c2 = 0 was
    // issued by the last 3x3 step)
    if (nchunks2 > 0) {
        uint4 r0, r1, r2, r3; unsigned m0, m1, m2, m3;
        const unsigned slab2_b = (unsigned)(p.cout_pad * CK) * 2u;
#define V5_SC_LOAD(C2) { m0 = load_piece1(C2, 0, r0); m1 = load_piece1(C2, 1, r1); m2 = load_piece1(C2, 2, r2); m3 = load_piece1(C2, 3, r3); }
        V5_SC_LOAD(0)
        for (int c2 = 0; c2 < nchunks2; ++c2) {
            const int hb = c2 & 1, wbi = (nchunks + c2) & 1;
            r0.x &= m0; r0.y &= m0; r0.z &= m0; r0.w &= m0; r1.x &= m1; r1.y &= m1; r1.z &= m1; r1.w &= m1;
            r2.x &= m2; r2.y &= m2; r2.z &= m2; r2.w &= m2; r3.x &= m3; r3.y &= m3; r3.z &= m3; r3.w &= m3;
            // halo buffer hb was last read two iterations ago (or by the 3x3 segment, whose last step ended with ... the
            // barrier below of the previous iteration); its writers wait for that barrier
            if (c2 == 0) V5_SYNC()                           // all waves have left the 3x3 segment (both halo buffers free)
            *reinterpret_cast<uint4*>(smem + piece1_dst(0, hb)) = r0; *reinterpret_cast<uint4*>(smem + piece1_dst(1, hb)) = r1;
            *reinterpret_cast<uint4*>(smem + piece1_dst(2, hb)) = r2; *reinterpret_cast<uint4*>(smem + piece1_dst(3, hb)) = r3;
            V5_SYNC()                                        // pieces visible, slab of c2 landed, previous iteration's reads done
            if (c2 + 1 < nchunks2) {
                dma_slab(p.w2b, (unsigned)(c2 + 1) * slab2_b + n0_b, wbi ^ 1);
                V5_SC_LOAD(c2 + 1)
            }
            const char* ha_ = smem + hb * HALO_BYTES + HPITCH + ROWB;         // centre tap
            const char* wbuf_ = smem + wbi * W_BYTES;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = MF::ld(ha_ + a_base[i] + kk * KB);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = MF::ld(wbuf_ + b_base[j][kk]);
            }
            V5_MFMA(0)
        }
#undef V5_SC_LOAD
    }
    V5_STAMP(4)
    V5_SYNC()                                                // the epilogue re-uses the LDS
    V5_STAMP(5)
#undef V5_MFMA
#undef V5_SYNC

    // ------------------------------ epilogue: per-wave LDS transpose, 16-byte I/O ------------------------------------------
    constexpr int STG_LD = BN + 4;
    constexpr int STG_WAVE = 32 * STG_LD * 4;                // 16,896 B per wave and round
    constexpr int CH = 8;                                    // channels per 16-byte piece
    constexpr int CPR = BN / CH;                             // 16-byte pieces per pixel row: 16
    constexpr int QN = 32 * CPR / 64;                        // passes per round: 8
    float* const stg = reinterpret_cast<float*>(smem + wave * STG_WAVE);
    float* const red = reinterpret_cast<float*>(smem + 4 * STG_WAVE);     // [4 waves][BN][2]
    TOUT* out = (TOUT*)p.out;
    const TOUT* res = (const TOUT*)p.res;
    const int ch = lane % CPR;
    const int co0 = n0 + ch * CH;
    const bool cok = co0 < p.Cout;
    // Combine ('sum') weights of this lane's channels: loop-invariant, fetched once (they were re-read per pixel piece)
    float4 w4r[CH]; float b4r[CH];
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
        const int gy = ty0 + wave * 2 + i;                   // this round's tile row
        uint4 resv[QN];                                      // residual pieces: fetched before the transposition
        if (res) {
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                const int gx = tx0 + (q * 64 + lane) / CPR;
                const bool ok = cok && gy < p.H && gx < p.W;
                const size_t pix = ok ? (size_t)(b * p.H + gy) * p.W + gx : 0;
                resv[q] = *reinterpret_cast<const uint4*>(res + pix * p.Cout + (ok ? co0 : 0));
            }
        }
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
            const int row = (q * 64 + lane) / CPR;           // pixel column inside the tile row
            const int gx = tx0 + row;
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
                    Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&resv[q]), rv);
#pragma unroll
                    for (int c = 0; c < CH; ++c) v[c] += rv[c];
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] *= p.out_scale;
                if (p.pyr) {
                    const float4 pq = *reinterpret_cast<const float4*>(p.pyr + pix * 4);
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        v[c] += b4r[c] + w4r[c].x * pq.x + w4r[c].y * pq.y + w4r[c].z * pq.z + w4r[c].w * pq.w;
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
        __builtin_amdgcn_wave_barrier();
        V5_STAMP(7)
    }
    if (p.stats) {
        // lanes holding the same 16-byte channel piece are CPR = 16 apart inside a wave
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
            for (int w = 0; w < 4; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
            const int co = n0 + tid;
            if (co < p.Cout) gn_accumulate(p.stats + ((size_t)b * p.Cout + co) * 2, s, q);
        }
    }
    V5_STAMP(8)
}

template <typename TIN, bool ACT>
static void v5_launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr int MAIN = 2 * V5_HH * V5_HW * 80 + 2 * V5_BN * 64 + 256 * 16 + 512 * 8;
    constexpr int EPI = 4 * 32 * (V5_BN + 4) * 4 + 4 * V5_BN * 2 * 4;
    constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
    static_assert(SMEM <= 80 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    auto kern = conv_v5_kernel<TIN, ACT>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        attr_set = true;
    }
    const long nwg = (long)conv_v5_tiles(a.H, a.W) * ((a.Cout + V5_BN - 1) / V5_BN) * a.B;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), SMEM, s, a);
}

static long g_v5_min_blocks = 1L << 40;          // off by default (opt-in through use_set_option): in-network it trails conv_v4
static int g_v5_stagger = -1;                                // -1: automatic (from the K extent), 0: off, > 0: s_sleep(127) count
void conv_v5_set_min_blocks(long n) { g_v5_min_blocks = n; }
void conv_v5_set_stagger(int n) { g_v5_stagger = n; }

bool conv_v5_eligible(const ConvArgs& a) {
    static const bool off = getenv("USE_HIP_NO_V5") != nullptr && atoi(getenv("USE_HIP_NO_V5")) != 0;   // A/B switch
    const int Ctot = a.C0 + a.C1, XC = a.XC0 + a.XC1;
    // per image, so that the kernel choice - and with it the summation order - does not depend on the batch size
    const long blocks = (long)conv_v5_tiles(a.H, a.W) * ((a.Cout + V5_BN - 1) / V5_BN);
    return !off && a.in_dtype != DT_F32 && a.wb != nullptr && (XC == 0 || a.w2b != nullptr) && a.ntaps == 9 && a.Cout > 32 &&
           a.in_dtype == a.out_dtype && Ctot % 32 == 0 && Ctot <= 512 && XC % 32 == 0 && a.cout_pad % V5_BN == 0 && a.Cout % 8 == 0 &&
           blocks >= g_v5_min_blocks;
}

void launch_conv_v5(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    // Stagger: the workgroups 256..511 of the dispatch order are (in practice) the second resident workgroup of every CU;
    // delaying them by about half a tile's main loop puts the two neighbours out of phase for the whole launch.
    const long nwg = (long)conv_v5_tiles(a.H, a.W) * ((a.Cout + V5_BN - 1) / V5_BN) * a.B;
    const int steps = (a.C0 + a.C1) / 32 * 9 + (a.XC0 + a.XC1) / 32;
    a.stagger = g_v5_stagger >= 0 ? g_v5_stagger : (nwg >= 512 ? std::max(1, steps / 24) : 0);   // ~ steps * 700 cycles / 2 / 8128
    a.stagger_lo = 256; a.stagger_hi = 512;
    if (a.in_dtype == DT_BF16)     { a.act ? v5_launch_t<__bf16, true>(a, s) : v5_launch_t<__bf16, false>(a, s); }
    else if (a.in_dtype == DT_F16) { a.act ? v5_launch_t<_Float16, true>(a, s) : v5_launch_t<_Float16, false>(a, s); }
}

}  // namespace use
