// conv_v2_kernel: software-pipelined implicit-GEMM 3x3 convolution for the large feature maps (the kernel that carries
// >90 % of the path's FLOPs).  Same arithmetic and argument struct as conv_kernel (use_kernels.hip), different schedule:
//
//   * one workgroup = 8 waves (4 along pixels x 2 along output channels) computes a 16x16-pixel x 128-channel tile,
//     one workgroup per CU (LDS: 2 halo buffers + 2 weight slabs = 130 KB of 160 KB);
//   * the halo tile of the NEXT 64-channel chunk is loaded, GroupNorm+SiLU-transformed and written to the second LDS
//     buffer piece by piece behind the MFMAs of the current chunk's taps (no exposed restaging);
//   * weights are prefetched two (tap, chunk) iterations ahead in registers, double-buffered in LDS; one barrier per
//     iteration (16 MFMAs per wave);
//   * workgroups are re-ordered so that each XCD (private L2) works on a contiguous band of tiles.
//
// Segment 1 (the fused 1x1 shortcut of a res-block, raw input, centre tap only) streams 4 raw pieces per thread per
// iteration through the same double buffer.
#include "use_kernels.h"
#include "use_device.h"

#include <cstdio>
#include <cstdlib>

namespace use {

constexpr int V2_T = 16;                      // tile edge (pixels)
constexpr int V2_HE = V2_T + 2;               // halo edge
constexpr int V2_HALO = V2_HE * V2_HE;        // 324 halo pixels
constexpr int V2_BN = 128;

// GroupNorm affine + SiLU on one 16-byte piece; `mask` = 0 zeroes it (conv zero padding / outside the image)
template <typename TIN, bool ACT>
DEVI uint4 v2_transform(const uint4 raw, const unsigned mask, const float (&ca)[16 / sizeof(TIN)],
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

template <typename TIN, typename TOUT, int CK, bool ACT>
__global__ __launch_bounds__(512) void conv_v2_kernel(ConvArgs p) {
    typedef Mfma<TIN> MF;
    constexpr int VEC = 16 / sizeof(TIN);
    constexpr int PARTS = CK / VEC;
    constexpr int ROWB = CK * (int)sizeof(TIN) + 16;
    constexpr int BN = V2_BN, WM = 4, WN = 2, MW = 64, NW = 64, TM = 2, TN = 2;
    constexpr int KSTEPS = CK / MF::KM;
    // halo row pitch padded to a multiple of 256 B: the two pixel rows a 32-row MFMA fragment read touches then fall on
    // disjoint LDS bank slots (ds_read_b128 lane groups mix lanes of both rows) -- removes the A-operand bank conflicts
    constexpr int HPITCH = (V2_HE * ROWB / 16 + 15) / 16 * 16 * 16;
    constexpr int HALO_BYTES = V2_HE * HPITCH, W_BYTES = BN * ROWB;
    constexpr int MAIN_BYTES = 2 * HALO_BYTES + 2 * W_BYTES;
    constexpr int NPIECE = V2_HALO * PARTS;                 // 16-byte pieces per halo chunk (2592)
    constexpr int PIECE_ITERS = (NPIECE + 511) / 512;       // taps that carry one piece per thread (6)
    static_assert(PARTS == 8 && 512 % PARTS == 0 && PIECE_ITERS <= 8, "v2 staging assumes 128-byte chunk rows");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [2][HALO_BYTES] halo tiles, [2][W_BYTES] weight slabs, [512][16] dummy slots (threads without a piece)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int b = blockIdx.z;
    // XCD-aware order: dispatch is round-robin over the 8 XCDs; give each XCD a contiguous band of tiles
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tiles_x = (p.W + V2_T - 1) / V2_T;
    const int ty0 = (tile / tiles_x) * V2_T, tx0 = (tile % tiles_x) * V2_T;
    const int n0 = blockIdx.y * BN;
    const int Ctot = p.C0 + p.C1;
    const int nchunks = Ctot / CK;
    const int XCtot = p.XC0 + p.XC1;
    const int nchunks2 = XCtot / CK;
    const int part = tid & (PARTS - 1);

    // per-lane epilogue constants (bias + time-embedding bias of this lane's output channels): fetched first so their
    // latency is hidden behind the whole main loop instead of being exposed at the start of the epilogue
    float addv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = n0 + wn * 64 + j * 32 + (lane & 31);
        float add = 0.f;
        if (co < p.Cout) {
            if (p.bias) add += p.bias[co];
            if (p.temb) add += p.temb[(size_t)b * p.temb_bstride + co];
        }
        addv[j] = add;
    }

    // optional timeline: lane 0 of waves 0 and 4 of workgroup (0,0,0) stamp s_memtime at phase boundaries
    const bool tracing = p.trace != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0 && (wave & 3) == 0;
    int trace_n = 0;
#ifdef USE_HIP_TRACE_BUILD   /* make CXXFLAGS+=-DUSE_HIP_TRACE_BUILD: the stamps' stores perturb the waitcnt placement */
#define V2_STAMP(ID)                                                                                   \
    if (tracing && trace_n < 120) {                                                                    \
        p.trace[(wave >> 2) * 256 + 2 * trace_n] = (unsigned long long)(ID);                           \
        p.trace[(wave >> 2) * 256 + 2 * trace_n + 1] = __builtin_readcyclecounter(); ++trace_n;        \
    }
#else
#define V2_STAMP(ID)
    (void)tracing; (void)trace_n;
#endif
#ifdef USE_HIP_TRACE_FINE
#define V2_STAMPF(ID) V2_STAMP(ID)
#else
#define V2_STAMPF(ID)
#endif
    V2_STAMP(1)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = addv[j];   // bias + time-embedding bias: the sum starts there

    int a_base[TM], b_base[TN];                              // LDS byte offsets of this lane's fragments
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wm * MW + i * 32 + (lane & 31);
        a_base[i] = (m >> 4) * HPITCH + (m & 15) * ROWB + (lane >> 5) * MF::KPL * (int)sizeof(TIN);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
        b_base[j] = 2 * HALO_BYTES + (wn * NW + j * 32 + (lane & 31)) * ROWB + (lane >> 5) * MF::KPL * (int)sizeof(TIN);

    // ---- segment-0 halo pieces: this thread's piece j (0..5) of every chunk ------------------------------------------
    //   ppix: pixel index in the image batch (0 with pmask 0 when outside the image / no piece)
    //   pdst: LDS byte offset inside a halo buffer, or -1 -> the thread's dummy slot
    int ppix[PIECE_ITERS], pdst[PIECE_ITERS]; unsigned pmask[PIECE_ITERS];
#pragma unroll
    for (int j = 0; j < PIECE_ITERS; ++j) {
        const int idx = j * 512 + tid;
        const int pix = idx / PARTS;
        const int hy = pix / V2_HE, hx = pix - hy * V2_HE;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        const bool inb = idx < NPIECE && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        ppix[j] = inb ? gy * p.W + gx : 0;                  // pixel offset inside the item's image (the item offset sits in the buffer base)
        pmask[j] = inb ? 0xffffffffu : 0u;
        pdst[j] = idx < NPIECE ? hy * HPITCH + hx * ROWB + part * 16 : -1;
    }
    const int dummy_off = MAIN_BYTES + tid * 16;
    // GroupNorm affine (a, b) of every input channel of this item in LDS: finalised here from the producers' totals (or copied
    // from a coefficient array, or the identity) - no separate finalize launch, and the per-chunk reads are LDS reads
    constexpr int COEF_OFF = MAIN_BYTES + 512 * 16;
    float2* const coef_lds = reinterpret_cast<float2*>(smem + COEF_OFF);   // filled in the prologue, behind the first loads
    float ca[VEC], cb[VEC];                                  // GroupNorm affine of the chunk being staged
    auto load_coef = [&](int chunk) {
        const float2* cf = coef_lds + chunk * CK + part * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { const float2 v = cf[k]; ca[k] = v.x; cb[k] = v.y; }
    };
    // Global loads of the main loop are buffer loads: a 4-SGPR descriptor of the tensor, a uniform SGPR offset and a 32-bit
    // per-lane VGPR offset - half the address traffic of a 64-bit-pointer global_load per issue.
    auto buf_ld = [&](const void* base, unsigned voff, unsigned soff) -> uint4 {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    auto src_ld0 = [&](int chunk, int pixoff) -> uint4 {
        const int c_glob = chunk * CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.C0) { src = (const TIN*)p.src0; Cs = p.C0; c_loc = c_glob; }
        else               { src = (const TIN*)p.src1; Cs = p.C1; c_loc = c_glob - p.C0; }
        const unsigned voff = (unsigned)pixoff * (unsigned)(Cs * (int)sizeof(TIN)) + (unsigned)(part * 16);
        // per-item buffer base: the 32-bit offsets only have to span one image (any batch size, < 2 GB per image tensor)
        return buf_ld(src + (size_t)b * p.H * p.W * Cs, voff, (unsigned)(c_loc * (int)sizeof(TIN)));
    };
    // ---- segment-1 (shortcut) pieces: centre 16x16 pixels only, 4 per thread, raw ---------------------------------
    auto load_piece1 = [&](int chunk2, int q, uint4& raw) -> unsigned {
        const int pix = (q * 512 + tid) / PARTS;             // 0..255
        const int gy = ty0 + (pix >> 4), gx = tx0 + (pix & 15);
        const bool inb = gy < p.H && gx < p.W;
        const int c_glob = chunk2 * CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.XC0) { src = (const TIN*)p.x0; Cs = p.XC0; c_loc = c_glob; }
        else                { src = (const TIN*)p.x1; Cs = p.XC1; c_loc = c_glob - p.XC0; }
        const size_t pixoff = inb ? (size_t)(b * p.H + gy) * p.W + gx : 0;
        raw = *reinterpret_cast<const uint4*>(src + pixoff * Cs + c_loc + part * VEC);
        return inb ? 0xffffffffu : 0u;
    };
    auto piece1_dst = [&](int q, int hb) -> int {
        const int pix = (q * 512 + tid) / PARTS;
        return hb * HALO_BYTES + ((pix >> 4) + 1) * HPITCH + ((pix & 15) + 1) * ROWB + part * 16;
    };

    // ---- weights: 2 pieces per thread per (tap, chunk) slab; 32-bit element offsets from a uniform slab pointer -------
    const int wrow0 = tid / PARTS, wrow1 = (tid + 512) / PARTS;
    const unsigned wo0 = wrow0 * 9 * Ctot + part * VEC, wo1 = wrow1 * 9 * Ctot + part * VEC;  // rows are 9*Ctot apart
    const unsigned wob0 = wo0 * (unsigned)sizeof(TIN), wob1 = wo1 * (unsigned)sizeof(TIN);    // the same in bytes
    const int wdst0 = 2 * HALO_BYTES + wrow0 * ROWB + part * 16, wdst1 = 2 * HALO_BYTES + wrow1 * ROWB + part * 16;
    const TIN* const wseg0 = (const TIN*)p.w + (size_t)n0 * 9 * Ctot;
    // weights of iteration (chunk CC, tap TT) -> R0/R1 ; TT may run past 8 (wraps into the next chunk)
#define V2_LOAD_W(CC, TT, R0, R1)                                                                                    \
    {                                                                                                                \
        const int cw_ = (TT) > 8 ? (CC) + 1 : (CC);                                                                  \
        const int tw_ = (TT) > 8 ? (TT)-9 : (TT);                                                                    \
        if (cw_ < nchunks) {                                                                                         \
            const unsigned so_ = (unsigned)(tw_ * Ctot + cw_ * CK) * (unsigned)sizeof(TIN);                          \
            R0 = buf_ld(wseg0, wob0, so_); R1 = buf_ld(wseg0, wob1, so_);                                            \
        }                                                                                                            \
    }
#define V2_STORE_W(BUF, R0, R1)                                                                     \
    {                                                                                               \
        *reinterpret_cast<uint4*>(smem + (BUF)*W_BYTES + wdst0) = R0;                               \
        *reinterpret_cast<uint4*>(smem + (BUF)*W_BYTES + wdst1) = R1;                               \
    }

    // ---- prologue: chunk 0 halo (synchronous), weights of iteration 0 -------------------------------------------------
    // all global loads of the prologue are issued together (one exposed memory latency, not three)
    uint4 wa0 = make_uint4(0, 0, 0, 0), wa1 = wa0;
    {
        uint4 w00 = wa0, w01 = wa0, raw[PIECE_ITERS];
        V2_LOAD_W(0, 0, w00, w01);
        V2_LOAD_W(0, 1, wa0, wa1);                           // weights of iteration 1, stored by LDS(0)
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j) raw[j] = src_ld0(0, ppix[j]);
        gn_fill_table(coef_lds, p, b, Ctot, tid, 512);       // while the halo / weight loads are in flight
        __syncthreads();                                     // coef_lds complete
        load_coef(0);
        V2_STORE_W(0, w00, w01);
#pragma unroll
        for (int j = 0; j < PIECE_ITERS; ++j)
            *reinterpret_cast<uint4*>(smem + (pdst[j] >= 0 ? pdst[j] : dummy_off)) = v2_transform<TIN, ACT>(raw[j], pmask[j], ca, cb);
    }

    V2_STAMP(2)
    typename MF::frag af[KSTEPS][TM], bf[KSTEPS][TN];
    uint4 hL = wa0, hT = wa0, t0 = wa0;                      // piece in flight, piece being transformed, transformed piece
    // Per (chunk CC, tap T) -- T is a literal, so every table index / LDS offset folds.  Piece k (0..5) of chunk CC+1:
    //   global load issued in LDS(k) -> GroupNorm+SiLU on the VALU behind the MFMAs of MFMA(k+1) -> written in LDS(k+2).
    // LDS phase: read the 16 fragments of (CC,T) first, then the LDS writes (transformed piece, next weight slab) and
    // the global-load issue for later iterations, while the reads are in flight.
#define V2_LDS(CC, T)                                                                                                \
    {                                                                                                                \
        const int cc_ = (CC);                                                                                        \
        const int par_ = cc_ & 1;                            /* halo buffer this chunk reads; it parity = par_ ^ (T&1) */ \
        const bool next_ = cc_ + 1 < nchunks;                                                                        \
        V2_STAMPF(100 + (T))                                                                                          \
        {                                                                                                            \
            const char* ha_ = smem + par_ * HALO_BYTES + ((T) / 3) * HPITCH + ((T) % 3) * ROWB;                      \
            const char* wbuf_ = smem + (par_ ^ ((T)&1)) * W_BYTES;                                                   \
            _Pragma("unroll") for (int kk = 0; kk < KSTEPS; ++kk) {                                                  \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) af[kk][i] = MF::ld(ha_ + a_base[i] + kk * MF::KM * (int)sizeof(TIN)); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) bf[kk][j] = MF::ld(wbuf_ + b_base[j] + kk * MF::KM * (int)sizeof(TIN)); \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        V2_STAMPF(500 + (T))                                                                                          \
        if ((T) >= 2 && (T) < PIECE_ITERS + 2 && next_) {                                                            \
            constexpr int k_ = (T) >= 2 && (T) < PIECE_ITERS + 2 ? (T)-2 : 0;                                        \
            *reinterpret_cast<uint4*>(smem + (pdst[k_] >= 0 ? (par_ ^ 1) * HALO_BYTES + pdst[k_] : dummy_off)) = t0;  \
        }                                                                                                            \
        if ((T) < 8 || next_) V2_STORE_W((par_ ^ ((T)&1)) ^ 1, wa0, wa1);                                            \
        V2_STAMPF(600 + (T))                                                                                          \
        if ((T) >= 1 && (T) < PIECE_ITERS + 1) {                                                                     \
            /* the piece loaded one iteration ago has landed (older than the weights just waited for): park it in  */ \
            /* plain registers so the MFMA-phase transform carries no vmcnt wait on this phase's fresh loads       */ \
            hT = hL;                                                                                                 \
            asm volatile("" : "+v"(hT.x), "+v"(hT.y), "+v"(hT.z), "+v"(hT.w));                                       \
        }                                                                                                            \
        if ((T) < PIECE_ITERS && next_) {                                                                            \
            if ((T) == 0) load_coef(cc_ + 1);                                                                        \
            constexpr int k_ = (T) < PIECE_ITERS ? (T) : 0;                                                          \
            hL = src_ld0(cc_ + 1, ppix[k_]);                                                                           \
        }                                                                                                            \
        V2_LOAD_W(cc_, (T) + 2, wa0, wa1);                                                                           \
        V2_STAMPF(200 + (T))                                                                                          \
    }
    // MFMA phase: 16 MFMAs on the fragments read in the preceding LDS phase; the piece loaded one iteration ago is
    // normalised + activated on the VALU in their shadow.
#define V2_MFMA(CC, T)                                                                                               \
    {                                                                                                                \
        V2_STAMPF(300 + (T))                                                                                          \
        _Pragma("unroll") for (int kk = 0; kk < KSTEPS; ++kk)                                                        \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                           \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(af[kk][i], bf[kk][j], acc[i][j]);  \
        if ((T) >= 1 && (T) < PIECE_ITERS + 1) {             /* unconditional at run time: same basic block as the MFMAs */ \
            constexpr int k_ = (T) >= 1 && (T) < PIECE_ITERS + 1 ? (T)-1 : 0;                                        \
            t0 = v2_transform<TIN, ACT>(hT, pmask[k_], ca, cb);                                                      \
            asm volatile("" : "+v"(t0.x), "+v"(t0.y), "+v"(t0.z), "+v"(t0.w));   /* materialise here, not at the ds_write */ \
            _Pragma("unroll") for (int g = 0; g < TM * TN * KSTEPS; ++g) {                                           \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   /* 1 MFMA  */                                   \
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   /* 3 VALU  */                                   \
                __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);   /* 1 TRANS */                                   \
            }                                                                                                        \
        }                                                                                                            \
        V2_STAMPF(400 + (T))                                                                                          \
    }

    // Ping-pong schedule over the 3x3 segment: the two waves that share a SIMD (w and w+4) are always in opposite
    // phases, one s_barrier per phase.
    //   phase:   0        1        2        3        4       ...
    //   G0:    LDS(0)  MFMA(0)  LDS(1)  MFMA(1)  LDS(2)
    //   G1:     --     LDS(0)  MFMA(0)  LDS(1)  MFMA(1)
    // register-only MFMAs may legally move across s_barrier; pin the phases so the ping-pong survives scheduling
#define V2_BAR() { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
    V2_BAR();
    if (wave < 4) {
        V2_LDS(0, 0)
        V2_BAR();
        for (int c = 0; c < nchunks; ++c) {
#define V2_G0_STEP(T) V2_MFMA(c, T) V2_BAR(); V2_LDS(c, (T) + 1) V2_BAR();
            V2_G0_STEP(0) V2_G0_STEP(1) V2_G0_STEP(2) V2_G0_STEP(3) V2_G0_STEP(4) V2_G0_STEP(5) V2_G0_STEP(6) V2_G0_STEP(7)
#undef V2_G0_STEP
            V2_MFMA(c, 8)
            V2_BAR();
            if (c + 1 < nchunks) V2_LDS(c + 1, 0)
            V2_BAR();
        }
    } else {
        V2_BAR();
        for (int c = 0; c < nchunks; ++c) {
#define V2_G1_STEP(T) V2_LDS(c, T) V2_BAR(); V2_MFMA(c, T) V2_BAR();
            V2_G1_STEP(0) V2_G1_STEP(1) V2_G1_STEP(2) V2_G1_STEP(3) V2_G1_STEP(4) V2_G1_STEP(5) V2_G1_STEP(6) V2_G1_STEP(7) V2_G1_STEP(8)
#undef V2_G1_STEP
        }
    }
#undef V2_BAR
#undef V2_LDS
#undef V2_LOAD_W

    V2_STAMP(3)
    // ---- segment 1: the fused 1x1 shortcut (2-4 iterations): raw centre pixels; double-buffered, one barrier per
    // iteration, the loads of chunk c2+1 fly behind the fragment reads + MFMAs of chunk c2 -------------------------------
    if (nchunks2 > 0) {
        uint4 r0, r1, r2, r3; unsigned m0, m1, m2, m3;
#define V2_SC_LOAD(C2)                                                                                        \
        {                                                                                                     \
            m0 = load_piece1(C2, 0, r0); m1 = load_piece1(C2, 1, r1); m2 = load_piece1(C2, 2, r2); m3 = load_piece1(C2, 3, r3); \
            const TIN* wb_ = (const TIN*)p.w2 + (size_t)n0 * XCtot + (C2)*CK;                                 \
            wa0 = *reinterpret_cast<const uint4*>(wb_ + wrow0 * XCtot + part * VEC);                          \
            wa1 = *reinterpret_cast<const uint4*>(wb_ + wrow1 * XCtot + part * VEC);                          \
        }
        V2_SC_LOAD(0)
        for (int c2 = 0; c2 < nchunks2; ++c2) {
            const int buf = c2 & 1;
            r0.x &= m0; r0.y &= m0; r0.z &= m0; r0.w &= m0; r1.x &= m1; r1.y &= m1; r1.z &= m1; r1.w &= m1;
            r2.x &= m2; r2.y &= m2; r2.z &= m2; r2.w &= m2; r3.x &= m3; r3.y &= m3; r3.z &= m3; r3.w &= m3;
            *reinterpret_cast<uint4*>(smem + piece1_dst(0, buf)) = r0; *reinterpret_cast<uint4*>(smem + piece1_dst(1, buf)) = r1;
            *reinterpret_cast<uint4*>(smem + piece1_dst(2, buf)) = r2; *reinterpret_cast<uint4*>(smem + piece1_dst(3, buf)) = r3;
            V2_STORE_W(buf, wa0, wa1);
            __syncthreads();
            if (c2 + 1 < nchunks2) V2_SC_LOAD(c2 + 1)
            const char* ha_ = smem + buf * HALO_BYTES + HPITCH + ROWB;        // centre tap
            const char* wb2_ = smem + buf * W_BYTES;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = MF::ld(ha_ + a_base[i] + kk * MF::KM * (int)sizeof(TIN));
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = MF::ld(wb2_ + b_base[j] + kk * MF::KM * (int)sizeof(TIN));
            }
            V2_MFMA(nchunks, 0)
        }
#undef V2_SC_LOAD
        __syncthreads();                                     // the epilogue re-uses the LDS
    }
#undef V2_MFMA
#undef V2_STORE_W

    V2_STAMP(4)
    // ------------------------------ epilogue: per-wave LDS transpose, 16-byte I/O -------------------------------------------
    // Same VALU diet as conv_v4's epilogue: 32-bit buffer offsets (lane-constant part + uniform part), the bias already in
    // the accumulators, out_scale skipped when it is 1, GroupNorm partial sums from the fp32 values.
    constexpr int STG_LD = NW + 4;
    constexpr int STG_WAVE = 32 * STG_LD * 4;
    constexpr int CH = 16 / (int)sizeof(TOUT);
    constexpr int CPR = NW / CH;
    constexpr int QN = 32 * CPR / 64;
    constexpr int PPP = 64 / CPR;                            // pixels per pass (8 / 4): a pass never straddles a 16-pixel tile row
    float* const stg = reinterpret_cast<float*>(smem + wave * STG_WAVE);
    float* const red = reinterpret_cast<float*>(smem + 8 * STG_WAVE);     // [WM][BN][2]
    const int ch = lane % CPR, lx = lane / CPR;
    const int co0 = n0 + wn * NW + ch * CH;
    const bool cok = co0 < p.Cout;
    const int wm_u = __builtin_amdgcn_readfirstlane(wm);
    const size_t img_elems = (size_t)p.H * p.W * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((TOUT*)p.out + (size_t)b * img_elems, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<TOUT*>((const TOUT*)p.res) + (size_t)b * img_elems, 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)((lx * p.Cout + co0) * (int)sizeof(TOUT));
    const bool has_res = p.res != nullptr, has_scale = p.out_scale != 1.f;
    // Combine ('sum') weights of this lane's channels: loop-invariant, fetched once (they were re-read per pixel piece)
    constexpr bool HOIST_W4 = sizeof(TOUT) == 2;             // fp32 parity kernels: no registers to spare, in-loop loads
    float4 w4r[CH]; float b4r[CH];
    if (HOIST_W4 && p.pyr && cok) {
#pragma unroll
        for (int c = 0; c < CH; ++c) { w4r[c] = *reinterpret_cast<const float4*>(p.w4 + (size_t)(co0 + c) * 4); b4r[c] = p.b4[co0 + c]; }
    } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) { w4r[c] = make_float4(0.f, 0.f, 0.f, 0.f); b4r[c] = 0.f; }
    }
    float st_s[CH], st_q[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { st_s[c] = 0.f; st_q[c] = 0.f; }
    // pixel of pass q of round i: tile row trow (uniform), tile column tcol0 + lx
#define V2_EPI_POS(I, Q)                                                                                             \
        const int m0_ = wm_u * MW + (I)*32 + (Q)*PPP;                                                                \
        const int gy = ty0 + (m0_ >> 4), gx = tx0 + (m0_ & 15) + lx;                                                 \
        const bool ok = cok && gy < p.H && gx < p.W;                                                                 \
        const unsigned off = ok ? voff + (unsigned)((gy * p.W + tx0 + (m0_ & 15)) * p.Cout) * (unsigned)sizeof(TOUT) : 0u;
    // residual pieces of both staging rounds are fetched up front (their HBM latency overlaps the LDS transposes)
    uint4 resv[TM][QN];
    if (has_res) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                V2_EPI_POS(i, q)
                resv[i][q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0));
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                stg[row * STG_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            }
        __builtin_amdgcn_wave_barrier();
        V2_STAMP(41 + 2 * i)
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int row = (q * 64 + lane) / CPR;
            V2_EPI_POS(i, q)
            float v[CH];
#pragma unroll
            for (int c4 = 0; c4 < CH / 4; ++c4) {
                const float4 t4 = *reinterpret_cast<const float4*>(stg + row * STG_LD + ch * CH + c4 * 4);
                v[c4 * 4] = t4.x; v[c4 * 4 + 1] = t4.y; v[c4 * 4 + 2] = t4.z; v[c4 * 4 + 3] = t4.w;
            }
            if (has_res) {
                float rv[CH];
                Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&resv[i][q]), rv);
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] += rv[c];
            }
            if (has_scale) {
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] *= p.out_scale;
            }
            if (p.pyr && ok) {
                const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
                const float4 pq = *reinterpret_cast<const float4*>(p.pyr + pix * 4);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const float4 wq = HOIST_W4 ? w4r[c] : *reinterpret_cast<const float4*>(p.w4 + (size_t)(co0 + c) * 4);
                    v[c] += (HOIST_W4 ? b4r[c] : p.b4[co0 + c]) + wq.x * pq.x + wq.y * pq.y + wq.z * pq.z + wq.w * pq.w;
                }
            }
            if (ok) {
                const uint4 packed = Vec16<TOUT>::pack(v);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, packed), rs_out, off, 0, 0);
#pragma unroll
                for (int c = 0; c < CH; ++c) { st_s[c] += v[c]; st_q[c] = fmaf(v[c], v[c], st_q[c]); }
            }
        }
        __builtin_amdgcn_wave_barrier();
        V2_STAMP(42 + 2 * i)
    }
#undef V2_EPI_POS
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
    V2_STAMP(5)
}

template <typename TIN, typename TOUT, int CK, bool ACT>
static void v2_launch_t(const ConvArgs& a, hipStream_t s) {
    constexpr int ROWB = CK * (int)sizeof(TIN) + 16;
    constexpr int HPITCH = (V2_HE * ROWB / 16 + 15) / 16 * 16 * 16;
    constexpr int MAIN = 2 * V2_HE * HPITCH + 2 * V2_BN * ROWB + 512 * 16 + 1024 * 8;
    constexpr int EPI = 8 * 32 * (64 + 4) * 4 + 4 * V2_BN * 2 * 4;
    constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
    static LdsAttrOnce attr;                                 // per (instantiation, device)
    auto kern = conv_v2_kernel<TIN, TOUT, CK, ACT>;
    attr(kern, SMEM);
    dim3 grid(conv_v2_tiles(a.H, a.W), (a.Cout + V2_BN - 1) / V2_BN, a.B);
    hipLaunchKernelGGL(kern, grid, dim3(512), SMEM, s, a);
}

bool conv_v2_eligible(const ConvArgs& a) {
    const int Ctot = a.C0 + a.C1, XC = a.XC0 + a.XC1;
    const int ck = a.in_dtype == DT_F32 ? 32 : 64;
    return a.ntaps == 9 && a.Cout > 32 && a.in_dtype == a.out_dtype && Ctot % ck == 0 && Ctot <= 1024 && XC % ck == 0 &&
           (a.C1 == 0 || a.C0 % ck == 0) && (a.XC1 == 0 || a.XC0 % ck == 0) &&      // a chunk never straddles the two sources
           a.H >= V2_T && a.W >= V2_T;
}

void launch_conv_v2(const ConvArgs& a0, hipStream_t s) {
    ConvArgs a = a0;
    static const int dbg = getenv("USE_HIP_DBG") ? atoi(getenv("USE_HIP_DBG")) : 0;
    a.dbg = dbg;
    static unsigned long long* trace_buf = nullptr;
    if (getenv("USE_HIP_TRACE")) {           // bring-up only: print the phase timeline of the first eligible launch of a given shape
        static int printed = 0;
        const int want_c = atoi(getenv("USE_HIP_TRACE"));
        if (!trace_buf) { (void)hipMalloc((void**)&trace_buf, 512 * 8); }
        if (!printed && a.H == 512 && a.C0 + a.C1 == want_c) {
            (void)hipMemsetAsync(trace_buf, 0, 512 * 8, s);
            a.trace = trace_buf;
            if (a.in_dtype == DT_BF16) { a.act ? v2_launch_t<__bf16, __bf16, 64, true>(a, s) : v2_launch_t<__bf16, __bf16, 64, false>(a, s); }
            (void)hipStreamSynchronize(s);
            unsigned long long hbuf[512];
            (void)hipMemcpy(hbuf, trace_buf, sizeof hbuf, hipMemcpyDeviceToHost);
            for (int g = 0; g < 2; ++g) {
                unsigned long long prev = hbuf[g * 256 + 1];
                for (int i = 0; i < 120 && hbuf[g * 256 + 2 * i]; ++i) {
                    fprintf(stderr, "[trace G%d] id %3llu  +%6llu\n", g, hbuf[g * 256 + 2 * i], hbuf[g * 256 + 2 * i + 1] - prev);
                    prev = hbuf[g * 256 + 2 * i + 1];
                }
            }
            printed = 1;
            a.trace = nullptr;
        }
    }
    if (a.in_dtype == DT_BF16)     { a.act ? v2_launch_t<__bf16, __bf16, 64, true>(a, s) : v2_launch_t<__bf16, __bf16, 64, false>(a, s); }
    else if (a.in_dtype == DT_F16) { a.act ? v2_launch_t<_Float16, _Float16, 64, true>(a, s) : v2_launch_t<_Float16, _Float16, 64, false>(a, s); }
    else                           { a.act ? v2_launch_t<float, float, 32, true>(a, s) : v2_launch_t<float, float, 32, false>(a, s); }
}

}  // namespace use
