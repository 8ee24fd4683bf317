// conv_v7_kernel: the persistent form of conv_v4_kernel (same operator, same tile geometry, same K order - the stored results are
// bit-identical), built to take the parts of a tile's life that conv_v4 runs with the matrix pipe idle out of the critical path:
//
//   * ONE workgroup per CU walks a contiguous range of (tile, 128-channel block) units of ONE batch item.  The halo pipeline does
//     not stop at a unit's last K chunk: its "next chunk" is chunk 0 of the next unit, staged (load -> GroupNorm+SiLU -> LDS) behind
//     the MFMAs of the current unit's last chunk like any other chunk, and the weight slabs run two iterations ahead across the unit
//     boundary as well.  conv_v4's synchronous prologue (~12 k cycles per K = 1152 tile: the whole chip bursting its first halo
//     chunk at once) is paid once per workgroup instead of once per tile.
//   * The MFMA operands are swapped (A = weights, B = pixels), so an accumulator register holds 4 consecutive CHANNELS of one pixel:
//     after one v_permlane32_swap per register pair a lane owns 8 consecutive channels = one 16-byte piece of the NHWC output.  The
//     epilogue needs no LDS transposition (conv_v4: 128 ds_write_b32 + 32 ds_read_b128 per wave and 135 KB of staging that a
//     prefetched next tile would have to share), so the LDS keeps the next unit's first halo chunk and weight slab while it runs.
//   * The residual is added by the matrix pipe: the residual tile in NHWC is exactly an MFMA B fragment (lane = pixel, 8 consecutive
//     channels), so acc += OneHot x residual is one MFMA per 16-byte piece, exact (a single product with 1.0 per element), instead of
//     unpack + add on the VALU.  GroupNorm partial sums likewise: packed outputs x OneHot turns "lane = pixel" back into
//     "lane = channel" inside an accumulator that sums over the wave's 64 pixels; squares go through the same path.
//   * GroupNorm totals of all the units of a workgroup are accumulated in LDS (64-bit fixed point, integer atomics: order-independent)
//     and flushed with ONE pair of global atomics per channel per workgroup (conv_v4: one pair per channel per tile).
//
// Main loop: conv_v4's ping-pong schedule (use_conv_v4.hip) - two wave groups one phase apart, "LDS phase" (12 fragment reads +
// staging stores + global-load issue) against "MFMA phase" (16 MFMAs with the GroupNorm+SiLU transform of one halo piece in their
// shadow), one s_barrier per phase.  Both groups run the SAME instruction stream here (the second group is one barrier late).
#include "use_kernels.h"
#include "use_device.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <algorithm>

namespace use {

constexpr int V7_TW = 32, V7_TH = 16, V7_HW = V7_TW + 2, V7_HH = V7_TH + 2, V7_BN = 128, V7_CK = 32;
constexpr int V7_ROWB = V7_CK * 2 + 16;                    // 80-byte pixel rows: conflict-free 16-lane ds_read_b128 groups
constexpr int V7_HPITCH = V7_HW * V7_ROWB;                 // 2720
constexpr int V7_HALO = V7_HH * V7_HPITCH;                 // 48,960
constexpr int V7_WSLAB = V7_BN * V7_CK * 2;                // 8,192: one (tap, chunk) slab of a 128-channel block, unpadded (piece-swizzled)
constexpr int V7_OFF_W = 2 * V7_HALO;                      // 97,920
constexpr int V7_OFF_COEF = V7_OFF_W + 2 * V7_WSLAB;       // 114,304: GroupNorm affine of the item's input channels (<= 512 x float2)
constexpr int V7_OFF_BIAS = V7_OFF_COEF + 4096;            // 118,400: bias + time embedding of the unit's 128 channels
constexpr int V7_OFF_TOT = V7_OFF_BIAS + 512;              // 118,912: [Cout <= 256][2] int64 GroupNorm totals of this workgroup
constexpr int V7_OFF_S = V7_OFF_TOT + 4096;                // 123,008: shortcut operand, even chunks: 512 centre pixels x 64 B, piece-swizzled
constexpr int V7_SBYTES = V7_TH * V7_TW * V7_CK * 2;       // 32,768   (odd chunks go to the halo buffer the last 3x3 chunk has left)
constexpr int V7_SMEM = V7_OFF_S + V7_SBYTES;              // 155,776

struct V7Plan {
    int wpi;          // workgroups per batch item
    int units;        // units per item = tiles x channel blocks
    int nb;           // channel blocks (Cout / 128)
    int tiles_x;      // tiles per image row
};

template <typename T> struct OneHot;
template <> struct OneHot<__bf16> { static constexpr unsigned ONE = 0x3F80u; };
template <> struct OneHot<_Float16> { static constexpr unsigned ONE = 0x3C00u; };

template <typename TIN, bool ACT>
__global__ __launch_bounds__(512) void conv_v7_kernel(ConvArgs p, V7Plan q) {
    typedef Mfma<TIN> MF;
    typedef typename MF::frag frag;
    constexpr int TM = 2, TN = 4, KSTEPS = 2, KB = 32;     // per wave: 2 tile rows x 4 channel blocks of 32; 2 MFMA k-steps per chunk
    constexpr int HPITCH = V7_HPITCH, ROWB = V7_ROWB, HALO = V7_HALO;
    static_assert(sizeof(TIN) == 2, "16-bit storage only (the fp32 parity mode runs on conv_v4)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int part = tid & 3, pix0 = tid >> 2;

    // ---- which units: workgroup -> (item, contiguous unit range); XCD-contiguous logical order ------------------------------------
    int L = blockIdx.x;
    if ((gridDim.x & 7) == 0) L = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int b = L / q.wpi, wg = L - b * q.wpi;
    int u = (int)(((long long)wg * q.units) / q.wpi);
    const int u_end = (int)(((long long)(wg + 1) * q.units) / q.wpi);
    if (u >= u_end) return;

#ifdef USE_HIP_TRACE_BUILD   /* bring-up: lane 0 of waves 0 and 4 of workgroup p.dbg stamp the cycle counter at coarse boundaries */
    const bool tracing = p.trace != nullptr && (int)blockIdx.x == (p.dbg & 0xffff) && lane == 0 && (wave & 3) == 0;
    int trace_n = 0;
#define V7_STAMP(ID)                                                                                   \
    if (tracing && trace_n < 120) {                                                                    \
        p.trace[(wave >> 2) * 256 + 2 * trace_n] = (unsigned long long)(ID);                           \
        p.trace[(wave >> 2) * 256 + 2 * trace_n + 1] = __builtin_readcyclecounter(); ++trace_n;        \
    }
#define V7_PST(ID) if (trace_fine) { V7_STAMP(ID) }
    bool trace_fine = false;
#else
#define V7_STAMP(ID)
#define V7_PST(ID)
#endif
#ifdef USE_HIP_TRACE_BUILD
    const bool abl_nostore = (p.dbg >> 16) & 1, abl_nostats = (p.dbg >> 17) & 1, abl_noepi = (p.dbg >> 18) & 1, abl_stagger = (p.dbg >> 19) & 1;
    if (abl_stagger) { for (int k = 0; k < (int)((blockIdx.x >> 3) & 3) * 40; ++k) __builtin_amdgcn_s_sleep(127); }
#else
    constexpr bool abl_nostore = false, abl_nostats = false, abl_noepi = false;
#endif
    V7_STAMP(1)
    const int Ctot = p.C0 + p.C1, nchunks = Ctot / V7_CK;
    const size_t img_px = (size_t)p.H * p.W;

    // unit -> tile origin and channel block
    auto unit_geom = [&](int uu, int& ty0, int& tx0, int& n0) {
        const int tile = uu / q.nb, nbi = uu - tile * q.nb;
        const int ty = tile / q.tiles_x;
        ty0 = ty * V7_TH; tx0 = (tile - ty * q.tiles_x) * V7_TW; n0 = nbi * V7_BN;
    };
    int ty0, tx0, n0, ty0n = 0, tx0n = 0, n0n = 0;
    unit_geom(u, ty0, tx0, n0);

    // ---- per-thread halo pieces: piece j of a chunk = 16 bytes of halo pixel j*128 + pix0 (row-major over the 18 x 34 halo) --------
    // ppix[j]: pixel offset inside the item's image, or -1 outside the image / beyond the halo (the zero padding of the convolution
    // applies AFTER the activation: such pieces are zeroed by the mask derived from the sign)
    int ppix[5];
    auto set_pieces = [&](int t_y0, int t_x0) {
        int pix0_o = pix0;                                   // opaque: recomputed per unit (a few dozen VALU instructions) instead of
        asm volatile("" : "+v"(pix0_o));                     // ten hoisted loop invariants spilled around the main loop
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int pix = j * 128 + pix0_o;
            const int hy = pix / V7_HW, hx = pix - hy * V7_HW;
            const int gy = t_y0 + hy - 1, gx = t_x0 + hx - 1;
            const bool inb = (j < 4 || tid < 400) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            ppix[j] = inb ? gy * p.W + gx : -1;
        }
    };
    set_pieces(ty0, tx0);
    const int pdst0 = pix0 * ROWB + part * 16;              // + j * 10240 (HPITCH = 34 x ROWB: the halo image is linear in the pixel index)

    // LDS byte offsets of this lane's fragments
    const int a_base = (wave * 2) * HPITCH + (lane & 31) * ROWB + (lane >> 5) * 16;           // pixel fragments; row i: + i * HPITCH
    const int wsw = (lane >> 2) & 3;                                                            // piece swizzle of the weight rows
    const int w_base0 = V7_OFF_W + (lane & 31) * 64 + (((lane >> 5)) ^ wsw) * 16;               // k-step 0; channel block j: + j * 2048
    const int w_base1 = V7_OFF_W + (lane & 31) * 64 + ((2 + (lane >> 5)) ^ wsw) * 16;           // k-step 1

    float2* const coef_lds = reinterpret_cast<float2*>(smem + V7_OFF_COEF);
    float* const bias_lds = reinterpret_cast<float*>(smem + V7_OFF_BIAS);
    unsigned long long* const tot_lds = reinterpret_cast<unsigned long long*>(smem + V7_OFF_TOT);

    // GroupNorm affine of the chunk being staged: (a, b) pairs of this thread's 8 channels exactly as the table holds them - four
    // 16-byte registers that double as staging registers of the shortcut segment (which has no transform)
    uint4 cq[4];
    auto load_coef = [&](int chunk) {
        int part_o = part;                                   // opaque: the table base is recomputed (1 VALU) instead of living in - or
        asm volatile("" : "+v"(part_o));                     // being spilled from - a register across the epilogue
        const uint4* cf = reinterpret_cast<const uint4*>(coef_lds + chunk * V7_CK + part_o * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) cq[k] = cf[k];
    };
    auto transform = [&](const uint4& raw, unsigned mask) -> uint4 {
        float ca[8], cb[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ca[2 * k] = __builtin_bit_cast(float, cq[k].x); cb[2 * k] = __builtin_bit_cast(float, cq[k].y);
            ca[2 * k + 1] = __builtin_bit_cast(float, cq[k].z); cb[2 * k + 1] = __builtin_bit_cast(float, cq[k].w);
        }
        return stage_transform<TIN, ACT>(raw, mask, ca, cb);
    };
    auto buf_ld = [&](const void* base, unsigned voff, unsigned soff) -> uint4 {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
    };
    const TIN* const src0_b = (const TIN*)p.src0 + (size_t)b * img_px * p.C0;
    const TIN* const src1_b = p.src1 ? (const TIN*)p.src1 + (size_t)b * img_px * p.C1 : nullptr;
    auto src_ld = [&](int chunk, int pixoff) -> uint4 {
        const int c_glob = chunk * V7_CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.C0) { src = src0_b; Cs = p.C0; c_loc = c_glob; }
        else               { src = src1_b; Cs = p.C1; c_loc = c_glob - p.C0; }
        const unsigned voff = (unsigned)max(pixoff, 0) * (unsigned)(Cs * 2) + (unsigned)(part * 16);
        return buf_ld(src, voff, (unsigned)(c_loc * 2));
    };

    // ---- weights: slab-major copy [tap][chunk][cout_pad][32], rows piece-swizzled in the blob: a linear 8 KB copy per slab ---------
    const unsigned wvoff = (unsigned)tid * 16u;
    const unsigned slab_b = (unsigned)(p.cout_pad * V7_CK) * 2u;
    auto w_ld = [&](int tap, int chunk, int n0w) -> uint4 {
        return buf_ld(p.wb, wvoff, (unsigned)(tap * nchunks + chunk) * slab_b + (unsigned)(n0w * V7_CK) * 2u);
    };
    auto w_st = [&](int buf, const uint4& r) { *reinterpret_cast<uint4*>(smem + V7_OFF_W + buf * V7_WSLAB + tid * 16) = r; };

    auto fill_bias = [&](int n0w) {                          // threads 0..127
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int co = n0w + tid_o;
        float add = 0.f;
        if (co < p.Cout) {
            if (p.bias) add += p.bias[co];
            if (p.temb) add += p.temb[(size_t)b * p.temb_bstride + co];
        }
        bias_lds[tid_o] = add;
    };

    // ---- prologue (once per workgroup): chunk 0 of the first unit synchronously, weights of iterations 0 and 1 ----------------------
    uint4 wa = make_uint4(0, 0, 0, 0);
    {
        uint4 w0, raw[5];
        w0 = w_ld(0, 0, n0);
        wa = w_ld(1, 0, n0);
#pragma unroll
        for (int j = 0; j < 5; ++j) raw[j] = src_ld(0, ppix[j]);
        gn_fill_table(coef_lds, p, b, Ctot, tid, 512);
        if (tid < V7_BN) fill_bias(n0);
        for (int c = tid; c < 2 * p.Cout; c += 512) tot_lds[c] = 0ull;
        __syncthreads();
        load_coef(0);
        w_st(0, w0);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned m = ~(unsigned)(ppix[j] >> 31);
            const uint4 t = transform(raw[j], m);
            if (j < 4 || tid < 400) *reinterpret_cast<uint4*>(smem + pdst0 + j * 10240) = t;
        }
    }

    f32x16 acc[TM][TN];
    frag af[KSTEPS][TM], bf[KSTEPS][TN];                      // af: pixel fragments (MFMA B operand), bf: weight fragments (A operand)
    uint4 hLa = wa, hLb = wa, hT = wa, t0 = wa;              // pieces in flight (even / odd), piece being transformed, transformed piece
    unsigned mT = 0;                                         // mask of the piece being transformed
    int gpar = 0;                                            // halo buffer the current chunk reads (flips per 3x3 chunk)
    int wpar = 0;                                            // weight buffer the current iteration reads is wpar ^ (T & 1) (flips per iteration)
    bool nx_valid = true;                                    // a next chunk exists (this unit's next chunk, or chunk 0 of the next unit)
    int c_next = 0, n0_w2 = n0;                              // next chunk index; channel block of the iterations that wrap past this chunk
    int c = 0;

#define V7_BAR() { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); }

    const size_t img_elems = img_px * p.Cout;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((TIN*)p.out + (size_t)b * img_elems, 0, 0x7fffffff, 0x00020000);
    const bool has_stats = p.stats != nullptr;

    // one-hot fragments: element e of lane (x = lane & 31, h = lane >> 5) is 1 where x == 16 gp + 8 h + e.  Rebuilt where they are
    // used (a dozen VALU instructions) instead of living in 16 registers across the main loop.
    auto one_hot = [&](int lane_o, int gp, unsigned one) -> uint4 {
        const int e = (lane_o & 31) - 16 * gp - 8 * (lane_o >> 5);
        const unsigned pat = (e & 1) ? one << 16 : one;
        return make_uint4((e >> 1) == 0 ? pat : 0u, (e >> 1) == 1 ? pat : 0u, (e >> 1) == 2 ? pat : 0u, (e >> 1) == 3 ? pat : 0u);
    };
    auto out_offsets = [&](int lane_o, int t_y0, int t_x0, int n0w, unsigned (&voff)[TM]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gy = t_y0 + wave_u * 2 + i;
            voff[i] = (unsigned)(((gy * p.W + t_x0 + (lane_o & 31)) * p.Cout + n0w) * 2 + (lane_o >> 5) * 16);
        }
    };
    // ---- shortcut segment (the res-block's 1x1 Conv_2 on the raw block input - or the residual, as an identity matrix): XC / 32 extra
    // chunks of ONE tap each after the 3x3 chunks.  Its operand is the tile's 512 raw centre pixels, 4 pieces per thread per chunk,
    // staged through the registers the halo pipeline leaves idle (no transform here): chunk s is loaded in iteration s-2, written to
    // LDS in iteration s-1 (even chunks: the region S0; odd chunks: the halo buffer the last 3x3 chunk has left), read in iteration s.
    const int nsc = (p.XC0 + p.XC1) / V7_CK;
    const TIN* const x0_b = nsc ? (const TIN*)p.x0 + (size_t)b * img_px * p.XC0 : nullptr;
    const TIN* const x1_b = (nsc && p.x1) ? (const TIN*)p.x1 + (size_t)b * img_px * p.XC1 : nullptr;
    auto sc_ld = [&](int chunk2, int qq) -> uint4 {
        int pix0_o = pix0;
        asm volatile("" : "+v"(pix0_o));
        const int c_glob = chunk2 * V7_CK;
        const TIN* src; int Cs, c_loc;
        if (c_glob < p.XC0) { src = x0_b; Cs = p.XC0; c_loc = c_glob; }
        else                { src = x1_b; Cs = p.XC1; c_loc = c_glob - p.XC0; }
        const int pixoff = (ty0 + qq * 4 + (pix0_o >> 5)) * p.W + tx0 + (pix0_o & 31);
        return buf_ld(src, (unsigned)pixoff * (unsigned)(Cs * 2) + (unsigned)(part * 16), (unsigned)(c_loc * 2));
    };
    auto sc_st = [&](int sbase, const uint4& r0, const uint4& r1, const uint4& r2, const uint4& r3) {
        int pix0_o = pix0;
        asm volatile("" : "+v"(pix0_o));
        char* d = smem + sbase + pix0_o * 64 + ((part ^ ((pix0_o >> 2) & 3)) * 16);       // 64-byte pixel rows, piece-swizzled like the weights
        *reinterpret_cast<uint4*>(d) = r0; *reinterpret_cast<uint4*>(d + 8192) = r1;
        *reinterpret_cast<uint4*>(d + 16384) = r2; *reinterpret_cast<uint4*>(d + 24576) = r3;
    };
    auto w2_ld = [&](int chunk2) -> uint4 {
        return buf_ld(p.w2b, wvoff, (unsigned)chunk2 * slab_b + (unsigned)(n0 * V7_CK) * 2u);
    };
    bool sc_now = false;                                     // this is the unit's last 3x3 chunk and a shortcut segment follows
    // Piece k (0..4) of the next chunk: global load issued in LDS(k) -> parked in plain registers in LDS(k+2) (two pieces in flight:
    // at a unit switch the whole chip asks for new pixels at once and one iteration does not cover that latency) -> GroupNorm+SiLU on
    // the VALU behind the MFMAs of MFMA(k+2) -> written to the other halo buffer in LDS(k+3).
    auto lds_phase = [&](auto Tc) {
        constexpr int T = decltype(Tc)::value;
        {
            const char* ha_ = smem + gpar * HALO + (T / 3) * HPITCH + (T % 3) * ROWB + a_base;
            const int wbuf_ = (wpar ^ (T & 1)) * V7_WSLAB;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = MF::ld(ha_ + i * HPITCH + kk * KB);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = MF::ld(smem + wbuf_ + (kk ? w_base1 : w_base0) + j * 2048);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (T >= 3 && T < 8) {
            constexpr int k_ = T - 3;
            if (nx_valid && (k_ < 4 || tid < 400)) *reinterpret_cast<uint4*>(smem + (gpar ^ 1) * HALO + pdst0 + k_ * 10240) = t0;
        }
        if (T < 8 || nx_valid || sc_now) w_st((wpar ^ (T & 1)) ^ 1, wa);
        if constexpr (T >= 2 && T < 7) {
            // the piece loaded two iterations ago has landed: park it in plain registers so that the MFMA-phase transform carries no
            // vmcnt wait on the fresh loads
            hT = ((T - 2) & 1) ? hLb : hLa; mT = ~(unsigned)(ppix[T - 2] >> 31);
            asm volatile("" : "+v"(hT.x), "+v"(hT.y), "+v"(hT.z), "+v"(hT.w), "+v"(mT));
        }
        if constexpr (T < 5) {
            if (nx_valid) {
                if constexpr (T == 0) load_coef(c_next);
                if constexpr (T & 1) hLb = src_ld(c_next, ppix[T]); else hLa = src_ld(c_next, ppix[T]);
            }
        }
        if constexpr (T == 7) {
            if (sc_now) { hLa = sc_ld(0, 0); hLb = sc_ld(0, 1); hT = sc_ld(0, 2); t0 = sc_ld(0, 3); }
        }
        if constexpr (T == 8) {
            if (sc_now) {
                sc_st(V7_OFF_S, hLa, hLb, hT, t0);
                cq[0] = sc_ld(1, 0); cq[1] = sc_ld(1, 1); cq[2] = sc_ld(1, 2); cq[3] = sc_ld(1, 3);
            }
        }
        // weights of iteration T+2 (wraps into the next chunk / the shortcut segment / the next unit's chunk 0)
        if constexpr (T + 2 <= 8) wa = w_ld(T + 2, c, n0);
        else if (sc_now) wa = w2_ld(T + 2 - 9);
        else if (nx_valid) wa = w_ld(T + 2 - 9, c_next, n0_w2);
    };
    // one shortcut chunk (parity P of its index is a compile-time constant: it selects the register set and the LDS region)
    auto sc_lds_phase = [&](auto Pc, int sidx, bool more_) {
        constexpr int P = decltype(Pc)::value;
        const int s1base = (gpar ^ 1) * HALO;                // (gpar already points at the next unit's first chunk)
        {
            // w_base carries the lane's swizzled offset (+ V7_OFF_W); opaque copies: the sums are rebuilt here (2 VALU) instead of being
            // hoisted out of the unit loop and spilled around the 3x3 chunks
            int wb0 = w_base0, wb1 = w_base1;
            asm volatile("" : "+v"(wb0), "+v"(wb1));
            const int sb_ = (P ? s1base : V7_OFF_S) + wave_u * 4096 - V7_OFF_W;
            const int sa0 = sb_ + wb0, sa1 = sb_ + wb1;
            const int wbuf_ = wpar * V7_WSLAB;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = MF::ld(smem + (kk ? sa1 : sa0) + i * 2048);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = MF::ld(smem + wbuf_ + (kk ? w_base1 : w_base0) + j * 2048);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (sidx + 1 < nsc) {                                // chunk s+1 (loaded one iteration ago) -> the other region
            if constexpr (P) sc_st(V7_OFF_S, hLa, hLb, hT, t0); else sc_st(s1base, cq[0], cq[1], cq[2], cq[3]);
        }
        if (sidx + 1 < nsc || more_) w_st(wpar ^ 1, wa);
        if (sidx + 2 < nsc) {                                // chunk s+2 -> this chunk's register set
            if constexpr (P) { cq[0] = sc_ld(sidx + 2, 0); cq[1] = sc_ld(sidx + 2, 1); cq[2] = sc_ld(sidx + 2, 2); cq[3] = sc_ld(sidx + 2, 3); }
            else { hLa = sc_ld(sidx + 2, 0); hLb = sc_ld(sidx + 2, 1); hT = sc_ld(sidx + 2, 2); t0 = sc_ld(sidx + 2, 3); }
            wa = w2_ld(sidx + 2);
        } else if (more_) {
            wa = w_ld(sidx + 2 - nsc, 0, n0n);               // the next unit's taps 0 / 1
        }
    };
    auto sc_mfma_phase = [&]() {
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(bf[kk][j], af[kk][i], acc[i][j]);
    };
    auto mfma_phase = [&](auto Tc) {
        constexpr int T = decltype(Tc)::value;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(bf[kk][j], af[kk][i], acc[i][j]);
        if constexpr (T >= 2 && T < 7) {                     // unconditional at run time: same basic block as the MFMAs
            t0 = transform(hT, mT);
            asm volatile("" : "+v"(t0.x), "+v"(t0.y), "+v"(t0.z), "+v"(t0.w));   // materialise here, not at the ds_write
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
                __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);   // 1 TRANS
            }
        }
    };

    // accumulators start at bias + time-embedding bias: register r of channel block j holds channel 32 j + 8 (r >> 2) + 4 h + (r & 3)
    auto init_acc = [&]() {
        // (the table sits beyond the 64 KB reach of a ds_read offset: an opaque base register + small immediates, or hipcc
        // materialises - and keeps alive across the main loop - one address register per read)
        int lane_i = lane;
        asm volatile("" : "+v"(lane_i));
        const int boff = V7_OFF_BIAS + 16 * (lane_i >> 5);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {               // (one read per tile row: cheaper than 64 register moves)
                    int bo = boff;
                    asm volatile("" : "+v"(bo));
                    const float4 bv = *reinterpret_cast<const float4*>(smem + bo + (j * 32 + 8 * g) * 4);
                    acc[i][j][4 * g] = bv.x; acc[i][j][4 * g + 1] = bv.y; acc[i][j][4 * g + 2] = bv.z; acc[i][j][4 * g + 3] = bv.w;
                }
            }
    };

    // ---- epilogue of one unit: no LDS transposition, no barrier ------------------------------------------------------------------------
    auto epilogue = [&]() {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));                     // keeps the one-hot fragments and offsets out of the main loop's live set
        uint4 selu[2], selq[2];
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) { selu[gp] = one_hot(lane_o, gp, OneHot<TIN>::ONE); selq[gp] = one_hot(lane_o, gp, 0x3F80u); }
        unsigned voff[TM];
        out_offsets(lane_o, ty0, tx0, n0, voff);
        const float scale = p.out_scale;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // GroupNorm partial sums of 16 channels on the matrix pipe: (lane = pixel, 8 channels) x OneHot -> accumulator with
                // lane = channel, summed over the wave's 64 pixels; columns 0-15 take the sums (of the stored, rounded values), columns
                // 16-31 the sums of squares (of the fp32 values, rounded to bf16 per element: unbiased, 2^-9 each)
                f32x16 sT;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // X = register group 2gp (channels 16gp + 4h + k), Y = group 2gp+1 (16gp + 8 + 4h + k): after the half exchange
                        // a lane holds channels 16gp + 8h + (0..7) of its pixel
                        // (element copies first: __builtin_bit_cast applied to a vector-element lvalue reads element 0)
                        const float xk = acc[i][j][8 * gp + k], yk = acc[i][j][8 * gp + 4 + k];
                        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xk), __builtin_bit_cast(unsigned, yk), false, false);
                        v[k] = __builtin_bit_cast(float, (unsigned)r[0]) * scale;          // (x 1.0 is exact: no branch on the scale)
                        v[4 + k] = __builtin_bit_cast(float, (unsigned)r[1]) * scale;
                    }
                    const uint4 packed = Vec16<TIN>::pack(v);
                    if (!abl_nostore)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, packed), rs_out,
                                                           voff[i] + (unsigned)(j * 64 + gp * 32), 0, 0);
                    else asm volatile("" :: "v"(packed.x), "v"(packed.y), "v"(packed.z), "v"(packed.w));
                    if (has_stats && !abl_nostats) {
                        bf16x8 sq;
#pragma unroll
                        for (int k = 0; k < 8; ++k) sq[k] = (__bf16)(v[k] * v[k]);
                        if (i == 0) {                             // chain head: C = 0
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            sT = MF::mma(__builtin_bit_cast(frag, packed), __builtin_bit_cast(frag, selu[0]), z);
                        } else {
                            sT = MF::mma(__builtin_bit_cast(frag, packed), __builtin_bit_cast(frag, selu[0]), sT);
                        }
                        sT = Mfma<__bf16>::mma(sq, __builtin_bit_cast(bf16x8, selq[1]), sT);
                    }
                }
                if (has_stats && !abl_nostats) {
                    // in-lane sum of the 16 pixel rows, pairwise (fixed order); then the other half-wave's 16 rows
                    float s8[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) s8[r] = sT[r] + sT[r + 8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) s8[r] += s8[r + 4];
                    float sv = (s8[0] + s8[2]) + (s8[1] + s8[3]);
                    sv = reduce_lanes_stride<32>(sv);
                    if (lane_o < 32) {                            // lanes 0-15: sum of channel 16gp + lane, lanes 16-31: its sum of squares
                        const int co = n0 + j * 32 + gp * 16 + (lane_o & 15);
                        const float fx = (lane_o & 16) ? GN_SQ_SCALE : GN_SUM_SCALE;
                        atomicAdd(tot_lds + co * 2 + ((lane_o >> 4) & 1), (unsigned long long)__float2ll_rn(sv * fx));
                    }
                }
            }
        }
    };

    // ---- the walk: both wave groups run this stream; waves 4-7 one barrier late, so that the two waves of a SIMD are always in
    // opposite phases:   G0: LDS(0) | MFMA(0) | LDS(1) | ...      G1:   --   | LDS(0) | MFMA(0) | ...
    // At a unit's end the groups line up for one interval - G0: epilogue, G1: MFMA(8) + epilogue (the epilogue is latency-, not
    // issue-bound: two at a time on a SIMD cost little more than one) - and G1 then idles through G0's LDS(0) to fall one phase behind.
    V7_STAMP(2)
    V7_BAR();                                                // prologue stores visible
    if (wave_u >= 4) V7_BAR();
    init_acc();
    for (;;) {
        V7_STAMP(10)
        const bool more = u + 1 < u_end;
        for (c = 0; c < nchunks; ++c) {
            const bool last = c + 1 == nchunks;
            if (last && more) {                              // the staging below belongs to the next unit
                unit_geom(u + 1, ty0n, tx0n, n0n);
                set_pieces(ty0n, tx0n);
            }
#ifdef USE_HIP_TRACE_BUILD
            trace_fine = ((p.dbg >> 20) & 1) && trace_n > 20 && trace_n < 100 && (c == 0 || last);
#endif
            nx_valid = !last || more;
            sc_now = last && nsc > 0;
            c_next = last ? 0 : c + 1;
            n0_w2 = last ? n0n : n0;
            lds_phase(std::integral_constant<int, 0>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 0>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 1>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 1>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 2>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 2>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 3>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 3>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 4>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 4>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 5>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 5>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 6>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 6>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 7>{});
            if (last && more && q.nb > 1 && tid < V7_BN) fill_bias(n0n);      // read after the epilogue, many barriers from here
            V7_BAR(); mfma_phase(std::integral_constant<int, 7>{}); V7_BAR();
            lds_phase(std::integral_constant<int, 8>{}); V7_BAR(); mfma_phase(std::integral_constant<int, 8>{});
            gpar ^= 1; wpar ^= 1;
            V7_STAMP(20 + c)
            if (!last || nsc > 0) V7_BAR();
        }
        for (int sidx = 0; sidx < nsc; sidx += 2) {          // shortcut chunks, two per trip (static register sets)
            sc_lds_phase(std::integral_constant<int, 0>{}, sidx, more); V7_BAR(); sc_mfma_phase(); V7_BAR(); wpar ^= 1;
            sc_lds_phase(std::integral_constant<int, 1>{}, sidx + 1, more); V7_BAR(); sc_mfma_phase(); wpar ^= 1;
            if (sidx + 2 < nsc) V7_BAR();
        }
        if (wave_u < 4) V7_BAR();                            // G0's MFMA(8) barrier; G1 runs on into its epilogue
        if (!abl_noepi) epilogue();
        V7_STAMP(40)
        if (!more) break;
        ++u; ty0 = ty0n; tx0 = tx0n; n0 = n0n;
        init_acc();
        V7_STAMP(41)
        V7_BAR();
        if (wave_u >= 4) V7_BAR();                           // G1 sits out G0's LDS(0)
    }
    V7_STAMP(50)
    __syncthreads();                                         // every wave's LDS totals are in
    if (has_stats) {
        for (int co = tid; co < p.Cout; co += 512) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.stats + ((size_t)b * p.Cout + co) * 2);
            atomicAdd(dst, tot_lds[co * 2]);
            atomicAdd(dst + 1, tot_lds[co * 2 + 1]);
        }
    }
    V7_STAMP(60)
#undef V7_BAR
#undef V7_STAMP
}

template <typename TIN, bool ACT>
static void v7_launch_t(const ConvArgs& a, const V7Plan& q, int grid, hipStream_t s) {
    static bool attr_set = false;
    auto kern = conv_v7_kernel<TIN, ACT>;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, V7_SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), V7_SMEM, s, a, q);
}

static long g_v7_min_units = 0;     // 0: off
static long g_v7_max_units = 1L << 40;
static int g_v7_cus = 0;
static int g_v7_modes = 7;          // bit 0: plain convolutions, 1: with a residual, 2: with a fused 1x1 shortcut
void conv_v7_set_min_units(long n) { g_v7_min_units = n; }
void conv_v7_set_max_units(long n) { g_v7_max_units = n; }
void conv_v7_set_modes(int m) { g_v7_modes = m; }
static int g_v7_upw = 0;            // units per workgroup (0: as many as an even split over the CUs gives)
void conv_v7_set_units_per_wg(int n) { g_v7_upw = n; }

// The residual of a convolution is served as an identity shortcut: out = conv3x3(in) + I x res.  Identity slabs in the slab-major,
// piece-swizzled weight layout ([chunk][cout_pad][32]; element (n, k = n) of chunk n / 32), one per (Cout, dtype), built on first use -
// the engine asks for them at plan time (conv_v7_prepare), never inside a stream capture.
struct V7Ident { int cout, dtype; void* dev; };
static V7Ident g_v7_ident[8];
static int g_v7_nident = 0;
static const void* v7_identity(int cout, int dtype, bool create) {
    for (int i = 0; i < g_v7_nident; ++i)
        if (g_v7_ident[i].cout == cout && g_v7_ident[i].dtype == dtype) return g_v7_ident[i].dev;
    if (!create || g_v7_nident == 8) return nullptr;
    const int nch = cout / V7_CK;
    const size_t elems = (size_t)nch * cout * V7_CK;
    unsigned short* h = (unsigned short*)calloc(elems, 2);
    if (!h) return nullptr;
    const unsigned short one = dtype == DT_BF16 ? 0x3F80 : 0x3C00;
    for (int n = 0; n < cout; ++n) {
        const int ch = n / V7_CK, kk = n % V7_CK, q = kk / 8, e = kk % 8, pos = q ^ ((n >> 2) & 3);
        h[((size_t)ch * cout + n) * V7_CK + pos * 8 + e] = one;
    }
    void* d = nullptr;
    if (hipMalloc(&d, elems * 2) != hipSuccess) { free(h); return nullptr; }
    (void)hipMemcpy(d, h, elems * 2, hipMemcpyHostToDevice);
    free(h);
    g_v7_ident[g_v7_nident++] = V7Ident{cout, dtype, d};
    return d;
}
void conv_v7_prepare(int cout, int dtype) {
    if (dtype != DT_F32 && cout % V7_BN == 0 && cout <= 256) (void)v7_identity(cout, dtype, true);
}

bool conv_v7_supports(const ConvArgs& a) {
    const int Ctot = a.C0 + a.C1, XC = a.XC0 + a.XC1;
    const bool sc_ok = XC == 0 || (a.w2b != nullptr && a.res == nullptr && XC % (2 * V7_CK) == 0 && (a.XC1 == 0 || a.XC0 % V7_CK == 0));
    return a.wb != nullptr && a.ntaps == 9 && a.in_dtype != DT_F32 && a.in_dtype == a.out_dtype && sc_ok && a.pyr == nullptr &&
           Ctot % V7_CK == 0 && Ctot >= 2 * V7_CK && Ctot <= 512 && (a.C1 == 0 || a.C0 % V7_CK == 0) && a.Cout % V7_BN == 0 && a.Cout <= 256 &&
           a.cout_pad == a.Cout && a.H % V7_TH == 0 && a.W % V7_TW == 0;
}
bool conv_v7_eligible(const ConvArgs& a) {
    if (g_v7_min_units <= 0) return false;
    const long units = (long)conv_v4_tiles(a.H, a.W) * ((a.Cout + V7_BN - 1) / V7_BN);
    if (units < g_v7_min_units || units > g_v7_max_units || !conv_v7_supports(a)) return false;
    const int mode = a.res ? 2 : (a.XC0 + a.XC1) ? 4 : 1;
    if (!(g_v7_modes & mode)) return false;
    return a.res == nullptr || v7_identity(a.Cout, a.in_dtype, false) != nullptr;     // (the identity slabs must exist already: no allocation here)
}

void launch_conv_v7(const ConvArgs& a0, hipStream_t s) {
    if (!g_v7_cus) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        g_v7_cus = n > 0 ? n : 256;
    }
    ConvArgs a = a0;
    if (a.res) {                                             // residual -> identity shortcut (exact: one product with 1.0 per element)
        a.x0 = a.res; a.XC0 = a.Cout; a.x1 = nullptr; a.XC1 = 0; a.w2b = v7_identity(a.Cout, a.in_dtype, true); a.res = nullptr;
    }
    V7Plan q;
    q.nb = a.Cout / V7_BN;
    q.tiles_x = a.W / V7_TW;
    q.units = conv_v4_tiles(a.H, a.W) * q.nb;
    q.wpi = g_v7_cus / a.B;
    if (q.wpi < 1) q.wpi = 1;
    // bounded walks: a workgroup that holds its CU for the whole launch keeps the other sub-batch's small kernels waiting
    if (g_v7_upw > 0) q.wpi = std::max(q.wpi, (q.units + g_v7_upw - 1) / g_v7_upw);
    if (q.wpi > q.units) q.wpi = q.units;
    const int grid = q.wpi * a.B;
    if (a.in_dtype == DT_BF16) { a.act ? v7_launch_t<__bf16, true>(a, q, grid, s) : v7_launch_t<__bf16, false>(a, q, grid, s); }
    else                       { a.act ? v7_launch_t<_Float16, true>(a, q, grid, s) : v7_launch_t<_Float16, false>(a, q, grid, s); }
}

}  // namespace use
