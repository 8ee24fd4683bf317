// conv_sk_kernel: the convolutions of the small maps (32x40 pixels and below: the four deepest levels of the U-Net and every
// level of short test shapes).  Same operator as conv_kernel (use_kernels.hip; reference layers.py:113-162 + layerspp.py:282-314
// fused around it), different parallel decomposition.
//
// On these maps a tile-per-workgroup schedule leaves the chip empty (8 ... 48 workgroups for 256 CUs) and each workgroup walks
// K = 2304 ... 4608 (+ shortcut) serially behind one global-load latency per step: 41 ... 55 us per launch whatever the map size.
// Here the reduction is what gets parallelised:
//   * a workgroup (8 waves) owns a 64-pixel x 32- or 64-channel output tile and its 8 waves split K: wave w takes the
//     (tap, 32-channel chunk) units w, w + 8, ... of the pass and accumulates a full-tile partial sum in its own registers;
//   * the weights of a wave's units go straight from global memory into MFMA operand registers (each weight element is used by
//     exactly one wave of the workgroup, LDS staging would buy nothing) and are requested before anything else, so their HBM
//     latency overlaps the GroupNorm table and the halo staging;
//   * the input halo of the pass (<= 128 pixels x <= 256 channels) is normalised + activated once, cooperatively, into LDS and
//     read as MFMA fragments by all waves; K passes of CP channels bound the LDS and register footprint;
//   * the eight partial tiles are summed through LDS in a fixed order (4 + 4, then 4 -> 1: deterministic), and the last step is
//     conv_kernel's epilogue: bias + temb, residual, 1/sqrt(2), Combine, 16-byte stores, GroupNorm totals of the stored values.
// fp32 parity mode uses v_mfma_f32_32x32x2_f32 on 16-byte pieces as well (a piece = 4 consecutive channels = 4 MFMAs).
#include "use_kernels.h"
#include <type_traits>
#include "use_device.h"

namespace use {

namespace {

// one 16-byte piece of A and of B -> MFMA(s).  Within a 32-channel unit, lanes 0-31 hold channels [8q', 8q'+8) (16-bit) resp.
// [4q', 4q'+4) (fp32) of the even 16-byte slots and lanes 32-63 those of the odd slots; A and B use the same assignment.
template <typename T> struct SkMma;
template <> struct SkMma<__bf16> {
    DEVI static f32x16 mma(const uint4 a, const uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct SkMma<_Float16> {
    DEVI static f32x16 mma(const uint4 a, const uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <> struct SkMma<float> {
    DEVI static f32x16 mma(const uint4 a, const uint4 b, f32x16 c) {
        const float4 fa = __builtin_bit_cast(float4, a), fb = __builtin_bit_cast(float4, b);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, c, 0, 0, 0);
        return c;
    }
};

constexpr int SK_HPMAX = 128;                    // halo pixels of a tile (host picks the tile shape accordingly)
constexpr int SK_MT = 64;                        // output pixels per workgroup

template <typename TIN, typename TOUT, int NJ, int CP>
__global__ __launch_bounds__(512) void conv_sk_kernel(ConvArgs p, int TH, int TW) {
    constexpr int ES = (int)sizeof(TIN);
    constexpr int VEC = 16 / ES;
    constexpr int NP = ES;                                   // 16-byte pieces per lane per 32-channel unit
    constexpr int MI = SK_MT / 32, MT = SK_MT, NT = 32 * NJ;
    constexpr int ROWB = CP * ES + 16;                       // LDS bytes per halo pixel (odd multiple of 16: conflict-free b128 reads)
    constexpr int UMAX = (9 * (CP / 32) + 7) / 8;            // units per wave per pass
    constexpr int NPC = SK_HPMAX * (CP * ES / 16) / 512;     // halo pieces per thread per pass
    constexpr int NTP = NT + 8;                              // fp32 row pitch of a partial tile (16 k + 8: conflict-free writes)
    constexpr int HALO_BYTES = SK_HPMAX * ROWB, RED_BYTES = 4 * MT * NTP * 4;
    constexpr int SMEM = HALO_BYTES > RED_BYTES ? HALO_BYTES : RED_BYTES;
    constexpr bool ACC = ES == 4;                            // fp32 parity mode: accurate SiLU
    static_assert(NPC >= 1 && SK_HPMAX * (CP * ES / 16) % 512 == 0, "halo staging");
    __shared__ __attribute__((aligned(16))) char smem[SMEM];
    __shared__ float2 coef_s[1024];
    __shared__ float2 st_red[8][NT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, n0 = blockIdx.y * NT;
    const int pad = p.ntaps == 9 ? 1 : 0;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH, tx0 = (blockIdx.x % tiles_x) * TW;
    const int HW_ = TW + 2 * pad, HP = HW_ * (TH + 2 * pad);
    const int Ctot = p.C0 + p.C1, XCtot = p.XC0 + p.XC1;
    const bool has_gn = p.coef || p.gn_st0;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[MI];                                           // this lane's pixel rows of the tile, as halo byte offsets (tap 0,0)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        int m = i * 32 + (lane & 31);
        if (m >= TH * TW) m = 0;                             // rows past the tile: any valid address, discarded by the epilogue
        const int ty = m / TW;
        a_off[i] = (ty * HW_ + (m - ty * TW)) * ROWB + (lane >> 5) * 16;
    }

    bool first = true;
    // the four K sources in order: the (normalised, activated) input concat(src0, src1) under all taps, then the raw shortcut
    // input concat(x0, x1) under the centre tap (Conv_2 fused into Conv_1)
    for (int si = 0; si < 4; ++si) {
        const bool seg1 = si >= 2;
        const TIN* src = (const TIN*)(si == 0 ? p.src0 : si == 1 ? p.src1 : si == 2 ? p.x0 : p.x1);
        const int Cs = si == 0 ? p.C0 : si == 1 ? p.C1 : si == 2 ? p.XC0 : p.XC1;
        const int cg0 = si == 1 ? p.C0 : si == 3 ? p.XC0 : 0;  // first channel of this source within its concatenation
        const char* wbase = seg1 ? (const char*)p.w2 : (const char*)p.w;
        const int wld = seg1 ? XCtot : Ctot, wtaps = seg1 ? 1 : p.ntaps;
        const bool use_coef = has_gn && !seg1, use_act = p.act && !seg1;
        const char* wslab = seg1 ? (const char*)p.w2b : (const char*)p.wb;   // slab-major copy (null: NIN layers)
        constexpr int SCK = ES == 4 ? 16 : 32;                   // channels per slab row (conv_v4_chunk): 64-byte rows
        const int nslabs = wld / SCK;
        for (int c_loc = 0; c_loc < Cs; c_loc += CP) {
            const int pc = Cs - c_loc < CP ? Cs - c_loc : CP;    // channels of this pass (multiple of 32)
            const int cpu_ = pc >> 5;                            // 32-channel chunks
            const int nunits = wtaps * cpu_;
            const int c_glob = cg0 + c_loc;
            // units of this wave: u = wave + 8 i -> (tap, chunk), stepped without divisions
            const int d_tap = 8 / cpu_, d_chunk = 8 - d_tap * cpu_;
            const int tap0 = wave / cpu_, chunk0 = wave - tap0 * cpu_;
            // 1. this wave's weights of the pass -> registers.  Slab-major copy: the 32 rows of a (tap, chunk, channel tile) are
            // 2 KB contiguous (16 cache lines per load instruction instead of 32 with the row-major layout)
            uint4 bw[UMAX][NJ][NP];
            {
                int tap = tap0, chunk = chunk0;
#pragma unroll
                for (int i = 0; i < UMAX; ++i) {
                    if (wave + 8 * i < nunits) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            int row = n0 + j * 32 + (lane & 31);
#ifdef USE_HIP_ABLATE
                            if (p.dbg & 1) row = n0 + j * 32 + (lane & 1);
#endif
                            if (wslab) {
#pragma unroll
                                for (int q = 0; q < NP; ++q) {
                                    const int cgl = c_glob + chunk * 32 + (2 * q + (lane >> 5)) * VEC;   // first channel of this lane's piece
                                    const int slab = cgl / SCK, piece = (cgl % SCK) / VEC;
                                    bw[i][j][q] = *reinterpret_cast<const uint4*>(
                                        wslab + (((size_t)tap * nslabs + slab) * p.cout_pad + row) * 64 + ((piece ^ ((row >> 2) & 3)) << 4));
                                }
                            } else {
                                const char* wp = wbase + (((size_t)row * wtaps + tap) * wld + c_glob + chunk * 32) * ES + (lane >> 5) * 16;
#pragma unroll
                                for (int q = 0; q < NP; ++q) bw[i][j][q] = *reinterpret_cast<const uint4*>(wp + q * 32);
                            }
                        }
                    }
                    tap += d_tap; chunk += d_chunk;
                    if (chunk >= cpu_) { chunk -= cpu_; ++tap; }
                }
            }
            if (first) {
                if (has_gn) gn_fill_table(coef_s, p, b, Ctot, tid, 512);
                first = false;
            }
            __syncthreads();                                     // table visible; fragment reads of the previous pass done
            // 2. halo of the pass: raw loads first (all in flight), then normalise + activate + store.  Piece idx = tid + 512 k
            // -> (pixel, 16-byte part) -> (halo row, column), stepped without divisions
            const int ppp = pc * ES / 16;                        // 16-byte pieces per pixel
            const int npieces = HP * ppp;
            const int d_pix = 512 / ppp, d_part = 512 - d_pix * ppp;
            const int d_hy = d_pix / HW_, d_hx = d_pix - d_hy * HW_;
            uint4 raw[NPC];
            int loff[NPC];                                       // LDS byte offset | in-image flag (bit 30) | valid flag (bit 31)
            {
                int pix = tid / ppp, part = tid - pix * ppp;
                int hy = pix / HW_, hx = pix - hy * HW_;
#pragma unroll
                for (int k = 0; k < NPC; ++k) {
                    raw[k] = make_uint4(0, 0, 0, 0);
                    loff[k] = 0;
                    if (tid + k * 512 < npieces) {
                        const int gy = ty0 + hy - pad, gx = tx0 + hx - pad;
                        const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                        loff[k] = (pix * ROWB + part * 16) | (inb ? 0x40000000 : 0) | 0x80000000;
                        if (inb) raw[k] = *reinterpret_cast<const uint4*>(src + ((size_t)(b * p.H + gy) * p.W + gx) * Cs + c_loc + part * VEC);
                    }
                    part += d_part; pix += d_pix; hx += d_hx; hy += d_hy;
                    if (part >= ppp) { part -= ppp; ++pix; ++hx; }
                    if (hx >= HW_) { hx -= HW_; ++hy; }
                }
            }
            {
                int part = tid % ppp;
#pragma unroll
                for (int k = 0; k < NPC; ++k) {
                    if (loff[k] < 0) {                           // valid piece
                        uint4 o = raw[k];
#ifdef USE_HIP_ABLATE
                        if (p.dbg & 2) { *reinterpret_cast<uint4*>(smem + (loff[k] & 0x3fffffff)) = o; continue; }
#endif
                        if ((loff[k] & 0x40000000) && (use_coef || use_act)) {   // zero padding applies after the activation: outside stays 0
                            float v[VEC];
                            Vec16<TIN>::load(reinterpret_cast<const TIN*>(&o), v);
                            if (use_coef) {
                                const float2* cf = coef_s + c_glob + part * VEC;
#pragma unroll
                                for (int e = 0; e < VEC; ++e) { const float2 ab = cf[e]; v[e] = fmaf(v[e], ab.x, ab.y); }
                            }
                            if (use_act) {
#pragma unroll
                                for (int e = 0; e < VEC; ++e) v[e] = silu_f<ACC>(v[e]);
                            }
                            o = Vec16<TIN>::pack(v);
                        }
                        *reinterpret_cast<uint4*>(smem + (loff[k] & 0x3fffffff)) = o;
                    }
                    part += d_part;
                    if (part >= ppp) part -= ppp;
                }
            }
            __syncthreads();
            // 3. this wave's units
            {
                int tap = tap0, chunk = chunk0;
#pragma unroll
                for (int i = 0; i < UMAX; ++i) {
#ifdef USE_HIP_ABLATE
                    if (p.dbg & 4) { if (wave + 8 * i < nunits) { _Pragma("unroll") for (int j = 0; j < NJ; ++j) _Pragma("unroll") for (int q = 0; q < NP; ++q) acc[0][j][q] += __builtin_bit_cast(float, bw[i][j][q].x); } continue; }
#endif
                    if (wave + 8 * i < nunits) {
                        const int dy = seg1 ? pad : (pad ? (tap * 11) >> 5 : 0);   // tap / 3 for tap <= 8
                        const int dx = seg1 ? pad : (pad ? tap - dy * 3 : 0);
                        const char* ha = smem + (dy * HW_ + dx) * ROWB + chunk * 32 * ES;
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            uint4 af[MI];
#pragma unroll
                            for (int m = 0; m < MI; ++m) af[m] = *reinterpret_cast<const uint4*>(ha + a_off[m] + q * 32);
#pragma unroll
                            for (int m = 0; m < MI; ++m)
#pragma unroll
                                for (int j = 0; j < NJ; ++j) acc[m][j] = SkMma<TIN>::mma(af[m], bw[i][j][q], acc[m][j]);
                        }
                    }
                    tap += d_tap; chunk += d_chunk;
                    if (chunk >= cpu_) { chunk -= cpu_; ++tap; }
                }
            }
        }
    }

    // ------------------------------ reduction over the 8 K slices ------------------------------
    float* const red = reinterpret_cast<float*>(smem);
    float* const mine = red + (wave & 3) * (MT * NTP);
    __syncthreads();                                             // all fragment reads done: the halo region becomes the buffer
#define SK_FOR_ACC(BODY)                                                                                   \
    _Pragma("unroll") for (int m = 0; m < MI; ++m) _Pragma("unroll") for (int j = 0; j < NJ; ++j)          \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                       \
        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = j * 32 + (lane & 31);     \
        BODY;                                                                                              \
    }
    if (wave >= 4) { SK_FOR_ACC(mine[row * NTP + col] = acc[m][j][r]) }
    __syncthreads();
    if (wave < 4) { SK_FOR_ACC(mine[row * NTP + col] += acc[m][j][r]) }     // same lane, same addresses: no hazard within the wave
    __syncthreads();
#undef SK_FOR_ACC

    // ------------------------------ epilogue: one 16-byte output chunk of one pixel per thread ------------------------------
    constexpr int CH = 16 / (int)sizeof(TOUT);
    constexpr int CPR = NT / CH;                                 // chunks per pixel row of the tile: 4, 8 or 16
    constexpr int ITEMS = MT * CPR;
    static_assert(512 % CPR == 0 && (CPR == 4 || CPR == 8 || CPR == 16), "epilogue chunking");
    TOUT* out = (TOUT*)p.out;
    const TOUT* res = (const TOUT*)p.res;
    const int ch = tid % CPR;
    const int co0 = n0 + ch * CH;
    float addv[CH], st_s[CH], st_q[CH];
    const int nvalid = p.Cout - co0;                             // channels of this chunk that exist (Cout < NT: the 4- / 8-channel
                                                                 // pyramid convolutions, whose weight rows are zero-padded to 32)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float add = 0.f;
        if (c < nvalid) {
            if (p.bias) add += p.bias[co0 + c];
            if (p.temb) add += p.temb[(size_t)b * p.temb_bstride + co0 + c];
        }
        addv[c] = add; st_s[c] = 0.f; st_q[c] = 0.f;
    }
#pragma unroll
    for (int it0 = 0; it0 < ITEMS; it0 += 512) {
        const int it = it0 + tid;
        const int row = it / CPR;
        if (it < ITEMS && row < TH * TW && nvalid > 0) {
            const int ty = row / TW, gy = ty0 + ty, gx = tx0 + row - ty * TW;
            if (gy < p.H && gx < p.W) {
                float v[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] = addv[c];
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int c4 = 0; c4 < CH / 4; ++c4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(red + w * (MT * NTP) + row * NTP + ch * CH + c4 * 4);
                        v[c4 * 4] += t4.x; v[c4 * 4 + 1] += t4.y; v[c4 * 4 + 2] += t4.z; v[c4 * 4 + 3] += t4.w;
                    }
                const size_t pix = (size_t)(b * p.H + gy) * p.W + gx;
                if (res) {
                    float rv[CH];
                    if (nvalid >= CH) Vec16<TOUT>::load(res + pix * p.Cout + co0, rv);
                    else {
#pragma unroll
                        for (int c = 0; c < CH; ++c) rv[c] = c < nvalid ? to_f(res[pix * p.Cout + co0 + c]) : 0.f;
                    }
#pragma unroll
                    for (int c = 0; c < CH; ++c) v[c] += rv[c];
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) v[c] *= p.out_scale;
                if (p.pyr && nvalid >= CH) {
                    const float4 pq = *reinterpret_cast<const float4*>(p.pyr + pix * 4);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const float4 wq = *reinterpret_cast<const float4*>(p.w4 + (size_t)(co0 + c) * 4);
                        v[c] += p.b4[co0 + c] + wq.x * pq.x + wq.y * pq.y + wq.z * pq.z + wq.w * pq.w;
                    }
                }
                const uint4 packed = Vec16<TOUT>::pack(v);
                if (nvalid >= CH) *reinterpret_cast<uint4*>(out + pix * p.Cout + co0) = packed;
                else {
#pragma unroll
                    for (int c = 0; c < CH; ++c) if (c < nvalid) out[pix * p.Cout + co0 + c] = reinterpret_cast<const TOUT*>(&packed)[c];
                }
                float vr[CH];
                Vec16<TOUT>::load(reinterpret_cast<const TOUT*>(&packed), vr);   // statistics of the stored values
#pragma unroll
                for (int c = 0; c < CH; ++c) { st_s[c] += vr[c]; st_q[c] += vr[c] * vr[c]; }
            }
        }
    }
    if (p.stats) {                                               // per-wave totals -> LDS -> one pair of integer atomics per channel
        constexpr int NWV = ITEMS / 64 < 8 ? ITEMS / 64 : 8;     // waves that own output chunks
#pragma unroll
        for (int c = 0; c < CH; ++c) { st_s[c] = reduce_lanes_stride<CPR>(st_s[c]); st_q[c] = reduce_lanes_stride<CPR>(st_q[c]); }
        if (lane < CPR && wave < NWV) {
#pragma unroll
            for (int c = 0; c < CH; ++c) st_red[wave][ch * CH + c] = make_float2(st_s[c], st_q[c]);
        }
        __syncthreads();
        if (tid < NT) {
            float ss = 0.f, qq = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) { const float2 t = st_red[w][tid]; ss += t.x; qq += t.y; }
            if (n0 + tid < p.Cout) gn_accumulate(p.stats + ((size_t)b * p.Cout + n0 + tid) * 2, ss, qq);
        }
    }
}

long g_sk_max_px = 16L * 20L;     // largest map (pixels) served by conv_sk (32x40: conv_v2 is faster, 30 vs 49 us)

// tile shape: TH x TW <= 64 pixels, halo <= 128 pixels, fewest tiles, then the smallest halo; rows balanced over the tiles
void sk_tile(int H, int W, int pad, int* th, int* tw) {
    long best = -1; int bh = 1, bw = 1;
    for (int w = 1; w <= W && w <= SK_MT; ++w) {
        int hmax = SK_MT / w; if (hmax > H) hmax = H;
        const int ny = (H + hmax - 1) / hmax;
        const int h = (H + ny - 1) / ny;
        const int halo = (h + 2 * pad) * (w + 2 * pad);
        if (halo > SK_HPMAX) continue;
        const long tiles = (long)ny * ((W + w - 1) / w);
        const long key = tiles * 1024 + halo;
        if (best < 0 || key < best) { best = key; bh = h; bw = w; }
    }
    *th = bh; *tw = bw;
}

template <typename T, typename TO, int NJ, int CP>
void sk_launch(const ConvArgs& a, int th, int tw, hipStream_t s) {
    const int tiles = ((a.H + th - 1) / th) * ((a.W + tw - 1) / tw);
    hipLaunchKernelGGL((conv_sk_kernel<T, TO, NJ, CP>), dim3(tiles, (a.Cout + 32 * NJ - 1) / (32 * NJ), a.B), dim3(512), 0, s, a, th, tw);
}

}  // namespace

void conv_sk_set_max_px(long n) { g_sk_max_px = n; }

bool conv_sk_eligible(const ConvArgs& a) {
    const int Ctot = a.C0 + a.C1, XC = a.XC0 + a.XC1;
    // outputs: full 32-channel tiles in the input's type, or the 4- / 8-channel fp32 pyramid heads (weight rows padded to 32)
    const bool full = a.in_dtype == a.out_dtype && a.Cout % 32 == 0 && a.Cout >= 32;
    const bool head = a.out_dtype == DT_F32 && a.Cout <= 8 && a.cout_pad >= 32 && !a.pyr && !a.stats;
    // fp32 (training / parity mode): 32-channel K chunks make conv_v2's serial K loop long, split-K pays up to 64x64 maps (105 vs 114 ms per training step)
    const long max_px = (a.in_dtype == DT_F32 && g_sk_max_px > 0) ? std::max(g_sk_max_px, 4096L) : g_sk_max_px;
    return (long)a.H * a.W <= max_px && (a.ntaps == 9 || a.ntaps == 1) && a.w && (full || head) &&
           Ctot % 32 == 0 && a.C0 % 32 == 0 && XC % 32 == 0 && a.XC0 % 32 == 0 && (XC == 0 || a.w2) && Ctot >= 32 && Ctot <= 1024;
}

void launch_conv_sk(const ConvArgs& a, hipStream_t s) {
    int th, tw;
    sk_tile(a.H, a.W, a.ntaps == 9 ? 1 : 0, &th, &tw);
    const long tiles = (long)((a.H + th - 1) / th) * ((a.W + tw - 1) / tw) * a.B;
    // 64-channel tiles once 32-channel ones would give more than two workgroups per CU (the halo staging is repeated per channel tile)
    const bool wide = a.Cout % 64 == 0 && tiles * (a.Cout / 32) > 512;
    if (a.in_dtype != a.out_dtype) {                             // pyramid heads of the 16-bit modes: fp32 out
        if (a.in_dtype == DT_BF16) sk_launch<__bf16, float, 1, 256>(a, th, tw, s); else sk_launch<_Float16, float, 1, 256>(a, th, tw, s);
        return;
    }
    if (a.in_dtype == DT_BF16)     { wide ? sk_launch<__bf16, __bf16, 2, 128>(a, th, tw, s) : sk_launch<__bf16, __bf16, 1, 256>(a, th, tw, s); }
    else if (a.in_dtype == DT_F16) { wide ? sk_launch<_Float16, _Float16, 2, 128>(a, th, tw, s) : sk_launch<_Float16, _Float16, 1, 256>(a, th, tw, s); }
    else                           { wide ? sk_launch<float, float, 2, 64>(a, th, tw, s) : sk_launch<float, float, 1, 128>(a, th, tw, s); }
}

}  // namespace use
