// Backward kernels of one ResnetBlockBigGANpp (the gradient half of ScoreModel.train_step, reference
// model_wrapper.py:147-208 driven by SGMSE_module.py:46-54) - the minimum slice: everything one res-block needs besides the data
// gradient of its convolutions, which is the forward implicit-GEMM kernel itself run on the flipped, transposed weights.
//
//   wgrad_kernel          dW[co][ci][tap] = alpha * sum_{b,p} dY[b,p,co] X[b,p+tap,ci]     (and db[co] = alpha * sum dY)
//   gn_stats_kernel       mean / rstd per (item, group) of an NHWC map                       (GroupNorm(eps) statistics)
//   gn_act_bwd_reduce     s1[b][c] = sum_p du, s2[b][c] = sum_p du xhat,  du = dy * act'(gamma xhat + beta)
//   gn_act_bwd_apply      dx = rstd (gamma du - m1 - xhat m2) [+ add_scale * add],  m1 / m2 = group means of gamma s1 / gamma s2
//   gn_act_fwd            y = act(gamma xhat + beta)                                        (operand of the next wgrad; recomputed)
//   gn_param_grads        dgamma[c] = sum_b s2, dbeta[c] = sum_b s1
//   colsum_kernel         out[b][c] = sum_p x[b,p,c]                                          (Dense_0's upstream gradient)
//   dense_bwd_kernel      Dense_0(act(temb)): dW, db, dtemb
//   attn_bwd_rows / _cols the attention core: dq, dk, dv from dO (P recomputed)
//
// fp32 storage, NHWC ([B][H][W][C]); the contraction of wgrad runs on the matrix pipe in exact fp32 (v_mfma_f32_32x32x2_f32: K = pixels,
// 2 per instruction, one float per lane and operand - both operands are read coalesced over channels straight from NHWC).
#include "use_kernels.h"
#include "use_device.h"

namespace use {

// ---- weight gradient ---------------------------------------------------------------------------------------------------------------
// grid (co blocks of 32, ci blocks of 32, taps x pixel slices); 256 threads = 4 waves, each wave walks every 4th pixel pair of its slice;
// the four partial 32x32 tiles are summed through LDS and added to dW with atomics (pixel slices > 1) or stored.
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                                    float* __restrict__ db, int B, int H, int W, int Cout, int Cin, int ntaps, int nslices,
                                                    float alpha) {
    __shared__ float red[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hh = lane >> 5;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    const int tap = blockIdx.z % ntaps, slice = blockIdx.z / ntaps;
    const int dyy = ntaps == 9 ? tap / 3 - 1 : 0, dxx = ntaps == 9 ? tap % 3 - 1 : 0;
    const long npix = (long)B * H * W;
    const long per = (npix + nslices - 1) / nslices, p_lo = (long)slice * per, p_hi = p_lo + per < npix ? p_lo + per : npix;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    const bool co_ok = co0 + l31 < Cout, ci_ok = ci0 + l31 < Cin;
    for (long p0 = p_lo + wave * 2; p0 < p_hi; p0 += 8) {
        const long p = p0 + hh;                               // this lane's pixel of the pair (k index of the MFMA)
        float a = 0.f, bv = 0.f;
        if (p < p_hi) {
            if (co_ok) a = dy[p * Cout + co0 + l31];
            const int xw = (int)(p % W), yh = (int)((p / W) % H);
            const int ys = yh + dyy, xs = xw + dxx;
            if (ci_ok && ys >= 0 && ys < H && xs >= 0 && xs < W) bv = x[(p + (long)dyy * W + dxx) * Cin + ci0 + l31];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);     // D[co][ci] += sum_k dY[k][co] X[k][ci]
        bsum += a;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * hh][l31] = acc[r];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
        const int m = e >> 5, n = e & 31;
        const float v = (red[0][m][n] + red[1][m][n]) + (red[2][m][n] + red[3][m][n]);
        if (co0 + m < Cout && ci0 + n < Cin) {
            float* d = dw + ((size_t)(co0 + m) * Cin + ci0 + n) * ntaps + tap;
            if (nslices > 1) atomicAdd(d, v * alpha); else *d = v * alpha;
        }
    }
    if (db && blockIdx.y == 0 && tap == ntaps / 2) {        // bias gradient: column sums of dY (once per co block and slice)
        bsum += __shfl_xor(bsum, 32);
        __syncthreads();
        if (lane < 32) red[wave][0][lane] = bsum;
        __syncthreads();
        if (threadIdx.x < 32 && co0 + threadIdx.x < Cout) {
            const float v = (red[0][0][threadIdx.x] + red[1][0][threadIdx.x]) + (red[2][0][threadIdx.x] + red[3][0][threadIdx.x]);
            if (nslices > 1) atomicAdd(db + co0 + threadIdx.x, v * alpha); else db[co0 + threadIdx.x] = v * alpha;
        }
    }
}

// ---- weight gradient, tiled form ---------------------------------------------------------------------------------------------------
// One workgroup = a 64 co x 64 ci tile of dW for ALL taps (4 waves as 2 x 2, each 32 x 32 x taps accumulators: 144 registers for a 3x3)
// over one slice of the pixels.  Pixels are walked in chunks of RH rows x CW columns (8 x 8 where the map allows, RH | H so that a chunk stays
// inside one image): the chunk of dY ([64 px][64 co]) and the (RH + 2) x (CW + 2) halo patch of X ([px][64 ci], zero outside the image) are staged in
// LDS once and feed 9 MFMAs (one per tap: the same dY operand against the patch shifted by the tap) per pixel pair -
// v_mfma_f32_32x32x2_f32, exact fp32.  Slices write their tile to part[slice][co][ci][tap] (the reference's weight layout);
// wgrad_reduce_kernel sums the slices (deterministic: no atomics).  Two workgroups per CU overlap each other's staging.
constexpr int WG_PX = 64, WG_T = 64;                           // pixels per chunk, tile edge
constexpr int WH_PATCH = 136;                                  // largest halo patch: (RH, CW) = (32, 2); 8 x 8 chunks need 100
constexpr int WG_PATCH = 112;                                  // fp32-MFMA kernel: (RH, CW) = (16, 4) -> 108; 7 float4 of prefetch per thread
constexpr int WG_SMEM = (WG_PX * WG_T + WG_PATCH * WG_T) * 4 + WG_PX * 4;

struct WgradPlan { int RH, CW, chunks_x, units, per_slice, nslices, tiles; };
// workgroup -> (tile of dW, pixel slice).  Workgroups go to the 8 XCDs round-robin by their linear id, and every XCD has its own L2: with the
// slices a multiple of 8 the tiles of one slice (which read the same dY / X pixels: a 128 x 128 gradient = 4 tiles reads every operand
// twice) are given ids 8 apart, i.e. the same XCD at nearly the same time, so that the second reader hits that L2 instead of HBM.
DEVI void wgrad_block(const WgradPlan& q, int& tile, int& slice) {
    const int L = blockIdx.x;
    if ((q.nslices & 7) == 0) { const int r = L >> 3; tile = r % q.tiles; slice = (L & 7) + 8 * (r / q.tiles); }
    else { tile = L % q.tiles; slice = L / q.tiles; }
}

template <typename T> DEVI void wg_ld4(const T* p, float (&v)[4]);                       // 4 consecutive channels -> fp32
template <> DEVI void wg_ld4<float>(const float* p, float (&v)[4]) { const float4 u = *reinterpret_cast<const float4*>(p); v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; }
template <> DEVI void wg_ld4<__bf16>(const __bf16* p, float (&v)[4]) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
}
template <> DEVI void wg_ld4<_Float16>(const _Float16* p, float (&v)[4]) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h4 u = *reinterpret_cast<const h4*>(p);
    v[0] = (float)u[0]; v[1] = (float)u[1]; v[2] = (float)u[2]; v[3] = (float)u[3];
}

template <int NT, typename T>
__global__ __launch_bounds__(256, 2) void wgrad_tile_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part,
                                                            float* __restrict__ bpart, int B, int H, int W, int Cout, int Cin, WgradPlan q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* dyS = reinterpret_cast<float*>(smem);                        // [64 px][64 co]
    float* xS = dyS + WG_PX * WG_T;                                     // [(RH + 2)(CW + 2) px][64 ci]
    int* pofs = reinterpret_cast<int*>(xS + WG_PATCH * WG_T);           // patch pixel of chunk pixel j (centre tap), -1: not a pixel
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const int tiles_ci = (Cin + WG_T - 1) / WG_T;
    int tile_i, slice_i;
    wgrad_block(q, tile_i, slice_i);
    const int co0 = (tile_i / tiles_ci) * WG_T, ci0 = (tile_i % tiles_ci) * WG_T;
    const int wm = wave >> 1, wn = wave & 1;                            // this wave's 32 x 32 quadrant of the tile
    const int PWp = q.CW + 2, npatch = (q.RH + 2) * PWp, nchunk = q.RH * q.CW;
    if (t < WG_PX) pofs[t] = t < nchunk ? (t / q.CW + 1) * PWp + t % q.CW + 1 : -1;
    f32x16 acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float bsum = 0.f;
    const int u_lo = slice_i * q.per_slice, u_hi = min(u_lo + q.per_slice, q.units);
    const int rows_per_img = H / q.RH;
    // Register double buffering (as wgrad16_kernel): the float4 pieces of chunk u + 1 (4 of dY, up to 7 of the patch per thread) are
    // requested before the 288 MFMAs of chunk u and stored to LDS after them.  The synchronous form ran at 0.56 of the fp32 MFMA rate
    // with the clock at 2.39 GHz and 1.06 kW: its load -> barrier -> MFMA cycle was the limit, not power or memory.
    constexpr int NDP = WG_PX * (WG_T / 4) / 256;                                            // 4
    constexpr int NXP = NT == 9 ? (WG_PATCH * (WG_T / 4) + 255) / 256 : NDP;                  // 7 | 4
    float4 rdy[NDP], rx[NXP];
    auto fetch = [&](int u) __attribute__((always_inline)) {
        const int cx = u % q.chunks_x, rr = u / q.chunks_x;             // column chunk, row chunk (over all images)
        const int b = rr / rows_per_img, y0 = (rr % rows_per_img) * q.RH, x0 = cx * q.CW;
#pragma unroll
        for (int it = 0; it < NDP; ++it) {                              // dY chunk
            const int e = t + 256 * it, j = e / (WG_T / 4), c4 = (e % (WG_T / 4)) * 4;
            const int r = j / q.CW, cxx = x0 + j % q.CW;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (j < nchunk && cxx < W && co0 + c4 < Cout) wg_ld4<T>(dy + (((size_t)b * H + y0 + r) * W + cxx) * Cout + co0 + c4, v);
            rdy[it] = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int it = 0; it < NXP; ++it) {
            const int e = t + 256 * it, c4 = (e % (WG_T / 4)) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (NT == 9) {                                              // X patch with a one-pixel halo, zero outside the image
                const int pp = e / (WG_T / 4);
                const int yy = y0 + pp / PWp - 1, xx = x0 + pp % PWp - 1;
                if (pp < npatch && yy >= 0 && yy < H && xx >= 0 && xx < W && ci0 + c4 < Cin) wg_ld4<T>(x + (((size_t)b * H + yy) * W + xx) * Cin + ci0 + c4, v);
            } else {                                                    // 1x1: the chunk's own pixels
                const int j = e / (WG_T / 4), r = j / q.CW, cxx = x0 + j % q.CW;
                if (j < nchunk && cxx < W && ci0 + c4 < Cin) wg_ld4<T>(x + (((size_t)b * H + y0 + r) * W + cxx) * Cin + ci0 + c4, v);
            }
            rx[it] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    auto stash = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NDP; ++it) {
            const int e = t + 256 * it, j = e / (WG_T / 4), c4 = (e % (WG_T / 4)) * 4;
            *reinterpret_cast<float4*>(dyS + j * WG_T + c4) = rdy[it];
        }
#pragma unroll
        for (int it = 0; it < NXP; ++it) {
            const int e = t + 256 * it, c4 = (e % (WG_T / 4)) * 4;
            if (NT == 9) {
                const int pp = e / (WG_T / 4);
                if (pp < npatch) *reinterpret_cast<float4*>(xS + pp * WG_T + c4) = rx[it];
            } else {                                                    // at their patch positions (pixels beyond the chunk: nowhere)
                const int j = e / (WG_T / 4);
                if (j < nchunk) *reinterpret_cast<float4*>(xS + ((j / q.CW + 1) * PWp + j % q.CW + 1) * WG_T + c4) = rx[it];
            }
        }
    };
    if (u_lo < u_hi) fetch(u_lo);
    for (int u = u_lo; u < u_hi; ++u) {
        __syncthreads();                                                // the previous chunk's MFMAs are done with the LDS
        stash();
        __syncthreads();
        if (u + 1 < u_hi) fetch(u + 1);
        const float* aP = dyS + wm * 32 + l31;
        const float* bP = xS + wn * 32 + l31;
#pragma unroll 4
        for (int j0 = 0; j0 < WG_PX; j0 += 2) {
            const int j = j0 + hh;                                      // this lane's pixel of the pair (the k index of the MFMA)
            const int po = pofs[j];
            float a = aP[j * WG_T];
            if (po < 0) a = 0.f;                                        // beyond the chunk (RH CW < 64): contributes nothing
            const int pb = po < 0 ? PWp + 1 : po;
            bsum += a;
            if (NT == 9) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float bv = bP[(pb + (k / 3 - 1) * PWp + (k % 3 - 1)) * WG_T];
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[k], 0, 0, 0);      // D[co][ci] += sum_k dY[k][co] X[k + tap][ci]
                }
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bP[pb * WG_T], acc[0], 0, 0, 0);
            }
        }
    }
    float* o = part + (size_t)slice_i * Cout * Cin * NT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, ci = ci0 + wn * 32 + l31;
        if (co < Cout && ci < Cin) {
#pragma unroll
            for (int k = 0; k < NT; ++k) o[((size_t)co * Cin + ci) * NT + k] = acc[k][r];
        }
    }
    if (bpart && ci0 == 0 && wn == 0) {                                // bias gradient: column sums of dY, once per co tile and slice
        bsum += __shfl_xor(bsum, 32);
        if (hh == 0 && co0 + wm * 32 + l31 < Cout) bpart[(size_t)slice_i * Cout + co0 + wm * 32 + l31] = bsum;
    }
}
// ---- weight gradient of 16-bit tensors on the 16-bit matrix pipe (mixed-precision training) -------------------------------------------
// Same decomposition as wgrad_tile_kernel (64 co x 64 ci tile, all taps, pixel slices, fp32 partial tiles), but the contraction runs on
// v_mfma_f32_32x32x16_{bf16,f16}: 16 pixels per instruction at 16x the fp32 rate.  The MFMA wants 8 consecutive k (= pixels) per lane for
// one channel, the NHWC data has the channels contiguous: the chunk ([64 px][64 co]) and the halo patch ([px][64 ci]) are staged in LDS
// as they lie in memory (plain 16-byte copies, row pitch 96 elements = 192 B so that the four rows of a transposing read fall into four
// different 64-byte bank groups) and read with ds_read_b64_tr_b16: a 16-lane group fetches a [4 px][16 ch] block and every lane receives
// 4 pixels of its own channel (lane i passes the address of row i >> 2, channel quad i & 3; measured: scripts/experiments/tr_probe.hip).
// Chunks are RH x CW = 8 x 8 pixels where the map allows (halo patch 100 px instead of 198 for 1 x 64); the tap shift is a row offset
// in the patch, i.e. an address offset - no alignment constraint.  Products of 16-bit values are exact in fp32: same numerics as the
// fp32-MFMA kernel on the same data, up to the summation order.
constexpr int WH_PITCH = 96;                                                     // LDS row pitch in elements (192 B)
constexpr int WH_SMEM = (WG_PX + WH_PATCH) * WH_PITCH * 2 + WG_PX * 4;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <typename T>
DEVI typename Mfma<T>::frag ld_tr8(const char* p0, const char* p1) {            // 8 k-values of this lane's channel: two transposing reads
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    const s16x8 c = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(typename Mfma<T>::frag, c);
}

template <int NT, typename T>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part,
                                                         float* __restrict__ bpart, int B, int H, int W, int Cout, int Cin, WgradPlan q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dyS = smem;                                                   // [64 px][pitch] T
    char* xS = dyS + WG_PX * WH_PITCH * 2;                              // [(RH + 2)(CW + 2) px][pitch] T
    int* pofs = reinterpret_cast<int*>(xS + WH_PATCH * WH_PITCH * 2);   // patch pixel of chunk pixel j (centre tap); beyond the chunk: pixel 0's
                                                                        // (always staged: dY is zero there, but 0 x stale LDS bits could be NaN)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const int tiles_ci = (Cin + WG_T - 1) / WG_T;
    int tile_i, slice_i;
    wgrad_block(q, tile_i, slice_i);
    const int co0 = (tile_i / tiles_ci) * WG_T, ci0 = (tile_i % tiles_ci) * WG_T;
    const int wm = wave >> 1, wn = wave & 1;
    const int PWp = q.CW + 2, npatch = (q.RH + 2) * PWp, nchunk = q.RH * q.CW;
    if (t < WG_PX) pofs[t] = t < nchunk ? (t / q.CW + 1) * PWp + t % q.CW + 1 : PWp + 1;
    __syncthreads();
    // this lane's rows of the transposing reads: k = 16 ks + 8 (lane >> 5) + 4 r + (i >> 2), i = lane & 15; channels 16 ((lane >> 4) & 1) + 4 (i & 3)
    const int li = lane & 15, chq = (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
    int aoff[4][2], boff[4][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = 16 * ks + 8 * hh + 4 * r + (li >> 2);
            aoff[ks][r] = j * WH_PITCH * 2 + wm * 64 + chq;
            boff[ks][r] = pofs[j] * WH_PITCH * 2 + wn * 64 + chq;
        }
    f32x16 acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float bsum = 0.f;
    const bool want_b = bpart && ci0 == 0 && wn == 0;
    const int u_lo = slice_i * q.per_slice, u_hi = min(u_lo + q.per_slice, q.units);
    const int rows_per_img = H / q.RH;
    // Register double buffering: the 16-byte pieces of chunk u + 1 (2 of dY, up to 5 of the patch per thread) are requested before the
    // MFMAs of chunk u and stored to LDS after them, so their latency hides behind the 36 MFMAs (PMC of the synchronous form: waves
    // waiting 0.61 of the time).
    constexpr int NXP = NT == 9 ? (WH_PATCH * (WG_T / 8) + 255) / 256 : (WG_PX * (WG_T / 8)) / 256;
    uint4 rdy[2], rx[NXP];
    auto fetch = [&](int u) {
        const int cx = u % q.chunks_x, rr = u / q.chunks_x;
        const int b = rr / rows_per_img, y0 = (rr % rows_per_img) * q.RH, x0 = cx * q.CW;
#pragma unroll
        for (int it = 0; it < 2; ++it) {                                // dY chunk
            const int e = t + 256 * it, j = e / (WG_T / 8), c8 = (e % (WG_T / 8)) * 8;
            const int r = j / q.CW, cxx = x0 + j % q.CW;
            rdy[it] = make_uint4(0u, 0u, 0u, 0u);
            if (j < nchunk && cxx < W && co0 + c8 < Cout) rdy[it] = *reinterpret_cast<const uint4*>(dy + (((size_t)b * H + y0 + r) * W + cxx) * Cout + co0 + c8);
        }
#pragma unroll
        for (int it = 0; it < NXP; ++it) {
            const int e = t + 256 * it, c8 = (e % (WG_T / 8)) * 8;
            rx[it] = make_uint4(0u, 0u, 0u, 0u);
            if (NT == 9) {                                              // X patch with a one-pixel halo, zero outside the image
                const int pp = e / (WG_T / 8);
                const int yy = y0 + pp / PWp - 1, xx = x0 + pp % PWp - 1;
                if (pp < npatch && yy >= 0 && yy < H && xx >= 0 && xx < W && ci0 + c8 < Cin)
                    rx[it] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + yy) * W + xx) * Cin + ci0 + c8);
            } else {                                                    // 1x1: the chunk's own pixels
                const int j = e / (WG_T / 8), r = j / q.CW, cxx = x0 + j % q.CW;
                if (j < nchunk && cxx < W && ci0 + c8 < Cin) rx[it] = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + y0 + r) * W + cxx) * Cin + ci0 + c8);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = t + 256 * it, j = e / (WG_T / 8), c8 = (e % (WG_T / 8)) * 8;
            *reinterpret_cast<uint4*>(dyS + (j * WH_PITCH + c8) * 2) = rdy[it];
        }
#pragma unroll
        for (int it = 0; it < NXP; ++it) {
            const int e = t + 256 * it, c8 = (e % (WG_T / 8)) * 8;
            if (NT == 9) {
                const int pp = e / (WG_T / 8);
                if (pp < npatch) *reinterpret_cast<uint4*>(xS + (pp * WH_PITCH + c8) * 2) = rx[it];
            } else {                                                    // at their patch positions (pixels beyond the chunk: nowhere)
                const int j = e / (WG_T / 8);
                if (j < nchunk) *reinterpret_cast<uint4*>(xS + (((j / q.CW + 1) * PWp + j % q.CW + 1) * WH_PITCH + c8) * 2) = rx[it];
            }
        }
    };
    if (u_lo < u_hi) fetch(u_lo);
    for (int u = u_lo; u < u_hi; ++u) {
        __syncthreads();                                                // the previous chunk's MFMAs are done with the LDS
        stash();
        __syncthreads();
        if (u + 1 < u_hi) fetch(u + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const typename Mfma<T>::frag a = ld_tr8<T>(dyS + aoff[ks][0], dyS + aoff[ks][1]);
            if (want_b) {
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum += (float)a[k];
            }
            if (NT == 9) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const int sh = ((k / 3 - 1) * PWp + (k % 3 - 1)) * WH_PITCH * 2;
                    const typename Mfma<T>::frag bv = ld_tr8<T>(xS + boff[ks][0] + sh, xS + boff[ks][1] + sh);
                    acc[k] = Mfma<T>::mma(a, bv, acc[k]);                // D[co][ci] += sum_k dY[k][co] X[k + tap][ci]
                }
            } else {
                acc[0] = Mfma<T>::mma(a, ld_tr8<T>(xS + boff[ks][0], xS + boff[ks][1]), acc[0]);
            }
        }
    }
    float* o = part + (size_t)slice_i * Cout * Cin * NT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, ci = ci0 + wn * 32 + l31;
        if (co < Cout && ci < Cin) {
#pragma unroll
            for (int k = 0; k < NT; ++k) o[((size_t)co * Cin + ci) * NT + k] = acc[k][r];
        }
    }
    if (want_b) {                                                      // bias gradient: column sums of dY, once per co tile and slice
        bsum += __shfl_xor(bsum, 32);
        if (hh == 0 && co0 + wm * 32 + l31 < Cout) bpart[(size_t)slice_i * Cout + co0 + wm * 32 + l31] = bsum;
    }
}

// out[i] = alpha * sum over the slices of part[slice][i]; the bias partials (nb entries per slice) ride along as elements n .. n + nb
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nslices, long n, float alpha, float* __restrict__ out,
                                                           const float* __restrict__ bpart, int nb, float* __restrict__ bout) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n + nb; i += (long)gridDim.x * 256) {
        float a = 0.f;
        if (i < n) {
            for (int sl = 0; sl < nslices; ++sl) a += part[(size_t)sl * n + i];
            out[i] = a * alpha;
        } else {
            const long c = i - n;
            for (int sl = 0; sl < nslices; ++sl) a += bpart[(size_t)sl * nb + c];
            bout[c] = a * alpha;
        }
    }
}

// ---- GroupNorm statistics: one block per (item, group) -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int HW, int C, int G, float eps, float* __restrict__ mean,
                                                       float* __restrict__ rstd) {
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G;
    double s = 0.0, q = 0.0;
    for (long i = threadIdx.x; i < (long)HW * cpg; i += 256) {
        const float v = x[((size_t)b * HW + i / cpg) * C + g * cpg + i % cpg];
        s += v; q += (double)v * v;
    }
    __shared__ double rs[4], rq[4];
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)HW * cpg, m = (rs[0] + rs[1] + rs[2] + rs[3]) / n;
        double var = (rq[0] + rq[1] + rq[2] + rq[3]) / n - m * m;
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)m; rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// d/du [u sigmoid(u)].  FAST (16-bit storage): v_exp_f32 / v_rcp_f32 approximations, far below the 16-bit rounding of the result
template <bool ACT, bool FAST = false>
__device__ __forceinline__ float act_grad(float u) {
    if (!ACT) return 1.f;
    const float sg = FAST ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.44269504088896341f)) : 1.0f / (1.0f + expf(-u));
    return sg * (1.0f + u * (1.0f - sg));
}

// s1[b][c] = sum_p du, s2[b][c] = sum_p du xhat; grid (channel blocks of 64, B), 256 threads = 4 pixel phases x 64 channels
template <bool ACT>
__global__ __launch_bounds__(256) void gn_act_bwd_reduce(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int HW, int C, int G, float* __restrict__ s1,
                                                         float* __restrict__ s2) {
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
        const int g = c / (C / G);
        const float mu = mean[b * G + g], rs = rstd[b * G + g], gm = gamma[c], bt = beta[c];
        for (int p = ph; p < HW; p += 4) {
            const size_t i = ((size_t)b * HW + p) * C + c;
            const float xh = (x[i] - mu) * rs;
            const float du = dy[i] * act_grad<ACT>(fmaf(gm, xh, bt));
            a1 += du; a2 += (double)du * xh;
        }
    }
    __shared__ double r1[4][64], r2[4][64];
    r1[ph][threadIdx.x & 63] = a1; r2[ph][threadIdx.x & 63] = a2;
    __syncthreads();
    if (ph == 0 && c < C) {
        const int l = threadIdx.x;
        s1[(size_t)b * C + c] = (float)((r1[0][l] + r1[1][l]) + (r1[2][l] + r1[3][l]));
        s2[(size_t)b * C + c] = (float)((r2[0][l] + r2[1][l]) + (r2[2][l] + r2[3][l]));
    }
}

// dx = rstd (gamma du - m1 - xhat m2) + add_scale * add; m1, m2 = means over the group (channels x pixels) of gamma du, gamma du xhat
template <bool ACT>
__global__ __launch_bounds__(256) void gn_act_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ s1, const float* __restrict__ s2,
                                                        const float* __restrict__ add, float add_scale, int HW, int C, int G,
                                                        float* __restrict__ dx, long n) {
    const int cpg = C / G;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C), b = (int)(i / ((long)HW * C)), g = c / cpg;
        double m1 = 0.0, m2 = 0.0;
        for (int k = 0; k < cpg; ++k) {
            const int cc = g * cpg + k;
            m1 += (double)gamma[cc] * s1[(size_t)b * C + cc]; m2 += (double)gamma[cc] * s2[(size_t)b * C + cc];
        }
        const double inv = 1.0 / ((double)HW * cpg);
        const float mu = mean[b * G + g], rs = rstd[b * G + g];
        const float xh = (x[i] - mu) * rs;
        const float du = dy[i] * act_grad<ACT>(fmaf(gamma[c], xh, beta[c]));
        float v = rs * (gamma[c] * du - (float)(m1 * inv) - xh * (float)(m2 * inv));
        if (add) v += add_scale * add[i];
        dx[i] = v;
    }
}

// y = act(gamma xhat + beta): the activation a convolution's weight gradient is taken against (recomputed, not stored by the forward)
template <bool ACT>
__global__ __launch_bounds__(256) void gn_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int G,
                                                         float* __restrict__ y, long n) {
    const int cpg = C / G;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C), b = (int)(i / ((long)HW * C)), g = c / cpg;
        const float u = fmaf(gamma[c], (x[i] - mean[b * G + g]) * rstd[b * G + g], beta[c]);
        y[i] = ACT ? u / (1.0f + expf(-u)) : u;
    }
}

// ---- sliced forms (C % 4 == 0, C <= 1024): grid (pixel slices, B); a block covers `ppi` pixels per iteration with C/4 threads per
// pixel (16-byte loads along the channels), folds its pixel lanes through LDS and leaves one fp64 partial per (slice, channel | group);
// a one-block-per-item kernel sums the slices.  These replace the one-block-per-(item, group | 64 channels) kernels above, which leave
// the chip idle on the full-resolution maps (8 blocks for a 128-channel map of 4 items).
__host__ __device__ inline int gn_slices(int HW, int C, int vec) {
    const int ppi = 256 / (C / vec);
    int n = HW / (ppi * 8);
    return n < 1 ? 1 : n > GN_MAX_SLICES ? GN_MAX_SLICES : n;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_part_kernel(const T* __restrict__ x, int HW, int C, int G, double* __restrict__ part) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float sh[2][256 * VEC];
    __shared__ double ch[2][1024];
    const int tpp = C / VEC, ppi = 256 / tpp, t = threadIdx.x, b = blockIdx.y, ns = gridDim.x;
    const int per = (HW + ns - 1) / ns, lo = blockIdx.x * per, hi = lo + per < HW ? lo + per : HW;
    const bool on = t < tpp * ppi;
    const int pl = t / tpp, c4 = (t % tpp) * VEC;
    float s[VEC], q[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { s[k] = 0.f; q[k] = 0.f; }
    if (on)
        for (int p = lo + pl; p < hi; p += ppi) {
            float v[VEC];
            Vec16<T>::load(x + ((size_t)b * HW + p) * C + c4, v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) { s[k] += v[k]; q[k] = fmaf(v[k], v[k], q[k]); }
        }
    if (on) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) { sh[0][pl * C + c4 + k] = s[k]; sh[1][pl * C + c4 + k] = q[k]; }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        double a = 0.0, e = 0.0;
        for (int l = 0; l < ppi; ++l) { a += sh[0][l * C + c]; e += sh[1][l * C + c]; }
        ch[0][c] = a; ch[1][c] = e;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = t; g < G; g += 256) {
        double a = 0.0, e = 0.0;
        for (int k = 0; k < cpg; ++k) { a += ch[0][g * cpg + k]; e += ch[1][g * cpg + k]; }
        double* o = part + (((size_t)b * ns + blockIdx.x) * G + g) * 2;
        o[0] = a; o[1] = e;
    }
}
// per-(item, channel) sums over the pixels, sliced like gn_stats_part_kernel: part[b][slice][c] (fp64)
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ x, int HW, int C, double* __restrict__ part) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float sh[256 * VEC];
    const int tpp = C / VEC, ppi = 256 / tpp, t = threadIdx.x, b = blockIdx.y, ns = gridDim.x;
    const int per = (HW + ns - 1) / ns, lo = blockIdx.x * per, hi = lo + per < HW ? lo + per : HW;
    const bool on = t < tpp * ppi;
    const int pl = t / tpp, c4 = (t % tpp) * VEC;
    float s[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.f;
    if (on) {
        for (int p = lo + pl; p < hi; p += ppi) {
            float v[VEC];
            Vec16<T>::load(x + ((size_t)b * HW + p) * C + c4, v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) s[k] += v[k];
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) sh[pl * C + c4 + k] = s[k];
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        double a = 0.0;
        for (int l = 0; l < ppi; ++l) a += sh[l * C + c];
        part[((size_t)b * ns + blockIdx.x) * C + c] = a;
    }
}
__global__ __launch_bounds__(256) void colsum_fin_kernel(const double* __restrict__ part, int ns, int C, int BC, float scale, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, c = i % C;
    double a = 0.0;
    for (int sl = 0; sl < ns; ++sl) a += part[((size_t)b * ns + sl) * C + c];
    out[i] = (float)(a * scale);
}

// one wave per (item, group): the slices are summed across the lanes
__global__ __launch_bounds__(64) void gn_stats_fin_kernel(const double* __restrict__ part, int ns, int G, double n, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd) {
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    double a = 0.0, e = 0.0;
    for (int sl = threadIdx.x; sl < ns; sl += 64) { const double* o = part + (((size_t)b * ns + sl) * G + g) * 2; a += o[0]; e += o[1]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); e += __shfl_xor(e, o); }
    if (threadIdx.x == 0) {
        const double m = a / n;
        double var = e / n - m * m;
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)m; rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <typename T, bool ACT>
__global__ __launch_bounds__(256) void gn_act_bwd_part_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int HW, int C, int G, double* __restrict__ part) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float sh[2][256 * VEC];
    const int tpp = C / VEC, ppi = 256 / tpp, t = threadIdx.x, b = blockIdx.y, ns = gridDim.x, cpg = C / G;
    const int per = (HW + ns - 1) / ns, lo = blockIdx.x * per, hi = lo + per < HW ? lo + per : HW;
    const bool on = t < tpp * ppi;
    const int pl = t / tpp, c4 = (t % tpp) * VEC;
    if (on) {
        float a1[VEC], a2[VEC], mu[VEC], rs[VEC], gm[VEC], bt[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int g = (c4 + k) / cpg;
            a1[k] = 0.f; a2[k] = 0.f;
            mu[k] = mean[b * G + g]; rs[k] = rstd[b * G + g]; gm[k] = gamma[c4 + k]; bt[k] = beta[c4 + k];
        }
        for (int p = lo + pl; p < hi; p += ppi) {
            const size_t i = ((size_t)b * HW + p) * C + c4;
            float xs[VEC], ds[VEC];
            Vec16<T>::load(x + i, xs); Vec16<T>::load(dy + i, ds);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float xh = (xs[k] - mu[k]) * rs[k];
                const float du = ds[k] * act_grad<ACT, sizeof(T) == 2>(fmaf(gm[k], xh, bt[k]));
                a1[k] += du; a2[k] = fmaf(du, xh, a2[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) { sh[0][pl * C + c4 + k] = a1[k]; sh[1][pl * C + c4 + k] = a2[k]; }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        double a = 0.0, e = 0.0;
        for (int l = 0; l < ppi; ++l) { a += sh[0][l * C + c]; e += sh[1][l * C + c]; }
        double* o = part + (((size_t)b * ns + blockIdx.x) * C + c) * 2;
        o[0] = a; o[1] = e;
    }
}
// one wave per (item, group): s1 / s2 per channel (the slices summed across the lanes) and the group means m1 / m2 of gamma s1, gamma s2
__global__ __launch_bounds__(64) void gn_act_bwd_fin_kernel(const double* __restrict__ part, const float* __restrict__ gamma, int ns, int HW, int C,
                                                            int G, float* __restrict__ s1, float* __restrict__ s2, float* __restrict__ m1,
                                                            float* __restrict__ m2) {
    const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G;
    double ga = 0.0, ge = 0.0;
    for (int k = 0; k < cpg; ++k) {
        const int c = g * cpg + k;
        double a = 0.0, e = 0.0;
        for (int sl = threadIdx.x; sl < ns; sl += 64) { const double* o = part + (((size_t)b * ns + sl) * C + c) * 2; a += o[0]; e += o[1]; }
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); e += __shfl_xor(e, o); }
        if (threadIdx.x == 0) { s1[(size_t)b * C + c] = (float)a; s2[(size_t)b * C + c] = (float)e; }
        ga += a * gamma[c]; ge += e * gamma[c];
    }
    if (threadIdx.x == 0) {
        const double inv = 1.0 / ((double)HW * cpg);
        m1[blockIdx.x] = (float)(ga * inv); m2[blockIdx.x] = (float)(ge * inv);
    }
}
template <typename T, bool ACT>
__global__ __launch_bounds__(256) void gn_act_bwd_applyv_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ m1,
                                                                const float* __restrict__ m2, const T* __restrict__ add, float add_scale, int HW,
                                                                int C, int G, T* __restrict__ dx, long nv) {
    constexpr int VEC = Vec16<T>::N;
    const int cpg = C / G, cvn = C / VEC;
    for (long iv = (long)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += (long)gridDim.x * 256) {
        const int c4 = (int)(iv % cvn) * VEC, b = (int)(iv / ((long)HW * cvn));
        const size_t i = (size_t)iv * VEC;
        float xs[VEC], ds[VEC], as[VEC], o[VEC];
        Vec16<T>::load(x + i, xs); Vec16<T>::load(dy + i, ds);
        if (add) Vec16<T>::load(add + i, as);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int c = c4 + k, g = c / cpg;
            const float rs = rstd[b * G + g], xh = (xs[k] - mean[b * G + g]) * rs, gm = gamma[c];
            const float du = ds[k] * act_grad<ACT, sizeof(T) == 2>(fmaf(gm, xh, beta[c]));
            o[k] = rs * (gm * du - m1[b * G + g] - xh * m2[b * G + g]);
            if (add) o[k] = fmaf(add_scale, as[k], o[k]);
        }
        Vec16<T>::store(dx + i, o);
    }
}
template <typename T, bool ACT>
__global__ __launch_bounds__(256) void gn_act_fwdv_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, int HW, int C, int G,
                                                          T* __restrict__ y, long nv) {
    constexpr int VEC = Vec16<T>::N;
    const int cpg = C / G, cvn = C / VEC;
    for (long iv = (long)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += (long)gridDim.x * 256) {
        const int c4 = (int)(iv % cvn) * VEC, b = (int)(iv / ((long)HW * cvn));
        float xs[VEC], o[VEC];
        Vec16<T>::load(x + (size_t)iv * VEC, xs);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int c = c4 + k, g = c / cpg;
            const float u = fmaf(gamma[c], (xs[k] - mean[b * G + g]) * rstd[b * G + g], beta[c]);
            o[k] = ACT ? silu_f<sizeof(T) == 4>(u) : u;
        }
        Vec16<T>::store(y + (size_t)iv * VEC, o);
    }
}

__global__ void gn_param_grads_kernel(const float* __restrict__ s1, const float* __restrict__ s2, int B, int C, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, bsum = 0.0;
    for (int b = 0; b < B; ++b) { a += s2[(size_t)b * C + c]; bsum += s1[(size_t)b * C + c]; }
    dgamma[c] = (float)a; dbeta[c] = (float)bsum;
}

// out[b][c] = scale * sum_p x[b,p,c]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int HW, int C, float scale, float* __restrict__ out) {
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    double a = 0.0;
    if (c < C) for (int p = ph; p < HW; p += 4) a += x[((size_t)b * HW + p) * C + c];
    __shared__ double r[4][64];
    r[ph][threadIdx.x & 63] = a;
    __syncthreads();
    if (ph == 0 && c < C) { const int l = threadIdx.x; out[(size_t)b * C + c] = (float)(((r[0][l] + r[1][l]) + (r[2][l] + r[3][l])) * scale); }
}

// Dense_0(act(temb)) [B][K] -> [B][Cout]: g = upstream gradient [B][Cout]; dW[co][k] = sum_b g act(temb), db[co] = sum_b g,
// dtemb[b][k] = act'(temb[b][k]) sum_co g[b][co] W[co][k]     (one block; these are [2..64] x [512] x [<=512] problems)
__global__ __launch_bounds__(256) void dense_bwd_kernel(const float* __restrict__ g, const float* __restrict__ temb, const float* __restrict__ Wd,
                                                        int B, int K, int Cout, float* __restrict__ dW, float* __restrict__ db,
                                                        float* __restrict__ dtemb) {
    for (int e = threadIdx.x; e < Cout * K; e += 256) {
        const int co = e / K, k = e % K;
        double a = 0.0;
        for (int b = 0; b < B; ++b) { const float t = temb[b * K + k]; a += (double)g[b * Cout + co] * (t / (1.0f + expf(-t))); }
        dW[e] = (float)a;
    }
    for (int co = threadIdx.x; co < Cout; co += 256) {
        double a = 0.0;
        for (int b = 0; b < B; ++b) a += g[b * Cout + co];
        db[co] = (float)a;
    }
    for (int e = threadIdx.x; e < B * K; e += 256) {
        const int b = e / K, k = e % K;
        double a = 0.0;
        for (int co = 0; co < Cout; ++co) a += (double)g[b * Cout + co] * Wd[co * K + k];
        dtemb[e] = (float)a * act_grad<true>(temb[e]);
    }
}

// ---- attention core backward: P = softmax(q k^T / sqrt(C)) recomputed; dP = dO v^T; dS = P (dP - rowsum(P dP)) ---------------------------
// pass 1, one block per (query i, item): the row i of P and dS (to scratch [B][N][N]) and dq_i = sum_j dS_ij k_j / sqrt(C)
__global__ __launch_bounds__(128) void attn_bwd_rows_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                            const float* __restrict__ dO, float* __restrict__ P, float* __restrict__ dS,
                                                            float* __restrict__ dq, int N, int C) {
    extern __shared__ float sm[];                            // [N] scores -> P, [N] dP -> dS, [C] q_i, [C] dO_i
    float* sp = sm; float* sd = sm + N; float* qi = sm + 2 * N; float* doi = qi + C;
    const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t base = (size_t)b * N * C;
    const float scale = 1.0f / sqrtf((float)C);
    for (int c = tid; c < C; c += 128) { qi[c] = q[base + (size_t)i * C + c]; doi[c] = dO[base + (size_t)i * C + c]; }
    __syncthreads();
    for (int j = tid; j < N; j += 128) {
        float s = 0.f, d = 0.f;
        for (int c = 0; c < C; ++c) { s = fmaf(qi[c], k[base + (size_t)j * C + c], s); d = fmaf(doi[c], v[base + (size_t)j * C + c], d); }
        sp[j] = s * scale; sd[j] = d;
    }
    __syncthreads();
    __shared__ float red[128];
    float m = -INFINITY;
    for (int j = tid; j < N; j += 128) m = fmaxf(m, sp[j]);
    red[tid] = m; __syncthreads();
    for (int o = 64; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    m = red[0]; __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < N; j += 128) { const float e = expf(sp[j] - m); sp[j] = e; sum += e; }
    red[tid] = sum; __syncthreads();
    for (int o = 64; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.0f / red[0]; __syncthreads();
    float dot = 0.f;
    for (int j = tid; j < N; j += 128) { sp[j] *= inv; dot = fmaf(sp[j], sd[j], dot); }
    red[tid] = dot; __syncthreads();
    for (int o = 64; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    dot = red[0]; __syncthreads();
    for (int j = tid; j < N; j += 128) {
        sd[j] = sp[j] * (sd[j] - dot);
        P[((size_t)b * N + i) * N + j] = sp[j]; dS[((size_t)b * N + i) * N + j] = sd[j];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 128) {
        float a = 0.f;
        for (int j = 0; j < N; ++j) a = fmaf(sd[j], k[base + (size_t)j * C + c], a);
        dq[base + (size_t)i * C + c] = a * scale;
    }
}
// pass 2, one block per (key j, item): dk_j = sum_i dS_ij q_i / sqrt(C), dv_j = sum_i P_ij dO_i
__global__ __launch_bounds__(128) void attn_bwd_cols_kernel(const float* __restrict__ q, const float* __restrict__ dO, const float* __restrict__ P,
                                                            const float* __restrict__ dS, float* __restrict__ dk, float* __restrict__ dv, int N,
                                                            int C) {
    const int j = blockIdx.x, b = blockIdx.y;
    const size_t base = (size_t)b * N * C;
    const float scale = 1.0f / sqrtf((float)C);
    for (int c = threadIdx.x; c < C; c += 128) {
        float a = 0.f, e = 0.f;
        for (int i = 0; i < N; ++i) {
            a = fmaf(dS[((size_t)b * N + i) * N + j], q[base + (size_t)i * C + c], a);
            e = fmaf(P[((size_t)b * N + i) * N + j], dO[base + (size_t)i * C + c], e);
        }
        dk[base + (size_t)j * C + c] = a * scale; dv[base + (size_t)j * C + c] = e;
    }
}

// ---- device-side weight packing (training: the parameters live in HBM and change every step) ----------------------------------------
// Writes the two layouts the convolution kernels read (use_kernels.h: plain [cout][tap][cin] and, when ck > 0, slab-major
// [tap][cin/ck][cout_pad][ck] with the 16-byte pieces of a row swizzled by the row) from an fp32 parameter tensor in HBM.
// mode 0: conv weight [cout][cin][taps];  mode 1: the data-gradient operand of a conv weight W[cin_op... = cout_fwd][cout_op = cin_fwd][taps],
// i.e. w'[co][ci][tap] = W[ci][co][taps-1-tap];  mode 2: NIN matrix [cin][cout] (layers.py:639-650).
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_dev_kernel(const float* __restrict__ src, int mode, int cout, int cin, int ntaps, int cout_pad, int ck,
                                                            T* __restrict__ dst, T* __restrict__ dstb, const float* __restrict__ bias,
                                                            float* __restrict__ bias_out) {
    const long total = (long)ntaps * cout_pad * cin;
    constexpr int vec = 16 / (int)sizeof(T);
    if (blockIdx.x == 0)                                                          // the zero-padded bias rides along
        for (int i = threadIdx.x; i < cout_pad; i += 256) bias_out[i] = (bias && i < cout) ? bias[i] : 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ci = (int)(idx % cin);
        const int tap = (int)((idx / cin) % ntaps);
        const int co = (int)(idx / ((long)cin * ntaps));
        float v = 0.f;
        if (co < cout)
            v = mode == 0 ? src[((size_t)co * cin + ci) * ntaps + tap]
              : mode == 1 ? src[((size_t)ci * cout + co) * ntaps + (ntaps - 1 - tap)]
                          : src[(size_t)ci * cout + co];
        const T o = (T)v;
        dst[idx] = o;                                                         // idx == (co * ntaps + tap) * cin + ci
        if (ck) {
            const int e = ci % ck;
            dstb[(((size_t)tap * (cin / ck) + ci / ck) * cout_pad + co) * ck + (((e / vec) ^ ((co >> 2) & 3)) * vec + e % vec)] = o;
        }
    }
}

// ---- launch wrappers ------------------------------------------------------------------------------------------------------------
void launch_attention_bwd(const float* q, const float* k, const float* v, const float* dO, float* work, float* dq, float* dk, float* dv, int B,
                          int N, int C, hipStream_t s) {
    float* P = work; float* dS = work + (size_t)B * N * N;
    hipLaunchKernelGGL(attn_bwd_rows_kernel, dim3(N, B), dim3(128), (size_t)(2 * N + 2 * C) * 4, s, q, k, v, dO, P, dS, dq, N, C);
    hipLaunchKernelGGL(attn_bwd_cols_kernel, dim3(N, B), dim3(128), 0, s, q, dO, P, dS, dk, dv, N, C);
}
static int g_wgrad_blocks = 512;      // two workgroups per CU in one round (101 vs 105 ms per training step with 1024)
void wgrad_set_blocks(int n) { g_wgrad_blocks = n > 0 ? n : 512; }
static WgradPlan wgrad_plan(int B, int H, int W, int Cout, int Cin, int cw_max = 8, int max_patch = WH_PATCH) {
    WgradPlan q;
    q.CW = std::min(W, cw_max);
    q.RH = std::min(WG_PX / q.CW, H);
    while (H % q.RH || (q.RH + 2) * (q.CW + 2) > max_patch) --q.RH;    // a chunk stays inside one image; the halo patch fits its buffer
    q.chunks_x = (W + q.CW - 1) / q.CW;
    q.units = B * (H / q.RH) * q.chunks_x;
    const int tiles = ((Cout + WG_T - 1) / WG_T) * ((Cin + WG_T - 1) / WG_T);
    int ns = std::max(1, g_wgrad_blocks / tiles);                     // about 2 workgroups per CU in total ...
    ns = std::min(ns, std::max(1, q.units / 4));                      // ... of at least 4 chunks each
    q.per_slice = (q.units + ns - 1) / ns;
    q.nslices = (q.units + q.per_slice - 1) / q.per_slice;
    q.tiles = tiles;
    return q;
}
static int g_wgrad_mfma16 = 1;        // 16-bit tensors: 1 = wgrad16_kernel (16-bit MFMA), 0 = wgrad_tile_kernel (converted while staging, fp32 MFMA)
void wgrad_set_mfma16(int v) { g_wgrad_mfma16 = v; }
static bool wgrad_use16(int dtype, int Cout, int Cin) { return dtype != DT_F32 && g_wgrad_mfma16 && Cout % 8 == 0 && Cin % 8 == 0; }
size_t wgrad_workspace_floats(int B, int H, int W, int Cout, int Cin, int ntaps, int dtype) {
    if (Cout % 4 || Cin % 4) return 0;
    const WgradPlan q = wgrad_plan(B, H, W, Cout, Cin, 8, wgrad_use16(dtype, Cout, Cin) ? WH_PATCH : WG_PATCH);
    return (size_t)q.nslices * ((size_t)Cout * Cin * ntaps + Cout);
}
template <typename T>
static void wgrad_tile_t(const void* dy, const void* x, float* part, float* bpart, int B, int H, int W, int Cout, int Cin, int ntaps, const WgradPlan& q,
                         hipStream_t s) {
    static LdsAttrOnce attr;
    attr.once([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tile_kernel<9, T>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tile_kernel<1, T>), hipFuncAttributeMaxDynamicSharedMemorySize, WG_SMEM);
    });
    const dim3 grid(q.tiles * q.nslices);
    if (ntaps == 9) hipLaunchKernelGGL((wgrad_tile_kernel<9, T>), grid, dim3(256), WG_SMEM, s, (const T*)dy, (const T*)x, part, bpart, B, H, W, Cout, Cin, q);
    else            hipLaunchKernelGGL((wgrad_tile_kernel<1, T>), grid, dim3(256), WG_SMEM, s, (const T*)dy, (const T*)x, part, bpart, B, H, W, Cout, Cin, q);
}
template <typename T>
static void wgrad16_t(const void* dy, const void* x, float* part, float* bpart, int B, int H, int W, int Cout, int Cin, int ntaps, const WgradPlan& q,
                      hipStream_t s) {
    static LdsAttrOnce attr;
    attr.once([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16_kernel<9, T>), hipFuncAttributeMaxDynamicSharedMemorySize, WH_SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad16_kernel<1, T>), hipFuncAttributeMaxDynamicSharedMemorySize, WH_SMEM);
    });
    const dim3 grid(q.tiles * q.nslices);
    if (ntaps == 9) hipLaunchKernelGGL((wgrad16_kernel<9, T>), grid, dim3(256), WH_SMEM, s, (const T*)dy, (const T*)x, part, bpart, B, H, W, Cout, Cin, q);
    else            hipLaunchKernelGGL((wgrad16_kernel<1, T>), grid, dim3(256), WH_SMEM, s, (const T*)dy, (const T*)x, part, bpart, B, H, W, Cout, Cin, q);
}
// dy / x in `dtype`: 16-bit tensors on the 16-bit matrix pipe (wgrad16_kernel), fp32 on the exact-fp32 one; dw / db fp32.  16-bit inputs need the
// tiled kernel (work != nullptr, channel counts multiples of 4): returns false otherwise.
bool launch_wgrad(const void* dy, const void* x, int dtype, float* dw, float* db, int B, int H, int W, int Cout, int Cin, int ntaps, float alpha,
                  float* work, hipStream_t s) {
    if (work && Cout % 4 == 0 && Cin % 4 == 0) {
        const bool m16 = wgrad_use16(dtype, Cout, Cin);
        const WgradPlan q = wgrad_plan(B, H, W, Cout, Cin, 8, m16 ? WH_PATCH : WG_PATCH);
        const long n = (long)Cout * Cin * ntaps;
        float* part = work; float* bpart = work + (size_t)q.nslices * n;
        if (m16) {
            if (dtype == DT_BF16) wgrad16_t<__bf16>(dy, x, part, db ? bpart : nullptr, B, H, W, Cout, Cin, ntaps, q, s);
            else wgrad16_t<_Float16>(dy, x, part, db ? bpart : nullptr, B, H, W, Cout, Cin, ntaps, q, s);
        } else if (dtype == DT_F32) wgrad_tile_t<float>(dy, x, part, db ? bpart : nullptr, B, H, W, Cout, Cin, ntaps, q, s);
        else if (dtype == DT_BF16) wgrad_tile_t<__bf16>(dy, x, part, db ? bpart : nullptr, B, H, W, Cout, Cin, ntaps, q, s);
        else wgrad_tile_t<_Float16>(dy, x, part, db ? bpart : nullptr, B, H, W, Cout, Cin, ntaps, q, s);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long>((n + Cout + 255) / 256, 4096)), dim3(256), 0, s, part, q.nslices, n, alpha, dw,
                           bpart, db ? Cout : 0, db);
        return true;
    }
    if (dtype != DT_F32) return false;
    const long npix = (long)B * H * W;
    int nslices = (int)std::min<long>(64, std::max<long>(1, npix / 4096));     // pixel slices (atomic accumulation when > 1)
    if (nslices > 1) {
        (void)hipMemsetAsync(dw, 0, (size_t)Cout * Cin * ntaps * 4, s);
        if (db) (void)hipMemsetAsync(db, 0, (size_t)Cout * 4, s);
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3((Cout + 31) / 32, (Cin + 31) / 32, ntaps * nslices), dim3(256), 0, s, (const float*)dy, (const float*)x, dw, db, B,
                       H, W, Cout, Cin, ntaps, nslices, alpha);
    return true;
}
static int vec_of(int dtype) { return dtype == DT_F32 ? 4 : 8; }
static bool gn_sliced(int C, int G, int dtype) { const int v = vec_of(dtype); return C % v == 0 && C / v <= 256 && C <= 1024 && G <= 1024; }
size_t gn_workspace_floats(int B, int C, int G) {
    // [fp64 partials: 2 B slices max(C, G)] [mean, rstd: 2 B G] [s1, s2: 2 B C] [m1, m2: 2 B G]
    return (size_t)4 * B * GN_MAX_SLICES * std::max(C, G) + (size_t)4 * B * G + (size_t)2 * B * C;
}
// x / dy / y / dx in `dtype` (fp32, bf16, fp16 storage); statistics, affine parameters and their gradients fp32.  The 16-bit types need
// the sliced kernels (part != nullptr, C a multiple of 8): returns false when the case cannot be served.
template <typename T>
static void gn_stats_t(const void* x, int B, int HW, int C, int G, float eps, float* mean, float* rstd, double* part, hipStream_t s) {
    const int ns = gn_slices(HW, C, Vec16<T>::N);
    hipLaunchKernelGGL(gn_stats_part_kernel<T>, dim3(ns, B), dim3(256), 0, s, (const T*)x, HW, C, G, part);
    hipLaunchKernelGGL(gn_stats_fin_kernel, dim3(B * G), dim3(64), 0, s, part, ns, G, (double)HW * (C / G), eps, mean, rstd);
}
bool launch_gn_stats(const void* x, int dtype, int B, int HW, int C, int G, float eps, float* mean, float* rstd, double* part, hipStream_t s) {
    if (part && gn_sliced(C, G, dtype)) {
        if (dtype == DT_F32) gn_stats_t<float>(x, B, HW, C, G, eps, mean, rstd, part, s);
        else if (dtype == DT_BF16) gn_stats_t<__bf16>(x, B, HW, C, G, eps, mean, rstd, part, s);
        else gn_stats_t<_Float16>(x, B, HW, C, G, eps, mean, rstd, part, s);
        return true;
    }
    if (dtype != DT_F32) return false;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * G), dim3(256), 0, s, (const float*)x, HW, C, G, eps, mean, rstd);
    return true;
}
template <typename T>
static void gn_act_bwd_t(const void* x, const void* dy, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                         const void* add, float add_scale, int B, int HW, int C, int G, float* s1, float* s2, float* m12, double* part, void* dx,
                         hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    const long n = (long)B * HW * C;
    const int ns = gn_slices(HW, C, VEC);
    float* m1 = m12; float* m2 = m12 + (size_t)B * G;
    const unsigned blocks = (unsigned)std::min<long>((n / VEC + 255) / 256, 8192);
    if (act) hipLaunchKernelGGL((gn_act_bwd_part_kernel<T, true>), dim3(ns, B), dim3(256), 0, s, (const T*)x, (const T*)dy, mean, rstd, gamma, beta, HW, C, G, part);
    else     hipLaunchKernelGGL((gn_act_bwd_part_kernel<T, false>), dim3(ns, B), dim3(256), 0, s, (const T*)x, (const T*)dy, mean, rstd, gamma, beta, HW, C, G, part);
    hipLaunchKernelGGL(gn_act_bwd_fin_kernel, dim3(B * G), dim3(64), 0, s, part, gamma, ns, HW, C, G, s1, s2, m1, m2);
    if (act) hipLaunchKernelGGL((gn_act_bwd_applyv_kernel<T, true>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const T*)dy, mean, rstd, gamma, beta, m1, m2, (const T*)add, add_scale, HW, C, G, (T*)dx, n / VEC);
    else     hipLaunchKernelGGL((gn_act_bwd_applyv_kernel<T, false>), dim3(blocks), dim3(256), 0, s, (const T*)x, (const T*)dy, mean, rstd, gamma, beta, m1, m2, (const T*)add, add_scale, HW, C, G, (T*)dx, n / VEC);
}
bool launch_gn_act_bwd(const void* x, const void* dy, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                       const void* add, float add_scale, int B, int HW, int C, int G, float* s1, float* s2, float* m12, double* part, void* dx,
                       float* dgamma, float* dbeta, hipStream_t s) {
    const long n = (long)B * HW * C;
    if (part && m12 && gn_sliced(C, G, dtype)) {
        if (dtype == DT_F32) gn_act_bwd_t<float>(x, dy, mean, rstd, gamma, beta, act, add, add_scale, B, HW, C, G, s1, s2, m12, part, dx, s);
        else if (dtype == DT_BF16) gn_act_bwd_t<__bf16>(x, dy, mean, rstd, gamma, beta, act, add, add_scale, B, HW, C, G, s1, s2, m12, part, dx, s);
        else gn_act_bwd_t<_Float16>(x, dy, mean, rstd, gamma, beta, act, add, add_scale, B, HW, C, G, s1, s2, m12, part, dx, s);
        hipLaunchKernelGGL(gn_param_grads_kernel, dim3((C + 127) / 128), dim3(128), 0, s, s1, s2, B, C, dgamma, dbeta);
        return true;
    }
    if (dtype != DT_F32) return false;
    const float* xf = (const float*)x; const float* dyf = (const float*)dy; const float* addf = (const float*)add; float* dxf = (float*)dx;
    const dim3 gr((C + 63) / 64, B);
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 4096);
    if (act) {
        hipLaunchKernelGGL(gn_act_bwd_reduce<true>, gr, dim3(256), 0, s, xf, dyf, mean, rstd, gamma, beta, HW, C, G, s1, s2);
        hipLaunchKernelGGL(gn_act_bwd_apply<true>, dim3(blocks), dim3(256), 0, s, xf, dyf, mean, rstd, gamma, beta, s1, s2, addf, add_scale, HW, C, G, dxf, n);
    } else {
        hipLaunchKernelGGL(gn_act_bwd_reduce<false>, gr, dim3(256), 0, s, xf, dyf, mean, rstd, gamma, beta, HW, C, G, s1, s2);
        hipLaunchKernelGGL(gn_act_bwd_apply<false>, dim3(blocks), dim3(256), 0, s, xf, dyf, mean, rstd, gamma, beta, s1, s2, addf, add_scale, HW, C, G, dxf, n);
    }
    hipLaunchKernelGGL(gn_param_grads_kernel, dim3((C + 127) / 128), dim3(128), 0, s, s1, s2, B, C, dgamma, dbeta);
    return true;
}
template <typename T>
static void gn_act_fwd_t(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int B, int HW, int C,
                         int G, void* y, hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    const long n = (long)B * HW * C;
    const unsigned blocks = (unsigned)std::min<long>((n / VEC + 255) / 256, 8192);
    if (act) hipLaunchKernelGGL((gn_act_fwdv_kernel<T, true>), dim3(blocks), dim3(256), 0, s, (const T*)x, mean, rstd, gamma, beta, HW, C, G, (T*)y, n / VEC);
    else     hipLaunchKernelGGL((gn_act_fwdv_kernel<T, false>), dim3(blocks), dim3(256), 0, s, (const T*)x, mean, rstd, gamma, beta, HW, C, G, (T*)y, n / VEC);
}
bool launch_gn_act_fwd(const void* x, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta, int act, int B, int HW,
                       int C, int G, void* y, hipStream_t s) {
    const long n = (long)B * HW * C;
    if (C % vec_of(dtype) == 0) {
        if (dtype == DT_F32) gn_act_fwd_t<float>(x, mean, rstd, gamma, beta, act, B, HW, C, G, y, s);
        else if (dtype == DT_BF16) gn_act_fwd_t<__bf16>(x, mean, rstd, gamma, beta, act, B, HW, C, G, y, s);
        else gn_act_fwd_t<_Float16>(x, mean, rstd, gamma, beta, act, B, HW, C, G, y, s);
        return true;
    }
    if (dtype != DT_F32) return false;
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 4096);
    if (act) hipLaunchKernelGGL(gn_act_fwd_kernel<true>, dim3(blocks), dim3(256), 0, s, (const float*)x, mean, rstd, gamma, beta, HW, C, G, (float*)y, n);
    else     hipLaunchKernelGGL(gn_act_fwd_kernel<false>, dim3(blocks), dim3(256), 0, s, (const float*)x, mean, rstd, gamma, beta, HW, C, G, (float*)y, n);
    return true;
}
template <typename T>
static void colsum_t(const void* x, int B, int HW, int C, float scale, float* out, double* part, hipStream_t s) {
    const int ns = std::min(64, gn_slices(HW, C, Vec16<T>::N));          // <= 64 slices: the finishing kernel walks them serially
    hipLaunchKernelGGL(colsum_part_kernel<T>, dim3(ns, B), dim3(256), 0, s, (const T*)x, HW, C, part);
    hipLaunchKernelGGL(colsum_fin_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, part, ns, C, B * C, scale, out);
}
// x in `dtype`; part = nullptr: the one-block-per-(item, 64 channels) fp32 kernel.  work: 2 * 64 * B * C floats (fp64 partials)
bool launch_colsum(const void* x, int dtype, int B, int HW, int C, float scale, float* out, double* part, hipStream_t s) {
    if (part && gn_sliced(C, 1, dtype)) {
        if (dtype == DT_F32) colsum_t<float>(x, B, HW, C, scale, out, part, s);
        else if (dtype == DT_BF16) colsum_t<__bf16>(x, B, HW, C, scale, out, part, s);
        else colsum_t<_Float16>(x, B, HW, C, scale, out, part, s);
        return true;
    }
    if (dtype != DT_F32) return false;
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 63) / 64, B), dim3(256), 0, s, (const float*)x, HW, C, scale, out);
    return true;
}
void launch_dense_bwd(const float* g, const float* temb, const float* Wd, int B, int K, int Cout, float* dW, float* db, float* dtemb, hipStream_t s) {
    hipLaunchKernelGGL(dense_bwd_kernel, dim3(1), dim3(256), 0, s, g, temb, Wd, B, K, Cout, dW, db, dtemb);
}

void launch_pack_conv_dev(const float* src, int mode, int cout, int cin, int ntaps, int cout_pad, int dtype, int ck, void* dst, void* dstb,
                          const float* bias, float* bias_out, hipStream_t s) {
    const long total = (long)ntaps * cout_pad * cin;
    const unsigned blocks = (unsigned)std::min<long>((total + 255) / 256, 8192);
    if (dtype == DT_F32)       hipLaunchKernelGGL(pack_conv_dev_kernel<float>, dim3(blocks), dim3(256), 0, s, src, mode, cout, cin, ntaps, cout_pad, dstb ? ck : 0, (float*)dst, (float*)dstb, bias, bias_out);
    else if (dtype == DT_BF16) hipLaunchKernelGGL(pack_conv_dev_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, src, mode, cout, cin, ntaps, cout_pad, dstb ? ck : 0, (__bf16*)dst, (__bf16*)dstb, bias, bias_out);
    else                       hipLaunchKernelGGL(pack_conv_dev_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, src, mode, cout, cin, ntaps, cout_pad, dstb ? ck : 0, (_Float16*)dst, (_Float16*)dstb, bias, bias_out);
}

}  // namespace use
