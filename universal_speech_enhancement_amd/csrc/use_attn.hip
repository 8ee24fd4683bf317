// attn_fused_kernel: the bottleneck self-attention block of the score network (AttnBlockpp, reference layerspp.py:60-93) as ONE
// launch, one workgroup per batch item:
//
//     h = GroupNorm(x);  q, k, v = NIN_0..2(h);  P = softmax(q k^T / sqrt(C));  out = (x + NIN_3(P v)) / sqrt(2)
//
// on the [N <= 96 tokens][C = 256] map of the deepest level (8 x 10 = 80 tokens at T' = 640).  Every contraction runs on the matrix
// pipe (v_mfma_f32_32x32x16 in the storage type): the three input projections as one pass over h with three weight matrices
// (9 accumulator tiles per wave), q k^T as 3 x 3 tiles over the waves, P v and the output projection with one 32-channel block per
// wave.  Operands move through three LDS regions (h -> q / S -> O, k -> P, v^T); weights are read from the blob in MFMA operand
// order (NIN matrices are stored [Cout][Cin], K-contiguous).  The GroupNorm is finalised in the prologue from the producer's
// fixed-point totals, the GroupNorm totals of the output are accumulated in the epilogue - the block replaces six launches
// (3 x NIN, attention core, NIN_3, and the finalize in front of them).
//
// Rounding points are those of the unfused path (q, k, v and the attention output are stored in the activation type between the
// operators there as well) plus one: the probabilities enter P v in the storage type (the unfused core kept them in fp32).
#include "use_kernels.h"
#include "use_device.h"

namespace use {

constexpr int AT_C = 256, AT_NMAX = 96;
constexpr int AT_PITCH = (AT_C + 8) * 2;                    // 528 B rows of the [token][channel] images: conflict-free ds_read_b128
constexpr int AT_TP = AT_NMAX + 8;                          // 104 elements: row pitch of v^T [channel][token] and P [token][token]
constexpr int AT_SP = AT_NMAX + 4;                          // 100 floats: row pitch of the fp32 scores
constexpr int AT_R0 = 0, AT_R1 = AT_NMAX * AT_PITCH, AT_R2 = 2 * AT_NMAX * AT_PITCH;        // 0, 50,688, 101,376
constexpr int AT_COEF = AT_R2 + AT_C * AT_TP * 2;           // 154,624
constexpr int AT_SMEM = AT_COEF + AT_C * 8;                 // 156,672

template <typename T>
__global__ __launch_bounds__(512) void attn_fused_kernel(AttnArgs p) {
    typedef Mfma<T> MF;
    typedef typename MF::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x, N = p.N;
    const int mt = (N + 31) / 32;                            // token tiles in use (1..3)
    const T* x = (const T*)p.x + (size_t)b * N * AT_C;
    float2* const coef = reinterpret_cast<float2*>(smem + AT_COEF);

    // ---- GroupNorm of x folded to (a, b) per channel, from the producer's totals -------------------------------------------------
    if (tid < AT_C)
        coef[tid] = gn_coef_of(p.gn_st, AT_C, nullptr, 0, p.gn_gamma, p.gn_beta, p.gn_groups, 1.0f / ((float)(AT_C / p.gn_groups) * (float)N),
                               p.gn_eps, b, tid);
    __syncthreads();
    // h = a x + b into R0 [token][channel] (rows N.. of the last tile: zero)
    for (int idx = tid; idx < mt * 32 * 32; idx += 512) {
        const int t = idx >> 5, c8 = idx & 31;
        uint4 o = make_uint4(0, 0, 0, 0);
        if (t < N) {
            float v[8];
            Vec16<T>::load(x + (size_t)t * AT_C + c8 * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float2 ab = coef[c8 * 8 + k]; v[k] = fmaf(v[k], ab.x, ab.y); }
            o = Vec16<T>::pack(v);
        }
        *reinterpret_cast<uint4*>(smem + AT_R0 + t * AT_PITCH + c8 * 16) = o;
    }
    __syncthreads();

    const int n0 = wave * 32;                                // this wave's 32 output channels
    auto wfrag = [&](const void* w, int kk) -> frag {        // B operand of a projection: row n0 + l31 of [Cout][Cin], k = 16 kk + 8 h ..
        return *reinterpret_cast<const frag*>((const char*)w + ((size_t)(n0 + l31) * AT_C + kk * 16 + hh * 8) * 2);
    };
    auto afrag = [&](int region, int pitch, int i, int kk) -> frag {      // A operand: row 32 i + l31 of an LDS image
        return *reinterpret_cast<const frag*>(smem + region + (i * 32 + l31) * pitch + (kk * 16 + hh * 8) * 2);
    };

    // ---- q, k, v = h W^T + bias: one pass over h, three weight matrices ---------------------------------------------------------------
    f32x16 aq[3], ak[3], av[3];
    {
        const float bq = p.bq[n0 + l31], bk = p.bk[n0 + l31], bv = p.bv[n0 + l31];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { aq[i][r] = bq; ak[i][r] = bk; av[i][r] = bv; }
#pragma unroll 4
        for (int kk = 0; kk < AT_C / 16; ++kk) {
            const frag fq = wfrag(p.wq, kk), fk = wfrag(p.wk, kk), fv = wfrag(p.wv, kk);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i < mt) {
                    const frag a = afrag(AT_R0, AT_PITCH, i, kk);
                    aq[i] = MF::mma(a, fq, aq[i]); ak[i] = MF::mma(a, fk, ak[i]); av[i] = MF::mma(a, fv, av[i]);
                }
            }
        }
    }
    __syncthreads();                                         // every wave has read h: R0 takes q, R1 k, R2 v^T
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                *reinterpret_cast<T*>(smem + AT_R0 + row * AT_PITCH + (n0 + l31) * 2) = (T)aq[i][r];
                *reinterpret_cast<T*>(smem + AT_R1 + row * AT_PITCH + (n0 + l31) * 2) = (T)ak[i][r];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {                    // v^T [channel][token]: 4 consecutive tokens per register group
                T v4[4] = {(T)av[i][4 * g], (T)av[i][4 * g + 1], (T)av[i][4 * g + 2], (T)av[i][4 * g + 3]};
                *reinterpret_cast<uint2*>(smem + AT_R2 + ((n0 + l31) * AT_TP + i * 32 + 8 * g + 4 * hh) * 2) = *reinterpret_cast<uint2*>(v4);
            }
        }
    }
    __syncthreads();

    // ---- scores: tile (ti, tj) of q k^T per wave (9 tiles at most: wave 0 takes the ninth) ----------------------------------------------
    const float sscale = 1.0f / sqrtf((float)AT_C);         // int(C) ** -0.5 (layerspp.py:84)
    f32x16 st[2];
    int tiles[2] = {wave, wave == 0 ? 8 : -1};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int tile = tiles[u], ti = tile / 3, tj = tile - ti * 3;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[u][r] = 0.f;
        if (tile >= 0 && ti < mt && tj < mt) {
#pragma unroll 4
            for (int kk = 0; kk < AT_C / 16; ++kk)
                st[u] = MF::mma(afrag(AT_R0, AT_PITCH, ti, kk), afrag(AT_R1, AT_PITCH, tj, kk), st[u]);
        }
    }
    __syncthreads();                                         // q, k consumed: R0 takes the fp32 scores [token][token]
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int tile = tiles[u], ti = tile / 3, tj = tile - ti * 3;
        if (tile >= 0 && ti < mt && tj < mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                reinterpret_cast<float*>(smem + AT_R0)[row * AT_SP + tj * 32 + l31] = st[u][r] * sscale;
            }
        }
    }
    __syncthreads();
    // ---- softmax over the keys (fp32), probabilities in the storage type into R1 [token][token] ------------------------------------------
    for (int row = wave; row < mt * 32; row += 8) {
        const float* srow = reinterpret_cast<const float*>(smem + AT_R0) + row * AT_SP;
        const float s0 = lane < N ? srow[lane] : -INFINITY, s1 = lane + 64 < N ? srow[lane + 64] : -INFINITY;
        float m = fmaxf(s0, s1);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e0 = lane < N ? expf(s0 - m) : 0.f, e1 = lane + 64 < N ? expf(s1 - m) : 0.f;
        float sum = e0 + e1;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float inv = row < N ? 1.0f / sum : 0.f;        // padding queries: zero rows
        T* prow = reinterpret_cast<T*>(smem + AT_R1) + row * AT_TP;
        prow[lane] = (T)(e0 * inv);
        if (lane + 64 < AT_NMAX) prow[lane + 64] = (T)(e1 * inv);
    }
    __syncthreads();

    // ---- O = P v: this wave's 32 channels --------------------------------------------------------------------------------------------
    f32x16 ao[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ao[i][r] = 0.f;
    for (int kk = 0; kk < mt * 2; ++kk) {                   // K = tokens
        const frag fv = *reinterpret_cast<const frag*>(smem + AT_R2 + ((n0 + l31) * AT_TP + kk * 16 + hh * 8) * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < mt) ao[i] = MF::mma(afrag(AT_R1, AT_TP * 2, i, kk), fv, ao[i]);
    }
    // (R0 holds the scores, consumed before the last barrier: it takes O [token][channel] in the storage type)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                *reinterpret_cast<T*>(smem + AT_R0 + row * AT_PITCH + (n0 + l31) * 2) = (T)ao[i][r];
            }
        }
    }
    __syncthreads();

    // ---- out = (x + O Wo^T + bo) / sqrt(2), GroupNorm totals of the stored map -----------------------------------------------------------
    {
        const float bo = p.bo[n0 + l31];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ao[i][r] = bo;
#pragma unroll 4
        for (int kk = 0; kk < AT_C / 16; ++kk) {
            const frag fo = wfrag(p.wo, kk);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < mt) ao[i] = MF::mma(afrag(AT_R0, AT_PITCH, i, kk), fo, ao[i]);
        }
    }
    T* out = (T*)p.out + (size_t)b * N * AT_C;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (row < N) {
                    const float v = ((float)x[(size_t)row * AT_C + n0 + l31] + ao[i][r]) * 0.70710678118654752440f;
                    out[(size_t)row * AT_C + n0 + l31] = (T)v;
                    ssum += v; ssq = fmaf(v, v, ssq);
                }
            }
        }
    }
    if (p.stats) {
        ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
        if (lane < 32) gn_accumulate(p.stats + ((size_t)b * AT_C + n0 + lane) * 2, ssum, ssq);
    }
}

bool attn_fused_eligible(int dtype, int N, int C) { return dtype != DT_F32 && C == AT_C && N >= 1 && N <= AT_NMAX; }

void launch_attn_fused(const AttnArgs& a, int dtype, int B, hipStream_t s) {
    static LdsAttrOnce attr;
    attr.once([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fused_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, AT_SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fused_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, AT_SMEM);
    });
    if (dtype == DT_BF16) hipLaunchKernelGGL(attn_fused_kernel<__bf16>, dim3(B), dim3(512), AT_SMEM, s, a);
    else                  hipLaunchKernelGGL(attn_fused_kernel<_Float16>, dim3(B), dim3(512), AT_SMEM, s, a);
}

}  // namespace use
