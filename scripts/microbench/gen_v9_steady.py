#!/usr/bin/env python3
"""Generates scripts/microbench/v9_steady.hip: the steady state of a one-wave-per-SIMD (4 waves, 512 registers) persistent form of
the 3x3 convolution - VERDICT r3 #2, gate (i): "can the bare structure beat 1.1 PFLOP/s on random data?" - without the tile switch,
the prologue and the epilogue, i.e. an upper bound for that design on the whole chip, power management included.

One workgroup per CU, 8 x 32 px x 128 output channels per tile, wave w owns tile rows 2w, 2w+1 (2 x 4 MFMA tiles, 128 accumulators in
AGPRs), K in 16-channel chunks with all nine taps of a chunk in LDS (72 MFMAs per wave per chunk, ONE barrier per chunk, 8 MFMAs before the
chunk's end so that the first fragments of the next chunk are read behind them), operands swapped (A = weights, B = pixels).
Per chunk and wave: 48 fragment reads (36 weight + 12 pixel), 9 LDS-DMA weight loads (chunk c+2), 3 halo loads (chunk c+3),
3 pieces transformed (GroupNorm affine + SiLU, conv_v4's 15-instruction dword chain, 4 per MFMA gap) and stored to LDS (chunk c+2).

MFMAs + VALU are asm statements (hipcc would move them), memory operations are compiler-visible (volatile) so that hipcc's waitcnt
insertion counts them.  MODE bits switch parts off: 1 fragment reads, 2 weight DMA, 4 halo loads + stores, 8 transform, 16 barrier."""
import sys, os
TWO = int(os.environ.get("V9_TWO", "0"))

HB = [0, 11264]                     # halo buffers: 10 x 34 px x 32 B = 10880
WB = [22528, 22528 + (0 if TWO else 36864)]   # weight buffers: 9 taps x 128 co x 32 B (TWO: aliased - timing only)
LDS = WB[1] + 36864

LO = 'v_lshlrev_b32 %[xl], 16, %[d]'
HI = 'v_and_b32 %[xh], 0xffff0000, %[d]'
SLICES = [
    [LO, HI, 'v_fma_f32 %[ul], %[xl], %[al], %[bl]', 'v_fma_f32 %[uh], %[xh], %[ah], %[bh]'],
    ['v_mul_f32 %[xl], 0xbfb8aa3b, %[ul]', 'v_mul_f32 %[xh], 0xbfb8aa3b, %[uh]', 'v_exp_f32 %[xl], %[xl]', 'v_exp_f32 %[xh], %[xh]'],
    ['v_add_f32 %[xl], 1.0, %[xl]', 'v_add_f32 %[xh], 1.0, %[xh]', 'v_rcp_f32 %[xl], %[xl]', 'v_rcp_f32 %[xh], %[xh]'],
    ['v_mul_f32 %[ul], %[ul], %[xl]', 'v_mul_f32 %[uh], %[uh], %[xh]', 'v_cvt_pk_bf16_f32 %[d], %[ul], %[uh]'],
]


def xidx(i, t):
    return (i + t // 3) * 3 + t % 3


SL3 = [   # the 15-instruction dword chain as five slices of three (a transcendental's result is read two instructions later)
    [LO, HI, 'v_fma_f32 %[ul], %[xl], %[al], %[bl]'],
    ['v_fma_f32 %[uh], %[xh], %[ah], %[bh]', 'v_mul_f32 %[xl], 0xbfb8aa3b, %[ul]', 'v_mul_f32 %[xh], 0xbfb8aa3b, %[uh]'],
    ['v_exp_f32 %[xl], %[xl]', 'v_exp_f32 %[xh], %[xh]', 'v_add_f32 %[xl], 1.0, %[xl]'],
    ['v_add_f32 %[xh], 1.0, %[xh]', 'v_rcp_f32 %[xl], %[xl]', 'v_rcp_f32 %[xh], %[xh]'],
    ['v_mul_f32 %[ul], %[ul], %[xl]', 'v_mul_f32 %[uh], %[uh], %[xh]', 'v_cvt_pk_bf16_f32 %[d], %[ul], %[uh]'],
]
XF_FIRST = 6          # first gap of the transform
DMA_GAPS = [4 + 5 * q for q in range(9)]
HLOAD_GAPS = [1, 2, 3]


def window(b):
    """one window = barrier, tap 8 of the chunk in buffers b, taps 0..7 of the chunk in buffers b^1 (72 MFMAs, gap g behind MFMA g)."""
    L = []
    nb = b ^ 1
    L.append('if (!(MODE & 16)) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\ts_barrier" ::: "memory"); }')
    newx = {0: [(0, 0), (1, 0)], 1: [(0, 1), (1, 1)], 2: [(0, 2), (1, 2)], 3: [(2, 0)], 4: [(2, 1)], 5: [(2, 2)], 6: [(3, 0)], 7: [(3, 1)], 8: [(3, 2)]}
    seq = [(8, b)] + [(t, nb) for t in range(8)]
    # fragment reads in the order of need: for sequence positions 1..8 of this window and position 0 of the next one (tap 8 of chunk nb)
    mem = [[] for _ in range(72)]
    g_next = 0
    R = 2 if TWO else 3
    for pos in list(range(1, 9)) + [9]:          # ring slot pos % R is free once position pos - R is done: not before gap (pos - R + 1) * 8
        t, buf = (seq[pos] if pos < 9 else (8, nb))
        g_next = max(g_next, (pos - R + 1) * 8)
        for j in range(4):
            mem[g_next].append(f'if (!(MODE & 1)) wf[{pos % R}][{j}] = LDSV(wbase + {WB[buf] + (t * 128 + j * 32) * 32});'); g_next += 1
        for (r, dx) in newx[t]:
            mem[g_next].append(f'if (!(MODE & 1)) xf[{r * 3 + dx}] = LDSV(xbase + {HB[buf] + (r * 34 + dx) * 32});'); g_next += 1
    for q, g in enumerate(DMA_GAPS):
        mem[g].append(f'if (!(MODE & 2)) {{ DMA(wsoff + {q * 4096}u, {WB[b] + q * 4096}); }}')
    mem[DMA_GAPS[-1]].append('wsoff += 36864u;')
    for k, g in enumerate(HLOAD_GAPS):
        mem[g].append(f'if (!(MODE & 4)) hs[{nb}][{k}] = HLOAD(hvoff[{k}], hsoff);')
    mem[HLOAD_GAPS[-1]].append('hsoff += 32u;')
    for g in range(72):
        gi, k8 = g // 8, g % 8
        t = seq[gi][0]
        i, j = k8 // 4, k8 % 4
        ops = f'[acc] "+a"(acc[{i}][{j}])'
        ins = f'[w] "v"(wf[{gi % (2 if TWO else 3)}][{j}]), [x] "v"(xf[{xidx(i, t)}])'
        bare = f'asm volatile("v_mfma_f32_32x32x16_bf16 %[acc], %[w], %[x], %[acc]" : {ops} : {ins});'
        n = g - XF_FIRST
        if 0 <= n < 60:
            p, q, sl = n // 20, (n % 20) // 5, n % 5
            if sl == 0:
                L.append(f'if (!(MODE & 12) || ((MODE & 12) == 4)) hcd[{q}] = hs[{b}][{p}][{q}];' if q else
                         f'{{ hcd[0] = hs[{b}][{p}][0]; hcd[1] = hs[{b}][{p}][1]; hcd[2] = hs[{b}][{p}][2]; hcd[3] = hs[{b}][{p}][3]; }}')
            body = '\\n\\t'.join(['v_mfma_f32_32x32x16_bf16 %[acc], %[w], %[x], %[acc]'] + SL3[sl])
            fops = ops + f', [d] "+v"(hcd[{q}]), [xl] "+v"(xl), [xh] "+v"(xh), [ul] "+v"(ul), [uh] "+v"(uh)'
            fins = ins + f', [al] "v"(ca[{2 * q}]), [bl] "v"(cb[{2 * q}]), [ah] "v"(ca[{2 * q + 1}]), [bh] "v"(cb[{2 * q + 1}])'
            L.append(f'if (MODE & 8) {bare}')
            L.append(f'else asm volatile("{body}" : {fops} : {fins});')
            if n % 20 == 19:
                mem[g].append(f'if (!(MODE & 4)) {{ u32x4 t_ = {{hcd[0], hcd[1], hcd[2], hcd[3]}}; LDSST(hdst[{p}] + {HB[b]}, t_); }}')
        else:
            L.append(bare)
        L += mem[g]
    return L


WPEV = 2 if TWO else 1
out = f'''// GENERATED by gen_v9_steady.py - do not edit.  hipcc --offload-arch=gfx950 -O3 v9_steady.hip -o v9_steady
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define WPE {WPEV}
#define GRID {512 if TWO else 256}
#define LDS_BYTES {LDS}
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) volatile u32x4 lds_vu4;
#define LDSV(OFF) (*(lds_vu4*)(sm3 + (OFF)))
#define LDSST(OFF, V) (*(lds_vu4*)(sm3 + (OFF)) = (V))
#define HLOAD(VO, SO) __builtin_amdgcn_raw_buffer_load_b128(rs_x, (VO), (SO), 0)
#define DMA(SO, LOFF) {{ char* dst_ = smem + (LOFF) + wave * 1024; __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)dst_, 16, wvoff, (SO), 0, 0); }}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void v9_steady(const void* gx, const void* gw, float* sink, unsigned long long* cyc, int nwin, unsigned xbytes, unsigned wbytes) {{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gx), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gw), 0, wbytes, 0x00020000);
    // LDS starts as random bf16 (the first fragments come from it)
    for (int i = tid; i < {LDS} / 16; i += 256)
        *reinterpret_cast<u32x4*>(smem + i * 16) = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)(i * 16 + blockIdx.x * 4096), 0, 0);
    __syncthreads();
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    u32x4 wf[3][4], xf[12], hs[2][3]; unsigned hcd[4];
    const unsigned fl = (unsigned)((lane & 31) * 32 + (lane >> 5) * 16);
    lds_char* const sm3 = (lds_char*)smem;
    const unsigned wbase = fl, xbase = fl + (unsigned)(2 * wave * 34 * 32);
    for (int j = 0; j < 4; ++j) {{ wf[0][j] = LDSV(wbase + {WB[0]} + j * 1024); wf[1][j] = LDSV(wbase + {WB[1]} + j * 1024); wf[2][j] = LDSV(wbase + {WB[1]} + j * 1024 + 4096); }}
    for (int n = 0; n < 12; ++n) xf[n] = LDSV(xbase + (n / 3 * 34 + n % 3) * 32);
    // halo pieces: piece p = tid + 256 k of the 680 of a chunk -> a 16-byte piece of pixel (row, px): addresses in a 512 x 640 x 128 map
    unsigned hvoff[3], hdst[3];
    const unsigned tile = blockIdx.x;                                  // tiles along a row of the map, 32 px apart
    for (int k = 0; k < 3; ++k) {{
        const int p = min(tid + 256 * k, 679), row = p / 68, px = (p % 68) >> 1, half = p & 1;
        hvoff[k] = (unsigned)(((row + (tile / 20) * 8) * 640 + (tile % 20) * 32 + px) * 256 + half * 16);
        hdst[k] = (unsigned)(p * 16);
        hs[0][k] = HLOAD(hvoff[k], 0u); hs[1][k] = hs[0][k];
    }}
    unsigned hsoff = 0u, wsoff = 0u;
    const unsigned wvoff = (unsigned)tid * 16u;
    for (int q = 0; q < 4; ++q) hcd[q] = 0x3f803f80u;
    float ca[8], cb[8], xl = 0.f, xh = 0.f, ul = 0.f, uh = 0.f;
    for (int c = 0; c < 8; ++c) {{ ca[c] = 0.7f + 0.01f * ((lane + c) & 7); cb[c] = 0.05f * (c - 3); }}
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < nwin; it += 2) {{
        if ((it & 7) == 0) {{ hsoff = 0u; wsoff = 0u; for (int k = 0; k < 3; ++k) {{ hvoff[k] += 8u * 640u * 256u * 12u; if (hvoff[k] >= xbytes - (1u << 24)) hvoff[k] -= xbytes - (1u << 25); }} }}   // next tile: 8 chunks later
'''
for b in (0, 1):
    out += '        // ---------------- window on buffers %d ----------------\n' % b
    out += ''.join('        ' + l + '\n' for l in window(b))
out += '''    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = xl + xh + ul + uh;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    for (int k = 0; k < 4; ++k) s += __builtin_bit_cast(float, hcd[k]);
    sink[blockIdx.x * 256 + tid] = s;
    if (lane == 0 && blockIdx.x == 97) cyc[wave] = t1 - t0;
}

template <int MODE> static void run(const char* what, const void* gx, const void* gw, float* sink, unsigned long long* cyc, int nwin, unsigned xb, unsigned wb) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&v9_steady<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, ''' + str(LDS) + ''');
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(cyc, 0, 64);
        hipEventRecord(e0);
        for (int l = 0; l < 5; ++l) hipLaunchKernelGGL(v9_steady<MODE>, dim3(''' + str(512 if TWO else 256) + '''), dim3(256), ''' + str(LDS) + ''', 0, gx, gw, sink, cyc, nwin, xb, wb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5.f;
        if (rep && ms < best) best = ms;
    }
    unsigned long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
    const double flops = ''' + str(512.0 if TWO else 256.0) + ''' * 4 * (double)nwin * 72 * 32768.0;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s   %7.1f cycles per 72-MFMA window (ideal 2304)  clock %.2f GHz\\n", what, best, flops / best * 1e-9,
           (double)h[0] / nwin, (double)h[0] / (best * 1e6));
}

int main(int argc, char** argv) {
    const size_t xb = (size_t)4 * 512 * 640 * 128 * 2, wb = (size_t)1 << 24;
    void *gx, *gw; float* sink; unsigned long long* cyc;
    hipMalloc(&gx, xb); hipMalloc(&gw, wb); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 64);
    std::vector<unsigned short> h(xb / 2);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) {                            // random bf16 of magnitude ~1 (sign, exponent 125..127, random mantissa)
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (unsigned short)(((s >> 20) & 0x8000) | ((125 + (s >> 40) % 3) << 7) | ((s >> 8) & 0x7f));
    }
    hipMemcpy(gx, h.data(), xb, hipMemcpyHostToDevice); hipMemcpy(gw, h.data(), wb, hipMemcpyHostToDevice);
    const int nwin = 8 * 10 * 8;                                       // 80 tiles of 8 chunks per workgroup
    if (argc > 2) {                                                    // soak: one mode back to back for argv[2] seconds (power / clock sampling from outside)
        const int mode = atoi(argv[1]); const double secs = atof(argv[2]);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        double total_ms = 0.0; long launches = 0;
        while (total_ms < secs * 1e3) {
            hipEventRecord(e0);
            for (int l = 0; l < 50; ++l) {
#define SOAK(M) case M: hipFuncSetAttribute(reinterpret_cast<const void*>(&v9_steady<M>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); hipLaunchKernelGGL(v9_steady<M>, dim3(GRID), dim3(256), LDS_BYTES, 0, gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb); break;
                switch (mode) { SOAK(31) SOAK(0) SOAK(8) SOAK(23) default: return 2; }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; launches += 50;
        }
        printf("mode %d: %ld launches in %.1f s -> %.1f TFLOP/s sustained\\n", mode, launches, total_ms * 1e-3,
               (double)GRID * 4 * (double)nwin * 72 * 32768.0 * launches / (total_ms * 1e-3) * 1e-12);
        return 0;
    }
    run<31>("MFMAs only", gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb);
    run<30>("+ fragment reads", gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb);
    run<14>("+ fragment reads + barrier", gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb);
    run<12>("+ weight DMA", gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb);
    run<8>("+ halo loads and stores", gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb);
    run<0>("+ transform = everything", gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb);
    run<23>("MFMAs + transform only", gx, gw, sink, cyc, nwin, (unsigned)xb, (unsigned)wb);
    return 0;
}
'''
open(sys.argv[1] if len(sys.argv) > 1 else ("scripts/microbench/v9_steady2.hip" if TWO else "scripts/microbench/v9_steady.hip"), "w").write(out)
