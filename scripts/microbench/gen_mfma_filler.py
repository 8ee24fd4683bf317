#!/usr/bin/env python3
"""Generates scripts/microbench/mfma_filler.hip: what does a stream of VALU / transcendental fillers cost behind 16 back-to-back
v_mfma_f32_32x32x16_bf16 on gfx950, by filler pattern and with / without a second wave on the SIMD that reads LDS?  (round 4: sizing
the GroupNorm + SiLU transform slices of conv_v4's MFMA phases.)

Each pattern is ONE asm block of 16 MFMAs (8 accumulators in rotation, as in conv_v4) with the fillers of gap g behind MFMA g.
Registers: %[x0]..%[x15] scratch floats, %[d0]..%[d3] the raw dwords, %[c0]..%[c3] coefficients."""
import sys

MFMA = "v_mfma_f32_32x32x16_bf16 %[acc{i}], %[a], %[b], %[acc{i}]"


def chain(e, dw, half):
    """the 7 ops of element e (registers x{2e}, x{2e+1}), reading dword dw"""
    x, y = f"%[x{2*e}]", f"%[x{2*e+1}]"
    un = f"v_lshlrev_b32 {x}, 16, %[d{dw}]" if half == 0 else f"v_and_b32 {x}, 0xffff0000, %[d{dw}]"
    return [un,                                  # A
            f"v_fma_f32 {y}, {x}, %[c0], %[c1]",  # B  u
            f"v_mul_f32 {x}, 0xbfb8aa3b, {y}",    # C
            f"v_exp_f32 {x}, {x}",                # D
            f"v_add_f32 {x}, 1.0, {x}",           # E
            f"v_rcp_f32 {x}, {x}",                # F
            f"v_mul_f32 {y}, {y}, {x}"]           # G


def pattern(name):
    gaps = [[] for _ in range(17)]               # gap 16 = tail behind the last MFMA
    if name == "bare":
        pass
    elif name == "serial":                       # round-4 first form: element k's chain A B C D in gap 2k, E F G(+H) in gap 2k+1
        for k in range(8):
            c = chain(k, k // 2, k & 1)
            gaps[2 * k] += c[:4]
            gaps[2 * k + 1] += c[4:6]
            if k > 0:
                gaps[2 * k + 1] += [chain(k - 1, 0, 0)[6]]
            if k >= 2 and k % 2 == 0:
                gaps[2 * k + 1] += [f"v_cvt_pk_bf16_f32 %[d{k//2-1}], %[x{2*(k-2)+1}], %[x{2*(k-1)+1}]"]
        gaps[15] += ["s_nop 0", chain(7, 0, 0)[6], "v_cvt_pk_bf16_f32 %[d3], %[x13], %[x15]"]
    elif name in ("dword", "dword_nop", "dword_notrans"):   # one dword at a time, lo / hi chains interleaved: gaps 4q .. 4q+3
        for q in range(4):
            lo, hi = chain(2 * q, q, 0), chain(2 * q + 1, q, 1)
            il = [op for pair in zip(lo, hi) for op in pair]            # A A B B C C D D E E F F G G
            if name == "dword_notrans":
                il = [op.replace("v_exp_f32", "v_mov_b32").replace("v_rcp_f32", "v_mov_b32") for op in il]
            gaps[4 * q] += il[0:4]
            gaps[4 * q + 1] += il[4:8]
            gaps[4 * q + 2] += il[8:12]
            gaps[4 * q + 3] += il[12:14]
            gaps[4 * q + 4] += [f"v_cvt_pk_bf16_f32 %[d{q}], %[x{4*q+1}], %[x{4*q+3}]"]
        if name == "dword_nop":
            for g in range(16):
                gaps[g].append("s_nop 0")
    elif name in ("late12", "late8", "late6"):   # the whole transform behind the LAST 12 / 8 / 6 MFMAs (dword-linear order): the first
        first = {"late12": 4, "late8": 8, "late6": 10}[name]   # MFMAs of the phase then cover the tail of the halo load's latency
        seq = []
        for q in range(4):
            lo, hi = chain(2 * q, q, 0), chain(2 * q + 1, q, 1)
            seq += [op for pair in zip(lo, hi) for op in pair]
            seq += [f"v_cvt_pk_bf16_f32 %[d{q}], %[x{4*q+1}], %[x{4*q+3}]"]
        n = 16 - first
        per = (len(seq) + n - 1) // n
        for i, op in enumerate(seq):
            gaps[min(first + i // per, 16)].append(op)
    elif name in ("lds8", "lds8_vm"):            # the one-wave-per-SIMD step of conv_v9: dword-linear transform + 8 fragment reads
        for q in range(4):                       # (+ 1 buffer load and 1 LDS store of a staged piece) behind 16 MFMAs
            lo, hi = chain(2 * q, q, 0), chain(2 * q + 1, q, 1)
            il = [op for pair in zip(lo, hi) for op in pair]
            gaps[4 * q] += il[0:4]; gaps[4 * q + 1] += il[4:8]; gaps[4 * q + 2] += il[8:12]; gaps[4 * q + 3] += il[12:14]
            gaps[4 * q + 4] += [f"v_cvt_pk_bf16_f32 %[d{q}], %[x{4*q+1}], %[x{4*q+3}]"]
        for r in range(8):                      # issued early in the phase, waited for at the start of the NEXT one (as the frags of
            gaps[r].append(f"ds_read_b128 %[f{r % 4}], %[la] offset:{r * 5120}")   # the next half-step would be)
        if name == "lds8_vm":
            gaps[9].append("buffer_load_dwordx4 %[f3], %[va], %[rs], 0 offen")
            gaps[12].append("ds_write_b128 %[la], %[f2] offset:40960")
    elif name == "pk":                           # packed f32 ops on the (lo, hi) pair
        for q in range(4):
            X, Y = f"%[p{2*q}]", f"%[p{2*q+1}]"      # 64-bit pairs
            ops = [f"v_lshlrev_b32 %[x{4*q}], 16, %[d{q}]", f"v_and_b32 %[x{4*q+1}], 0xffff0000, %[d{q}]"]
            # pk ops need consecutive register pairs: x{4q},x{4q+1} are passed as one 64-bit operand p{2q}; x{4q+2},x{4q+3} = p{2q+1}
            ops = [f"v_lshlrev_b32 %[x{4*q}], 16, %[d{q}]".replace(f"%[x{4*q}]", X + "_lo"), ]
            gaps[4 * q] += []
        raise SystemExit("pk pattern needs sub-register syntax: not generated")
    elif name == "skew":                         # two dwords in flight, the second one two stages behind: <= 2 transcendentals per gap
        # stage lists per dword: S0 = A A B B, S1 = C C, S2 = D D, S3 = E E, S4 = F F, S5 = G G, S6 = H
        def stages(q):
            lo, hi = chain(2 * q, q, 0), chain(2 * q + 1, q, 1)
            return [[lo[0], hi[0], lo[1], hi[1]], [lo[2], hi[2]], [lo[3], hi[3]], [lo[4], hi[4]], [lo[5], hi[5]], [lo[6], hi[6]],
                    [f"v_cvt_pk_bf16_f32 %[d{q}], %[x{4*q+1}], %[x{4*q+3}]"]]
        start = [0, 3, 7, 10]                    # gap of S0 of dword q
        for q in range(4):
            for j, st in enumerate(stages(q)):
                gaps[min(start[q] + j, 16)] += st
    else:
        raise SystemExit(name)
    lines = []
    if name.startswith("lds8"):
        lines.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    for g in range(16):
        lines.append(MFMA.format(i=g % 8))
        lines += gaps[g]
    lines += gaps[16]
    return lines


PATTERNS = ["bare", "dword", "late12"]
out = []
out.append('''// GENERATED by gen_mfma_filler.py - do not edit.  hipcc --offload-arch=gfx950 -O3 mfma_filler.hip -o mfma_filler
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define OPS "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7), \\
    [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]), [x6] "+v"(x[6]), [x7] "+v"(x[7]), \\
    [x8] "+v"(x[8]), [x9] "+v"(x[9]), [x10] "+v"(x[10]), [x11] "+v"(x[11]), [x12] "+v"(x[12]), [x13] "+v"(x[13]), [x14] "+v"(x[14]), [x15] "+v"(x[15]), \\
    [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3])
''')
for p in PATTERNS:
    body = "\\n\\t".join(pattern(p))
    # named accumulators: acc0..acc7 operands are positional 0..7 -> use names
    body = body
    out.append(f'''
template <int LDSW> __global__ __launch_bounds__(512) void k_{p}(unsigned long long* out, float* sink, int iters, void* gsrc) {{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc0 = {{}}, acc1 = {{}}, acc2 = {{}}, acc3 = {{}}, acc4 = {{}}, acc5 = {{}}, acc6 = {{}}, acc7 = {{}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {{ a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane ^ i)); }}
    float x[16]; unsigned d[4];
    for (int i = 0; i < 16; ++i) x[i] = 0.01f * (lane + i);
    for (int i = 0; i < 4; ++i) d[i] = 0x3f803f80u + lane * 65537u * (i + 1);
    float c0 = 0.5f, c1 = 0.25f;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 f0 = {{}}, f1 = {{}}, f2 = {{}}, f3 = {{}};
    const unsigned la = lane * 80, va = (threadIdx.x + blockIdx.x * 512) * 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gsrc, 0, 0x7fffffff, 0x00020000);
    reinterpret_cast<float*>(lds)[threadIdx.x] = 1.f;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (LDSW && wave >= 4) {{                  // partner waves (one per SIMD): an LDS phase's traffic per ~500 cycles, no barriers
        float4 s = {{}};                        // LDSW = 1: 12 fragment reads; 2: + one buffer_load_dwordx4; 3: + two; 4: two loads + a 16-byte LDS store
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4p;
        const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(gsrc, 0, 0x7fffffff, 0x00020000);
        for (int it = 0; it < iters; ++it) {{
#pragma unroll
            for (int r = 0; r < 12; ++r) {{ const float4 v = *reinterpret_cast<const float4*>(lds + ((lane * 80 + r * 5120 + it * 16) & 65520)); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }}
            if (LDSW >= 2) {{ const u32x4p v = __builtin_amdgcn_raw_buffer_load_b128(rsp, (threadIdx.x * 16 + it * 8192 + blockIdx.x * 65536) & 0x1ffff0, 0, 0); s.x += __builtin_bit_cast(float, v[0]); }}
            if (LDSW >= 3) {{ const u32x4p v = __builtin_amdgcn_raw_buffer_load_b128(rsp, (threadIdx.x * 16 + it * 8192 + blockIdx.x * 65536 + 1048576) & 0x1ffff0, 0, 0); s.y += __builtin_bit_cast(float, v[1]); }}
            if (LDSW >= 4) *reinterpret_cast<float4*>(lds + 32768 + threadIdx.x * 16) = s;
            __builtin_amdgcn_s_sleep(6);
        }}
        sink[threadIdx.x] = s.x + s.y + s.z + s.w;
        return;
    }}
    if (!LDSW && wave >= 4) return;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {{
        asm volatile("{body}"
                     : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [acc2] "+v"(acc2), [acc3] "+v"(acc3), [acc4] "+v"(acc4), [acc5] "+v"(acc5), [acc6] "+v"(acc6), [acc7] "+v"(acc7),
                       [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]), [x6] "+v"(x[6]), [x7] "+v"(x[7]),
                       [x8] "+v"(x[8]), [x9] "+v"(x[9]), [x10] "+v"(x[10]), [x11] "+v"(x[11]), [x12] "+v"(x[12]), [x13] "+v"(x[13]), [x14] "+v"(x[14]), [x15] "+v"(x[15]),
                       [d0] "+v"(d[0]), [d1] "+v"(d[1]), [d2] "+v"(d[2]), [d3] "+v"(d[3]), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3)
                     : [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [la] "v"(la), [va] "v"(va), [rs] "s"(rs));
    }}
    t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + acc2[i] + acc3[i] + acc4[i] + acc5[i] + acc6[i] + acc7[i] + x[i];
    sink[threadIdx.x] = s + d[0] + d[1] + d[2] + d[3] + f0[0] + f1[0] + f2[0] + f3[0];
    if (lane == 0 && blockIdx.x == 7) out[wave] = t1 - t0;
}}
''')
out.append('''
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096); void* gsrc; hipMalloc(&gsrc, 1 << 26); hipMemset(gsrc, 0, 1 << 26);
    const int iters = 2000;
    unsigned long long h[8];
#define RUN(NAME, L)                                                                                          \\
    { hipMemset(out, 0, 64);                                                                                  \\
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_##NAME<L>), dim3(256), dim3(512), 0, 0, out, sink, iters, gsrc); hipDeviceSynchronize(); \\
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_##NAME<L>), dim3(256), dim3(512), 0, 0, out, sink, iters, gsrc); hipDeviceSynchronize(); \\
      hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);                                                           \\
      printf("%-14s partner-LDS-wave=%d : %7.1f cycles per 16-MFMA phase (waves 0-3: %llu %llu %llu %llu)\\n", #NAME, L, (double)h[0] / iters, h[0], h[1], h[2], h[3]); }
''')
for p in PATTERNS:
    out.append(f"    RUN({p}, 0) RUN({p}, 1) RUN({p}, 2) RUN({p}, 3) RUN({p}, 4)\n")
out.append("    return 0;\n}\n")
open(sys.argv[1] if len(sys.argv) > 1 else "scripts/microbench/mfma_filler.hip", "w").write("".join(out))
