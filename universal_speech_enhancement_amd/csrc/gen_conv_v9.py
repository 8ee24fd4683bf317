#!/usr/bin/env python3
"""Generates use_conv_v9.hip (committed next to this script; `make` does not run it).

conv_v9: the 3x3 convolution of the large maps as ONE wave per SIMD (4 waves, 512 registers: 128 accumulators in AGPRs), persistent
over the tiles of one item.  Why a generator: hipcc places side-effect-free MFMAs and VALU wherever it likes
(round 4, use_device.h), so the 72-MFMA "window" is emitted as one asm statement per MFMA with its share of the GroupNorm + SiLU
transform behind it (3 VALU per gap), and the memory operations of the same gap as compiler-visible volatile accesses in between
(hipcc's waitcnt insertion counts them).  The schedule is the one sized with scripts/microbench/gen_v9_steady.py.

  tile      8 x 32 px x 128 output channels; wave w owns tile rows 2w, 2w+1: acc[i][j] = 32 output channels (j) x 32 pixels of row i
            (operands swapped against conv_v4: A = weights, B = pixels, so a lane ends up with 4 consecutive channels of its pixel)
  K         16-channel chunks; a chunk's halo (10 x 34 px x 32 B) and its nine weight slabs (9 x 128 x 32 B) are in LDS, twice
  window    s_barrier | tap 8 of chunk c-1 (buffers b) | taps 0..7 of chunk c (buffers b^1)   - the fragments of tap 8 were read
            before the barrier, so buffers b are free behind it: weights of chunk c+1 by LDS-DMA, the halo pieces of chunk c+1
            (loaded one window earlier) transformed behind the MFMAs and stored, the raw pieces of chunk c+2 loaded
  tile end  the window that starts a tile runs the previous tile's epilogue between its tap 8 and its tap 0 (conv_v7's form: one
            v_permlane32_swap per register pair -> 16-byte pieces of the NHWC output, GroupNorm partial sums through the matrix pipe)
"""
import os

HB = [0, 11264]
WB = [22528, 22528 + 36864]
COEF = 96256            # [Cin <= 512][2] fp32
BINIT = COEF + 4096     # [Cout <= 512] fp32
TOT = BINIT + 2048      # [Cout <= 512][2] u64
LDS = TOT + 8192
DUMMY = 10880           # spare bytes behind a halo buffer

LO = {'bf16': 'v_lshlrev_b32 %[xl], 16, %[d]', 'f16': 'v_cvt_f32_f16 %[xl], %[d]'}
HI = {'bf16': 'v_and_b32 %[xh], 0xffff0000, %[d]', 'f16': 'v_cvt_f32_f16_sdwa %[xh], %[d] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1'}
PK = {'bf16': 'v_cvt_pk_bf16_f32 %[d], %[ul], %[uh]', 'f16': 'v_cvt_pk_f16_f32 %[d], %[ul], %[uh]'}
MF = {'bf16': 'v_mfma_f32_32x32x16_bf16', 'f16': 'v_mfma_f32_32x32x16_f16'}


def slices(ty, act):
    """the per-dword transform as five slices of <= 3 instructions (a transcendental's result is read two instructions later)"""
    if act:
        return [[LO[ty], HI[ty], 'v_fma_f32 %[ul], %[xl], %[al], %[bl]'],
                ['v_fma_f32 %[uh], %[xh], %[ah], %[bh]', 'v_mul_f32 %[xl], 0xbfb8aa3b, %[ul]', 'v_mul_f32 %[xh], 0xbfb8aa3b, %[uh]'],
                ['v_exp_f32 %[xl], %[xl]', 'v_exp_f32 %[xh], %[xh]', 'v_add_f32 %[xl], 1.0, %[xl]'],
                ['v_add_f32 %[xh], 1.0, %[xh]', 'v_rcp_f32 %[xl], %[xl]', 'v_rcp_f32 %[xh], %[xh]'],
                ['v_mul_f32 %[ul], %[ul], %[xl]', 'v_mul_f32 %[uh], %[uh], %[xh]', PK[ty]]]
    return [[LO[ty], HI[ty]], ['v_fma_f32 %[ul], %[xl], %[al], %[bl]'], ['v_fma_f32 %[uh], %[xh], %[ah], %[bh]'], [PK[ty]], []]


def xidx(i, t):
    return (i + t // 3) * 3 + t % 3


XF_FIRST = 6
DMA_GAPS = [4 + 5 * q for q in range(9)]
HLOAD_GAPS = [1, 2, 3]
NEWX = {0: [(0, 0), (1, 0)], 1: [(0, 1), (1, 1)], 2: [(0, 2), (1, 2)], 3: [(2, 0)], 4: [(2, 1)], 5: [(2, 2)], 6: [(3, 0)], 7: [(3, 1)], 8: [(3, 2)]}


def window(b, ty, act, switch):
    """C++ statements of one window on buffers b (tap 8 of the previous chunk reads buffers b, taps 0..7 of the current one b^1)."""
    L = []
    nb = b ^ 1
    mf = MF[ty]
    sl = slices(ty, act)
    seq = [(8, b)] + [(t, nb) for t in range(8)]
    mem = [[] for _ in range(72)]
    g_next = 0
    for pos in list(range(1, 9)) + [9]:          # ring slot pos % 3 is free once position pos - 3 is done: not before gap (pos - 2) * 8
        t, buf = (seq[pos] if pos < 9 else (8, nb))
        g_next = max(g_next, (pos - 2) * 8)
        for j in range(4):
            mem[g_next].append(f'wf[{pos % 3}][{j}] = LDSV(wbase + {WB[buf] + (t * 128 + j * 32) * 32});'); g_next += 1
        for (r, dx) in NEWX[t]:
            mem[g_next].append(f'xf[{r * 3 + dx}] = LDSV(xbase + {HB[buf] + (r * 34 + dx) * 32});'); g_next += 1
    for q, g in enumerate(DMA_GAPS):
        mem[g].append(f'{{ V9_DMA(w1_soff + {q}u * tap_b, {WB[b] + q * 4096}); }}')
    for k, g in enumerate(HLOAD_GAPS):
        mem[g].append(f'hs[{nb}][{k}] = V9_HLOAD({k});')
    # coefficient rows of the chunk being staged (chunk c+1): 8 channels x (a, b) = 64 bytes, read at the top of the window
    for q in range(4):
        mem[0].append(f'cf[{q}] = LDSF(coef1 + {q * 16});')
    for g in range(72):
        gi, k8 = g // 8, g % 8
        t = seq[gi][0]
        i, j = k8 // 4, k8 % 4
        ops = f'[acc] "+a"(acc[{i}][{j}])'
        ins = f'[w] "v"(wf[{gi % 3}][{j}]), [x] "v"(xf[{xidx(i, t)}])'
        stmt = None
        n = g - XF_FIRST
        if 0 <= n < 60:
            p, q, s = n // 20, (n % 20) // 5, n % 5
            if s == 0:
                L.append(f'hcd[{q}] = hs[{b}][{p}][{q}];' if q else f'{{ hcd[0] = hs[{b}][{p}][0]; hcd[1] = hs[{b}][{p}][1]; hcd[2] = hs[{b}][{p}][2]; hcd[3] = hs[{b}][{p}][3]; }}')
            if sl[s]:
                body = '\\n\\t'.join([f'{mf} %[acc], %[w], %[x], %[acc]'] + sl[s])
                fops = ops + f', [d] "+v"(hcd[{q}]), [xl] "+v"(xl), [xh] "+v"(xh), [ul] "+v"(ul), [uh] "+v"(uh)'
                fins = ins + f', [al] "v"(cf[{q}][0]), [bl] "v"(cf[{q}][1]), [ah] "v"(cf[{q}][2]), [bh] "v"(cf[{q}][3])'
                stmt = f'asm volatile("{body}" : {fops} : {fins});'
            if n % 20 == 19:
                mem[g].append(f'{{ u32x4 t_ = {{hcd[0], hcd[1], hcd[2], hcd[3]}}; if (!((pv1 >> {p}) & 1u)) t_ = u32x4{{0u, 0u, 0u, 0u}}; LDSST(hdst[{p}] + {HB[b]}, t_); }}')
        if stmt is None:
            stmt = f'asm volatile("{mf} %[acc], %[w], %[x], %[acc]" : {ops} : {ins});'
        L.append(stmt)
        if g == 7 and switch:
            L.append('V9_TILE_SWITCH()')
        L += mem[g]
    return L


def emit(ty, act):
    suf = f'{ty.upper()}_{"ACT" if act else "LIN"}'
    out = []
    for name, b, sw in (('0S', 0, True), ('0', 0, False), ('1', 1, False)):
        body = window(b, ty, act, sw)
        out.append(f'#define V9_WIN_{name}_{suf} \\')
        out += ['    ' + l + ' \\' for l in body]
        out.append('')
    return '\n'.join(out)


here = os.path.dirname(os.path.abspath(__file__))
tmpl = open(os.path.join(here, 'use_conv_v9.hip.in')).read()
gen = '\n'.join(emit(ty, act) for ty in ('bf16', 'f16') for act in (True, False))
consts = '\n'.join(f'constexpr int V9_{k} = {v};' for k, v in
                   [('HB0', HB[0]), ('HB1', HB[1]), ('WB0', WB[0]), ('WB1', WB[1]), ('COEF', COEF), ('BINIT', BINIT), ('TOT', TOT), ('LDS', LDS), ('DUMMY', DUMMY)])
open(os.path.join(here, 'use_conv_v9.hip'), 'w').write(
    '// GENERATED by gen_conv_v9.py from use_conv_v9.hip.in - edit those, then run the script.\n' +
    tmpl.replace('//@@CONSTS@@', consts).replace('//@@WINDOWS@@', gen))
