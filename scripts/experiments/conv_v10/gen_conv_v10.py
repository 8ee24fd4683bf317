#!/usr/bin/env python3
"""Generates use_conv_v10.hip from use_conv_v10.hip.in (committed next to this script; `make` does not run it).

conv_v10: conv_v9's asm-pinned MFMA stream in the shape that survives the two-stream schedule - TWO co-resident, NON-persistent 4-wave
workgroups per CU (256 registers per wave: 128 accumulators in AGPRs + 128, <= 80 KB LDS), each walking a short strip of tiles; one
workgroup's prologue, barriers and epilogue run under the other's MFMAs (the steady state of this shape measured
1.40 PFLOP/s on the whole chip, scripts/microbench/gen_v9_steady.py with V9_TWO=1).

  tile      8 x 32 px x 128 output channels, wave w owns rows 2w, 2w+1 (A = weights, B = pixels: conv_v9's layout and epilogue)
  K         16-channel chunks; the chunk's halo (10 x 34 px x 32 B) twice in LDS; the weights as THREE tap-row groups (3 taps x 128 x 32 B
            = 12 KB each, 36 KB in all: the nine-tap double buffer of conv_v9 does not fit twice on a CU): slot r holds tap row r
  chunk     three groups of 24 MFMAs, one s_barrier in front of each.  Group (c, r) multiplies tap row r; behind its barrier slot r-1 is
            free: LDS-DMA of tap row r-1 of chunk c+1 (r = 0: tap row 2 of chunk c itself) - two groups ahead of its first read.
            Fragments are read behind the barrier of their own group (tap 0) and one tap ahead after that; the halo pieces of chunk c+1
            are transformed behind the MFMAs (3 instructions per gap) from ONE register set that is reloaded (chunk c+2) piece by piece.
"""
import os

HB = [0, 11264]
WB = 22528                      # three tap-row slots of 12288 bytes
COEF = WB + 3 * 12288           # [Cin <= 512][2] fp32
BINIT = COEF + 4096             # [128] fp32
TOT = BINIT + 512               # [128][2] u64
LDS = TOT + 2048
DUMMY = 10880

LO = {'bf16': 'v_lshlrev_b32 %[xl], 16, %[d]', 'f16': 'v_cvt_f32_f16 %[xl], %[d]'}
HI = {'bf16': 'v_and_b32 %[xh], 0xffff0000, %[d]', 'f16': 'v_cvt_f32_f16_sdwa %[xh], %[d] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1'}
PK = {'bf16': 'v_cvt_pk_bf16_f32 %[d], %[ul], %[uh]', 'f16': 'v_cvt_pk_f16_f32 %[d], %[ul], %[uh]'}
MF = {'bf16': 'v_mfma_f32_32x32x16_bf16', 'f16': 'v_mfma_f32_32x32x16_f16'}


def slices(ty, act):
    if act:
        return [[LO[ty], HI[ty], 'v_fma_f32 %[ul], %[xl], %[al], %[bl]'],
                ['v_fma_f32 %[uh], %[xh], %[ah], %[bh]', 'v_mul_f32 %[xl], 0xbfb8aa3b, %[ul]', 'v_mul_f32 %[xh], 0xbfb8aa3b, %[uh]'],
                ['v_exp_f32 %[xl], %[xl]', 'v_exp_f32 %[xh], %[xh]', 'v_add_f32 %[xl], 1.0, %[xl]'],
                ['v_add_f32 %[xh], 1.0, %[xh]', 'v_rcp_f32 %[xl], %[xl]', 'v_rcp_f32 %[xh], %[xh]'],
                ['v_mul_f32 %[ul], %[ul], %[xl]', 'v_mul_f32 %[uh], %[uh], %[xh]', PK[ty]]]
    return [[LO[ty], HI[ty]], ['v_fma_f32 %[ul], %[xl], %[al], %[bl]'], ['v_fma_f32 %[uh], %[xh], %[ah], %[bh]'], [PK[ty]], []]


XF_FIRST = 6
# vmcnt allowed at the barrier in front of group r (VMEM operations issued after the DMA that must have landed: see the module text)
BAR_VMCNT = {0: 5, 1: 4, 2: 5}


def chunk(par, ty, act):
    """C++ statements of one chunk whose halo is in buffer `par` (stages chunk c+1 into buffer par^1)."""
    L = []
    mf = MF[ty]
    sl = slices(ty, act)
    mem = [[] for _ in range(72)]

    def wread(t, j):                       # weight fragment of tap t (row t // 3 -> slot, position t % 3), block j; ring slot by tap parity
        return f'wf[{(t + par) % 2}][{j}] = LDSV(wbase + {WB + (t // 3) * 12288 + ((t % 3) * 128 + j * 32) * 32});'

    def xread(r, dx):
        return f'xf[{r * 3 + dx}] = LDSV(xbase + {HB[par] + (r * 34 + dx) * 32});'

    newx = {0: [(0, 0), (1, 0)], 1: [(0, 1), (1, 1)], 2: [(0, 2), (1, 2)], 3: [(2, 0)], 4: [(2, 1)], 5: [(2, 2)], 6: [(3, 0)], 7: [(3, 1)], 8: [(3, 2)]}
    first = {0: [], 1: [], 2: []}          # reads behind the barrier of group r: its first tap
    for r in range(3):
        t = 3 * r
        first[r] = [wread(t, j) for j in range(4)] + [xread(rr, dx) for (rr, dx) in newx[t]]
        for k in (1, 2):                   # taps 1, 2 of the group are read during the tap before
            tt = t + k
            items = [wread(tt, j) for j in range(4)] + [xread(rr, dx) for (rr, dx) in newx[tt]]
            g0 = (tt - 1) * 8
            for n, it in enumerate(items):
                mem[g0 + n].append(it)
    # LDS-DMA of the tap row whose slot the barrier in front of this group has freed: 3 wave-level instructions of 1 KB per wave
    for r in range(3):
        for k in range(3):
            src_row = (r - 1) % 3
            so = 'w0_soff' if r == 0 else 'w1_soff'            # r = 0: tap row 2 of THIS chunk; r = 1, 2: tap rows 0, 1 of chunk c+1
            mem[24 * r + 1 + k].append(f'{{ V10_DMA({so} + {src_row * 3 + k}u * tap_b, {WB + src_row * 12288 + k * 4096}); }}')
    # coefficient rows of the chunk being staged
    for q in range(4):
        mem[0].append(f'cf[{q}] = LDSF(coef1 + {q * 16});')
    for g in range(72):
        t, k8 = g // 8, g % 8
        i, j = k8 // 4, k8 % 4
        if g % 24 == 0:
            L.append(f'V10_BAR({BAR_VMCNT[g // 24]})')
            L += first[g // 24]
        ops = f'[acc] "+a"(acc[{i}][{j}])'
        ins = f'[w] "v"(wf[{(t + par) % 2}][{j}]), [x] "v"(xf[{(i + t // 3) * 3 + t % 3}])'
        stmt = None
        n = g - XF_FIRST
        if 0 <= n < 60:
            p, q, s = n // 20, (n % 20) // 5, n % 5
            if sl[s]:
                body = '\\n\\t'.join([f'{mf} %[acc], %[w], %[x], %[acc]'] + sl[s])
                fops = ops + f', [d] "+v"(hsd[{p}][{q}]), [xl] "+v"(xl), [xh] "+v"(xh), [ul] "+v"(ul), [uh] "+v"(uh)'
                fins = ins + f', [al] "v"(cf[{q}][0]), [bl] "v"(cf[{q}][1]), [ah] "v"(cf[{q}][2]), [bh] "v"(cf[{q}][3])'
                stmt = f'asm volatile("{body}" : {fops} : {fins});'
            if n % 20 == 19:               # piece p transformed: store it (chunk c+1's halo), reload the registers with chunk c+2's piece
                mem[g].append(f'{{ u32x4 t_ = {{hsd[{p}][0], hsd[{p}][1], hsd[{p}][2], hsd[{p}][3]}}; if (!((pv >> {p}) & 1u)) t_ = u32x4{{0u, 0u, 0u, 0u}}; LDSST(hdst[{p}] + {HB[par ^ 1]}, t_); }}')
                mem[g].append(f'{{ const u32x4 t_ = V10_HLOAD({p}); hsd[{p}][0] = t_[0]; hsd[{p}][1] = t_[1]; hsd[{p}][2] = t_[2]; hsd[{p}][3] = t_[3]; }}')
        if stmt is None:
            stmt = f'asm volatile("{mf} %[acc], %[w], %[x], %[acc]" : {ops} : {ins});'
        L.append(stmt)
        L += mem[g]
    return L


def emit(ty, act):
    suf = f'{ty.upper()}_{"ACT" if act else "LIN"}'
    out = []
    for par in (0, 1):
        body = chunk(par, ty, act)
        out.append(f'#define V10_CHUNK_{par}_{suf} \\')
        out += ['    ' + l + ' \\' for l in body]
        out.append('')
    return '\n'.join(out)


here = os.path.dirname(os.path.abspath(__file__))
tmpl = open(os.path.join(here, 'use_conv_v10.hip.in')).read()
gen = '\n'.join(emit(ty, act) for ty in ('bf16', 'f16') for act in (True, False))
consts = '\n'.join(f'constexpr int V10_{k} = {v};' for k, v in
                   [('HB0', HB[0]), ('HB1', HB[1]), ('WB', WB), ('COEF', COEF), ('BINIT', BINIT), ('TOT', TOT), ('LDS', LDS), ('DUMMY', DUMMY)])
open(os.path.join(here, 'use_conv_v10.hip'), 'w').write(
    '// GENERATED by gen_conv_v10.py from use_conv_v10.hip.in - edit those, then run the script.\n' +
    tmpl.replace('//@@CONSTS@@', consts).replace('//@@CHUNKS@@', gen))
