// Host side of libuse_hip.so: NCSN++ architecture walk (reference backbones/ncsnpp.py:116-316, 324-501),
// weight packing, workspace planning, the score evaluation, the predictor-corrector loop
// (reference sampling/__init__.py:59-71) with hipGraph capture, and the C ABI of include/use_hip.h.
#include "../../include/use_hip.h"
#include "use_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

using namespace use;

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
// error reporting for the other translation units of the library (use_io.cpp)
int use_set_error(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return fail(USE_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static inline uint16_t f32_to_bf16(float f) {   // round to nearest even
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline uint16_t f32_to_f16(float f) {    // round to nearest even, overflow -> inf, subnormals kept
    uint32_t u; memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                 // NaN
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                // >= 65520 -> inf
    if (u < 0x38800000u) {                                                  // subnormal half (or zero)
        if (u < 0x33000000u) return (uint16_t)sign;
        const int shift = 126 - (int)(u >> 23);                             // 14..24
        uint32_t m = (u & 0x7fffffu) | 0x800000u;
        const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1);
        m >>= shift;
        if (rest > half || (rest == half && (m & 1))) ++m;
        return (uint16_t)(sign | m);
    }
    uint32_t v = u - 0x38000000u;                                           // rebias exponent
    const uint32_t rest = v & 0x1fffu;
    v >>= 13;
    if (rest > 0x1000u || (rest == 0x1000u && (v & 1))) ++v;
    return (uint16_t)(sign | v);
}

// ---------------------------------------------------------------------------------------------------------
// architecture description
// ---------------------------------------------------------------------------------------------------------
struct ConvW { std::string wname, bname; int cin = 0, cout = 0, cout_pad = 0, ntaps = 1, w_dtype = DT_F32;
               bool nin = false; size_t w_off = 0, b_off = 0;
               int cin_src = 0, cout_src = 0;                 // extents of the host tensor when it is zero-padded to cin / cout
               size_t wb_off = 0; bool has_wb = false;        // slab-major copy for conv_v4_kernel (see pack_conv)
               bool split_in = false; };                      // wb = the bf16 hi / lo split copy of the input convolution (pack_conv_in_split)
struct GNW { std::string prefix; int C = 0; size_t g_off = 0, b_off = 0; };
struct ResW { int idx = 0, in_ch = 0, out_ch = 0; bool up = false, down = false, has_c2 = false;
              GNW gn0, gn1; ConvW c0, c1, c2; int dense_row0 = 0; size_t b12_off = 0; };   // b12 = Conv_1.bias + Conv_2.bias
struct CombineW { int idx = 0, C = 0; size_t w_off = 0, b_off = 0; };      // conv1x1 4->C, fp32 [C][4]
struct AttnW { int idx = 0, C = 0; GNW gn; ConvW q, k, v, o; };
struct PyrW { GNW gn; ConvW conv; };

struct Expected { std::string name; std::vector<int64_t> shape; };

constexpr int MAX_SUB = 8;                       // sub-batches of the pipelined evaluation (run_score)

struct Act { void* p = nullptr; int C = 0, H = 0, W = 0, dtype = DT_F32; long long* stats = nullptr;     // stats: [B][C][2] fixed-point totals
             long long* part = nullptr; int ntiles = 0; };   // or (large maps, conv_v4 producers) [B][ntiles][C][2] per-workgroup partial totals (ConvArgs::stats_part)

struct Arena {
    char* base = nullptr; size_t cap = 0, off = 0, peak = 0;
    bool overflow = false;           // sticky: an allocation did not fit (cannot happen for the plan the arena was sized for)
    bool* dry_on_overflow = nullptr; // ... and then the evaluation switches itself to the bookkeeping-only mode: no further launch
    void reset() { off = 0; }
    void* alloc(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? (void*)(base + off) : nullptr;
        off += bytes;
        if (off > peak) peak = off;
        if (base && off > cap) {     // never a write past the allocation and never abort() behind a C ABI: the entry point that
            overflow = true;         // ran this evaluation reports USE_E_STATE (see eval_status); the tensor that did not fit is not touched
            if (dry_on_overflow) *dry_on_overflow = true;
            p = base;
        }
        return p;
    }
};

struct use_handle {
    use_config cfg{};
    int device = 0;
    int act_dtype = DT_F32;
    // architecture
    std::vector<Expected> expected;
    std::unordered_map<std::string, size_t> expected_index;
    ConvW conv_in;
    std::vector<ResW> res;            // in forward order
    std::vector<CombineW> combines;   // one per down level
    AttnW attn;
    std::vector<PyrW> pyrs;           // in forward order (coarsest level first)
    int dense_rows = 0;
    size_t gfp_off = 0, l1w_off = 0, l1b_off = 0, l2w_off = 0, l2b_off = 0, dense_w_off = 0, dense_b_off = 0,
           outw_off = 0, outb_off = 0;
    size_t blob_bytes = 0;
    // weights
    std::unordered_map<std::string, std::vector<float>> host_w;
    char* blob = nullptr; bool weights_ready = false;
    // plan
    int B = 0, T = 0;
    Arena arena;
    char* persist = nullptr; size_t persist_bytes = 0;
    size_t arena_alloc = 0, persist_alloc = 0;   // sizes of the two device allocations: kept at their high-water marks across re-plans
    size_t arena_plan_bytes = 0;                 // what the current plan uses of the first (all sub-batch arenas + GroupNorm totals)
    float *x4 = nullptr, *silu_temb = nullptr, *tembias = nullptr, *t_dev = nullptr;
    float2 *Y = nullptr, *X = nullptr, *Xmean = nullptr, *score = nullptr, *xin = nullptr;
    float2 *cond_buf = nullptr, *cond2_buf = nullptr;   // cond2: the second conditioning spectrogram of condition="both" (6 input channels)
    int pcp = 4;                                 // input / pyramid channels as stored: the reference's 2, 4 or 6 zero-padded to 4 or 8
    float2 *Cond = nullptr;                      // score conditioning of the sampler: == Y unless use_sample_cond gave another
    float *lang_partial = nullptr, *lang_step = nullptr;
    unsigned long long* rng_state = nullptr;
    int lang_blocks = 0;
    // sampler
    use_sampler_config sc{};
    bool sampler_set = false;
    std::vector<float> timesteps;
    float* ts_dev = nullptr; float* temb_table = nullptr; float* silu_table = nullptr;
    float2* noise_copy = nullptr; size_t noise_copy_bytes = 0;
    hipStream_t cap_stream = nullptr;
    // sub-batch pipelining (see run_score): the batch is evaluated as up to MAX_SUB sub-batches (default three) on as many streams, the second started when
    // the first reaches its small feature maps, so that one half's latency-bound kernels hide behind the other's large ones
    int nsub = 1, sub_B[MAX_SUB] = {};  // sub-batch sizes (nsub = 1: not split)
    Arena sub_arena[MAX_SUB];                    // workspaces of sub-batches 1.. (inside the same allocation as `arena`)
    Arena st_arena[MAX_SUB];                     // per sub-batch: the GroupNorm totals of all its activations, contiguous (one memset per evaluation)
    hipStream_t aux_stream[MAX_SUB] = {};
    hipEvent_t ev_fork = nullptr, ev_stagger[MAX_SUB] = {},
               ev_join[MAX_SUB] = {};
    int debug_B = 0;
    std::vector<hipGraphExec_t> graph_exec[2];           // [0]: device RNG, [1]: injected noise; one graph per segment of steps
    hipGraphExec_t score_graph = nullptr;
    // plan cache: a predict run over files of different lengths alternates between a handful of (B, T') shapes; the plans of the
    // most recently used ones are parked here - workspace, state buffers, time-embedding tables and the captured graphs with the
    // addresses they hold - so that a shape that returns is neither re-planned nor re-captured.  use_set_option("plan_cache", k).
    struct PlanState;
    std::vector<PlanState*> plan_cache;          // most recently used last
    long long opt_gen_at_plan = -1;              // g_opt_gen when the current plan was built
    long long n_graph_captures = 0, n_plans_built = 0, n_plan_cache_hits = 0;
    // scratch for the stand-alone use_sde_* entry points (independent of weights / plan)
    char* sde_buf = nullptr; unsigned long long* sde_rng = nullptr; float* sde_step = nullptr; float* sde_partial = nullptr;
    static constexpr int SDE_MAX_B = 1024, SDE_BLOCKS = 128;
    // per-launch HIP-event profiling of the dominant conv kernel (use_profile_score)
    bool profile = false, profile_all = false; std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events; std::vector<double> prof_flops, prof_bytes; std::vector<char> prof_main; std::vector<std::pair<int, int>> prof_hw;
    // ... and of the HBM-bound kernels around it (FIR resampling, pyramid heads, input convolution): name, map, algorithmic bytes
    struct AuxProf { std::string name; int H = 0, W = 0; double bytes = 0.0; hipEvent_t e0 = nullptr, e1 = nullptr; double ms = 0.0; double flops = 0.0; };
    std::vector<AuxProf> prof_aux;
    std::vector<std::string> prof_desc;
    // introspection
    bool dry = false;
    std::map<std::string, Act> debug;
    double flops = 0.0;
};

static void add_expected(use_handle* h, const std::string& name, std::vector<int64_t> shape) {
    h->expected_index[name] = h->expected.size();
    h->expected.push_back({name, std::move(shape)});
}

static ConvW make_conv(use_handle* h, const std::string& prefix, int cin, int cout, int ntaps, int w_dtype,
                       int cin_src = 0, int cout_src = 0) {
    ConvW c; c.wname = prefix + ".weight"; c.bname = prefix + ".bias"; c.cin = cin; c.cout = cout; c.ntaps = ntaps;
    c.w_dtype = w_dtype;
    c.cin_src = cin_src ? cin_src : cin; c.cout_src = cout_src ? cout_src : cout;
    c.cout_pad = cout <= 32 ? 32 : (cout + 127) / 128 * 128;
    const int k = ntaps == 9 ? 3 : 1;
    add_expected(h, c.wname, {c.cout_src, c.cin_src, k, k});
    add_expected(h, c.bname, {c.cout_src});
    return c;
}
static ConvW make_nin(use_handle* h, const std::string& prefix, int C, int w_dtype) {
    ConvW c; c.wname = prefix + ".W"; c.bname = prefix + ".b"; c.cin = C; c.cout = C; c.ntaps = 1; c.w_dtype = w_dtype;
    c.nin = true; c.cout_pad = (C + 127) / 128 * 128; c.cin_src = C; c.cout_src = C;
    add_expected(h, c.wname, {C, C});
    add_expected(h, c.bname, {C});
    return c;
}
static GNW make_gn(use_handle* h, const std::string& prefix, int C) {
    GNW g; g.prefix = prefix; g.C = C;
    add_expected(h, prefix + ".weight", {C});
    add_expected(h, prefix + ".bias", {C});
    return g;
}

static ResW make_res(use_handle* h, int idx, int in_ch, int out_ch, bool up, bool down) {
    const std::string p = "all_modules." + std::to_string(idx);
    const int dt = h->act_dtype;
    ResW r; r.idx = idx; r.in_ch = in_ch; r.out_ch = out_ch; r.up = up; r.down = down;
    r.gn0 = make_gn(h, p + ".GroupNorm_0", in_ch);
    r.c0 = make_conv(h, p + ".Conv_0", in_ch, out_ch, 9, dt);
    add_expected(h, p + ".Dense_0.weight", {out_ch, 4 * h->cfg.nf});
    add_expected(h, p + ".Dense_0.bias", {out_ch});
    r.dense_row0 = h->dense_rows; h->dense_rows += out_ch;
    r.gn1 = make_gn(h, p + ".GroupNorm_1", out_ch);
    r.c1 = make_conv(h, p + ".Conv_1", out_ch, out_ch, 9, dt);
    r.has_c2 = (in_ch != out_ch) || up || down;
    if (r.has_c2) r.c2 = make_conv(h, p + ".Conv_2", in_ch, out_ch, 1, dt);
    return r;
}

// Mirrors the module construction order of reference ncsnpp.py:181-316.
static int build_arch(use_handle* h) {
    const use_config& c = h->cfg;
    const int nf = c.nf, L = c.n_levels, nrb = c.num_res_blocks, dt = h->act_dtype;
    // pc real input / pyramid channels in the reference tensors; stored zero-padded to pcp = 4 (8 for the 6 channels of
    // condition="both", model_wrapper.py:43-46) everywhere in the engine
    const int pc = c.input_channels ? c.input_channels : 4;
    const int pcp = pc > 4 ? 8 : 4;
    h->pcp = pcp;
    add_expected(h, "output_layer.weight", {2, pc, 1, 1});
    add_expected(h, "output_layer.bias", {2});
    int m = 0;
    add_expected(h, "all_modules.0.W", {nf}); m++;          // the Fourier projection exists even when unconditional
    if (!c.unconditional) {
        add_expected(h, "all_modules.1.weight", {4 * nf, 2 * nf}); add_expected(h, "all_modules.1.bias", {4 * nf}); m++;
        add_expected(h, "all_modules.2.weight", {4 * nf, 4 * nf}); add_expected(h, "all_modules.2.bias", {4 * nf}); m++;
    }
    h->conv_in = make_conv(h, "all_modules." + std::to_string(m), pcp, nf, 9, DT_F32, pc, 0); m++;   // fp32 input always
    std::vector<int> hs_c{nf};
    int in_ch = nf;
    for (int lvl = 0; lvl < L; ++lvl) {
        for (int k = 0; k < nrb; ++k) {
            const int out_ch = nf * c.ch_mult[lvl];
            h->res.push_back(make_res(h, m++, in_ch, out_ch, false, false));
            in_ch = out_ch; hs_c.push_back(in_ch);
        }
        if (lvl != L - 1) {
            h->res.push_back(make_res(h, m++, in_ch, in_ch, false, true));
            CombineW cb; cb.idx = m; cb.C = in_ch;
            add_expected(h, "all_modules." + std::to_string(m) + ".Conv_0.weight", {in_ch, pc, 1, 1});
            add_expected(h, "all_modules." + std::to_string(m) + ".Conv_0.bias", {in_ch});
            h->combines.push_back(cb); m++;
            hs_c.push_back(in_ch);
        }
    }
    in_ch = hs_c.back();
    h->res.push_back(make_res(h, m++, in_ch, in_ch, false, false));
    {
        const std::string p = "all_modules." + std::to_string(m);
        h->attn.idx = m; h->attn.C = in_ch;
        h->attn.gn = make_gn(h, p + ".GroupNorm_0", in_ch);
        h->attn.q = make_nin(h, p + ".NIN_0", in_ch, dt);
        h->attn.k = make_nin(h, p + ".NIN_1", in_ch, dt);
        h->attn.v = make_nin(h, p + ".NIN_2", in_ch, dt);
        h->attn.o = make_nin(h, p + ".NIN_3", in_ch, dt);
        m++;
    }
    h->res.push_back(make_res(h, m++, in_ch, in_ch, false, false));
    for (int lvl = L - 1; lvl >= 0; --lvl) {
        for (int k = 0; k < nrb + 1; ++k) {
            const int out_ch = nf * c.ch_mult[lvl];
            const int skip = hs_c.back(); hs_c.pop_back();
            h->res.push_back(make_res(h, m++, in_ch + skip, out_ch, false, false));
            in_ch = out_ch;
        }
        PyrW pw;
        pw.gn = make_gn(h, "all_modules." + std::to_string(m), in_ch); m++;
        pw.conv = make_conv(h, "all_modules." + std::to_string(m), in_ch, pcp, 9, dt, 0, pc); m++;
        h->pyrs.push_back(pw);
        if (lvl != 0) h->res.push_back(make_res(h, m++, in_ch, in_ch, true, false));
    }
    if (!hs_c.empty()) return fail(USE_E_INVALID, "internal: skip stack not empty");

    // ---- device blob layout ----
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) & ~(size_t)255; size_t o = off; off += bytes; return o; };
    auto lay_conv = [&](ConvW& w) {
        w.w_off = take((size_t)w.ntaps * w.cout_pad * w.cin * dtype_size(w.w_dtype));
        w.b_off = take((size_t)w.cout * 4);
        // 3x3 convolutions and the res-block shortcuts fused into them get a second, slab-major copy
        w.has_wb = !w.nin && w.cout_pad % 128 == 0 && w.cin % conv_v4_chunk(w.w_dtype) == 0 && (w.ntaps == 9 || w.ntaps == 1);
        if (w.has_wb) w.wb_off = take((size_t)w.ntaps * w.cout_pad * w.cin * dtype_size(w.w_dtype));
    };
    // the input convolution in the 16-bit modes: split-bf16 copy [cout_pad][CONV_IN_SPLIT_K] for conv_in_split_kernel
    auto lay_conv_in = [&](ConvW& w) {
        lay_conv(w);
        if (dt != DT_F32 && w.cin == 4 && w.ntaps == 9 && w.cout_pad % 128 == 0) {
            w.split_in = true; w.has_wb = true;
            w.wb_off = take((size_t)w.cout_pad * CONV_IN_SPLIT_K * 2);
        }
    };
    auto lay_gn = [&](GNW& g) { g.g_off = take((size_t)g.C * 4); g.b_off = take((size_t)g.C * 4); };
    h->outw_off = take((size_t)2 * pcp * 4); h->outb_off = take(2 * 4);
    h->gfp_off = take((size_t)nf * 4);
    h->l1w_off = take((size_t)4 * nf * 2 * nf * 4); h->l1b_off = take((size_t)4 * nf * 4);
    h->l2w_off = take((size_t)4 * nf * 4 * nf * 4); h->l2b_off = take((size_t)4 * nf * 4);
    h->dense_w_off = take((size_t)h->dense_rows * 4 * nf * 4); h->dense_b_off = take((size_t)h->dense_rows * 4);
    lay_conv_in(h->conv_in);
    for (auto& r : h->res) {
        lay_gn(r.gn0); lay_conv(r.c0); lay_gn(r.gn1); lay_conv(r.c1);
        if (r.has_c2) { lay_conv(r.c2); r.b12_off = take((size_t)r.out_ch * 4); }
    }
    for (auto& cb : h->combines) { cb.w_off = take((size_t)cb.C * pcp * 4); cb.b_off = take((size_t)cb.C * 4); }
    lay_gn(h->attn.gn); lay_conv(h->attn.q); lay_conv(h->attn.k); lay_conv(h->attn.v); lay_conv(h->attn.o);
    for (auto& p : h->pyrs) { lay_gn(p.gn); lay_conv(p.conv); }
    h->blob_bytes = (off + 255) & ~(size_t)255;
    return USE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------------------
// host float weights -> device layouts: dst [cout_pad][tap][cin] (see use_kernels.h) and, when dstb != null, the slab-major
// copy [tap][chunk][cout_pad][ck] with piece-swizzled 64-byte rows (ConvArgs::wb)
static void pack_conv_raw(const float* src, const ConvW& w, char* dst, char* dstb) {
    const size_t es = dtype_size(w.w_dtype);
    auto put = [&](char* base, size_t o, float v) {
        if (w.w_dtype == DT_F32) ((float*)base)[o] = v;
        else ((uint16_t*)base)[o] = w.w_dtype == DT_F16 ? f32_to_f16(v) : f32_to_bf16(v);
    };
    memset(dst, 0, (size_t)w.ntaps * w.cout_pad * w.cin * es);
    for (int tap = 0; tap < w.ntaps; ++tap)
        for (int co = 0; co < w.cout_src; ++co)
            for (int ci = 0; ci < w.cin_src; ++ci) {
                // reference conv weight [cout][cin][kh][kw]; NIN W is [cin][cout] (layers.py:639-650)
                const float v = w.nin ? src[(size_t)ci * w.cout + co] : src[((size_t)co * w.cin_src + ci) * w.ntaps + tap];
                put(dst, ((size_t)co * w.ntaps + tap) * w.cin + ci, v);       // [cout][tap][cin]: see use_kernels.h
            }
    if (dstb) {                                               // [tap][chunk][cout_pad][ck]: one (tap, chunk) slab contiguous
        const int ck = conv_v4_chunk(w.w_dtype), nchunks = w.cin / ck;
        memset(dstb, 0, (size_t)w.ntaps * w.cout_pad * w.cin * es);
        for (int tap = 0; tap < w.ntaps; ++tap)
            for (int co = 0; co < w.cout_src; ++co)
                for (int ci = 0; ci < w.cin_src; ++ci) {
                    const float v = src[((size_t)co * w.cin_src + ci) * w.ntaps + tap];
                    const int vec = 16 / (int)es, e = ci % ck;             // 64-byte rows, 16-byte pieces swizzled by the row
                    put(dstb, (((size_t)tap * nchunks + ci / ck) * w.cout_pad + co) * ck + (((e / vec) ^ ((co >> 2) & 3)) * vec + e % vec), v);
                }
    }
}
// Input convolution, 16-bit modes: w = wh + wl with wh = bf16(w), wl = bf16(w - wh); row n of the copy holds the K = 112 operand
// of conv_in_split_kernel: k = 4 u + ci, unit u = (block, tap): block 0 -> wh (meets xh), block 1 -> wl (meets xh), block 2 -> wh
// (meets xl), unit 27 = zero padding.  x w = xh wh + xh wl + xl wh + O(2^-16 |x w|).
static void pack_conv_in_split(const float* src, const ConvW& w, char* dst) {
    uint16_t* d = (uint16_t*)dst;
    memset(dst, 0, (size_t)w.cout_pad * CONV_IN_SPLIT_K * 2);
    for (int co = 0; co < w.cout_src; ++co)
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < w.cin_src; ++ci) {
                const float v = src[((size_t)co * w.cin_src + ci) * 9 + tap];     // reference conv weight [cout][cin][kh][kw]
                const uint16_t hi = f32_to_bf16(v);
                uint32_t hb = (uint32_t)hi << 16; float hf; memcpy(&hf, &hb, 4);
                const uint16_t lo = f32_to_bf16(v - hf);
                uint16_t* row = d + (size_t)co * CONV_IN_SPLIT_K;
                row[(0 * 9 + tap) * 4 + ci] = hi; row[(1 * 9 + tap) * 4 + ci] = lo; row[(2 * 9 + tap) * 4 + ci] = hi;
            }
}
static void pack_conv(const use_handle* h, const ConvW& w, char* blob) {
    const std::vector<float>& bias = h->host_w.at(w.bname);
    pack_conv_raw(h->host_w.at(w.wname).data(), w, blob + w.w_off, (w.has_wb && !w.split_in) ? blob + w.wb_off : nullptr);
    if (w.split_in) pack_conv_in_split(h->host_w.at(w.wname).data(), w, blob + w.wb_off);
    memset(blob + w.b_off, 0, (size_t)w.cout * 4);
    memcpy(blob + w.b_off, bias.data(), (size_t)w.cout_src * 4);
}
static void pack_gn(const use_handle* h, const GNW& g, char* blob) {
    memcpy(blob + g.g_off, h->host_w.at(g.prefix + ".weight").data(), (size_t)g.C * 4);
    memcpy(blob + g.b_off, h->host_w.at(g.prefix + ".bias").data(), (size_t)g.C * 4);
}

static int pack_all(use_handle* h, char* blob) {
    for (const auto& e : h->expected)
        if (!h->host_w.count(e.name)) return fail(USE_E_STATE, "weight '%s' was never set", e.name.c_str());
    const int nf = h->cfg.nf;
    auto cp = [&](size_t off, const char* name) {
        const auto& v = h->host_w.at(name); memcpy(blob + off, v.data(), v.size() * 4);
    };
    const int pc = h->cfg.input_channels ? h->cfg.input_channels : 4;
    const int pcp = h->pcp;
    auto cp_rows = [&](size_t off, const std::string& name, int rows) {     // [rows][pc] -> [rows][pcp], zero-padded
        const auto& v = h->host_w.at(name);
        float* d = (float*)(blob + off);
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < pcp; ++k) d[r * pcp + k] = k < pc ? v[(size_t)r * pc + k] : 0.f;
    };
    cp_rows(h->outw_off, "output_layer.weight", 2); cp(h->outb_off, "output_layer.bias");
    cp(h->gfp_off, "all_modules.0.W");
    if (!h->cfg.unconditional) {
        cp(h->l1w_off, "all_modules.1.weight"); cp(h->l1b_off, "all_modules.1.bias");
        cp(h->l2w_off, "all_modules.2.weight"); cp(h->l2b_off, "all_modules.2.bias");
    }
    pack_conv(h, h->conv_in, blob);
    for (const auto& r : h->res) {
        pack_gn(h, r.gn0, blob); pack_conv(h, r.c0, blob); pack_gn(h, r.gn1, blob); pack_conv(h, r.c1, blob);
        const std::string p = "all_modules." + std::to_string(r.idx);
        if (r.has_c2) {
            pack_conv(h, r.c2, blob);
            const auto& b1 = h->host_w.at(p + ".Conv_1.bias"); const auto& b2 = h->host_w.at(p + ".Conv_2.bias");
            float* d = (float*)(blob + r.b12_off);
            for (int i = 0; i < r.out_ch; ++i) d[i] = b1[i] + b2[i];
        }
        memcpy(blob + h->dense_w_off + (size_t)r.dense_row0 * 4 * nf * 4, h->host_w.at(p + ".Dense_0.weight").data(),
               (size_t)r.out_ch * 4 * nf * 4);
        memcpy(blob + h->dense_b_off + (size_t)r.dense_row0 * 4, h->host_w.at(p + ".Dense_0.bias").data(),
               (size_t)r.out_ch * 4);
    }
    for (const auto& cb : h->combines) {
        const std::string p = "all_modules." + std::to_string(cb.idx);
        cp_rows(cb.w_off, p + ".Conv_0.weight", cb.C); cp(cb.b_off, (p + ".Conv_0.bias").c_str());
    }
    pack_gn(h, h->attn.gn, blob);
    pack_conv(h, h->attn.q, blob); pack_conv(h, h->attn.k, blob); pack_conv(h, h->attn.v, blob); pack_conv(h, h->attn.o, blob);
    for (const auto& p : h->pyrs) { pack_gn(h, p.gn, blob); pack_conv(h, p.conv, blob); }
    return USE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// forward pass (one score-network evaluation)
// ---------------------------------------------------------------------------------------------------------
static int g_subbatch_min_items = 2;              // use_set_option("subbatch_min_items", n): no sub-batch smaller than n items
static int g_subbatch = -1;                      // use_set_option("subbatch", n): sub-batches per evaluation (0 / 1: off; -1: by batch size, below)
// GroupNorm finalisation inside the consuming conv (no launch) for maps of at most g_gn_inline pixels per item, a separate
// finalize launch above: on the large maps thousands of workgroups would each redo the finalisation (measured slower), on the
// small, latency-bound maps the saved launch is what counts.  use_set_option("gn_inline", pixels); 0: never inline
static long g_gn_inline = 128L * 160L;
static int g_attn_fused = 1;                     // use_set_option("attn_fused", 0): the unfused attention block (3 NIN, core, NIN_3)
static int g_stats_part = 1;                     // use_set_option("stats_part", 0): GroupNorm totals by atomics on the large maps too
static int g_stagger_level = 2;                  // use_set_option("stagger_level", l): the next sub-batch starts after level l

__global__ __launch_bounds__(256) void zero16_kernel(uint4* __restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}

struct Fwd {
    use_handle* h; hipStream_t s;
    const float* tembias; int temb_bstride;      // [B or 1][dense_rows] (already offset to this sub-batch)
    const float* t; int t_stride;                 // per-item time (stride 0: shared)
    int B = 0;                                    // items of this (sub-)batch
    Arena* arena = nullptr;                       // its activation workspace
    Arena* st_arena = nullptr;                    // its GroupNorm-totals region
    bool primary = true;                          // the first sub-batch owns the FLOP count and the debug tensors
    hipEvent_t ev_stagger = nullptr;              // recorded when the large maps of the down path are done (or null)
    int stagger_level = 2;                        // ... i.e. after this level of the down path
    template <typename T> const T* W(size_t off) const { return (const T*)(h->blob + off); }

    Act new_act(int C, int H, int Wd, int dtype, bool stats) {
        Act a; a.C = C; a.H = H; a.W = Wd; a.dtype = dtype;
        a.p = arena->alloc((size_t)B * H * Wd * C * dtype_size(dtype));
        if (stats) a.stats = (long long*)st_arena->alloc((size_t)B * C * 2 * sizeof(long long));
        return a;
    }

    // use_profile_score: HIP events around one launch of an HBM-bound kernel, with its algorithmic bytes (every operand once)
    template <typename F> void timed_aux(const char* name, int Hm, int Wm, double bytes, F&& launch) {
        if (!h->profile) { launch(); return; }
        use_handle::AuxProf a{name, Hm, Wm, bytes, nullptr, nullptr, 0.0};
        (void)hipEventCreate(&a.e0); (void)hipEventCreate(&a.e1);
        (void)hipEventRecord(a.e0, s);
        launch();
        (void)hipEventRecord(a.e1, s);
        h->prof_aux.push_back(a);
    }

    // coefficient array for the consumers that cannot finalise the GroupNorm themselves (the FIR resampling kernels)
    float* gn_coef(const Act& a, const Act* a2, const GNW& g) {
        const int C = a.C + (a2 ? a2->C : 0);
        float* coef = (float*)arena->alloc((size_t)B * C * 2 * 4);
        if (!h->dry)
            launch_gn_finalize(a.stats, a.C, a2 ? a2->stats : nullptr, a2 ? a2->C : 0, W<float>(g.g_off), W<float>(g.b_off),
                               std::min(C / 4, 32), a.H * a.W, 1e-6f, coef, B, s, a.part, a.ntiles, a2 ? a2->part : nullptr, a2 ? a2->ntiles : 0);
        return coef;
    }

    // gn != null: the input concat(a, a2) goes through this GroupNorm, finalised inside the conv kernel from the producers' totals
    Act conv(const Act& a, const Act* a2, const GNW* gn, int act, const ConvW& w, const float* temb,
             const Act* res, float scale, const float* pyr, const CombineW* cb, int out_dtype, bool stats,
             const Act* sx0 = nullptr, const Act* sx1 = nullptr, const ConvW* w2 = nullptr, size_t bias_off = 0) {
        Act o = new_act(w.cout, a.H, a.W, out_dtype, stats);
        double fl = 2.0 * B * a.H * a.W * (double)w.cout * w.cin * w.ntaps;
        if (w2) fl += 2.0 * B * a.H * a.W * (double)w2->cout * w2->cin;
        h->flops += fl;
        // large maps: separate finalize launch into a coefficient array (allocated in the dry run as well: the arena is sized by it)
        const float* coef_arr = (gn && (long)a.H * a.W > g_gn_inline) ? gn_coef(a, a2, *gn) : nullptr;
        // Large maps: the producer writes per-workgroup partial totals instead of queueing 640 atomics per item on each total; every consumer
        // of such a map goes through gn_coef (same threshold), which sums them.  (allocated in the dry run as well: the arena is sized by it)
        // Only for the two launch shapes that can write them (pointer-free test, so that the dry run sizes the arena like the real one):
        // the input convolution's walk (conv_in_split_wgs workgroups per item) and conv_v4's tile grid (the buffer used to be
        // max(tiles, 256, conv_in_wgs) rows for every stats-producing convolution of a large map, whatever kernel ran it).
        const bool in_shape = a.dtype == DT_F32 && w.cin <= 8 && w.ntaps == 9 && out_dtype != DT_F32;
        const bool v4_shape = w.ntaps == 9 && w.cout > 32 && a.dtype == out_dtype && a.H % 16 == 0 && a.W % 32 == 0;
        const int part_rows = in_shape ? conv_in_split_wgs(a.H, a.W) : v4_shape ? conv_v4_tiles(a.H, a.W) : 0;
        long long* part = (stats && g_stats_part && part_rows > 0 && (long)a.H * a.W > g_gn_inline)
                              ? (long long*)arena->alloc((size_t)B * part_rows * w.cout * 2 * sizeof(long long)) : nullptr;
        if (h->dry) return o;
        ConvArgs p{};
        p.src0 = a.p; p.C0 = a.C; p.src1 = a2 ? a2->p : nullptr; p.C1 = a2 ? a2->C : 0; p.in_dtype = a.dtype;
        p.coef = coef_arr; p.act = act; p.w = h->blob + w.w_off; p.cout_pad = w.cout_pad;
        if (gn && !coef_arr) {
            const int C = a.C + (a2 ? a2->C : 0), groups = std::min(C / 4, 32);
            p.gn_st0 = a.stats; p.gn_st1 = a2 ? a2->stats : nullptr;
            p.gn_gamma = W<float>(gn->g_off); p.gn_beta = W<float>(gn->b_off); p.gn_groups = groups;
            p.gn_inv_n = 1.0f / ((float)(C / groups) * (float)(a.H * a.W)); p.gn_eps = 1e-6f;
        }
        p.wb = w.has_wb ? h->blob + w.wb_off : nullptr;
        p.bias = W<float>(w2 ? bias_off : w.b_off);
        if (w2) { p.x0 = sx0->p; p.XC0 = sx0->C; p.x1 = sx1 ? sx1->p : nullptr; p.XC1 = sx1 ? sx1->C : 0; p.w2 = h->blob + w2->w_off;
                  p.w2b = w2->has_wb ? h->blob + w2->wb_off : nullptr; }
        p.temb = temb; p.temb_bstride = temb_bstride;
        p.res = res ? res->p : nullptr; p.out_scale = scale;
        p.pyr = pyr; p.w4 = cb ? W<float>(cb->w_off) : nullptr; p.b4 = cb ? W<float>(cb->b_off) : nullptr;
        p.out = o.p; p.out_dtype = out_dtype; p.stats = o.stats;
        p.B = B; p.H = a.H; p.W = a.W; p.Cout = w.cout; p.ntaps = w.ntaps;
        const bool main_variant = conv_v4_eligible(p);        // the dominant kernel (conv_v4_kernel, large maps)
        if (const int nparts = part ? conv_stats_parts(p) : 0) {
            if (nparts <= part_rows) { p.stats_part = part; p.stats = nullptr; o.part = part; o.ntiles = nparts; }   // (else: atomics; cannot happen, both count the same grid)
        }
        // consumers that read the totals directly (inline GroupNorm below gn_inline pixels, the fused attention) must never meet a producer
        // that wrote partial totals instead: both sides test the same H * W > gn_inline - checked here, where the input is consumed
        if (gn && !coef_arr && (a.part || (a2 && a2->part))) { arena->overflow = true; h->dry = true; return o; }   // reported by eval_status as a stale plan
        if (h->profile && (main_variant || (h->profile_all && conv_v2_eligible(p)))) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, s);
            launch_conv(p, s);
            (void)hipEventRecord(e1, s);
            h->prof_events.push_back({e0, e1});
            h->prof_flops.push_back(fl); h->prof_main.push_back(main_variant); h->prof_hw.push_back({a.H, a.W});
            {   // algorithmic HBM bytes of this launch: every operand once (input, shortcut input, residual, output, weights)
                const double es = (double)dtype_size(a.dtype), px = (double)B * a.H * a.W;
                double by = px * (w.cin + (w2 ? w2->cin : 0)) * es + px * w.cout * dtype_size(out_dtype) * (res ? 2.0 : 1.0);
                by += (double)w.ntaps * w.cin * w.cout * es + (w2 ? (double)w2->cin * w2->cout * es : 0.0);
                if (pyr) by += px * 4 * 4;
                h->prof_bytes.push_back(by);
            }
            char d[160]; snprintf(d, sizeof d, "%-28s H=%3d W=%3d Cin=%3d Cout=%3d taps=%d gn=%d res=%d sc=%d", w.wname.c_str(), a.H, a.W, w.cin, w.cout, w.ntaps, gn != nullptr, res != nullptr, w2 ? w2->cin : 0);
            h->prof_desc.push_back(d);
        } else if (h->profile && w.ntaps == 9 && w.cout > 8 && !(a.dtype == DT_F32 && w.cin <= 8)) {
            // the other implicit-GEMM kernels (conv_v2 on the middle maps, conv_sk on the smallest): MFMA-bound records of the aux list
            const double es = (double)dtype_size(a.dtype), px = (double)B * a.H * a.W;
            double by = px * (w.cin + (w2 ? w2->cin : 0)) * es + px * w.cout * dtype_size(out_dtype) * (res ? 2.0 : 1.0);
            by += (double)w.ntaps * w.cin * w.cout * es + (w2 ? (double)w2->cin * w2->cout * es : 0.0);
            timed_aux(conv_sk_eligible(p) ? "conv_sk" : conv_v2_eligible(p) ? "conv_v2" : "conv_generic", a.H, a.W, by, [&] { launch_conv(p, s); });
            h->prof_aux.back().flops = fl;
        } else if (h->profile && (w.cout <= 8 || (a.dtype == DT_F32 && w.cin <= 8))) {
            const double px = (double)B * a.H * a.W;
            const double by = px * w.cin * dtype_size(a.dtype) + px * w.cout * dtype_size(out_dtype) * (res ? 2.0 : 1.0) + (double)w.ntaps * w.cin * w.cout * dtype_size(a.dtype);
            timed_aux(w.cout <= 8 ? "pyr_conv" : "conv_in", a.H, a.W, by, [&] { launch_conv(p, s); });
        } else {
            launch_conv(p, s);
        }
        return o;
    }

    // ResnetBlockBigGANpp.forward (reference layerspp.py:282-314). `skip` = second half of the channel concat.
    Act resblock(const Act& x, const Act* skip, const ResW& r, const float* pyr = nullptr, const CombineW* cb = nullptr) {
        const float rs = 0.70710678118654752440f;   // 1/sqrt(2)
        const float* temb = tembias ? tembias + r.dense_row0 : nullptr;    // unconditional: no Dense_0(temb) term
        const int dt = h->act_dtype;
        Act hcur, xr;
        const Act* sx0 = &x; const Act* sx1 = skip;          // inputs of the 1x1 shortcut (Conv_2)
        if (r.up || r.down) {
            const int H2 = r.up ? x.H * 2 : x.H / 2, W2 = r.up ? x.W * 2 : x.W / 2;
            Act hr = new_act(x.C, H2, W2, dt, false);
            xr = new_act(x.C, H2, W2, dt, false);
            float* coef0 = gn_coef(x, skip, r.gn0);          // the resampling kernels take the GroupNorm as a coefficient array
            if (!h->dry) {
                // reads the map once, writes the activated and the raw resampled copy
                const double by = (double)B * x.H * x.W * x.C * dtype_size(dt) + 2.0 * B * H2 * W2 * x.C * dtype_size(dt);
                if (r.up) timed_aux("fir_up", x.H, x.W, by, [&] { launch_fir_up2(x.p, dt, coef0, 1, hr.p, xr.p, B, x.H, x.W, x.C, s); });
                else      timed_aux("fir_down", x.H, x.W, by, [&] { launch_fir_down2(x.p, dt, coef0, 1, hr.p, xr.p, B, x.H, x.W, x.C, s); });
            }
            hcur = conv(hr, nullptr, nullptr, 0, r.c0, temb, nullptr, 1.f, nullptr, nullptr, dt, true);
            sx0 = &xr; sx1 = nullptr;
        } else {
            hcur = conv(x, skip, &r.gn0, 1, r.c0, temb, nullptr, 1.f, nullptr, nullptr, dt, true);
        }
        if (r.has_c2)   // Conv_1 and the Conv_2 shortcut accumulate into the same MFMA tile (one K loop)
            return conv(hcur, nullptr, &r.gn1, 1, r.c1, nullptr, nullptr, rs, pyr, cb, dt, true, sx0, sx1, &r.c2, r.b12_off);
        return conv(hcur, nullptr, &r.gn1, 1, r.c1, nullptr, &x, rs, pyr, cb, dt, true);
    }

    // AttnBlockpp.forward (reference layerspp.py:77-93)
    Act attention(const Act& x, const AttnW& aw) {
        const int dt = h->act_dtype;
        if (g_attn_fused && x.stats && attn_fused_eligible(dt, x.H * x.W, x.C)) {       // one launch, MFMA contractions (use_attn.hip)
            Act o = new_act(x.C, x.H, x.W, dt, true);
            h->flops += 4.0 * 2.0 * B * x.H * x.W * (double)x.C * x.C;                  // the four NIN, as the unfused path counts them
            if (h->dry) return o;
            if (x.part) { arena->overflow = true; h->dry = true; return o; }   // (the fused block reads x.stats: a producer with partial totals cannot feed it)
            AttnArgs a{};
            a.x = x.p; a.out = o.p; a.N = x.H * x.W;
            a.gn_st = x.stats; a.gn_gamma = W<float>(aw.gn.g_off); a.gn_beta = W<float>(aw.gn.b_off); a.gn_groups = std::min(x.C / 4, 32); a.gn_eps = 1e-6f;
            a.wq = h->blob + aw.q.w_off; a.wk = h->blob + aw.k.w_off; a.wv = h->blob + aw.v.w_off; a.wo = h->blob + aw.o.w_off;
            a.bq = W<float>(aw.q.b_off); a.bk = W<float>(aw.k.b_off); a.bv = W<float>(aw.v.b_off); a.bo = W<float>(aw.o.b_off);
            a.stats = o.stats;
            launch_attn_fused(a, dt, B, s);
            return o;
        }
        Act q = conv(x, nullptr, &aw.gn, 0, aw.q, nullptr, nullptr, 1.f, nullptr, nullptr, dt, false);
        Act k = conv(x, nullptr, &aw.gn, 0, aw.k, nullptr, nullptr, 1.f, nullptr, nullptr, dt, false);
        Act v = conv(x, nullptr, &aw.gn, 0, aw.v, nullptr, nullptr, 1.f, nullptr, nullptr, dt, false);
        Act a = new_act(x.C, x.H, x.W, dt, false);
        const int N = x.H * x.W;
        const int ck = dt == DT_F32 ? 32 : 64;
        if (N > 512 && N % 128 == 0 && x.C % 128 == 0 && N % ck == 0) {
            // long sequences (64x80 bottleneck of the refine generator): per batch item two implicit GEMMs on conv_kernel.
            // The key tensor [N][C] already is a packed 1x1 weight [Cout = N][1][Cin = C]; v^T plays that part for P.V
            const size_t es = dtype_size(dt);
            char* sc = (char*)arena->alloc((size_t)B * N * N * es);           // scores -> probabilities, [B][N][N]
            char* vt = (char*)arena->alloc((size_t)B * x.C * N * es);         // [B][C][N]
            if (!h->dry) {
                launch_transpose_nc(v.p, vt, dt, B, N, x.C, s);
                for (int b = 0; b < B; ++b) {
                    ConvArgs p{};
                    p.src0 = (char*)q.p + (size_t)b * N * x.C * es; p.C0 = x.C; p.in_dtype = dt;
                    p.w = (char*)k.p + (size_t)b * N * x.C * es; p.cout_pad = N;
                    p.out_scale = 1.0f / std::sqrt((float)x.C);               // int(C) ** -0.5 (layerspp.py:84)
                    p.out = sc + (size_t)b * N * N * es; p.out_dtype = dt;
                    p.B = 1; p.H = x.H; p.W = x.W; p.Cout = N; p.ntaps = 1;
                    launch_conv(p, s);
                }
                launch_softmax_rows(sc, dt, (long)B * N, N, s);
                for (int b = 0; b < B; ++b) {
                    ConvArgs p{};
                    p.src0 = sc + (size_t)b * N * N * es; p.C0 = N; p.in_dtype = dt;
                    p.w = vt + (size_t)b * x.C * N * es; p.cout_pad = x.C;
                    p.out_scale = 1.f;
                    p.out = (char*)a.p + (size_t)b * N * x.C * es; p.out_dtype = dt;
                    p.B = 1; p.H = x.H; p.W = x.W; p.Cout = x.C; p.ntaps = 1;
                    launch_conv(p, s);
                }
            }
        } else if (!h->dry) {
            launch_attention(q.p, k.p, v.p, a.p, dt, B, N, x.C, s);
        }
        return conv(a, nullptr, nullptr, 0, aw.o, nullptr, &x, 0.70710678118654752440f, nullptr, nullptr, dt, true);
    }

    // NCSNpp.forward (reference ncsnpp.py:324-501): x4 = packed network input, returns the fp32 pyramid [B,F,T,4]
    Act run(const float* x4) {
        use_handle* H = h;
        const use_config& c = H->cfg;
        const int L = c.n_levels, nrb = c.num_res_blocks, dt = H->act_dtype;
        arena->reset();
        st_arena->reset();
        // all GroupNorm totals of this evaluation; a kernel, not hipMemsetAsync: a captured memset node of this (small) size does not
        // clear the region again on the second replay of a graph that holds four or more evaluations (ROCm 7.2; seen as NaN from the
        // second sampler call on for batches of 1-3 items, i.e. without the sub-batch split)
        if (!H->dry && st_arena->base) {
            const size_t n16 = st_arena->cap / 16;
            hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 1024)), dim3(256), 0, s, (uint4*)st_arena->base, n16);
        }
        if (primary) { H->flops = 0.0; H->debug.clear(); H->debug_B = B; }
        const int pcp = H->pcp;
        Act xin; xin.p = (void*)x4; xin.C = pcp; xin.H = c.n_freq; xin.W = H->T; xin.dtype = DT_F32;
        std::vector<Act> hs;
        hs.push_back(conv(xin, nullptr, nullptr, 0, H->conv_in, nullptr, nullptr, 1.f, nullptr, nullptr, dt, true));
        if (primary) H->debug["h_in"] = hs.back();
        Act ipyr = xin;
        size_t ri = 0, ci = 0;
        for (int lvl = 0; lvl < L; ++lvl) {
            for (int k = 0; k < nrb; ++k) hs.push_back(resblock(hs.back(), nullptr, H->res[ri++]));
            if (lvl == std::min(stagger_level, L - 1) && ev_stagger && !H->dry) (void)hipEventRecord(ev_stagger, s);   // small maps follow
            if (lvl != L - 1) {
                Act nip = new_act(pcp, ipyr.H / 2, ipyr.W / 2, DT_F32, false);        // pyramid_downsample
                if (!H->dry) launch_fir_down2(ipyr.p, DT_F32, nullptr, 0, nullptr, nip.p, B, ipyr.H, ipyr.W, pcp, s);
                ipyr = nip;
                if (pcp == 4) {                                // Combine fused into the block's last convolution
                    hs.push_back(resblock(hs.back(), nullptr, H->res[ri++], (const float*)ipyr.p, &H->combines[ci++]));
                } else {                                       // 6-channel input: Combine as its own pass over the block's output
                    Act o = resblock(hs.back(), nullptr, H->res[ri++]);
                    const CombineW& cb = H->combines[ci++];
                    if (!H->dry) {
                        hipLaunchKernelGGL(zero16_kernel, dim3((unsigned)(((size_t)B * o.C + 255) / 256)), dim3(256), 0, s, (uint4*)o.stats,
                                           (size_t)B * o.C);                                             // totals of the combined map (16 bytes per channel)
                        launch_combine_add(o.p, o.dtype, (const float*)ipyr.p, W<float>(cb.w_off), W<float>(cb.b_off), o.stats, B,
                                           (long)o.H * o.W, o.C, s);
                    }
                    o.part = nullptr; o.ntiles = 0;            // (the totals of the combined map are what the consumers read)
                    hs.push_back(o);
                }
            }
        }
        if (primary) H->debug["down_out"] = hs.back();
        Act hc = resblock(hs.back(), nullptr, H->res[ri++]);
        if (primary) H->debug["pre_attn"] = hc;
        hc = attention(hc, H->attn);
        if (primary) H->debug["post_attn"] = hc;
        hc = resblock(hc, nullptr, H->res[ri++]);
        Act pyr; bool have_pyr = false; size_t pi = 0;
        for (int lvl = L - 1; lvl >= 0; --lvl) {
            for (int k = 0; k < nrb + 1; ++k) {
                Act sk = hs.back(); hs.pop_back();
                hc = resblock(hc, &sk, H->res[ri++]);
            }
            const PyrW& pw = H->pyrs[pi++];
            if (!have_pyr) {
                pyr = conv(hc, nullptr, &pw.gn, 1, pw.conv, nullptr, nullptr, 1.f, nullptr, nullptr, DT_F32, false);
                have_pyr = true;
            } else {
                Act up = new_act(pcp, pyr.H * 2, pyr.W * 2, DT_F32, false);             // pyramid_upsample
                if (!H->dry) launch_fir_up2(pyr.p, DT_F32, nullptr, 0, nullptr, up.p, B, pyr.H, pyr.W, pcp, s);
                pyr = conv(hc, nullptr, &pw.gn, 1, pw.conv, nullptr, &up, 1.f, nullptr, nullptr, DT_F32, false);
            }
            if (lvl != 0) hc = resblock(hc, nullptr, H->res[ri++]);
        }
        if (primary) H->debug["h_last"] = hc;
        if (primary) H->debug["pyramid"] = pyr;
        return pyr;
    }
};

// time embedding for `rows` time values (batch items, or sampler steps) -> tembias[rows][dense_rows]
static void run_temb(use_handle* h, const float* t, int n, float* silu_buf, float* out, hipStream_t s) {
    const int nf = h->cfg.nf;
    launch_temb_mlp(t, 1, (const float*)(h->blob + h->gfp_off), (const float*)(h->blob + h->l1w_off),
                    (const float*)(h->blob + h->l1b_off), (const float*)(h->blob + h->l2w_off),
                    (const float*)(h->blob + h->l2b_off), silu_buf, n, nf, s);
    launch_temb_dense(silu_buf, (const float*)(h->blob + h->dense_w_off), (const float*)(h->blob + h->dense_b_off), out,
                      n, h->dense_rows, 4 * nf, s);
}

// score = -net(cat[x, y], t): x, y device complex64
static void run_score(use_handle* h, const float2* x, const float2* y, const float* tembias, int temb_bstride,
                      const float* t, int t_stride, float2* out, hipStream_t s, float sign = -1.f, const float2* y2 = nullptr) {
    if (h->cfg.no_sigma_scale) t = nullptr;                   // score_out: no division by t
    const long n_per_b = (long)h->cfg.n_freq * h->T;
    const int pcp = h->pcp;
    launch_pack_input(x, y, y2, h->x4, (long)h->B * n_per_b, s);
    const float* outw = (const float*)(h->blob + h->outw_off); const float* outb = (const float*)(h->blob + h->outb_off);
    if (h->nsub == 1) {
        Fwd f0{h, s, tembias, temb_bstride, t, t_stride};
        f0.B = h->B; f0.arena = &h->arena; f0.st_arena = &h->st_arena[0];
        Act pyr = f0.run(h->x4);
        if (!h->dry) launch_score_out((const float*)pyr.p, pcp, t, t_stride, outw, outb, out, h->B, n_per_b, sign, s);
        return;
    }
    // Sub-batches on separate streams.  The items are independent inside the network (GroupNorm is per item), so this is
    // a pure re-scheduling: sub-batch i+1 starts when sub-batch i has finished its large down-path maps, and from then on
    // the latency-bound kernels of one sub-batch (small maps, GroupNorm finalisation, attention: a few workgroups each)
    // execute beside the large convolutions of another instead of leaving the chip idle.
    const bool overlap = !h->profile;                         // per-launch timing wants the kernels one at a time
    if (overlap) (void)hipEventRecord(h->ev_fork, s);
    int b0 = 0;
    for (int i = 0; i < h->nsub; ++i) {
        hipStream_t si = (i == 0 || !overlap) ? s : h->aux_stream[i];
        if (overlap && i) {
            (void)hipStreamWaitEvent(si, h->ev_fork, 0);
            (void)hipStreamWaitEvent(si, h->ev_stagger[i - 1], 0);
        }
        Fwd f{h, si, tembias ? tembias + (size_t)b0 * temb_bstride : nullptr, temb_bstride,
              t ? t + (size_t)b0 * t_stride : nullptr, t_stride};
        f.B = h->sub_B[i]; f.arena = i ? &h->sub_arena[i] : &h->arena; f.st_arena = &h->st_arena[i]; f.primary = i == 0;
        if (overlap && i + 1 < h->nsub) { f.ev_stagger = h->ev_stagger[i]; f.stagger_level = g_stagger_level; }
        Act pyr = f.run(h->x4 + (size_t)b0 * n_per_b * pcp);
        if (!h->dry)
        launch_score_out((const float*)pyr.p, pcp, t ? t + (size_t)b0 * t_stride : nullptr, t_stride, outw, outb,
                         out + (size_t)b0 * n_per_b, h->sub_B[i], n_per_b, sign, si);
        if (overlap && i) (void)hipEventRecord(h->ev_join[i], si);
        b0 += h->sub_B[i];
    }
    if (overlap)
        for (int i = 1; i < h->nsub; ++i) (void)hipStreamWaitEvent(s, h->ev_join[i], 0);
}

// ---------------------------------------------------------------------------------------------------------
// SDE scalars (reference sdes.py:205-243, 88-92)
// ---------------------------------------------------------------------------------------------------------
static double ouve_logsig(const use_config& c) { return std::log((double)c.sigma_max / (double)c.sigma_min); }
static float ouve_std(const use_config& c, float t) {
    const double th = c.theta, ls = ouve_logsig(c), sm = c.sigma_min, tt = t;
    return (float)std::sqrt(sm * sm * std::exp(-2 * th * tt) * (std::exp(2 * (th + ls) * tt) - 1) * ls / (th + ls));
}
static float ouve_diffusion(const use_config& c, float t) {
    const float sigma = c.sigma_min * std::pow(c.sigma_max / c.sigma_min, t);     // float32 like the reference
    return (float)((double)sigma * std::sqrt(2 * ouve_logsig(c)));
}
// torch.linspace(start, end, steps) float32 CPU semantics: symmetric fill from both ends, each element one fused
// multiply-add (matches the ATen CPU kernel bit-for-bit on the build container for N in 1..200)
static void linspace_f32(float start, float end, int steps, std::vector<float>& out) {
    out.resize(steps);
    if (steps == 1) { out[0] = start; return; }
    const float step = (end - start) / (float)(steps - 1);
    const int half = steps / 2;
    for (int i = 0; i < steps; ++i) out[i] = i < half ? std::fmaf(step, (float)i, start) : std::fmaf(-step, (float)(steps - 1 - i), end);
}

static void predictor_coeffs(const use_config& c, int predictor, float t, int N, float& cd, float& cs, float& cn) {
    const float g = ouve_diffusion(c, t);
    if (predictor == USE_PRED_REVERSE_DIFFUSION) {            // predictors.py:61-68 over sdes.py:88-92,159-173
        const float dt = (float)(1.0 / N);
        const float G = g * std::sqrt(dt);
        cd = c.theta * dt; cs = G * G; cn = G;
    } else {                                                  // Euler-Maruyama: predictors.py:44-53, sdes.py:125-157
        const double dt = 1.0 / N;
        cd = (float)(c.theta * dt); cs = (float)((double)(g * g) * dt); cn = (float)((double)g * std::sqrt(dt));
    }
}

// ---------------------------------------------------------------------------------------------------------
// the PC loop (reference sampling/__init__.py:59-71)
// ---------------------------------------------------------------------------------------------------------
__global__ void set_rng_kernel(unsigned long long* st, unsigned long long seed, unsigned long long base) { st[0] = seed; st[1] = base; }

// steps [i0, i1) of the loop (i0 == 0: preceded by the prior sampling); the whole loop is run_sampler(h, noise, s, 0, N)
static void run_sampler(use_handle* h, const float2* noise, hipStream_t s, int i0, int i1) {
    const use_sampler_config& sc = h->sc;
    const long n = (long)h->B * h->cfg.n_freq * h->T, n_per_b = (long)h->cfg.n_freq * h->T;
    const int ncorr = sc.corrector == USE_CORR_NONE ? 0 : sc.corrector_steps;
    auto nz = [&](unsigned d) { return noise ? noise + (size_t)d * n : nullptr; };
    unsigned d = 1 + (unsigned)i0 * (unsigned)(ncorr + (sc.predictor == USE_PRED_NONE ? 0 : 1));   // noise draws consumed so far
    if (i0 == 0) launch_prior(h->Y, nz(0), RngRef{h->rng_state, 0}, ouve_std(h->cfg, 1.0f), h->X, n, s);
    for (int i = i0; i < i1; ++i) {
        const float t = h->timesteps[i];
        const float* temb = h->temb_table + (size_t)i * h->dense_rows;
        for (int k = 0; k < ncorr; ++k) {
            run_score(h, h->X, h->Cond, temb, 0, h->ts_dev + i, 0, h->score, s, -1.f, h->pcp == 8 ? h->cond2_buf : nullptr);
            RngRef rr{h->rng_state, d};
            if (sc.corrector == USE_CORR_LANGEVIN) {
                launch_langevin_norms(h->score, nz(d), rr, h->lang_partial, h->B, n_per_b, h->lang_blocks, s);
                launch_langevin_step(h->lang_partial, h->B, h->lang_blocks, sc.snr, h->lang_step, s);
                launch_corrector(h->X, h->score, nz(d), rr, h->lang_step, 0.f, h->X, nullptr, n, s);
            } else {                                               // ALD: correctors.py:79-98
                const float sd = sc.snr * ouve_std(h->cfg, t);
                launch_corrector(h->X, h->score, nz(d), rr, nullptr, sd * sd * 2.f, h->X, nullptr, n, s);
            }
            d++;
        }
        if (sc.predictor == USE_PRED_NONE) {
            if (i == sc.N - 1) (void)hipMemcpyAsync(h->Xmean, h->X, (size_t)n * 8, hipMemcpyDeviceToDevice, s);
        } else {
            run_score(h, h->X, h->Cond, temb, 0, h->ts_dev + i, 0, h->score, s, -1.f, h->pcp == 8 ? h->cond2_buf : nullptr);
            float cd, cs, cn; predictor_coeffs(h->cfg, sc.predictor, t, sc.N, cd, cs, cn);
            launch_predictor(h->X, h->Y, h->score, nz(d), RngRef{h->rng_state, d}, cd, cs, cn, h->X,
                             i == sc.N - 1 ? h->Xmean : nullptr, n, s);
            d++;
        }
    }
}

static void drop_graphs(use_handle* h) {
    for (auto& v : h->graph_exec) { for (auto g : v) if (g) (void)hipGraphExecDestroy(g); v.clear(); }
    if (h->score_graph) { (void)hipGraphExecDestroy(h->score_graph); h->score_graph = nullptr; }
}

// everything of the handle that belongs to ONE plan (the current plan lives in the handle's own fields)
#define USE_PLAN_FIELDS(X)                                                                                            \
    X(B) X(T) X(nsub) X(arena) X(arena_alloc) X(arena_plan_bytes) X(persist) X(persist_bytes) X(persist_alloc) X(x4) X(silu_temb) X(tembias) X(t_dev) \
    X(Y) X(X) X(Xmean) X(score) X(xin) X(cond_buf) X(cond2_buf) X(Cond) X(lang_partial) X(lang_step) X(rng_state) X(lang_blocks)   \
    X(sc) X(sampler_set) X(timesteps) X(ts_dev) X(temb_table) X(silu_table) X(noise_copy) X(noise_copy_bytes) X(score_graph)      \
    X(debug) X(debug_B) X(opt_gen_at_plan)
struct use_handle::PlanState {
#define X(f) decltype(use_handle::f) f;
    USE_PLAN_FIELDS(X)
#undef X
    int sub_B[MAX_SUB]; Arena sub_arena[MAX_SUB], st_arena[MAX_SUB];
    std::vector<hipGraphExec_t> graph_exec[2];
};
static long long g_opt_gen = 0;                  // bumped by every use_set_option: plans built under other options are not reused
static int g_plan_cache = 4;                     // plans kept besides the current one
static void plan_stash(use_handle* h, use_handle::PlanState& p) {      // handle -> p; the handle's plan fields are left empty
#define X(f) p.f = std::move(h->f);
    USE_PLAN_FIELDS(X)
#undef X
    for (int i = 0; i < MAX_SUB; ++i) { p.sub_B[i] = h->sub_B[i]; p.sub_arena[i] = h->sub_arena[i]; p.st_arena[i] = h->st_arena[i]; }
    for (int i = 0; i < 2; ++i) { p.graph_exec[i] = std::move(h->graph_exec[i]); h->graph_exec[i].clear(); }
    h->B = h->T = 0; h->arena = Arena{}; h->arena_alloc = 0; h->persist = nullptr; h->persist_bytes = h->persist_alloc = 0;
    h->ts_dev = h->temb_table = h->silu_table = nullptr; h->noise_copy = nullptr; h->noise_copy_bytes = 0; h->score_graph = nullptr;
    h->sampler_set = false; h->timesteps.clear(); h->debug.clear();
}
static void plan_restore(use_handle* h, use_handle::PlanState& p) {    // p -> handle
#define X(f) h->f = std::move(p.f);
    USE_PLAN_FIELDS(X)
#undef X
    for (int i = 0; i < MAX_SUB; ++i) { h->sub_B[i] = p.sub_B[i]; h->sub_arena[i] = p.sub_arena[i]; h->st_arena[i] = p.st_arena[i]; }
    for (int i = 0; i < 2; ++i) h->graph_exec[i] = std::move(p.graph_exec[i]);
}
static void plan_free(use_handle::PlanState* p) {
    for (auto& v : p->graph_exec) for (auto g : v) if (g) (void)hipGraphExecDestroy(g);
    if (p->score_graph) (void)hipGraphExecDestroy(p->score_graph);
    if (p->arena.base) (void)hipFree(p->arena.base);
    if (p->persist) (void)hipFree(p->persist);
    if (p->temb_table) (void)hipFree(p->temb_table);
    if (p->silu_table) (void)hipFree(p->silu_table);
    if (p->ts_dev) (void)hipFree(p->ts_dev);
    if (p->noise_copy) (void)hipFree(p->noise_copy);
    delete p;
}
static void plan_cache_clear(use_handle* h) {                          // weights changed / handle destroyed: every parked plan is stale
    for (auto* p : h->plan_cache) plan_free(p);
    h->plan_cache.clear();
}

static int ensure_sde_scratch(use_handle* h) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!h->sde_buf) {
        const size_t bytes = 512 + (size_t)use_handle::SDE_MAX_B * use_handle::SDE_BLOCKS * 2 * 4;
        HIPCHK(hipMalloc((void**)&h->sde_buf, bytes));
        HIPCHK(hipMemset(h->sde_buf, 0, bytes));
        h->sde_rng = (unsigned long long*)h->sde_buf; h->sde_step = (float*)(h->sde_buf + 256);
        h->sde_partial = (float*)(h->sde_buf + 512);
    }
    return USE_OK;
}

static int ensure_cap_stream(use_handle* h) {
    if (!h->cap_stream) HIPCHK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    return USE_OK;
}

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" {

int use_get_stat(use_handle* h, const char* name, long long* value) {
    if (!h || !name || !value) return fail(USE_E_INVALID, "null argument");
    if (!strcmp(name, "graph_captures")) { *value = h->n_graph_captures; return USE_OK; }      // sampling-loop segments captured so far
    if (!strcmp(name, "plans_built")) { *value = h->n_plans_built; return USE_OK; }
    if (!strcmp(name, "plan_cache_hits")) { *value = h->n_plan_cache_hits; return USE_OK; }
    if (!strcmp(name, "plans_parked")) { *value = (long long)h->plan_cache.size(); return USE_OK; }
    if (!strcmp(name, "plan_stale")) { *value = h->B && h->opt_gen_at_plan != g_opt_gen ? 1 : 0; return USE_OK; }   // an option changed since use_plan
    return fail(USE_E_INVALID, "unknown statistic '%s'", name);
}

int use_set_option(const char* name, long long value) {
    if (!name) return fail(USE_E_INVALID, "option name is null");
    ++g_opt_gen;                                               // plans built under the previous options are not reused
    if (!strcmp(name, "attn_fused")) { g_attn_fused = (int)value; return USE_OK; }
    if (!strcmp(name, "plan_cache")) { g_plan_cache = (int)std::max(0LL, std::min(16LL, value)); return USE_OK; }   // parked plans per handle
    if (!strcmp(name, "subbatch_min_items")) { g_subbatch_min_items = (int)std::max(1LL, value); return USE_OK; }
    if (!strcmp(name, "subbatch")) { g_subbatch = (int)value; return USE_OK; }          // takes effect at the next use_plan
    if (!strcmp(name, "stagger_level")) { g_stagger_level = (int)value; return USE_OK; }
    if (!strcmp(name, "stats_part")) { g_stats_part = (int)value; return USE_OK; }                  // takes effect at the next use_plan
    if (!strcmp(name, "gn_inline")) { g_gn_inline = (long)value; return USE_OK; }                  // takes effect at the next use_plan
    if (!strcmp(name, "conv_v4_min_blocks")) { conv_v4_set_min_blocks((long)value); return USE_OK; }
    if (!strcmp(name, "conv_v5")) { conv_v5_set((int)value); return USE_OK; }
    if (!strcmp(name, "fir_strip")) { fir_set_strip((int)value); return USE_OK; }
    if (!strcmp(name, "pyr_ws")) { pyr_conv_set_ws((int)value); return USE_OK; }
    if (!strcmp(name, "wgrad_mfma16")) { wgrad_set_mfma16((int)value); return USE_OK; }
    if (!strcmp(name, "wgrad_blocks")) { wgrad_set_blocks((int)value); return USE_OK; }
    if (!strcmp(name, "conv_in_wgs")) { if (value < 1 || value > 4096) return fail(USE_E_INVALID, "conv_in_wgs: 1 ... 4096"); g_conv_in_wgs = (int)value; return USE_OK; }
    if (!strcmp(name, "conv_sk_max_px")) { conv_sk_set_max_px((long)value); return USE_OK; }             // 0: conv_sk off
    return fail(USE_E_INVALID, "unknown option '%s'", name);
}
const char* use_last_error(void) { return g_err.c_str(); }
const char* use_version(void) { return "use_hip 0.1 (gfx950)"; }

int use_create(const use_config* cfg, int device, use_handle** out) {
    if (!cfg || !out) return fail(USE_E_INVALID, "null argument");
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->nf % 32 != 0 || cfg->nf < 32 || cfg->num_res_blocks < 1)
        return fail(USE_E_INVALID, "unsupported architecture (nf must be a multiple of 32, 1..8 levels)");
    for (int i = 0; i < cfg->n_levels; ++i)
        if (cfg->ch_mult[i] < 1 || cfg->nf * cfg->ch_mult[i] > 512) return fail(USE_E_INVALID, "unsupported architecture (1 <= nf * ch_mult <= 512)");
    if (cfg->n_freq % (1 << (cfg->n_levels - 1)) != 0)
        return fail(USE_E_INVALID, "n_freq=%d is not divisible by 2^(levels-1)", cfg->n_freq);
    if (cfg->precision != USE_PREC_FP32 && cfg->precision != USE_PREC_BF16 && cfg->precision != USE_PREC_FP16) return fail(USE_E_INVALID, "bad precision");
    if (cfg->input_channels != 0 && cfg->input_channels != 2 && cfg->input_channels != 4 && cfg->input_channels != 6)
        return fail(USE_E_INVALID, "input_channels must be 4 (x and y), 6 (x, y and a second conditioning) or 2 (y alone), got %d", cfg->input_channels);
    use_handle* h = new use_handle();
    h->cfg = *cfg; h->device = device;
    h->act_dtype = cfg->precision == USE_PREC_BF16 ? DT_BF16 : cfg->precision == USE_PREC_FP16 ? DT_F16 : DT_F32;
    int rc = build_arch(h);
    if (rc) { delete h; return rc; }
    *out = h;
    return USE_OK;
}

int use_destroy(use_handle* h) {
    if (!h) return USE_OK;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    drop_graphs(h); plan_cache_clear(h);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    for (int i = 0; i < MAX_SUB; ++i) {
        if (h->aux_stream[i]) (void)hipStreamDestroy(h->aux_stream[i]);
        if (h->ev_stagger[i]) { (void)hipEventDestroy(h->ev_stagger[i]); (void)hipEventDestroy(h->ev_join[i]); }
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->blob) (void)hipFree(h->blob);
    if (h->arena.base) (void)hipFree(h->arena.base);
    if (h->persist) (void)hipFree(h->persist);
    if (h->temb_table) (void)hipFree(h->temb_table);
    if (h->silu_table) (void)hipFree(h->silu_table);
    if (h->ts_dev) (void)hipFree(h->ts_dev);
    if (h->noise_copy) (void)hipFree(h->noise_copy);
    if (h->sde_buf) (void)hipFree(h->sde_buf);
    delete h;
    return USE_OK;
}

int use_num_expected_weights(use_handle* h) { return h ? (int)h->expected.size() : USE_E_INVALID; }
int use_expected_weight(use_handle* h, int index, const char** name, int64_t* shape4, int* ndim) {
    if (!h || index < 0 || index >= (int)h->expected.size()) return fail(USE_E_INVALID, "bad index");
    const Expected& e = h->expected[index];
    if (name) *name = e.name.c_str();
    if (ndim) *ndim = (int)e.shape.size();
    if (shape4) for (size_t i = 0; i < 4; ++i) shape4[i] = i < e.shape.size() ? e.shape[i] : 1;
    return USE_OK;
}

int use_set_weight(use_handle* h, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!h || !name || !data || !shape) return fail(USE_E_INVALID, "null argument");
    auto it = h->expected_index.find(name);
    if (it == h->expected_index.end()) return fail(USE_E_INVALID, "unexpected weight name '%s'", name);
    const Expected& e = h->expected[it->second];
    if ((int)e.shape.size() != ndim) return fail(USE_E_INVALID, "weight '%s': rank %d, expected %zu", name, ndim, e.shape.size());
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] != e.shape[i]) return fail(USE_E_INVALID, "weight '%s': dim %d is %lld, expected %lld", name, i, (long long)shape[i], (long long)e.shape[i]);
        n *= (size_t)shape[i];
    }
    h->host_w[name].assign(data, data + n);
    h->weights_ready = false;
    return USE_OK;
}

int use_alloc_weight_blob(use_handle* h) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (!h->blob) HIPCHK(hipMalloc((void**)&h->blob, h->blob_bytes));
    h->weights_ready = true;   // contents to be filled by the caller's broadcast
    h->sampler_set = false;    // the time-embedding table was built from the previous contents
    drop_graphs(h); plan_cache_clear(h);
    return USE_OK;
}

int use_commit_weights(use_handle* h) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    std::vector<char> host(h->blob_bytes, 0);
    int rc = pack_all(h, host.data());
    if (rc) return rc;
    if (!h->blob) HIPCHK(hipMalloc((void**)&h->blob, h->blob_bytes));
    HIPCHK(hipMemcpy(h->blob, host.data(), h->blob_bytes, hipMemcpyHostToDevice));
    h->host_w.clear();
    h->weights_ready = true;
    h->sampler_set = false;    // the temb table depends on the weights
    drop_graphs(h); plan_cache_clear(h);
    return USE_OK;
}

// ---- packed weight file: header + the device blob, so that a deployment starts without a state dict ------
// The blob layout is private to a library build: BLOB_LAYOUT is bumped whenever pack_all / the blob offsets change.
namespace {
constexpr uint32_t BLOB_LAYOUT = 7;          // 7: the 16-channel-chunk copies of round 4 (conv_v10, removed) are gone again; 5: + split-bf16 copy of the input convolution (16-bit modes); 4: piece-swizzled slab copies
struct BlobHeader {
    char magic[8];                           // "USEHIPWB"
    uint32_t header_bytes, layout;
    int32_t nf, n_levels, ch_mult[8], num_res_blocks, precision, input_channels, unconditional, no_sigma_scale;
    uint32_t crc32, reserved;
    uint64_t blob_bytes;
};
uint32_t crc32_of(const unsigned char* p, size_t n) {
    static uint32_t table[256]; static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = c & 1 ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
BlobHeader make_header(const use_handle* h) {
    BlobHeader hd{};
    memcpy(hd.magic, "USEHIPWB", 8);
    hd.header_bytes = sizeof(BlobHeader); hd.layout = BLOB_LAYOUT;
    const use_config& c = h->cfg;
    hd.nf = c.nf; hd.n_levels = c.n_levels; for (int i = 0; i < 8; ++i) hd.ch_mult[i] = i < c.n_levels ? c.ch_mult[i] : 0;
    hd.num_res_blocks = c.num_res_blocks; hd.precision = c.precision;
    hd.input_channels = c.input_channels ? c.input_channels : 4; hd.unconditional = c.unconditional != 0; hd.no_sigma_scale = c.no_sigma_scale != 0;
    hd.blob_bytes = h->blob_bytes;
    return hd;
}
}  // namespace

int use_save_weight_blob(use_handle* h, const char* path) {
    if (!h || !path) return fail(USE_E_INVALID, "null argument");
    std::vector<char> host(h->blob_bytes, 0);
    if (!h->host_w.empty()) {                                 // weights set but not committed: pack on the host, no GPU needed
        int rc = pack_all(h, host.data());
        if (rc) return rc;
    } else if (h->blob && h->weights_ready) {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipMemcpy(host.data(), h->blob, h->blob_bytes, hipMemcpyDeviceToHost));
    } else {
        return fail(USE_E_STATE, "no weights to save (use_set_weight every tensor, or commit / load first)");
    }
    BlobHeader hd = make_header(h);
    hd.crc32 = crc32_of((const unsigned char*)host.data(), host.size());
    FILE* f = fopen(path, "wb");
    if (!f) return fail(USE_E_INVALID, "cannot open '%s' for writing", path);
    const bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && fwrite(host.data(), 1, host.size(), f) == host.size();
    if (fclose(f) != 0 || !ok) return fail(USE_E_INVALID, "short write to '%s'", path);
    return USE_OK;
}

int use_load_weight_blob(use_handle* h, const char* path) {
    if (!h || !path) return fail(USE_E_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(USE_E_INVALID, "cannot open '%s'", path);
    BlobHeader hd{}, want = make_header(h);
    std::vector<char> host(h->blob_bytes);
    if (fread(&hd, sizeof hd, 1, f) != 1) { fclose(f); return fail(USE_E_INVALID, "'%s' is truncated", path); }
    if (memcmp(hd.magic, want.magic, 8) != 0 || hd.header_bytes != sizeof hd) { fclose(f); return fail(USE_E_INVALID, "'%s' is not a use_hip weight file", path); }
    if (hd.layout != want.layout) { fclose(f); return fail(USE_E_INVALID, "'%s' has blob layout %u, this build reads %u: re-pack the checkpoint", path, hd.layout, want.layout); }
    {
        BlobHeader a = hd; a.crc32 = 0;                       // everything but the checksum describes the network
        if (memcmp(&a, &want, sizeof a) != 0) { fclose(f); return fail(USE_E_INVALID, "'%s' was packed for a different network configuration / precision", path); }
    }
    const bool ok = fread(host.data(), 1, host.size(), f) == host.size();
    fclose(f);
    if (!ok) return fail(USE_E_INVALID, "'%s' is truncated", path);
    if (crc32_of((const unsigned char*)host.data(), host.size()) != hd.crc32) return fail(USE_E_INVALID, "'%s': checksum mismatch", path);
    HIPCHK(hipSetDevice(h->device));
    if (!h->blob) HIPCHK(hipMalloc((void**)&h->blob, h->blob_bytes));
    HIPCHK(hipMemcpy(h->blob, host.data(), h->blob_bytes, hipMemcpyHostToDevice));
    h->host_w.clear();
    h->weights_ready = true;
    h->sampler_set = false;
    drop_graphs(h); plan_cache_clear(h);                     // parked plans hold time-embedding tables + graphs of the OLD weights
    return USE_OK;
}

int use_weight_blob(use_handle* h, void** dev_ptr, size_t* bytes) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    if (dev_ptr) *dev_ptr = h->blob;
    if (bytes) *bytes = h->blob_bytes;
    return USE_OK;
}

int use_plan(use_handle* h, int B, int Tpad) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    if (B < 1 || Tpad < 64 || Tpad % 64 != 0) return fail(USE_E_INVALID, "plan needs B >= 1 and T' a positive multiple of 64 (got B=%d T'=%d)", B, Tpad);
    if ((Tpad >> (h->cfg.n_levels - 1)) < 1) return fail(USE_E_INVALID, "T' too small for %d levels", h->cfg.n_levels);
    HIPCHK(hipSetDevice(h->device));
    if (h->arena.base && h->B == B && h->T == Tpad && h->opt_gen_at_plan == g_opt_gen) return USE_OK;     // the current plan
    HIPCHK(hipDeviceSynchronize());
    use_handle::PlanState* hit = nullptr;                     // the requested plan, if it is parked: taken out BEFORE the current one
    for (size_t i = 0; i < h->plan_cache.size(); ++i) {       // goes in (else a cycle over capacity + 1 shapes would evict what it needs)
        use_handle::PlanState* p = h->plan_cache[i];
        if (p->B == B && p->T == Tpad && p->opt_gen_at_plan == g_opt_gen) { hit = p; h->plan_cache.erase(h->plan_cache.begin() + (long)i); break; }
    }
    if (h->arena.base) {                                      // park the current plan (with its graphs)
        auto* p = new use_handle::PlanState();
        plan_stash(h, *p);
        if (g_plan_cache > 0 && p->opt_gen_at_plan == g_opt_gen) h->plan_cache.push_back(p); else plan_free(p);
        while ((int)h->plan_cache.size() > g_plan_cache) { plan_free(h->plan_cache.front()); h->plan_cache.erase(h->plan_cache.begin()); }
    }
    if (hit) {
        plan_restore(h, *hit);
        delete hit;
        ++h->n_plan_cache_hits;
        return USE_OK;
    }
    ++h->n_plans_built;
    h->opt_gen_at_plan = g_opt_gen;
    h->B = B; h->T = Tpad; h->sampler_set = false;
    // sub-batch pipelining (run_score): `subbatch` sub-batches of at least 2 items each
    // How many sub-batches: a question of grid quantisation on the 256 CUs (same-box sweeps, profiles/r5_e2e_ab_subbatch_sweep.txt).  The 3x3
    // convolutions of the 256 x 320 / 128 x 160 maps launch 160 / 80 workgroups per item: sub-batches of 4 items leave 2.5 / 1.25 rounds (83 % /
    // 62 % of the last round's CUs busy), of 3 items 1.9 / 0.94 - batch 8 runs 1.6 % faster as 3 + 3 + 2 than as 4 + 4 (bf16 and fp16), five or
    // more streams 12-17 % slower; batch 16 is fastest as 8 + 8 (5 / 2.5 rounds; 6 + 5 + 5: +0.7 %).
    const int want = g_subbatch >= 0 ? g_subbatch : (B >= 6 && B <= 11) ? 3 : 2;
    h->nsub = std::max(1, std::min(std::min(want, MAX_SUB), B / std::max(1, g_subbatch_min_items)));
    for (int i = 0; i < MAX_SUB; ++i) h->sub_B[i] = i < h->nsub ? B / h->nsub + (i < B % h->nsub ? 1 : 0) : 0;
    // dry runs to size the activation arenas (one per sub-batch, carved from one allocation).  The allocation itself is kept
    // when it is large enough: a predict run over files of different lengths re-plans for almost every batch
    char* old_base = h->arena.base;
    h->arena = Arena{};
    h->dry = true;
    size_t caps[MAX_SUB] = {}, stcaps[MAX_SUB] = {}, total = 0;
    for (int i = 0; i < h->nsub; ++i) {
        Arena& ar = i ? h->sub_arena[i] : h->arena;
        ar = Arena{}; h->st_arena[i] = Arena{};
        Fwd f{h, nullptr, nullptr, 0, nullptr, 0};
        f.B = h->sub_B[i]; f.arena = &ar; f.st_arena = &h->st_arena[i]; f.primary = i == 0;
        f.run(nullptr);
        caps[i] = (ar.peak + 4096 + 255) & ~(size_t)255; total += caps[i];
        stcaps[i] = (h->st_arena[i].peak + 255) & ~(size_t)255; total += stcaps[i];
    }
    h->dry = false;
    char* base = old_base;
    if (!base || total > h->arena_alloc) {
        if (base) { HIPCHK(hipFree(base)); base = nullptr; h->arena_alloc = 0; }
        if (hipMalloc((void**)&base, total) != hipSuccess) {
            (void)hipGetLastError();
            plan_cache_clear(h);                              // parked plans hold their workspaces: give them up and try once more
            if (hipMalloc((void**)&base, total) != hipSuccess)
                return fail(USE_E_NOMEM, "cannot allocate %.1f MB of activation workspace", total / 1e6);
        }
        h->arena_alloc = total;
    }
    h->arena.base = base; h->arena.cap = caps[0];             // owns the allocation (arena_alloc bytes); its own slice is caps[0]
    h->arena_plan_bytes = total;
    {
        size_t off = caps[0];
        for (int i = 1; i < h->nsub; ++i) { h->sub_arena[i].base = base + off; h->sub_arena[i].cap = caps[i]; off += caps[i]; }
        for (int i = 0; i < h->nsub; ++i) { h->st_arena[i].base = base + off; h->st_arena[i].cap = stcaps[i]; off += stcaps[i]; }
        h->arena.dry_on_overflow = &h->dry;
        for (int i = 0; i < h->nsub; ++i) { h->sub_arena[i].dry_on_overflow = &h->dry; h->st_arena[i].dry_on_overflow = &h->dry; }
    }
    if (h->nsub > 1 && !h->ev_fork) HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < h->nsub; ++i) {
        if (i && !h->aux_stream[i]) HIPCHK(hipStreamCreateWithFlags(&h->aux_stream[i], hipStreamNonBlocking));
        if (h->nsub > 1 && !h->ev_stagger[i]) {
            HIPCHK(hipEventCreateWithFlags(&h->ev_stagger[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
        }
    }
    // persistent buffers
    const size_t n = (size_t)B * h->cfg.n_freq * Tpad;
    h->lang_blocks = (int)std::min<size_t>(256, ((size_t)h->cfg.n_freq * Tpad + 255) / 256);
    size_t off = 0;
    auto take = [&](size_t bytes) { off = (off + 255) & ~(size_t)255; size_t o = off; off += bytes; return o; };
    const size_t o_x4 = take(n * 4 * h->pcp), o_c2 = take(h->pcp == 8 ? n * 8 : 0), o_Y = take(n * 8), o_X = take(n * 8), o_Xm = take(n * 8), o_sc = take(n * 8),
                 o_xin = take(n * 8), o_cond = take(n * 8), o_st = take((size_t)B * 4 * h->cfg.nf * 4), o_tb = take((size_t)B * h->dense_rows * 4),
                 o_t = take((size_t)B * 4), o_lp = take((size_t)B * h->lang_blocks * 2 * 4), o_ls = take(256), o_rng = take(256);
    h->persist_bytes = off;
    if (!h->persist || off > h->persist_alloc) {
        if (h->persist) { HIPCHK(hipFree(h->persist)); h->persist = nullptr; h->persist_alloc = 0; }
        if (hipMalloc((void**)&h->persist, off) != hipSuccess) {
            (void)hipGetLastError();
            plan_cache_clear(h);
            if (hipMalloc((void**)&h->persist, off) != hipSuccess) return fail(USE_E_NOMEM, "cannot allocate %.1f MB of state", off / 1e6);
        }
        h->persist_alloc = off;
    }
    h->x4 = (float*)(h->persist + o_x4); h->Y = (float2*)(h->persist + o_Y); h->X = (float2*)(h->persist + o_X);
    h->Xmean = (float2*)(h->persist + o_Xm); h->score = (float2*)(h->persist + o_sc); h->xin = (float2*)(h->persist + o_xin);
    h->cond_buf = (float2*)(h->persist + o_cond); h->Cond = h->Y;
    h->cond2_buf = h->pcp == 8 ? (float2*)(h->persist + o_c2) : nullptr;
    h->silu_temb = (float*)(h->persist + o_st); h->tembias = (float*)(h->persist + o_tb); h->t_dev = (float*)(h->persist + o_t);
    h->lang_partial = (float*)(h->persist + o_lp); h->lang_step = (float*)(h->persist + o_ls);
    h->rng_state = (unsigned long long*)(h->persist + o_rng);
    HIPCHK(hipMemset(h->persist, 0, off));
    return USE_OK;
}

int use_workspace_bytes(use_handle* h, size_t* bytes) {
    if (!h || !bytes) return fail(USE_E_INVALID, "null argument");
    *bytes = h->arena_plan_bytes + h->persist_bytes + h->blob_bytes;
    return USE_OK;
}

double use_flops_per_score(use_handle* h) { return h ? h->flops : 0.0; }

static int check_ready(use_handle* h) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    if (!h->weights_ready) return fail(USE_E_STATE, "weights not committed (use_commit_weights / use_alloc_weight_blob)");
    if (!h->B) return fail(USE_E_STATE, "use_plan has not been called");
    // Options are read at plan time (buffer sizes: conv_in_wgs, stats_part, gn_inline, subbatch*; kernel choice: every threshold): a plan
    // built under other options is stale.  An error, not a silent re-plan: a re-plan frees the captured graphs and the sampler tables
    // behind the caller's back (and round 5's answer - overflowing the arena and abort() - was no answer at all).
    if (h->opt_gen_at_plan != g_opt_gen)
        return fail(USE_E_STATE, "use_set_option was called after use_plan: the plan is stale, call use_plan (and use_set_sampler) again");
    return USE_OK;
}

// after an evaluation was issued: did every tensor fit its arena?  (defence in depth behind check_ready's stale-plan test)
static int eval_status(use_handle* h) {
    bool bad = h->arena.overflow;
    for (int i = 0; i < MAX_SUB; ++i) bad = bad || h->sub_arena[i].overflow || h->st_arena[i].overflow;
    if (!bad) return USE_OK;
    h->arena.overflow = false; h->dry = false;
    for (int i = 0; i < MAX_SUB; ++i) h->sub_arena[i].overflow = h->st_arena[i].overflow = false;
    return fail(USE_E_STATE, "activation workspace too small for this evaluation (an option that sizes plan buffers changed after use_plan?): call use_plan again; the output is invalid");
}

// one network evaluation; sign -1: the score (use_score), +1: the raw backbone output (use_forward)
static int eval_net(use_handle* h, const void* x, const void* y, const float* t, void* out, use_stream_t stream, float sign,
                    const void* y2 = nullptr) {
    int rc = check_ready(h); if (rc) return rc;
    const use_config& c = h->cfg;
    const bool two_ch = c.input_channels == 2;
    if (!x || !out) return fail(USE_E_INVALID, "null tensor");
    if ((c.input_channels == 6) != (y2 != nullptr))
        return fail(USE_E_INVALID, c.input_channels == 6 ? "a 6-channel network takes two conditioning tensors (use_score2)" : "this network takes one conditioning tensor");
    if (two_ch ? y != nullptr : y == nullptr) return fail(USE_E_INVALID, two_ch ? "a 2-channel network takes x alone (y must be null)" : "null tensor");
    if (!t && (!c.unconditional || !c.no_sigma_scale)) return fail(USE_E_INVALID, "this network needs the time t");
    hipStream_t s = (hipStream_t)stream;
    if (!c.unconditional) run_temb(h, t, h->B, h->silu_temb, h->tembias, s);
    run_score(h, (const float2*)x, (const float2*)y, c.unconditional ? nullptr : h->tembias, h->dense_rows, t, 1, (float2*)out, s, sign,
              (const float2*)y2);
    HIPCHK(hipGetLastError());
    return eval_status(h);
}
int use_score2(use_handle* h, const void* x, const void* y, const void* y2, const float* t, void* out, use_stream_t stream) {
    return eval_net(h, x, y, t, out, stream, -1.f, y2);
}
int use_score(use_handle* h, const void* x, const void* y, const float* t, void* out, use_stream_t stream) {
    return eval_net(h, x, y, t, out, stream, -1.f);
}
int use_forward(use_handle* h, const void* x, const void* y, const float* t, void* out, use_stream_t stream) {
    return eval_net(h, x, y, t, out, stream, +1.f);
}

int use_profile_score(use_handle* h, const void* x, const void* y, const float* t, void* out, use_stream_t stream,
                      double* conv_ms, double* conv_flops, double* conv_bytes, int* conv_launches, double* total_ms) {
    int rc = check_ready(h); if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t t0, t1; HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1));
    h->profile = true; h->prof_events.clear(); h->prof_flops.clear(); h->prof_bytes.clear(); h->prof_desc.clear(); h->prof_main.clear();
    for (auto& a : h->prof_aux) { if (a.e0) (void)hipEventDestroy(a.e0); if (a.e1) (void)hipEventDestroy(a.e1); }
    h->prof_aux.clear();
    const bool verbose = getenv("USE_HIP_PROFILE_VERBOSE") != nullptr;
    h->profile_all = verbose;                                 // verbose: also list the conv_v2_kernel launches
    HIPCHK(hipEventRecord(t0, s));
    rc = use_score(h, x, y, t, out, stream);
    h->profile = false;
    if (rc) return rc;
    HIPCHK(hipEventRecord(t1, s));
    HIPCHK(hipStreamSynchronize(s));
    double ms = 0.0, fl = 0.0, by = 0.0; int nmain = 0;
    for (size_t i = 0; i < h->prof_events.size(); ++i) {
        float e = 0.f; HIPCHK(hipEventElapsedTime(&e, h->prof_events[i].first, h->prof_events[i].second));
        if (h->prof_main[i]) {
            ms += e; fl += h->prof_flops[i]; by += h->prof_bytes[i]; ++nmain;
            // also as a record of the aux list, so that callers can break the dominant kernel down by map (bench.py: roofline.by_map)
            use_handle::AuxProf r; r.name = "conv_v4"; r.H = h->prof_hw[i].first; r.W = h->prof_hw[i].second; r.bytes = h->prof_bytes[i]; r.ms = e; r.flops = h->prof_flops[i];
            h->prof_aux.push_back(r);
        }
        if (verbose) fprintf(stderr, "[use_profile] %s  %8.3f ms  %7.1f TFLOP/s\n", h->prof_desc[i].c_str(), e, h->prof_flops[i] / e / 1e9);
        (void)hipEventDestroy(h->prof_events[i].first); (void)hipEventDestroy(h->prof_events[i].second);
    }
    for (auto& a : h->prof_aux) {
        if (!a.e0) continue;                                 // (the per-launch records of the dominant kernel appended above)
        float e = 0.f; HIPCHK(hipEventElapsedTime(&e, a.e0, a.e1));
        a.ms = e; (void)hipEventDestroy(a.e0); (void)hipEventDestroy(a.e1); a.e0 = a.e1 = nullptr;
    }
    float tot = 0.f; HIPCHK(hipEventElapsedTime(&tot, t0, t1));
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    if (conv_ms) *conv_ms = ms;
    if (conv_flops) *conv_flops = fl;
    if (conv_bytes) *conv_bytes = by;
    if (conv_launches) *conv_launches = nmain;
    if (total_ms) *total_ms = tot;
    h->profile_all = false;
    h->prof_events.clear(); h->prof_flops.clear(); h->prof_bytes.clear(); h->prof_main.clear(); h->prof_hw.clear();
    return USE_OK;
}

int use_profile_aux(use_handle* h, int index, char* name, int name_cap, int* H, int* W, double* bytes, double* ms) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    if (index < 0 || index >= (int)h->prof_aux.size()) return 1;          // past the end (not an error: the caller iterates)
    const auto& a = h->prof_aux[(size_t)index];
    if (name && name_cap > 0) { strncpy(name, a.name.c_str(), (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (H) *H = a.H; if (W) *W = a.W; if (bytes) *bytes = a.bytes; if (ms) *ms = a.ms;
    return USE_OK;
}

int use_profile_aux_flops(use_handle* h, int index, double* flops) {
    if (!h) return fail(USE_E_INVALID, "null handle");
    if (index < 0 || index >= (int)h->prof_aux.size()) return 1;
    if (flops) *flops = h->prof_aux[(size_t)index].flops;
    return USE_OK;
}

int use_timesteps(int N, float t_eps, float* out) {
    if (N < 1 || !out) return fail(USE_E_INVALID, "bad arguments");
    std::vector<float> ts; linspace_f32(1.0f, t_eps, N, ts);
    memcpy(out, ts.data(), (size_t)N * 4);
    return USE_OK;
}

int use_set_sampler(use_handle* h, const use_sampler_config* sc) {
    int rc = check_ready(h); if (rc) return rc;
    if (!sc || sc->N < 1) return fail(USE_E_INVALID, "sampler needs N >= 1");
    if (h->cfg.unconditional || h->cfg.input_channels == 2 || h->cfg.no_sigma_scale)
        return fail(USE_E_STATE, "the reverse-SDE sampler needs the conditional 4-channel score network");
    if (sc->predictor < 0 || sc->predictor > 2 || sc->corrector < 0 || sc->corrector > 2) return fail(USE_E_INVALID, "unknown predictor/corrector id");
    if (sc->corrector != USE_CORR_NONE && sc->corrector_steps < 0) return fail(USE_E_INVALID, "negative corrector_steps");
    HIPCHK(hipSetDevice(h->device));
    if (h->sampler_set && !memcmp(&h->sc, sc, sizeof *sc)) return USE_OK;       // unchanged (e.g. a parked plan taken back): tables and graphs stay
    HIPCHK(hipDeviceSynchronize());
    drop_graphs(h);
    h->sc = *sc;
    linspace_f32(1.0f, sc->t_eps, sc->N, h->timesteps);         // sampling/__init__.py:63 (sde.T == 1)
    if (h->temb_table) { HIPCHK(hipFree(h->temb_table)); h->temb_table = nullptr; }
    if (h->silu_table) { HIPCHK(hipFree(h->silu_table)); h->silu_table = nullptr; }
    if (h->ts_dev) { HIPCHK(hipFree(h->ts_dev)); h->ts_dev = nullptr; }
    HIPCHK(hipMalloc((void**)&h->temb_table, (size_t)sc->N * h->dense_rows * 4));
    HIPCHK(hipMalloc((void**)&h->silu_table, (size_t)sc->N * 4 * h->cfg.nf * 4));
    HIPCHK(hipMalloc((void**)&h->ts_dev, (size_t)sc->N * 4));
    HIPCHK(hipMemcpy(h->ts_dev, h->timesteps.data(), (size_t)sc->N * 4, hipMemcpyHostToDevice));
    run_temb(h, h->ts_dev, sc->N, h->silu_table, h->temb_table, nullptr);   // t is batch-uniform and known up front
    HIPCHK(hipDeviceSynchronize());
    h->sampler_set = true;
    return USE_OK;
}

int use_num_noise_draws(use_handle* h) {
    if (!h || !h->sampler_set) return fail(USE_E_STATE, "sampler not configured");
    const int ncorr = h->sc.corrector == USE_CORR_NONE ? 0 : h->sc.corrector_steps;
    return 1 + h->sc.N * (ncorr + (h->sc.predictor == USE_PRED_NONE ? 0 : 1));
}

int use_get_timesteps(use_handle* h, float* out, int n) {
    if (!h || !h->sampler_set || !out) return fail(USE_E_STATE, "sampler not configured");
    for (int i = 0; i < n && i < (int)h->timesteps.size(); ++i) out[i] = h->timesteps[i];
    return (int)h->timesteps.size();
}

int use_sample(use_handle* h, const void* y, const void* noise, uint64_t seed, void* out, use_stream_t stream) {
    return use_sample_cond(h, y, nullptr, noise, seed, out, stream);
}

int use_sample_cond(use_handle* h, const void* y, const void* cond, const void* noise, uint64_t seed, void* out, use_stream_t stream) {
    return use_sample_cond2(h, y, cond, nullptr, noise, seed, out, stream);
}

int use_sample_cond2(use_handle* h, const void* y, const void* cond, const void* cond2, const void* noise, uint64_t seed, void* out,
                     use_stream_t stream) {
    int rc = check_ready(h); if (rc) return rc;
    if (!h->sampler_set) return fail(USE_E_STATE, "use_set_sampler has not been called");
    if (!y || !out) return fail(USE_E_INVALID, "null tensor");
    if ((h->cfg.input_channels == 6) != (cond2 != nullptr))
        return fail(USE_E_INVALID, h->cfg.input_channels == 6 ? "a 6-channel network samples with two conditioning tensors (use_sample_cond2)"
                                                              : "this network takes one conditioning tensor");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)h->B * h->cfg.n_freq * h->T;
    HIPCHK(hipMemcpyAsync(h->Y, y, n * 8, hipMemcpyDeviceToDevice, s));
    if (cond2) HIPCHK(hipMemcpyAsync(h->cond2_buf, cond2, n * 8, hipMemcpyDeviceToDevice, s));
    {   // the conditioning spectrogram the network sees beside x: the SDE's y itself, or a separate one (condition="denoised")
        float2* want = cond ? h->cond_buf : h->Y;
        if (want != h->Cond) { HIPCHK(hipStreamSynchronize(s)); drop_graphs(h); h->Cond = want; }    // captured graphs hold the pointer
        if (cond) HIPCHK(hipMemcpyAsync(h->cond_buf, cond, n * 8, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(set_rng_kernel, dim3(1), dim3(1), 0, s, h->rng_state, (unsigned long long)seed, 0ull);
    if (!h->sc.use_graph) {
        run_sampler(h, (const float2*)noise, s, 0, h->sc.N);
        rc = eval_status(h); if (rc) return rc;
    } else {
        const int gi = noise ? 1 : 0;
        const float2* nz = nullptr;
        if (noise) {
            const size_t nb = (size_t)use_num_noise_draws(h) * n * 8;
            if (nb > h->noise_copy_bytes) {
                HIPCHK(hipStreamSynchronize(s));
                if (h->noise_copy) HIPCHK(hipFree(h->noise_copy));
                h->noise_copy = nullptr; h->noise_copy_bytes = 0;
                for (auto g : h->graph_exec[1]) if (g) (void)hipGraphExecDestroy(g);
                h->graph_exec[1].clear();
                if (hipMalloc((void**)&h->noise_copy, nb) != hipSuccess) return fail(USE_E_NOMEM, "cannot allocate %.1f MB noise staging", nb / 1e6);
                h->noise_copy_bytes = nb;
            }
            HIPCHK(hipMemcpyAsync(h->noise_copy, noise, nb, hipMemcpyDeviceToDevice, s));
            nz = h->noise_copy;
        }
        if (h->graph_exec[gi].empty()) {
            // The loop is captured in segments of at most 64 score evaluations (~16 k kernel nodes each, the size the
            // configs[1] graph has): one graph of the N = 200 config's 400 evaluations (~10^5 nodes) crashes the HIP runtime
            // at instantiation.  The segments are launched back to back on the caller's stream.
            rc = ensure_cap_stream(h); if (rc) return rc;
            const int ncorr = h->sc.corrector == USE_CORR_NONE ? 0 : h->sc.corrector_steps;
            const int per_seg = std::max(1, 64 / (ncorr + 1));
            for (int i0 = 0; i0 < h->sc.N; i0 += per_seg) {
                hipGraph_t g = nullptr;
                hipGraphExec_t ge = nullptr;
                HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
                run_sampler(h, nz, h->cap_stream, i0, std::min(h->sc.N, i0 + per_seg));
                // (nothing returns between begin and end: a failed launch invalidates the capture and surfaces here, with the
                // stream out of capture mode either way)
                hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
                if (eval_status(h)) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); drop_graphs(h); return USE_E_STATE; }
                if (e != hipSuccess || !g) {
                    if (g) (void)hipGraphDestroy(g);
                    (void)hipGetLastError();
                    drop_graphs(h);
                    return fail(USE_E_HIP, "capturing the sampling loop failed: %s", hipGetErrorString(e));
                }
                e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (e != hipSuccess) { drop_graphs(h); return fail(USE_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
                h->graph_exec[gi].push_back(ge);
                ++h->n_graph_captures;
            }
        }
        for (auto ge : h->graph_exec[gi]) HIPCHK(hipGraphLaunch(ge, s));
    }
    HIPCHK(hipMemcpyAsync(out, h->Xmean, n * 8, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipGetLastError());
    return USE_OK;
}

int use_spec_fwd(const void* stft, void* Y, int B, int F, int T, int Tpad, float factor, float exponent, use_stream_t s) {
    if (!stft || !Y || B < 1 || F < 1 || T < 1 || Tpad < T) return fail(USE_E_INVALID, "bad arguments");
    launch_spec_map((const float2*)stft, (float2*)Y, (long)B * F, T, T, Tpad, 1.f, exponent, factor, (hipStream_t)s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}
int use_spec_back(const void* X, void* stft, int B, int F, int T, int Tpad, float factor, float exponent, use_stream_t s) {
    if (!stft || !X || B < 1 || F < 1 || T < 1 || Tpad < T || factor == 0.f || exponent == 0.f) return fail(USE_E_INVALID, "bad arguments");
    launch_spec_map((const float2*)X, (float2*)stft, (long)B * F, T, Tpad, T, 1.f / factor, 1.f / exponent, 1.f, (hipStream_t)s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}

// ---- device STFT / iSTFT fused with the spectrogram glue (handle-free) -----------------------------------------
namespace {
std::mutex g_tw_mutex;
std::map<std::pair<int, int>, float2*> g_tw_tables;          // (device, n_fft) -> (cos, sin)(2 pi m / n_fft); lives for the process
int twiddles(int n_fft, hipStream_t s, const float2** out) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    auto it = g_tw_tables.find({dev, n_fft});
    if (it == g_tw_tables.end()) {                            // first use on this device: build the table (synchronises once)
        float2* tw = nullptr;
        HIPCHK(hipMalloc((void**)&tw, (size_t)n_fft * sizeof(float2)));
        launch_twiddle_table(tw, n_fft, s);
        HIPCHK(hipStreamSynchronize(s));
        it = g_tw_tables.emplace(std::make_pair(dev, n_fft), tw).first;
    }
    *out = it->second;
    return USE_OK;
}
int check_stft_args(int B, int L, int n_fft, int hop, int Tpad) {
    if (B < 1 || L < 1 || n_fft < 4 || (n_fft & 1) || hop < 1 || hop > n_fft) return fail(USE_E_INVALID, "bad STFT arguments (n_fft must be even, 1 <= hop <= n_fft)");
    if (L <= n_fft / 2) return fail(USE_E_INVALID, "signal of %d samples is too short for reflect padding with n_fft=%d", L, n_fft);
    if (Tpad < 1 + L / hop) return fail(USE_E_INVALID, "Tpad=%d is smaller than the %d frames of the signal", Tpad, 1 + L / hop);
    if ((size_t)n_fft * 8 + (size_t)((n_fft + hop - 1) / hop) * (n_fft / 2 + 1) * 8 > 64 * 1024) return fail(USE_E_INVALID, "n_fft / hop combination exceeds the synthesis kernel's LDS");
    return USE_OK;
}
}  // namespace

int use_stft_fwd(const float* wav, void* Y, int B, int L, int n_fft, int hop, const float* window, int Tpad, float factor,
                 float exponent, use_stream_t s) {
    if (!wav || !Y || !window) return fail(USE_E_INVALID, "null argument");
    int rc = check_stft_args(B, L, n_fft, hop, Tpad); if (rc) return rc;
    const float2* tw = nullptr;
    rc = twiddles(n_fft, (hipStream_t)s, &tw); if (rc) return rc;
    launch_stft_fwd(wav, window, tw, (float2*)Y, B, L, n_fft, hop, 1 + L / hop, Tpad, factor, exponent, (hipStream_t)s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}
int use_istft_back(const void* X, float* wav, int B, int L, int n_fft, int hop, const float* window, int Tpad, float factor,
                   float exponent, use_stream_t s) {
    if (!wav || !X || !window || factor == 0.f || exponent == 0.f) return fail(USE_E_INVALID, "bad arguments");
    int rc = check_stft_args(B, L, n_fft, hop, Tpad); if (rc) return rc;
    const float2* tw = nullptr;
    rc = twiddles(n_fft, (hipStream_t)s, &tw); if (rc) return rc;
    launch_istft_back((const float2*)X, window, tw, wav, B, L, n_fft, hop, Tpad, factor, exponent, (hipStream_t)s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}

int use_sde_prior(use_handle* h, const void* y, const void* noise, uint64_t seed, void* x, int64_t n, use_stream_t s) {
    int rc = ensure_sde_scratch(h); if (rc) return rc;
    hipLaunchKernelGGL(set_rng_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, h->sde_rng, (unsigned long long)seed, 0ull);
    launch_prior((const float2*)y, (const float2*)noise, RngRef{h->sde_rng, 0}, ouve_std(h->cfg, 1.0f), (float2*)x, n, (hipStream_t)s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}

int use_fill_noise(use_handle* h, uint64_t seed, int draw, void* out, int64_t n, use_stream_t s) {
    // draw `draw` of the device noise stream of use_sample(noise = NULL, seed): element i = Philox4x32-10(key = seed, counter = (i, draw))
    int rc = ensure_sde_scratch(h); if (rc) return rc;
    if (!out || n < 1 || draw < 0) return fail(USE_E_INVALID, "bad arguments");
    hipLaunchKernelGGL(set_rng_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, h->sde_rng, (unsigned long long)seed, 0ull);
    launch_fill_noise((float2*)out, RngRef{h->sde_rng, (unsigned)draw}, (long)n, (hipStream_t)s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}

int use_sde_predictor(use_handle* h, int predictor, float t, int N, const void* x, const void* y, const void* score,
                      const void* noise, uint64_t seed, void* x_out, void* x_mean, int64_t n, use_stream_t s) {
    int rc = ensure_sde_scratch(h); if (rc) return rc;
    if (predictor != USE_PRED_REVERSE_DIFFUSION && predictor != USE_PRED_EULER_MARUYAMA) return fail(USE_E_INVALID, "predictor id %d has no update kernel", predictor);
    float cd, cs, cn; predictor_coeffs(h->cfg, predictor, t, N, cd, cs, cn);
    hipLaunchKernelGGL(set_rng_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, h->sde_rng, (unsigned long long)seed, 0ull);
    launch_predictor((const float2*)x, (const float2*)y, (const float2*)score, (const float2*)noise, RngRef{h->sde_rng, 0},
                     cd, cs, cn, (float2*)x_out, (float2*)x_mean, n, (hipStream_t)s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}

int use_sde_corrector(use_handle* h, int corrector, float t, float snr, int B, const void* x, const void* score,
                      const void* noise, uint64_t seed, void* x_out, void* x_mean, int64_t n, use_stream_t st) {
    int rc = ensure_sde_scratch(h); if (rc) return rc;
    hipStream_t s = (hipStream_t)st;
    if (B < 1 || n % B != 0) return fail(USE_E_INVALID, "n must be a multiple of B");
    hipLaunchKernelGGL(set_rng_kernel, dim3(1), dim3(1), 0, s, h->sde_rng, (unsigned long long)seed, 0ull);
    RngRef rr{h->sde_rng, 0};
    if (corrector == USE_CORR_LANGEVIN) {
        if (B > use_handle::SDE_MAX_B) return fail(USE_E_INVALID, "B=%d exceeds %d", B, use_handle::SDE_MAX_B);
        launch_langevin_norms((const float2*)score, (const float2*)noise, rr, h->sde_partial, B, n / B, use_handle::SDE_BLOCKS, s);
        launch_langevin_step(h->sde_partial, B, use_handle::SDE_BLOCKS, snr, h->sde_step, s);
        launch_corrector((const float2*)x, (const float2*)score, (const float2*)noise, rr, h->sde_step, 0.f, (float2*)x_out, (float2*)x_mean, n, s);
    } else if (corrector == USE_CORR_ALD) {
        const float sd = snr * ouve_std(h->cfg, t);
        launch_corrector((const float2*)x, (const float2*)score, (const float2*)noise, rr, nullptr, sd * sd * 2.f, (float2*)x_out, (float2*)x_mean, n, s);
    } else return fail(USE_E_INVALID, "corrector id %d has no update kernel", corrector);
    HIPCHK(hipGetLastError());
    return USE_OK;
}

// ---- single-convolution harness (kernel bring-up / A-B timing; no reference counterpart) ----------------------------------
namespace {
__global__ void fill_uniform_kernel(void* p, size_t n, int dtype, unsigned seed, float lo, float hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + ((unsigned long long)seed << 32 | 0x1234567u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const float v = lo + (hi - lo) * (float)((z >> 40) * (1.0 / 16777216.0));
    if (dtype == DT_F32) ((float*)p)[i] = v;
    else if (dtype == DT_BF16) ((__bf16*)p)[i] = (__bf16)v;
    else ((_Float16*)p)[i] = (_Float16)v;
}
__global__ void to_float_kernel(const void* p, float* out, size_t n, int dtype) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = dtype == DT_F32 ? ((const float*)p)[i] : dtype == DT_BF16 ? (float)((const __bf16*)p)[i] : (float)((const _Float16*)p)[i];
}
}  // namespace

int use_conv_bench(const use_conv_case* c, float* out_host, float* stats_host, double* ms_avg, double* flops) {
    if (!c || c->B < 1 || c->H < 1 || c->W < 1 || c->C0 < 1 || c->Cout < 1) return fail(USE_E_INVALID, "bad conv case");
    const int dt = c->dtype;
    if (dt != DT_F32 && dt != DT_BF16 && dt != DT_F16) return fail(USE_E_INVALID, "bad dtype");
    const size_t es = dtype_size(dt);
    const int Cin = c->C0 + c->C1, XC = c->XC0 + c->XC1;
    const size_t px = (size_t)c->B * c->H * c->W;
    std::vector<void*> bufs;
    auto dalloc = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, bytes) != hipSuccess) return nullptr; bufs.push_back(q); return q; };
    auto cleanup = [&]() { for (void* q : bufs) (void)hipFree(q); };
    auto fill = [&](void* q, size_t n, int dtype, unsigned seed, float lo, float hi) {
        hipLaunchKernelGGL(fill_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, q, n, dtype, seed, lo, hi);
    };
    ConvW w; w.cin = Cin; w.cout = c->Cout; w.ntaps = 9; w.w_dtype = dt; w.cin_src = Cin; w.cout_src = c->Cout;
    w.cout_pad = c->Cout <= 32 ? 32 : (c->Cout + 127) / 128 * 128;
    ConvW w2 = w; w2.cin = XC; w2.cin_src = XC; w2.ntaps = 1;
    ConvArgs a{};
    a.B = c->B; a.H = c->H; a.W = c->W; a.Cout = c->Cout; a.ntaps = 9; a.in_dtype = dt; a.out_dtype = dt;
    a.C0 = c->C0; a.C1 = c->C1; a.XC0 = c->XC0; a.XC1 = c->XC1; a.act = c->act; a.out_scale = c->res || XC ? 0.70710678f : 1.f;
    a.cout_pad = w.cout_pad;
    void* src0 = dalloc(px * c->C0 * es); void* src1 = c->C1 ? dalloc(px * c->C1 * es) : nullptr;
    void* x0 = c->XC0 ? dalloc(px * c->XC0 * es) : nullptr; void* x1 = c->XC1 ? dalloc(px * c->XC1 * es) : nullptr;
    void* res = c->res ? dalloc(px * c->Cout * es) : nullptr;
    void* out = dalloc(px * c->Cout * es);
    float* outf = (float*)dalloc(px * c->Cout * 4);
    float* coef = c->gn ? (float*)dalloc((size_t)c->B * Cin * 2 * 4) : nullptr;
    float* bias = (float*)dalloc((size_t)c->Cout * 4); float* temb = (float*)dalloc((size_t)c->B * c->Cout * 4);
    const size_t wbytes = (size_t)9 * w.cout_pad * Cin * es, w2bytes = (size_t)w2.cout_pad * std::max(XC, 1) * es;
    char* dw = (char*)dalloc(wbytes); char* dwb = (char*)dalloc(wbytes);
    char* dw2 = XC ? (char*)dalloc(w2bytes) : nullptr; char* dw2b = XC ? (char*)dalloc(w2bytes) : nullptr;
    const size_t stats_bytes = (size_t)c->B * c->Cout * 2 * sizeof(long long);
    long long* stats = c->stats ? (long long*)dalloc(stats_bytes) : nullptr;
    if (!src0 || !out || !outf || !dw || !dwb || (c->stats && !stats)) { cleanup(); return fail(USE_E_NOMEM, "conv bench allocation failed"); }
    fill(src0, px * c->C0, dt, 1, -2.f, 2.f); if (src1) fill(src1, px * c->C1, dt, 2, -2.f, 2.f);
    if (x0) fill(x0, px * c->XC0, dt, 3, -1.f, 1.f); if (x1) fill(x1, px * c->XC1, dt, 4, -1.f, 1.f);
    if (res) fill(res, px * c->Cout, dt, 5, -1.f, 1.f);
    if (coef) fill(coef, (size_t)c->B * Cin * 2, DT_F32, 6, -0.9f, 1.1f);
    fill(bias, c->Cout, DT_F32, 7, -0.5f, 0.5f); fill(temb, (size_t)c->B * c->Cout, DT_F32, 8, -0.5f, 0.5f);
    {
        std::vector<float> hw((size_t)c->Cout * Cin * 9), hw2((size_t)c->Cout * std::max(XC, 1));
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 40) * (1.0 / 16777216.0)) * 2.f - 1.f; };
        const float sc = 1.0f / std::sqrt((float)Cin * 9.f / 3.f);
        for (auto& v : hw) v = rnd() * sc;
        for (auto& v : hw2) v = rnd() * (1.0f / std::sqrt((float)std::max(XC, 1) / 3.f));
        std::vector<char> hp(wbytes), hpb(wbytes);
        const bool slab = w.cout_pad % 128 == 0 && Cin % conv_v4_chunk(dt) == 0;
        pack_conv_raw(hw.data(), w, hp.data(), slab ? hpb.data() : nullptr);
        (void)hipMemcpy(dw, hp.data(), wbytes, hipMemcpyHostToDevice);
        if (slab) (void)hipMemcpy(dwb, hpb.data(), wbytes, hipMemcpyHostToDevice);
        a.w = dw; a.wb = slab ? dwb : nullptr;
        if (XC) {
            std::vector<char> hq(w2bytes), hqb(w2bytes);
            const bool slab2 = XC % conv_v4_chunk(dt) == 0 && w2.cout_pad % 128 == 0;
            pack_conv_raw(hw2.data(), w2, hq.data(), slab2 ? hqb.data() : nullptr);
            (void)hipMemcpy(dw2, hq.data(), w2bytes, hipMemcpyHostToDevice);
            if (slab2) (void)hipMemcpy(dw2b, hqb.data(), w2bytes, hipMemcpyHostToDevice);
            a.w2 = dw2; a.w2b = slab2 ? dw2b : nullptr;
        }
    }
    a.src0 = src0; a.src1 = src1; a.x0 = x0; a.x1 = x1; a.coef = coef; a.bias = bias; a.temb = c->temb ? temb : nullptr;
    a.temb_bstride = c->Cout; a.res = res; a.out = out; a.stats = stats;
    auto run = [&]() -> int {
        switch (c->variant) {
            case 0: launch_conv(a, 0); return 0;
            case 1: launch_conv_generic(a, 0); return 0;
            case 7: if (!conv_sk_eligible(a)) return -1; launch_conv_sk(a, 0); return 0;
            case 2: if (!conv_v2_eligible(a)) return -1; launch_conv_v2(a, 0); return 0;
            case 4: if (!a.wb || (XC && !a.w2b) || a.H % 16 || a.W % 32) return -1; launch_conv_v4(a, 0); return 0;   // (conv_v4 has no partial tiles)
            case 5: if (!a.wb || (XC && !a.w2b) || a.H % 16 || a.W % 32 || a.in_dtype == DT_F32) return -1; launch_conv_v5(a, 0); return 0;
            default: return -1;
        }
    };
    if (getenv("USE_HIP_DBG")) a.dbg = atoi(getenv("USE_HIP_DBG"));     // ablation bits (USE_HIP_ABLATE kernels)
    unsigned long long* trace = nullptr;
    if (getenv("USE_HIP_TRACE")) {                            // bring-up (USE_HIP_TRACE_BUILD kernels): stamps of workgroup $USE_HIP_TRACE
        trace = (unsigned long long*)dalloc(512 * 8);
        if (trace) { (void)hipMemset(trace, 0, 512 * 8); a.trace = trace; a.dbg = atoi(getenv("USE_HIP_TRACE")); }
    }
    if (stats) (void)hipMemset(stats, 0, stats_bytes);
    if (run() != 0) { cleanup(); return fail(USE_E_INVALID, "variant %d cannot run this case", c->variant); }
    if (hipDeviceSynchronize() != hipSuccess) { cleanup(); return fail(USE_E_HIP, "conv bench: launch failed: %s", hipGetErrorString(hipGetLastError())); }
    if (trace) {
        unsigned long long hb[512];
        (void)hipMemcpy(hb, trace, sizeof hb, hipMemcpyDeviceToHost);
        for (int g = 0; g < 2; ++g) {                           // wave 0 / wave 4 (conv_v4 stamps both wave groups)
            unsigned long long prev = hb[g * 256 + 1];
            for (int i = 0; i < 125 && hb[g * 256 + 2 * i]; ++i) {
                fprintf(stderr, "[trace G%d] id %3llu  +%6llu  @%8llu\n", g, hb[g * 256 + 2 * i], hb[g * 256 + 2 * i + 1] - prev, hb[g * 256 + 2 * i + 1] - hb[1]);
                prev = hb[g * 256 + 2 * i + 1];
            }
        }
        a.trace = nullptr;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = std::max(1, c->iters);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) run();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms_avg) *ms_avg = ms / iters;
    if (flops) *flops = 2.0 * (double)px * c->Cout * ((double)Cin * 9 + XC);
    int rc = USE_OK;
    if (out_host) {
        hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)((px * c->Cout + 255) / 256)), dim3(256), 0, 0, out, outf, px * c->Cout, dt);
        if (hipMemcpy(out_host, outf, px * c->Cout * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(USE_E_HIP, "copy back failed");
    }
    if (stats_host && stats) {                                // per (item, channel) totals of ONE launch, as floats
        (void)hipMemset(stats, 0, stats_bytes);
        run();
        std::vector<long long> hs((size_t)c->B * c->Cout * 2);
        if (hipMemcpy(hs.data(), stats, stats_bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(USE_E_HIP, "copy back failed");
        for (size_t i = 0; i < hs.size(); i += 2) {
            stats_host[i] = (float)((double)hs[i] / 1048576.0); stats_host[i + 1] = (float)((double)hs[i + 1] / 1048576.0);
        }
    }
    cleanup();
    return rc;
}

// ---- single operators on caller-owned device tensors (NHWC) with host fp32 weights: the kernels of the path one at a time, so that
// the per-operator golden vectors of the reference (tests/golden/fir.npz, resblock_*.npz, attn.npz) reach the HIP code itself ----
int use_op_conv(const use_conv_op* c, use_stream_t stream) {
    if (!c || c->B < 1 || c->H < 1 || c->W < 1 || c->C0 < 1 || c->Cout < 1 || !c->src0 || !c->w || !c->out) return fail(USE_E_INVALID, "use_op_conv: bad argument");
    const int dt = c->dtype, odt = c->out_dtype;
    if ((dt != DT_F32 && dt != DT_BF16 && dt != DT_F16) || (odt != DT_F32 && odt != DT_BF16 && odt != DT_F16)) return fail(USE_E_INVALID, "use_op_conv: bad dtype");
    const int Cin = c->C0 + c->C1, XC = c->XC0 + c->XC1, ntaps = c->ntaps == 1 ? 1 : 9;
    if (Cin % 32 != 0 || XC % 32 != 0 || (c->C1 && c->C0 % 32) || (c->XC1 && c->XC0 % 32)) return fail(USE_E_INVALID, "use_op_conv: channel counts must be multiples of 32 (zero-pad)");
    hipStream_t s = (hipStream_t)stream;
    const size_t es = dtype_size(dt);
    ConvW w; w.cin = Cin; w.cout = c->Cout; w.ntaps = ntaps; w.w_dtype = dt; w.cin_src = Cin; w.cout_src = c->Cout;
    w.cout_pad = c->Cout <= 32 ? 32 : (c->Cout + 127) / 128 * 128;
    ConvW w2 = w; w2.cin = XC; w2.cin_src = XC; w2.ntaps = 1;
    std::vector<void*> bufs;
    auto dalloc = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, bytes) != hipSuccess) return nullptr; bufs.push_back(q); return q; };
    auto cleanup = [&]() { for (void* q : bufs) (void)hipFree(q); };
    const size_t wbytes = (size_t)ntaps * w.cout_pad * Cin * es, w2bytes = (size_t)w2.cout_pad * std::max(XC, 1) * es;
    const bool slab = ntaps == 9 && w.cout_pad % 128 == 0 && Cin % conv_v4_chunk(dt) == 0;
    const bool slab2 = XC > 0 && XC % conv_v4_chunk(dt) == 0 && w2.cout_pad % 128 == 0;
    char* dw = (char*)dalloc(wbytes); char* dwb = slab ? (char*)dalloc(wbytes) : nullptr;
    char* dw2 = XC ? (char*)dalloc(w2bytes) : nullptr; char* dw2b = slab2 ? (char*)dalloc(w2bytes) : nullptr;
    float* dbias = (float*)dalloc((size_t)w.cout_pad * 4);
    if (!dw || !dbias || (slab && !dwb) || (XC && !dw2) || (slab2 && !dw2b)) { cleanup(); return fail(USE_E_NOMEM, "use_op_conv: allocation failed"); }
    {
        std::vector<char> hp(wbytes), hpb(slab ? wbytes : 0);
        pack_conv_raw(c->w, w, hp.data(), slab ? hpb.data() : nullptr);
        (void)hipMemcpy(dw, hp.data(), wbytes, hipMemcpyHostToDevice);
        if (slab) (void)hipMemcpy(dwb, hpb.data(), wbytes, hipMemcpyHostToDevice);
        if (XC) {
            if (!c->w2 || !c->x0) { cleanup(); return fail(USE_E_INVALID, "use_op_conv: shortcut without weights / input"); }
            std::vector<char> hq(w2bytes), hqb(slab2 ? w2bytes : 0);
            pack_conv_raw(c->w2, w2, hq.data(), slab2 ? hqb.data() : nullptr);
            (void)hipMemcpy(dw2, hq.data(), w2bytes, hipMemcpyHostToDevice);
            if (slab2) (void)hipMemcpy(dw2b, hqb.data(), w2bytes, hipMemcpyHostToDevice);
        }
        std::vector<float> hb((size_t)w.cout_pad, 0.f);
        if (c->bias) memcpy(hb.data(), c->bias, (size_t)c->Cout * 4);
        (void)hipMemcpy(dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    }
    ConvArgs a{};
    a.B = c->B; a.H = c->H; a.W = c->W; a.Cout = c->Cout; a.ntaps = ntaps; a.in_dtype = dt; a.out_dtype = odt;
    a.src0 = c->src0; a.src1 = c->C1 ? c->src1 : nullptr; a.C0 = c->C0; a.C1 = c->C1; a.coef = c->coef; a.act = c->act;
    a.w = dw; a.wb = dwb; a.cout_pad = w.cout_pad;
    a.x0 = XC ? c->x0 : nullptr; a.x1 = c->XC1 ? c->x1 : nullptr; a.XC0 = c->XC0; a.XC1 = c->XC1; a.w2 = dw2; a.w2b = dw2b;
    a.bias = dbias; a.temb = c->temb; a.temb_bstride = c->Cout; a.res = c->res; a.out_scale = c->out_scale;
    a.out = c->out; a.stats = c->stats;
    int rc = USE_OK;
    switch (c->variant) {
        case 0: launch_conv(a, s); break;
        case 1: launch_conv_generic(a, s); break;
        case 2: if (!conv_v2_eligible(a)) rc = fail(USE_E_INVALID, "conv_v2 cannot run this case"); else launch_conv_v2(a, s); break;
        case 4: if (!slab || (XC && !slab2) || a.H % 16 || a.W % 32 || dt != odt) rc = fail(USE_E_INVALID, "conv_v4 cannot run this case"); else launch_conv_v4(a, s); break;
        case 7: if (!conv_sk_eligible(a)) rc = fail(USE_E_INVALID, "conv_sk cannot run this case"); else launch_conv_sk(a, s); break;
        default: rc = fail(USE_E_INVALID, "use_op_conv: unknown variant %d", c->variant);
    }
    if (hipStreamSynchronize(s) != hipSuccess && rc == USE_OK) rc = fail(USE_E_HIP, "use_op_conv: %s", hipGetErrorString(hipGetLastError()));
    cleanup();
    return rc;
}
// use_op_conv with the weights already in HBM (fp32 parameter tensors: the training path, where they change every step) and a
// caller-provided workspace: the weights are laid out by a kernel on `stream`, nothing is allocated and nothing synchronises.
static size_t conv_dev_layout(const use_conv_op* c, size_t* wbytes, bool* slab, int* cout_pad) {
    const int Cin = c->C0 + c->C1, ntaps = c->ntaps == 1 ? 1 : 9;
    *cout_pad = c->Cout <= 32 ? 32 : (c->Cout + 127) / 128 * 128;
    *wbytes = ((size_t)ntaps * *cout_pad * Cin * dtype_size(c->dtype) + 255) / 256 * 256;
    *slab = ntaps == 9 && *cout_pad % 128 == 0 && Cin % conv_v4_chunk(c->dtype) == 0;
    return *wbytes * (*slab ? 2 : 1) + (size_t)*cout_pad * 4;
}
size_t use_op_conv_dev_workspace(const use_conv_op* c) {
    if (!c || c->C0 < 1 || c->Cout < 1) return 0;
    size_t wb; bool slab; int cp;
    return conv_dev_layout(c, &wb, &slab, &cp);
}
int use_op_conv_dev(const use_conv_op* c, int w_mode, void* work, size_t work_bytes, use_stream_t stream) {
    if (!c || c->B < 1 || c->H < 1 || c->W < 1 || c->C0 < 1 || c->Cout < 1 || !c->src0 || !c->w || !c->out || !work) return fail(USE_E_INVALID, "use_op_conv_dev: bad argument");
    const int dt = c->dtype, odt = c->out_dtype;
    if ((dt != DT_F32 && dt != DT_BF16 && dt != DT_F16) || (odt != DT_F32 && odt != DT_BF16 && odt != DT_F16)) return fail(USE_E_INVALID, "use_op_conv_dev: bad dtype");
    if (w_mode < 0 || w_mode > 2 || (w_mode == 2 && c->ntaps != 1)) return fail(USE_E_INVALID, "use_op_conv_dev: bad weight layout %d", w_mode);
    const int Cin = c->C0 + c->C1, ntaps = c->ntaps == 1 ? 1 : 9;
    if (Cin % 32 != 0 || (c->C1 && c->C0 % 32)) return fail(USE_E_INVALID, "use_op_conv_dev: channel counts must be multiples of 32 (zero-pad)");
    if (c->XC0 || c->XC1 || c->x0 || c->w2) return fail(USE_E_INVALID, "use_op_conv_dev: no fused shortcut in this form");
    size_t wbytes; bool slab; int cout_pad;
    if (conv_dev_layout(c, &wbytes, &slab, &cout_pad) > work_bytes) return fail(USE_E_INVALID, "use_op_conv_dev: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* dw = (char*)work; char* dwb = slab ? dw + wbytes : nullptr;
    float* dbias = (float*)(dw + wbytes * (slab ? 2 : 1));
    launch_pack_conv_dev((const float*)c->w, w_mode, c->Cout, Cin, ntaps, cout_pad, dt, conv_v4_chunk(dt), dw, dwb, (const float*)c->bias, dbias, s);
    ConvArgs a{};
    a.B = c->B; a.H = c->H; a.W = c->W; a.Cout = c->Cout; a.ntaps = ntaps; a.in_dtype = dt; a.out_dtype = odt;
    a.src0 = c->src0; a.src1 = c->C1 ? c->src1 : nullptr; a.C0 = c->C0; a.C1 = c->C1; a.coef = c->coef; a.act = c->act;
    a.w = dw; a.wb = dwb; a.cout_pad = cout_pad;
    a.bias = dbias; a.temb = c->temb; a.temb_bstride = c->Cout; a.res = c->res; a.out_scale = c->out_scale;
    a.out = c->out; a.stats = c->stats;
    launch_conv(a, s);
    HIPCHK(hipGetLastError());
    return USE_OK;
}
int use_op_fir(const void* src, int dtype, const float* coef, int act, void* out_act, void* out_raw, int B, int H, int W, int C, int up,
               use_stream_t stream) {
    if (!src || (!out_act && !out_raw) || B < 1 || H < 1 || W < 1 || C < 1) return fail(USE_E_INVALID, "use_op_fir: bad argument");
    if (up) launch_fir_up2(src, dtype, coef, act, out_act, out_raw, B, H, W, C, (hipStream_t)stream);
    else    launch_fir_down2(src, dtype, coef, act, out_act, out_raw, B, H, W, C, (hipStream_t)stream);
    return USE_OK;
}
int use_op_attention(const void* q, const void* k, const void* v, void* out, int dtype, int B, int N, int C, use_stream_t stream) {
    if (!q || !k || !v || !out || B < 1 || N < 1 || C < 1) return fail(USE_E_INVALID, "use_op_attention: bad argument");
    launch_attention(q, k, v, out, dtype, B, N, C, (hipStream_t)stream);
    return USE_OK;
}
int use_op_gn_finalize(const long long* st0, int C0, const long long* st1, int C1, const float* gamma, const float* beta, int groups,
                       int hw, float eps, float* coef, int B, use_stream_t stream) {
    if (!st0 || !gamma || !beta || !coef || C0 < 1 || groups < 1 || (C0 + C1) % groups) return fail(USE_E_INVALID, "use_op_gn_finalize: bad argument");
    launch_gn_finalize(st0, C0, C1 ? st1 : nullptr, C1, gamma, beta, groups, hw, eps, coef, B, (hipStream_t)stream);
    return USE_OK;
}

// ---- backward operators (fp32 NHWC device tensors; SURVEY 8f4 minimum slice: one res-block) ----
size_t use_op_wgrad_workspace(int B, int H, int W, int Cout, int Cin, int ntaps, int dtype) {
    return (B < 1 || H < 1 || W < 1 || Cout < 1 || Cin < 1) ? 0 : wgrad_workspace_floats(B, H, W, Cout, Cin, ntaps == 1 ? 1 : 9, dtype);
}
int use_op_wgrad(const void* dy, const void* x, int dtype, float* dw, float* db, int B, int H, int W, int Cout, int Cin, int ntaps, float alpha,
                 float* work, size_t work_floats, use_stream_t stream) {
    if (!dy || !x || !dw || B < 1 || H < 1 || W < 1 || Cout < 1 || Cin < 1 || (ntaps != 1 && ntaps != 9)) return fail(USE_E_INVALID, "use_op_wgrad: bad argument");
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_F16) return fail(USE_E_INVALID, "use_op_wgrad: bad dtype");
    if (work && work_floats < wgrad_workspace_floats(B, H, W, Cout, Cin, ntaps, dtype)) return fail(USE_E_INVALID, "use_op_wgrad: workspace too small");
    if (!launch_wgrad(dy, x, dtype, dw, db, B, H, W, Cout, Cin, ntaps, alpha, (work && work_floats) ? work : nullptr, (hipStream_t)stream))
        return fail(USE_E_INVALID, "use_op_wgrad: 16-bit inputs need the workspace and channel counts that are multiples of 4");
    HIPCHK(hipGetLastError());
    return USE_OK;
}
// workspace of the two GroupNorm operators, in floats (8-byte aligned): [fp64 slice partials][mean, rstd][s1, s2][m1, m2]
size_t use_op_gn_workspace(int B, int C, int groups) { return (B < 1 || C < 1 || groups < 1) ? 0 : gn_workspace_floats(B, C, groups); }
int use_op_gn_act_bwd(const void* x, const void* dy, int dtype, const float* gamma, const float* beta, int groups, float eps, int act, const void* add,
                      float add_scale, int B, int HW, int C, float* work, int have_stats, void* dx, float* dgamma, float* dbeta, use_stream_t stream) {
    if (!x || !dy || !gamma || !beta || !work || !dx || !dgamma || !dbeta || groups < 1 || C % groups) return fail(USE_E_INVALID, "use_op_gn_act_bwd: bad argument");
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_F16) return fail(USE_E_INVALID, "use_op_gn_act_bwd: bad dtype");
    if ((uintptr_t)work % 8) return fail(USE_E_INVALID, "use_op_gn_act_bwd: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)work;
    float* mean = work + (size_t)4 * B * GN_MAX_SLICES * std::max(C, groups); float* rstd = mean + (size_t)B * groups;
    float* s1 = rstd + (size_t)B * groups; float* s2 = s1 + (size_t)B * C; float* m12 = s2 + (size_t)B * C;
    // have_stats: `work` is the workspace use_op_gn_act_fwd ran in for the same x - its mean / rstd are reused, one pass over x saved
    if ((!have_stats && !launch_gn_stats(x, dtype, B, HW, C, groups, eps, mean, rstd, part, s)) ||
        !launch_gn_act_bwd(x, dy, dtype, mean, rstd, gamma, beta, act, add, add_scale, B, HW, C, groups, s1, s2, m12, part, dx, dgamma, dbeta, s))
        return fail(USE_E_INVALID, "use_op_gn_act_bwd: 16-bit tensors need C to be a multiple of 8 (and C / 8 <= 256)");
    HIPCHK(hipGetLastError());
    return USE_OK;
}
int use_op_gn_act_fwd(const void* x, int dtype, const float* gamma, const float* beta, int groups, float eps, int act, int B, int HW, int C, float* work,
                      void* y, use_stream_t stream) {
    if (!x || !gamma || !beta || !work || !y || groups < 1 || C % groups) return fail(USE_E_INVALID, "use_op_gn_act_fwd: bad argument");
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_F16) return fail(USE_E_INVALID, "use_op_gn_act_fwd: bad dtype");
    if ((uintptr_t)work % 8) return fail(USE_E_INVALID, "use_op_gn_act_fwd: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)work;
    float* mean = work + (size_t)4 * B * GN_MAX_SLICES * std::max(C, groups); float* rstd = mean + (size_t)B * groups;
    if (!launch_gn_stats(x, dtype, B, HW, C, groups, eps, mean, rstd, part, s) ||
        !launch_gn_act_fwd(x, dtype, mean, rstd, gamma, beta, act, B, HW, C, groups, y, s))
        return fail(USE_E_INVALID, "use_op_gn_act_fwd: 16-bit tensors need C to be a multiple of 8 (and C / 8 <= 256)");
    HIPCHK(hipGetLastError());
    return USE_OK;
}
int use_op_colsum(const void* x, int dtype, int B, int HW, int C, float scale, float* out, float* work, use_stream_t stream) {
    if (!x || !out || B < 1 || HW < 1 || C < 1) return fail(USE_E_INVALID, "use_op_colsum: bad argument");
    if (dtype != DT_F32 && dtype != DT_BF16 && dtype != DT_F16) return fail(USE_E_INVALID, "use_op_colsum: bad dtype");
    if (work && (uintptr_t)work % 8) return fail(USE_E_INVALID, "use_op_colsum: workspace must be 8-byte aligned");
    if (!launch_colsum(x, dtype, B, HW, C, scale, out, (double*)work, (hipStream_t)stream))
        return fail(USE_E_INVALID, "use_op_colsum: 16-bit tensors need the workspace (128 B C floats) and C a multiple of 8");
    HIPCHK(hipGetLastError());
    return USE_OK;
}
int use_op_attention_bwd(const float* q, const float* k, const float* v, const float* dO, float* work, float* dq, float* dk, float* dv, int B, int N,
                         int C, use_stream_t stream) {
    if (!q || !k || !v || !dO || !work || !dq || !dk || !dv || B < 1 || N < 1 || C < 1) return fail(USE_E_INVALID, "use_op_attention_bwd: bad argument");
    if ((size_t)(2 * N + 2 * C) * 4 > 60000) return fail(USE_E_INVALID, "use_op_attention_bwd: N + C too large for one row in LDS");
    launch_attention_bwd(q, k, v, dO, work, dq, dk, dv, B, N, C, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return USE_OK;
}
int use_op_dense_bwd(const float* g, const float* temb, const float* Wd, int B, int K, int Cout, float* dW, float* db, float* dtemb, use_stream_t stream) {
    if (!g || !temb || !Wd || !dW || !db || !dtemb) return fail(USE_E_INVALID, "use_op_dense_bwd: null tensor");
    launch_dense_bwd(g, temb, Wd, B, K, Cout, dW, db, dtemb, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return USE_OK;
}

int use_debug_tensor(use_handle* h, const char* name, void** dev_ptr, int* dims4, int* dtype) {
    if (!h || !name) return fail(USE_E_INVALID, "null argument");
    auto it = h->debug.find(name);
    if (it == h->debug.end()) return fail(USE_E_INVALID, "no debug tensor '%s'", name);
    if (dev_ptr) *dev_ptr = it->second.p;
    if (dims4) { dims4[0] = h->debug_B; dims4[1] = it->second.H; dims4[2] = it->second.W; dims4[3] = it->second.C; }
    if (dtype) *dtype = it->second.dtype;
    return USE_OK;
}

}  // extern "C"
