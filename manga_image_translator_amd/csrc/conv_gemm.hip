// conv_gemm.hip — configuration table, tile choice, kernel-time probe and C entry points of mit_conv_gemm.
// The kernels are in conv_gemm_kernels.h; their instantiations are compiled in conv_gemm_inst<group>.hip.
#include "conv_gemm_kernels.h"
#include <string.h>
#include <string>
#include <atomic>

using namespace mitcg;

#define X(g, name, fast, BM, BN, BK, fn, ...) \
    extern template void mitcg::fn<BM, BN, BK, __VA_ARGS__>(const MitConvGemm &, int, int, int, int, hipStream_t);
#include "conv_gemm_cfgs.inc"
#undef X

namespace {

const CfgEntry kCfgs[] = {
#define KNAME_launch_cfg "conv_gemm_kernel"
#define KNAME_launch_fast "conv_gemm_fast_kernel"
#define KNAME_launch_gemv "conv_gemv_kernel"
#define KNAME_launch_split "conv_gemm_split_kernel"
#define KNAME_launch_split_pp "conv_gemm_split_pp_kernel"
#define X(g, name, fast, BM, BN, BK, fn, ...) \
    {name, BM, BN, BK, fn<BM, BN, BK, __VA_ARGS__>, fast, KNAME_##fn "<" #BM ", " #BN ", " #BK ", " #__VA_ARGS__ ">"},
#include "conv_gemm_cfgs.inc"
#undef X
};

// fast kernel preconditions: whole K-tiles inside one tap, table fits, 32-bit element offsets
bool fast_eligible(const MitConvGemm &p, int BK) {
    if (p.Cin % BK || p.ntaps > FAST_MAX_TAPS) return false;
    int64_t maxoff = (int64_t)(p.NB - 1) * p.a_bs + (int64_t)(p.Hi - 1) * p.a_ys + (int64_t)(p.Wi - 1) * p.a_xs + p.Cin;
    int64_t minoff = 0;
    if (p.a_bs < 0 || p.a_ys < 0 || p.a_xs < 0) return false;
    int tmax = 0;
    for (int t = 0; t < p.ntaps; ++t) {
        if (p.tap_off[t] < 0) return false;
        if (p.tap_off[t] > tmax) tmax = p.tap_off[t];
    }
    (void)minoff;
    return maxoff + tmax < 0x7fffffffLL;
}
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

int cfg_by_name(const char *name) {
    for (int i = 0; i < (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); ++i)
        if (!strcmp(kCfgs[i].name, name)) return i;
    return -1;
}

// conv_gemm_split_kernel preconditions: the fast kernel's, plus split planes of W laid out by mit_gemm_split_pack
bool split_eligible(const MitConvGemm &p, int BK) {
    if (!p.w_split || (reinterpret_cast<uintptr_t>(p.w_split) & 15) || (p.ws_zs0 & 7)) return false;
    if (p.w_zs1 != 0 || (p.Kw & 7) || p.ntaps * p.Cin > p.Kw) return false;
    if ((int64_t)3 * (p.Kw >> 3) * p.ldw > 0x7fffffffLL) return false;  // 32-bit cell indices
    return fast_eligible(p, BK);
}

// the "u" tiles (operand loads through buffer instructions: 32-bit byte offsets against a 2 GB descriptor, conv_gemm_split.h VAR bit
// 2048): every byte offset of A — relative to the slice base the kernel forms — and of the packed W planes must stay below 2^31
bool buf_eligible(const MitConvGemm &p) {
    if (p.a_bs < 0 || p.a_ys < 0 || p.a_xs < 0) return false;
    int tmax = 0;
    for (int t = 0; t < p.ntaps; ++t)
        if (p.tap_off[t] > tmax) tmax = p.tap_off[t];
    const int64_t maxoff = (int64_t)(p.NB - 1) * p.a_bs + (int64_t)(p.Hi - 1) * p.a_ys + (int64_t)(p.Wi - 1) * p.a_xs + p.Cin + tmax;
    if (maxoff * 4 >= 0x80000000LL) return false;
    return (int64_t)3 * (p.Kw >> 3) * p.ldw * 16 < 0x80000000LL;
}
// the buffer-load twin of a shipped p6 tile ("...p6o" -> "...p6u"), or the tile itself (MIT_CONV_NO_BUF=1: A/B knob)
int buf_twin(int c) {
    static const std::vector<int> twin = [] {
        std::vector<int> t(kNumCfgs);
        const bool off = getenv("MIT_CONV_NO_BUF") != nullptr;
        for (int i = 0; i < kNumCfgs; ++i) {
            t[i] = i;
            const std::string n = kCfgs[i].name;
            if (!off && n.size() > 3 && n.compare(n.size() - 3, 3, "p6o") == 0) {
                const int u = cfg_by_name((n.substr(0, n.size() - 1) + "u").c_str());
                if (u >= 0) t[i] = u;
            }
        }
        return t;
    }();
    return c >= 0 && c < kNumCfgs ? twin[c] : c;
}

// conv_gemv_kernel preconditions: <= 4 output columns, plain (unbatched, unsplit) maps, the whole weight panel in LDS
bool gemv_eligible(const MitConvGemm &p, int lpr) {
    if (p.N > 4 || p.Z != 1 || p.Cin % (4 * lpr) || (int64_t)p.NB * p.Ho > 65535) return false;  // one output row per blockIdx.y
    if ((int64_t)p.ntaps * p.Cin * (p.N == 1 ? 1 : 4) * 4 > 60 * 1024) return false;            // the transposed weight panel must fit the default dynamic-LDS limit
    if (p.c.nsplit || p.pre.nsplit || p.post.nsplit || p.lut_rows) return false;
    return true;
}

// GEMM mode (include/mit_hip.h, mit_gemm_mode_set): -1 = not read yet
std::atomic<int> g_gemm_mode{-1};
int gemm_mode_now() {
    int m = g_gemm_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *v = getenv("MIT_GEMM_SPLIT");
        m = (v && *v) ? atoi(v) : 6;
        if (m != 6 && m != 9) m = 0;
        g_gemm_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

// smallest launch (in 128 x 64 tiles) the automatic choice gives to the split tiles.  Default 0: every eligible launch takes them, so
// that a layer's arithmetic — and with it a page's result — does not depend on how many pages share the batch (a size threshold made
// the same layer run on the fp32 tiles at B = 1 and on the split tiles at B = 16).  -1 = not read yet (MIT_GEMM_SPLIT_MIN_TILES)
std::atomic<long long> g_split_min{-1};
int64_t split_min_now() {
    long long m = g_split_min.load(std::memory_order_relaxed);
    if (m < 0) {
        const char *v = getenv("MIT_GEMM_SPLIT_MIN_TILES");
        m = (v && *v) ? atoll(v) : 0;
        if (m < 0) m = 0;
        g_split_min.store(m, std::memory_order_relaxed);
    }
    return m;
}

int env_cfg(const char *name, const char *dflt_name) {  // tuning knob for scripts/: replaces a default fast tile by another fast tile, BY NAME
    const int dflt = dflt_name ? cfg_by_name(dflt_name) : -1;  // -1: rule off
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    const int c = cfg_by_name(v);
    return (c >= 0 && (kCfgs[c].fast == 1 || kCfgs[c].fast == 2) && kCfgs[c].BK == 16) ? c : dflt;
}

int pick_cfg(const MitConvGemm &p, int64_t M) {
    // measured on MI355X (scripts/bench_conv.py)
    static const int wide = env_cfg("MIT_CONV_TILE_WIDE", "fast128x128x16w4c"), narrow = env_cfg("MIT_CONV_TILE_NARROW", "fast128x64x16w5c");
    static const int m192 = env_cfg("MIT_CONV_TILE_M192", "fast192x64x16w4c"), bigk = env_cfg("MIT_CONV_TILE_BIGK", nullptr);
    static const int wide_l = env_cfg("MIT_CONV_TILE_WIDE_L", nullptr), narrow_l = env_cfg("MIT_CONV_TILE_NARROW_L", nullptr);  // experiments: Cin % 32 == 0
    static const int kCfgGemv16 = cfg_by_name("gemv16"), kCfgGemv4 = cfg_by_name("gemv4"), kCfgGemv16N1 = cfg_by_name("gemv16n1"), kCfgGemv4N1 = cfg_by_name("gemv4n1");
    static const int kCfgSmall = cfg_by_name("fast64x64x16w8c"), kCfgGen128 = cfg_by_name("128x128x16"), kCfgGen64 = cfg_by_name("128x64x16"), kCfgGen32 = cfg_by_name("128x32x16");
    static const int narrow_max = getenv("MIT_CONV_NARROW_MAX") ? atoi(getenv("MIT_CONV_NARROW_MAX")) : 64;
    const bool f16 = fast_eligible(p, 16);
    static const bool gemv_off = getenv("MIT_CONV_NO_GEMV") != nullptr;  // A/B knob for scripts/
    if (!gemv_off && gemv_eligible(p, 16)) return p.N == 1 ? kCfgGemv16N1 : kCfgGemv16;
    if (!gemv_off && gemv_eligible(p, 4)) return p.N == 1 ? kCfgGemv4N1 : kCfgGemv4;
    if (p.N <= 32) {  // ESRGAN's growth-32 convolutions, small heads: a 128 x 32 tile on the best kernel the layer is eligible for
        static const int n32_split6 = cfg_by_name("split128x32x16p6o"), n32_split9 = cfg_by_name("split128x32x16p9m");
        static const int n32_fast = getenv("MIT_CONV_NO_N32_FAST") ? -1 : cfg_by_name("fast128x32x16w4c");
        const int sp = gemm_mode_now();
        if ((sp == 6 || sp == 9) && p.w_split && split_eligible(p, 16) && ((M + 127) / 128) * p.Z >= split_min_now()) {
            const int c = sp == 6 ? n32_split6 : n32_split9;
            if (c >= 0) return buf_eligible(p) ? buf_twin(c) : c;
        }
        if (f16 && n32_fast >= 0) return n32_fast;
        return kCfgGen32;
    }
    // split-bf16 tiles (GEMM mode 6 | 9, mit_gemm_mode_set): layers whose packer attached split planes of W, large enough to fill the chip
    const int split = gemm_mode_now();
    const int64_t split_min = split_min_now();
    const int64_t tiles128 = ((M + 127) / 128) * ((p.N + 63) / 64);
    if ((split == 6 || split == 9) && p.w_split && split_eligible(p, 16) && tiles128 * p.Z >= split_min) {
        static const int wide6 = cfg_by_name("split128x128x16p6o"), wide9 = cfg_by_name("split128x128x16p9m");
        static const int narrow6 = cfg_by_name("split128x64x16p6o"), narrow9 = cfg_by_name("split128x64x16p9");
        static const int small6 = getenv("MIT_CONV_NO_SMALL_TILE") ? -1 : cfg_by_name("split64x64x16p6o");
        static const int small9 = getenv("MIT_CONV_NO_SMALL_TILE") ? -1 : cfg_by_name("split64x64x16p9m");
        static const int64_t ssmall_max = getenv("MIT_CONV_SPLIT_SMALL_MAX") ? atoll(getenv("MIT_CONV_SPLIT_SMALL_MAX")) : 768;  // one wave of 128-row split tiles (3 workgroups per CU)
        const int r = p.N % 128;
        int c = (p.N <= 64 || (r != 0 && r <= 64)) ? (split == 6 ? narrow6 : narrow9) : (split == 6 ? wide6 : wide9);
        // Exact-N tiles (round 5; wave tile 32 x BN, the A tile split once for all BN columns) where the 64-column tile would otherwise
        // run 3 or 5 times over the same rows, or the 128-column tile would compute 48 padded columns — measured per shape
        // (profiles/r07f_split_check_exact_n_tiles.log, r07g_split_check_tile192.log; same bits as every other p6 tile):
        //   N = 160, K = 640 (ConvNeXt stage-2 pw2): 1.36x of 3 x 64;   N = 320, K = 1280 (stage-3 pw2): 1.13x of 5 x 64;
        //   N = 80, K = 320 (stage-1 pw2): 1.08x of the 128-column tile;   N = 192, K = 384 (LaMa spectral conv1): 1.09x of 3 x 64.
        // The short-K expansions (pw1: K = 80 / 160 / 320 into N = 4K) are 5-10 % SLOWER on them and keep the tiles above.
        static const bool exact_n_off = getenv("MIT_CONV_NO_EXACT_N") != nullptr;  // A/B knob
        if (split == 6 && p.Z == 1 && !exact_n_off) {
            static const int t160 = cfg_by_name("split128x160x16p6o"), t96 = cfg_by_name("split128x96x16p6o"), t192 = cfg_by_name("split128x192x16p6o");
            const int K = p.ntaps * p.Cin;
            if (t160 >= 0 && (p.N == 160 || p.N == 320) && K >= 512) c = t160;
            else if (t96 >= 0 && p.N > 64 && p.N <= 96 && K >= 256) c = t96;
            else if (t192 >= 0 && p.N == 192 && K >= 256) c = t192;
        }
        // Ping-pong tile (round 6; conv_gemm_split_pp.h: eight waves, one workgroup per CU, compute and load segments alternating on
        // every SIMD): 1.12-1.15x of the 128 x 128 tile on long-K layers whose N is a multiple of 256 (LaMa's stride-2 and transposed
        // convolutions at 256 / 512 channels, the detector's 2048 -> 256 layers), 0.8-1.0x on short K (an 8-32 step loop does not
        // amortise a lone workgroup's prologue and 128 KB epilogue) — profiles/r10d_split_check_pp.log.  Same bits as every p6 tile.
        // OPT-IN (MIT_CONV_PP=1): those layers are 7 % of a step's GEMM time, the step gains 0.6 % (inside the box-to-box noise,
        // profiles/r10e_ab_pp_end_to_end.log), and they are the best launches of the tile the bench's roofline is priced on.
        static const bool pp_on = getenv("MIT_CONV_PP") != nullptr && atoi(getenv("MIT_CONV_PP")) != 0;
        static const int pp256 = cfg_by_name("split128x256x16p6pp");
        if (split == 6 && p.Z == 1 && pp_on && pp256 >= 0 && p.N % 256 == 0 && p.ntaps * p.Cin >= 1024 && ((M + 127) / 128) * (p.N / 256) >= 512) c = pp256;
        // under-filled launches (one page through the plugins, the decoder's Linears): 64 x 64 tiles quadruple the workgroup count;
        // the arithmetic per output element is that of the large tiles, so a result does not depend on the choice
        const int sm = split == 6 ? small6 : small9;
        if (sm >= 0 && p.Z == 1 && tiles128 < ssmall_max) {
            c = sm;
            // launches of at most two workgroups per CU (one page through the plugins: the decoder at M = 160 rows, the detector's deep
            // layers) are bound by the latency of a K-loop iteration, not by its throughput: two MFMA steps per barrier (BK = 32) take
            // 10-22 % off them and cost 2 % on fuller launches (profiles/r03l_split_check_bk32.log).  Same MFMA sequence per element.
            static const int small6k = getenv("MIT_CONV_NO_SMALL_BK32") ? -1 : cfg_by_name("split64x64x32p6o");
            static const int small9k = getenv("MIT_CONV_NO_SMALL_BK32") ? -1 : cfg_by_name("split64x64x32p9m");
            const int smk = split == 6 ? small6k : small9k;
            if (smk >= 0 && ((M + 63) / 64) * ((p.N + 63) / 64) <= 512 && split_eligible(p, 32)) c = smk;
        }
        if (c >= 0) return buf_eligible(p) ? buf_twin(c) : c;  // same arithmetic, operand loads through buffer instructions where offsets fit
    }
    if (f16 && m192 >= 0 && M > 128 && M <= 192 && p.Z >= 8) return m192;  // batched launches (Z entries of M = 184 rows: W-axis DFTs): 2 x 128 rows would run a 40 % empty second tile.  Unbatched, one row of 192 x 64 tiles leaves the chip empty (the decoder at B = 1: M = 160)
    if (f16 && bigk >= 0 && p.N % 128 == 0 && p.N <= 128 && p.ntaps * p.Cin >= 4096 && M >= 256 * 1024) return bigk;
    // under-filled launches (the decoder's GEMMs: M = lines x beams = 10240): a 128-row tiling leaves most CUs with one workgroup or
    // none, 64 x 64 tiles double the count
    static const int small = getenv("MIT_CONV_NO_SMALL_TILE") ? -1 : kCfgSmall;
    static const int64_t small_max = getenv("MIT_CONV_SMALL_MAX") ? atoll(getenv("MIT_CONV_SMALL_MAX")) : 1280;  // swept 640 .. 5120 on the OCR and detector stages (same-box A/B): 1280 = one full wave of workgroups
    if (f16 && small >= 0 && p.Z == 1 && p.N > 32 && ((M + 127) / 128) * ((p.N + 63) / 64) < small_max) return small;
    const int rem = p.N % 128;
    const bool lines = f16 && p.Cin % 32 == 0;
    if (p.N <= 64 || (rem != 0 && rem <= narrow_max)) return f16 ? (lines && narrow_l >= 0 ? narrow_l : narrow) : kCfgGen64;  // e.g. N = 192: 3 x 64 beats 2 x 128 with a half-empty tile
    return f16 ? (lines && wide_l >= 0 ? wide_l : wide) : kCfgGen128;  // 4 waves of 128 x 32, <= 128 registers: 4 workgroups per CU (+3-7 % over the 2 x 2 layout)
}

// ---- kernel-time probe (mit_prof_*): HIP events around every launch while enabled ----
struct ProbeRec {
    hipEvent_t start, stop;
    int cfg;
    double exec_flops, alg_flops;
    int M, N, K, ntaps, Z, act;
};
std::mutex g_probe_mu;
bool g_probe_on = false;
std::vector<ProbeRec> g_probe;
thread_local double g_next_alg_flops = -1.0;

}  // namespace

extern "C" int mit_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_probe_mu);
    for (auto &r : g_probe) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    g_probe.clear();
    g_probe_on = on != 0;
    mit_probe_reset(on != 0);
    return 0;
}

extern "C" int mit_prof_tag_next(double alg_flops) {
    g_next_alg_flops = alg_flops;
    return 0;
}

extern "C" int mit_prof_dump(const char *path) {
    if (!path) return mit_set_error("mit_prof_dump: null path");
    std::lock_guard<std::mutex> lk(g_probe_mu);
    FILE *f = fopen(path, "w");
    if (!f) return mit_set_error("mit_prof_dump: cannot open %s", path);
    fprintf(f, "tile,M,N,K,taps,Z,act,ms,exec_flops,alg_flops\n");
    for (auto &r : g_probe) {
        float ms = 0.f;
        if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) {
            fclose(f);
            return mit_set_error("mit_prof_dump: event query failed");
        }
        fprintf(f, "%s,%d,%d,%d,%d,%d,%d,%.6f,%.0f,%.0f\n", kCfgs[r.cfg].name, r.M, r.N, r.K, r.ntaps, r.Z, r.act, ms, r.exec_flops,
                r.alg_flops);
    }
    fclose(f);
    return 0;
}

extern "C" int mit_prof_read(MitProfStat *stats, int max_cfgs, int *n_cfgs) {
    if (!stats || !n_cfgs) return mit_set_error("mit_prof_read: null");
    std::lock_guard<std::mutex> lk(g_probe_mu);
    const int n = max_cfgs < kNumCfgs ? max_cfgs : kNumCfgs;
    for (int i = 0; i < n; ++i) {
        stats[i].launches = 0;
        stats[i].ms = stats[i].exec_flops = stats[i].alg_flops = 0.0;
    }
    for (auto &r : g_probe) {
        MIT_CHECK_HIP(hipEventSynchronize(r.stop));
        float ms = 0.f;
        MIT_CHECK_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
        if (r.cfg < n) {
            stats[r.cfg].launches += 1;
            stats[r.cfg].ms += ms;
            stats[r.cfg].exec_flops += r.exec_flops;
            stats[r.cfg].alg_flops += r.alg_flops;
        }
    }
    *n_cfgs = n;
    return 0;
}

extern "C" const char *mit_conv_gemm_config_name(int cfg) {
    if (cfg < 0 || cfg >= kNumCfgs) return nullptr;
    return kCfgs[cfg].name;
}

extern "C" const char *mit_conv_gemm_config_kernel(int cfg) {
    if (cfg < 0 || cfg >= kNumCfgs) return nullptr;
    return kCfgs[cfg].kernel;
}

namespace {
bool map_vec_ok(const MitTensorMap &m, bool split_ok = false) {
    if (m.nsplit != 0 && !(split_ok && !(m.nsplit & 3) && !(m.nhi & 3))) return false;  // (only the C map's float4 store handles a column split)
    return !(reinterpret_cast<uintptr_t>(m.base) & 15) && !((m.zs1 | m.zs0 | m.bs | m.ys | m.xs) & 3);
}
// dwordx4 epilogue (epilogue_store_vec): whole float4 column groups, contiguous and 16-byte aligned in every tensor it touches
bool vec_epilogue_ok(const MitConvGemm &p) {
    static const bool off = getenv("MIT_CONV_SCALAR_EPILOGUE") != nullptr;  // A/B knob for scripts/
    if (off || (p.N & 3) || !map_vec_ok(p.c, true)) return false;
    if (p.pre.base && !map_vec_ok(p.pre)) return false;
    if (p.post.base && !map_vec_ok(p.post)) return false;
    if (p.lut_rows && ((p.lut_ld & 3) || (reinterpret_cast<uintptr_t>(p.lut1) & 15) || (reinterpret_cast<uintptr_t>(p.lut2) & 15))) return false;
    return !(reinterpret_cast<uintptr_t>(p.scale) & 15) && !(reinterpret_cast<uintptr_t>(p.bias) & 15);
}
}  // namespace

extern "C" int mit_conv_gemm_cfg(const MitConvGemm *d, int cfg, void *stream) {
    if (!d) return mit_set_error("mit_conv_gemm: null descriptor");
    MitConvGemm pv = *d;
    if (pv.act & MIT_ACT_VEC_OK) return mit_set_error("mit_conv_gemm: reserved activation bits set");
    if (vec_epilogue_ok(pv)) pv.act |= MIT_ACT_VEC_OK;
    const MitConvGemm &p = pv;
    if (!p.a || !p.w || !p.c.base) return mit_set_error("mit_conv_gemm: null operand");
    if (p.Cin <= 0 || (p.Cin & 3)) return mit_set_error("mit_conv_gemm: Cin must be a positive multiple of 4 (got %d)", p.Cin);
    if ((p.ldw & 3) || (p.Nw & 3)) return mit_set_error("mit_conv_gemm: ldw/Nw must be multiples of 4 (ldw=%lld Nw=%d)", (long long)p.ldw, p.Nw);
    if (p.ntaps <= 0 || p.ntaps > MIT_MAX_TAPS) return mit_set_error("mit_conv_gemm: ntaps %d out of range", p.ntaps);
    if (p.NB <= 0 || p.Ho <= 0 || p.Wo <= 0 || p.N <= 0 || p.Z <= 0 || p.zdiv <= 0)
        return mit_set_error("mit_conv_gemm: empty problem");
    if ((reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.w) & 15))
        return mit_set_error("mit_conv_gemm: operands must be 16-byte aligned");
    if ((p.a_xs & 3) || (p.a_ys & 3) || (p.a_bs & 3) || (p.a_zs0 & 3) || (p.a_zs1 & 3) || (p.w_zs0 & 3) || (p.w_zs1 & 3))
        return mit_set_error("mit_conv_gemm: strides must be multiples of 4 elements");
    for (int t = 0; t < p.ntaps; ++t)
        if (p.tap_off[t] & 3) return mit_set_error("mit_conv_gemm: tap_off must be multiples of 4");
    if (p.pad_mode == MIT_PAD_REFLECT) {
        for (int t = 0; t < p.ntaps; ++t) {
            int ady = p.tap_dy[t] < 0 ? -p.tap_dy[t] : p.tap_dy[t];
            int adx = p.tap_dx[t] < 0 ? -p.tap_dx[t] : p.tap_dx[t];
            if (ady >= p.Hi || adx >= p.Wi) return mit_set_error("mit_conv_gemm: reflect pad larger than input");
        }
    }
    const int64_t M64 = (int64_t)p.NB * p.Ho * p.Wo;
    if (M64 > 0x7fffffffLL) return mit_set_error("mit_conv_gemm: M too large");
    if (p.lut_rows) {  // the row-lookup epilogue: both tables, rows long enough, one slice (lut_rows is indexed by the output row)
        if (!p.lut1 || !p.lut2 || p.lut_ld < p.N) return mit_set_error("mit_conv_gemm: lut_rows needs lut1, lut2 and lut_ld >= N");
        if (p.Z != 1) return mit_set_error("mit_conv_gemm: the row-lookup epilogue is for Z == 1 launches");
        if (p.lut_ld > 0x7fff) return mit_set_error("mit_conv_gemm: lut_ld too large (row offsets are 16-bit row x lut_ld in 32 bits)");
    }
    if (p.Z > 65535) return mit_set_error("mit_conv_gemm: Z too large");
    // The fast kernels index A with 32-bit element offsets.  A batch whose activations exceed 2^31 elements (16 pages of
    // 2048 x 1456 x 64: LaMa's first stride-2 conv) is cut into runs of whole images that fit, instead of falling to the generic kernel.
    // Round 6: in the split mode the runs are cut to what the buffer-load tiles address (2^31 BYTES per run: buf_eligible) when one
    // image fits that — every run is still thousands of workgroups, and each takes the "u" tile instead of its "o" twin.
    static const bool cut_for_buf = getenv("MIT_CONV_NO_BUF") == nullptr && getenv("MIT_CONV_NO_BUF_CUT") == nullptr;
    bool want_buf = false;
    if (cfg < 0 && cut_for_buf && p.Z == 1 && p.NB > 1 && gemm_mode_now() == 6 && p.w_split != nullptr && split_eligible(p, 16)) {
        MitConvGemm one = p;
        one.NB = 1;
        want_buf = buf_eligible(one);
    }
    auto run_ok = [&](const MitConvGemm &q) { return fast_eligible(q, 16) && (!want_buf || buf_eligible(q)); };
    if (cfg < 0 && p.Z == 1 && p.NB > 1 && p.Cin % 16 == 0 && p.ntaps <= FAST_MAX_TAPS && p.a_bs > 0 && !run_ok(p)) {
        MitConvGemm one = p;
        one.NB = 1;
        if (fast_eligible(one, 16)) {
            int nbc = p.NB;
            while (nbc > 1) {
                one.NB = nbc;
                if (run_ok(one)) break;
                nbc = (nbc + 1) / 2;
            }
            for (int b0 = 0; b0 < p.NB; b0 += nbc) {
                MitConvGemm sub = *d;
                sub.NB = p.NB - b0 < nbc ? p.NB - b0 : nbc;
                sub.a = d->a + (int64_t)b0 * d->a_bs;
                sub.c.base = d->c.base + (int64_t)b0 * d->c.bs;
                if (d->pre.base) sub.pre.base = d->pre.base + (int64_t)b0 * d->pre.bs;
                if (d->post.base) sub.post.base = d->post.base + (int64_t)b0 * d->post.bs;
                if (d->lut_rows) sub.lut_rows = d->lut_rows + (int64_t)b0 * p.Ho * p.Wo;
                if (g_next_alg_flops >= 0.0) g_next_alg_flops = -1.0;  // a tagged cost does not survive the split
                const int rc = mit_conv_gemm_cfg(&sub, -1, stream);
                if (rc) return rc;
            }
            return 0;
        }
    }
    if (cfg < 0) cfg = pick_cfg(p, M64);
    if (cfg >= kNumCfgs) return mit_set_error("mit_conv_gemm: bad cfg %d", cfg);
    const CfgEntry &c = kCfgs[cfg];
    if (p.lut_rows) {  // the row-lookup epilogue exists as an instantiation of the vector store path of the fast / split tiles only
        if (c.fast != 1 && c.fast != 2 && c.fast != 4)
            return mit_set_error("mit_conv_gemm: the row-lookup epilogue (lut_rows) needs a fast or split tile (Cin %% 16 == 0, <= %d taps); this launch takes %s", FAST_MAX_TAPS, c.name);
        if (!(p.act & MIT_ACT_VEC_OK)) return mit_set_error("mit_conv_gemm: lut_rows needs the float4 epilogue (N %% 4 == 0, 16-byte aligned maps, tables and lut_ld %% 4 == 0)");
        if (p.post.base) return mit_set_error("mit_conv_gemm: lut_rows together with a post residual is not implemented");
        if ((p.act & 0xff) != MIT_ACT_NONE && (p.act & 0xff) != MIT_ACT_RELU) return mit_set_error("mit_conv_gemm: lut_rows is implemented for act none / relu");
    }
    if (p.dyn && c.fast == 3) return mit_set_error("mit_conv_gemm: the device-side step offset (dyn) is not implemented by the N <= 4 kernel");
    if (p.dyn && ((p.a_dyn | p.c_dyn) & 3)) return mit_set_error("mit_conv_gemm: a_dyn / c_dyn must be multiples of 4 floats");
    if (c.fast == 2 && (p.Cin % 32)) return mit_set_error("mit_conv_gemm: cfg %s needs Cin %% 32 == 0", c.name);
    if (c.fast == 3) {
        const int lpr = (!strcmp(c.name, "gemv16") || !strcmp(c.name, "gemv16n1")) ? 16 : 4;
        if (!gemv_eligible(p, lpr) || p.N > c.BN)
            return mit_set_error("mit_conv_gemm: cfg %s needs N <= %d, Z == 1, unsplit maps and Cin %% %d == 0", c.name, c.BN, 4 * lpr);
    }
    if (c.fast == 4 && !split_eligible(p, c.BK))
        return mit_set_error("mit_conv_gemm: cfg %s needs w_split (mit_gemm_split_pack, 16-byte aligned, w_zs1 == 0, Kw %% 8 == 0) and the fast tiles' preconditions", c.name);
    if ((c.fast == 1 || c.fast == 2) && !fast_eligible(p, c.BK))
        return mit_set_error("mit_conv_gemm: cfg %s needs Cin %% %d == 0, <= %d taps and 32-bit element offsets", c.name, c.BK, FAST_MAX_TAPS);
    const int M = (int)M64;
    const int MT = (M + c.BM - 1) / c.BM;
    const int NT = (p.N + c.BN - 1) / c.BN;
    const int Ktot = p.ntaps * p.Cin;
    const int KT = (Ktot + c.BK - 1) / c.BK;
    if ((int64_t)MT * NT > 0x7fffffffLL) return mit_set_error("mit_conv_gemm: grid too large");
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    const double tagged = g_next_alg_flops;
    g_next_alg_flops = -1.0;
    if (g_probe_on) {
        std::lock_guard<std::mutex> lk(g_probe_mu);
        ProbeRec r;
        MIT_CHECK_HIP(hipEventCreate(&r.start));
        MIT_CHECK_HIP(hipEventCreate(&r.stop));
        r.cfg = cfg;
        r.exec_flops = 2.0 * (double)M * p.N * Ktot * p.Z;
        r.alg_flops = tagged >= 0.0 ? tagged : r.exec_flops;
        r.M = M, r.N = p.N, r.K = Ktot, r.ntaps = p.ntaps, r.Z = p.Z, r.act = p.act;
        MIT_CHECK_HIP(hipEventRecord(r.start, hs));
        c.launch(p, M, MT, NT, KT, hs);
        MIT_CHECK_HIP(hipEventRecord(r.stop, hs));
        g_probe.push_back(r);
    } else {
        c.launch(p, M, MT, NT, KT, hs);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mit_set_error("mit_conv_gemm: launch failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int mit_conv_gemm(const MitConvGemm *d, void *stream) { return mit_conv_gemm_cfg(d, -1, stream); }

extern "C" int mit_gemm_mode_set(int mode) {
    if (mode != 0 && mode != 6 && mode != 9) return mit_set_error("mit_gemm_mode_set: mode must be 0, 6 or 9 (got %d)", mode);
    g_gemm_mode.store(mode, std::memory_order_relaxed);
    return 0;
}

extern "C" int mit_gemm_mode_get(void) { return gemm_mode_now(); }

extern "C" int64_t mit_gemm_split_min_tiles(int64_t n) {
    const int64_t prev = split_min_now();
    if (n >= 0) g_split_min.store(n, std::memory_order_relaxed);
    return prev;
}

extern "C" int mit_gemm_split_pack(const float *w_dev, int64_t w_zs, int nz, int Kw, int64_t ldw, uint16_t *out_dev, void *stream) {
    if (!w_dev || !out_dev) return mit_set_error("mit_gemm_split_pack: null pointer");
    if (nz <= 0 || Kw <= 0 || (Kw & 7) || ldw <= 0 || (ldw & 3) || ldw > 0x7fffffffLL)
        return mit_set_error("mit_gemm_split_pack: need nz > 0, Kw %% 8 == 0, ldw %% 4 == 0 (nz=%d Kw=%d ldw=%lld)", nz, Kw, (long long)ldw);
    if ((reinterpret_cast<uintptr_t>(out_dev) & 15)) return mit_set_error("mit_gemm_split_pack: output must be 16-byte aligned");
    const int64_t total = (int64_t)nz * (Kw >> 3) * ldw;
    hipLaunchKernelGGL(gemm_split_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       w_dev, w_zs, Kw >> 3, (int)ldw, out_dev, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mit_set_error("mit_gemm_split_pack: launch failed: %s", hipGetErrorString(e));
    return 0;
}
