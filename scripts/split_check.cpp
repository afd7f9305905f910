// split_check.cpp — stand-alone check of the split-bf16 tiles of mit_conv_gemm against its fp32 MFMA tiles and a float64 host
// reference, with timings.  No Python / torch: links libmit_hip.so and the HIP runtime only, so it starts in seconds on a GPU box.
//
//   hipcc -O2 -std=c++17 scripts/split_check.cpp -o scripts/split_check -Iinclude -Lmanga_image_translator_amd -lmit_hip \
//         -Wl,-rpath,'$ORIGIN/../manga_image_translator_amd'          (scripts/build_split_check.sh)
//   scripts/split_check [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <string>
#include <vector>
#include "mit_hip.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)

struct Case {
    const char *name;
    int NB, H, W, Cin, N, k, stride, pad_mode, Z, act;
    int ref_cfg;
    std::vector<int> cfgs;
};

static int find_cfg(const char *name) {
    for (int i = 0;; ++i) {
        const char *n = mit_conv_gemm_config_name(i);
        if (!n) return -1;
        if (!strcmp(n, name)) return i;
    }
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);  // a GPU fault kills the process: keep what was printed
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    const bool ablate = argc > 2 && !strcmp(argv[2], "--ablate");  // time the xs* ablation tiles of an MIT_CONV_EXPERIMENTS build (no checks)
    const int f_wide = find_cfg("fast128x128x16w4c"), f_narrow = find_cfg("fast128x64x16w5c");
    const int s6 = find_cfg("split128x128x16p6"), s9 = find_cfg("split128x128x16p9"), s3 = find_cfg("split128x128x16p3");
    const int n6 = find_cfg("split128x64x16p6"), n9 = find_cfg("split128x64x16p9"), s6k32 = find_cfg("split128x128x32p6");
    const int s6s = find_cfg("split128x128x16p6s"), s9s = find_cfg("split128x128x16p9s"), n6s = find_cfg("split128x64x16p6s"), s6k32s = find_cfg("split128x128x32p6s");
    const int s6m = find_cfg("split128x128x16p6m"), s9m = find_cfg("split128x128x16p9m"), n6m = find_cfg("split128x64x16p6m"), s6k32m = find_cfg("split128x128x32p6m");
    const int s6o = find_cfg("split128x128x16p6o"), n6o = find_cfg("split128x64x16p6o");
    if (s6o < 0 || n6o < 0) return 2;  // (tiles of MIT_CONV_EXPERIMENTS builds come back as -1 from a default build and are skipped below)
    const int s64 = find_cfg("split64x64x16p6o"), s32 = find_cfg("split128x32x16p6o"), f32t = find_cfg("fast128x32x16w4c");  // small / narrow tiles (-1: skipped)
    const int w6o = find_cfg("split64x256x16p6o");
    const int t192 = find_cfg("split128x192x16p6o"), t256 = find_cfg("split256x64x16p6o");
    const int t160 = find_cfg("split128x160x16p6o"), t96 = find_cfg("split128x96x16p6o");  // exact-N tiles for N = 160 k / N = 80  // wide-N tile (-1 in builds without it: skipped)
    const int pp = find_cfg("split256x128x16p6pp"), pq = find_cfg("split128x256x16p6pp");
    const int bd = find_cfg("split128x128x16p6u");  // round 6: operand loads through buffer instructions
    const int bv = find_cfg("split128x128x16p6v");  // ... + two K-tiles per loop trip with constant LDS buffer parity  // round 6: the eight-wave ping-pong tiles
    const int gen32 = find_cfg("128x32x16"), f64 = find_cfg("fast64x64x16w8c"), s64k = find_cfg("split64x64x32p6o");
    if (f_wide < 0 || f_narrow < 0 || s6 < 0 || s9 < 0 || s3 < 0 || n6 < 0 || n9 < 0 || s9m < 0) {
        fprintf(stderr, "tile names not found\n");
        return 2;
    }
    std::vector<Case> cases = {
        {"probe: A[m][k] = (k == m % 16), W[k][n] = 1000 k + n", 1, 1, 256, 16, 128, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s6, s9, s3, n6, s6s, n6s, s6m, n6m, s6o, n6o, s64}},
        {"1x1 320->1280 (ConvNeXt pw1), M=65536, gelu", 1, 256, 256, 320, 1280, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {s6, s9, s3, s6s, s6m, s9m, s6k32m, s6o, w6o, s6o, w6o, s6o}},
        {"3x3 reflect 128->128, 2x96x160, relu", 2, 96, 160, 128, 128, 3, 1, MIT_PAD_REFLECT, 1, MIT_ACT_RELU, f_wide, {s6, s9, s6s, s6m, s9m, s6k32m, s6o, s6m, s6o, s6m, s6o}},
        {"winograd-like Z=36, T=8192, 128->384", 1, 1, 8192, 128, 384, 1, 1, MIT_PAD_ZERO, 36, MIT_ACT_NONE, f_wide, {s6, s9, s6m, s9m, s6o, n6o, w6o, s6o, n6o, w6o}},
        {"1x1 192->384 (spectral conv2), M=131072, relu", 1, 256, 512, 192, 384, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, n6o, w6o, s6o, n6o, w6o}},
        {"1x1 160->640 (ConvNeXt stage 2 pw1), M=131072, gelu", 1, 256, 512, 160, 640, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {s6o, n6o, w6o, s6o, n6o, w6o}},
        {"1x1 80->320 (ConvNeXt stage 1 pw1), M=262144, gelu", 1, 512, 512, 80, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {s6o, n6o, w6o, s6o, n6o, w6o}},
        {"1x1 640->160 (ConvNeXt stage 2 pw2), M=131072", 1, 256, 512, 640, 160, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s6o, n6o, t160, s6o, n6o, t160}},
        {"1x1 320->80 (ConvNeXt stage 1 pw2), M=262144", 1, 512, 512, 320, 80, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s6o, n6o, t96, s6o, n6o, t96}},
        {"1x1 1280->320 (pw2), M=131072", 1, 256, 512, 1280, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s6o, n6o, t160, n6o, t160}},
        {"1x1 320->1280 (pw1), M=131072, gelu", 1, 256, 512, 320, 1280, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {s6o, t160, s6o, t160}},
        {"1x1 160->640 (pw1), M=262144, gelu", 1, 512, 512, 160, 640, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {s6o, t160, s6o, t160}},
        {"1x1 80->320 (pw1), M=524288, gelu", 1, 1024, 512, 80, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {s6o, n6o, t160, n6o, t160}},
        {"1x1 384->192 (spectral conv1), M=262144, relu", 1, 512, 512, 384, 192, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, n6o, t192, n6o, t192}},
        {"1x1 192->384 (spectral conv2), M=262144, relu", 1, 512, 512, 192, 384, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, t192, s6o, t192}},
        {"winograd-like Z=36, T=16384, 128->384", 1, 1, 16384, 128, 384, 1, 1, MIT_PAD_ZERO, 36, MIT_ACT_NONE, f_wide, {s6o, t192, s6o, t192}},
        {"1x1 2 taps-like: 768->384, M=131072", 1, 256, 512, 768, 384, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, t192, s6o, t192}},
        {"1x1 224->64 (stem-like K), M=1048576, relu", 1, 1024, 1024, 224, 64, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_narrow, {n6o, t256, n6o, t256}},
        {"1x1 256->64 (convT parity-like), M=1048576, relu", 1, 1024, 1024, 256, 64, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_narrow, {n6o, t256, n6o, t256}},
        {"1x1 512->64, M=1048576, relu", 1, 1024, 1024, 512, 64, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_narrow, {n6o, t256, n6o, t256}},
        {"3x3 zero 64->64, 2x512x512, leaky", 2, 512, 512, 64, 64, 3, 1, MIT_PAD_ZERO, 1, MIT_ACT_LEAKY, f_narrow, {n6o, t256, n6o, t256}},
        {"1x1 80->320 (pw1) again, M=524288, gelu", 1, 1024, 512, 80, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {n6o, t256, n6o, t256}},
        {"decoder-like M=10240: 320->960 exact-N", 1, 1, 10240, 320, 960, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {n6o, t160, n6o, t160}},
        {"encoder-like M=60208: 320->960 exact-N", 1, 1, 60208, 320, 960, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {n6o, t160, n6o, t160}},
        {"3x3 s2 zero 128->256 (LaMa down), 2x128x96", 2, 256, 192, 128, 256, 3, 2, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, w6o, s6o, w6o}},
        {"1x1 384x2taps-like: 768->384, M=65536", 1, 256, 256, 768, 384, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, w6o, s6o, w6o}},
        {"3x3 s2 zero 64->64, 4x128x128", 4, 128, 128, 64, 64, 3, 2, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_narrow, {n6, n9, n6s, n6m, n6o, n6m, n6o}},
        {"1x1 1280->320 (pw2), M=65536", 1, 256, 256, 1280, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s6, s9, n6, s6m, n6m, s6k32m, s6o, n6o, s6m, s6o}},
        {"3x3 zero 128->32 (ESRGAN dense conv3), 2x256x256, leaky", 2, 256, 256, 128, 32, 3, 1, MIT_PAD_ZERO, 1, MIT_ACT_LEAKY, gen32, {f32t, s32, n6o}},
        {"decoder-like M=160: 320->960", 1, 1, 160, 320, 960, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {f64, s6o, n6o, s64, s64k, s64, s64k}},
        {"decoder-like M=160: 320->320", 1, 1, 160, 320, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {f64, s64, s64k, s64, s64k}},
        {"decoder-like M=160: 320->2048 (FFN in), relu", 1, 1, 160, 320, 2048, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {f64, s64, s64k, s64, s64k}},
        {"decoder-like M=160: 2048->320 (FFN out)", 1, 1, 160, 2048, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {f64, s64, s64k, s64, s64k}},
        {"decoder-like M=160: 320->6004 (logits)", 1, 1, 160, 320, 6004, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {f64, s6o, s64, s64k}},
        {"decoder-like M=10240: 320->960", 1, 1, 10240, 320, 960, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s6o, n6o, s64, s64k, s64, s64k}},
        {"decoder-like M=10240: 2048->320", 1, 1, 10240, 2048, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s64, s64k, s64, s64k}},
        {"detector-like 3x3 256->256, 1x32x32", 1, 32, 32, 256, 256, 3, 1, MIT_PAD_ZERO, 1, MIT_ACT_LEAKY, f_wide, {s64, s64k, s64, s64k}},
        {"ragged: M=1000, 48->200 3x3 zero", 1, 25, 40, 48, 200, 3, 1, MIT_PAD_ZERO, 1, MIT_ACT_LEAKY, f_wide, {s6, n6, n9, s6m, n6m, s6o, n6o, pp, pq}},
        // round 6: the five shapes of the ping-pong experiment, each against the shipped 128 x 128 tile, twice
        {"pp: long K 1x1 2304->512, M=46592, relu", 1, 182, 256, 2304, 512, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, pp, pq, s6o, pp, pq}},
        {"pp: 3x3 s2 zero 256->512 (LaMa down3), 4x91x128", 4, 182, 256, 256, 512, 3, 2, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, pp, pq, s6o, pp, pq}},
        {"pp: winograd Z=36, T=11776, 512->128", 1, 1, 11776, 512, 128, 1, 1, MIT_PAD_ZERO, 36, MIT_ACT_NONE, f_wide, {s6o, pp, s6o, pp}},
        {"pp: winograd Z=36, T=11776, 128->384", 1, 1, 11776, 128, 384, 1, 1, MIT_PAD_ZERO, 36, MIT_ACT_NONE, f_wide, {s6o, pp, pq, s6o, pp, pq}},
        {"pp: spectral 1x1 192->384, M=186368, relu", 1, 364, 512, 192, 384, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_RELU, f_wide, {s6o, t192, pp, pq, s6o, t192, pp, pq}},
        {"pp: pw1 1x1 320->1280, M=131072, gelu", 1, 256, 512, 320, 1280, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_GELU, f_wide, {s6o, pp, pq, s6o, pp, pq}},
        {"pp: pw2 1x1 1280->320, M=131072", 1, 256, 512, 1280, 320, 1, 1, MIT_PAD_ZERO, 1, MIT_ACT_NONE, f_wide, {s6o, t160, pp, pq, s6o, t160, pp, pq}},
        {"pp: 3x3 reflect 128->128, 2x96x160, relu", 2, 96, 160, 128, 128, 3, 1, MIT_PAD_REFLECT, 1, MIT_ACT_RELU, f_wide, {s6o, pp, s6o, pp}},
    };
    for (auto &c : cases)   // the direct-W tile wherever the shipped wide tile is measured
        if (!strncmp(c.name, "pp:", 3) || !strncmp(c.name, "ragged", 6) || !strncmp(c.name, "probe", 5)) {
            c.cfgs.push_back(bd);
            c.cfgs.push_back(bv);
            c.cfgs.push_back(s6o);
            c.cfgs.push_back(bd);
            c.cfgs.push_back(bv);
        }
    if (const char *only = getenv("SC_CASE")) {  // substring filter on the case name
        std::vector<Case> keep;
        for (auto &c : cases)
            if (strstr(c.name, only)) keep.push_back(c);
        cases.swap(keep);
    }
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    int bad = 0;
    for (auto &cs : cases) {
        const int pad = cs.k / 2;
        const int Ho = (cs.H + 2 * pad - cs.k) / cs.stride + 1, Wo = (cs.W + 2 * pad - cs.k) / cs.stride + 1;
        const int ntaps = cs.k * cs.k, K = ntaps * cs.Cin, Kp = (K + 15) / 16 * 16, Np = (cs.N + 3) / 4 * 4;
        const int64_t a_elems = (int64_t)cs.Z * cs.NB * cs.H * cs.W * cs.Cin, w_elems = (int64_t)cs.Z * Kp * Np;
        const int64_t M = (int64_t)cs.NB * Ho * Wo, c_elems = (int64_t)cs.Z * M * cs.N;
        std::vector<float> ha(a_elems), hw(w_elems, 0.f), hbias(Np), hscale(Np);
        for (auto &v : ha) v = nd(rng) * (1.f + 3.f * (float)((&v - ha.data()) % 7 == 0));
        for (int z = 0; z < cs.Z; ++z)
            for (int k = 0; k < K; ++k)
                for (int n = 0; n < cs.N; ++n) hw[((int64_t)z * Kp + k) * Np + n] = nd(rng) * 0.05f;
        for (int n = 0; n < Np; ++n) hbias[n] = nd(rng) * 0.1f, hscale[n] = 1.f + 0.1f * nd(rng);
        const bool probe = !strncmp(cs.name, "probe", 5);
        if (probe) {  // output[m][n] must be exactly W[m % 16][n]: any row / column / k-group mix-up shows as a readable number
            for (int64_t i = 0; i < a_elems; ++i) ha[i] = (i % cs.Cin) == (i / cs.Cin) % 16 ? 1.f : 0.f;
            for (int k = 0; k < K; ++k)
                for (int n = 0; n < cs.N; ++n) hw[(int64_t)k * Np + n] = 1000.f * k + n;
            for (int n = 0; n < Np; ++n) hbias[n] = 0.f, hscale[n] = 1.f;
        }
        float *da, *dw, *dc, *dref, *dbias, *dscale;
        uint16_t *dsplit;
        CK(hipMalloc(&da, a_elems * 4));
        CK(hipMalloc(&dw, w_elems * 4));
        CK(hipMalloc(&dc, c_elems * 4));
        CK(hipMalloc(&dref, c_elems * 4));
        CK(hipMalloc(&dbias, Np * 4));
        CK(hipMalloc(&dscale, Np * 4));
        CK(hipMalloc(&dsplit, w_elems * 3 * 2));
        CK(hipMemcpy(da, ha.data(), a_elems * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), w_elems * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, hbias.data(), Np * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dscale, hscale.data(), Np * 4, hipMemcpyHostToDevice));
        if (mit_gemm_split_pack(dw, (int64_t)Kp * Np, cs.Z, Kp, Np, dsplit, nullptr)) {
            fprintf(stderr, "pack: %s\n", mit_last_error());
            return 2;
        }
        CK(hipDeviceSynchronize());
        {   // the planes must add up to the fp32 weights exactly
            std::vector<uint16_t> hs(w_elems * 3);
            CK(hipMemcpy(hs.data(), dsplit, w_elems * 3 * 2, hipMemcpyDeviceToHost));
            int64_t wrong = 0;
            const int K8 = Kp / 8;
            for (int z = 0; z < cs.Z && wrong == 0; ++z)
                for (int k = 0; k < Kp; ++k)
                    for (int n = 0; n < Np; ++n) {
                        float sum = 0.f;
                        for (int pl = 2; pl >= 0; --pl) {
                            const uint16_t b = hs[((((int64_t)z * 3 + pl) * K8 + k / 8) * Np + n) * 8 + (k & 7)];
                            uint32_t u = (uint32_t)b << 16;
                            float f;
                            memcpy(&f, &u, 4);
                            sum += f;
                        }
                        if (sum != hw[((int64_t)z * Kp + k) * Np + n]) ++wrong;
                    }
            printf("%-48s pack: %s\n", cs.name, wrong ? "PLANES DO NOT SUM TO W" : "hi + mid + lo == w exactly");
            bad += wrong != 0;
        }
        MitConvGemm d;
        memset(&d, 0, sizeof(d));
        d.a = da;
        d.a_zs0 = (int64_t)cs.NB * cs.H * cs.W * cs.Cin;
        d.a_bs = (int64_t)cs.H * cs.W * cs.Cin, d.a_ys = (int64_t)cs.W * cs.Cin, d.a_xs = cs.Cin;
        d.NB = cs.NB, d.Hi = cs.H, d.Wi = cs.W, d.Cin = cs.Cin, d.Ho = Ho, d.Wo = Wo, d.sy = d.sx = cs.stride;
        d.ntaps = ntaps, d.pad_mode = cs.pad_mode;
        for (int t = 0; t < ntaps; ++t) d.tap_dy[t] = t / cs.k - pad, d.tap_dx[t] = t % cs.k - pad, d.tap_off[t] = 0;
        d.w = dw, d.w_zs0 = (int64_t)Kp * Np, d.ldw = Np, d.Kw = Kp, d.Nw = Np;
        d.N = cs.N, d.Z = cs.Z, d.zdiv = 1 << 30;
        d.c.base = dref, d.c.zs0 = M * cs.N, d.c.bs = (int64_t)Ho * Wo * cs.N, d.c.ys = (int64_t)Wo * cs.N, d.c.xs = cs.N;
        d.scale = dscale, d.bias = dbias, d.act = cs.act, d.act_alpha = 0.1f;
        d.w_split = dsplit, d.ws_zs0 = (int64_t)3 * Kp * Np;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        auto run = [&](int cfg, float *out, float *ms) -> int {
            d.c.base = out;
            if (mit_conv_gemm_cfg(&d, cfg, nullptr)) {
                fprintf(stderr, "  cfg %s: %s\n", mit_conv_gemm_config_name(cfg), mit_last_error());
                return 1;
            }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, nullptr));
            for (int r = 0; r < reps; ++r) mit_conv_gemm_cfg(&d, cfg, nullptr);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(ms, e0, e1));
            *ms /= reps;
            return 0;
        };
        float ms_ref = 0.f;
        if (run(cs.ref_cfg, dref, &ms_ref)) return 2;
        std::vector<float> href(c_elems), hc(c_elems);
        CK(hipMemcpy(href.data(), dref, c_elems * 4, hipMemcpyDeviceToHost));
        // float64 reference of the pre-activation sums on a sample of outputs
        const int NS = 4000;
        std::vector<int64_t> sm(NS);
        std::vector<double> exact(NS);
        double ymax = 0;
        for (auto v : href) ymax = fmax(ymax, fabs((double)v));
        std::uniform_int_distribution<int64_t> pick(0, c_elems - 1);
        auto act64 = [&](double v) {
            if (cs.act == MIT_ACT_RELU) return v > 0 ? v : 0.0;
            if (cs.act == MIT_ACT_LEAKY) return v > 0 ? v : 0.1 * v;
            if (cs.act == MIT_ACT_GELU) return 0.5 * v * (1.0 + erf(v * 0.70710678118654752440));
            return v;
        };
        for (int s = 0; s < NS; ++s) {
            const int64_t idx = pick(rng);
            sm[s] = idx;
            const int n = (int)(idx % cs.N);
            const int64_t m = (idx / cs.N) % M;
            const int z = (int)(idx / cs.N / M);
            const int nb = (int)(m / ((int64_t)Ho * Wo)), oy = (int)(m / Wo % Ho), ox = (int)(m % Wo);
            double acc = 0;
            for (int t = 0; t < ntaps; ++t) {
                int iy = oy * cs.stride + t / cs.k - pad, ix = ox * cs.stride + t % cs.k - pad;
                if (cs.pad_mode == MIT_PAD_REFLECT) {
                    iy = iy < 0 ? -iy : (iy >= cs.H ? 2 * cs.H - 2 - iy : iy);
                    ix = ix < 0 ? -ix : (ix >= cs.W ? 2 * cs.W - 2 - ix : ix);
                } else if (iy < 0 || iy >= cs.H || ix < 0 || ix >= cs.W) {
                    continue;
                }
                const float *ap = ha.data() + (((int64_t)z * cs.NB + nb) * cs.H + iy) * cs.W * cs.Cin + (int64_t)ix * cs.Cin;
                const float *wp = hw.data() + ((int64_t)z * Kp + (int64_t)t * cs.Cin) * Np + n;
                for (int c = 0; c < cs.Cin; ++c) acc += (double)ap[c] * (double)wp[(int64_t)c * Np];
            }
            exact[s] = act64(acc * (double)hscale[n] + (double)hbias[n]);
        }
        auto err64 = [&](const std::vector<float> &y) {
            double e = 0;
            for (int s = 0; s < NS; ++s) e = fmax(e, fabs((double)y[sm[s]] - exact[s]));
            return e / ymax;
        };
        const double flops = 2.0 * M * cs.N * K * cs.Z;
        printf("  %-22s %9.3f ms %8.1f TFLOP/s   err vs f64 %.2e (relative to max|y| = %.3g)\n", mit_conv_gemm_config_name(cs.ref_cfg), ms_ref,
               flops / ms_ref * 1e-9, err64(href), ymax);
        const double e_ref = err64(href);
        std::vector<float> hfirst6;  // the first 6-pair split tile's output: every other one must reproduce it bit for bit
        for (int cfg : cs.cfgs) {
            if (cfg < 0) continue;
            float ms = 0.f;
            printf("  %-22s ...\n", mit_conv_gemm_config_name(cfg));
            CK(hipMemset(dc, 0xff, c_elems * 4));
            if (run(cfg, dc, &ms)) {
                ++bad;
                continue;
            }
            CK(hipMemcpy(hc.data(), dc, c_elems * 4, hipMemcpyDeviceToHost));
            double dmax = 0;
            int64_t nan = 0;
            for (int64_t i = 0; i < c_elems; ++i) {
                if (!(hc[i] == hc[i])) ++nan;
                dmax = fmax(dmax, fabs((double)hc[i] - (double)href[i]));
            }
            const std::string nm = mit_conv_gemm_config_name(cfg);
            if (probe || dmax / ymax > 1e-2) {
                int shown = 0;
                for (int64_t i = 0; i < c_elems && shown < 12; ++i)
                    if (hc[i] != href[i] && (probe || fabs((double)hc[i] - href[i]) > 1e-2 * ymax)) {
                        printf("    [%s] z %lld m %lld n %lld: got %.9g want %.9g\n", nm.c_str(), (long long)(i / cs.N / M), (long long)(i / cs.N % M),
                               (long long)(i % cs.N), hc[i], href[i]);
                        ++shown;
                    }
            }
            const bool p3 = nm.find("p3") != std::string::npos;
            const double e = err64(hc), tol = p3 ? 2e-2 : 4.0 * e_ref + 2e-6;
            const bool fp32_tile = nm.rfind("fast", 0) == 0;  // another fp32 MFMA tile: same k-sequential chain, so the very same bits
            const bool ok = nan == 0 && e <= tol && dmax / ymax <= (p3 ? 2e-2 : 2e-5) && (!fp32_tile || dmax == 0.0);
            const char *same = "";
            if (nm.rfind("split", 0) == 0 && nm.find("p6") != std::string::npos) {
                if (hfirst6.empty()) hfirst6 = hc;
                else if (memcmp(hfirst6.data(), hc.data(), c_elems * 4)) same = "  BITS DIFFER from the first p6 tile", ++bad;
                else same = "  == first p6 tile";
            }
            printf("  %-22s %9.3f ms %8.1f TFLOP/s   err vs f64 %.2e   vs fp32 tile %.2e   nan %lld   %s  (x%.2f)%s\n", nm.c_str(), ms, flops / ms * 1e-9, e,
                   dmax / ymax, (long long)nan, ok ? "ok" : "FAIL", ms_ref / ms, same);
            bad += !ok;
        }
        if (ablate && (strstr(cs.name, "pw1") || strstr(cs.name, "3x3 reflect") || strstr(cs.name, "pw2"))) {
            for (const char *nm : {"split128x128x16p6", "split128x128x16p6s", "xsAsmSub", "xsAsmSubNP", "xsNoLoad", "xsNoSplit", "xsNoWrite", "xsNoFrag", "xsNoBar",
                                   "xsMfmaOnly", "xsNoLoadNP", "xsNoWriteNP", "xsNoFragNP", "xsNoLoadWriteNP", "xsMfmaOnlyNP"}) {
                const int cfg = find_cfg(nm);
                float ms = 0.f;
                if (cfg < 0) continue;
                printf("  ablation %-20s ...\n", nm);
                if (run(cfg, dc, &ms)) continue;
                printf("  ablation %-20s %9.3f ms %8.1f TFLOP/s-equivalent\n", nm, ms, flops / ms * 1e-9);
            }
        }
        CK(hipFree(da)); CK(hipFree(dw)); CK(hipFree(dc)); CK(hipFree(dref)); CK(hipFree(dbias)); CK(hipFree(dscale)); CK(hipFree(dsplit));
    }
    printf(bad ? "SPLIT CHECK FAILED (%d)\n" : "SPLIT CHECK OK\n", bad);
    return bad ? 1 : 0;
}
