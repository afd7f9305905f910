// pgemm_check.cpp — stand-alone check of mit_pgemm (planar-operand GEMM, csrc/pgemm.hip) against the split-bf16 tiles of
// mit_conv_gemm, with timings.  No Python / torch: links libmit_hip.so and the HIP runtime only.
//
//   scripts/build_pgemm_check.sh && scripts/pgemm_check [reps] [--quick]
//
// Per case: A fp32 [Z][M][K], W fp32 [Z][K][N] (random, heavy-tailed), scale / bias / activation, optional pre / post.
//   reference  : mit_conv_gemm on the split tile (split128x128x16p6o or split128x64x16p6o), fp32 output
//   planes     : mit_split_planes(A) must join back to A exactly (mit_join_planes)
//   fp32 out   : every "pg*" tile with fp32 output must reproduce the reference BIT FOR BIT
//   planar out : every "pg*P" / "pg*Q" tile must equal mit_split_planes(reference) BIT FOR BIT (the Q tile does the lane exchange of the
//                planar epilogue with __shfl_xor: if P fails and Q passes, v_permlane32_swap pairs the halves the other way round)
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

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)

struct Case {
    const char *name;
    int M, K, N, Z, act;
    bool pre, post, post_first;
    int reps_scale;  // timing repetitions divisor for the big cases
};

static int find_cfg(const char *name) {
    for (int i = 0;; ++i) {
        const char *n = mit_conv_gemm_config_name(i);
        if (!n) return -1;
        if (!strcmp(n, name)) return i;
    }
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    int reps = 10;
    bool quick = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--quick")) quick = true;
        else reps = atoi(argv[i]);
    }
    const int s_wide = find_cfg("split128x128x16p6o"), s_narrow = find_cfg("split128x64x16p6o"), s_wide9 = find_cfg("split128x128x16p9m");
    if (s_wide < 0 || s_narrow < 0) {
        fprintf(stderr, "split tile names not found\n");
        return 2;
    }
    std::vector<Case> cases = {
        {"tiny ragged: M=1000 K=48 N=200, relu, post", 1000, 48, 200, 1, MIT_ACT_RELU, false, true, false, 1},
        {"tiny: M=130 K=16 N=8 (one K-tile)", 130, 16, 8, 1, MIT_ACT_NONE, false, false, false, 1},
        {"ConvNeXt pw1 320->1280 gelu, M=65536", 65536, 320, 1280, 1, MIT_ACT_GELU, false, false, false, 1},
        {"ConvNeXt pw2 1280->320 + residual, M=65536", 65536, 1280, 320, 1, MIT_ACT_NONE, false, true, false, 1},
        {"ConvNeXt pw1 160->640 gelu, M=131072", 131072, 160, 640, 1, MIT_ACT_GELU, false, false, false, 1},
        {"ConvNeXt pw2 640->160 + residual, M=131072", 131072, 640, 160, 1, MIT_ACT_NONE, false, true, false, 1},
        {"ConvNeXt pw1 80->320 gelu, M=262144", 262144, 80, 320, 1, MIT_ACT_GELU, false, false, false, 1},
        {"ConvNeXt pw2 320->80 + residual, M=262144", 262144, 320, 80, 1, MIT_ACT_NONE, false, true, false, 1},
        {"winograd Z=36, T=11776 (4 pages), 512->128", 11776, 512, 128, 36, MIT_ACT_NONE, false, false, false, 1},
        {"winograd Z=36, T=11776 (4 pages), 128->384", 11776, 128, 384, 36, MIT_ACT_NONE, false, false, false, 1},
        {"spectral st_in 384->192 relu, M=186368", 186368, 384, 192, 1, MIT_ACT_RELU, false, false, false, 1},
        {"spectral st_out 192->384 pre + relu + post, M=186368", 186368, 192, 384, 1, MIT_ACT_RELU, true, true, false, 1},
        {"post-first (ResNet block): 256->256, M=40000", 40000, 256, 256, 1, MIT_ACT_RELU | MIT_ACT_POST_FIRST, false, true, true, 1},
        {"long K: 2304->512, M=46592", 46592, 2304, 512, 1, MIT_ACT_RELU, false, false, false, 1},
        {"encoder FFN 320->2048 relu, M=10240", 10240, 320, 2048, 1, MIT_ACT_RELU, false, false, false, 1},
        {"logits 320->6004, M=10240 (N % 8 != 0)", 10240, 320, 6004, 1, MIT_ACT_NONE, false, false, false, 1},
        // one page's decoder (32 lines x 5 beams): launches bound by the latency of their K loop, not by throughput
        {"decoder qkv 320->960, M=160", 160, 320, 960, 1, MIT_ACT_NONE, false, false, false, 1},
        {"decoder out 320->320 + residual, M=160", 160, 320, 320, 1, MIT_ACT_NONE, false, true, false, 1},
        {"decoder ff1 320->2048 relu, M=160", 160, 320, 2048, 1, MIT_ACT_RELU, false, false, false, 1},
        {"decoder ff2 2048->320 + residual, M=160", 160, 2048, 320, 1, MIT_ACT_NONE, false, true, false, 1},
        {"decoder logits 320->6004, M=160", 160, 320, 6004, 1, MIT_ACT_NONE, false, false, false, 1},
        {"decoder qkv 320->960, M=2560 (16 pages)", 2560, 320, 960, 1, MIT_ACT_NONE, false, false, false, 1},
        {"decoder ff2 2048->320 + residual, M=2560 (16 pages)", 2560, 2048, 320, 1, MIT_ACT_NONE, false, true, false, 1},
        {"decoder qkv 320->960, M=640 (4 pages)", 640, 320, 960, 1, MIT_ACT_NONE, false, false, false, 1},
        {"decoder ff2 2048->320 + residual, M=640 (4 pages)", 640, 2048, 320, 1, MIT_ACT_NONE, false, true, false, 1},
    };
    if (quick) cases.resize(4);
    const char *only_case = getenv("PG_CASE"), *only_tile = getenv("PG_TILE");  // substrings: run only matching cases / pgemm tiles
    // a cheap generator (the big cases fill a few hundred million floats on the host): sum of four uniforms, zero mean, unit-ish variance
    struct Fast {
        uint64_t s = 0x9E3779B97F4A7C15ull;
        uint32_t next() {
            s ^= s << 13, s ^= s >> 7, s ^= s << 17;
            return (uint32_t)(s >> 32);
        }
        float operator()(Fast &) {
            const uint32_t a = next(), b = next();
            return ((float)(a & 0xffff) + (float)(a >> 16) + (float)(b & 0xffff) + (float)(b >> 16) - 131070.f) * (1.f / 37837.f);
        }
    } nd;
    Fast &rng = nd;
    int bad = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (auto &cs : cases) {
        if (only_case && *only_case && !strstr(cs.name, only_case)) continue;
        const int M = cs.M, K = cs.K, N = cs.N, Z = cs.Z;
        const int Kp = (K + 15) / 16 * 16, Np = (N + 3) / 4 * 4;
        const int64_t a_el = (int64_t)Z * M * K, w_el = (int64_t)Z * Kp * Np, c_el = (int64_t)Z * M * N;
        const int64_t lda = M, ld_cp = M;
        std::vector<float> ha(a_el), hw(w_el, 0.f), hbias(Np), hscale(Np), hres(c_el);
        for (int64_t i = 0; i < a_el; ++i) ha[i] = nd(rng) * (1.f + 3.f * (float)(i % 7 == 0));
        for (int z = 0; z < Z; ++z)
            for (int k = 0; k < K; ++k)
                for (int n = 0; n < N; ++n) hw[((int64_t)z * Kp + k) * Np + n] = nd(rng) * 0.05f;
        for (int n = 0; n < Np; ++n) hbias[n] = nd(rng) * 0.1f, hscale[n] = 1.f + 0.1f * nd(rng);
        for (auto &v : hres) v = nd(rng);
        float *da, *dw, *dref, *dc, *dbias, *dscale, *dres, *dres2, *djoin;
        uint16_t *dwsplit, *dapl, *dcpl, *drefpl;
        const bool planar_ok = N % 8 == 0 && !cs.pre && !cs.post;
        CK(hipMalloc(&da, a_el * 4));
        CK(hipMalloc(&djoin, a_el * 4));
        CK(hipMalloc(&dw, w_el * 4));
        CK(hipMalloc(&dref, c_el * 4));
        CK(hipMalloc(&dc, c_el * 4));
        CK(hipMalloc(&dres, c_el * 4));
        CK(hipMalloc(&dres2, c_el * 4));
        CK(hipMalloc(&dbias, Np * 4));
        CK(hipMalloc(&dscale, Np * 4));
        CK(hipMalloc(&dwsplit, w_el * 3 * 2));
        CK(hipMalloc(&dapl, (int64_t)Z * 3 * K * lda * 2));
        CK(hipMalloc(&dcpl, (int64_t)Z * 3 * N * ld_cp * 2 + 16));
        CK(hipMalloc(&drefpl, (int64_t)Z * 3 * N * ld_cp * 2 + 16));
        CK(hipMemcpy(da, ha.data(), a_el * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), w_el * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbias, hbias.data(), Np * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dscale, hscale.data(), Np * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dres, hres.data(), c_el * 4, hipMemcpyHostToDevice));
        for (auto &v : hres) v = nd(rng);
        CK(hipMemcpy(dres2, hres.data(), c_el * 4, hipMemcpyHostToDevice));
        if (mit_gemm_split_pack(dw, (int64_t)Kp * Np, Z, Kp, Np, dwsplit, nullptr)) {
            fprintf(stderr, "pack: %s\n", mit_last_error());
            return 2;
        }
        printf("%s\n", cs.name);
        // ---- planes of A, and back
        for (int z = 0; z < Z; ++z) {
            if (mit_split_planes(da + (int64_t)z * M * K, K, M, K, dapl + (int64_t)z * 3 * K * lda, lda, nullptr) ||
                mit_join_planes(dapl + (int64_t)z * 3 * K * lda, lda, M, K, djoin + (int64_t)z * M * K, K, nullptr)) {
                fprintf(stderr, "planes: %s\n", mit_last_error());
                return 2;
            }
        }
        CK(hipDeviceSynchronize());
        {
            std::vector<float> hj(a_el);
            CK(hipMemcpy(hj.data(), djoin, a_el * 4, hipMemcpyDeviceToHost));
            const bool same = !memcmp(hj.data(), ha.data(), a_el * 4);
            printf("  planes: split -> join %s\n", same ? "== A exactly" : "DIFFERS FROM A");
            bad += !same;
        }
        // ---- reference: the split tile of mit_conv_gemm
        MitConvGemm d;
        memset(&d, 0, sizeof(d));
        d.a = da, d.a_zs0 = (int64_t)M * K, d.a_xs = K;
        d.NB = 1, d.Hi = 1, d.Wi = M, d.Cin = K, d.Ho = 1, d.Wo = M, d.sy = d.sx = 1, d.ntaps = 1, d.pad_mode = MIT_PAD_ZERO;
        d.w = dw, d.w_zs0 = (int64_t)Kp * Np, d.ldw = Np, d.Kw = Kp, d.Nw = Np;
        d.N = N, d.Z = Z, d.zdiv = 1 << 30;
        d.c.base = dref, d.c.zs0 = (int64_t)M * N, d.c.xs = N;
        if (cs.pre) d.pre.base = dres, d.pre.zs0 = (int64_t)M * N, d.pre.xs = N;
        if (cs.post) d.post.base = dres2, d.post.zs0 = (int64_t)M * N, d.post.xs = N;
        d.scale = dscale, d.bias = dbias, d.act = cs.act, d.act_alpha = 0.1f;
        d.w_split = dwsplit, d.ws_zs0 = (int64_t)3 * Kp * Np;
        const int r128 = N % 128;
        const bool narrow = N <= 64 || (r128 != 0 && r128 <= 64);
        const int ref_cfg = narrow ? s_narrow : s_wide;
        auto time_it = [&](auto fn, float *ms) -> int {
            if (fn()) return 1;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, nullptr));
            for (int r = 0; r < reps; ++r) fn();
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(ms, e0, e1));
            *ms /= reps;
            return 0;
        };
        float ms_ref = 0.f;
        if (time_it([&] { return mit_conv_gemm_cfg(&d, ref_cfg, nullptr); }, &ms_ref)) {
            fprintf(stderr, "  reference: %s\n", mit_last_error());
            return 2;
        }
        const double flops = 2.0 * M * (double)N * K * Z;
        printf("  %-20s %9.3f ms %8.1f TFLOP/s (fp32-equivalent)\n", mit_conv_gemm_config_name(ref_cfg), ms_ref, flops / ms_ref * 1e-9);
        {   // the other split tile of mit_conv_gemm on the same problem (which of the two should the automatic choice take?)
            const int other = narrow ? s_wide : s_narrow;
            float ms_o = 0.f;
            d.c.base = dc;
            if (!time_it([&] { return mit_conv_gemm_cfg(&d, other, nullptr); }, &ms_o))
                printf("  %-20s %9.3f ms %8.1f TFLOP/s   x%.2f   (the other split tile)\n", mit_conv_gemm_config_name(other), ms_o, flops / ms_o * 1e-9, ms_ref / ms_o);
            if (M <= 4096)   // under-filled launches: the small split tiles the automatic choice gives them
                for (const char *nm : {"split64x64x16p6o", "split64x64x32p6o"}) {
                    const int c = find_cfg(nm);
                    if (c >= 0 && !time_it([&] { return mit_conv_gemm_cfg(&d, c, nullptr); }, &ms_o))
                        printf("  %-20s %9.3f ms %8.1f TFLOP/s   x%.2f\n", nm, ms_o, flops / ms_o * 1e-9, ms_ref / ms_o);
                }
            d.c.base = dref;
        }
        std::vector<float> href(c_el), hc(c_el);
        CK(hipMemcpy(href.data(), dref, c_el * 4, hipMemcpyDeviceToHost));
        std::vector<uint16_t> hrefpl, hcpl;
        if (planar_ok) {
            for (int z = 0; z < Z; ++z)
                if (mit_split_planes(dref + (int64_t)z * M * N, N, M, N, drefpl + (int64_t)z * 3 * N * ld_cp, ld_cp, nullptr)) return 2;
            CK(hipDeviceSynchronize());
            hrefpl.resize((int64_t)Z * 3 * N * ld_cp);
            hcpl.resize(hrefpl.size());
            CK(hipMemcpy(hrefpl.data(), drefpl, hrefpl.size() * 2, hipMemcpyDeviceToHost));
        }
        // ---- mit_pgemm tiles
        MitPGemm g;
        memset(&g, 0, sizeof(g));
        g.a_planes = dapl, g.a_zs = (int64_t)3 * K * lda, g.lda = lda;
        g.w_planes = dwsplit, g.w_zs = (int64_t)3 * Kp * Np, g.ldw = Np;
        g.M = M, g.N = N, g.K = K, g.Z = Z;
        g.scale = dscale, g.bias = dbias, g.act = cs.act, g.act_alpha = 0.1f;
        for (int tile = 0;; ++tile) {
            const char *tn = mit_pgemm_tile_name(tile);
            if (!tn) break;
            const std::string nm = tn;
            if (only_tile && *only_tile && nm.find(only_tile) == std::string::npos) continue;
            const bool is_planar = nm.back() == 'P' || nm.back() == 'Q';
            const bool is9 = nm.find("p9") != std::string::npos;
            if (is9 && s_wide9 < 0) continue;
            if (is_planar && !planar_ok) continue;
            if (is9 && (Z > 1 || M > 70000)) continue;  // the 9-pair tiles: a couple of cases are enough
            g.tile = tile, g.nprod = 0;
            g.c = nullptr, g.c_planes = nullptr, g.pre = g.post = nullptr;
            if (is_planar) {
                g.c_planes = dcpl, g.ld_cp = ld_cp, g.cp_zs = (int64_t)3 * N * ld_cp;
                CK(hipMemset(dcpl, 0xff, (int64_t)Z * 3 * N * ld_cp * 2));
            } else {
                g.c = dc, g.ldc = N, g.c_zs = (int64_t)M * N;
                if (cs.pre) g.pre = dres, g.ld_pre = N, g.pre_zs = (int64_t)M * N;
                if (cs.post) g.post = dres2, g.ld_post = N, g.post_zs = (int64_t)M * N;
                CK(hipMemset(dc, 0xff, c_el * 4));
            }
            float ms = 0.f;
            if (time_it([&] { return mit_pgemm(&g, nullptr); }, &ms)) {
                printf("  %-20s REFUSED: %s\n", tn, mit_last_error());
                ++bad;
                continue;
            }
            int64_t diff = 0, first = -1;
            if (nm[0] == 'x') {  // timing ablation (MIT_CONV_EXPERIMENTS builds): wrong results by construction
                printf("  %-20s %9.3f ms %8.1f TFLOP/s-equivalent   x%.2f   (ablation: not compared)\n", tn, ms, flops / ms * 1e-9, ms_ref / ms);
                continue;
            }
            if (is9) {  // nine pairs: compare with the nine-pair split tile
                static std::vector<float> href9;
                href9.resize(c_el);
                // (dref is overwritten by the nine-pair tile and restored afterwards)
                if (mit_conv_gemm_cfg(&d, s_wide9, nullptr)) return 2;
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(href9.data(), dref, c_el * 4, hipMemcpyDeviceToHost));
                if (mit_conv_gemm_cfg(&d, ref_cfg, nullptr)) return 2;  // restore the 6-pair reference
                CK(hipDeviceSynchronize());
                if (is_planar) {
                    printf("  %-20s %9.3f ms %8.1f TFLOP/s   (9-pair planar output: not compared)\n", tn, ms, flops / ms * 1e-9);
                    continue;
                }
                CK(hipMemcpy(hc.data(), dc, c_el * 4, hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < c_el; ++i)
                    if (memcmp(&hc[i], &href9[i], 4)) {
                        if (first < 0) first = i;
                        ++diff;
                    }
            } else if (is_planar) {
                CK(hipMemcpy(hcpl.data(), dcpl, hcpl.size() * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < hcpl.size(); ++i)
                    if (hcpl[i] != hrefpl[i]) {
                        if (first < 0) first = (int64_t)i;
                        ++diff;
                    }
            } else {
                CK(hipMemcpy(hc.data(), dc, c_el * 4, hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < c_el; ++i)
                    if (memcmp(&hc[i], &href[i], 4)) {
                        if (first < 0) first = i;
                        ++diff;
                    }
            }
            printf("  %-20s %9.3f ms %8.1f TFLOP/s   x%.2f   %s", tn, ms, flops / ms * 1e-9, ms_ref / ms, diff ? "BITS DIFFER" : "== reference bit for bit");
            if (diff) {
                if (is_planar) {
                    const int64_t per = (int64_t)3 * N * ld_cp;
                    const int64_t z = first / per, rem = first % per, pl = rem / ((int64_t)N * ld_cp), c8 = rem % ((int64_t)N * ld_cp) / (ld_cp * 8),
                                  m = rem % (ld_cp * 8) / 8, j = rem % 8;
                    printf("  (%lld elements; first: z %lld plane %lld n %lld m %lld: got %04x want %04x)", (long long)diff, (long long)z, (long long)pl,
                           (long long)(c8 * 8 + j), (long long)m, hcpl[first], hrefpl[first]);
                } else {
                    printf("  (%lld elements; first: z %lld m %lld n %lld: got %.9g want %.9g)", (long long)diff, (long long)(first / N / M),
                           (long long)(first / N % M), (long long)(first % N), hc[first], href[first]);
                }
                ++bad;
            }
            printf("\n");
        }
        CK(hipFree(da)); CK(hipFree(djoin)); CK(hipFree(dw)); CK(hipFree(dref)); CK(hipFree(dc)); CK(hipFree(dres)); CK(hipFree(dres2));
        CK(hipFree(dbias)); CK(hipFree(dscale)); CK(hipFree(dwsplit)); CK(hipFree(dapl)); CK(hipFree(dcpl)); CK(hipFree(drefpl));
    }
    printf(bad ? "PGEMM CHECK FAILED (%d)\n" : "PGEMM CHECK OK\n", bad);
    return bad ? 1 : 0;
}
