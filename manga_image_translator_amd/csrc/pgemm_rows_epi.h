// pgemm_rows_epi.h — the epilogue of the few-row planar GEMMs (pgemm_rows_kernel in pgemm.hip, pgemm_rows_ln_kernel in
// pgemm_rows_ln.hip): one definition, so both apply bit for bit the same arithmetic per element.
#pragma once
#include "conv_gemm_kernels.h"
#include "pgemm_rows.h"

namespace mitcg {

template <int ACT>
__device__ __forceinline__ f32x4 pgr_act4(f32x4 v, const float alpha) {
    v.x = apply_act<ACT>(v.x, alpha);
    v.y = apply_act<ACT>(v.y, alpha);
    v.z = apply_act<ACT>(v.z, alpha);
    v.w = apply_act<ACT>(v.w, alpha);
    return v;
}

// acc: the TRANSPOSED 32 x 32 block (MFMA rows = output columns) of one wave: lane (li, lh) holds row m0 + li, columns
// n0 + (r & 3) + 4 lh + 8 (r >> 2); after v_permlane32_swap the cells 2 pp + lh (eight consecutive columns of one row): 16-byte fp32
// stores and / or a split into planes.
__device__ __forceinline__ void pg_rows_epilogue(const MitPGemm &p, const PgRowsExt &x, const f32x16 &acc, const int m0, const int n0, const int z,
                                                 const int lane) {
    const int li = lane & 31, lh = lane >> 5;
    const int m = m0 + li;
    const int act = p.act & 0xff;
    const bool post_first = (p.act & MIT_ACT_POST_FIRST) != 0;
    float *cbase = p.c ? p.c + (int64_t)z * p.c_zs + (x.dyn ? (int64_t)(*x.dyn) * x.c_dyn : 0) : nullptr;
    u32x4 *pl_out = p.c_planes ? reinterpret_cast<u32x4 *>(p.c_planes + (int64_t)z * p.cp_zs) : reinterpret_cast<u32x4 *>(x.also_planes);
    const int64_t pl_ld = p.c_planes ? p.ld_cp : x.also_ld;
    const int64_t pl_plane = (int64_t)(p.N >> 3) * pl_ld;
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int n = n0 + 8 * (2 * pp + lh);
        f32x4 v0, v1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * pp + q]), __float_as_uint(acc[8 * pp + 4 + q]), false, false);
            v0[q] = __uint_as_float(sw[0]);
            v1[q] = __uint_as_float(sw[1]);
        }
        if (n >= p.N || m >= p.M) continue;
        const bool ok1 = n + 4 < p.N;   // N % 4 == 0: the cell's second half may lie past N (fp32 output only; planar output has N % 8 == 0)
        f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, bi0 = {0.f, 0.f, 0.f, 0.f}, bi1 = bi0;
        if (p.scale) sc0 = *reinterpret_cast<const f32x4 *>(p.scale + n);
        if (p.scale && ok1) sc1 = *reinterpret_cast<const f32x4 *>(p.scale + n + 4);
        if (p.bias) bi0 = *reinterpret_cast<const f32x4 *>(p.bias + n);
        if (p.bias && ok1) bi1 = *reinterpret_cast<const f32x4 *>(p.bias + n + 4);
        if (p.pre) {
            const float *q = p.pre + (int64_t)z * p.pre_zs + (int64_t)m * p.ld_pre + n;
            v0 += *reinterpret_cast<const f32x4 *>(q);
            if (ok1) v1 += *reinterpret_cast<const f32x4 *>(q + 4);
        }
        v0 = v0 * sc0 + bi0;
        v1 = v1 * sc1 + bi1;
        f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0;
        if (p.post) {
            const float *q = p.post + (int64_t)z * p.post_zs + (int64_t)m * p.ld_post + n;
            r0 = *reinterpret_cast<const f32x4 *>(q);
            if (ok1) r1 = *reinterpret_cast<const f32x4 *>(q + 4);
        }
        if (p.post && post_first) v0 += r0, v1 += r1;
        if (act == MIT_ACT_RELU) v0 = pgr_act4<MIT_ACT_RELU>(v0, p.act_alpha), v1 = pgr_act4<MIT_ACT_RELU>(v1, p.act_alpha);
        else if (act == MIT_ACT_GELU) v0 = pgr_act4<MIT_ACT_GELU>(v0, p.act_alpha), v1 = pgr_act4<MIT_ACT_GELU>(v1, p.act_alpha);
        if (p.post && !post_first) v0 += r0, v1 += r1;
        if (cbase) {
            const int64_t col = x.nsplit ? (int64_t)(n / x.nsplit) * x.nhi + (n % x.nsplit) : n;
            float *o = cbase + (int64_t)m * p.ldc + col;
            *reinterpret_cast<f32x4 *>(o) = v0;
            if (ok1) *reinterpret_cast<f32x4 *>(o + 4) = v1;
        }
        if (pl_out) {
            u32x4 h, md, l;
            split8(v0, v1, h, md, l);
            u32x4 *o = pl_out + (int64_t)(n >> 3) * pl_ld + m;
            o[0] = h;
            o[pl_plane] = md;
            o[2 * pl_plane] = l;
        }
    }
}

}  // namespace mitcg
