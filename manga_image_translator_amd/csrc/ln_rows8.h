// ln_rows8.h — LayerNorm of a 320-wide row by EIGHT LANES (lane q of the row's aligned group of eight holds the k-cells 8 b + q, b < 5,
// i.e. the elements d = 64 b + 8 q + j), bit for bit layernorm_kernel's result (ocr_kernels.hip: lane l of a wave sums d = l, l + 64, ...
// from 0 and the 64 partials go through an xor butterfly 32, 16, ..., 1): the partials l = 8 q + j are this lane's, butterfly levels
// 32, 16, 8 pair q with q ^ 4, q ^ 2, q ^ 1 (ds_swizzle / DPP quad permutes inside the group), levels 4, 2, 1 pair j with j ^ 4, j ^ 2,
// j ^ 1 (registers).  Shared by pgemm_rows_ln.hip and the q-projecting cross-attention of ocr_kernels.hip — both compiled without
// packed-fp32 code generation: the statistics are scalar fp32 chains.
#pragma once
#include <hip/hip_runtime.h>
#include "bf16_split.h"

namespace mitln {

constexpr int LN_K = 320;

__device__ __forceinline__ float lane_xor1(const float v) {   // quad_perm [1, 0, 3, 2]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_xor2(const float v) {   // quad_perm [2, 3, 0, 1]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_xor4(const float v) {   // ds_swizzle, bit mode: and 0x1f, or 0, xor 4
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (4 << 10) | 0x1f));
}

// xr = the row + 8 q, gw / gb = LayerNorm weight / bias + 8 q.  On return v[b] = the normalised cell 8 b + q (eight consecutive values).
// Every lane of the wave must call it (the exchanges are wave-wide); lanes of a group must hold the same row.
__device__ __forceinline__ void ln_row_cells(const float *xr, const float *gw, const float *gb, const float eps, f32x4 (&v)[5][2]) {
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        v[b][0] = *reinterpret_cast<const f32x4 *>(xr + 64 * b);
        v[b][1] = *reinterpret_cast<const f32x4 *>(xr + 64 * b + 4);
    }
    f32x4 wv[5][2], bv[5][2];   // requested before the statistics
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            wv[b][hf] = *reinterpret_cast<const f32x4 *>(gw + 64 * b + 4 * hf);
            bv[b][hf] = *reinterpret_cast<const f32x4 *>(gb + 64 * b + 4 * hf);
        }
    __builtin_amdgcn_sched_barrier(0);
    auto reduce = [&](float (&pt)[8]) __attribute__((always_inline)) -> float {
#pragma unroll
        for (int j = 0; j < 8; ++j) pt[j] += lane_xor4(pt[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) pt[j] += lane_xor2(pt[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) pt[j] += lane_xor1(pt[j]);
        const float r0 = pt[0] + pt[4], r1 = pt[1] + pt[5], r2 = pt[2] + pt[6], r3 = pt[3] + pt[7];
        return (r0 + r2) + (r1 + r3);
    };
    float pt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float sm = 0.f;
#pragma unroll
        for (int b = 0; b < 5; ++b) sm += v[b][j >> 2][j & 3];
        pt[j] = sm;
    }
    const float mean = reduce(pt) / (float)LN_K;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float sm = 0.f;
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            const float tt = v[b][j >> 2][j & 3] - mean;
            sm += tt * tt;
        }
        pt[j] = sm;
    }
    const float rstd = 1.0f / sqrtf(reduce(pt) / (float)LN_K + eps);
#pragma unroll
    for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[b][hf][c] = (v[b][hf][c] - mean) * rstd * wv[b][hf][c] + bv[b][hf][c];
}

}  // namespace mitln
