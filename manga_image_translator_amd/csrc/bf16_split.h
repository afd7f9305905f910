// bf16_split.h — x = hi + mid + lo: an fp32 value as three bf16 planes, the operand form of the split-bf16 GEMM tiles (conv_gemm_split.h,
// pgemm.hip).  Shared by the tiles and by the kernels that PRODUCE planes for them (pgemm.hip, ocr_kernels.hip): one definition, so a
// producer's planes are bit for bit the ones a tile would form from the fp32 value itself.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace mitcg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32: a in the low half, round to nearest even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned int pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int pk) { return __uint_as_float(pk & 0xffff0000u); }

template <bool ASM_SUB>
__device__ __forceinline__ float sub_f32(float a, float b) {
    if (ASM_SUB) {
        float r;
        asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    return a - b;
}
template <bool ASM_SUB = false>
__device__ __forceinline__ void split3(const f32x4 x, u32x2 &h, u32x2 &m, u32x2 &l) {
    h.x = pack_bf16(x.x, x.y);
    h.y = pack_bf16(x.z, x.w);
    const float r0 = sub_f32<ASM_SUB>(x.x, bf16_lo(h.x)), r1 = sub_f32<ASM_SUB>(x.y, bf16_hi(h.x));
    const float r2 = sub_f32<ASM_SUB>(x.z, bf16_lo(h.y)), r3 = sub_f32<ASM_SUB>(x.w, bf16_hi(h.y));
    m.x = pack_bf16(r0, r1);
    m.y = pack_bf16(r2, r3);
    const float s0 = sub_f32<ASM_SUB>(r0, bf16_lo(m.x)), s1 = sub_f32<ASM_SUB>(r1, bf16_hi(m.x));
    const float s2 = sub_f32<ASM_SUB>(r2, bf16_lo(m.y)), s3 = sub_f32<ASM_SUB>(r3, bf16_hi(m.y));
    l.x = pack_bf16(s0, s1);
    l.y = pack_bf16(s2, s3);
}

// eight consecutive values -> their three 16-byte cells
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, u32x4 &h, u32x4 &m, u32x4 &l) {
    u32x2 h0, m0, l0, h1, m1, l1;
    split3<false>(lo, h0, m0, l0);
    split3<false>(hi, h1, m1, l1);
    h = u32x4{h0.x, h0.y, h1.x, h1.y};
    m = u32x4{m0.x, m0.y, m1.x, m1.y};
    l = u32x4{l0.x, l0.y, l1.x, l1.y};
}

// plane pairs of a split product, smallest products first; a tile with NPROD products takes the last NPROD entries (conv_gemm_split.h)
__device__ constexpr int kSplitPA[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
__device__ constexpr int kSplitPB[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};

}  // namespace mitcg
