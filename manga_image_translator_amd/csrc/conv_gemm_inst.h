// conv_gemm_inst.h — body of the conv_gemm_inst<N>.hip translation units: each defines MIT_INST_GROUP = N and includes this file,
// which instantiates the launchers of the group-N tile configurations of conv_gemm_cfgs.inc (parallel compilation only).
#include "conv_gemm_kernels.h"

#define MIT_INST_YES(BM, BN, BK, fn, ...) \
    template void mitcg::fn<BM, BN, BK, __VA_ARGS__>(const MitConvGemm &, int, int, int, int, hipStream_t);
#define MIT_INST_NO(...)
#define MIT_INST_CAT2(a, b) a##b
#define MIT_INST_CAT(a, b) MIT_INST_CAT2(a, b)
// MIT_INST_EQ_<g>_<group> is MIT_INST_YES when g == group, MIT_INST_NO otherwise
#define MIT_INST_PICK(g) MIT_INST_CAT(MIT_INST_CAT(MIT_INST_EQ_, g), MIT_INST_CAT(_, MIT_INST_GROUP))
#define MIT_INST_EQ_0_0 MIT_INST_YES
#define MIT_INST_EQ_0_1 MIT_INST_NO
#define MIT_INST_EQ_0_2 MIT_INST_NO
#define MIT_INST_EQ_0_3 MIT_INST_NO
#define MIT_INST_EQ_0_4 MIT_INST_NO
#define MIT_INST_EQ_0_5 MIT_INST_NO
#define MIT_INST_EQ_0_6 MIT_INST_NO
#define MIT_INST_EQ_0_7 MIT_INST_NO
#define MIT_INST_EQ_1_0 MIT_INST_NO
#define MIT_INST_EQ_1_1 MIT_INST_YES
#define MIT_INST_EQ_1_2 MIT_INST_NO
#define MIT_INST_EQ_1_3 MIT_INST_NO
#define MIT_INST_EQ_1_4 MIT_INST_NO
#define MIT_INST_EQ_1_5 MIT_INST_NO
#define MIT_INST_EQ_1_6 MIT_INST_NO
#define MIT_INST_EQ_1_7 MIT_INST_NO
#define MIT_INST_EQ_2_0 MIT_INST_NO
#define MIT_INST_EQ_2_1 MIT_INST_NO
#define MIT_INST_EQ_2_2 MIT_INST_YES
#define MIT_INST_EQ_2_3 MIT_INST_NO
#define MIT_INST_EQ_2_4 MIT_INST_NO
#define MIT_INST_EQ_2_5 MIT_INST_NO
#define MIT_INST_EQ_2_6 MIT_INST_NO
#define MIT_INST_EQ_2_7 MIT_INST_NO
#define MIT_INST_EQ_3_0 MIT_INST_NO
#define MIT_INST_EQ_3_1 MIT_INST_NO
#define MIT_INST_EQ_3_2 MIT_INST_NO
#define MIT_INST_EQ_3_3 MIT_INST_YES
#define MIT_INST_EQ_3_4 MIT_INST_NO
#define MIT_INST_EQ_3_5 MIT_INST_NO
#define MIT_INST_EQ_3_6 MIT_INST_NO
#define MIT_INST_EQ_3_7 MIT_INST_NO
#define MIT_INST_EQ_4_0 MIT_INST_NO
#define MIT_INST_EQ_4_1 MIT_INST_NO
#define MIT_INST_EQ_4_2 MIT_INST_NO
#define MIT_INST_EQ_4_3 MIT_INST_NO
#define MIT_INST_EQ_4_4 MIT_INST_YES
#define MIT_INST_EQ_4_5 MIT_INST_NO
#define MIT_INST_EQ_4_6 MIT_INST_NO
#define MIT_INST_EQ_4_7 MIT_INST_NO
#define MIT_INST_EQ_5_0 MIT_INST_NO
#define MIT_INST_EQ_5_1 MIT_INST_NO
#define MIT_INST_EQ_5_2 MIT_INST_NO
#define MIT_INST_EQ_5_3 MIT_INST_NO
#define MIT_INST_EQ_5_4 MIT_INST_NO
#define MIT_INST_EQ_5_5 MIT_INST_YES
#define MIT_INST_EQ_5_6 MIT_INST_NO
#define MIT_INST_EQ_5_7 MIT_INST_NO
#define MIT_INST_EQ_6_0 MIT_INST_NO
#define MIT_INST_EQ_6_1 MIT_INST_NO
#define MIT_INST_EQ_6_2 MIT_INST_NO
#define MIT_INST_EQ_6_3 MIT_INST_NO
#define MIT_INST_EQ_6_4 MIT_INST_NO
#define MIT_INST_EQ_6_5 MIT_INST_NO
#define MIT_INST_EQ_6_6 MIT_INST_YES
#define MIT_INST_EQ_6_7 MIT_INST_NO
#define MIT_INST_EQ_7_0 MIT_INST_NO
#define MIT_INST_EQ_7_1 MIT_INST_NO
#define MIT_INST_EQ_7_2 MIT_INST_NO
#define MIT_INST_EQ_7_3 MIT_INST_NO
#define MIT_INST_EQ_7_4 MIT_INST_NO
#define MIT_INST_EQ_7_5 MIT_INST_NO
#define MIT_INST_EQ_7_6 MIT_INST_NO
#define MIT_INST_EQ_7_7 MIT_INST_YES
#define X(g, name, fast, BM, BN, BK, fn, ...) MIT_INST_PICK(g)(BM, BN, BK, fn, __VA_ARGS__)
#include "conv_gemm_cfgs.inc"
#undef X
