// conv_gemm_inst1.hip — instantiates the group-1 tile configurations of conv_gemm_cfgs.inc (parallel compilation).
#include "conv_gemm_kernels.h"

#define MIT_INST_1(BM, BN, BK, fn, ...) \
    template void mitcg::fn<BM, BN, BK, __VA_ARGS__>(const MitConvGemm &, int, int, int, int, hipStream_t);
#ifndef MIT_INST_0
#define MIT_INST_0(...)
#endif
#ifndef MIT_INST_1
#define MIT_INST_1(...)
#endif
#ifndef MIT_INST_2
#define MIT_INST_2(...)
#endif
#ifndef MIT_INST_3
#define MIT_INST_3(...)
#endif
#ifndef MIT_INST_4
#define MIT_INST_4(...)
#endif
#ifndef MIT_INST_5
#define MIT_INST_5(...)
#endif
#ifndef MIT_INST_6
#define MIT_INST_6(...)
#endif
#ifndef MIT_INST_7
#define MIT_INST_7(...)
#endif
#define X(g, name, fast, BM, BN, BK, fn, ...) MIT_INST_##g(BM, BN, BK, fn, __VA_ARGS__)
#include "conv_gemm_cfgs.inc"
#undef X
