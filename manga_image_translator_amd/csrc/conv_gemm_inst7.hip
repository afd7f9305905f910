// conv_gemm_inst7.hip — instantiates the group-7 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 7
#include "conv_gemm_inst.h"
