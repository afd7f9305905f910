// conv_gemm_inst3.hip — instantiates the group-3 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 3
#include "conv_gemm_inst.h"
