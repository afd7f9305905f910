// conv_gemm_inst0.hip — instantiates the group-0 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 0
#include "conv_gemm_inst.h"
