// conv_gemm_inst1.hip — instantiates the group-1 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 1
#include "conv_gemm_inst.h"
