// conv_gemm_inst2.hip — instantiates the group-2 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 2
#include "conv_gemm_inst.h"
