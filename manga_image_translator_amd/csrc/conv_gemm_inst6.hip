// conv_gemm_inst6.hip — instantiates the group-6 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 6
#include "conv_gemm_inst.h"
