// conv_gemm_inst5.hip — instantiates the group-5 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 5
#include "conv_gemm_inst.h"
