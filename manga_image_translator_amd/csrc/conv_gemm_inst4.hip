// conv_gemm_inst4.hip — instantiates the group-4 tile configurations of conv_gemm_cfgs.inc (see conv_gemm_inst.h).
#define MIT_INST_GROUP 4
#include "conv_gemm_inst.h"
