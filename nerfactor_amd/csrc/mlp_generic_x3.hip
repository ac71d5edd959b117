// mlp_generic_x3.hip — the fp32-class (bf16 hi / lo operand pair) instantiations of the runtime-shaped kernels: a second
// translation unit of mlp_generic.hip (NFX_PREC_FP32; see that file), so that the three operand modes compile in parallel.
#define NFX_GENERIC_TU 1
#include "mlp_generic.hip"
