// mlp_generic_native.hip — the native-fp32 (v_mfma_f32_32x32x2_f32) instantiations of the runtime-shaped kernels: a third
// translation unit of mlp_generic.hip (NFX_PREC_FP32_NATIVE; see that file).
#define NFX_GENERIC_TU 2
#include "mlp_generic.hip"
