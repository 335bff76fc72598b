#define WDM_T float
#define WDM_PAIR_NAME launch_gemm_pair_f32
#define WDM_LAUNCH_NAME launch_conv_f32
#include "conv_dispatch.inc"
