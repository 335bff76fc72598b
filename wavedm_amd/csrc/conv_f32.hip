#define WDM_T float
#define WDM_LAUNCH_NAME launch_conv_f32
#include "conv_dispatch.inc"
