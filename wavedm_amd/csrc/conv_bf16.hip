#define WDM_T __bf16
#define WDM_LAUNCH_NAME launch_conv_bf16
#include "conv_dispatch.inc"
