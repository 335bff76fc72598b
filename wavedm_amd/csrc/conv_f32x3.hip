#define WDM_T f32x3_t
#define WDM_LAUNCH_NAME launch_conv_f32x3
#define WDM_DTYPE_NAME "f32x3"
#define WDM_HAS_DMAX3 1
#include "conv_dmax3_kernel.h"
#include "conv_dmax3t_kernel.h"
#include "conv_dma8x3_kernel.h"
#include "conv_up4x3_kernel.h"
#include "conv_gemmx3_kernel.h"
#include "conv_s2x3_kernel.h"
#include "conv_dispatch.inc"
