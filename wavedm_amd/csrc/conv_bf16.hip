#define WDM_T __bf16
#define WDM_LAUNCH_NAME launch_conv_bf16
#define WDM_HAS_GEMM 1
#include "conv_gemm_kernel.h"
#include "conv_dma_kernel.h"
#include "conv_dma256_kernel.h"
#include "conv_up4_kernel.h"
#include "conv_dma8_kernel.h"
#include "conv_s2_kernel.h"
#include "conv_dispatch.inc"
