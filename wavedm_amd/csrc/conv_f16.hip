// WDM_F16: the kernels of conv_bf16.hip instantiated on IEEE half operands (same LDS maps, rings and schedules; v_mfma_f32_16x16x32_f16)
#define WDM_T f16_t
#define WDM_LAUNCH_NAME launch_conv_f16
#define WDM_DTYPE_NAME "f16"
#define WDM_H16_NAME "f16"
#define WDM_HAS_GEMM 1
#include "conv_gemm_kernel.h"
#include "conv_dma_kernel.h"
#include "conv_dma256_kernel.h"
#include "conv_up4_kernel.h"
#include "conv_dma8_kernel.h"
#include "conv_s2_kernel.h"
#include "conv_dispatch.inc"
