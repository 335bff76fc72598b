// The AttnBlock's q|k projection and V^T = W_v . h^T in ONE launch (round 3, WDM_GEMM_PAIR=1): built, bit-identical, measured null end to end
// (244.96 / 245.06 vs 245.14 / 244.43 img/s at 50 steps: 256 + 128 workgroups of 160 KB LDS still run one after the other on the 256 CUs, only a kernel
// boundary is saved).  Removed from the library in round 4; kept here as evidence (EXPERIMENTS.md).  Needs conv_gemm_kernel.h's conv_gemm_body.
#pragma once
#include "conv_gemm_kernel.h"
namespace wdm {
// TWO independent GEMMs in one launch: blocks [0, nblk0) work on a0 (wave tiles of WN0 fragments), the rest on a1 -- the AttnBlock's q|k projection and
// its V^T = W_v . h^T read the same normalised map and neither fills more than one round of workgroups, so side by side they share one launch, one
// fill and one drain of the chip.  nblk0 is a multiple of 8 (XCD-aware tile order of either grid).
template <int WN0, int WN1>
__global__ __launch_bounds__(512, 2) void conv_gemm_pair_kernel(const ConvArgs a0, const ConvArgs a1, const int nblk0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < nblk0) conv_gemm_body<16, 16, 1, 4, 2, 4, WN0>(a0, blockIdx.x, smem);
    else conv_gemm_body<16, 16, 1, 4, 2, 4, WN1>(a1, blockIdx.x - nblk0, smem);
}

}  // namespace wdm
