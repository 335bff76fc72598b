// GroupNorm finalised in the PROLOGUE of the consuming convolution (LDS-DMA 3x3 kernels, bf16): instead of a gn_finalize launch between producer and
// consumer (5.7 us + a kernel boundary, 21 times per UNet call on a strictly serial chain), the producer's epilogue also leaves GROUP-level partial
// statistics (conv_kernel.h: ConvArgs::gst, 12 bytes per (image, 64-row slab, group)) and every workgroup of the consumer turns its image's partials --
// 1.5 KB (16 x 16 maps) ... 24 KB (64 x 64) -- into the scale / shift table it used to fetch:
//   issue:  the image's partials, gamma and beta go to LDS by DMA like every other operand of these kernels (no compiler-visible load in the prologue);
//   table:  16 lanes per group walk the slabs in ascending order (fp64, re-centred on slab 0's pivot), a fixed xor tree joins them, the group's lanes
//           write scale = rstd * gamma * premul, shift = (beta - mean * rstd * gamma) * premul  (premul = -log2 e: conv_kernel.h gn_silu_unit).
// No atomics, no cross-workgroup traffic: an image's table depends on nothing but its own partials, whoever computes it.
// Single-input consumers with Cin = 128 / 256 / 512 (group width 4 / 8 / 16 <= the 16 lanes of a group); channel-concat inputs, whose groups straddle
// the seam, and the pass consumers keep the per-channel partials and gn_finalize / gn_finalize_apply.
#pragma once
#include "conv_kernel.h"

namespace wdm {

constexpr int GN_INLINE_MAX_BYTES = 24 * 1024;        // scratch: the second halo buffer, idle until the K loop starts
constexpr int GN_INLINE_MAX_NSLAB = 16;               // measured (batch 64): +1.2 ... 1.7 us per launch at 4 / 16 slabs (16 x 16 / 32 x 32 maps) against the ~7 us of a
                                                      // gn_finalize launch and its boundary; at 64 slabs (64 x 64 maps: 24 KB of partials, +2.3 us per table) +9.3 us per
                                                      // launch with a table per tile (round 3), level with gn_finalize with a table per image (round 4, persistent kernel
                                                      // walking an image's tiles back to back) and again level with it on the 512 x 128 tile (two tables per CU: 611.2 / 611.1 / 611.6 img/s with the launches,
                                                      // 613.6 / 609.3 / 610.1 without): those keep gn_finalize
__host__ __device__ inline bool gn_inline_shape_ok(int cin, int nslab) {
    return (cin == 128 || cin == 256 || cin == 512) && nslab >= 1 && nslab <= GN_INLINE_MAX_NSLAB && nslab * 384 <= GN_INLINE_MAX_BYTES;
}

// DMA requests of one wave: its share of the image's partials (<= 3 pieces of 1 KB), waves 0, 1: gamma, waves 2, 3: beta (<= 2 KB each).
// scr_lds / tab_lds: LDS byte addresses of the scratch area and of the scale / shift table; gamma and beta are parked 2 KB behind the scale and the shift
// row (both rows are MAX_CIN floats long, Cin <= 512 uses the first 2 KB of each).
template <int MAX_CIN, class AT, class DMA, class MAKEQ>
__device__ __forceinline__ void gn_inline_issue(const AT& a, int img0, int wave, int lane, unsigned scr_lds, unsigned tab_lds, DMA&& dma16, MAKEQ&& make_q) {
    const unsigned bytes = (unsigned)(a.gin_nslab * 384);
    const auto q_g = make_q(a.gin + (long long)img0 * a.gin_nslab * 96, bytes);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned off = (unsigned)((wave * 3 + i) * 1024);
        if (off < bytes) dma16(q_g, scr_lds + off, off + (unsigned)(lane * 16), 0);
    }
    if (wave < 4) {
        const auto q_v = make_q(wave < 2 ? a.gn_gamma : a.gn_beta, (unsigned)(a.Cin * 4));
        const unsigned off = (unsigned)((wave & 1) * 1024);
        if (off < (unsigned)(a.Cin * 4)) dma16(q_v, tab_lds + (wave < 2 ? 2048u : (unsigned)(MAX_CIN * 4 + 2048)) + off, off + (unsigned)(lane * 16), 0);
    }
}

// after the requests have landed and a workgroup barrier: all 512 threads; the caller puts a workgroup barrier behind it before the table is read
template <int MAX_CIN>
__device__ __forceinline__ void gn_inline_table(const float* gs_lds, float* tab, int nslab, int cin, int hw, float eps, int tid) {
    const int g = tid >> 4, j = tid & 15;
    const int gs = cin >> 5;
    const double P = (double)gs_lds[g * 3];
    const double n = 64.0 * gs;
    double A1 = 0.0, A2 = 0.0;
    for (int s = j; s < nslab; s += 16) {
        const float* e = gs_lds + (s * 32 + g) * 3;
        const double d = (double)e[0] - P, s1 = (double)e[1], s2 = (double)e[2];
        A1 += s1 + n * d;
        A2 += s2 + 2.0 * d * s1 + n * d * d;
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { A1 += __shfl_xor(A1, o); A2 += __shfl_xor(A2, o); }
    const double N = (double)gs * (double)hw;
    const double m = A1 / N;
    double var = A2 / N - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)(P + m);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (j < gs) {
        const int c = g * gs + j;
        const float sc = rstd * tab[512 + c];                        // gamma parked 2 KB into the scale row
        tab[c] = sc * -1.4426950408889634f;
        tab[MAX_CIN + c] = (tab[MAX_CIN + 512 + c] - mean * sc) * -1.4426950408889634f;      // beta parked 2 KB into the shift row
    }
}

}  // namespace wdm
