// GroupNorm of a conv's CONSUMER finalised inside the producing kernel by the last workgroup of each image -- instead of a gn_finalize launch (5.7 us + a kernel
// boundary, 17 times per UNet call on a strictly serial chain: skipping them altogether measured +4.7 % end to end, the upper bound of this file).
// OPT-IN (WDM_GN_INLINE=2), NOT the default: measured in round 4 (same box, 20 DDIM steps): 637.9 / 637.7 img/s with the launches, 614.3 / 615.2 with this file
// (-3.6 %), 678 with the arrival protocol alone (no finalize; garbage rows).  The protocol is free; what costs is the finalize itself: ONE workgroup per image
// reduces 32 groups x 256 ... 768 float4 partials in fp64 (+8.6 us on a 64 x 64 conv launch, +17 with a concat consumer, +76 on the sub-pixel upsample whose
// consumer regroups 384 channels), where the stand-alone kernel spreads them over 2 048 waves in 5.7 us.  What would make it pay: tile-level instead of 64-row
// partials (4x fewer items) -- a change of the statistics format in every epilogue; not built.  Kept because it is exact, tested and the evidence for the above.
//
// Protocol (no workgroup ever waits for another, so nothing can deadlock whatever the dispatch order or residency):
//   every workgroup   statistics stores go out write-through (conv_store_stat: sc0 sc1)  ->  s_waitcnt vmcnt(0) per wave  ->  workgroup barrier  ->
//                     ONE agent-scope atomic add of its tile count on the image's counter
//   the last arriver  (old + count == fin_total; exactly one workgroup per image sees that) reduces the image's groups with gn_group_stats<FRESH0 = true> -- the
//                     instruction sequence of gn_finalize_kernel over the same float4 partials in the same order, loads past the caches (sc0 sc1: the partials
//                     were written by other CUs / XCDs of this very launch) -- writes scale / shift like gn_finalize_kernel does and resets the counter.
// The result does not depend on which workgroup arrives last: bit-identical to the gn_finalize launch (tests/test_gpu_switches.py).  The consumer is a later
// kernel on the same stream: it sees scale / shift through the ordinary kernel boundary.
// Visibility rules used: MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility" ({sc0 sc1 stores and loads both sides}; the
// asm waitcnt between the stores and the atomic because the compiler may drop its own).
#pragma once
#include "gn_group.h"

namespace wdm {

// all NTHREADS threads of the workgroup; flag: 4 bytes of LDS nobody else touches until the next workgroup barrier.  img: image index, count: tiles of that image this
// workgroup has finished since its last arrival for it, HW: pixels per image of the conv's OUTPUT map
template <int NTHREADS, class AT>
__device__ __forceinline__ void gn_arrive(const AT& a, int img, int count, int HW, int* flag, int tid) {
    if (a.fin_cnt == nullptr || img >= a.B || count <= 0) return;              // (uniform over the workgroup)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // this wave's statistics stores have left the chip
    __syncthreads();
    if (tid == 0) *flag = __hip_atomic_fetch_add(a.fin_cnt + img, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int old = *flag;
    if (old + count != a.fin_total) return;
    constexpr int NW = NTHREADS / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const int C0 = a.Cout, C = a.Cout + a.fin_C1, gw = C >> 5;
    const float4* st0 = (const float4*)a.stats;
    const float4* st1 = a.fin_st1 ? (const float4*)a.fin_st1 : st0;
    const int ns1 = a.fin_st1 ? a.fin_nslab1 : 1;
    // a wave's groups: all their loads in flight before the first reduction
    constexpr int GPW = (32 + NW - 1) / NW;
    GnGroupLoad L[GPW];
#pragma unroll
    for (int k = 0; k < GPW; ++k) { const int g = wave + k * NW; if (g < 32) gn_group_load<true>(st0, a.stats_nslab, C0, st1, ns1, C, g, img, lane, L[k]); }
#pragma unroll
    for (int k = 0; k < GPW; ++k) {
        const int g = wave + k * NW;
        if (g >= 32) continue;
        float mean, rstd;
        gn_group_reduce<true>(st0, a.stats_nslab, C0, st1, ns1, C, HW, a.fin_eps, img, lane, L[k], mean, rstd);
        for (int ci = lane; ci < gw; ci += 64) {
            const int c = g * gw + ci;
            float sc, sh;
            gn_scale_shift(mean, rstd, a.fin_gamma[c], a.fin_beta[c], sc, sh);
            a.fin_scale[(long long)img * C + c] = sc * a.fin_premul;
            a.fin_shift[(long long)img * C + c] = sh * a.fin_premul;
        }
    }
    if (tid == 0) __hip_atomic_store(a.fin_cnt + img, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
}

}  // namespace wdm
