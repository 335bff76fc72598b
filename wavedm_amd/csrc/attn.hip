// Launcher of the fused attention core (attn_fused_kernel.h).
#include <atomic>
#include <type_traits>

#include "common.h"
#include "attn_fused_kernel.h"

namespace wdm {

bool attn_fused_eligible(int dtype, int N, int C) {
    // one phase-2 pass over <= 512 channels, two over the halves above that (the halves must be whole wave fragments: multiples of 128)
    return env_cfg().attn_fused && is_h16(dtype) && N == AttnFusedCfg::N && C % 128 == 0 && C >= 128 && C <= AttnFusedCfg::MAX_C &&
           (C <= AttnFusedCfg::MAX_CP || (C / 2) % 128 == 0);
}

template <typename T, bool VTOK>
static int launch_attn_fused_t(const AttnOperands& in, void* o, int B, int C, hipStream_t s, const float* vbias, const ConvArgs* proj) {
    using Cf = AttnFusedCfg;
    const bool qproj = in.qw != nullptr;
    if ((!in.q && !qproj) || !in.k || !in.v || (!o && !proj) || B <= 0) WDM_FAIL(WDM_EINVAL, "attn(fused): bad argument");
    if (qproj && (!VTOK || !proj || in.q || in.k != in.v || in.qw_ld < C || in.qw_ld % 8 || in.qw_bytes == 0 || in.qw_bytes >= 4294901760.0))
        WDM_FAIL(WDM_EINVAL, "attn(fused): the in-kernel query projection needs the folded block's operands (K = V = the normalised input, proj_out fused)");
    if ((!qproj && in.q_ld < C) || in.k_ld < C || ((qproj ? 0 : in.q_ld) | in.k_ld) % 8 || (VTOK && (in.v_ld < C || in.v_ld % 8))) WDM_FAIL(WDM_EINVAL, "attn(fused): row strides must be multiples of 8 elements >= C");
    if (proj && (C > 512 || proj->Cout != C || proj->Cin != C || proj->Hout != 16 || proj->Wout != 16 || proj->B != B || !proj->w || proj->w_bytes == 0 || proj->y_mode != Y_NHWC ||
                 proj->temb || proj->m_valid || proj->up4 || (proj->stats && proj->stats_nslab != 4)))
        WDM_FAIL(WDM_EINVAL, "attn(fused): proj_out epilogue arguments do not describe a C = %d 1x1 conv on 16 x 16 maps", C);
    if (in.bdiag && (!VTOK || proj || qproj)) WDM_FAIL(WDM_EINVAL, "attn(fused): the block-diagonal form (64 tokens per image) takes token-major V and writes O");
    const int NT = in.bdiag ? 64 : Cf::N;                 // tokens per real image
    const int G = in.bdiag ? (B + 3) / 4 : B;             // "images" of the kernel
    const double qb = (double)B * NT * in.q_ld * 2.0, kb = (double)B * NT * in.k_ld * 2.0, vb = VTOK ? (double)B * NT * in.v_ld * 2.0 : (double)B * C * Cf::N * 2.0;
    if (qb >= 4294901760.0 || kb >= 4294901760.0 || vb >= 4294901760.0) WDM_FAIL(WDM_EINVAL, "attn(fused): an operand exceeds the 4 GB buffer-offset range");
    AttnFusedArgs a{};
    a.q = in.q; a.k = in.k; a.v = in.v; a.o = o; a.vbias = vbias; a.B = G; a.C = C; a.bdiag = in.bdiag; a.nimg = B;
    a.q_ld = in.q_ld; a.k_ld = in.k_ld; a.v_ld = in.v_ld;
    a.alpha = (float)std::pow((double)C, -0.5);
    // extents behind the three pointers (q and k may be column ranges of one tensor: the last row ends C elements behind its start)
    a.qw = in.qw; a.qbias = in.qbias; a.qw_ld = in.qw_ld; a.qw_bytes = (unsigned)in.qw_bytes; a.qw_slab = in.qw_slab;
    a.q_bytes = qproj ? 0u : (unsigned)(qb - (in.q_ld - C) * 2.0); a.k_bytes = (unsigned)(kb - (in.k_ld - C) * 2.0); a.v_bytes = (unsigned)(VTOK ? vb - (in.v_ld - C) * 2.0 : vb);
    static std::atomic<unsigned> devs{0};
    int dev = 0;
    WDM_HIP(hipGetDevice(&dev));
    if (!(devs.load() & (1u << (dev & 31)))) {
        WDM_HIP(hipFuncSetAttribute((const void*)attn_fused_kernel<false, T, VTOK>, hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS_BYTES));
        WDM_HIP(hipFuncSetAttribute((const void*)attn_fused_kernel<true, T, VTOK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if constexpr (VTOK) WDM_HIP(hipFuncSetAttribute((const void*)attn_fused_kernel<true, T, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        devs.fetch_or(1u << (dev & 31));
    }
    const bool prof = prof_enabled();
    if (prof) {
        char name[96];
        snprintf(name, sizeof(name), "attn_fused_n%s%s%s_%s|%s C=%d%s%s", in.bdiag ? "64x4" : "256", VTOK ? "t" : "", qproj ? "q" : "", std::is_same<T, f16_t>::value ? "f16" : "bf16",
                 in.bdiag ? "8x8" : "16x16", C, qproj ? " qproj+" : "", proj ? " +proj" : "");
        prof_begin(s, name, 4.0 * B * NT * (double)NT * C + (proj ? 2.0 * B * Cf::N * (double)C * C : 0.0) + (qproj ? 2.0 * B * Cf::N * (double)C * C : 0.0),
                   (double)B * NT * C * 2.0 * (qproj ? 3.0 : proj ? 5.0 : 4.0) + (proj ? (double)C * C * 2.0 : 0.0) + (qproj ? (double)C * C * 2.0 : 0.0));
    }
    ConvArgs pe{};
    if (proj) pe = *proj;
    pe.fin_total = AttnFusedCfg::N / AttnFusedCfg::QB;          // gn_arrive.h: the image's query blocks
    if (qproj) { if constexpr (VTOK) hipLaunchKernelGGL((attn_fused_kernel<true, T, true, true>), dim3(((G + 7) / 8) * 32), dim3(Cf::NTHREADS), 160 * 1024, s, a, pe); }
    else if (proj) hipLaunchKernelGGL((attn_fused_kernel<true, T, VTOK>), dim3(((G + 7) / 8) * 32), dim3(Cf::NTHREADS), 160 * 1024, s, a, pe);
    else hipLaunchKernelGGL((attn_fused_kernel<false, T, VTOK>), dim3(((G + 7) / 8) * 32), dim3(Cf::NTHREADS), Cf::LDS_BYTES, s, a, pe);      // 8 images x 4 query blocks per group of 32
    if (prof) prof_end(s);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
int launch_attn_fused(const AttnOperands& in, void* o, int B, int C, hipStream_t s, const float* vbias, const ConvArgs* proj, int dtype) {
    if (dtype == WDM_F16) return in.v_tok ? launch_attn_fused_t<f16_t, true>(in, o, B, C, s, vbias, proj) : launch_attn_fused_t<f16_t, false>(in, o, B, C, s, vbias, proj);
    if (dtype == WDM_BF16) return in.v_tok ? launch_attn_fused_t<__bf16, true>(in, o, B, C, s, vbias, proj) : launch_attn_fused_t<__bf16, false>(in, o, B, C, s, vbias, proj);
    WDM_FAIL(WDM_EINVAL, "attn(fused): 16-bit modes only");
}

}  // namespace wdm
