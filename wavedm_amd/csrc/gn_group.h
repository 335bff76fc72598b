// GroupNorm from partial statistics: the group reduction and the per-element arithmetic shared by gn_finalize_kernel, gn_finalize_apply_kernel
// (elementwise.hip) and the conv epilogues' in-tile GroupNorm (gn_out_tail below), so that all of them produce the same bits.
#pragma once
#include "conv_kernel.h"

#ifndef WDM_FIN_LD_AUX
#define WDM_FIN_LD_AUX 17
#endif
namespace wdm {

// mean / rstd of group g of image b from the partial statistics of [x0 | x1]: the work of ONE wave (all 64 lanes take part; every lane returns the result),
// in two steps so that a wave that owns several groups can have all their loads in flight at once.  Shared by gn_finalize_kernel and
// gn_finalize_apply_kernel so that both produce the same bits: the accumulation order per lane (ascending item index) and the shuffle tree are fixed.
struct GnGroupLoad { float kgf; float4 v0[4]; int items, items0, n0c, cg0; };
// FRESH0: tensor 0's partials were written by OTHER workgroups of the kernel that is reading them now (gn_arrive.h): they were stored write-through
// (sc0 sc1) and must be loaded past the caches (sc0 sc1 as well) -- a plain load may be served by a stale L1 / L2 line (MI355X_MICROARCH.md, "Workgroup
// dispatch, XCD placement & inter-workgroup visibility").  Tensor 1 (the other half of a channel concat) always comes from an earlier kernel.
template <bool FRESH0>
__device__ __forceinline__ float4 gn_ld0(const float4* __restrict__ st0, long long idx) {
    if constexpr (FRESH0) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)st0, 0, 0x7FFFFFF0, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(idx * 16), 0, WDM_FIN_LD_AUX);          // aux 17 = sc0 | sc1
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    } else {
        return st0[idx];
    }
}
template <bool FRESH0 = false>
__device__ __forceinline__ float4 gn_load_item(const float4* __restrict__ st0, int nslab0, int C0, const float4* __restrict__ st1, int nslab1, int C1, int b,
                                               const GnGroupLoad& L, int it) {
    // slab counts are powers of two for every map of the model: shifts instead of ~35-instruction run-time divisions (four per item -- they were most of
    // this kernel's instruction count)
    auto divmod = [](int x, int d, int& q, int& r) __attribute__((always_inline)) {
        if ((d & (d - 1)) == 0) { const int sh = __builtin_ctz(d); q = x >> sh; r = x & (d - 1); } else { q = x / d; r = x - q * d; }
    };
    int ci, sl;
    if (it < L.items0) { divmod(it, nslab0, ci, sl); return gn_ld0<FRESH0>(st0, ((long long)b * nslab0 + sl) * C0 + L.cg0 + ci); }
    divmod(it - L.items0, nslab1, ci, sl);
    return st1[((long long)b * nslab1 + sl) * C1 + (L.cg0 + L.n0c + ci - C0)];
}
template <bool FRESH0 = false>
__device__ __forceinline__ void gn_group_load(const float4* __restrict__ st0, int nslab0, int C0, const float4* __restrict__ st1, int nslab1, int C, int g, int b, int lane,
                                              GnGroupLoad& L) {
    const int gw = C / 32, C1 = C - C0;
    L.cg0 = g * gw;
    // ONE memory round trip: the group's pivot (first channel, slab 0) and the first batch of partials are requested before anything is used
    L.kgf = (L.cg0 < C0) ? gn_ld0<FRESH0>(st0, ((long long)b * nslab0) * C0 + L.cg0).x : st1[((long long)b * nslab1) * C1 + (L.cg0 - C0)].x;
    // items = (channel of the group, slab of that channel's tensor); the slab counts of the two tensors may differ
    L.n0c = max(0, min(C0 - L.cg0, gw));        // channels of this group that live in tensor 0
    L.items0 = L.n0c * nslab0;
    L.items = L.items0 + (gw - L.n0c) * nslab1;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (lane + 64 * u < L.items) L.v0[u] = gn_load_item<FRESH0>(st0, nslab0, C0, st1, nslab1, C1, b, L, lane + 64 * u);
}
template <bool FRESH0 = false>
__device__ __forceinline__ void gn_group_reduce(const float4* __restrict__ st0, int nslab0, int C0, const float4* __restrict__ st1, int nslab1, int C, int HW, float eps,
                                                int b, int lane, const GnGroupLoad& L, float& mean, float& rstd) {
    const int gw = C / 32, C1 = C - C0;
    double S1 = 0.0, S2 = 0.0;
    const double kg = (double)L.kgf;
    auto accumulate = [&](const float4& v) __attribute__((always_inline)) {
        const double n = (double)v.w, d = (double)v.x - kg, a1 = (double)v.y, a2 = (double)v.z;
        S1 += a1 + n * d;
        S2 += a2 + 2.0 * d * a1 + n * d * d;
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) if (lane + 64 * u < L.items) accumulate(L.v0[u]);
    for (int it0 = lane + 256; it0 < L.items; it0 += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (it0 + 64 * u < L.items) v[u] = gn_load_item<FRESH0>(st0, nslab0, C0, st1, nslab1, C1, b, L, it0 + 64 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (it0 + 64 * u < L.items) accumulate(v[u]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        S1 += __shfl_xor(S1, o);
        S2 += __shfl_xor(S2, o);
    }
    const double N = (double)gw * (double)HW;
    const double m = S1 / N;
    double var = S2 / N - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)(kg + m);
    rstd = (float)(1.0 / sqrt(var + (double)eps));
}
template <bool FRESH0 = false>
__device__ __forceinline__ void gn_group_stats(const float4* __restrict__ st0, int nslab0, int C0, const float4* __restrict__ st1, int nslab1, int C, int HW, float eps,
                                               int g, int b, int lane, float& mean, float& rstd) {
    GnGroupLoad L;
    gn_group_load<FRESH0>(st0, nslab0, C0, st1, nslab1, C, g, b, lane, L);
    gn_group_reduce<FRESH0>(st0, nslab0, C0, st1, nslab1, C, HW, eps, b, lane, L, mean, rstd);
}

// scale / shift of one channel from its group's mean / rstd and the norm's weights (explicit fma: every caller rounds alike)
__device__ __forceinline__ void gn_scale_shift(float mean, float rstd, float gamma, float beta, float& scale, float& shift) {
    scale = rstd * gamma;
    shift = __builtin_fmaf(-mean, scale, beta);
}
// act(x * scale + shift) of one 16-byte vector; sc / sh point at the vector's first channel (LDS or global)
template <typename T>
__device__ __forceinline__ uint4 gn_apply_f8(float* f, const float* sc, const float* sh, int silu) {
    constexpr int VEC = TI<T>::VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) f[e] = __builtin_fmaf(f[e], sc[e], sh[e]);
    if (silu) {
        // bf16 outputs: v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~12 instructions per element: it was most of the pass's VALU time -- 64 K
        // elements per CU); the fp32 modes keep the division
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] = VEC == 8 ? f[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-f[e])) : f[e] / (1.0f + __expf(-f[e]));
    }
    return TI<T>::pack(f);
}
template <typename T>
__device__ __forceinline__ uint4 gn_apply_vec(const uint4& u, const float* sc, const float* sh, int silu) {
    float f[TI<T>::VEC];
    TI<T>::unpack(u, f);
    return gn_apply_f8<T>(f, sc, sh, silu);
}

// ------------------------------------------------------------------------------------------------
// In-tile GroupNorm of a conv's OUTPUT (ConvArgs::yn): where a workgroup's tile holds whole images x whole groups (the 8 x 8 kernel: two images x 48
// columns; the 16 x 16 maps on 256 x 128 tiles: one image x 128 columns) the consumer's act(GroupNorm(y)) is written by the producer itself, behind its
// epilogue -- the gn_finalize_apply launch (8.3 us + a kernel boundary) of the 8 x 8 ResnetBlocks and of the AttnBlocks disappears.
// Everything stays in LDS: the epilogue keeps every pass's tile of final values and a table of the tile's per-(image, slab, column) partials
// (conv_epilogue_w: keep_tab); wave k finalises (image, group) pair k, k + NW, ... with gn_group_stats -- the reduction of the stand-alone kernels over
// the same float4 partials, hence the same mean / rstd bits -- and all threads apply scale / shift (+ SiLU) with the arithmetic of gn_apply_vec to the
// values the statistics pass left in the tiles (exactly the values stored to y: bf16-rounded, or the fp32 ones of the f32 / f32x3 modes).  A first form re-read y and the statistics from L2 after waiting
// for the stores: 6-8 us of exposed round trips per launch (the whole chip is in the tail at the same time) -- as much as the launch it replaced.
// Geometry of the epilogue tiles: wave (wave_m, wave_n) holds rows [wave_m * EROWS, +EROWS) x its 16 WN columns, ECOLS = 16 NJ of them per pass.
// ------------------------------------------------------------------------------------------------
template <int TH, int TW, int WM, int WN, int NJ_, int WAVES_N>
struct GnTailGeom {
    static constexpr int NJ = NJ_ ? NJ_ : ((WN >= 2) ? 2 : 1);
    static constexpr int EROWS = 16 * WM, ECOLS = 16 * NJ, ESTR = ECOLS + 4, NPASS = (WN + NJ - 1) / NJ;
    static constexpr int SPT = (TH * TW) / conv_stat_rows(TH, TW, EROWS);        // statistics slabs per image
    static constexpr int HW = TH * TW;                                           // the tile's images are whole: one tile = HW pixels of each
    __host__ __device__ static constexpr int tiles_bytes(int nwaves) { return NPASS * nwaves * EROWS * ESTR * 4; }
    __host__ __device__ static constexpr int keep_bytes(int nimg, int bn) { return nimg * SPT * bn * 16; }
    __host__ __device__ static constexpr int total_bytes(int nwaves, int nimg, int bn) { return tiles_bytes(nwaves) + keep_bytes(nimg, bn) + 2 * nimg * bn * 4; }
};
template <typename T, int NTHREADS, class G, int WAVES_N, int WN, int BN, class AT>
__device__ __forceinline__ void gn_out_tail(const AT& a, int img0, int nimg, int n0, char* smem, const float4* keep_tab, float* tab, int tid) {
    // (every index below divides by compile-time constants only: with run-time divisors the address arithmetic of the apply loop -- four ~35-instruction
    // divisions per vector -- cost more than the arithmetic itself)
    constexpr int VEC = TI<T>::VEC, ES = 16 / VEC, NW = NTHREADS / 64, bn = BN, HW = G::HW, ncols = BN;      // host check: Cout % BN == 0, gw divides BN
    const int lane = tid & 63, wave = tid >> 6;
    const int gw = a.Cout >> 5;
    const int ng = ncols / gw;
    const int ni = min(nimg, a.B - img0);
    __syncthreads();                                        // the tiles and keep_tab are complete
    for (int p = wave; p < ni * ng; p += NW) {
        const int il = p / ng, gl = p - il * ng;
        float gam = 0.f, bet = 0.f;
        if (lane < gw) { gam = a.on_gamma[n0 + gl * gw + lane]; bet = a.on_beta[n0 + gl * gw + lane]; }
        float mean, rstd;
        // the tile's table as a [nimg][SPT][bn] statistics tensor of 32 gw "channels": local group gl of local image il
        gn_group_stats(keep_tab, G::SPT, bn, keep_tab, 1, 32 * gw, HW, a.on_eps, gl, il, lane, mean, rstd);
        if (lane < gw) gn_scale_shift(mean, rstd, gam, bet, tab[il * bn + gl * gw + lane], tab[(nimg + il) * bn + gl * gw + lane]);
    }
    __syncthreads();
    constexpr int cols = ncols / VEC;
    const int nv = ni * HW * cols;
    const __amdgpu_buffer_rsrc_t r_n = __builtin_amdgcn_make_buffer_rsrc(a.yn, 0, (int)(unsigned)((long long)a.B * HW * a.Cout * ES), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const float* tiles = (const float*)smem;
    for (int id = tid; id < nv; id += NTHREADS) {
        const int col = (id % cols) * VEC, m = id / cols;                            // m = image-local index x HW + pixel = row of the tile
        const int wm = m / G::EROWS, r = m - wm * G::EROWS;
        const int wn = col / (16 * WN), cw = col - wn * (16 * WN);
        const int ps = cw / G::ECOLS, c = cw - ps * G::ECOLS;
        const float* src = tiles + ((ps * NW + wm * WAVES_N + wn) * G::EROWS + r) * G::ESTR + c;
        float f[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) { const float4 v = *(const float4*)(src + e); f[e] = v.x; f[e + 1] = v.y; f[e + 2] = v.z; f[e + 3] = v.w; }
        const int il = m / HW;
        const uint4 o = gn_apply_f8<T>(f, &tab[il * bn + col], &tab[(nimg + il) * bn + col], a.on_silu);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_n, (int)(unsigned)((((long long)img0 * HW + m) * a.Cout + n0 + col) * ES), 0, WDM_STORE_AUX);
    }
}

// The same tail behind the bf16-tile epilogue (conv_epilogue_packed; 16 x 16 maps on 256 x 128 tiles: ONE image x 128 columns, one pass): wave (wm, wn) keeps its 64 rows x
// 64 columns as the swizzled bf16 tile of that epilogue -- row r = 128 bytes, 16-byte unit u in slot u ^ (r & 7), the unit's 8-byte halves swapped when r & 8 --,
// i.e. exactly the values stored to y; statistics table, group reduction and per-element arithmetic are those of gn_out_tail: the same bits.
template <typename T, int NTHREADS, int WAVES_N, int BN, class AT>
__device__ __forceinline__ void gn_out_tail_packed(const AT& a, int img0, int n0, char* smem, const float4* keep_tab, float* tab, int tid) {
    constexpr int NW = NTHREADS / 64, HW = 256, SPT = 4, bn = BN;
    const int lane = tid & 63, wave = tid >> 6;
    const int gw = a.Cout >> 5;
    const int ng = BN / gw;
    __syncthreads();                                        // the tiles and keep_tab are complete
    for (int p = wave; p < ng; p += NW) {
        float gam = 0.f, bet = 0.f;
        if (lane < gw) { gam = a.on_gamma[n0 + p * gw + lane]; bet = a.on_beta[n0 + p * gw + lane]; }
        float mean, rstd;
        gn_group_stats(keep_tab, SPT, bn, keep_tab, 1, 32 * gw, HW, a.on_eps, p, 0, lane, mean, rstd);
        if (lane < gw) gn_scale_shift(mean, rstd, gam, bet, tab[p * gw + lane], tab[bn + p * gw + lane]);
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r_n = __builtin_amdgcn_make_buffer_rsrc(a.yn, 0, (int)(unsigned)((long long)a.B * HW * a.Cout * 2), 0x00020000);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int cols = BN / 8;                            // 16-byte units per pixel
#pragma unroll
    for (int id = tid; id < HW * cols; id += NTHREADS) {
        const int cu = id % cols, m = id / cols;            // unit, pixel (= row of the workgroup's tile)
        const int wm = m >> 6, r = m & 63, wn = cu >> 3, u = cu & 7;
        uint4 v = *(const uint4*)(smem + (wm * WAVES_N + wn) * 8192 + r * 128 + ((u ^ (r & 7)) << 4));
        if (r & 8) v = uint4{v.z, v.w, v.x, v.y};
        float f[8];
        TI<T>::unpack(v, f);
        const int col = cu * 8;
        const uint4 o = gn_apply_f8<T>(f, &tab[col], &tab[bn + col], a.on_silu);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_n, (int)(unsigned)((((long long)img0 * HW + m) * a.Cout + n0 + col) * 2), 0, WDM_STORE_AUX);
    }
}

}  // namespace wdm
