// Producer / consumer ("wave-specialised") variant of the fused 3x3 stride-1 convolution.
//
// Why: in conv_kernel.h every wave alternates between a fill phase (global loads, GroupNorm+SiLU on the A operand,
// ds_write) and an MFMA phase, two workgroups per CU overlapping by chance.  The ablation tool (tools/conv_ablate.hip)
// shows that kernel is NOT matrix-bound: dropping the MFMAs only saves 20 % of its time, the LDS->MFMA loop alone
// tops out at ~1.5 PFLOP/s and the fill phases stretch that to ~0.85.  Here the two roles are split for good:
//
//   * workgroup = 12 waves, one per CU: waves 0-3 are CONSUMERS (LDS fragment reads + MFMAs, nothing else),
//     waves 4-11 are PRODUCERS (buffer loads two stages ahead -> GN-apply + SiLU in registers -> ds_write);
//     a consumer and a producer share each SIMD, so the matrix pipe and the VALU/LDS/VMEM pipes run concurrently
//     by construction instead of by luck;
//   * the LDS stage (A halo tile + 9 weight taps of one 32-channel slab, 64.5 KB) is double-buffered (129 KB of the
//     160 KB): producers fill stage s+1 while consumers read stage s; ONE workgroup barrier per stage;
//   * producers hold no accumulators, so they afford two register sets: the loads of stage s+3 are issued as soon
//     as the registers of stage s+1 have been written to LDS -> ~2 stage times (> 2 us) of latency cover.
//
// Everything else (LDS image, fragment mapping, loop-invariant buffer-load offsets, epilogue) is shared with
// conv_kernel.h, and the per-output accumulation order (slab-major, then tap) is the same as there.
#pragma once
#include <type_traits>

#include "conv_kernel.h"

namespace wdm {

template <typename T, int TH, int TW, int NI, int WAVES_M, int WAVES_N, int WM, int WN, int PMULT = 2>
struct ConvWsCfg {
    static constexpr int NCONS = WAVES_M * WAVES_N;             // consumer waves
    static constexpr int PTHREADS = 64 * NCONS * PMULT;         // producer threads (PMULT producer waves per consumer wave)
    static constexpr int NTHREADS = 64 * NCONS + PTHREADS;
    static constexpr int VEC = TI<T>::VEC;
    static constexpr int NU = 4;
    static constexpr int BK = NU * VEC;
    static constexpr int M = TH * TW * NI;
    static constexpr int BN = 16 * WN * WAVES_N;
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int RS = (PW + 7) / 8 * 8;
    static constexpr int NPIX = PH * PW;
    static constexpr int PLANE_IMG = PH * RS;
    static constexpr int PLANE = NI * PLANE_IMG;
    static constexpr int A_BYTES = PLANE * 64;
    static constexpr int B_BYTES = 9 * BN * 64;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int RPI = PTHREADS / NU;                   // rows covered per item index
    static constexpr int A_IPI = (NPIX * NU + PTHREADS - 1) / PTHREADS;
    static constexpr int B_IPT = (9 * BN * NU + PTHREADS - 1) / PTHREADS;
    static_assert(M == 16 * WM * WAVES_M, "tile M mismatch");
    static_assert(NCONS == 4, "4 consumer waves, one per SIMD");
    static_assert(LDS_BYTES <= 160 * 1024, "double-buffered stage exceeds the 160 KB LDS");
    static_assert(BN % RPI == 0 || RPI % BN == 0, "producer B mapping: weight rows per item and BN must nest");
};

template <typename T, int TH, int TW, int NI, int WAVES_M, int WAVES_N, int WM, int WN, int PMULT = 2>
__global__ __launch_bounds__((ConvWsCfg<T, TH, TW, NI, WAVES_M, WAVES_N, WM, WN, PMULT>::NTHREADS)) void conv_ws_kernel(const ConvArgs a) {
    using C = ConvWsCfg<T, TH, TW, NI, WAVES_M, WAVES_N, WM, WN, PMULT>;
    constexpr int VEC = C::VEC, NU = C::NU, BK = C::BK, BN = C::BN, PW = C::PW, RS = C::RS, NPIX = C::NPIX;
    constexpr int PLANE_IMG = C::PLANE_IMG, A_BYTES = C::A_BYTES, STAGE = C::STAGE_BYTES;
    constexpr int A_IPI = C::A_IPI, B_IPT = C::B_IPT, RPI = C::RPI;
    constexpr int ES = (int)sizeof(T);

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_prod = wave >= C::NCONS;
    const int cw = is_prod ? wave - C::NCONS : wave;             // index inside the role
    const int wave_m = cw / WAVES_N, wave_n = cw % WAVES_N;

    const int bid = blockIdx.x;
    int mt, nt;
    {   // same XCD-aware tile order as conv_kernel.h
        const int gn = a.grid_gn, gm = 8 / gn;
        const int xcd = bid & 7, seq = bid >> 3;
        const int xn = xcd % gn, xm = xcd / gn;
        const int ncnt = (a.ntiles - xn + gn - 1) / gn, mcnt = (a.mtiles - xm + gm - 1) / gm;
        if (ncnt <= 0 || mcnt <= 0 || seq >= mcnt * ncnt) return;
        if (gn == 1) { nt = seq % ncnt; mt = xm + gm * (seq / ncnt); }
        else { mt = xm + gm * (seq % mcnt); nt = xn + gn * (seq / mcnt); }
    }
    const int n0 = nt * BN;
    int img0, oy0, ox0, tile_in_img = 0;
    if (NI == 1) {
        const int twn = a.Wout / TW;
        const int tpi = (a.Hout / TH) * twn;
        img0 = mt / tpi;
        const int t = mt - img0 * tpi;
        tile_in_img = t;
        oy0 = (t / twn) * TH;
        ox0 = (t % twn) * TW;
    } else {
        img0 = mt * NI; oy0 = 0; ox0 = 0;
    }
    const int nst = a.Cin / BK;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (is_prod) {
        // =============================================================== PRODUCER waves
        const int ptid = tid - 64 * C::NCONS;
        const int unit = ptid & (NU - 1);
        constexpr unsigned OOB = 0xFFFF0000u;
        const __amdgpu_buffer_rsrc_t r_x0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x0, 0, a.x0_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x1 ? a.x1 : a.x0), 0, a.x1_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.w_bytes, 0x00020000);
        unsigned a_v0[NI][A_IPI], a_v1[NI][A_IPI];
        int a_l[NI][A_IPI];
        unsigned inb_mask = 0;
        static_assert(NI * A_IPI <= 32, "inb_mask too small");
#pragma unroll
        for (int im = 0; im < NI; ++im) {
            const int img_g = img0 + im;
#pragma unroll
            for (int i = 0; i < A_IPI; ++i) {
                const int q = (ptid >> 2) + i * RPI;
                const int hy = q / PW, hx = q - hy * PW;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                const bool ok = (q < NPIX) && (img_g < a.B) && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
                const unsigned gp = (unsigned)((img_g * a.Hin + iy) * a.Win + ix);
                a_v0[im][i] = ok ? gp * (unsigned)(a.xs0 * ES) + (unsigned)(unit * 16) : OOB;
                a_v1[im][i] = ok ? gp * (unsigned)(a.xs1 * ES) + (unsigned)(unit * 16) : OOB;
                a_l[im][i] = lds_off(im * PLANE_IMG + hy * RS + hx, unit);
                if (ok) inb_mask |= 1u << (im * A_IPI + i);
            }
        }
        const int rb0 = ptid >> 2;
        const int b_l0 = A_BYTES + lds_off(rb0, unit);
        const unsigned b_v0 = (unsigned)((((long long)(n0 + rb0 % BN)) * a.w_row_stride + (rb0 / BN) * a.w_tap_stride) * ES + unit * 16);

        uint4 ra[2][NI][A_IPI];
        uint4 rb[2][B_IPT];
        float sc[2][NI][VEC], sh[2][NI][VEC];

        auto load16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) __attribute__((always_inline)) -> uint4 {
            return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
        };
        auto load_stage = [&](auto set_c, int st) __attribute__((always_inline)) {
            constexpr int S = decltype(set_c)::value;
            const int c = st * BK;
#pragma unroll
            for (int im = 0; im < NI; ++im) {
#pragma unroll
                for (int i = 0; i < A_IPI; ++i) {
                    if (c < a.C0) ra[S][im][i] = load16(r_x0, a_v0[im][i], c * ES);
                    else ra[S][im][i] = load16(r_x1, a_v1[im][i], (c - a.C0) * ES);
                }
                if (a.pro) {
                    const int ig = img0 + im < a.B ? img0 + im : a.B - 1;
                    const float* ps = a.scale + (long long)ig * a.Cin + c + unit * VEC;
                    const float* pf = a.shift + (long long)ig * a.Cin + c + unit * VEC;
#pragma unroll
                    for (int e = 0; e < VEC; e += 4) {
                        const float4 s4 = *(const float4*)(ps + e), f4 = *(const float4*)(pf + e);
                        sc[S][im][e] = s4.x; sc[S][im][e + 1] = s4.y; sc[S][im][e + 2] = s4.z; sc[S][im][e + 3] = s4.w;
                        sh[S][im][e] = f4.x; sh[S][im][e + 1] = f4.y; sh[S][im][e + 2] = f4.z; sh[S][im][e + 3] = f4.w;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < B_IPT; ++i) {
                const int tap = (i * RPI) / BN, nof = (i * RPI) % BN;
                const long long so = (long long)nof * a.w_row_stride + tap * a.w_tap_stride + c;
                rb[S][i] = load16(r_w, b_v0, (int)(so * ES));
            }
        };
        auto fill_stage = [&](auto set_c, int buf) __attribute__((always_inline)) {
            constexpr int S = decltype(set_c)::value;
            char* base = smem + buf * STAGE;
#pragma unroll
            for (int im = 0; im < NI; ++im) {
#pragma unroll
                for (int i = 0; i < A_IPI; ++i) {
                    uint4 v = ra[S][im][i];
                    if (a.pro) {     // wave-uniform; out-of-image pixels stay zero (padding comes AFTER the activation)
                        const uint4 tv = gn_silu_unit<T>(v, sc[S][im], sh[S][im]);
                        const bool in = (inb_mask >> (im * A_IPI + i)) & 1u;
                        v = make_uint4(in ? tv.x : 0u, in ? tv.y : 0u, in ? tv.z : 0u, in ? tv.w : 0u);
                    }
                    const bool may_overrun = (i + 1) * RPI > NPIX;
                    if (!may_overrun || (ptid >> 2) + i * RPI < NPIX) *(uint4*)(base + a_l[im][i]) = v;
                }
            }
#pragma unroll
            for (int i = 0; i < B_IPT; ++i) {
                const bool may_overrun = (i + 1) * RPI > 9 * BN;
                if (!may_overrun || rb0 + i * RPI < 9 * BN) *(uint4*)(base + b_l0 + i * (RPI * 64)) = rb[S][i];
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;

        load_stage(S0{}, 0);
        if (nst > 1) load_stage(S1{}, 1);
        fill_stage(S0{}, 0);
        if (nst > 2) load_stage(S0{}, 2);
        __syncthreads();                                           // stage 0 is in buffer 0
        for (int st = 0; st < nst; st += 2) {
            if (st + 1 < nst) {                                    // while the consumers run stage st: fill stage st+1
                fill_stage(S1{}, 1);
                if (st + 3 < nst) load_stage(S1{}, st + 3);
            }
            __syncthreads();
            if (st + 1 < nst) {
                if (st + 2 < nst) {                                // while the consumers run stage st+1: fill stage st+2
                    fill_stage(S0{}, 0);
                    if (st + 4 < nst) load_stage(S0{}, st + 4);
                }
                __syncthreads();
            }
        }
    } else {
        // =============================================================== CONSUMER waves
        const int ku = lane >> 4;
        int a_addr[WM][3];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m = (wave_m * WM + i) * 16 + (lane & 15);
            const int img = m / (TH * TW), r = m % (TH * TW);
            const int q0 = img * PLANE_IMG + (r / TW) * RS + (r % TW);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) a_addr[i][dx] = lds_off(q0 + dx, ku);
        }
        int b_addr[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) b_addr[j] = A_BYTES + lds_off((wave_n * WN + j) * 16 + (lane & 15), ku);

        __syncthreads();                                           // stage 0 is in buffer 0
        for (int st = 0; st < nst; ++st) {
            const char* base = smem + (st & 1) * STAGE;
            if (TW == 16) {
                // halo row i+dy serves output row i at tap dy: WM+2 fragment reads per dx instead of 3*WM
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    uint4 ah[WM + 2];
#pragma unroll
                    for (int r = 0; r < WM + 2; ++r) ah[r] = *(const uint4*)(base + a_addr[0][dx] + r * (RS * 64));
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        uint4 bfr[WN];
#pragma unroll
                        for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(base + b_addr[j] + (dy * 3 + dx) * (BN * 64));
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < WN; ++j) mma16<T>(acc[i][j], ah[i + dy], bfr[j]);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    const int dy = s / 3, dx = s % 3;
                    uint4 af[WM], bfr[WN];
#pragma unroll
                    for (int i = 0; i < WM; ++i) af[i] = *(const uint4*)(base + a_addr[i][dx] + dy * (RS * 64));
#pragma unroll
                    for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(base + b_addr[j] + s * (BN * 64));
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) mma16<T>(acc[i][j], af[i], bfr[j]);
                }
            }
            __syncthreads();
        }
    }

    static_assert(4 * 16 * WM * (16 * (WN >= 2 ? 2 : 1) + 4) * 4 <= C::LDS_BYTES, "epilogue tile does not fit in LDS");
    conv_epilogue<T, TH, TW, WM, WN>(a, acc, smem, !is_prod, cw, lane, wave_m, wave_n, img0, oy0, ox0, n0, tile_in_img);
}

}  // namespace wdm
