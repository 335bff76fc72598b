// 3x3 stride-1 convolution on 8 x 8 maps, 16-bit operands by LDS-DMA, no prologue -- the K-SPLIT form of conv_dma8_kernel.h (round 5, VERDICT r4 item 5:
// "the 128 x 96 tile on eight waves").
//
// Why: conv_dma8_kernel.h is LDS-bound, not matrix-bound -- its 32 x 48 wave tiles read 0.78 fragments per MFMA (14 ds_read_b128 for 18 MFMAs per sub-stage),
// and with two 128 x 48 workgroups per CU every CU pulls 92 KB through the LDS-DMA path per 32-channel slab (MFMAs + fragment reads alone 32 us, DMA alone 30 us of
// a 768 -> 768 launch at batch 64).  M = B x 64 rows is all the 8 x 8 level has: at batch 64 a 128 x 96 tile per CU is exactly one round of 256 workgroups,
// so a larger WAVE tile has to come out of the same workgroup tile.  Here the eight waves are two K groups of four: group g owns the sub-stages (slab, dx) of parity
// g in the order t = 3 slab + dx, each of its waves holds a 64 x 48 tile (one image x half the columns, 48 accumulator registers):
//   fragment reads per MFMA   0.78 -> 0.50   (9 halo-row + 9 weight fragments for 36 MFMAs per sub-stage)
//   LDS-DMA bytes per CU/slab 92 KB -> 74 KB (one halo of 20 KB instead of two, the same 54 KB of weights)
//   barriers per MFMA         halved         (a step = two sub-stages, 36 MFMAs per wave)
// After the K loop (and the fused 1x1 shortcut, split the same way: k slice g of every 64-channel step) the two groups swap halves through LDS -- group g keeps rows
// [32 g, 32 g + 32) of its image and adds the other group's partial sums for them -- and all eight waves run the epilogue of the 32 x 48 wave tiles of
// conv_dma8_kernel.h (same statistics slabs, same in-tile GroupNorm of the output).
// Sum order per output: (even sub-stages in ascending order) + (odd sub-stages in ascending order), fp32 -- fixed, so runs are bit-reproducible and batch-independent,
// but NOT the single ascending chain of conv_dma8_kernel.h: the two kernels agree to fp32 rounding, not bit for bit.
//
// Schedule: steps j = 0 .. 3 nslab / 2 - 1, step j = sub-stages 2 j (group 0) and 2 j + 1 (group 1); three steps = two slabs = one period of the loop.
//   LDS   two halo buffers of 20 KB (2 images x 10 x 10 pixels, 16-slot rows: conv_dma8_kernel.h's conflict-free image) + a ring of SIX 18 KB weight
//         sub-stages (3 taps x 96 rows x 64 B) = 148 KB: one workgroup per CU
//   DMA   waves 0..5 fetch weights (3 pieces of 1 KB per sub-stage each, two steps = four sub-stages ahead: vmcnt(6) before a step's barrier), waves 6, 7 fetch
//         the halo (10 pieces per slab each, one step ahead of its first use: vmcnt(0))
//   one raw s_barrier per step.
#pragma once
#include "conv_kernel.h"
#include "gn_group.h"

#ifndef WDM_D8KABL
#define WDM_D8KABL 0        // tools/dma8k_bench.hip: 2 = no MFMAs, 16 (with 2) = no fragment reads either, 4 = no halo DMA, 8 = no weight DMA (timing only)
#endif

namespace wdm {

struct ConvDma8kCfg {
    static constexpr int TH = 8, TW = 8, NI = 2, BN = 96, BK = 32;
    static constexpr int NWAVES = 8, NTHREADS = 512;
    static constexpr int WM = 4, WN = 3;                        // K loop: 64 x 48 per wave
    static constexpr int WM_E = 2, WAVES_N = 2, NJ = 1;         // epilogue: 32 x 48 per wave, waves 4 (M) x 2 (N)
    static constexpr int PH = 10, PW = 10, RS = 16;
    static constexpr int PLANE_IMG = PH * RS, A_ROWS = NI * PLANE_IMG;
    static constexpr int A_BYTES = A_ROWS * 64;                 // 20 KB = 20 pieces
    static constexpr int B_ROWS = 3 * BN, B_SUB = B_ROWS * 64;  // 18 KB = 18 pieces
    static constexpr int NRING = 6, W_WAVES = 6, B_CPW = 3, H_WAVES = 2, A_CPW = 10;
    static constexpr int B_OFF = 2 * A_BYTES;
    static constexpr int LDS_BYTES = B_OFF + NRING * B_SUB;     // 148 KB
    static constexpr int X_OFF = 96 * 1024, X_WAVE = 2 * WN * 64 * 16;      // the half-tile exchange: 6 KB per wave behind everything the epilogue overlays
    static constexpr int G_A = NI * 64 * 128, G_STAGE = G_A + 128 * 128;    // shortcut GEMM stage: 128 pixel rows + 96 (of 128) weight rows x 64 channels
    static_assert(W_WAVES * B_CPW * 16 == B_ROWS && H_WAVES * A_CPW * 16 == A_ROWS && W_WAVES + H_WAVES == NWAVES, "DMA roles");
    static_assert(LDS_BYTES <= 160 * 1024 && 3 * G_STAGE <= LDS_BYTES && X_OFF + NWAVES * X_WAVE <= LDS_BYTES, "LDS");
};

template <typename T_ = __bf16>
__global__ __launch_bounds__(512, 1) void conv_dma8k_kernel(const ConvArgs a) {
    using C = ConvDma8kCfg;
    using T = T_;
    constexpr int TH = C::TH, TW = C::TW, NI = C::NI, WM = C::WM, WN = C::WN, BN = C::BN, RS = C::RS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16_mode_init<T>();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const bool is_w = wave < C::W_WAVES;

    int mt, nt;
    if (!conv_decode_tile(a, blockIdx.x, mt, nt)) return;
    const int n0 = nt * BN;
    const int img0 = mt * NI;

    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto make_q = [](const void* p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)p;
        return i32x4{(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    };
    const i32x4 q_x0 = make_q(a.x0, a.x0_bytes), q_w = make_q(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const i32x4& rsrc, unsigned lds_addr, unsigned voff, int soff) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff)
                     : "memory");
    };

    constexpr unsigned OOB = 0xFFFF0000u;
    const int un = (lane & 3) ^ ((lane >> 3) & 2);          // channel unit this lane fetches (conv_dma_kernel.h)
    // source offsets of this wave's DMA pieces: weight waves use dv[0..2] (rows [dy][n] of a sub-stage), halo waves dv[0..9] (row slots of the halo image)
    unsigned dv[C::A_CPW];
    if (is_w) {
#pragma unroll
        for (int i = 0; i < C::B_CPW; ++i) {
            const int r = (wave * C::B_CPW + i) * 16 + (lane >> 2);
            const int dy = r / BN, n = n0 + (r - dy * BN);
            dv[i] = n < a.w_rows ? (unsigned)(((long long)dy * 3 * a.w_tap_stride + (long long)n * a.w_row_stride) * 2 + un * 16) : OOB;
        }
#pragma unroll
        for (int i = C::B_CPW; i < C::A_CPW; ++i) dv[i] = OOB;
    } else {
#pragma unroll
        for (int i = 0; i < C::A_CPW; ++i) {
            const int q = ((wave - C::W_WAVES) * C::A_CPW + i) * 16 + (lane >> 2);
            const int im = q / C::PLANE_IMG, qi = q - im * C::PLANE_IMG;
            const int hy = qi / RS, hx = qi - hy * RS;
            const int iy = hy - 1, ix = hx - 1;
            const bool ok = hx < C::PW && img0 + im < a.B && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const unsigned gp = (unsigned)(((img0 + im) * a.Hin + iy) * a.Win + ix);
            dv[i] = ok ? gp * (unsigned)(a.xs0 * 2) + (unsigned)(un * 16) : OOB;
        }
    }
    const int nslab = a.Cin / C::BK;
    const int wslab = a.w_slab_stride ? a.w_slab_stride : C::BK;
    auto issue_w = [&](int s, int dx, int slot) __attribute__((always_inline)) {       // weight waves only
        const int sc_ = s < nslab ? s : nslab - 1;          // clamped: uniform DMA counts, the extra pieces land in slots nobody reads again
        const int soff = (int)(((long long)dx * a.w_tap_stride + (long long)sc_ * wslab) * 2);
        const unsigned base = lds0 + C::B_OFF + slot * C::B_SUB + wave * (C::B_CPW * 1024);
#pragma unroll
        for (int i = 0; i < C::B_CPW; ++i) if (!(WDM_D8KABL & 8)) dma16(q_w, base + i * 1024, dv[i], soff);
    };
    auto issue_h = [&](int s, int buf) __attribute__((always_inline)) {                // halo waves only
        const int sc_ = s < nslab ? s : nslab - 1;
        const unsigned base = lds0 + buf * C::A_BYTES + (wave - C::W_WAVES) * (C::A_CPW * 1024);
#pragma unroll
        for (int i = 0; i < C::A_CPW; ++i) if (!(WDM_D8KABL & 4)) dma16(q_x0, base + i * 1024, dv[i], sc_ * C::BK * 2);
    };

    // fragment addresses per step of the period: group 0 walks (slab 2m, dx 0), (2m, dx 2), (2m + 1, dx 1); group 1 (2m, dx 1), (2m + 1, dx 0), (2m + 1, dx 2)
    const int ku = lane >> 4;
    int a_ad[3][WM], b_base[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int dx = kg ? (p == 0 ? 1 : p == 1 ? 0 : 2) : (p == 0 ? 0 : p == 1 ? 2 : 1);
        const int hb = kg ? (p == 0 ? 0 : 1) : (p == 2 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int ly = i * 2 + ((lane & 15) >> 3), lx = lane & 7;
            a_ad[p][i] = hb * C::A_BYTES + lds_off(wm * C::PLANE_IMG + ly * RS + lx + dx, ku);
        }
        b_base[p] = C::B_OFF + (2 * p + kg) * C::B_SUB;
    }
    int b_addr[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) b_addr[j] = lds_off((wn * WN + j) * 16 + (lane & 15), ku);

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one sub-stage: a 16-row MFMA group is two image rows; tap row dy = 2 of group i is tap row 0 of group i + 1 (conv_dma8_kernel.h): five "even" and four "odd"
    // row-pair fragments serve the three tap rows of the image's four groups
    auto compute = [&](const int (&aad)[WM], int bb) __attribute__((always_inline)) {
        if ((WDM_D8KABL & 18) == 18) return;
        uint4 ae[WM + 1], ao[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) { ae[i] = *(const uint4*)(smem + aad[i]); ao[i] = *(const uint4*)(smem + aad[i] + RS * 64); }
        ae[WM] = *(const uint4*)(smem + aad[WM - 1] + 2 * (RS * 64));
        const char* pb = smem + bb;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            if (dy == 0) __builtin_amdgcn_s_setprio(2); else if (dy == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            uint4 bfr[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(pb + b_addr[j] + dy * (BN * 64));
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const uint4& af = dy == 0 ? ae[i] : dy == 1 ? ao[i] : ae[i + 1];
                    if (WDM_D8KABL & 2) { acc[i][j][0] += __uint_as_float(af.x ^ bfr[j].x); } else mma16t<T>(acc[i][j], af, bfr[j]);
                }
        }
    };
#define WDM_D8K_SYNC() do { if (is_w) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                            asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    if (is_w) { issue_w(0, 0, 0); issue_w(0, 1, 1); issue_w(0, 2, 2); issue_w(1, 0, 3); }
    else issue_h(0, 0);
    for (int m = 0; 2 * m < nslab; ++m) {
        const int s0 = 2 * m;
        WDM_D8K_SYNC();                    // sub-stages (s0, 0), (s0, 1) and the halo of slab s0 have landed
        if (is_w) { issue_w(s0 + 1, 1, 4); issue_w(s0 + 1, 2, 5); }
        else issue_h(s0 + 1, 1);
        compute(a_ad[0], b_base[0]);
        WDM_D8K_SYNC();                    // (s0, 2), (s0 + 1, 0), halo s0 + 1
        if (is_w) { issue_w(s0 + 2, 0, 0); issue_w(s0 + 2, 1, 1); }
        compute(a_ad[1], b_base[1]);
        WDM_D8K_SYNC();                    // (s0 + 1, 1), (s0 + 1, 2); every wave is done with slab s0's halo
        if (is_w) { issue_w(s0 + 2, 2, 2); issue_w(s0 + 3, 0, 3); }
        else issue_h(s0 + 2, 0);
        compute(a_ad[2], b_base[2]);
    }
#undef WDM_D8K_SYNC
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");            // no DMA may land on what follows
    __builtin_amdgcn_sched_barrier(0);

    // ---- second contraction into the same accumulators: the ResnetBlock's 1x1 shortcut over the block input (a.sx0 | a.sx1): a GEMM over the tile's 128 pixels,
    // 64 channels per K step through three 32 KB stages; K group g takes the 32-channel slice g of every step
    if (a.sx0 != nullptr) {
        constexpr int G_A = C::G_A, G_STAGE = C::G_STAGE;
        const i32x4 q_s0 = make_q(a.sx0, a.sx0_bytes), q_s1 = make_q(a.sx1 ? a.sx1 : a.sx0, a.sx1_bytes), q_sw = make_q(a.sw, a.sw_bytes);
        unsigned g_a0[2], g_a1[2], g_b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + (lane >> 3);          // 0 .. 127: image row / 64, pixel row % 64
            const int u = (lane & 7) ^ ((row >> 1) & 7);
            const bool ok = img0 + row / 64 < a.B;
            const unsigned gp = (unsigned)((img0 + row / 64) * 64 + row % 64);
            g_a0[i] = ok ? gp * (unsigned)(a.sxs0 * 2) + (unsigned)(u * 16) : OOB;
            g_a1[i] = ok ? gp * (unsigned)(a.sxs1 * 2) + (unsigned)(u * 16) : OOB;
            const int n = n0 + row;
            g_b[i] = (row < BN && n < a.sw_rows) ? (unsigned)(n * a.sw_row_stride * 2 + u * 16) : OOB;
        }
        auto issue2 = [&](int k, int buf) __attribute__((always_inline)) {
            const int c = k * 64;
            const unsigned base = lds0 + buf * G_STAGE;
            if (c < a.sC0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) dma16(q_s0, base + (wave * 2 + i) * 1024, g_a0[i], c * 2);
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) dma16(q_s1, base + (wave * 2 + i) * 1024, g_a1[i], (c - a.sC0) * 2);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) dma16(q_sw, base + G_A + (wave * 2 + i) * 1024, g_b[i], c * 2);
        };
        const int sw7 = (lane >> 1) & 7;
        const int slot = (kg * 4 + ku) ^ sw7;
        const int a2 = (wm * 64 + (lane & 15)) * 128 + slot * 16;
        const int b2 = G_A + (wn * WN * 16 + (lane & 15)) * 128 + slot * 16;
        const int nk = (a.sC0 + a.sC1) / 64;
        issue2(0, 0);
        if (nk > 1) issue2(1, 1);
        int buf = 0;
        for (int k = 0; k < nk; ++k) {
            if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (k + 2 < nk) issue2(k + 2, buf >= 1 ? buf - 1 : 2);
            const char* base = smem + buf * G_STAGE;
            uint4 af[WM], bfr[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *(const uint4*)(base + a2 + i * (16 * 128));
#pragma unroll
            for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(base + b2 + j * (16 * 128));
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) mma16t<T>(acc[i][j], af[i], bfr[j]);
            buf = buf == 2 ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- the two K groups swap halves: group g keeps rows [32 g, 32 g + 32) of its image (fragments 2 g, 2 g + 1) and adds the partner's sums for them
    {
        float4* xw = (float4*)(smem + C::X_OFF + wave * C::X_WAVE);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const f32x4 g = kg ? acc[ii][j] : acc[2 + ii][j];          // what this wave gives away
                xw[(ii * WN + j) * 64 + lane] = make_float4(g[0], g[1], g[2], g[3]);
            }
    }
    __syncthreads();
    f32x4 acc_e[C::WM_E][WN];
    {
        const float4* xr = (const float4*)(smem + C::X_OFF + (wave ^ 4) * C::X_WAVE);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const f32x4 k = kg ? acc[2 + ii][j] : acc[ii][j];
                const float4 o = xr[(ii * WN + j) * 64 + lane];
                // (even sub-stages) + (odd sub-stages), whichever group holds which: one rounding per component, the same in both orders
                const f32x4 ev = kg ? f32x4{o.x, o.y, o.z, o.w} : k, od = kg ? k : f32x4{o.x, o.y, o.z, o.w};
                acc_e[ii][j] = f32x4{ev[0] + od[0], ev[1] + od[1], ev[2] + od[2], ev[3] + od[3]};
            }
    }
    const int wave_m_e = wm * 2 + kg, ewave = wave_m_e * C::WAVES_N + wn;
    // the tile is NI whole images x BN columns: the consumer's act(GroupNorm(y)) from here when it asked for it (gn_group.h; the epilogue keeps its tiles)
    using G = GnTailGeom<TH, TW, C::WM_E, WN, C::NJ, C::WAVES_N>;
    static_assert(G::total_bytes(C::NWAVES, NI, BN) <= C::X_OFF, "in-tile GroupNorm: LDS");
    float4* keep_tab = a.yn != nullptr ? (float4*)(smem + G::tiles_bytes(C::NWAVES)) : nullptr;
    conv_epilogue<T, TH, TW, C::WM_E, WN, C::NJ>(a, acc_e, smem, true, ewave, lane, wave_m_e, wn, img0, 0, 0, n0, 0, 0, EpiNoHook(), true, keep_tab, BN);
    if (a.yn != nullptr) gn_out_tail<T, C::NTHREADS, G, C::WAVES_N, WN, BN>(a, img0, NI, n0, smem, keep_tab, (float*)(smem + G::tiles_bytes(C::NWAVES) + G::keep_bytes(NI, BN)), tid);
}

}  // namespace wdm
