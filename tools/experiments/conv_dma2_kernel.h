// EXPERIMENT (round 4; not in the library): measured at PARITY with the shipped kernels, see the table at the end of this comment.
//
// 3x3 stride-1 convolution, 256 x 128 output tile, in HALF the LDS (80 KB) on FOUR waves of 128 x 64 (<= 256 registers per lane), so that TWO workgroups share a CU:
// two waves per SIMD from DIFFERENT workgroups, whose phases are not tied together by a barrier -- whatever one workgroup cannot issue MFMAs through (its first
// DMA round trip, the GroupNorm+SiLU transform of a halo slab, a barrier, its whole epilogue) is matrix-pipe time for the other one.  conv_dma_kernel.h /
// conv_dmap_kernel.h hold all 160 KB with one eight-wave workgroup whose waves are all in the same phase at the same time: of a 64 x 64-map tile's ~44 k ticks
// 18.4 k are MFMA issue (tools/dmap_timeline.hip), the epilogue alone is 8 k.
//
// LDS map (bytes): A[2] = 2 x 21 KB halo slabs (dense 18 x 18 x 64 B: 21 DMA pieces) | weight ring = 4 x 8 KB TAP stages (128 rows x 64 B; a tap's weights are
// requested three taps ahead) | scale / shift table 2 x 768 floats: 80 KB.  Sub-stage = one tap (32 MFMAs per wave), taps walked dx-major so that the ten halo-row
// fragments of a dx column stay in registers for its three tap rows.  Same LDS images, fragment layouts, K order (slab, dx, dy) and epilogues as conv_dma_kernel.h
// => bit-identical outputs and statistics.
// DMA queue per wave (in order): per tap 2 weight pieces; at a slab's first tap the 5 or 6 halo pieces of the NEXT slab behind them.  The counted waits exclude a
// fixed number of YOUNGER pieces (conservatively 9 where the halo pieces are among them: wave 0 issues 6 of them, the others 5), so the uneven piece counts of
// older requests never matter.
//
// MEASURED (tools/conv_bench256.hip, batch 64, same process, bit-identical to the shipped tile on all 22 shapes incl. concat inputs, residuals and fused shortcuts):
//   64x64 128->128 gn   93.8 us vs 93.3 (persistent 256 x 128)      128->128 no prologue 82.9 vs 82.8      128->128 gn +1x1 (256 ch) 116.0 vs 118.8
//   64x64 256->128 gn  156.0 vs 151.1      384->128 gn 220.3 vs 205.6      96->128 68.8 vs 69.7      32x32 256->256 gn 81.8 vs 79.0 (68.2 on the 256 x 256 tile)
// A first, cruder form (single-buffered halo, dx-column sub-stages) measured the same.  Two desynchronised workgroups per CU hide every exposed phase of one
// workgroup behind the other's MFMAs and still take the same time as one workgroup whose eight waves move in lock step: the tile's ~44 k ticks are not exposed
// latency, they are issue time of the SIMDs (MFMA, the transform's VALU, fragment reads, DMA issue, epilogue -- two waves per SIMD either way).  What shortens a
// tile is fewer issued instructions per output (the packed epilogue: -7 %), not another arrangement of the same ones.
#pragma once
#include "conv_dma_kernel.h"

namespace wdm {

struct ConvDma2Cfg {
    static constexpr int TH = 16, TW = 16, WAVES_M = 2, WAVES_N = 2, WM = 8, WN = 4, NWAVES = 4, NTHREADS = 256, BN = 128, BK = 32;
    static constexpr int RS = 18, A_ROWS = 18 * 18, A_PIECES = 21;
    static constexpr int A_BYTES = A_PIECES * 1024;             // 21 KB
    static constexpr int B_TAP = BN * 64;                       // 8 KB
    static constexpr int B_OFF = 2 * A_BYTES;                   // 42 KB
    static constexpr int NRING = 4;
    static constexpr int SC_OFF = B_OFF + NRING * B_TAP;        // 74 KB
    static constexpr int MAX_CIN = 768;
    static constexpr int LDS_BYTES = SC_OFF + 2 * MAX_CIN * 4;  // 80 KB
    static constexpr int EPI_PACKED = NWAVES * EPI_PACK_TILE;   // 32 KB per half
    static constexpr int EPI_F32 = NWAVES * 64 * (64 + 4) * 4;  // 68 KB per half
    static constexpr int G_STAGE = (256 + 128) * 64;            // the fused 1x1 shortcut: 32-channel K steps, ring of three 24 KB stages
    static_assert(EPI_F32 <= LDS_BYTES && 3 * G_STAGE <= LDS_BYTES && LDS_BYTES <= 80 * 1024, "LDS");
};

template <bool PACKED>
__global__ __launch_bounds__(256, 2) void conv_dma2_kernel(const ConvArgs a) {
    using C = ConvDma2Cfg;
    using T = __bf16;
    constexpr int TW = C::TW, WM = C::WM, WN = C::WN, BN = C::BN, RS = C::RS;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / C::WAVES_N, wave_n = wave % C::WAVES_N;

    int mt, nt;
    if (!conv_decode_tile(a, (int)blockIdx.x, mt, nt)) return;
    const int n0 = nt * BN;
    int img0, tile_in_img, oy0, ox0;
    conv_decode_image<16, TW>(a, mt, img0, tile_in_img, oy0, ox0);
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;

    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto make_q = [](const void* p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)p;
        return i32x4{(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    };
    const i32x4 q_x0 = make_q(a.x0, a.x0_bytes), q_x1 = make_q(a.x1 ? a.x1 : a.x0, a.x1_bytes), q_w = make_q(a.w, a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const i32x4& rsrc, unsigned lds_addr, unsigned voff, int soff) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff)
                     : "memory");
    };
    constexpr unsigned OOB = 0xFFFF0000u;
    const int un = (lane & 3) ^ ((lane >> 3) & 2);                // channel unit this lane fetches (pieces start on multiples of 16 row slots: a function of the lane)
    // halo pieces of this wave: p = wave, wave + 4, ... < 21 (wave 0: six, the others five); per piece the source pixel of this lane's row slot
    constexpr int ACPM = 6;
    const int acp = wave == 0 ? 6 : 5;
    // (the source pixel of a piece's row slot is re-derived at every issue -- ~10 VALU per piece against 288 MFMAs per slab -- instead of kept in six registers: the
    // allocator spills them next to 128 accumulator registers, and a scratch reload in the K loop makes the compiler wait for the whole DMA queue)
    auto a_src = [&](int i, bool& ok) __attribute__((always_inline)) {
        int lq = lane >> 2;
        asm volatile("" : "+v"(lq));
        const int q = (wave + 4 * i) * 16 + lq;
        const int hy = (q * 3641) >> 16, hx = q - hy * RS;              // q / 18 for q < 336
        const int iy = iy0 + hy, ix = ix0 + hx;
        ok = i < acp && q < C::A_ROWS && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        return (unsigned)((img0 * a.Hin + iy) * a.Win + ix);
    };
    unsigned inb = 0;
#pragma unroll
    for (int i = 0; i < ACPM; ++i) { bool ok; (void)a_src(i, ok); if (ok) inb |= 1u << i; }
    // weight pieces: 16 rows of a tap stage each; piece p = wave * 2 + i covers rows [16 p, 16 p + 16): a scalar part + this lane's row (the launcher only picks the
    // kernel when rows n0 ... n0 + 127 exist)
    const unsigned b_lane = (unsigned)(((lane >> 2) * a.w_row_stride) * 2 + un * 16);
    const int nslab = a.Cin / C::BK;
    const int wslab = a.w_slab_stride ? a.w_slab_stride : C::BK;
    const int ntap = 9 * nslab;
    // tap index g = 9 s + 3 dx + dy; its weights [dy][dx] = tap row dy, column dx of slab s -> ring slot g & 3 (taps past the end: nothing)
    auto issue_b = [&](int g) __attribute__((always_inline)) {
        if (g >= ntap) return;
        const int s = g / 9, r = g - 9 * s, dx = r / 3, dy = r - 3 * dx;
        const long long soff0 = (long long)(dy * 3 + dx) * a.w_tap_stride + (long long)s * wslab;
        const unsigned base = lds0 + C::B_OFF + (g & 3) * C::B_TAP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wave * 2 + i;
            dma16(q_w, base + p * 1024, b_lane, (int)((soff0 + (long long)(n0 + p * 16) * a.w_row_stride) * 2));
        }
    };
    auto issue_a = [&](int s) __attribute__((always_inline)) {
        if (s >= nslab) return;
        const int c = s * C::BK;
        const bool first = c < a.C0;
        const unsigned xs2 = (unsigned)((first ? a.xs0 : a.xs1) * 2);
        const int so = (first ? c : c - a.C0) * 2;
        const unsigned base = lds0 + (s & 1) * C::A_BYTES;
#pragma unroll
        for (int i = 0; i < ACPM; ++i) {
            if (i >= acp) break;
            bool ok;
            const unsigned gp = a_src(i, ok);
            const unsigned vo = ok ? gp * xs2 + (unsigned)(un * 16) : OOB;
            if (first) dma16(q_x0, base + (wave + 4 * i) * 1024, vo, so); else dma16(q_x1, base + (wave + 4 * i) * 1024, vo, so);
        }
    };
    const float* sct = (const float*)(smem + C::SC_OFF);
    auto transform = [&](int s) __attribute__((always_inline)) {
        const int c = s * C::BK + un * 8;
        float sc[8], sh[8];
        *(float4*)&sc[0] = *(const float4*)(sct + c); *(float4*)&sc[4] = *(const float4*)(sct + c + 4);
        *(float4*)&sh[0] = *(const float4*)(sct + C::MAX_CIN + c); *(float4*)&sh[4] = *(const float4*)(sct + C::MAX_CIN + c + 4);
        char* base = smem + (s & 1) * C::A_BYTES + lane * 16;
#pragma unroll
        for (int i = 0; i < ACPM; ++i) {
            if (i >= acp) break;
            uint4* p = (uint4*)(base + (wave + 4 * i) * 1024);
            const uint4 tv = gn_silu_unit<T>(*p, sc, sh);
            if ((inb >> i) & 1u) *p = tv;
        }
    };

    const int ku = lane >> 4;
    // halo rows r and r + 4 are 72 slots apart: the same unit rotation, 4608 bytes further on.  The fragment addresses are formed when a dx column is read (a handful
    // of VALU per 96 MFMAs) instead of living in twelve registers through the whole loop: with 128 accumulator registers the allocator spills them otherwise
    constexpr int AR_STEP = 4 * RS * 64;
    const int q00 = (wave_m * WM) * RS + (lane & 15);
    auto a_at = [&](int r, int dx) __attribute__((always_inline)) {
        int qq = q00;
        asm volatile("" : "+v"(qq));                 // (keeps the compiler from hoisting the twelve results out of the K loop again)
        return lds_off(qq + (r & 3) * RS + dx, ku) + (r >> 2) * AR_STEP;
    };
    const int b_addr0 = C::B_OFF + lds_off(wave_n * WN * 16 + (lane & 15), ku);      // fragment column j: + j KB (16 rows on: the same unit rotation)

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define WDM_D2_SYNC(N) do { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: table rows, halo slab 0, the first three taps' weights (in that order: what is needed first is requested first)
    const bool pro = a.pro != 0;
    if (pro && wave < 3) {        // 768 floats = three 1 KB pieces per row
        const i32x4 q_sc = make_q(a.scale + (long long)img0 * a.Cin, (unsigned)(a.Cin * 4)), q_sh = make_q(a.shift + (long long)img0 * a.Cin, (unsigned)(a.Cin * 4));
        const unsigned vo = (unsigned)((wave * 256 + lane * 4) * 4);
        dma16(q_sc, lds0 + C::SC_OFF + wave * 1024, vo, 0);
        dma16(q_sh, lds0 + C::SC_OFF + C::MAX_CIN * 4 + wave * 1024, vo, 0);
    }
    issue_a(0);
    issue_b(0); issue_b(1); issue_b(2);
    if (pro) {
        WDM_D2_SYNC(6);                 // every wave's table pieces and this lane's halo pieces landed (younger: the three taps' six weight pieces)
        transform(0);
    }
    WDM_D2_SYNC(4);                     // tap 0's weights in (taps 1, 2 may be in flight), every lane's transform visible
    int g = 0;
    for (int s = 0; s < nslab; ++s) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            uint4 ah[WM + 2];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy, ++g) {
                // queue behind this point: the weights of tap g + 3 (slot of tap g - 1, which every wave left before the barrier it just passed), and at a slab's
                // first tap the next slab's halo (its buffer was last read at the previous slab's last tap)
                issue_b(g + 3);
                if (dx == 0 && dy == 0) issue_a(s + 1);
                const char* pa = smem + (s & 1) * C::A_BYTES;
                if (dy == 0) {
#pragma unroll
                    for (int r = 0; r < WM + 2; ++r) ah[r] = *(const uint4*)(pa + a_at(r, dx));
                }
                const char* pb = smem + (g & 3) * C::B_TAP;
                uint4 bfr[WN];
#pragma unroll
                for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(pb + b_addr0 + j * 1024);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16t<T>(acc[i][j], ah[i + dy], bfr[j]);
                if (dx == 2 && dy == 2) {
                    if (pro && s + 1 < nslab) {
                        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this lane's pieces of halo slab s + 1 (younger: at most three taps' weights)
                        __builtin_amdgcn_sched_barrier(0);
                        transform(s + 1);
                    }
                    WDM_D2_SYNC(4);     // weights of tap g + 1 in (younger: taps g + 2, g + 3); the transform visible
                } else if (dx == 0 && s + 1 < nslab) {
                    WDM_D2_SYNC(9);     // ... younger as well: the halo pieces requested at this slab's first tap (5 or 6 of them: 4 + 5, conservative for wave 0)
                } else {
                    WDM_D2_SYNC(4);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- second contraction into the same accumulators: the ResnetBlock's 1x1 shortcut over the block input (a.sx0 | a.sx1), 32 channels per K step:
    // stage = the tile's 256 pixel rows + 128 weight rows of 64 bytes in the conv kernels' rotated layout, ring of three, two steps of lead
    if (a.sx0 != nullptr) {
        const i32x4 q_s0 = make_q(a.sx0, a.sx0_bytes), q_s1 = make_q(a.sx1 ? a.sx1 : a.sx0, a.sx1_bytes), q_sw = make_q(a.sw, a.sw_bytes);
        unsigned g_px[4];               // four pixel pieces (16 rows each) per wave and step: pieces wave * 4 + i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 16 + (lane >> 2);
            g_px[i] = (unsigned)((img0 * a.Hout + oy0 + row / TW) * a.Wout + ox0 + row % TW);
        }
        const unsigned g_wl = (unsigned)(((lane >> 2) * a.sw_row_stride) * 2 + un * 16);
        auto issue2 = [&](int k, int buf) __attribute__((always_inline)) {
            const int c = k * 32;
            const bool first = c < a.sC0;
            const unsigned xs2 = (unsigned)((first ? a.sxs0 : a.sxs1) * 2);
            const int so = (first ? c : c - a.sC0) * 2;
            const unsigned base = lds0 + buf * C::G_STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned vo = g_px[i] * xs2 + (unsigned)(un * 16);
                if (first) dma16(q_s0, base + (wave * 4 + i) * 1024, vo, so); else dma16(q_s1, base + (wave * 4 + i) * 1024, vo, so);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = wave * 2 + i;
                dma16(q_sw, base + 256 * 64 + p * 1024, g_wl, (int)(((long long)(n0 + p * 16) * a.sw_row_stride + c) * 2));
            }
        };
        const int a2 = lds_off(wave_m * WM * 16 + (lane & 15), ku);                        // + i * 1024
        const int b2 = 256 * 64 + lds_off(wave_n * WN * 16 + (lane & 15), ku);             // + j * 1024
        const int nk = (a.sC0 + a.sC1) / 32;
        issue2(0, 0);
        if (nk > 1) issue2(1, 1);
        int buf = 0;
        for (int k = 0; k < nk; ++k) {
            if (k + 1 < nk) WDM_D2_SYNC(6); else WDM_D2_SYNC(0);
            if (k + 2 < nk) issue2(k + 2, buf >= 1 ? buf - 1 : 2);
            const char* base = smem + buf * C::G_STAGE;
            uint4 af[WM], bfr[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *(const uint4*)(base + a2 + i * 1024);
#pragma unroll
            for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(base + b2 + j * 1024);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) mma16t<T>(acc[i][j], af[i], bfr[j]);
            buf = buf + 1 == 3 ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
#undef WDM_D2_SYNC

    // a wave's 128 rows = wave rows 2 wave_m and 2 wave_m + 1 of the eight-wave kernels: two epilogues through the same (wave-private) LDS tile
#pragma unroll
    for (int p = 0; p < 2; ++p)
        conv_epilogue<T, 16, TW, 4, WN, WN, EpiNoHook, false, (PACKED ? 2 : 0)>(a, *(f32x4 (*)[4][WN])&acc[4 * p], smem, true, wave, lane, wave_m * 2 + p, wave_n, img0, oy0, ox0, n0, tile_in_img, 0,
                                                                                EpiNoHook(), false);
    gn_arrive<C::NTHREADS>(a, img0, 1, a.Hout * a.Wout, (int*)smem, tid);
}

}  // namespace wdm
