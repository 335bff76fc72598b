// 1x1 convolutions (plain GEMMs over pixels) with direct global->LDS staging -- bf16 only.
//
// A 1x1 conv has no tap reuse: every K step needs a fresh [pixels][64 ch] and [cout][64 ch] tile, ~12 KB per MFLOP, three
// times the traffic of a 3x3 stage.  Staging that through registers (conv_kernel.h) costs two instructions per KB (the
// load and the ds_write, ~100 + ~70 issue cycles each) and the kernel ends up bound by issuing them.  Here both operands go
// HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging registers, no ds_write pass, and the copy for
// K step k+1 is in flight while step k's MFMAs run (two LDS buffers, one workgroup barrier per K step).
//
// LDS image of one operand stage: rows (pixels / output channels) of 128 bytes = 64 channels = 8 16-byte units; unit u of
// row r sits in slot  u ^ ((r >> 1) & 7)  of its row.  An LDS-DMA instruction writes 1 KB lane-linearly (lane L -> byte
// 16 L of the chunk = row L/8, slot L%8), so the swizzle is applied to the SOURCE address: lane L fetches unit
// (L%8) ^ ((row>>1)&7) of its row -- still one full 128-byte line per 8 lanes.  A 16-lane ds_read_b128 group reads 16
// consecutive rows at one unit: two rows share a 256-byte bank sweep only if they have the same parity, and then their slots
// differ in (r>>1)&7 -> conflict-free.
//
// Tile: 16x16 pixels (one MFMA-row group = one image row) x 128 output channels, 8 waves as 4 (M) x 2 (N), wave tile 64 x 64:
// the accumulator layout of conv_kernel.h's main configuration, so conv_epilogue (bias, temb, residual, 16-byte stores,
// GroupNorm partial statistics, NCHW / fp32 output modes) is shared.  Out-of-range rows (pixels past the tensor, weight rows
// past the packed matrix) are outside the buffer descriptor: the DMA writes zeros.
#pragma once
#include "conv_kernel.h"
#ifndef WDM_GABL
#define WDM_GABL 0      // ablation mask for tools/gemm_ablate.hip: 1 no epilogue, 2 no MFMA, 4 DMA for the first two stages only
#endif

namespace wdm {

template <int TH, int TW, int NI, int WAVES_M, int WAVES_N, int WM, int WN>
struct GemmCfg {
    static constexpr int NWAVES = WAVES_M * WAVES_N;
    static constexpr int NTHREADS = 64 * NWAVES;
    static constexpr int M = TH * TW * NI;
    static constexpr int BN = 16 * WN * WAVES_N;
    static constexpr int BK = 64;                                   // channels per K step (two MFMA k-substeps)
    static constexpr int A_BYTES = M * 128;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int EPI_NJ = WN > 4 ? 4 : WN;                  // epilogue passes of 64 columns (whole 128-byte rows per wave)
    static constexpr int EPI_BYTES = NWAVES * 16 * WM * (16 * EPI_NJ + 4) * 4;
    static constexpr int NBUF = WN > 4 ? 2 : 3;                     // LDS ring: two stages in flight while one is consumed (256-column tiles: 64 KB stages, one in flight)
    static constexpr int LDS_BYTES = NBUF * STAGE > EPI_BYTES ? NBUF * STAGE : EPI_BYTES;
    static constexpr int A_CPW = (M / 8) / NWAVES;                  // 1 KB chunks (8 rows) per wave per stage
    static constexpr int B_CPW = (BN / 8) / NWAVES;
    static_assert(M == 16 * WM * WAVES_M && NWAVES == 8, "8 waves");
    static_assert((M / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0, "chunks must split evenly over the waves");
};

// the workgroup's work: block `bid` of the launch described by `a` (a kernel may hold two launches: tools/experiments/conv_gemm_pair_kernel.h)
template <int TH, int TW, int NI, int WAVES_M, int WAVES_N, int WM, int WN, typename T_ = __bf16>
__device__ __forceinline__ void conv_gemm_body(const ConvArgs& a, const int bid, char* smem) {
    using C = GemmCfg<TH, TW, NI, WAVES_M, WAVES_N, WM, WN>;
    using T = T_;
    h16_mode_init<T>();
    constexpr int BN = C::BN, A_BYTES = C::A_BYTES, STAGE = C::STAGE, A_CPW = C::A_CPW, B_CPW = C::B_CPW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    // workgroup -> (M tile, N tile): same XCD-aware order as conv_kernel
    int mt, nt;
    if (!conv_decode_tile(a, bid, mt, nt)) return;
    const int n0 = nt * BN;
    int img0, oy0, ox0, tile_in_img = 0;
    if (NI == 1) conv_decode_image<TH, TW>(a, mt, img0, tile_in_img, oy0, ox0);
    else { img0 = mt * NI; oy0 = 0; ox0 = 0; }

    constexpr unsigned OOB = 0xFFFF0000u;
    // ---- per-lane source offsets of this wave's chunks (loop-invariant; the K step is a scalar offset)
    unsigned a_v0[A_CPW], a_v1[A_CPW], b_v[B_CPW];
#pragma unroll
    for (int j = 0; j < A_CPW; ++j) {
        const int row = (wave * A_CPW + j) * 8 + (lane >> 3);
        const int u = (lane & 7) ^ ((row >> 1) & 7);
        const int img = row / (TH * TW), r = row % (TH * TW);
        const int img_g = img0 + img;
        const int iy = oy0 + r / TW, ix = ox0 + r % TW;
        const bool ok = img_g < a.B && iy < a.Hin && ix < a.Win;
        const unsigned gp = (unsigned)((conv_x_img(a, img_g) * a.Hin + iy) * a.Win + ix);
        a_v0[j] = ok ? gp * (unsigned)(a.xs0 * 2) + (unsigned)(u * 16) : OOB;
        a_v1[j] = ok ? gp * (unsigned)(a.xs1 * 2) + (unsigned)(u * 16) : OOB;
    }
#pragma unroll
    for (int j = 0; j < B_CPW; ++j) {
        const int row = (wave * B_CPW + j) * 8 + (lane >> 3);
        const int u = (lane & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_v[j] = n < a.w_rows ? (unsigned)(n * a.w_row_stride * 2 + u * 16) : OOB;
    }

    // The LDS-DMA is issued from inline asm: hipcc waits vmcnt(0) before the first ds_read after a DMA it knows about (it cannot
    // prove the read does not alias the destination), which would drain the ring every K step.  It does not count asm loads, so
    // every wait on them below is explicit.  M0 carries the wave-uniform LDS byte address; it is saved / restored inside the
    // statement because the compiler does not expect it to change.
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto make_q = [](const void* p, unsigned bytes) __attribute__((always_inline)) {      // raw buffer descriptor: base, stride 0, num_records, flags
        const unsigned long long v = (unsigned long long)p;
        return i32x4{(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    };
    const i32x4 q_x0 = make_q(a.x0, a.x0_bytes), q_x1 = make_q(a.x1 ? a.x1 : a.x0, a.x1_bytes);
    const i32x4 q_w = make_q((const T*)a.w + conv_w_img_offset(a, img0), a.w_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const i32x4& rsrc, unsigned lds_addr, unsigned voff, int soff) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff)
                     : "memory");
    };
    auto issue = [&](int k, int buf) __attribute__((always_inline)) {
        const int c = k * C::BK;
        const unsigned base = lds0 + buf * STAGE;
        if (c < a.C0) {
#pragma unroll
            for (int j = 0; j < A_CPW; ++j) dma16(q_x0, base + (wave * A_CPW + j) * 1024, a_v0[j], c * 2);
        } else {
#pragma unroll
            for (int j = 0; j < A_CPW; ++j) dma16(q_x1, base + (wave * A_CPW + j) * 1024, a_v1[j], (c - a.C0) * 2);
        }
#pragma unroll
        for (int j = 0; j < B_CPW; ++j) dma16(q_w, base + A_BYTES + (wave * B_CPW + j) * 1024, b_v[j], c * 2);
    };

    // ---- fragment addresses: row (16-aligned base + lane%16), k-unit ks*4 + lane/16, slot = unit ^ ((lane>>1)&7)
    const int sw = (lane >> 1) & 7;
    const int ku = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int slot = (ks * 4 + ku) ^ sw;
        a_off[ks] = (wave_m * WM * 16 + (lane & 15)) * 128 + slot * 16;
        b_off[ks] = A_BYTES + (wave_n * WN * 16 + (lane & 15)) * 128 + slot * 16;
    }
    static_assert(C::NBUF == 3 || C::NBUF == 2, "ring of three (two stages of lead) or two (one)");

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K loop: ring of NBUF stage buffers, the DMA runs two stages ahead.  These GEMMs stream their operands once (a 512 -> 512
    // 1x1 conv is 256 FLOP/B: HBM-bound), so what matters is bytes in flight per CU -- with one stage of prefetch every K step
    // pays a full memory round trip.  The waits are counted by hand: `vmcnt(CPW)` leaves only the newest stage's DMA
    // outstanding (i.e. stage k has landed for this wave), the raw barrier extends that to every wave's part and also says
    // everyone is done reading the buffer that stage k+2 is about to overwrite.  (__syncthreads() would drain the queue.)
    const int nk = a.Cin / C::BK;
    constexpr int CPW = A_CPW + B_CPW;
    issue(0, 0);
    if (C::NBUF == 3 && nk > 1) issue(1, 1);
    int buf = 0;
    for (int k = 0; k < nk; ++k) {
        if (C::NBUF == 3 && k + 1 < nk && !(WDM_GABL & 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (C::NBUF == 3) { if (k + 2 < nk && !(WDM_GABL & 4)) issue(k + 2, buf >= 1 ? buf - 1 : C::NBUF - 1); }      // (buf + 2) % 3
        else if (k + 1 < nk) issue(k + 1, buf ^ 1);                                                                // the buffer stage k - 1 was read from
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);      // see conv_dma_kernel.h
            uint4 af[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = *(const uint4*)(base + a_off[ks] + i * (16 * 128));
            constexpr int NB = WN < 4 ? WN : 4;          // weight fragments held at a time
            static_assert(WN % NB == 0, "column fragments in groups of four (or all of them)");
#pragma unroll
            for (int h = 0; h < WN / NB; ++h) {
                uint4 bfr[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) bfr[j] = *(const uint4*)(base + b_off[ks] + (h * NB + j) * (16 * 128));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if (WDM_GABL & 2) acc[i][h * NB + j][0] += __uint_as_float(af[i].x ^ bfr[j].y);
                        else mma16t<T>(acc[i][h * NB + j], af[i], bfr[j]);
                    }
            }
        }
        buf = buf + 1 == C::NBUF ? 0 : buf + 1;
    }
    __syncthreads();
    static_assert(C::EPI_BYTES <= C::LDS_BYTES, "epilogue tile does not fit in LDS");
    if (WDM_GABL & 1) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 123.456f) ((float*)a.y)[tid] = t;
        return;
    }
    // 8 x 8 maps (four images per tile): half-image statistics slabs need the two-fragment passes (conv_stat_rows)
    conv_epilogue<T, TH, TW, WM, WN, (TH * TW == 64 ? 2 : C::EPI_NJ), EpiNoHook, false, ((TH * TW) % 64 == 0 && TW == 16 && WM == 4 && WN % 4 == 0 && TH * TW != 64 ? 1 : 0)>(a, acc, smem, true, wave, lane, wave_m, wave_n, img0, oy0, ox0, n0, tile_in_img);
}

template <int TH, int TW, int NI, int WAVES_M, int WAVES_N, int WM, int WN, typename T_ = __bf16>
__global__ __launch_bounds__(512, 2) void conv_gemm_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv_gemm_body<TH, TW, NI, WAVES_M, WAVES_N, WM, WN, T_>(a, blockIdx.x, smem);
}

}  // namespace wdm
