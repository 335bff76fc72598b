// 1x1 convolutions / batched GEMMs of the "f32x3" mode with direct global->LDS staging: conv_gemm_kernel.h (256 pixels x 128 columns on 8 waves, both operands
// streamed by LDS-DMA through a ring of three stages, two in flight) with the operand handling of conv_dmax3_kernel.h.
//
// A stage is 32 fp32 channels: rows of 128 bytes = 8 16-byte units = TWO 16-channel groups; unit u of row r sits in slot u ^ ((r >> 1) & 7) as in the bf16
// kernel.  Both operands are activations or per-image "weights" as often as model weights here (Q.K^T, P.V), so both are split in LDS, by the lane that
// fetched the unit, behind the MFMAs of the previous stage: group g of a row becomes [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15] (logical slots 4g .. 4g + 3;
// the eight lanes of a row read their units with one instruction and write their 8-byte halves with the next).  A product is two v_mfma_f32_16x16x32_bf16:
// the weight-side fragment is the group as stored ([w_hi | w_lo] x 16 channels), the pixel-side fragment its hi half in all four k-groups, then its lo half.
#pragma once
#include "conv_kernel.h"

namespace wdm {

struct GemmX3Cfg {
    static constexpr int TH = 16, TW = 16, WAVES_M = 4, WAVES_N = 2, WM = 4, WN = 4;
    static constexpr int NWAVES = 8, NTHREADS = 512, M = 256, BN = 128, BK = 32;
    static constexpr int A_BYTES = M * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;      // 48 KB
    static constexpr int NBUF = 3;
    static constexpr int EPI_BYTES = NWAVES * 64 * (16 * WN + 4) * 4;
    static constexpr int LDS_BYTES = NBUF * STAGE > EPI_BYTES ? NBUF * STAGE : EPI_BYTES;      // 144 KB
    static constexpr int A_CPW = (M / 8) / NWAVES, B_CPW = (BN / 8) / NWAVES;                  // 1 KB chunks (8 rows) per wave per stage: 4 + 2
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// The K loop of the GEMM as a phase other kernels can run too (conv_dmax3_kernel.h: the ResnetBlock's 1x1 shortcut accumulated into conv2's tile): rows of the A
// operand = the tile's 256 pixels at a_v0 / a_v1 (per-lane byte offsets into q_a0 / q_a1 of this wave's chunks; the second tensor takes over at channel C0),
// rows of the B operand at b_v into q_w, nk stages of 32 channels; every LDS byte from `smem` up to 3 stages is overwritten.  The caller has drained its DMA
// queue and passed a barrier.
typedef int gx3_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gemmx3_phase(f32x4 (&acc)[4][4], char* smem, const gx3_i32x4& q_a0, const gx3_i32x4& q_a1, const gx3_i32x4& q_w, const unsigned (&a_v0)[4],
                                             const unsigned (&a_v1)[4], const unsigned (&b_v)[2], int C0, int nk, int lane, int wave, int wave_m, int wave_n) {
    using C = GemmX3Cfg;
    constexpr int WM = C::WM, WN = C::WN, A_BYTES = C::A_BYTES, STAGE = C::STAGE, A_CPW = C::A_CPW, B_CPW = C::B_CPW;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const gx3_i32x4& rsrc, unsigned lds_addr, unsigned voff, int soff) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lds_addr), "s"(rsrc), "s"(soff)
                     : "memory");
    };
    auto issue = [&](int k, int buf) __attribute__((always_inline)) {
        const int c = k * C::BK;
        const unsigned base = lds0 + buf * STAGE;
        if (c < C0) {
#pragma unroll
            for (int j = 0; j < A_CPW; ++j) dma16(q_a0, base + (wave * A_CPW + j) * 1024, a_v0[j], c * 4);
        } else {
#pragma unroll
            for (int j = 0; j < A_CPW; ++j) dma16(q_a1, base + (wave * A_CPW + j) * 1024, a_v1[j], (c - C0) * 4);
        }
#pragma unroll
        for (int j = 0; j < B_CPW; ++j) dma16(q_w, base + A_BYTES + (wave * B_CPW + j) * 1024, b_v[j], c * 4);
    };
    // hi / lo split of the units this lane fetched into stage buffer `buf`: unit u = 4 g + uu of a row goes to logical slots 4 g + (uu >> 1) (hi) and
    // 4 g + 2 + (uu >> 1) (lo), bytes 8 (uu & 1) ..; logical slot d of row r sits at physical slot d ^ ((r >> 1) & 7).  A chunk is 8 rows: the row's
    // swizzle depends on the chunk's parity position only through (r >> 1) & 7 with r = 8 chunk + (lane >> 3), i.e. ((4 chunk) + (lane >> 4)) & 7.
    auto split_chunk = [&](char* pc, int chunk) __attribute__((always_inline)) {
        const int sw = (4 * chunk + (lane >> 4)) & 7;
        const int u = (lane & 7) ^ sw;
        const int dhi = ((u & 4) | ((u & 3) >> 1)) ^ sw;
        char* rowp = pc + (lane >> 3) * 128;
        const uint4 v = *(const uint4*)(pc + lane * 16);
        const float x0 = __uint_as_float(v.x), x1 = __uint_as_float(v.y), x2 = __uint_as_float(v.z), x3 = __uint_as_float(v.w);
        const unsigned h01 = TI<__bf16>::pack2(x0, x1), h23 = TI<__bf16>::pack2(x2, x3);
        const unsigned l01 = TI<__bf16>::pack2(x0 - __uint_as_float(h01 << 16), x1 - __uint_as_float(h01 & 0xffff0000u));
        const unsigned l23 = TI<__bf16>::pack2(x2 - __uint_as_float(h23 << 16), x3 - __uint_as_float(h23 & 0xffff0000u));
        *(uint2*)(rowp + dhi * 16 + (u & 1) * 8) = make_uint2(h01, h23);
        *(uint2*)(rowp + (dhi ^ 2) * 16 + (u & 1) * 8) = make_uint2(l01, l23);
    };
    auto split = [&](int buf) __attribute__((always_inline)) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < A_CPW; ++j) split_chunk(base + (wave * A_CPW + j) * 1024, wave * A_CPW + j);
#pragma unroll
        for (int j = 0; j < B_CPW; ++j) split_chunk(base + A_BYTES + (wave * B_CPW + j) * 1024, wave * B_CPW + j);
    };

    // fragment addresses: row (16-aligned base + lane % 16); weight side: logical slot 4 ks + kg (the group as stored); pixel side: its hi half, logical slot
    // 4 ks + (kg & 1) -- the lo half is that ^ 2, i.e. the address ^ 32
    const int sw = (lane >> 1) & 7;
    const int ku = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a_off[ks] = (wave_m * WM * 16 + (lane & 15)) * 128 + (((ks * 4 + (ku & 1)) ^ sw) << 4);
        b_off[ks] = A_BYTES + (wave_n * WN * 16 + (lane & 15)) * 128 + (((ks * 4 + ku) ^ sw) << 4);
    }

    // K loop: stage k + 2 is requested behind the barrier that frees its buffer; behind the MFMAs of stage k the wave waits for ITS pieces of stage k + 1 and
    // splits them; the next barrier publishes the split.
    constexpr int CPW = A_CPW + B_CPW;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    split(0);
    int buf = 0;
    for (int k = 0; k < nk; ++k) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int nb = buf + 1 == C::NBUF ? 0 : buf + 1;
        if (k + 2 < nk) issue(k + 2, buf >= 1 ? buf - 1 : C::NBUF - 1);           // (buf + 2) % 3: the buffer stage k - 1 was read from
        const char* base = smem + buf * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);      // see conv_dma_kernel.h
            uint4 ah[WM], al[WM], bfr[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) { ah[i] = *(const uint4*)(base + a_off[ks] + i * (16 * 128)); al[i] = *(const uint4*)(base + (a_off[ks] ^ 32) + i * (16 * 128)); }
#pragma unroll
            for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(base + b_off[ks] + j * (16 * 128));
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const bf16x8 w = __builtin_bit_cast(bf16x8, bfr[j]);          // [w_hi | w_lo]: the MFMA's row operand (mma16t); small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, al[i]), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, ah[i]), acc[i][j], 0, 0, 0);
                }
        }
        if (k + 1 < nk) {
            if (k + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            split(nb);
        }
        buf = nb;
    }
}

__global__ __launch_bounds__(512, 2) void conv_gemmx3_kernel(const ConvArgs a) {
    using C = GemmX3Cfg;
    constexpr int TH = C::TH, TW = C::TW, WM = C::WM, WN = C::WN, BN = C::BN, A_BYTES = C::A_BYTES, STAGE = C::STAGE, A_CPW = C::A_CPW, B_CPW = C::B_CPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / C::WAVES_N, wave_n = wave % C::WAVES_N;

    int mt, nt;
    if (!conv_decode_tile(a, blockIdx.x, mt, nt)) return;
    const int n0 = nt * BN;
    int img0, oy0, ox0, tile_in_img = 0;
    conv_decode_image<TH, TW>(a, mt, img0, tile_in_img, oy0, ox0);

    constexpr unsigned OOB = 0xFFFF0000u;
    unsigned a_v0[A_CPW], a_v1[A_CPW], b_v[B_CPW];
#pragma unroll
    for (int j = 0; j < A_CPW; ++j) {
        const int row = (wave * A_CPW + j) * 8 + (lane >> 3);
        const int u = (lane & 7) ^ ((row >> 1) & 7);
        const int iy = oy0 + row / TW, ix = ox0 + row % TW;
        const bool ok = img0 < a.B && iy < a.Hin && ix < a.Win;
        const unsigned gp = (unsigned)((conv_x_img(a, img0) * a.Hin + iy) * a.Win + ix);
        a_v0[j] = ok ? gp * (unsigned)(a.xs0 * 4) + (unsigned)(u * 16) : OOB;
        a_v1[j] = ok ? gp * (unsigned)(a.xs1 * 4) + (unsigned)(u * 16) : OOB;
    }
#pragma unroll
    for (int j = 0; j < B_CPW; ++j) {
        const int row = (wave * B_CPW + j) * 8 + (lane >> 3);
        const int u = (lane & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        b_v[j] = n < a.w_rows ? (unsigned)(n * a.w_row_stride * 4 + u * 16) : OOB;
    }

    auto make_q = [](const void* p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)p;
        return gx3_i32x4{(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    };
    const gx3_i32x4 q_x0 = make_q(a.x0, a.x0_bytes), q_x1 = make_q(a.x1 ? a.x1 : a.x0, a.x1_bytes);
    const gx3_i32x4 q_w = make_q((const float*)a.w + conv_w_img_offset(a, img0), a.w_bytes);
    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemmx3_phase(acc, smem, q_x0, q_x1, q_w, a_v0, a_v1, b_v, a.C0, a.Cin / C::BK, lane, wave, wave_m, wave_n);
    __syncthreads();
    conv_epilogue<float, TH, TW, WM, WN, 4>(a, acc, smem, true, wave, lane, wave_m, wave_n, img0, oy0, ox0, n0, tile_in_img);
}

}  // namespace wdm
