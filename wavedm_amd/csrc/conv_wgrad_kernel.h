// Weight gradient of a 3x3 stride-1 convolution, direct form (bf16, training step):
//     dW[tap][co][ci] = sum over pixels p of dy[p][co] * x[p + tap][ci]                                   (autograd of models/unet.py:45,65,91,100)
// a GEMM whose CONTRACTION index is the pixel.  The batched-GEMM form (train.hip: conv_wgrad) first writes channel-major copies of dy and of three dx-shifted
// x to HBM (gather_t) and then streams them again at 86 FLOP per byte.  Here both operands stay as they are -- NHWC -- all the way into LDS (LDS-DMA, like the
// forward kernels: an 8 x 16-pixel chunk of dy and its 10 x 18 halo of x per stage, double-buffered), and the MFMA fragments, which need 8 consecutive PIXELS of
// one channel per lane, are read with ds_read_b64_tr_b16 (gfx950's transposing LDS read: the 16 lanes of a group pass the addresses of a [4 rows][16 columns]
// block, 4 x 8 bytes per row, and lane i receives column i).  The nine taps are nine row offsets into the same halo tile; their accumulators stay in registers
// (a wave owns 64 co x 16 ci x 9 taps = 144 registers), the pixels are split over workgroups (split K) and every workgroup writes one fp32 partial
// [tap][split][co][ci] that reduce_wgrad_kernel sums in a fixed order: no atomics, deterministic.
//
// Workgroup = 128 co x 64 ci x 9 taps, 8 waves as 2 (co halves) x 4 (ci fragments).  Per k-step of 32 pixels (two image rows of the chunk) a wave reads
// 4 dy fragments + 9 x fragments (26 transposing reads) for 36 MFMAs.
//
// LDS images (bank = (byte / 4) % 64; one LDS cycle serves 32 lanes = two 16-lane groups = 8 rows x 32 bytes, which must fall on 8 different bank octets):
//   dy chunk   row = pixel y * 16 + x of the chunk, 256 bytes = 128 co = eight 32-byte segments; segment b of row r sits at position (b + f(r)) & 7,
//              f(r) = (r & 3) + 4 ((r >> 3) & 1): the rows r .. r + 3 and r + 8 .. r + 11 of one cycle take the eight positions once each
//   x halo     row = hy * 24 + hx (10 x 18 halo pixels in 24-slot rows), 128 bytes = 64 ci = four segments; segment b of row R sits at position
//              (b + g(R)) & 3, g(R) = ((R >> 1) & 1) + 2 ((R >> 3) & 1): with the row's parity that separates R .. R + 3 and R + 8 .. R + 11 for every start
//              R (the dx taps shift the start by one row); a tap row (+ 24 rows) flips g by 2 and a k-step (+ 48 rows) leaves it alone, so a lane needs
//              twelve base addresses (dx x half x tap-row parity) and immediate offsets.
// The DMA is lane-linear (lane L writes bytes [16 L, 16 L + 16) of its 1 KB piece), so the rotation is applied to the SOURCE address of each lane.
#pragma once
#include "conv_kernel.h"

namespace wdm {

struct WgradArgs {
    const void* dy;             // [B][H][W][cout] bf16, dense
    const void* x0;             // [B][H][W][xs0] bf16: channels [0, C0) of the conv input
    const void* x1;             // channels [C0, C0 + C1) (nullptr: single input)
    float* part;                // [9][S][rows_g][cin] fp32
    int B, H, W, cout, C0, C1, xs0, xs1, cin, rows_g;
    int S, nchunk, cps;         // splits, chunks (8 x 16 pixels) in the batch, chunks per split
    int n_co, n_ci;             // tiles of 128 co / 64 ci
    unsigned dy_bytes, x0_bytes, x1_bytes;
};

// M8 = false: maps that are multiples of 8 x 16 pixels, chunk = 8 rows x 16 columns of one image, halo 10 x 18 in 24-slot rows.
// M8 = true: 8 x 8 maps, chunk = TWO images (the same 128 dy rows: row = image * 64 + y * 8 + x, a k-step is four image rows of eight pixels), halo 2 x (10 x 10) in
// 16-slot rows; the two 16-lane groups of an LDS cycle then read rows R .. R + 3 and R + 16 .. R + 19 (the next image row), so g uses bit 4 where the 16-wide form uses bit 3.
template <bool M8>
struct WgradCfg {
    static constexpr int NTHREADS = 512;
    static constexpr int DY_BYTES = 128 * 256;                 // 32 KB
    static constexpr int RS = M8 ? 16 : 24;                     // halo row slots per halo row
    static constexpr int XROWS = M8 ? 2 * 10 * 16 : 10 * 24;    // 320 | 240 halo row slots
    static constexpr int X_BYTES = XROWS * 128;                 // 40 | 30 KB
    static constexpr int STAGE = DY_BYTES + X_BYTES;            // 72 | 62 KB
    static constexpr int LDS_BYTES = 2 * STAGE;                 // 144 | 124 KB
    static constexpr int PIECES = STAGE / 1024;                 // 72 | 62
    static constexpr int NP = (PIECES + 7) / 8;                 // pieces per wave: 9 | 8
    static constexpr int GBIT = M8 ? 4 : 3;                     // g(R) = ((R >> 1) & 1) + 2 ((R >> GBIT) & 1)
    // byte offset of k-step ks / tap row ty inside the x halo image (both leave (R >> 1) & 1 alone; a tap row flips bit GBIT, a k-step does not)
    static constexpr int ks_off(int ks) { return M8 ? ((ks & 1) * 64 + (ks >> 1) * 160) * 128 : ks * 48 * 128; }
    static constexpr int ty_off(int ty) { return ty * RS * 128; }
};

template <int OFF>
__device__ __forceinline__ unsigned long long wg_tr(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// The reads above are invisible to the compiler's wait-count bookkeeping: a fragment is used only after it went through one of these (LDS reads return in
// order: lgkmcnt(N) = everything but the N youngest has landed).  The "+v" operands pin the registers between the read and the wait and order the MFMAs behind it.
template <int N>
__device__ __forceinline__ void wg_wait2(unsigned long long& a, unsigned long long& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wg_wait10(unsigned long long (&l)[4], unsigned long long (&h)[4], unsigned long long& a, unsigned long long& b) {
    asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(a), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ uint4 wg_frag(unsigned long long lo, unsigned long long hi) { return uint4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)}; }

// One k-step (32 pixels = two image rows of the chunk), software-pipelined: on entry the four dy fragments and the x fragment of tap 0 are in flight; every
// tap requests the next tap's fragment before it waits for its own; the last tap requests the next k-step's first five (not across a chunk boundary:
// the next chunk's buffer is published by a barrier).
template <int KS, bool M8>
__device__ __forceinline__ void wg_kstep(f32x4 (&acc)[9][4], const unsigned (&aa)[4], const unsigned (&ba)[3][2][2], unsigned so, unsigned long long (&al)[4],
                                         unsigned long long (&ah)[4], unsigned long long& b0l, unsigned long long& b0h) {
    using C = WgradCfg<M8>;
    uint4 af[4];
    unsigned long long bl = b0l, bh = b0h, nl = 0, nh = 0;
#define WDM_WG_TAP(T)                                                                                                                              \
    do {                                                                                                                                            \
        constexpr int ty = (T) / 3, dx = (T) % 3, ty1 = ((T) + 1) / 3, dx1 = ((T) + 1) % 3;                                                         \
        if ((T) < 8) {                                                                                                                              \
            nl = wg_tr<C::ks_off(KS) + C::ty_off(ty1)>(ba[dx1][0][ty1 & 1] + so);                                                                           \
            nh = wg_tr<C::ks_off(KS) + C::ty_off(ty1)>(ba[dx1][1][ty1 & 1] + so);                                                                           \
        } else if (KS < 3) {                                                                                                                        \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) { al[m] = wg_tr<(KS + 1) * 8192>(aa[m] + so); ah[m] = wg_tr<(KS + 1) * 8192 + 1024>(aa[m] + so); } \
            nl = wg_tr<C::ks_off(KS + 1)>(ba[0][0][0] + so);                                                                                          \
            nh = wg_tr<C::ks_off(KS + 1)>(ba[0][1][0] + so);                                                                                          \
        }                                                                                                                                           \
        if ((T) == 0) {                                                                                                                             \
            wg_wait10<2>(al, ah, bl, bh);                                                                                                           \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) af[m] = wg_frag(al[m], ah[m]);                                                            \
        } else if ((T) < 8) wg_wait2<2>(bl, bh);                                                                                                    \
        else if (KS < 3) wg_wait2<10>(bl, bh);                                                                                                      \
        else wg_wait2<0>(bl, bh);                                                                                                                   \
        const uint4 bf = wg_frag(bl, bh);                                                                                                           \
        _Pragma("unroll") for (int m = 0; m < 4; ++m) mma16<__bf16>(acc[ty * 3 + dx][m], af[m], bf);                                                \
        bl = nl; bh = nh;                                                                                                                           \
        (void)dx1;                                                                                                                                  \
    } while (0)
    WDM_WG_TAP(0); WDM_WG_TAP(1); WDM_WG_TAP(2); WDM_WG_TAP(3); WDM_WG_TAP(4); WDM_WG_TAP(5); WDM_WG_TAP(6); WDM_WG_TAP(7); WDM_WG_TAP(8);
#undef WDM_WG_TAP
    b0l = bl; b0h = bh;
}

template <bool M8>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const WgradArgs a) {
    using C = WgradCfg<M8>;
    constexpr int NP = C::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_co = wave >> 2, wave_ci = wave & 3;

    const int ntile = a.n_co * a.n_ci;
    const int bid = blockIdx.x;
    const int split = bid / ntile, tile = bid - split * ntile;
    const int cot = tile / a.n_ci, cit = tile - cot * a.n_ci;
    const int co0 = cot * 128, ci0 = cit * 64;
    const bool from0 = ci0 < a.C0;
    const int cil0 = from0 ? ci0 : ci0 - a.C0;                  // first channel of the tile inside its tensor
    const int xs = from0 ? a.xs0 : a.xs1;
    const int cx = from0 ? a.C0 : a.C1;                         // channels of that tensor

    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto make_q = [](const void* p, unsigned bytes) __attribute__((always_inline)) {
        const unsigned long long v = (unsigned long long)p;
        return i32x4{(int)(unsigned)v, (int)((unsigned)(v >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
    };
    const i32x4 q_dy = make_q(a.dy, a.dy_bytes);
    const i32x4 q_x = from0 ? make_q(a.x0, a.x0_bytes) : make_q(a.x1, a.x1_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto dma16 = [&](const i32x4& rsrc, unsigned lds_addr, unsigned voff) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(lds_addr), "s"(rsrc)
                     : "memory");
    };
    constexpr unsigned OOB = 0xFFFFFFF0u;
    constexpr int NONE = -2147483647 - 1;

    // ---- DMA sources of this wave's pieces, relative to the chunk's first pixel (chunk-invariant part)
    // pieces 0 .. 31: dy (4 pixel rows each), 32 .. PIECES - 1: x halo (8 row slots each), the rest: none
    int rel[NP];                // byte offset relative to the chunk origin (dy: of the row's pixel; x: of the halo pixel), or NONE: never fetched
    int hyx[NP];                // x pieces: (second image of the chunk) << 16 | hy << 8 | hx of this lane's row (validity is per chunk)
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int q = wave * NP + j;
        rel[j] = NONE; hyx[j] = 0;
        if (q < 32) {
            const int r = 4 * q + (lane >> 4);
            const int s16 = lane & 15;
            const int f = (r & 3) + 4 * ((r >> 3) & 1);
            const int b = ((s16 >> 1) - f) & 7;
            const int co = co0 + 16 * b + 8 * (s16 & 1);
            const int pix = M8 ? r : (r >> 4) * a.W + (r & 15);            // pixel index relative to the chunk origin
            if (co < a.cout) { rel[j] = (pix * a.cout + co) * 2; hyx[j] = M8 ? (r >> 6) << 16 : 0; }
        } else if (q < C::PIECES) {
            const int R = 8 * (q - 32) + (lane >> 3);
            const int im = M8 ? R / 160 : 0, Rr = M8 ? R - im * 160 : R;
            const int hy = Rr / C::RS, hx = Rr - hy * C::RS;
            const int s8 = lane & 7;
            const int g = ((R >> 1) & 1) + 2 * ((R >> C::GBIT) & 1);
            const int b = ((s8 >> 1) - g) & 3;
            const int ci = cil0 + 16 * b + 8 * (s8 & 1);
            const int wpx = M8 ? 8 : a.W;
            if (hx < (M8 ? 10 : 18) && ci < cx) { rel[j] = ((im * 64 + (hy - 1) * wpx + (hx - 1)) * xs + ci) * 2; hyx[j] = (im << 16) | (hy << 8) | hx; }
        }
    }
    auto issue = [&](int chunk, int buf) __attribute__((always_inline)) {
        int img, y0, x0;
        if (M8) { img = 2 * chunk; y0 = 0; x0 = 0; }
        else {
            const int xbn = a.W >> 4, ybn = a.H >> 3;
            const int xb = chunk % xbn, t = chunk / xbn;
            const int yb = t % ybn;
            img = t / ybn; y0 = yb * 8; x0 = xb * 16;
        }
        const long long origin = ((long long)img * a.H + y0) * a.W + x0;              // pixel index of the chunk's first pixel
        const unsigned o_dy = (unsigned)(origin * a.cout * 2), o_x = (unsigned)(origin * xs * 2);
        const unsigned base = lds0 + buf * C::STAGE;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = wave * NP + j;
            const int im = hyx[j] >> 16, hy = (hyx[j] >> 8) & 255, hx = hyx[j] & 255;
            if (q < 32) {
                dma16(q_dy, base + q * 1024, rel[j] != NONE && img + im < a.B ? o_dy + (unsigned)rel[j] : OOB);
            } else if (q < C::PIECES) {
                const bool ok = rel[j] != NONE && img + im < a.B && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
                dma16(q_x, base + C::DY_BYTES + (q - 32) * 1024, ok ? o_x + (unsigned)rel[j] : OOB);
            }
        }
    };

    // ---- fragment addresses (bytes inside a stage)
    const int kq = lane >> 4, i16 = lane & 15;
    unsigned a_addr[4];         // dy fragment m (co block 16 m of this wave's 64): rows of k-step 0, first half
    {
        const int r = 8 * kq + (i16 >> 2);            // 16-wide chunks: image row kq >> 1, columns 8 (kq & 1) ..; 8 x 8 maps: image row kq
        const int f = (r & 3) + 4 * ((r >> 3) & 1);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int cb = wave_co * 4 + m;
            a_addr[m] = lds0 + (unsigned)(r * 256 + ((cb + f) & 7) * 32 + (i16 & 3) * 8);
        }
    }
    unsigned b_addr[3][2][2];   // [dx][half][tap-row parity]: k-step 0, tap row 0 (odd tap rows: the flipped segment)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int R = (M8 ? kq * 16 : (kq >> 1) * 24 + 8 * (kq & 1)) + 4 * h + dx + (i16 >> 2);
            const int g = ((R >> 1) & 1) + 2 * ((R >> C::GBIT) & 1);
            const unsigned v = lds0 + C::DY_BYTES + (unsigned)(R * 128 + (i16 & 3) * 8);
            b_addr[dx][h][0] = v + (unsigned)(((wave_ci + g) & 3) * 32);
            b_addr[dx][h][1] = v + (unsigned)(((wave_ci + g + 2) & 3) * 32);
        }

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int c_begin = split * a.cps, c_end = min(c_begin + a.cps, a.nchunk);
    if (c_begin < c_end) issue(c_begin, 0);
    int buf = 0;
    for (int c = c_begin; c < c_end; ++c) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // chunk c landed everywhere; everyone is done with the other buffer
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < c_end) issue(c + 1, buf ^ 1);
        const unsigned so = (unsigned)(buf * C::STAGE);
        unsigned long long al[4], ah[4], bl, bh;
#pragma unroll
        for (int m = 0; m < 4; ++m) { al[m] = wg_tr<0>(a_addr[m] + so); ah[m] = wg_tr<1024>(a_addr[m] + so); }
        bl = wg_tr<0>(b_addr[0][0][0] + so);
        bh = wg_tr<0>(b_addr[0][1][0] + so);
        wg_kstep<0, M8>(acc, a_addr, b_addr, so, al, ah, bl, bh);
        wg_kstep<1, M8>(acc, a_addr, b_addr, so, al, ah, bl, bh);
        wg_kstep<2, M8>(acc, a_addr, b_addr, so, al, ah, bl, bh);
        wg_kstep<3, M8>(acc, a_addr, b_addr, so, al, ah, bl, bh);
        buf ^= 1;
    }

    // ---- partial [tap][split][co][ci]: a lane holds co = 4 (lane >> 4) + j of fragment m, ci = lane & 15 of this wave's fragment
    const int ci = ci0 + wave_ci * 16 + i16;
    if (ci < a.cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = co0 + wave_co * 64 + m * 16 + kq * 4 + j;
                    if (co < a.cout) a.part[(((long long)t * a.S + split) * a.rows_g + co) * a.cin + ci] = acc[t][m][j];
                }
    }
}

}  // namespace wdm
