// Fused implicit-GEMM convolution for gfx950 (MI355X) -- the kernel that carries >98 % of the
// UNet's FLOPs (conv3x3 91 %, conv1x1 8 %, attention QK^T / PV 1 %).
//
// One kernel template covers
//   MODE_S1  conv3x3 stride 1 pad 1            (ResnetBlock conv1/conv2, conv_in, conv_out; unet.py:91,100,233,303)
//   MODE_S2  pad(0,1,0,1) + conv3x3 stride 2    (Downsample, unet.py:71-78)
//   MODE_UPS nearest x2 + conv3x3 pad 1         (Upsample, unet.py:51-56; the upsample is folded into the LDS
//                                                read address, the x2 tensor is never materialised)
//   MODE_P1  conv1x1 / plain GEMM               (nin_shortcut, q/k/v/proj_out, and the attention products
//                                                Q.K^T and P.V with per-image "weights"; unet.py:113,147-189)
// with, fused in:
//   * channel concat of two inputs (torch.cat([h, skip]) at unet.py:380 is never materialised),
//   * GroupNorm-apply + SiLU on the A operand while it is staged into LDS (x*scale[b,c]+shift[b,c], then
//     x*sigmoid(x)); zero padding is applied AFTER the activation like the reference,
//   * epilogue: alpha*acc + bias[n] + temb[b,n] + residual, store as NHWC / NCHW in bf16 or f32.
//
// GEMM view: M = output pixels (a TH x TW patch of NI images per workgroup), N = output channels,
// K = taps x Cin.  Activations are NHWC so the K (channel) axis is contiguous for both operands
// (weights are packed [tap][cout][cin]): every MFMA fragment is one 16-byte LDS read per lane.
//
// LDS image (both operands):   [row slot q][4 k-units, rotated][16 bytes]   -- 64 bytes per pixel / weight row
//   a "unit" is 16 bytes of consecutive channels (8 bf16 / 4 f32); unit u of row q lives in 16-byte slot
//   4q + (u ^ ((q>>1)&2)).  Lane l of a wave reads unit (l>>4) of row base+(l&15).  With that rotation
//     * ds_read_b128 (16-lane service groups {0-3,12-15,20-27}, ..., banks = 16 slots of 16 B): the 8 lanes of a
//       group that share a unit cover every (q&3) twice with opposite ((q>>2)&1) -> 8 distinct slots, and the two
//       units of a group differ in parity -> all 16 distinct for ANY base row (taps shift the base): conflict-free;
//     * ds_write_b128 (8 contiguous lanes = 2 pixels x 4 units = one 128-byte bank row): conflict-free.
//   (The first version used [unit][row][16 B]: conflict-free reads but 4-way conflicts on every staging store;
//   rocprofv3 showed SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.46 -- see profiles/.)
//   bf16: one v_mfma_f32_16x16x32_bf16 consumes the 4 units (K = 32);  f32 (parity mode): four
//   v_mfma_f32_16x16x4_f32, MFMA j taking element j of every unit (K = 16) -- exact fp32 FMA chains.
//
// Pipeline per K stage (one 32/16-channel slab x all 9 taps, or 4 slabs for 1x1):
//   barrier | registers -> LDS (A transform here) | barrier | issue global loads of the NEXT stage into
//   registers | MFMAs of this stage from LDS.   Global latency hides under the MFMAs; two workgroups per CU
//   (<= 58 KB LDS each for the main configs) overlap one's LDS-write phase with the other's MFMA phase.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef WDM_ABL
#define WDM_ABL 0      // ablation mask, only ever set by tools/conv_ablate.hip
#endif
// cache policy of the epilogues' output stores (buffer instruction aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1).  Non-temporal: the outputs are
// streamed once, the consumer is another kernel.  Measured (round 4, same box, 20 DDIM steps, five A/B pairs): nt +0.4 ... +1.1 % end to end in every pair,
// write-through (sc1) +0.8 %, both together +-0; non-temporal LOADS of the halo tiles -1.5 % (the 16 tiles of an image re-read each other's borders from L2).
#ifndef WDM_STORE_AUX
#define WDM_STORE_AUX 2
#endif

namespace wdm {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short bf16_raw;

enum { MODE_S1 = 0, MODE_S2 = 1, MODE_UPS = 2, MODE_P1 = 3, MODE_UP4 = 4 };   // MODE_UP4: conv_up4_kernel.h (host-side selection only)
enum { Y_NHWC = 0, Y_NCHW = 1, Y_NCHW_F32 = 2, Y_NHWC_F32 = 3 };

struct ConvArgs {
    const void* x0;
    const void* x1;
    int C0, C1;            // channels taken from x0 / x1 (Cin = C0 + C1, C1 == 0: single input)
    int xs0, xs1;          // pixel stride of x0 / x1 in elements (>= C0 / C1)
    int B, Hin, Win, Hout, Wout;
    int Cin, Cout;
    const void* w;         // [tap][row][cin] (row = output channel), model dtype
    long long w_tap_stride, w_img_stride;  // elements; w_img_stride != 0: per-image weights (attention)
    int w_row_stride;      // elements
    const void* w_sm;      // slab-major copy of w, [slab][tap][row][32] (bf16 3x3 layers, k_pack_conv_sm), or nullptr: the 8x8 LDS-DMA kernel reads it
    int w_slab_stride;     // LDS-DMA 3x3 kernels: elements between the 32-channel slabs of a weight row (0 = 32: the plain [tap][row][cin] matrix;
                           // slab-major copies [slab][tap][row][32], ConvW::w_sm, make every 1 KB DMA piece one contiguous run of whole cache lines)
    int w_split;           // conv_dmax3_kernel.h: 1 = w is the pre-split copy (k_pack_conv_sm, f32x3): no split pass over the weight sub-stages
    int w_rows;            // informational: rows of the weight matrix (columns >= Cout are never stored)
    unsigned x0_bytes, x1_bytes, w_bytes;   // extents for the buffer descriptors (reads past them return 0)
    const float* bias;     // [Cout] or nullptr
    float alpha;           // accumulator scale (attention: C^-0.5), 1 otherwise
    int pro;               // 0: none, 1: x*scale+shift then SiLU  (MODE_S1 only)
    const float* scale;    // [B][Cin]
    const float* shift;    // [B][Cin]
    const float* temb;     // [n_t][temb_ld] or nullptr, added per (image, n)
    int temb_ld;
    int temb_per_image;    // 1: row = image index, 0: row 0 for every image
    const void* res;       // residual, NHWC model dtype, pixel stride res_s; or nullptr
    int res_s;
    void* y;
    int y_mode;            // Y_*
    int y_s;               // NHWC pixel stride of y in elements
    int mtiles, ntiles;
    int grid_gn;           // XCD N-groups (1, 2, 4 or 8), see the kernel's tile mapping
    float* stats;          // optional: GroupNorm partial statistics of the output, float4[B][stats_nslab][Cout] (see elementwise.hip)
    int stats_nslab;       // slabs per image = tiles per image x wave tiles (in M) per tile
    float* gst;            // optional, with stats: GROUP-level partials of the output, float[B][stats_nslab][32][3] = (pivot, sum(x-K), sum((x-K)^2)) over the
                           // slab's rows x the group's Cout / 32 channels (4, 8 or 16: a group never straddles a wave's columns) -- what a consumer conv needs
                           // to finalise GroupNorm in its own prologue (gn_inline.h) instead of a gn_finalize launch
    // consumer side (LDS-DMA 3x3 kernels): pro != 0 with gin set = the GroupNorm of x0 (single input, Cin = 128 / 256 / 512) is finalised in the prologue from
    // x0's group partials and the norm's weights; scale / shift are then unused
    const float* gin;      // float[B][gin_nslab][32][3]
    int gin_nslab;
    const float* gn_gamma; // [Cin]
    const float* gn_beta;  // [Cin]
    float gn_eps;
    // producer side, in-tile GroupNorm of the OUTPUT (gn_group.h: gn_out_tail): with stats set, kernels whose tile holds whole images x whole groups also
    // write yn = act(GroupNorm(y)) -- the normalised (+ SiLU) copy the consumer would otherwise get from a gn_finalize_apply launch
    void* yn;              // dense [B][Hout][Wout][Cout] model dtype, or nullptr
    const float* on_gamma; // [Cout]
    const float* on_beta;  // [Cout]
    float on_eps;
    int on_silu;
    // producer side, GroupNorm of the CONSUMER finalised by the last workgroup of an image (gn_arrive.h): with stats and fin_cnt set, every workgroup adds its tiles
    // to fin_cnt[image] once its statistics are out; the one that completes fin_total runs gn_finalize_kernel's reduction for the image -- over this conv's partials
    // [| fin_st1, the partials of the tensor the consumer concatenates behind it] -- and writes the consumer's scale / shift rows.  No gn_finalize launch.
    int* fin_cnt;          // [B], zero between launches (the last arriver resets its image's counter), or nullptr
    int fin_total;         // tiles that complete an image: (M tiles per image) x (N tiles) of this launch
    const float* fin_st1;  // float4[B][fin_nslab1][fin_C1] or nullptr
    int fin_nslab1, fin_C1;
    const float* fin_gamma; // [Cout + fin_C1]
    const float* fin_beta;
    float fin_eps, fin_premul;      // premul: -log2(e) for a consumer with the GroupNorm+SiLU prologue (k_gn_finalize: for_silu_conv), else 1
    float* fin_scale;      // [B][Cout + fin_C1]
    float* fin_shift;
    int* query_fin;        // host only: when set, the launcher stores 1 if the kernel it would pick arrives (else 0); used with query_nslab
    int* query_yn;         // host only: when set, the launcher stores 1 if the kernel it would pick for this shape can write yn (else 0) and does not launch
    int* query_nslab;      // host only: when set, the launcher stores stats_nslab for this shape here and does not launch
    long long m_valid;     // 0: every pixel of the (B,Hout,Wout) grid exists; > 0: only the first m_valid flattened pixels do
                           //    (plain GEMMs over M rows that do not fill the last row of the 16-wide pixel grid)
    // optional second contraction accumulated into the same output tile (conv_dma_kernel.h only): the ResnetBlock's 1x1
    // nin_shortcut over the block input [sx0 | sx1] (unet.py:134-137), so that `x_shortcut + h` is one accumulator
    const void* sx0;       // nullptr: none
    const void* sx1;
    int sC0, sC1, sxs0, sxs1;
    const void* sw;        // [row][sC0 + sC1] model dtype
    int sw_row_stride, sw_rows;
    unsigned sx0_bytes, sx1_bytes, sw_bytes;
    const float* sbias;    // [Cout] added like bias
    // batched-GEMM addressing for the training wgrad (1x1 mode, per-image weights): image i = tap * img_mod + b reads the input of
    // image b (x0 is shared by the taps) and the weight matrix of image b displaced by the tap:
    //   w + b * w_img_stride + (tap % 3) * w_tx_stride + (tap / 3 - 1) * w_ty_stride        (img_mod == 0: off)
    int img_mod;
    long long w_tx_stride, w_ty_stride;
    int x_img_shared;      // 1: every image of the batch reads the rows of image 0 of x0 (the A operand is a weight matrix and the per-image
                           // "weights" are activations: V^T = W_v . h^T of the attention block)
    // sub-pixel form of Upsample (conv_up4_kernel.h): the grid is the LOW-resolution map, N tile nt belongs to output phase nt / up4_ntp
    // (py = phase >> 1, px = phase & 1) and writes pixel (2 oy + py, 2 ox + px) of the (2 Hout) x (2 Wout) output
    int up4, up4_ntp;
#ifdef WDM_EPI_TS
    unsigned long long* ts;    // ablation builds (tools/dma_ablate.hip): s_memtime stamps of workgroup WDM_EPI_TS, [wave][8]
#endif
};
#ifdef WDM_EPI_TS
#define WDM_ETS(k) do { if (blockIdx.x == WDM_EPI_TS && (threadIdx.x & 63) == 0) a.ts[(threadIdx.x >> 6) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WDM_ETS(k) do { } while (0)
#endif

// element offset of the weight matrix image `img` reads, and the image whose input it reads
__host__ __device__ inline long long conv_w_img_offset(const ConvArgs& a, int img) {
    if (a.w_img_stride == 0) return 0;
    if (a.img_mod == 0) return (long long)img * a.w_img_stride;
    const int tap = img / a.img_mod, b = img - tap * a.img_mod;
    return (long long)b * a.w_img_stride + (long long)(tap % 3) * a.w_tx_stride + (long long)(tap / 3 - 1) * a.w_ty_stride;
}
__host__ __device__ inline int conv_x_img(const ConvArgs& a, int img) { return a.x_img_shared ? 0 : a.img_mod ? img % a.img_mod : img; }

// Workgroup -> (M tile, N tile) and tile -> (image, tile of the image) without runtime integer divisions when the divisors are powers of two (they
// are for every layer of the model): a scalar division by a run-time value is a ~25-instruction dependent chain through v_rcp_iflag_f32, and the ten
// of them at the head of every conv kernel were most of the 2 400 ticks a workgroup spent before its first DMA (tools/dma_ablate.hip -DWDM_EPI_TS).
__device__ __forceinline__ void udivmod_fast(int x, int d, int& q, int& r) {        // x >= 0, d > 0
    if ((d & (d - 1)) == 0) { const int sh = __builtin_ctz(d); q = x >> sh; r = x & (d - 1); }
    else { q = x / d; r = x - q * d; }
}
// false: this workgroup has no tile (the grid is rounded up per XCD)
template <class AT>
__device__ __forceinline__ bool conv_decode_tile(const AT& a, int bid, int& mt, int& nt) {      // AT: ConvArgs in any address space
    const int gn = a.grid_gn;
    const int xcd = bid & 7, seq = bid >> 3;
    if (gn == 1) {                   // N fastest: the N tiles of one M tile back to back on one XCD
        const int mcnt = (a.mtiles - xcd + 7) >> 3;
        if (seq >= mcnt * a.ntiles) return false;
        int q, r;
        udivmod_fast(seq, a.ntiles, q, r);
        nt = r; mt = xcd + 8 * q;
    } else {                         // M fastest inside an (xm, xn) group of XCDs
        const int gm = 8 / gn;
        const int xn = xcd % gn, xm = xcd / gn;
        const int ncnt = (a.ntiles - xn + gn - 1) / gn, mcnt = (a.mtiles - xm + gm - 1) / gm;
        if (ncnt <= 0 || mcnt <= 0 || seq >= mcnt * ncnt) return false;
        int q, r;
        udivmod_fast(seq, mcnt, q, r);
        mt = xm + gm * r; nt = xn + gn * q;
    }
    return true;
}
template <int TH, int TW, class AT>
__device__ __forceinline__ void conv_decode_image(const AT& a, int mt, int& img0, int& tile_in_img, int& oy0, int& ox0) {
    const int twn = a.Wout / TW;
    const int tpi = (a.Hout / TH) * twn;
    udivmod_fast(mt, tpi, img0, tile_in_img);
    int ty, tx;
    udivmod_fast(tile_in_img, twn, ty, tx);
    oy0 = ty * TH; ox0 = tx * TW;
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_raw v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) { return __builtin_bit_cast(bf16_raw, (__bf16)f); }

template <typename T> struct TI;
template <> struct TI<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
    __device__ static __forceinline__ float ld(const void* p, long long i) { return ((const float*)p)[i]; }
    __device__ static __forceinline__ void st(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct TI<__bf16> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    __device__ static __forceinline__ unsigned pack2(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (RNE)
        typedef float v2f __attribute__((ext_vector_type(2)));
        typedef __bf16 v2b __attribute__((ext_vector_type(2)));
        const v2f v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, v2b));
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    __device__ static __forceinline__ float raw16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }
    __device__ static __forceinline__ void unpack2(unsigned u, float& lo, float& hi) { lo = __uint_as_float(u << 16); hi = __uint_as_float(u & 0xffff0000u); }
    __device__ static __forceinline__ float ld(const void* p, long long i) { return bf16_to_f32(((const bf16_raw*)p)[i]); }
    __device__ static __forceinline__ void st(void* p, long long i, float v) { ((bf16_raw*)p)[i] = f32_to_bf16(v); }
};

// IEEE half ("f16" mode): the bf16 kernels on fp16 operands.  Same instruction counts as bf16 -- v_cvt_f32_f16 (SDWA for the upper half) per unpacked element,
// one v_cvt_pk_f16_f32 (RNE) per packed pair, v_mfma_f32_16x16x32_f16 at the bf16 MFMA's rate -- with three more mantissa bits and a range of +-65504.
typedef _Float16 f16_t;
template <> struct TI<f16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void unpack(const uint4& u, float* f) {
        typedef f16_t v8h __attribute__((ext_vector_type(8)));
        typedef float v8f __attribute__((ext_vector_type(8)));
        const v8f v = __builtin_convertvector(__builtin_bit_cast(v8h, u), v8f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = v[e];
    }
    __device__ static __forceinline__ unsigned pack2(float lo, float hi) {      // one v_cvt_pk_f16_f32 (RNE)
        typedef float v2f __attribute__((ext_vector_type(2)));
        typedef f16_t v2h __attribute__((ext_vector_type(2)));
        const v2f v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, v2h));
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
    }
    __device__ static __forceinline__ float raw16_to_f32(unsigned short v) { return (float)__builtin_bit_cast(f16_t, v); }
    __device__ static __forceinline__ void unpack2(unsigned u, float& lo, float& hi) {
        typedef f16_t v2h __attribute__((ext_vector_type(2)));
        const v2h h = __builtin_bit_cast(v2h, u);
        lo = (float)h[0]; hi = (float)h[1];
    }
    __device__ static __forceinline__ float ld(const void* p, long long i) { return (float)((const f16_t*)p)[i]; }
    __device__ static __forceinline__ void st(void* p, long long i, float v) { ((f16_t*)p)[i] = (f16_t)v; }
};

// Range of the f16 mode: every kernel that packs fp16 sets MODE.FP16_OVFL at its start -- an overflowing conversion then gives +-65504 instead of +-inf (true
// infinities and NaNs pass: measured on gfx950, tools: v_cvt_pk_f16_f32 of 7e4 -> 0x7bff with the bit, 0x7c00 without).  An activation beyond the fp16 range
// (none occurs in this architecture: GroupNorm bounds every conv input) saturates instead of turning the rest of the trajectory into NaNs; weights outside the
// range are refused when they are loaded (wdm_unet_load_param).  One scalar instruction per kernel; nothing for the other element types.
template <typename T> __device__ __forceinline__ void h16_mode_init() {}
template <> __device__ __forceinline__ void h16_mode_init<f16_t>() { __builtin_amdgcn_s_setreg((1 - 1) << 11 | 23 << 6 | 1, 1); }      // hwreg(HW_REG_MODE, 23, 1) <- 1

__device__ __forceinline__ float silu_f(float v) {
    // x * sigmoid(x) = x / (1 + e^-x)   (unet.py:31-33); v_exp_f32 + v_rcp_f32 (1 ulp each) instead of an IEEE divide:
    // this runs once per staged element in the conv prologue, where VALU issue slots compete with the MFMA stream
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

// GroupNorm-apply + SiLU on one 16-byte unit, written for the fewest issue slots (the SIMD's instruction issue, not a
// pipe, bounds the conv kernel): scale/shift arrive pre-multiplied by -log2(e), so per PAIR of elements it is
//   t = x*sc' + sh' (v_pk_fma)   e = 2^t (2 v_exp)   d = 1 + e (v_pk_add)   r = 1/d (2 v_rcp)
//   v = t * (-ln 2) (v_pk_mul)   y = v * r (v_pk_mul)          =>  y = v * sigmoid(v),  v = x*scale + shift
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ uint4 gn_silu_unit(const uint4& u, const float* sc, const float* sh) {
    constexpr int VEC = TI<T>::VEC;
    float f[VEC];
    TI<T>::unpack(u, f);
#pragma unroll
    for (int e = 0; e < VEC; e += 2) {
        const f32x2 x = {f[e], f[e + 1]}, s2 = {sc[e], sc[e + 1]}, h2 = {sh[e], sh[e + 1]};
        const f32x2 t = x * s2 + h2;
#if defined(WDM_SILU_ABL) && WDM_SILU_ABL == 1      // tools: the transform without its four transcendentals per pair (timing only, wrong values)
        f32x2 d = t;
        d = d + 1.0f;
        const f32x2 r = d;
#elif defined(WDM_SILU_ABL) && WDM_SILU_ABL == 2    // tools: without the two reciprocals
        f32x2 d = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        d = d + 1.0f;
        const f32x2 r = d;
#else
        f32x2 d = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
        d = d + 1.0f;
        const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
#endif
        const f32x2 y = (t * -0.6931471805599453f) * r;
        f[e] = y.x; f[e + 1] = y.y;
    }
    return TI<T>::pack(f);
}

// "f32x3": fp32 tensors everywhere (same memory format and the same elementwise math as float), but every product of the contractions is
// taken as three bf16 MFMAs on operands split hi + lo in registers: a*b ~ ah*bh + ah*bl + al*bh with fp32 accumulation.  hi = bf16(x)
// (RNE), lo = bf16(x - hi) carry 16-17 mantissa bits of x, the dropped al*bl term is <= 2^-16 of the product: ~1e-5 end to end, inside
// north_star's 1e-3, at several times the rate of the exact v_mfma_f32_16x16x4_f32 chain (which runs at the fp32 VALU rate).
struct f32x3_t { float v; };
template <> struct TI<f32x3_t> : TI<float> {};

template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<__bf16>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<f16_t>(f32x4& acc, const uint4& a, const uint4& b) {
    typedef f16_t f16x8 __attribute__((ext_vector_type(8)));
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// A fragment unit holds 4 consecutive k (fp32) of a row: exactly the A / B layout of v_mfma_f32_16x16x16_bf16 (lane = row, k = 4 (lane >> 4) + 0..3)
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_bf16(const uint4& u, s16x4& hi, s16x4& lo) {
    const float x0 = __uint_as_float(u.x), x1 = __uint_as_float(u.y), x2 = __uint_as_float(u.z), x3 = __uint_as_float(u.w);
    const unsigned h01 = TI<__bf16>::pack2(x0, x1), h23 = TI<__bf16>::pack2(x2, x3);
    const unsigned l01 = TI<__bf16>::pack2(x0 - __uint_as_float(h01 << 16), x1 - __uint_as_float(h01 & 0xffff0000u));
    const unsigned l23 = TI<__bf16>::pack2(x2 - __uint_as_float(h23 << 16), x3 - __uint_as_float(h23 & 0xffff0000u));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(s16x4, u32x2{h01, h23});
    lo = __builtin_bit_cast(s16x4, u32x2{l01, l23});
}
template <> __device__ __forceinline__ void mma16<f32x3_t>(f32x4& acc, const uint4& a, const uint4& b) {
    s16x4 ah, al, bh, bl;
    split_bf16(a, ah, al);                       // inlined per (i, j): the compiler keeps one split per fragment (common subexpressions)
    split_bf16(b, bh, bl);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
}

// The conv kernels multiply with the WEIGHT fragment as the MFMA's row operand and the pixel fragment as its column operand: the 16 x 16 result
// fragment is then [channel][pixel], i.e. a lane holds 4 CONSECUTIVE CHANNELS (4 (lane >> 4) + 0..3) of ONE pixel (lane & 15) -- 8 / 16 contiguous
// bytes of an NHWC row, stored straight from the accumulators (conv_epilogue_direct).  Both operands have the same register layout, so this costs
// nothing in the main loop.
template <typename T> __device__ __forceinline__ void mma16t(f32x4& acc, const uint4& pix, const uint4& wgt) { mma16<T>(acc, wgt, pix); }

template <int N> __device__ __forceinline__ float dpp_row_ror(float x) {      // lane l of each row of 16 lanes gets the value of lane (l + N) mod 16
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}

// ------------------------------------------------------------------------------------------------
// epilogue, fallback form: accumulators -> per-wave LDS tile (fp32) -> row-contiguous 16-byte global accesses (channel-major and fp32-NCHW outputs,
// channel counts that are not a multiple of 4).
// The MFMA C layout gives a lane ONE channel of 4 pixels; storing from it directly costs one 2-byte store per
// output (64 store instructions per lane, issue-bound).  Through LDS every lane owns 8 consecutive channels of
// one pixel: alpha*acc + bias + temb + residual in fp32, then one 16-byte store (and one 16-byte residual load).
// Every wave of the workgroup must call it (it contains workgroup barriers); waves with active == false (the
// producer waves of the specialised kernel) only take part in the barriers.  `wave` indexes the LDS tile.
// ------------------------------------------------------------------------------------------------
// Rows per GroupNorm-statistics slab.  It depends on the pixel tile only -- not on how many images a workgroup covers --
// so the partial sums (hence scale/shift, hence every output bit) are the same whatever batch an image sits in:
// 8x8 tiles always use half-image slabs (32 rows) whether one or two images share a workgroup.
__host__ __device__ constexpr int conv_stat_rows(int TH, int TW, int EROWS) { return (TH * TW == 64) ? (EROWS < 32 ? EROWS : 32) : EROWS; }

#ifndef WDM_EABL
#define WDM_EABL 0          // tools/dma_ablate.hip: 1 = no global stores of the output tile, 2 = return at once
#endif
// `write_pass(ep, jp)` puts the wave's accumulators of the 16-column fragments [jp, jp + NJ) into its fp32 tile ep[row][ESTR] (row = pixel of the wave
// tile in row-major order); it is the only part that knows the MFMA C layout (16x16 fragments below, 32x32 ones in conv_pp_kernel.h).
// `hook()` runs once, at the end of the first pass (every value that pass loaded -- bias, temb, residual -- has been consumed, its stores are issued):
// the persistent kernel issues the next tile's first DMAs there.  Earlier, the compiler's waits for the epilogue's own loads would also wait for the
// DMAs queued behind them (one in-order counter); the remaining passes and the statistics then run while the DMAs are in flight.
// entry_barrier = false: the caller has already closed the main loop with a workgroup barrier (and must not have its DMA queue drained by
// __syncthreads(), which waits vmcnt(0) while an LDS-DMA is pending).
struct EpiNoHook { __device__ __forceinline__ void operator()() const {} };
// one float4 of partial statistics.  When the launch finalises its consumer's GroupNorm itself (ConvArgs::fin_cnt, gn_arrive.h) another workgroup of the SAME
// kernel reads it: written through to memory (sc0 sc1) instead of into this XCD's write-back L2
template <class AT>
__device__ __forceinline__ void conv_store_stat(const AT& a, long long idx, const float4& v) {
    if (a.fin_cnt != nullptr) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a.stats, 0, 0x7FFFFFF0, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, (int)(idx * 16), 0, 17);
    } else {
        ((float4*)a.stats)[idx] = v;
    }
}
// CANON: the statistics of a slab are summed in ascending order of its 16-row chunks whatever NJ is -- the association of the one-pass form
// (NJ = WN = 4: one lane walks all rows of a column) -- so that a multi-pass epilogue writes the bits of the one-pass one.
template <typename T, int TH, int TW, int WM, int WN, int NJ_, class WritePass, class Hook = EpiNoHook, bool CANON = false, class AT = ConvArgs>
__device__ __forceinline__ void conv_epilogue_w(const AT& a, WritePass&& write_pass, char* smem, bool active, int wave, int lane, int wave_m,
                                                int wave_n, int img0, int oy0, int ox0, int n0, int tile_in_img, int phase = 0, Hook hook = Hook(),
                                                bool entry_barrier = true, bool pre_applied = false, float4* keep_tab = nullptr, int keep_bn = 0) {
    // keep_tab (LDS; tiles of whole images only): the in-tile GroupNorm of the output (gn_group.h: gn_out_tail) follows -- every pass gets its own LDS tile, so
    // that the final values (as stored: the statistics pass writes them back) are all still there afterwards, and the (pivot, s1, s2, n) of every
    // (image, slab, column) of the tile also goes into keep_tab[(image * slabs + slab) * keep_bn + column]
    // pre_applied: the tile already holds the FINAL values (alpha * acc + bias + temb, no residual) and the statistics are written (conv_epilogue's
    // direct path): rows are only rounded and stored
    constexpr int VEC = TI<T>::VEC;
    constexpr int ES = 16 / VEC;                           // bytes per element of T
    constexpr int NJ = NJ_ ? NJ_ : ((WN >= 2) ? 2 : 1);   // 16-column fragments per pass (NJ_ = WN: one pass, whole 128-byte rows per wave)
    constexpr int ECOLS = 16 * NJ;
    constexpr int ESTR = ECOLS + 4;                   // row stride (floats): 4*ESTR = 16 (mod 32) -> conflict-free writes
    constexpr int EROWS = 16 * WM;
    constexpr int LPR = ECOLS / 8;                    // lanes per row
    constexpr int RPI = 64 / LPR;                     // rows per iteration
    float* const ep0 = (float*)smem + wave * (EROWS * ESTR);
    const int keep_stride = keep_tab != nullptr ? (int)(blockDim.x >> 6) * (EROWS * ESTR) : 0;      // floats between the tiles of consecutive passes
    const bool vec_ok = (a.y_mode == Y_NHWC || a.y_mode == Y_NHWC_F32) && (a.Cout % 8 == 0);
    // output / residual through raw buffer descriptors (extents checked on the host: conv_dispatch.inc)
    const long long out_rows = a.up4 ? (long long)a.B * 4 * a.Hout * a.Wout : (a.m_valid ? (long long)a.m_valid : (long long)a.B * a.Hout * a.Wout);
    const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(unsigned)(out_rows * a.y_s * (a.y_mode == Y_NHWC ? ES : 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : a.y), 0, a.res ? (int)(unsigned)(out_rows * a.res_s * ES) : 0, 0x00020000);
    // additive per-channel terms of BOTH passes, fetched up front as 16-byte loads so that their latency overlaps the LDS
    // transposes instead of opening every pass: bias + shortcut bias + temb.  A wave tile lies inside one image
    // (static_assert), so temb's row is a per-wave constant.
    constexpr int NPASS = (WN + NJ - 1) / NJ;
    static_assert((TH * TW) % EROWS == 0 || EROWS % (TH * TW) == 0, "wave tile vs image geometry");
    float add8[NPASS][8];
    if (vec_ok && active && !pre_applied) {
        const int img_w = img0 + (wave_m * EROWS) / (TH * TW);
        const long long trow = (a.temb != nullptr && a.temb_per_image) ? (img_w < a.B ? img_w : a.B - 1) : 0;
        constexpr bool ONE_IMG = (TH * TW) % EROWS == 0;         // else (8x8 tiles, 128-row wave tiles) temb stays in the row loop
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int n = n0 + (wave_n * WN + ps * NJ) * 16 + (lane % LPR) * 8;
            float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            if (n < a.Cout) {          // Cout % 8 == 0 and n % 8 == 0: all eight channels exist
                if (a.bias != nullptr) { b0 = *(const float4*)(a.bias + n); b1 = *(const float4*)(a.bias + n + 4); }
                if (a.sbias != nullptr) {
                    const float4 c0 = *(const float4*)(a.sbias + n), c1 = *(const float4*)(a.sbias + n + 4);
                    b0.x += c0.x; b0.y += c0.y; b0.z += c0.z; b0.w += c0.w; b1.x += c1.x; b1.y += c1.y; b1.z += c1.z; b1.w += c1.w;
                }
                if (ONE_IMG && a.temb != nullptr) {
                    const float* tp = a.temb + trow * a.temb_ld + n;
                    const float4 c0 = *(const float4*)tp, c1 = *(const float4*)(tp + 4);
                    b0.x += c0.x; b0.y += c0.y; b0.z += c0.z; b0.w += c0.w; b1.x += c1.x; b1.y += c1.y; b1.z += c1.z; b1.w += c1.w;
                }
            }
            add8[ps][0] = b0.x; add8[ps][1] = b0.y; add8[ps][2] = b0.z; add8[ps][3] = b0.w;
            add8[ps][4] = b1.x; add8[ps][5] = b1.y; add8[ps][6] = b1.z; add8[ps][7] = b1.w;
        }
    }
    constexpr bool TEMB_IN_ADD = (TH * TW) % EROWS == 0;
#pragma unroll
    for (int jp = 0; jp < WN; jp += NJ) {
        float* const ep = ep0 + (jp / NJ) * keep_stride;
        // The fp32 tile is private to the wave, and the LDS executes one wave's instructions in order: only the hand-over from the
        // main loop (other waves may still read the operand images this tile overlays) needs the workgroup barrier.
        if (jp == 0) { WDM_ETS(1); if (entry_barrier) __syncthreads(); WDM_ETS(2); }
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        if (active) write_pass(ep, jp);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (jp == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); WDM_ETS(3); }
        if (!active) continue;
        const int ncol0 = n0 + (wave_n * WN + jp) * 16;          // first channel of this pass
        if (vec_ok) {
            // Addresses: pixel index = (wave-uniform part of iteration `it`) + (lane part, the same in every iteration) -- RPI and TW are powers of two,
            // so the lane's row offset never carries into the uniform part.  The uniform part goes into the buffer instructions' scalar offset: no
            // per-iteration vector integer math (the 64-bit index arithmetic of a plain pointer store was most of this loop's issue slots).
            const int c8 = (lane % LPR) * 8, lp = lane / LPR;
            const int n = ncol0 + c8;
            const bool ncol = n < a.Cout;
            constexpr unsigned OOBV = 0xFFFF0000u;                 // beyond every extent (extents stay below it: conv_dispatch.inc)
            const int es_y = a.y_mode == Y_NHWC ? ES : 4;
            const int lane_pix = !a.up4 ? (lp / TW) * a.Wout + (lp % TW) : ((lp / TW) * 2) * (2 * a.Wout) + (lp % TW) * 2;
            const unsigned vo_y = ncol ? (unsigned)((lane_pix * a.y_s + n) * es_y) : OOBV;
            const unsigned vo_r = ncol ? (unsigned)((lane_pix * a.res_s + n) * ES) : OOBV;
            float bias8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[e] = add8[jp / NJ][e];
            // GroupNorm partial statistics of the values as stored (optional): the final values go back into the LDS tile
            // and a column pass (lane = channel) sums them -- no cross-lane shuffles
            const bool do_stats = a.stats != nullptr && !pre_applied;
            constexpr int NIT = (EROWS + RPI - 1) / RPI;
            constexpr bool RAGGED = EROWS % RPI != 0;              // fewer rows than one iteration covers (16-row wave tiles): the lanes past them idle
            int so_pix[NIT];
            bool img_ok[NIT];
            int img_it[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {                    // wave-uniform (scalar) part
                const int mu = wave_m * EROWS + it * RPI;
                const int img_u = mu / (TH * TW), rru = mu % (TH * TW);
                const int oyu = oy0 + rru / TW, oxu = ox0 + rru % TW;
                const int img_g = img0 + img_u;
                img_it[it] = img_g;
                img_ok[it] = img_g < a.B;
                so_pix[it] = !a.up4 ? (img_g * a.Hout + oyu) * a.Wout + oxu
                                    : (img_g * (2 * a.Hout) + 2 * oyu + (phase >> 1)) * (2 * a.Wout) + 2 * oxu + (phase & 1);
            }
            // residual rows of the whole pass requested up front (bf16: one 16-byte load per iteration)
            uint4 resv[VEC == 8 ? NIT : 1];
            if (VEC == 8 && a.res != nullptr) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    unsigned vo = vo_r;
                    if (a.m_valid != 0 && so_pix[it] + lane_pix >= a.m_valid) vo = OOBV;
                    if (RAGGED && it * RPI + lp >= EROWS) vo = OOBV;
                    resv[it] = img_ok[it] ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)vo, (int)((unsigned)so_pix[it] * (unsigned)a.res_s * (unsigned)ES), 0)) : make_uint4(0u, 0u, 0u, 0u);      // unsigned: tensors up to 3.75 GB (conv_dispatch.inc)
                }
            }
            // the lane's rows of the tile, all requested before the first is used (one LDS round trip for the pass instead of one per iteration)
            float4 tl[NIT][2];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int rloc = it * RPI + lp;
                const bool row_ok = !RAGGED || rloc < EROWS;
                const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                tl[it][0] = row_ok ? *(const float4*)(ep + rloc * ESTR + c8) : zero4;
                tl[it][1] = row_ok ? *(const float4*)(ep + rloc * ESTR + c8 + 4) : zero4;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (!img_ok[it]) continue;                         // wave-uniform: images past the batch (their statistics slabs are never written)
                const int rloc = it * RPI + lp;
                unsigned voy = vo_y, vor = vo_r;
                if (a.m_valid != 0 && so_pix[it] + lane_pix >= a.m_valid) { voy = OOBV; vor = OOBV; }
                const bool row_ok = !RAGGED || rloc < EROWS;
                if (!row_ok) { voy = OOBV; vor = OOBV; }
                const float4 v0 = tl[it][0], v1 = tl[it][1];
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (!pre_applied) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], a.alpha, bias8[e]);       // explicit: the direct path (conv_epilogue) computes the same
                }
                if (!pre_applied && !TEMB_IN_ADD && a.temb != nullptr && ncol) {
                    const float* tp = a.temb + (long long)(a.temb_per_image ? img_it[it] : 0) * a.temb_ld + n;
                    const float4 t0 = *(const float4*)tp, t1 = *(const float4*)(tp + 4);
                    v[0] += t0.x; v[1] += t0.y; v[2] += t0.z; v[3] += t0.w; v[4] += t1.x; v[5] += t1.y; v[6] += t1.z; v[7] += t1.w;
                }
                if (a.res != nullptr) {
                    float rf[8];
                    if (VEC == 8) TI<T>::unpack(resv[it], rf);
                    else {
                        const int so = (int)((unsigned)so_pix[it] * (unsigned)a.res_s * (unsigned)ES);
                        TI<T>::unpack(__builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)vor, so, 0)), rf);
                        TI<T>::unpack(__builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)vor + 16, so, 0)), rf + 4);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rf[e];
                }
                float vr[8];                                   // the values as the consumer will read them back
                // The uniform part is added to the lane's offset instead of riding in the store's scalar-offset field: a 16-byte buffer store with an
                // SGPR soffset was seen (gfx950, fp32 tiles) to pick up a data register that the NEXT VALU instruction overwrote -- the compiler
                // inserts the wait state for that hazard only when soffset is not a register.  (The loads above have no data operand to race on.)
                const unsigned vst = voy == OOBV ? OOBV : voy + (unsigned)so_pix[it] * (unsigned)a.y_s * (unsigned)es_y;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                if (a.y_mode == Y_NHWC && VEC == 8) {
                    const uint4 pk = TI<T>::pack(v);
                    TI<T>::unpack(pk, vr);
                    if (!(WDM_EABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk), r_y, (int)vst, 0, WDM_STORE_AUX);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vr[e] = v[e];
                    if (!(WDM_EABL & 1)) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), r_y, (int)vst, 0, WDM_STORE_AUX);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[4], v[5], v[6], v[7])), r_y, (int)vst + 16, 0, WDM_STORE_AUX);
                    }
                }
                if (do_stats && row_ok) {
                    *(float4*)(ep + rloc * ESTR + c8) = make_float4(vr[0], vr[1], vr[2], vr[3]);
                    *(float4*)(ep + rloc * ESTR + c8 + 4) = make_float4(vr[4], vr[5], vr[6], vr[7]);
                }
            }
            WDM_ETS(4);
            if (jp == 0) hook();
            if (do_stats) {
                // wave-local: the LDS executes one wave's instructions in order, so the column reads below see the
                // write-backs above; the fence only stops the compiler from reordering them
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                constexpr int SROWS = conv_stat_rows(TH, TW, EROWS);     // rows per statistics slab
                constexpr int PARTS = 64 / ECOLS;                         // lane groups splitting the rows
                constexpr int RPP = EROWS / PARTS;                        // rows per lane
                static_assert(SROWS % RPP == 0 && EROWS % SROWS == 0, "statistics slab geometry");
                const int col = lane % ECOLS, part = lane / ECOLS;
                const int r0 = part * RPP;
                const int srow = (r0 / SROWS) * SROWS;                    // first row of this lane's slab
                const float K = ep[srow * ESTR + col];
                // fixed association: 16-row chunks summed in row order, chunks added in ascending order, then lane groups --
                // identical whether a slab's 32 rows sit in one lane (two images per workgroup) or in two (one image)
                float s1 = 0.f, s2 = 0.f;
                constexpr int CH = RPP < 16 ? RPP : 16;
                static_assert(RPP % CH == 0, "statistics rows per lane must be a multiple of the chunk");
                constexpr int NCH = RPP / CH;
                float cs1[NCH], cs2[NCH];
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    float c1 = 0.f, c2 = 0.f;
#pragma unroll
                    for (int r = 0; r < CH; ++r) { const float d = ep[(r0 + ch * CH + r) * ESTR + col] - K; c1 += d; c2 = __builtin_fmaf(d, d, c2); }      // explicit fma: every tiling rounds alike
                    cs1[ch] = c1; cs2[ch] = c2;
                    if (ch == 0) { s1 = c1; s2 = c2; } else { s1 += c1; s2 += c2; }
                }
                if (CANON && SROWS / RPP > 1) {
                    // every lane walks the slab's chunks in row order, fetching the other lane groups' chunk sums (K is the slab's, the same in all of them)
                    constexpr int GPS = SROWS / RPP;                          // lane groups per slab
                    const int base = (part / GPS) * GPS;
#pragma unroll
                    for (int pg = 0; pg < GPS; ++pg)
#pragma unroll
                        for (int ch = 0; ch < NCH; ++ch) {
                            const float v1 = __shfl(cs1[ch], col + ECOLS * (base + pg)), v2 = __shfl(cs2[ch], col + ECOLS * (base + pg));
                            if (pg == 0 && ch == 0) { s1 = v1; s2 = v2; } else { s1 += v1; s2 += v2; }
                        }
                } else {
#pragma unroll
                    for (int off = ECOLS; off < ECOLS * (SROWS / RPP); off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
                }
                // all rows of a wave tile belong to one image
                const int m0 = wave_m * EROWS + srow;
                const int img_g = img0 + m0 / (TH * TW);
                constexpr int SPT = (TH * TW) / SROWS;                    // slabs per tile per image
                const int slab = tile_in_img * SPT + (m0 % (TH * TW)) / SROWS + (a.up4 ? phase * (a.stats_nslab >> 2) : 0);
                const int nn = ncol0 + col;
                if ((r0 % SROWS) == 0 && nn < a.Cout && img_g < a.B)
                    conv_store_stat(a, ((long long)img_g * a.stats_nslab + slab) * a.Cout + nn, make_float4(K, s1, s2, (float)SROWS));
                if (keep_tab != nullptr && (r0 % SROWS) == 0 && nn < a.Cout)
                    keep_tab[((m0 / (TH * TW)) * SPT + (m0 % (TH * TW)) / SROWS) * keep_bn + (nn - n0)] = make_float4(K, s1, s2, (float)SROWS);
                if (a.gst != nullptr) {
                    // group-level partials: the gs = Cout / 32 channels of a group are gs consecutive lanes (gs divides the pass's columns: host check);
                    // re-centred on the group's first channel and summed by a fixed xor tree, explicit fma (every instantiation rounds alike)
                    const int gs = a.Cout >> 5;
                    const float Kg = __shfl(K, lane & ~(gs - 1));
                    const float d = K - Kg, nr = (float)SROWS;
                    float g1 = __builtin_fmaf(nr, d, s1);
                    float g2 = __builtin_fmaf(nr * d, d, __builtin_fmaf(2.0f * d, s1, s2));
                    for (int off = 1; off < gs; off <<= 1) { g1 += __shfl_xor(g1, off); g2 += __shfl_xor(g2, off); }
                    if ((r0 % SROWS) == 0 && (lane & (gs - 1)) == 0 && nn < a.Cout && img_g < a.B) {
                        float* q = a.gst + (((long long)img_g * a.stats_nslab + slab) * 32 + nn / gs) * 3;
                        q[0] = Kg; q[1] = g1; q[2] = g2;
                    }
                }
            }
        } else {
            // channel-major (NCHW) outputs and odd channel counts: lane = pixel row, loop over channels, so that
            // for each channel the 64 lanes write runs of consecutive pixels
#pragma unroll 1
            for (int it = 0; it < (EROWS + 63) / 64; ++it) {
                const int rloc = it * 64 + lane;
                if (rloc >= EROWS) continue;
                const int m = wave_m * EROWS + rloc;
                const int img = m / (TH * TW), rr = m % (TH * TW);
                const int oy = oy0 + rr / TW, ox = ox0 + rr % TW;
                const int img_g = img0 + img;
                if (img_g >= a.B) continue;
                const long long opix = ((long long)img_g * a.Hout + oy) * a.Wout + ox;
                if (a.m_valid != 0 && opix >= a.m_valid) continue;
#pragma unroll 4
                for (int c = 0; c < ECOLS; ++c) {
                    const int n = ncol0 + c;
                    if (n >= a.Cout) break;
                    float v = ep[rloc * ESTR + c] * a.alpha + (a.bias != nullptr ? a.bias[n] : 0.f) + (a.sbias != nullptr ? a.sbias[n] : 0.f);
                    if (a.temb != nullptr) v += a.temb[(long long)(a.temb_per_image ? img_g : 0) * a.temb_ld + n];
                    if (a.res != nullptr) v += TI<T>::ld(a.res, opix * a.res_s + n);
                    if (a.y_mode == Y_NHWC) TI<T>::st(a.y, opix * a.y_s + n, v);
                    else if (a.y_mode == Y_NHWC_F32) ((float*)a.y)[opix * a.y_s + n] = v;
                    else {
                        const long long o = (((long long)img_g * a.Cout + n) * a.Hout + oy) * a.Wout + ox;
                        if (a.y_mode == Y_NCHW) TI<T>::st(a.y, o, v);
                        else ((float*)a.y)[o] = v;
                    }
                }
            }
            if (jp == 0) hook();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// epilogue, packed form: bf16 NHWC outputs WITHOUT a residual operand (conv1 of every ResnetBlock, conv2 with the fused 1x1 shortcut, conv_in, Up /
// Downsample) on 16-wide pixel tiles, wave tile = 64 rows x 64 columns per pass.
// The fp32 form above moves every output through LDS three times in fp32 (accumulators in: 16 ds_write_b128 per lane, the rounded values back for the
// statistics: 16 more, the column pass) -- and ds_write_b128 runs at ~79 B/clk/CU: ~3.3 k of a 256 x 128 tile's ~8 k epilogue ticks are those two store
// phases.  Without a residual nothing needs the row layout in fp32: alpha * acc + (bias + shortcut bias + temb) is formed and rounded in the ACCUMULATOR
// layout (a lane holds 4 consecutive channels of a pixel: the additive terms are 4 floats per fragment column), the tile goes to LDS once, as bf16 (8 KB per
// wave instead of 17), the row loop only moves 16-byte units LDS -> HBM, and the column pass (lane = channel, rows in the same chunk order) reads the
// rounded values -- the same operations on the same numbers as the fp32 form, so outputs and statistics are bit-identical to it.
// LDS tile: row r = 128 bytes = eight 16-byte units; unit u sits in slot u ^ (r & 7) and its two 8-byte halves are swapped when r & 8 -- the 16 pixel
// lanes of a ds_write_b64 then cover all 32 banks, the 2 x 8 lanes of a ds_read_b128 service group cover all 64, and the swap is a compile-time register
// order in the row loop (rows of iteration `it` have (r >> 3) & 1 == it & 1).
// `hook()` runs when the row loop's stores are issued (first pass only).
// ------------------------------------------------------------------------------------------------
template <class AT>
__host__ __device__ __forceinline__ bool conv_epilogue_can_pack(const AT& a) {
#ifdef WDM_NO_PACK          // tools: A/B against the fp32 form
    return false;
#else
    return a.y_mode == Y_NHWC && (a.Cout % 8) == 0 && a.m_valid == 0 && !(a.up4 && a.res != nullptr);      // (round 5: a residual operand no longer forces the fp32 tile)
#endif
}
constexpr int EPI_PACK_TILE = 64 * 128;      // LDS bytes per wave
template <typename T, int TH, int TW, int WN, class Hook = EpiNoHook, class AT = ConvArgs>
__device__ __forceinline__ void conv_epilogue_packed(const AT& a, f32x4 (&acc)[4][WN], int jp, char* smem, int wave, int lane_, int wave_m, int wave_n, int img0, int oy0,
                                                     int ox0, int n0, int tile_in_img, int phase, Hook hook, bool call_hook, float4* keep_tab = nullptr, int keep_bn = 0) {
    constexpr int EROWS = 64, ROWB = 128;
    int lane = lane_;
    asm volatile("" : "+v"(lane));             // every address below is a function of the lane: formed HERE, not hoisted above the K loop (where ~40 of them would
                                               // stay live through it and spill the accumulators of the 256-column tiles)
    char* const tp = smem + wave * EPI_PACK_TILE;
    const int quad = lane >> 4, px = lane & 15;
    const int ncol0 = n0 + (wave_n * WN + jp) * 16;            // first channel of this pass
    if (jp != 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }       // the previous pass's column reads (same wave: LDS in order)
    __builtin_amdgcn_sched_barrier(0);            // a pass's loads stay inside the pass (hoisted across passes they spill the accumulators of 256-column tiles)
    // ---- residual operand (round 5): its rows come in like the output rows go out (a lane moves 16 bytes = 8 channels of a pixel, 8 rows per iteration) and are
    // parked in the tile at the very place the final values of those outputs will sit, so that the accumulator-layout pass below finds the four residual values of
    // a lane as ONE 8-byte read and puts alpha * acc + (bias + temb) + residual back in their place -- the same operations on the same numbers as conv_epilogue_w
    // (fma, + residual, one rounding), hence the same bits, without the fp32 tile's three trips through LDS.
    const bool has_res = a.res != nullptr;
    if (has_res) {
        constexpr unsigned OOBV = 0xFFFF0000u;
        const long long out_rows = (long long)a.B * a.Hout * a.Wout;                   // (up4 / m_valid launches have no residual: conv_dispatch.inc)
        const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)a.res, 0, (int)(unsigned)(out_rows * a.res_s * 2), 0x00020000);
        const int c8 = (lane & 7) * 8, lp = lane >> 3;
        const int n = ncol0 + c8;
        const unsigned vo_r = n < a.Cout ? (unsigned)((lp * a.res_s + n) * 2) : OOBV;
        const int rd = lp * ROWB + (((lane & 7) ^ lp) << 4);
        uint4 rv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int mu = wave_m * EROWS + it * 8;                     // wave-uniform
            const int img_u = mu / (TH * TW), rru = mu % (TH * TW);
            const int oyu = oy0 + rru / TW, oxu = ox0 + rru % TW;
            const int img_g = img0 + img_u;
            const int so_pix = (img_g * a.Hout + oyu) * a.Wout + oxu;
            rv[it] = img_g < a.B ? __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r_res, (int)vo_r, (int)((unsigned)so_pix * (unsigned)a.res_s * 2u), 0)) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) *(uint4*)(tp + rd + it * (8 * ROWB)) = (it & 1) ? make_uint4(rv[it].z, rv[it].w, rv[it].x, rv[it].y) : rv[it];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- final values in the accumulator layout -> bf16 tile
    {
        const int img_w = img0 + (wave_m * EROWS) / (TH * TW);
        const long long trow = (a.temb != nullptr && a.temb_per_image) ? (img_w < a.B ? img_w : a.B - 1) : 0;
        const int wr0 = px * ROWB + (((quad & 1) ^ (px >> 3)) << 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = ncol0 + j * 16 + quad * 4;
            float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < a.Cout) {                 // the order of conv_epilogue_w: bias, + shortcut bias, + temb
                if (a.bias != nullptr) ad = *(const float4*)(a.bias + n);
                if (a.sbias != nullptr) { const float4 c = *(const float4*)(a.sbias + n); ad.x += c.x; ad.y += c.y; ad.z += c.z; ad.w += c.w; }
                if (a.temb != nullptr) { const float4 c = *(const float4*)(a.temb + trow * a.temb_ld + n); ad.x += c.x; ad.y += c.y; ad.z += c.z; ad.w += c.w; }
            }
            const int wr = wr0 + (((j * 2 + (quad >> 1)) ^ (px & 7)) << 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = acc[i][jp + j];
                float f0 = __builtin_fmaf(v[0], a.alpha, ad.x), f1 = __builtin_fmaf(v[1], a.alpha, ad.y), f2 = __builtin_fmaf(v[2], a.alpha, ad.z), f3 = __builtin_fmaf(v[3], a.alpha, ad.w);
                if (has_res) {
                    const uint2 rr = *(const uint2*)(tp + wr + i * (16 * ROWB));
                    float r0, r1, r2, r3;
                    TI<T>::unpack2(rr.x, r0, r1); TI<T>::unpack2(rr.y, r2, r3);
                    f0 += r0; f1 += r1; f2 += r2; f3 += r3;
                }
                uint2 pk;
                pk.x = TI<T>::pack2(f0, f1);
                pk.y = TI<T>::pack2(f2, f3);
                *(uint2*)(tp + wr + i * (16 * ROWB)) = pk;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- row loop: a lane moves 8 consecutive channels (one 16-byte unit) of a pixel, 8 rows per iteration (conv_epilogue_w's addresses for NJ = 4)
    {
        constexpr unsigned OOBV = 0xFFFF0000u;
        const long long out_rows = a.up4 ? (long long)a.B * 4 * a.Hout * a.Wout : (long long)a.B * a.Hout * a.Wout;
        const __amdgpu_buffer_rsrc_t r_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(unsigned)(out_rows * a.y_s * 2), 0x00020000);
        const int c8 = (lane & 7) * 8, lp = lane >> 3;
        const int n = ncol0 + c8;
        const int lane_pix = !a.up4 ? lp : lp * 2;                      // lp < 8 <= TW: the lane's row of an iteration lies in one image row
        const unsigned vo_y = n < a.Cout ? (unsigned)((lane_pix * a.y_s + n) * 2) : OOBV;
        const int rd = lp * ROWB + (((lane & 7) ^ lp) << 4);
        uint4 tl[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) tl[it] = *(const uint4*)(tp + rd + it * (8 * ROWB));
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int mu = wave_m * EROWS + it * 8;                     // wave-uniform
            const int img_u = mu / (TH * TW), rru = mu % (TH * TW);
            const int oyu = oy0 + rru / TW, oxu = ox0 + rru % TW;
            const int img_g = img0 + img_u;
            if (img_g >= a.B) continue;
            const int so_pix = !a.up4 ? (img_g * a.Hout + oyu) * a.Wout + oxu
                                      : (img_g * (2 * a.Hout) + 2 * oyu + (phase >> 1)) * (2 * a.Wout) + 2 * oxu + (phase & 1);
            const unsigned vst = vo_y == OOBV ? OOBV : vo_y + (unsigned)so_pix * (unsigned)a.y_s * 2u;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 pk = (it & 1) ? u32x4{tl[it].z, tl[it].w, tl[it].x, tl[it].y} : u32x4{tl[it].x, tl[it].y, tl[it].z, tl[it].w};
            if (!(WDM_EABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(pk, r_y, (int)vst, 0, WDM_STORE_AUX);
        }
    }
    if (call_hook) hook();
    // ---- GroupNorm partial statistics of the values as stored: lane = channel, 64 rows in four 16-row chunks (the association of conv_epilogue_w's
    // one-pass form); group-level partials as there
    if (a.stats != nullptr) {
        const int col = lane, u = col >> 3, e = col & 7;
        int va[2][8];                                   // [(r >> 3) & 1][r & 7]: byte offset of this channel inside row r
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 8; ++k) va[h][k] = ((u ^ k) << 4) | ((((e >> 2) ^ h)) << 3) | ((e & 3) << 1);
        auto val = [&](int r) __attribute__((always_inline)) {
            return TI<T>::raw16_to_f32(*(const unsigned short*)(tp + r * ROWB + va[(r >> 3) & 1][r & 7]));
        };
        const float K = val(0);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = val(ch * 16 + r) - K; c1 += d; c2 = __builtin_fmaf(d, d, c2); }
            if (ch == 0) { s1 = c1; s2 = c2; } else { s1 += c1; s2 += c2; }
            __builtin_amdgcn_sched_barrier(0);          // one chunk's 16 reads in flight at a time (all 64 hoisted cost more registers than the kernels have)
        }
        const int m0 = wave_m * EROWS;
        const int img_g = img0 + m0 / (TH * TW);
        constexpr int SPT = (TH * TW) / 64;
        const int slab = tile_in_img * SPT + (m0 % (TH * TW)) / 64 + (a.up4 ? phase * (a.stats_nslab >> 2) : 0);
        const int nn = ncol0 + col;
        if (nn < a.Cout && img_g < a.B) conv_store_stat(a, ((long long)img_g * a.stats_nslab + slab) * a.Cout + nn, make_float4(K, s1, s2, 64.f));
        // (in-tile GroupNorm of the output, gn_group.h: gn_out_tail_packed -- the tile's own table of these partials, as conv_epilogue_w keeps it)
        if (keep_tab != nullptr && nn < a.Cout) keep_tab[((m0 / (TH * TW)) * SPT + (m0 % (TH * TW)) / 64) * keep_bn + (nn - n0)] = make_float4(K, s1, s2, 64.f);
        if (a.gst != nullptr) {
            const int gs = a.Cout >> 5;
            const float Kg = __shfl(K, lane & ~(gs - 1));
            const float d = K - Kg, nr = 64.f;
            float g1 = __builtin_fmaf(nr, d, s1);
            float g2 = __builtin_fmaf(nr * d, d, __builtin_fmaf(2.0f * d, s1, s2));
            for (int off = 1; off < gs; off <<= 1) { g1 += __shfl_xor(g1, off); g2 += __shfl_xor(g2, off); }
            if ((lane & (gs - 1)) == 0 && nn < a.Cout && img_g < a.B) {
                float* q = a.gst + (((long long)img_g * a.stats_nslab + slab) * 32 + nn / gs) * 3;
                q[0] = Kg; q[1] = g1; q[2] = g2;
            }
        }
    }
}

template <typename T, int TH, int TW, int WM, int WN, int NJ_ = 0, class Hook = EpiNoHook, bool CANON = false, int PACK = 0, class AT = ConvArgs>
__device__ __forceinline__ void conv_epilogue(const AT& a, f32x4 (&acc)[WM][WN], char* smem, bool active, int wave, int lane, int wave_m,
                                              int wave_n, int img0, int oy0, int ox0, int n0, int tile_in_img, int phase = 0, Hook hook = Hook(),
                                              bool entry_barrier = true, float4* keep_tab = nullptr, int keep_bn = 0) {
    if (WDM_EABL & 2) { float t = 0.f; for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; if (t == 123.456f) ((float*)a.y)[0] = t; return; }
    constexpr int NJ = NJ_ ? NJ_ : ((WN >= 2) ? 2 : 1);
    constexpr int ESTR = 16 * NJ + 4;
    // (Round 3's "direct" form -- final values and GroupNorm statistics taken in the accumulator layout, 16-lane DPP reductions per channel -- measured
    // 3.5 % slower end to end and is gone; conv_epilogue_packed below keeps the column pass and drops the fp32 round trips instead.)
    // PACK = 1: the packed form when the arguments allow it (run-time test), 2: always (the launcher has tested: no fp32 form in the kernel at all)
    if constexpr (PACK != 0) {
        static_assert(TI<T>::VEC == 8 && WM == 4 && (WN % 4) == 0 && TW == 16 && (TH * TW) % 64 == 0, "packed epilogue: 16-bit tiles, 64-row wave tiles of a 16-wide pixel tile");
        if (PACK == 2 || (keep_tab == nullptr && conv_epilogue_can_pack(a))) {
            if (entry_barrier) __syncthreads();
#pragma unroll
            for (int jp = 0; jp < WN; jp += 4)
                conv_epilogue_packed<T, TH, TW, WN, Hook, AT>(a, acc, jp, smem, wave, lane, wave_m, wave_n, img0, oy0, ox0, n0, tile_in_img, phase, hook, jp == 0, keep_tab, keep_bn);
            return;
        }
    }
    auto write_pass = [&](float* ep, int jp) __attribute__((always_inline)) {      // [channel][pixel] fragments -> ep[pixel][channel]
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
            for (int i = 0; i < WM; ++i)
                *(float4*)(ep + (i * 16 + (lane & 15)) * ESTR + jj * 16 + (lane >> 4) * 4) = make_float4(acc[i][jp + jj][0], acc[i][jp + jj][1], acc[i][jp + jj][2], acc[i][jp + jj][3]);
    };
    conv_epilogue_w<T, TH, TW, WM, WN, NJ_, decltype(write_pass)&, Hook, CANON>(a, write_pass, smem, active, wave, lane, wave_m, wave_n, img0, oy0, ox0, n0, tile_in_img, phase, hook, entry_barrier,
                                                                                false, keep_tab, keep_bn);
    WDM_ETS(5);
#ifdef WDM_EPI_TS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WDM_ETS(6);
#endif
}

// ------------------------------------------------------------------------------------------------
// compile-time geometry of one kernel configuration
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, int TH, int TW, int NI, int WAVES_M, int WAVES_N, int WM, int WN, int KSUB = 4>
struct ConvCfg {
    static constexpr int NWAVES = WAVES_M * WAVES_N;   // 4 (two workgroups per CU) or 8 (one workgroup per CU)
    static constexpr int NTHREADS = 64 * NWAVES;
    static constexpr int VEC = TI<T>::VEC;
    static constexpr int NU = 4;                 // 16-byte k-units per slab
    static constexpr int BK = NU * VEC;          // channels per slab: 32 (bf16) / 16 (f32)
    static constexpr int M = TH * TW * NI;
    static constexpr int BN = 16 * WN * WAVES_N;
    static constexpr int NSUB = (MODE == MODE_P1) ? KSUB : 9;   // B sub-blocks per stage (taps, or slabs for 1x1)
    static constexpr int NSUBA = (MODE == MODE_P1) ? KSUB : 1;  // A sub-planes per stage
    static constexpr int PH = MODE == MODE_S1 ? TH + 2 : MODE == MODE_S2 ? 2 * TH + 1 : MODE == MODE_UPS ? TH / 2 + 2 : TH;
    static constexpr int PW = MODE == MODE_S1 ? TW + 2 : MODE == MODE_S2 ? 2 * TW + 1 : MODE == MODE_UPS ? TW / 2 + 2 : TW;
    // row stride in pixel slots, a multiple of 8: (a) a tap's dy*RS never changes the rotation bit (q>>2)&1, so
    // only the three dx variants of a fragment address are kept in registers; (b) for 8-wide tiles a 16-row MFMA
    // group spans two image rows RS apart and stays conflict-free
    static constexpr int RS = (PW + 7) / 8 * 8;
    static constexpr int NPIX = PH * PW;                      // pixels actually staged per image
    static constexpr int PLANE_IMG = PH * RS;                 // pixel slots per image
    static constexpr int PLANE = NI * PLANE_IMG;              // pixel slots per A sub-plane
    static constexpr int A_BYTES = NSUBA * PLANE * 64;
    static constexpr int B_BYTES = NSUB * BN * 64;
    static constexpr int LDS_BYTES = A_BYTES + B_BYTES;
    static constexpr int A_IPI = (NSUBA * NPIX * NU + NTHREADS - 1) / NTHREADS;   // A items per thread per image
    static constexpr int B_IPT = (NSUB * BN * NU + NTHREADS - 1) / NTHREADS;      // B items per thread
    static constexpr bool PREFETCH = (MODE != MODE_S2);   // S2 tiles stage 4x the pixels: keep registers low
    // two workgroups per CU (one's LDS-fill phase overlaps the other's MFMA phase) need <= 256 registers per lane
    static constexpr int MIN_WAVES = (NWAVES == 8 || LDS_BYTES <= 80 * 1024) ? 2 : 1;
    static_assert(M == 16 * WM * WAVES_M, "tile M mismatch");
    static_assert(NWAVES == 4 || NWAVES == 8, "4 or 8 waves per workgroup");
    static_assert(NI == 1 || (MODE == MODE_S1 || MODE == MODE_P1), "multi-image tiles: s1 / 1x1 only");
};

// byte offset of unit u of row slot q inside an operand image (see the header comment)
__device__ __forceinline__ int lds_off(int q, int u) { return (q << 6) | ((u ^ ((q >> 1) & 2)) << 4); }

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, int TH, int TW, int NI, int WAVES_M, int WAVES_N, int WM, int WN, int KSUB = 4>
__global__ __launch_bounds__((ConvCfg<T, MODE, TH, TW, NI, WAVES_M, WAVES_N, WM, WN, KSUB>::NTHREADS),
                             (ConvCfg<T, MODE, TH, TW, NI, WAVES_M, WAVES_N, WM, WN, KSUB>::MIN_WAVES)) void conv_kernel(const ConvArgs a) {
    using C = ConvCfg<T, MODE, TH, TW, NI, WAVES_M, WAVES_N, WM, WN, KSUB>;
    constexpr int VEC = C::VEC, NU = C::NU, BK = C::BK, BN = C::BN;
    constexpr int NSUB = C::NSUB, NSUBA = C::NSUBA, PW = C::PW, RS = C::RS, NPIX = C::NPIX;
    constexpr int PLANE = C::PLANE, PLANE_IMG = C::PLANE_IMG, A_BYTES = C::A_BYTES;
    constexpr int A_IPI = C::A_IPI, B_IPT = C::B_IPT;
    constexpr int NDX = (MODE == MODE_S1 || MODE == MODE_S2) ? 3 : 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    h16_mode_init<T>();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    // ---- workgroup -> (M tile, N tile), XCD-aware (block id % 8 is the XCD the dispatcher picks; used for speed only).
    // The 8 XCDs are split into gn N-groups x (8/gn) M-groups; XCD (xm, xn) owns the tiles mt = xm (mod gm), nt = xn (mod gn)
    // and walks them M-fastest, so the ~64 workgroups resident on an XCD at any time share ONE (or few) weight tiles --
    // which then stay in that XCD's 4 MB L2 -- while each A tile is streamed once per N tile of the XCD.  gn = 1 is the
    // plain "N tiles of one M tile back to back" order used when the whole weight tensor fits in L2 anyway.
    const int bid = blockIdx.x;
    int mt, nt;
    if (!conv_decode_tile(a, bid, mt, nt)) return;
    const int n0 = nt * BN;

    int img0, oy0, ox0, tile_in_img = 0;
    if (NI == 1) conv_decode_image<TH, TW>(a, mt, img0, tile_in_img, oy0, ox0);
    else { img0 = mt * NI; oy0 = 0; ox0 = 0; }
    // origin of the staged input region in source coordinates
    const int iy0 = MODE == MODE_S1 ? oy0 - 1 : MODE == MODE_S2 ? 2 * oy0 : MODE == MODE_UPS ? (oy0 >> 1) - 1 : oy0;
    const int ix0 = MODE == MODE_S1 ? ox0 - 1 : MODE == MODE_S2 ? 2 * ox0 : MODE == MODE_UPS ? (ox0 >> 1) - 1 : ox0;

    // ---- per-lane fragment addresses (bytes into smem).  A: one address per dx (the rotation bit depends on the
    // row slot), dy and the sub-plane are immediate offsets.
    const int ku = lane >> 4;                  // k-unit this lane feeds to the MFMA
    int a_addr[WM][NDX];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = (wave_m * WM + i) * 16 + (lane & 15);
        const int img = m / (TH * TW), r = m % (TH * TW);
        const int ly = r / TW, lx = r % TW;
        if (MODE == MODE_UPS) {
            a_addr[i][0] = (ly << 8) | lx;     // resolved per tap in compute_stage
        } else {
            const int q0 = img * PLANE_IMG + (MODE == MODE_S2 ? 2 * ly * RS + 2 * lx : ly * RS + lx);
#pragma unroll
            for (int dx = 0; dx < NDX; ++dx) a_addr[i][dx] = lds_off(q0 + dx, ku);
        }
    }
    int b_addr[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) b_addr[j] = A_BYTES + lds_off((wave_n * WN + j) * 16 + (lane & 15), ku);

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging registers
    uint4 ra[NI][A_IPI];
    uint4 rb[B_IPT];
    float sc[NI][VEC], sh[NI][VEC];
    static_assert(NI * A_IPI <= 32, "inb_mask too small");

    const int unit = tid & (NU - 1);
    const int nchunks = a.Cin / BK;                                   // slabs
    const int nstages = (MODE == MODE_P1) ? (nchunks + NSUB - 1) / NSUB : nchunks;
    constexpr int ES = (int)sizeof(T);
    constexpr int RPI_ = C::NTHREADS / NU;                            // rows (pixels / weight rows) covered per item index

    // ---- per-item geometry, computed ONCE.  All global traffic of the main loop goes through buffer loads whose
    // per-lane byte offset (voffset) is loop-invariant; the channel-slab / tap advance is a wave-uniform SGPR
    // offset.  Out-of-image halo pixels (and everything past the end of a tensor) carry an out-of-range voffset:
    // the hardware bounds check returns zeros, so there are no branches and no address arithmetic in the loop.
    constexpr unsigned OOB = 0xFFFF0000u;      // >= num_records of any tensor (host asserts tensors < 4 GB - 64 KB)
    const __amdgpu_buffer_rsrc_t r_x0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x0, 0, a.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x1 ? a.x1 : a.x0), 0, a.x1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)a.w + conv_w_img_offset(a, img0)), 0,
                                                                        a.w_bytes, 0x00020000);
    unsigned a_v0[NI][A_IPI], a_v1[NI][A_IPI];   // byte offset of (pixel, unit) in x0 / x1
    int a_l[NI][A_IPI];                          // LDS byte offset of the item (sub-plane included)
    unsigned inb_mask = 0;                       // bit (img*A_IPI + i): item lies inside the image (transform applies)
#pragma unroll
    for (int im = 0; im < NI; ++im) {
        const int img_g = img0 + im;
#pragma unroll
        for (int i = 0; i < A_IPI; ++i) {
            const int pq = (tid >> 2) + i * RPI_;
            const int sub = (NSUBA == 1) ? 0 : pq / NPIX;
            const int q = (NSUBA == 1) ? pq : pq - sub * NPIX;
            const int hy = q / PW, hx = q - hy * PW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = (pq < NSUBA * NPIX) && (img_g < a.B) && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const unsigned gp = (unsigned)((conv_x_img(a, img_g) * a.Hin + iy) * a.Win + ix);
            a_v0[im][i] = ok ? gp * (unsigned)(a.xs0 * ES) + (unsigned)(unit * 16) : OOB;
            a_v1[im][i] = ok ? gp * (unsigned)(a.xs1 * ES) + (unsigned)(unit * 16) : OOB;
            a_l[im][i] = sub * (PLANE * 64) + lds_off(im * PLANE_IMG + hy * RS + hx, unit);
            if (ok) inb_mask |= 1u << (im * A_IPI + i);
        }
    }
    // B items: row index rn = tid/4 + 64 i walks the [sub][n] rows of the stage linearly, so both the LDS offset and the
    // global offset are (per-thread base) + (uniform function of i)
    const int rb0 = tid >> 2;
    const int b_l0 = A_BYTES + lds_off(rb0, unit);                    // + i * RPI_ * 64
    unsigned b_v0;
    if (BN >= RPI_) b_v0 = (unsigned)((n0 + rb0) * a.w_row_stride * ES + unit * 16);
    else b_v0 = (unsigned)(((n0 + rb0 % BN) * (long long)a.w_row_stride + (rb0 / BN) * ((MODE == MODE_P1) ? (long long)BK : a.w_tap_stride)) * ES + unit * 16);

    auto load16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff) __attribute__((always_inline)) -> uint4 {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
    };

    // Loads of stage st, restricted to the items k with k % nparts == part (nparts == 1: everything).
    // (Spreading the next stage's loads over the nine tap groups of the current one was tried: the ~1300 cycles the
    // wave spends issuing 15 KB of loads just move into the MFMA phase -- the vector-memory path, 64 B/clk/CU for
    // 61 KB per stage, is the limit, not the position of the loads.  Phase timestamps: tools/conv_ablate.hip.)
    auto load_part = [&](int st, int part, int nparts) __attribute__((always_inline)) {
        const int cbase = (MODE == MODE_P1) ? st * NSUB * BK : st * BK;
        int k = 0;
        // ---- A items (thread -> pixel tid/4 + 64 i, unit tid&3: a quad of lanes reads one pixel's 64 contiguous bytes)
#pragma unroll
        for (int im = 0; im < NI; ++im) {
#pragma unroll
            for (int i = 0; i < A_IPI; ++i, ++k) {
                if (k % nparts != part) continue;
                // NPIX is a multiple of 64 for 1x1 tiles, so the slab of item i is a compile-time function of i
                const int c = cbase + ((NSUBA == 1) ? 0 : (i * RPI_ / NPIX) * BK);       // wave-uniform
                if (c < a.C0) ra[im][i] = load16(r_x0, a_v0[im][i], c * ES);
                else ra[im][i] = load16(r_x1, a_v1[im][i], (c - a.C0) * ES);
            }
            if (MODE == MODE_S1) {
                if (k++ % nparts == part && a.pro) {
                    const int c = cbase + unit * VEC;
                    const int ig = img0 + im < a.B ? img0 + im : a.B - 1;
                    const float* ps = a.scale + (long long)ig * a.Cin + c;
                    const float* pf = a.shift + (long long)ig * a.Cin + c;
#pragma unroll
                    for (int e = 0; e < VEC; e += 4) {
                        const float4 s4 = *(const float4*)(ps + e), f4 = *(const float4*)(pf + e);
                        sc[im][e] = s4.x; sc[im][e + 1] = s4.y; sc[im][e + 2] = s4.z; sc[im][e + 3] = s4.w;
                        sh[im][e] = f4.x; sh[im][e + 1] = f4.y; sh[im][e + 2] = f4.z; sh[im][e + 3] = f4.w;
                    }
                }
            }
        }
        // ---- B items
#pragma unroll
        for (int i = 0; i < B_IPT; ++i, ++k) {
            if (k % nparts != part) continue;
            long long so;                                             // wave-uniform element offset of item i
            if (BN >= RPI_) {
                const int sub = (i * RPI_) / BN, nof = (i * RPI_) % BN;
                so = (long long)nof * a.w_row_stride + ((MODE == MODE_P1) ? (long long)(cbase + sub * BK) : sub * a.w_tap_stride + cbase);
            } else {
                const int sub = i * (RPI_ / BN);
                so = (MODE == MODE_P1) ? (long long)(cbase + sub * BK) : sub * a.w_tap_stride + cbase;
            }
            rb[i] = load16(r_w, b_v0, (int)(so * ES));
        }
    };
    auto load_stage = [&](int st) __attribute__((always_inline)) { load_part(st, 0, 1); };

    // GroupNorm-apply + SiLU on the prefetched A registers.  Runs right after the MFMAs of the previous stage were
    // issued (VALU and matrix pipes overlap), so only the ds_writes sit between the two barriers.
    auto transform_stage = [&]() __attribute__((always_inline)) {
        if (MODE == MODE_S1 && !(WDM_ABL & 1)) {
            if (a.pro) {     // wave-uniform; out-of-image pixels stay zero (padding comes AFTER the activation)
#pragma unroll
                for (int im = 0; im < NI; ++im) {
#pragma unroll
                    for (int i = 0; i < A_IPI; ++i) {
                        const uint4 tv = gn_silu_unit<T>(ra[im][i], sc[im], sh[im]);
                        const bool in = (inb_mask >> (im * A_IPI + i)) & 1u;
                        ra[im][i] = make_uint4(in ? tv.x : 0u, in ? tv.y : 0u, in ? tv.z : 0u, in ? tv.w : 0u);
                    }
                }
            }
        }
    };

    auto store_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int im = 0; im < NI; ++im) {
#pragma unroll
            for (int i = 0; i < A_IPI; ++i) {
                const bool may_overrun = (i + 1) * RPI_ > NSUBA * NPIX;   // compile-time per unrolled i
                if (!may_overrun || (tid >> 2) + i * RPI_ < NSUBA * NPIX) *(uint4*)(smem + a_l[im][i]) = ra[im][i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_IPT; ++i) {
            const bool may_overrun = (i + 1) * RPI_ > NSUB * BN;
            if (!may_overrun || rb0 + i * RPI_ < NSUB * BN) *(uint4*)(smem + b_l0 + i * (RPI_ * 64)) = rb[i];
        }
    };

    auto compute_stage = [&](int st) __attribute__((always_inline)) {
        if (MODE == MODE_S1 && TW == 16) {
            // every MFMA row group is one image row, so the fragment of output row i at tap (dy,dx) is halo row i+dy:
            // read the WM+2 halo rows once per dx and reuse them for the three dy taps (18 A reads instead of 36)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                // wave priority falling with progress inside the stage (conv_dma_kernel.h explains): the wave that is behind wins the MFMA slot
                if (dx == 0) __builtin_amdgcn_s_setprio(2); else if (dx == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
                uint4 ah[WM + 2];
#pragma unroll
                for (int r = 0; r < WM + 2; ++r) ah[r] = *(const uint4*)(smem + a_addr[0][dx] + r * (RS * 64));
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    uint4 bfr[WN];
#pragma unroll
                    for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(smem + b_addr[j] + (dy * 3 + dx) * (BN * 64));
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) {
                            if (WDM_ABL & 8) { acc[i][j][0] += __uint_as_float(ah[i + dy].x ^ bfr[j].y); }
                            else mma16t<T>(acc[i][j], ah[i + dy], bfr[j]);
                        }
                }
            }
            return;
        }
        int nsub = NSUB;
        if (MODE == MODE_P1) { const int rem = nchunks - st * NSUB; nsub = rem < NSUB ? rem : NSUB; }
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            if (MODE == MODE_P1 && s >= nsub) break;
            if (NSUB >= 3) { if (s == 0) __builtin_amdgcn_s_setprio(2); else if (s == NSUB / 3) __builtin_amdgcn_s_setprio(1); else if (s == 2 * NSUB / 3) __builtin_amdgcn_s_setprio(0); }
            uint4 af[WM], bfr[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                int off;
                if (MODE == MODE_P1) off = a_addr[i][0] + s * (PLANE * 64);
                else if (MODE == MODE_UPS) {
                    const int dy = s / 3, dx = s % 3;
                    const int ly = a_addr[i][0] >> 8, lx = a_addr[i][0] & 255;
                    off = lds_off(((ly + dy + 1) >> 1) * RS + ((lx + dx + 1) >> 1), ku);
                } else {
                    const int dy = s / 3, dx = s % 3;
                    off = a_addr[i][dx % NDX] + dy * (RS * 64);
                }
                af[i] = *(const uint4*)(smem + off);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) bfr[j] = *(const uint4*)(smem + b_addr[j] + s * (BN * 64));
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    if (WDM_ABL & 8) { acc[i][j][0] += __uint_as_float(af[i].x ^ bfr[j].y); }   // keep the LDS reads alive
                    else mma16t<T>(acc[i][j], af[i], bfr[j]);
                }
        }
    };

    // ---- main loop
#if (WDM_ABL & 16)
    // phase timestamps (ablation builds only): block 0 and a middle block, every wave, first 12 stages
    unsigned long long* tsbuf = (unsigned long long*)a.temb;
    const bool ts_on = (bid == 0 || bid == 1000) && lane == 0;
    const int ts_base = ((bid == 0 ? 0 : 1) * 4 + wave) * 12 * 8;
#define WDM_TS(ph) do { if (ts_on && st < 12) tsbuf[ts_base + st * 8 + (ph)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WDM_TS(ph) do { } while (0)
#endif
    if (C::PREFETCH) load_stage(0);
    for (int st = 0; st < nstages; ++st) {
        if (!C::PREFETCH) load_stage(st);
        WDM_TS(0);
        transform_stage();
        WDM_TS(1);
        __syncthreads();                  // everyone finished reading the previous stage
        WDM_TS(2);
        if (!(WDM_ABL & 4) || st == 0) store_stage();
        WDM_TS(3);
        __syncthreads();
        WDM_TS(4);
        if (C::PREFETCH && st + 1 < nstages && !(WDM_ABL & 2)) load_stage(st + 1);
        WDM_TS(5);
        compute_stage(st);
        WDM_TS(6);
    }
#undef WDM_TS

    // ---- epilogue
    static_assert(C::NWAVES * 16 * WM * (16 * (WN >= 2 ? 2 : 1) + 4) * 4 <= C::LDS_BYTES, "epilogue tile does not fit in LDS");
    conv_epilogue<T, TH, TW, WM, WN>(a, acc, smem, true, wave, lane, wave_m, wave_n, img0, oy0, ox0, n0, tile_in_img);
}

// host-side launchers implemented per dtype in conv_bf16.hip / conv_f32.hip
int launch_conv_bf16(const ConvArgs& a, int mode, hipStream_t s);
int launch_conv_f32(const ConvArgs& a, int mode, hipStream_t s);
int launch_conv_f32x3(const ConvArgs& a, int mode, hipStream_t s);
int launch_conv_f16(const ConvArgs& a, int mode, hipStream_t s);

}  // namespace wdm
