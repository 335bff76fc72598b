// Training step, backward primitives (reference: the autograd graph of models/unet.py under
// models/ddm_wavelet.py:108-124, :259-272).  Every contraction reuses the forward conv kernels:
//   * dgrad of a 3x3 / 1x1 conv = the same conv kernel on dy with the weights transposed (cin <-> cout) and the taps mirrored;
//     the stride-2 Downsample's dgrad scatters dy onto the odd positions of a zero map first, the Upsample's dgrad sum-pools 2x2;
//   * wgrad: dW[tap][co][ci] = sum_pixels dy[p][co] * a[p + tap][ci] is a GEMM whose contraction index is the PIXEL, so both operands
//     are gathered into channel-major ("transposed") images first -- dyT[b][co][k], aT_tap[b][ci][k], k = output pixel -- and each tap is
//     one batched 1x1 GEMM (image b = K chunk b, per-image "weights" = aT_tap[b]) writing fp32 partials [tap][b][co][ci]; a fixed-order
//     reduction over b writes the OIHW gradient.  Deterministic: no atomics.
//   * bias gradients and the per-image temb gradients are fixed-order column sums.
#include <algorithm>

#include "common.h"

namespace wdm {

static inline int nblk(long long n, int bs) { long long g = (n + bs - 1) / bs; return (int)(g > 16384 ? 16384 : g); }

// dst[b][c][k] (row length kp, k = oy*Wo + ox) = src[b][stride*oy + off_y][stride*ox + off_x][c_off + c]  (0 outside the map / past Ho*Wo)
template <typename T>
__global__ __launch_bounds__(256) void gather_t_kernel(const T* __restrict__ src, int xs, int c_off, int C, int H, int W, int Ho, int Wo, int stride, int off_y, int off_x,
                                                       T* __restrict__ dst, int kp, long long total) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(id % kp);
        const int c = (int)((id / kp) % C);
        const long long b = id / ((long long)kp * C);
        float v = 0.f;
        if (k < Ho * Wo) {
            const int oy = k / Wo, ox = k - oy * Wo;
            const int y = stride * oy + off_y, x = stride * ox + off_x;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) v = TI<T>::ld(src, ((b * H + y) * W + x) * xs + c_off + c);
        }
        TI<T>::st(dst, id, v);
    }
}
// grad[co][ci][tap] (OIHW f32) (+)= sum_b partial[tap][b][co][ci]   (rows_g rows per image in the partial buffer)
__global__ __launch_bounds__(256) void reduce_wgrad_kernel(const float* __restrict__ part, int taps, int B, int rows_g, int cout, int cin, float* __restrict__ grad,
                                                           int accumulate) {
    const long long total = (long long)taps * cout * cin;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(id % cin);
        const int co = (int)((id / cin) % cout);
        const int tap = (int)(id / ((long long)cin * cout));
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += part[(((long long)tap * B + b) * rows_g + co) * cin + ci];
        const long long o = ((long long)co * cin + ci) * taps + tap;
        grad[o] = accumulate ? grad[o] + s : s;
    }
}
// out[g][c] (+)= sum over the rows of group g of x[row][c]; rows_per_group rows per group (bias grad: one group; temb grad: one per image)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int xs, int C, long long rows_per_group, float* __restrict__ out, int accumulate) {
    __shared__ float red[256];
    const int g = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (long long r = sl; r < rows_per_group; r += 4) s += TI<T>::ld(x, (g * rows_per_group + r) * xs + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && c < C) {
        const float t = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
        out[(long long)g * C + c] = accumulate ? out[(long long)g * C + c] + t : t;
    }
}
// Downsample dgrad helper: z[b][2oy+1][2ox+1][c] = dy[b][oy][ox][c], zero elsewhere (z is H x W, dy is H/2 x W/2)
template <typename T>
__global__ __launch_bounds__(256) void scatter_odd_kernel(const T* __restrict__ dy, int C, int Ho, int Wo, T* __restrict__ z, long long total) {
    const int H = 2 * Ho, W = 2 * Wo;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % C);
        const long long p = id / C;
        const int x = (int)(p % W), y = (int)((p / W) % H);
        const long long b = p / ((long long)W * H);
        float v = 0.f;
        if ((y & 1) && (x & 1)) v = TI<T>::ld(dy, ((b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c);
        TI<T>::st(z, id, v);
    }
}
// nearest x2 upsample (forward, materialised for the wgrad) and its adjoint (2x2 sum pool, optionally accumulated)
template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(const T* __restrict__ x, int C, int h, int w, T* __restrict__ y, long long total) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % C);
        const long long p = id / C;
        const int X = (int)(p % (2 * w)), Y = (int)((p / (2 * w)) % (2 * h));
        const long long b = p / ((long long)4 * w * h);
        TI<T>::st(y, id, TI<T>::ld(x, ((b * h + (Y >> 1)) * w + (X >> 1)) * C + c));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void sumpool2_kernel(const T* __restrict__ dy, int C, int h, int w, T* __restrict__ dx, int accumulate, long long total) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % C);
        const long long p = id / C;
        const int x = (int)(p % w), y = (int)((p / w) % h);
        const long long b = p / ((long long)w * h);
        const long long r0 = ((b * 2 * h + 2 * y) * 2 * w + 2 * x) * C + c;
        float s = TI<T>::ld(dy, r0) + TI<T>::ld(dy, r0 + C) + TI<T>::ld(dy, r0 + (long long)2 * w * C) + TI<T>::ld(dy, r0 + (long long)2 * w * C + C);
        if (accumulate) s += TI<T>::ld(dx, id);
        TI<T>::st(dx, id, s);
    }
}
// dgrad weights: dst[tap'][row = ci][k = co] = w[co][ci][taps-1-tap'] (transposed, taps mirrored), k zero-padded to kpad, rows to rows_total
template <typename T>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ w, int cout, int cin, int kk, T* __restrict__ dst, int rows_total, int kpad) {
    const long long total = (long long)kk * rows_total * kpad;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(id % kpad);
        const int ci = (int)((id / kpad) % rows_total);
        const int tp = (int)(id / ((long long)kpad * rows_total));
        const float v = (co < cout && ci < cin) ? w[((long long)co * cin + ci) * kk + (kk - 1 - tp)] : 0.f;
        TI<T>::st(dst, id, v);
    }
}
// y = x with channels zero-padded from C to Cp (dense)
template <typename T>
__global__ __launch_bounds__(256) void pad_channels_kernel(const T* __restrict__ x, int C, int Cp, T* __restrict__ y, long long total) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % Cp);
        const long long p = id / Cp;
        TI<T>::st(y, id, c < C ? TI<T>::ld(x, p * C + c) : 0.f);
    }
}

static int kalign(int dtype) { return dtype == WDM_BF16 ? 32 : 16; }

// typed launch helpers ------------------------------------------------------------------------------------------------
template <typename T>
static void gather_t(hipStream_t s, const void* src, int xs, int c_off, int C, int B, int H, int W, int Ho, int Wo, int stride, int off_y, int off_x, void* dst,
                     int rows_per_img, int kp) {
    // dst image stride is rows_per_img * kp (rows past C stay as they are: the buffer is zeroed once by the caller)
    for (int b = 0; b < B; ++b) {      // one launch per image keeps the index math 32-bit and the image stride free
        const long long total = (long long)C * kp;
        hipLaunchKernelGGL(gather_t_kernel<T>, dim3(nblk(total, 256)), dim3(256), 0, s, (const T*)src + (long long)b * H * W * xs, xs, c_off, C, H, W, Ho, Wo, stride,
                           off_y, off_x, (T*)dst + (long long)b * rows_per_img * kp, kp, total);
    }
}
#define BY_DTYPE(dtype, FN, ...) do { if ((dtype) == WDM_BF16) FN<__bf16>(__VA_ARGS__); else FN<float>(__VA_ARGS__); } while (0)

template <typename T> static void l_scatter_odd(hipStream_t s, const void* dy, int C, int Ho, int Wo, void* z, long long total) {
    hipLaunchKernelGGL(scatter_odd_kernel<T>, dim3(nblk(total, 256)), dim3(256), 0, s, (const T*)dy, C, Ho, Wo, (T*)z, total);
}
template <typename T> static void l_upsample2(hipStream_t s, const void* x, int C, int h, int w, void* y, long long total) {
    hipLaunchKernelGGL(upsample2_kernel<T>, dim3(nblk(total, 256)), dim3(256), 0, s, (const T*)x, C, h, w, (T*)y, total);
}
template <typename T> static void l_sumpool2(hipStream_t s, const void* dy, int C, int h, int w, void* dx, int acc, long long total) {
    hipLaunchKernelGGL(sumpool2_kernel<T>, dim3(nblk(total, 256)), dim3(256), 0, s, (const T*)dy, C, h, w, (T*)dx, acc, total);
}
template <typename T> static void l_pack_dgrad(hipStream_t s, const float* w, int cout, int cin, int kk, void* dst, int rows, int kpad) {
    const long long total = (long long)kk * rows * kpad;
    hipLaunchKernelGGL(pack_dgrad_kernel<T>, dim3(nblk(total, 256)), dim3(256), 0, s, w, cout, cin, kk, (T*)dst, rows, kpad);
}
template <typename T> static void l_pad_channels(hipStream_t s, const void* x, int C, int Cp, void* y, long long total) {
    hipLaunchKernelGGL(pad_channels_kernel<T>, dim3(nblk(total, 256)), dim3(256), 0, s, (const T*)x, C, Cp, (T*)y, total);
}
template <typename T> static void l_colsum(hipStream_t s, const void* x, int xs, int C, long long rows_per_group, int groups, float* out, int acc) {
    hipLaunchKernelGGL(colsum_kernel<T>, dim3((C + 63) / 64, groups), dim3(256), 0, s, (const T*)x, xs, C, rows_per_group, out, acc);
}

// ---- dgrad: dx (+)= conv^T(dy).  (H, W) is the forward INPUT map; dy is dense NHWC [B][Ho][Wo][cout]; dx dense [B][H][W][cin].
int conv_dgrad(Ctx& c, int mode, const float* w_oihw, int cin, int cout, const Tens& dy, int H, int W, void* dx, bool accumulate) {
    const int k = mode == MODE_P1 ? 1 : 3, kk = k * k;
    const size_t es = dsize(c.dtype);
    const int kpad = (int)align_up((size_t)cout, kalign(c.dtype));             // contraction length (forward cout), padded
    const int rows = conv_rows_pad(cin);
    void* wd = c.ar->alloc((size_t)kk * rows * kpad * es);
    if (!wd) WDM_FAIL(WDM_ENOMEM, "workspace too small (dgrad weights)");
    if (!c.dry) BY_DTYPE(c.dtype, l_pack_dgrad, c.s, w_oihw, cout, cin, kk, wd, rows, kpad);
    // what the transposed conv reads: dy (3x3 s1, 1x1, upsample), dy scattered onto the odd grid (Downsample); channel-padded if needed
    const void* src = dy.p;
    int Hs = dy.H, Ws = dy.W;
    void* t_sc = nullptr; void* t_pad = nullptr; void* t_up = nullptr;
    if (mode == MODE_S2) {
        Hs = 2 * dy.H; Ws = 2 * dy.W;
        const long long total = (long long)c.B * Hs * Ws * cout;
        t_sc = c.ar->alloc((size_t)total * es);
        if (!t_sc) WDM_FAIL(WDM_ENOMEM, "workspace too small (dgrad scatter)");
        if (!c.dry) BY_DTYPE(c.dtype, l_scatter_odd, c.s, dy.p, cout, dy.H, dy.W, t_sc, total);
        src = t_sc;
    }
    if (kpad != cout) {
        const long long total = (long long)c.B * Hs * Ws * kpad;
        t_pad = c.ar->alloc((size_t)total * es);
        if (!t_pad) WDM_FAIL(WDM_ENOMEM, "workspace too small (dgrad channel pad)");
        if (!c.dry) BY_DTYPE(c.dtype, l_pad_channels, c.s, src, cout, kpad, t_pad, total);
        src = t_pad;
    }
    void* out = dx;
    if (mode == MODE_UPS) {      // gradient on the upsampled map first, then 2x2 sum pool
        t_up = c.ar->alloc((size_t)c.B * Hs * Ws * cin * es);
        if (!t_up) WDM_FAIL(WDM_ENOMEM, "workspace too small (dgrad upsample)");
        out = t_up;
    }
    int rc = WDM_OK;
    if (!c.dry) {
        ConvArgs a{};
        a.x0 = src; a.C0 = kpad; a.xs0 = kpad; a.C1 = 0;
        a.B = c.B; a.Hin = a.Hout = Hs; a.Win = a.Wout = Ws;
        a.Cin = kpad; a.Cout = cin;
        a.w = wd; a.w_tap_stride = (long long)rows * kpad; a.w_img_stride = 0; a.w_row_stride = kpad; a.w_rows = rows;
        a.w_bytes = (unsigned)((size_t)kk * rows * kpad * es);
        a.alpha = 1.f;
        if (accumulate && mode != MODE_UPS) { a.res = dx; a.res_s = cin; }
        a.y = out; a.y_mode = Y_NHWC; a.y_s = cin;
        rc = launch_conv(a, mode == MODE_P1 ? MODE_P1 : MODE_S1, c.dtype, c.s);
        if (rc == WDM_OK && mode == MODE_UPS) {
            const long long total = (long long)c.B * H * W * cin;
            BY_DTYPE(c.dtype, l_sumpool2, c.s, t_up, cin, H, W, dx, accumulate ? 1 : 0, total);
        }
    }
    if (t_up) c.ar->free(t_up);
    if (t_pad) c.ar->free(t_pad);
    if (t_sc) c.ar->free(t_sc);
    c.ar->free(wd);
    return rc;
}

// ---- wgrad: dw[co][ci][tap] (OIHW f32) (+)= sum_pixels dy[p][co] * x[p + tap][ci];  x = [x0 | x1] (the forward input; for MODE_UPS the
// low-resolution map, upsampled here), dy dense [B][Ho][Wo][cout]
int conv_wgrad(Ctx& c, int mode, const Tens& x0, const Tens* x1, const Tens& dy, int cout, float* dw, bool accumulate) {
    const int cin = x0.C + (x1 ? x1->C : 0);
    const int k = mode == MODE_P1 ? 1 : 3, kk = k * k;
    const size_t es = dsize(c.dtype);
    const int Ho = dy.H, Wo = dy.W;
    const int kp = (int)align_up((size_t)Ho * Wo, kalign(c.dtype));
    const int rows_g = (int)align_up((size_t)cout, 64);                         // the GEMM's M grid: rows_g = Hg x Wg "pixels"
    const int Wg = (rows_g % 128 == 0) ? 16 : 8, Hg = rows_g / Wg;
    // forward input as the conv saw it
    const Tens* s0 = &x0; const Tens* s1 = x1;
    Tens up0, up1;
    void* t_up0 = nullptr; void* t_up1 = nullptr;
    int H = x0.H, W = x0.W;
    if (mode == MODE_UPS) {
        if (x1) WDM_FAIL(WDM_EINVAL, "wgrad: upsample conv takes a single input");
        H = 2 * x0.H; W = 2 * x0.W;
        const long long total = (long long)c.B * H * W * x0.C;
        t_up0 = c.ar->alloc((size_t)total * es);
        if (!t_up0) WDM_FAIL(WDM_ENOMEM, "workspace too small (wgrad upsample)");
        if (x0.xs != x0.C) WDM_FAIL(WDM_EINVAL, "wgrad: upsample input must be dense");
        if (!c.dry) BY_DTYPE(c.dtype, l_upsample2, c.s, x0.p, x0.C, x0.H, x0.W, t_up0, total);
        up0 = x0; up0.p = t_up0; up0.H = H; up0.W = W; up0.xs = x0.C;
        s0 = &up0;
    }
    (void)up1; (void)t_up1;
    void* dyT = c.ar->alloc((size_t)c.B * rows_g * kp * es);
    void* aT = c.ar->alloc((size_t)c.B * cin * kp * es);
    float* part = (float*)c.ar->alloc((size_t)kk * c.B * rows_g * cin * sizeof(float));
    if (!dyT || !aT || !part) WDM_FAIL(WDM_ENOMEM, "workspace too small (wgrad)");
    int rc = WDM_OK;
    if (!c.dry) {
        WDM_HIP(hipMemsetAsync(dyT, 0, (size_t)c.B * rows_g * kp * es, c.s));
        BY_DTYPE(c.dtype, gather_t, c.s, dy.p, dy.xs, 0, cout, c.B, Ho, Wo, Ho, Wo, 1, 0, 0, dyT, rows_g, kp);
        const int stride = mode == MODE_S2 ? 2 : 1;
        for (int tap = 0; tap < kk && rc == WDM_OK; ++tap) {
            const int ty = tap / k, tx = tap % k;
            const int oy = mode == MODE_P1 ? 0 : (mode == MODE_S2 ? ty : ty - 1), ox = mode == MODE_P1 ? 0 : (mode == MODE_S2 ? tx : tx - 1);
            BY_DTYPE(c.dtype, gather_t, c.s, s0->p, s0->xs, 0, s0->C, c.B, H, W, Ho, Wo, stride, oy, ox, aT, cin, kp);
            if (s1) BY_DTYPE(c.dtype, gather_t, c.s, s1->p, s1->xs, 0, s1->C, c.B, H, W, Ho, Wo, stride, oy, ox, (char*)aT + (size_t)s0->C * kp * es, cin, kp);
            ConvArgs a{};
            a.x0 = dyT; a.C0 = kp; a.xs0 = kp; a.C1 = 0;
            a.B = c.B; a.Hin = a.Hout = Hg; a.Win = a.Wout = Wg;
            a.Cin = kp; a.Cout = cin;
            a.w = aT; a.w_tap_stride = 0; a.w_img_stride = (long long)cin * kp; a.w_row_stride = kp; a.w_rows = cin;
            a.w_bytes = (unsigned)((size_t)cin * kp * es);
            a.alpha = 1.f;
            a.y = part + (size_t)tap * c.B * rows_g * cin; a.y_mode = Y_NHWC_F32; a.y_s = cin;
            rc = launch_conv(a, MODE_P1, c.dtype, c.s);
        }
        if (rc == WDM_OK) {
            const long long total = (long long)kk * cout * cin;
            // partial rows per image = rows_g: compact view for the reduction
            hipLaunchKernelGGL(reduce_wgrad_kernel, dim3(nblk(total, 256)), dim3(256), 0, c.s, part, kk, c.B, rows_g, cout, cin, dw, accumulate ? 1 : 0);
            WDM_HIP(hipGetLastError());
        }
    }
    c.ar->free(part); c.ar->free(aT); c.ar->free(dyT);
    if (t_up0) c.ar->free(t_up0);
    return rc;
}

// db[c] (+)= sum over all rows of dy;  per_image: out[b][c] = sum over the image's rows (temb gradient)
int colsum(Ctx& c, const Tens& dy, float* out, bool per_image, bool accumulate) {
    if (c.dry) return WDM_OK;
    const long long rows = (long long)dy.H * dy.W * (per_image ? 1 : c.B);
    BY_DTYPE(c.dtype, l_colsum, c.s, dy.p, dy.xs, dy.C, rows, per_image ? c.B : 1, out, accumulate ? 1 : 0);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

}  // namespace wdm

// =================================================================================================
// C ABI: per-op test entry point
// =================================================================================================
using namespace wdm;

extern "C" int wdm_conv_backward(wdm_handle* h, const float* w, int cin, int cout, int mode, const float* x, const float* dy, int B, int H, int W, float* dx,
                                 float* dw, float* db, int dtype, void* scratch, size_t scratch_bytes, void* stream) {
    if (!h || !w || !x || !dy || !dw || !scratch) WDM_FAIL(WDM_EINVAL, "wdm_conv_backward: null argument");
    if (mode < 0 || mode > 3) WDM_FAIL(WDM_EINVAL, "wdm_conv_backward: bad mode");
    Arena ar(scratch, scratch_bytes);
    Ctx c{(hipStream_t)stream, dtype, B, &ar, false};
    const size_t es = dsize(dtype);
    const int Ho = mode == MODE_S2 ? H / 2 : mode == MODE_UPS ? 2 * H : H, Wo = mode == MODE_S2 ? W / 2 : mode == MODE_UPS ? 2 * W : W;
    Tens tx, tdy;
    tx.p = ar.alloc((size_t)B * H * W * cin * es); tx.C = cin; tx.H = H; tx.W = W; tx.xs = cin;
    tdy.p = ar.alloc((size_t)B * Ho * Wo * cout * es); tdy.C = cout; tdy.H = Ho; tdy.W = Wo; tdy.xs = cout;
    void* tdx = dx ? ar.alloc((size_t)B * H * W * cin * es) : nullptr;
    if (!tx.p || !tdy.p || (dx && !tdx)) WDM_FAIL(WDM_ENOMEM, "wdm_conv_backward: scratch too small");
    WDM_TRY(k_nchw_to_nhwc(x, tx.p, B, cin, H, W, dtype, c.s));
    WDM_TRY(k_nchw_to_nhwc(dy, tdy.p, B, cout, Ho, Wo, dtype, c.s));
    if (dx) {
        WDM_TRY(conv_dgrad(c, mode, w, cin, cout, tdy, H, W, tdx, false));
        WDM_TRY(k_nhwc_to_nchw(tdx, dx, B, cin, H, W, dtype, c.s));
    }
    WDM_TRY(conv_wgrad(c, mode, tx, nullptr, tdy, cout, dw, false));
    if (db) WDM_TRY(colsum(c, tdy, db, false, false));
    return WDM_OK;
}
