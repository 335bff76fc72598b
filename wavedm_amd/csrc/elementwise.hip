// HBM-bound kernels of the sampling path: Haar wavelet-packet DWT/IDWT, patch gather into the UNet input,
// scatter-mean + DDIM update, layout conversion, GroupNorm statistics, attention softmax, the timestep-embedding
// MLP, and weight packing.  All deterministic (no atomics): every reduction has a fixed order.
#include "common.h"
#include "gn_group.h"

namespace wdm {

static inline int nblocks(long long n, int bs) { return (int)((n + bs - 1) / bs); }

// =================================================================================================
// Haar wavelet packet, 2 levels (models/wavelet.py:37-49).
// The 16 analysis filters are f_j[p][q] = 0.25 * (-1)^(j0*(q>>1) + j1*(p>>1) + j2*(q&1) + j3*(p&1)), i.e. a 16-point
// Walsh-Hadamard transform of the 4x4 block re-indexed as idx = (p&1)<<3 | (q&1)<<2 | (p>>1)<<1 | (q>>1): four
// add/sub lifting stages and one scale.  Output channel = j*3 + c (sub-band major), integer-exact bookkeeping.
// One thread per 4x4 block: 4 coalesced 16-byte reads, 16 coalesced 4-byte writes (one per sub-band plane).
// =================================================================================================
__device__ __forceinline__ void wht16(float* v) {
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if ((i & s) == 0) {
                const float a = v[i], b = v[i | s];
                v[i] = a + b;
                v[i | s] = a - b;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] *= 0.25f;
}

// pre-affine: the transform of (scale * x + shift) -- data_transform (2x - 1, restoration.py:8-9) folded into the load
__global__ __launch_bounds__(256) void dwt_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, float scale, float shift) {
    const int h = H >> 2, w = W >> 2;
    const long long total = (long long)B * 3 * h * w;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(id % w);
        const int u = (int)((id / w) % h);
        const int c = (int)((id / ((long long)w * h)) % 3);
        const int b = (int)(id / ((long long)w * h * 3));
        const float* src = x + (((long long)b * 3 + c) * H + 4 * u) * W + 4 * v;
        float t[16];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 r = *(const float4*)(src + (long long)p * W);
            const float rq[4] = {r.x * scale + shift, r.y * scale + shift, r.z * scale + shift, r.w * scale + shift};
#pragma unroll
            for (int q = 0; q < 4; ++q) t[((p & 1) << 3) | ((q & 1) << 2) | ((p >> 1) << 1) | (q >> 1)] = rq[q];
        }
        wht16(t);
#pragma unroll
        for (int j = 0; j < 16; ++j) y[(((long long)b * 48 + j * 3 + c) * h + u) * w + v] = t[j];
    }
}

// Coefficient channel ch = j*3 + c comes from y_lo (B, lo_total, h, w) when ch < n_lo, else from y (B, 48, h, w): the
// torch.cat([x0[:, :pc], hf_wav[:, pc:]]) of restoration.py:114 without materialising it (n_lo = 0: plain inverse).
// post != 0: clamp((x + 1) / 2, 0, 1) = inverse_data_transform (restoration.py:12-13) on the way out.
__global__ __launch_bounds__(256) void dwt_inv_kernel(const float* __restrict__ y, float* __restrict__ x, int B, int h, int w, const float* __restrict__ y_lo,
                                                      int lo_total, int n_lo, int post) {
    const int H = h << 2, W = w << 2;
    const long long total = (long long)B * 3 * h * w;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(id % w);
        const int u = (int)((id / w) % h);
        const int c = (int)((id / ((long long)w * h)) % 3);
        const int b = (int)(id / ((long long)w * h * 3));
        float t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int chn = j * 3 + c;
            t[j] = chn < n_lo ? y_lo[(((long long)b * lo_total + chn) * h + u) * w + v] : y[(((long long)b * 48 + chn) * h + u) * w + v];
        }
        wht16(t);   // the basis is orthonormal and symmetric in (j, idx): the inverse is the same butterfly
        if (post) {
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = fminf(fmaxf((t[i] + 1.0f) * 0.5f, 0.0f), 1.0f);
        }
        float* dst = x + (((long long)b * 3 + c) * H + 4 * u) * W + 4 * v;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float rq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) rq[q] = t[((p & 1) << 3) | ((q & 1) << 2) | ((p >> 1) << 1) | (q >> 1)];
            *(float4*)(dst + (long long)p * W) = make_float4(rq[0], rq[1], rq[2], rq[3]);
        }
    }
}

int k_dwt_fwd(const float* x, float* y, int B, int H, int W, hipStream_t s, float scale, float shift) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3)) WDM_FAIL(WDM_EINVAL, "dwt_fwd: H=%d W=%d must be positive multiples of 4", H, W);
    const long long total = (long long)B * 3 * (H / 4) * (W / 4);
    hipLaunchKernelGGL(dwt_fwd_kernel, dim3(nblocks(total, 256) > 8192 ? 8192 : nblocks(total, 256)), dim3(256), 0, s, x, y, B, H, W, scale, shift);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
int k_dwt_inv(const float* y, float* x, int B, int h, int w, hipStream_t s, const float* y_lo, int lo_total, int n_lo, int post) {
    if (B <= 0 || h <= 0 || w <= 0) WDM_FAIL(WDM_EINVAL, "dwt_inv: bad shape");
    if (n_lo < 0 || n_lo > 48 || (n_lo > 0 && (!y_lo || lo_total < n_lo))) WDM_FAIL(WDM_EINVAL, "dwt_inv: bad low-band source (%d of %d channels)", n_lo, lo_total);
    const long long total = (long long)B * 3 * h * w;
    hipLaunchKernelGGL(dwt_inv_kernel, dim3(nblocks(total, 256) > 8192 ? 8192 : nblocks(total, 256)), dim3(256), 0, s, y, x, B, h, w, y_lo, lo_total, n_lo, post);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// =================================================================================================
// patch gather: crop()+cat of ddm_wavelet.py:467-478, written straight into the NHWC UNet input
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void pack_channels_kernel(const float* __restrict__ src, int nch, int H, int W, const int32_t* __restrict__ patches,
                                                            int n, int p, T* __restrict__ x96, int c_total, int c_off) {
    h16_mode_init<T>();
    const long long total = (long long)n * p * p * nch;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % nch);
        const int xx = (int)((id / nch) % p);
        const int yy = (int)((id / ((long long)nch * p)) % p);
        const int k = (int)(id / ((long long)nch * p * p));
        int img = k, hi = 0, wi = 0;
        if (patches != nullptr) { img = patches[3 * k]; hi = patches[3 * k + 1]; wi = patches[3 * k + 2]; }
        const float v = src[(((long long)img * nch + c) * H + hi + yy) * W + wi + xx];
        TI<T>::st(x96, (((long long)k * p + yy) * p + xx) * c_total + c_off + c, v);
    }
}

// The same gather with coalesced accesses on BOTH sides (the kernel above reads NCHW with the channel as the fastest thread index: lanes H W floats apart).
// One workgroup per (patch, patch row): the row's nch x p values are read channel by channel -- consecutive lanes = consecutive pixels of one channel
// plane --, turned in LDS (row stride p + 1 floats: odd, conflict-free both ways) and written pixel by pixel, channels fastest, as the NHWC row wants them.
constexpr int PACK_MAX_P = 128, PACK_MAX_CH = 48;
template <typename T>
__global__ __launch_bounds__(256) void pack_channels_rows_kernel(const float* __restrict__ src, int nch, int H, int W, const int32_t* __restrict__ patches,
                                                                 int p, T* __restrict__ x96, int c_total, int c_off) {
    h16_mode_init<T>();
    __shared__ float tile[PACK_MAX_CH * (PACK_MAX_P + 1)];
    const int k = (int)(blockIdx.x / (unsigned)p), yy = (int)(blockIdx.x - (unsigned)k * (unsigned)p);
    int img = k, hi = 0, wi = 0;
    if (patches != nullptr) { img = patches[3 * k]; hi = patches[3 * k + 1]; wi = patches[3 * k + 2]; }
    const int ld = p + 1, tot = nch * p;
    const float* row0 = src + ((long long)img * nch * H + hi + yy) * W + wi;
    for (int id = threadIdx.x; id < tot; id += 256) {
        const int c = id / p, xx = id - c * p;
        tile[c * ld + xx] = row0[(long long)c * H * W + xx];
    }
    __syncthreads();
    T* out = x96 + ((long long)k * p + yy) * p * c_total + c_off;
    for (int id = threadIdx.x; id < tot; id += 256) {
        const int xx = id / nch, c = id - xx * nch;
        TI<T>::st(out, (long long)xx * c_total + c, tile[c * ld + xx]);
    }
}

int k_pack_channels(const float* src, int nch, int H, int W, const int32_t* patches, int n, int p, void* x96, int c_total, int c_off, int dtype,
                    hipStream_t s) {
    if (n <= 0 || p <= 0 || nch <= 0 || c_off < 0 || c_off + nch > c_total) WDM_FAIL(WDM_EINVAL, "pack_channels: bad arguments");
    if (patches == nullptr && (p != H || p != W)) WDM_FAIL(WDM_EINVAL, "pack_channels: identity patch list needs p == H == W");
    if (p <= PACK_MAX_P && nch <= PACK_MAX_CH && (long long)n * p < 2147483647LL) {
        if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(pack_channels_rows_kernel<H16>, dim3(n * p), dim3(256), 0, s, src, nch, H, W, patches, p, (H16*)x96, c_total, c_off));
        else hipLaunchKernelGGL(pack_channels_rows_kernel<float>, dim3(n * p), dim3(256), 0, s, src, nch, H, W, patches, p, (float*)x96, c_total, c_off);
        WDM_HIP(hipGetLastError());
        return WDM_OK;
    }
    const long long total = (long long)n * p * p * nch;
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(pack_channels_kernel<H16>, dim3(g), dim3(256), 0, s, src, nch, H, W, patches, n, p, (H16*)x96, c_total, c_off));
    else hipLaunchKernelGGL(pack_channels_kernel<float>, dim3(g), dim3(256), 0, s, src, nch, H, W, patches, n, p, (float*)x96, c_total, c_off);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// =================================================================================================
// scatter-mean + DDIM update (ddm_wavelet.py:485-502).  Gather form: every full-image element sums the patches that
// cover it in patch-list order -- the same order as the reference's sequential "+=", so the fp32 sum is identical.
// =================================================================================================
__global__ __launch_bounds__(256) void ddim_update_kernel(const float* __restrict__ eps, const int32_t* __restrict__ patches, int n, int p,
                                                          const float* __restrict__ x_t, int nimg, int H, int W, float s1m, float sa, float san, float c2,
                                                          float* __restrict__ x0o, float* __restrict__ xno, const float* __restrict__ noise, float c1) {
    const long long total = (long long)nimg * 3 * H * W;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(id % W);
        const int yy = (int)((id / W) % H);
        const int c = (int)((id / ((long long)W * H)) % 3);
        const int img = (int)(id / ((long long)W * H * 3));
        float acc = 0.f, cnt = 0.f;
        if (patches == nullptr) {
            acc = eps[id];
            cnt = 1.f;
        } else {
            for (int k = 0; k < n; ++k) {
                const int pi = patches[3 * k], hi = patches[3 * k + 1], wi = patches[3 * k + 2];
                if (pi == img && (unsigned)(yy - hi) < (unsigned)p && (unsigned)(xx - wi) < (unsigned)p) {
                    acc += eps[(((long long)k * 3 + c) * p + (yy - hi)) * p + (xx - wi)];
                    cnt += 1.f;
                }
            }
        }
        const float et = acc / cnt;
        const float xt = x_t[id];
        const float x0 = (xt - et * s1m) / sa;
        x0o[id] = x0;
        // eta != 0 (ddm_wavelet.py:500-502): at_next.sqrt() * x0_t + c1 * randn_like(x) + c2 * et, summed left to right like the reference's expression
        xno[id] = noise ? san * x0 + c1 * noise[id] + c2 * et : san * x0 + c2 * et;
    }
}

int k_ddim_update(const float* eps, const int32_t* patches, int n, int p, const float* x_t, int nimg, int H, int W, float s1m, float sa, float san,
                  float c2, float* x0, float* xn, hipStream_t s, const float* noise, float c1) {
    if (n <= 0 || nimg <= 0) WDM_FAIL(WDM_EINVAL, "ddim_update: bad arguments");
    if (patches == nullptr && (p != H || p != W || n != nimg)) WDM_FAIL(WDM_EINVAL, "ddim_update: identity patch list needs n == nimg, p == H == W");
    const long long total = (long long)nimg * 3 * H * W;
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    hipLaunchKernelGGL(ddim_update_kernel, dim3(g), dim3(256), 0, s, eps, patches, n, p, x_t, nimg, H, W, s1m, sa, san, c2, x0, xn, noise, c1);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// ---- patch-sharded single image: each rank owns a subset of the patches.  patch_accumulate writes the rank's
// partial sums [0 .. nimg*3*H*W) and partial counts [nimg*3*H*W ..) into ONE buffer (one all-reduce per step); ddim_from_sums
// divides the reduced sums by the reduced counts and applies the same DDIM update as ddim_update_kernel.
__global__ __launch_bounds__(256) void patch_accumulate_kernel(const float* __restrict__ eps, const int32_t* __restrict__ patches, int n, int p, int nimg, int H,
                                                               int W, float* __restrict__ acc_cnt) {
    const long long total = (long long)nimg * 3 * H * W;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(id % W);
        const int yy = (int)((id / W) % H);
        const int c = (int)((id / ((long long)W * H)) % 3);
        const int img = (int)(id / ((long long)W * H * 3));
        float acc = 0.f, cnt = 0.f;
        for (int k = 0; k < n; ++k) {
            const int pi = patches[3 * k], hi = patches[3 * k + 1], wi = patches[3 * k + 2];
            if (pi == img && (unsigned)(yy - hi) < (unsigned)p && (unsigned)(xx - wi) < (unsigned)p) {
                acc += eps[(((long long)k * 3 + c) * p + (yy - hi)) * p + (xx - wi)];
                cnt += 1.f;
            }
        }
        acc_cnt[id] = acc;
        acc_cnt[total + id] = cnt;
    }
}
__global__ __launch_bounds__(256) void ddim_from_sums_kernel(const float* __restrict__ acc_cnt, const float* __restrict__ x_t, long long total, float s1m, float sa,
                                                             float san, float c2, float* __restrict__ x0o, float* __restrict__ xno) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const float et = acc_cnt[id] / acc_cnt[total + id];
        const float x0 = (x_t[id] - et * s1m) / sa;
        x0o[id] = x0;
        xno[id] = san * x0 + c2 * et;
    }
}
int k_patch_accumulate(const float* eps, const int32_t* patches, int n, int p, int nimg, int H, int W, float* acc_cnt, hipStream_t s) {
    if (n < 0 || nimg <= 0 || !patches) WDM_FAIL(WDM_EINVAL, "patch_accumulate: bad arguments");
    const long long total = (long long)nimg * 3 * H * W;
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    hipLaunchKernelGGL(patch_accumulate_kernel, dim3(g), dim3(256), 0, s, eps, patches, n, p, nimg, H, W, acc_cnt);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
int k_ddim_from_sums(const float* acc_cnt, const float* x_t, int nimg, int H, int W, float s1m, float sa, float san, float c2, float* x0, float* xn, hipStream_t s) {
    const long long total = (long long)nimg * 3 * H * W;
    if (total <= 0) WDM_FAIL(WDM_EINVAL, "ddim_from_sums: empty image");
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    hipLaunchKernelGGL(ddim_from_sums_kernel, dim3(g), dim3(256), 0, s, acc_cnt, x_t, total, s1m, sa, san, c2, x0, xn);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// =================================================================================================
// layout conversion at the drop-in model(x, t) boundary
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int HW) {
    h16_mode_init<T>();
    const long long total = (long long)B * C * HW;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % C);
        const long long bp = id / C;               // b*HW + pix
        const long long b = bp / HW, pix = bp % HW;
        TI<T>::st(dst, id, src[(b * C + c) * HW + pix]);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int HW) {
    const long long total = (long long)B * C * HW;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const long long pix = id % HW;
        const int c = (int)((id / HW) % C);
        const long long b = id / ((long long)HW * C);
        dst[id] = TI<T>::ld(src, (b * HW + pix) * C + c);
    }
}
int k_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dtype, hipStream_t s) {
    const long long total = (long long)B * C * H * W;
    if (total <= 0) WDM_FAIL(WDM_EINVAL, "nchw_to_nhwc: empty tensor");
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<H16>, dim3(g), dim3(256), 0, s, src, (H16*)dst, B, C, H * W));
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, s, src, (float*)dst, B, C, H * W);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
int k_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int dtype, hipStream_t s) {
    const long long total = (long long)B * C * H * W;
    if (total <= 0) WDM_FAIL(WDM_EINVAL, "nhwc_to_nchw: empty tensor");
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<H16>, dim3(g), dim3(256), 0, s, (const H16*)src, dst, B, C, H * W));
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)src, dst, B, C, H * W);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// =================================================================================================
// GroupNorm(32 groups, eps, biased variance) statistics -> per-(image, channel) scale / shift   (unet.py:36-37)
//
// Partial statistics live in a "stats" buffer float4[B][nslab][C] = (K, S1, S2, n): over the n pixels of a slab,
// S1 = sum(x-K), S2 = sum((x-K)^2) with a per-(slab, channel) pivot K taken from the data (shifted-data variance: no
// catastrophic cancellation).  They are produced either
//   * by the epilogue of the convolution that writes the tensor (conv_kernel.h: the values are in registers anyway;
//     one slab = the rows of one wave's output tile) -- the normal case inside the UNet, no extra pass over HBM; or
//   * by gn_partial_kernel (one workgroup per (pixel slab, image), 16-byte loads, fixed-order LDS reduction) for
//     tensors that did not come out of a conv (block-level entry points, the tests).
// gn_finalize_kernel: one wave per (image, group) re-centres the partials of the group's channels -- which may come
// from TWO tensors, the channel concat [x0 | x1] (1280 = 768 + 512 channels -> 40-channel groups straddle the seam) --
// on a common pivot, reduces them in fp64 with a fixed shuffle tree and writes
//     scale[b,c] = rstd*gamma[c],  shift[b,c] = beta[c] - mean*scale[b,c]      (times -log2(e) for the SiLU conv prologue)
// =================================================================================================
int gn_default_nslab(int HW) { int n = HW / 64; return n < 1 ? 1 : (n > 64 ? 64 : n); }
size_t gn_stats_bytes(int B, int nslab, int C) { return (size_t)B * nslab * C * 4 * sizeof(float); }

template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, int xs, int C, int HW, int nslab, float4* __restrict__ stats) {
    constexpr int VEC = TI<T>::VEC;
    __shared__ float red[256 * VEC * 2];
    const int cols = C / VEC;                         // 16-byte channel vectors per pixel
    const int cb = blockIdx.z;                        // column block (cols may exceed 256 in f32 mode)
    const int cols_here = min(cols - cb * 256, 256);
    const int rows = 256 / cols_here;
    const int tid = threadIdx.x;
    const int col = tid % cols_here, row = tid / cols_here;
    const int b = blockIdx.y, slab = blockIdx.x;
    const int pps = HW / nslab;
    const int p0 = slab * pps, p1 = (slab == nslab - 1) ? HW : p0 + pps;
    const int c = (cb * 256 + col) * VEC;
    float s1[VEC], s2[VEC], piv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = 0.f; s2[e] = 0.f; piv[e] = 0.f; }
    if (row < rows) {
        const T* base = x + (long long)b * HW * xs + c;
        { uint4 u = *(const uint4*)(base + (long long)p0 * xs); TI<T>::unpack(u, piv); }
        for (int p = p0 + row; p < p1; p += rows) {
            const uint4 u = *(const uint4*)(base + (long long)p * xs);
            float f[VEC];
            TI<T>::unpack(u, f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = f[e] - piv[e]; s1[e] += d; s2[e] += d * d; }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[(tid * VEC + e) * 2] = s1[e]; red[(tid * VEC + e) * 2 + 1] = s2[e]; }
    __syncthreads();
    if (row == 0) {
        for (int r = 1; r < rows; ++r) {
            const int o = (r * cols_here + col) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { s1[e] += red[(o + e) * 2]; s2[e] += red[(o + e) * 2 + 1]; }
        }
        float4* dst = stats + ((long long)b * nslab + slab) * C + c;
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[e] = make_float4(piv[e], s1[e], s2[e], (float)(p1 - p0));
    }
}

// gn_group_load / gn_group_reduce / gn_group_stats, gn_scale_shift, gn_apply_vec: gn_group.h (shared with the conv epilogues' in-tile GroupNorm)

__global__ __launch_bounds__(64) void gn_finalize_kernel(const float4* __restrict__ st0, int nslab0, int C0, const float4* __restrict__ st1, int nslab1,
                                                         int C, int HW, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         float premul, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_rstd) {
    const int g = blockIdx.x, b = blockIdx.y;
    const int gw = C / 32;
    const int lane = threadIdx.x;
    const int cg0 = g * gw;
    float gam0 = 0.f, bet0 = 0.f;
    if (lane < gw) { gam0 = gamma[cg0 + lane]; bet0 = beta[cg0 + lane]; }
    float mean, rstd;
    gn_group_stats(st0, nslab0, C0, st1, nslab1, C, HW, eps, g, b, lane, mean, rstd);
    if (mean_rstd != nullptr && lane == 0) { mean_rstd[((long long)b * 32 + g) * 2] = mean; mean_rstd[((long long)b * 32 + g) * 2 + 1] = rstd; }
    for (int ci = lane; ci < gw; ci += 64) {
        const int c = cg0 + ci;
        const float gm = ci < 64 ? gam0 : gamma[c], bt = ci < 64 ? bet0 : beta[c];
        float sc, sh;
        gn_scale_shift(mean, rstd, gm, bt, sc, sh);
        scale[(long long)b * C + c] = sc * premul;
        shift[(long long)b * C + c] = sh * premul;
    }
}

// GroupNorm finalize + apply (+ SiLU, + channel concat) in ONE launch, for the consumers that normalise in a pass of their own (the 8 x 8 ResnetBlocks,
// the AttnBlocks).  Workgroup (q, image) owns FOUR consecutive groups: wave k finalises group 4 q + k with gn_group_stats (the bits of
// gn_finalize_kernel; the four reductions run side by side) and all 256 threads then apply scale / shift to the groups' channels of every pixel of
// the image with the arithmetic of gn_apply_kernel -- no statistics are reduced twice, and the channel runs of four groups (>= 128 bytes) keep the
// accesses on whole cache lines.  The thread's first vectors of x are requested BEFORE the statistics (they do not depend on them).
// History: a first form with a few fat workgroups per image, each repeating the image's whole reduction (16 waves x 2 groups), took 12.8 us against
// 5.7 + 5.9 us for finalize + apply (-1 % end to end); this form has no redundant work.
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_apply_kernel(const float4* __restrict__ st0, int nslab0, int C0, const float4* __restrict__ st1, int nslab1, int C,
                                                                int HW, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                const T* __restrict__ x0, int xs0, const T* __restrict__ x1, int xs1, T* __restrict__ y, int silu) {
    h16_mode_init<T>();
    constexpr int VEC = TI<T>::VEC;
    constexpr int GPW = 4;                            // groups per workgroup = waves per workgroup
    __shared__ float tab[2][GPW * 64];                // scale | shift of the workgroup's channels (group widths up to 64)
    const int q = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = C / 32;
    const int cw = GPW * gw, c_first = q * cw;        // the workgroup's channel range [c_first, c_first + cw)
    const int cols = cw / VEC;                        // 16-byte vectors per pixel in that range (C0, gw * GPW multiples of VEC: host check)
    // thread = (pixel phase, vector column), fixed for the whole loop: no division per vector (the 64-bit `id % cols`, `id / cols` of the first form were
    // ~200 instructions per vector, more than the arithmetic); 256 % cols threads idle (cols = 12: 4 of 256)
    const int ppi = blockDim.x / cols;                // pixels per iteration
    const int p0 = threadIdx.x / cols, col = threadIdx.x - p0 * cols;
    const bool live = p0 < ppi;
    const int c = c_first + col * VEC, cl = col * VEC;
    const T* xsrc = c < C0 ? x0 + ((long long)b * HW) * xs0 + c : x1 + ((long long)b * HW) * xs1 + (c - C0);
    const int xs = c < C0 ? xs0 : xs1;
    T* ydst = y + ((long long)b * HW) * C + c;
    constexpr int NPF = 4;
    uint4 xr[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k) { const int p = p0 + k * ppi; if (live && p < HW) xr[k] = *(const uint4*)(xsrc + (long long)p * xs); }
    {
        const int g = q * GPW + wave;
        float gam = 0.f, bet = 0.f;
        if (lane < gw) { gam = gamma[g * gw + lane]; bet = beta[g * gw + lane]; }
        float mean, rstd;
        gn_group_stats(st0, nslab0, C0, st1, nslab1, C, HW, eps, g, b, lane, mean, rstd);
        if (lane < gw) gn_scale_shift(mean, rstd, gam, bet, tab[0][wave * gw + lane], tab[1][wave * gw + lane]);
    }
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int k = 0; k < NPF; ++k) { const int p = p0 + k * ppi; if (p < HW) *(uint4*)(ydst + (long long)p * C) = gn_apply_vec<T>(xr[k], &tab[0][cl], &tab[1][cl], silu); }
    for (int p = p0 + NPF * ppi; p < HW; p += ppi) *(uint4*)(ydst + (long long)p * C) = gn_apply_vec<T>(*(const uint4*)(xsrc + (long long)p * xs), &tab[0][cl], &tab[1][cl], silu);
}

int k_gn_partial(const Tens& x, int B, float* stats, int nslab, int dtype, hipStream_t s) {
    const int vec = is_h16(dtype) ? 8 : 4;
    const int HW = x.H * x.W;
    if (x.C % vec) WDM_FAIL(WDM_EINVAL, "groupnorm: channel count %d must be a multiple of %d", x.C, vec);
    const int cols = x.C / vec;
    const dim3 grid(nslab, B, (cols + 255) / 256);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(gn_partial_kernel<H16>, grid, dim3(256), 0, s, (const H16*)x.p, x.xs, x.C, HW, nslab, (float4*)stats));
    else hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), 0, s, (const float*)x.p, x.xs, x.C, HW, nslab, (float4*)stats);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

int k_gn_finalize(int B, int HW, const float* st0, int nslab0, int C0, const float* st1, int nslab1, int C1, const NormW& nw, float eps, int for_silu_conv,
                  float* scale, float* shift, hipStream_t s, float* mean_rstd) {
    const int C = C0 + C1;
    if (C != nw.c || C % 32) WDM_FAIL(WDM_EINVAL, "groupnorm: %d channels vs %d weights (must be a multiple of 32)", C, nw.c);
    // for_silu_conv: the consumer is a conv with the fused GN+SiLU prologue, which wants scale/shift pre-multiplied by -log2(e)
    const float premul = for_silu_conv ? -1.4426950408889634f : 1.0f;
    const bool prof = prof_enabled();
    if (prof) {
        char name[64];
        snprintf(name, sizeof(name), "gn_finalize_kernel|%d px C=%d", HW, C);
        prof_begin(s, name, 0.0, 16.0 * B * (nslab0 * (double)C0 + (st1 ? nslab1 * (double)C1 : 0.0)) + 8.0 * B * C);
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, B), dim3(64), 0, s, (const float4*)st0, nslab0, C0, (const float4*)(st1 ? st1 : st0), st1 ? nslab1 : 1, C, HW,
                       nw.g, nw.b, eps, premul, scale, shift, mean_rstd);
    if (prof) prof_end(s);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// y (dense, C0 + C1 channels per pixel) = act(GroupNorm([x0 | x1])) from the tensors' partial statistics, one launch (gn_finalize_apply_kernel)
bool gn_fused_pass_eligible(int C0, int C1, int dtype) {
    const int vec = is_h16(dtype) ? 8 : 4;
    const int C = C0 + C1, gw = C / 32;
    // a 16-byte vector must not straddle the seam of the concat, and the four groups of a workgroup must be whole vectors
    return C % 32 == 0 && gw <= 64 && C0 % vec == 0 && C1 % vec == 0 && (4 * gw) % vec == 0;
}
int k_gn_finalize_apply(int B, const Tens& x0, const Tens* x1, const float* st0, int nslab0, const float* st1, int nslab1, const NormW& nw, float eps, int silu, void* y,
                        int dtype, hipStream_t s) {
    const int C0 = x0.C, C1 = x1 ? x1->C : 0, C = C0 + C1, HW = x0.H * x0.W;
    if (C != nw.c || !gn_fused_pass_eligible(C0, C1, dtype)) WDM_FAIL(WDM_EINVAL, "groupnorm: %d + %d channels vs %d weights unsupported by the fused pass", C0, C1, nw.c);
    const dim3 grid(8, B);
    const bool prof = prof_enabled();
    if (prof) {
        const double es = is_h16(dtype) ? 2.0 : 4.0;
        char name[64];
        snprintf(name, sizeof(name), "gn_finalize_apply_kernel|%dx%d C=%d%s", x0.H, x0.W, C, silu ? " silu" : "");
        prof_begin(s, name, 0.0, 2.0 * B * HW * C * es + 16.0 * B * (nslab0 * (double)C0 + (st1 ? nslab1 * (double)C1 : 0.0)));
    }
    if (is_h16(dtype))
        WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(gn_finalize_apply_kernel<H16>, grid, dim3(256), 0, s, (const float4*)st0, nslab0, C0, (const float4*)(st1 ? st1 : st0), st1 ? nslab1 : 1, C, HW, nw.g,
                           nw.b, eps, (const H16*)x0.p, x0.xs, (const H16*)(x1 ? x1->p : x0.p), x1 ? x1->xs : x0.xs, (H16*)y, silu));
    else
        hipLaunchKernelGGL(gn_finalize_apply_kernel<float>, grid, dim3(256), 0, s, (const float4*)st0, nslab0, C0, (const float4*)(st1 ? st1 : st0), st1 ? nslab1 : 1, C, HW, nw.g,
                           nw.b, eps, (const float*)x0.p, x0.xs, (const float*)(x1 ? x1->p : x0.p), x1 ? x1->xs : x0.xs, (float*)y, silu);
    if (prof) prof_end(s);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// GroupNorm apply, optionally followed by SiLU: y = act(x*scale + shift).  Output pixel stride ys / channel offset via the y
// pointer, scale/shift rows of length sc_ld: two calls materialise the channel concat of two tensors (AttnBlock.norm,
// unet.py:169-170; the normalised+activated conv input of the 8x8 ResnetBlocks, unet.py:121-123 / 130-133).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int xs, int C, int HW, long long nvec, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int sc_ld, T* __restrict__ y, int ys, int silu) {
    h16_mode_init<T>();
    constexpr int VEC = TI<T>::VEC;
    const int cols = C / VEC;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nvec; id += (long long)gridDim.x * blockDim.x) {
        const int col = (int)(id % cols);
        const long long bp = id / cols;
        const long long b = bp / HW;
        const int c = col * VEC;
        const uint4 u = *(const uint4*)(x + bp * xs + c);
        *(uint4*)(y + bp * ys + c) = gn_apply_vec<T>(u, scale + b * sc_ld + c, shift + b * sc_ld + c, silu);       // the arithmetic of the fused pass (gn_group.h)
    }
}
int k_gn_apply(const Tens& x, int B, const float* scale, const float* shift, int sc_ld, void* y, int y_stride, int y_choff, int silu, int dtype,
               hipStream_t s) {
    const int HW = x.H * x.W;
    const int vec = is_h16(dtype) ? 8 : 4;
    const long long nvec = (long long)B * HW * (x.C / vec);
    const int g = nblocks(nvec, 256) > 16384 ? 16384 : nblocks(nvec, 256);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(gn_apply_kernel<H16>, dim3(g), dim3(256), 0, s, (const H16*)x.p, x.xs, x.C, HW, nvec, scale, shift, sc_ld, (H16*)y + y_choff, y_stride, silu));
    else hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x.p, x.xs, x.C, HW, nvec, scale, shift, sc_ld, (float*)y + y_choff, y_stride, silu);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// =================================================================================================
// attention softmax over keys (unet.py:179): one wave per query row, wavefront-shuffle max / sum reductions
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, T* __restrict__ P, long long rows, int n) {
    h16_mode_init<T>();
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* src = S + row * n;
    float v[8];
    const int per = n / 64;    // n in {64, 128, 256, 512}
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < per) { v[i] = src[lane + i * 64]; mx = fmaxf(mx, v[i]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < per) { v[i] = expf(v[i] - mx); sum += v[i]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < per) TI<T>::st(P, row * n + lane + i * 64, v[i] * inv);
}
int k_softmax_rows(const float* S, void* P, long long rows, int n, int dtype, hipStream_t s) {
    if (n % 64 || n > 512 || n <= 0) WDM_FAIL(WDM_EINVAL, "softmax: row length %d must be a multiple of 64 and <= 512", n);
    const int g = (int)((rows + 3) / 4);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(softmax_rows_kernel<H16>, dim3(g), dim3(256), 0, s, S, (H16*)P, rows, n));
    else hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(g), dim3(256), 0, s, S, (float*)P, rows, n);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// =================================================================================================
// timestep embedding + Linear layers of the temb path (unet.py:10-28, 354-357, 125), all fp32
// =================================================================================================
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int n_t, int dim, float* __restrict__ emb) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dim / 2;
    if (id >= n_t * half) return;
    const int n = id / half, i = id % half;
    const float w = expf((float)i * -(logf(10000.0f) / (float)(half - 1)));
    const float a = t[n] * w;
    emb[n * dim + i] = sinf(a);
    emb[n * dim + half + i] = cosf(a);
}
int k_timestep_embedding(const float* t, int n_t, int dim, float* emb, hipStream_t s) {
    if (dim % 2 || dim < 4) WDM_FAIL(WDM_EINVAL, "timestep embedding: dim %d must be even", dim);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(nblocks((long long)n_t * dim / 2, 64)), dim3(64), 0, s, t, n_t, dim, emb);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ in, int n, int k, const float* __restrict__ W, const float* __restrict__ bias,
                                                     int o, float* __restrict__ out, int act) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)n * o) return;
    const int row = (int)(wid / o), oc = (int)(wid % o);
    const float* x = in + (long long)row * k;
    const float* w = W + (long long)oc * k;
    float acc = 0.f;
    for (int i = lane; i < k; i += 64) {
        float v = x[i];
        if (act == 1) v = v / (1.0f + expf(-v));
        acc += v * w[i];
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) {
        float r = acc + (bias ? bias[oc] : 0.f);
        if (act == 2) r = r / (1.0f + expf(-r));
        out[(long long)row * o + oc] = r;
    }
}
int k_linear(const float* in, int n, int k, const float* W, const float* b, int o, float* out, int act, hipStream_t s) {
    const long long waves = (long long)n * o;
    hipLaunchKernelGGL(linear_kernel, dim3((int)((waves + 3) / 4)), dim3(256), 0, s, in, n, k, W, b, o, out, act);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

// =================================================================================================
// weight packing: OIHW f32 -> [tap][rows_total][cin] in the model dtype
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, int cout, int cin, int cin_dst, int kk, T* __restrict__ dst, int rows_total,
                                                        int row_off, int rows_span) {
    // rows [row_off, row_off + cout) get the weights, rows [row_off + cout, row_off + rows_span) are zero padding; so are the
    // columns [cin, cin_dst) (conv_in of models whose input channel count is not a multiple of the K slab)
    const long long total = (long long)rows_span * cin_dst * kk;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(id % cin_dst);
        const int o = (int)((id / cin_dst) % rows_span);
        const int tap = (int)(id / ((long long)cin_dst * rows_span));
        const float v = (o < cout && ci < cin) ? w[((long long)o * cin + ci) * kk + tap] : 0.f;
        TI<T>::st(dst, ((long long)tap * rows_total + row_off + o) * cin_dst + ci, v);
    }
}
int k_pack_conv(const float* w_oihw, int cout, int cin, int k, void* dst, int rows_total, int row_off, int zero_tail, int dtype, hipStream_t s, int cin_dst) {
    if (cin_dst <= 0) cin_dst = cin;
    const int rows_span = zero_tail ? rows_total - row_off : cout;
    const long long total = (long long)rows_span * cin_dst * k * k;
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(pack_conv_kernel<H16>, dim3(g), dim3(256), 0, s, w_oihw, cout, cin, cin_dst, k * k, (H16*)dst, rows_total, row_off, rows_span));
    else hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(g), dim3(256), 0, s, w_oihw, cout, cin, cin_dst, k * k, (float*)dst, rows_total, row_off, rows_span);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_sm_kernel(const float* __restrict__ w, int cout, int cin, T* __restrict__ dst, int rows_total, int taps) {
    const long long total = (long long)taps * rows_total * cin;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id & 31);
        const int o = (int)((id >> 5) % rows_total);
        const int tap = (int)((id / ((long long)32 * rows_total)) % taps);
        const int slab = (int)(id / ((long long)32 * rows_total * taps));
        const float v = o < cout ? w[((long long)o * cin + slab * 32 + c) * taps + tap] : 0.f;
        TI<T>::st(dst, id, v);
    }
}
// f32x3 mode (conv_dmax3_kernel.h): OIHW f32 3x3 -> [tap][rows_total][cin] with every 16-channel group (64 bytes) already split and laid out as the kernel's
// LDS rows: [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15] bf16 (hi = RNE(w), lo = RNE(w - hi)) -- the kernel then DMAs weight sub-stages ready to multiply
__global__ __launch_bounds__(256) void pack_conv_x3_kernel(const float* __restrict__ w, int cout, int cin, unsigned* __restrict__ dst, int rows_total) {
    const int upr = cin / 4;                                     // 4-channel units per row
    const long long total = (long long)9 * rows_total * upr;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int u = (int)(id % upr);
        const int o = (int)((id / upr) % rows_total);
        const int tap = (int)(id / ((long long)upr * rows_total));
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (o < cout) for (int e = 0; e < 4; ++e) x[e] = w[((long long)o * cin + u * 4 + e) * 9 + tap];
        unsigned hi[2], lo[2];
        for (int h = 0; h < 2; ++h) {
            const unsigned ph = TI<__bf16>::pack2(x[2 * h], x[2 * h + 1]);
            hi[h] = ph;
            lo[h] = TI<__bf16>::pack2(x[2 * h] - __uint_as_float(ph << 16), x[2 * h + 1] - __uint_as_float(ph & 0xffff0000u));
        }
        const int grp = u >> 2, uu = u & 3;
        unsigned* row = dst + ((long long)tap * rows_total + o) * cin + grp * 16;      // dwords: one per fp32 element of the plain matrix
        unsigned* ph = row + (uu >> 1) * 4 + (uu & 1) * 2;
        ph[0] = hi[0]; ph[1] = hi[1]; ph[8] = lo[0]; ph[9] = lo[1];
    }
}
int k_pack_conv_sm(const float* w_oihw, int cout, int cin, void* dst, int rows_total, hipStream_t s, int dtype, int k) {
    if (dtype == WDM_F32X3) {
        if (k != 3) WDM_FAIL(WDM_EINVAL, "k_pack_conv_sm: the f32x3 pre-split copy is for 3x3 convs");
        if (cin % 16) WDM_FAIL(WDM_EINVAL, "k_pack_conv_sm: cin %d is not a multiple of 16", cin);
        const long long total = (long long)9 * rows_total * (cin / 4);
        hipLaunchKernelGGL(pack_conv_x3_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 65535)), dim3(256), 0, s, w_oihw, cout, cin, (unsigned*)dst, rows_total);
        WDM_HIP(hipGetLastError());
        return WDM_OK;
    }

    if (cin % 32) WDM_FAIL(WDM_EINVAL, "k_pack_conv_sm: cin %d is not a multiple of 32", cin);
    const long long total = (long long)k * k * rows_total * cin;
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(pack_conv_sm_kernel<H16>, dim3(g), dim3(256), 0, s, w_oihw, cout, cin, (H16*)dst, rows_total, k * k));
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
// Upsample in sub-pixel form (conv_up4_kernel.h): OIHW f32 3x3 -> SLAB-MAJOR [cin / 32][phase = 2 py + px][dy'][dx'][rows_total][32] 16-bit (round 5: every 1 KB
// LDS-DMA piece of the kernel -- 16 rows x 64 B -- is then one run of whole cache lines; as [tap][row][cin] it was 16 half lines a row apart), the taps of the
// upsampled grid that fall on the same low-resolution pixel summed in fp32:  py = 0: {w0}, {w1 + w2};  py = 1: {w0 + w1}, {w2}
template <typename T>
__global__ __launch_bounds__(256) void pack_up4_kernel(const float* __restrict__ w, int cout, int cin, T* __restrict__ dst, int rows_total) {
    const long long total = (long long)16 * rows_total * cin;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(id % cin);
        const int o = (int)((id / cin) % rows_total);
        const int t = (int)(id / ((long long)cin * rows_total));       // phase * 4 + dy' * 2 + dx'
        const int py = t >> 3, px = (t >> 2) & 1, dyl = (t >> 1) & 1, dxl = t & 1;
        float v = 0.f;
        if (o < cout) {
            const float* p = w + ((long long)o * cin + ci) * 9;
            const int y0 = py == 0 ? (dyl == 0 ? 0 : 1) : (dyl == 0 ? 0 : 2), y1 = py == 0 ? (dyl == 0 ? 0 : 2) : (dyl == 0 ? 1 : 2);
            const int x0 = px == 0 ? (dxl == 0 ? 0 : 1) : (dxl == 0 ? 0 : 2), x1 = px == 0 ? (dxl == 0 ? 0 : 2) : (dxl == 0 ? 1 : 2);
            for (int ty = y0; ty <= y1; ++ty)
                for (int tx = x0; tx <= x1; ++tx) v += p[ty * 3 + tx];
        }
        TI<T>::st(dst, (((long long)(ci >> 5) * 16 + t) * rows_total + o) * 32 + (ci & 31), v);
    }
}
// f32x3 mode (conv_up4x3_kernel.h): the same 16 pre-summed taps (fp32 sums), every 16-channel group split and laid out as [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]
__global__ __launch_bounds__(256) void pack_up4_x3_kernel(const float* __restrict__ w, int cout, int cin, unsigned* __restrict__ dst, int rows_total) {
    const int upr = cin / 4;
    const long long total = (long long)16 * rows_total * upr;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int u = (int)(id % upr);
        const int o = (int)((id / upr) % rows_total);
        const int t = (int)(id / ((long long)upr * rows_total));       // phase * 4 + dy' * 2 + dx'
        const int py = t >> 3, px = (t >> 2) & 1, dyl = (t >> 1) & 1, dxl = t & 1;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (o < cout) {
            const int y0 = py == 0 ? (dyl == 0 ? 0 : 1) : (dyl == 0 ? 0 : 2), y1 = py == 0 ? (dyl == 0 ? 0 : 2) : (dyl == 0 ? 1 : 2);
            const int x0 = px == 0 ? (dxl == 0 ? 0 : 1) : (dxl == 0 ? 0 : 2), x1 = px == 0 ? (dxl == 0 ? 0 : 2) : (dxl == 0 ? 1 : 2);
            for (int e = 0; e < 4; ++e) {
                const float* p = w + ((long long)o * cin + u * 4 + e) * 9;
                float v = 0.f;
                for (int ty = y0; ty <= y1; ++ty)
                    for (int tx = x0; tx <= x1; ++tx) v += p[ty * 3 + tx];
                x[e] = v;
            }
        }
        unsigned hi[2], lo[2];
        for (int h = 0; h < 2; ++h) {
            const unsigned ph = TI<__bf16>::pack2(x[2 * h], x[2 * h + 1]);
            hi[h] = ph;
            lo[h] = TI<__bf16>::pack2(x[2 * h] - __uint_as_float(ph << 16), x[2 * h + 1] - __uint_as_float(ph & 0xffff0000u));
        }
        const int grp = u >> 2, uu = u & 3;
        unsigned* q = dst + ((long long)t * rows_total + o) * cin + grp * 16 + (uu >> 1) * 4 + (uu & 1) * 2;
        q[0] = hi[0]; q[1] = hi[1]; q[8] = lo[0]; q[9] = lo[1];
    }
}
int k_pack_up4(const float* w_oihw, int cout, int cin, void* dst, int rows_total, hipStream_t s, int dtype) {
    if (dtype == WDM_F32X3) {
        if (cin % 16) WDM_FAIL(WDM_EINVAL, "k_pack_up4: cin %d is not a multiple of 16", cin);
        const long long tot = (long long)16 * rows_total * (cin / 4);
        hipLaunchKernelGGL(pack_up4_x3_kernel, dim3(nblocks(tot, 256) > 16384 ? 16384 : nblocks(tot, 256)), dim3(256), 0, s, w_oihw, cout, cin, (unsigned*)dst, rows_total);
        WDM_HIP(hipGetLastError());
        return WDM_OK;
    }
    if (cin % 32) WDM_FAIL(WDM_EINVAL, "k_pack_up4: cin %d is not a multiple of 32", cin);
    const long long total = (long long)16 * rows_total * cin;
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(pack_up4_kernel<H16>, dim3(g), dim3(256), 0, s, w_oihw, cout, cin, (H16*)dst, rows_total));
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
// y[row][0..Cp) = x[row][0..C) followed by zeros (dense rows)
template <typename T>
__global__ __launch_bounds__(256) void pad_channels2_kernel(const T* __restrict__ x, int C, int Cp, T* __restrict__ y, long long total) {
    h16_mode_init<T>();
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % Cp);
        const long long r = id / Cp;
        TI<T>::st(y, id, c < C ? TI<T>::ld(x, r * C + c) : 0.f);
    }
}
int k_pad_channels(const void* x, int C, int Cp, void* y, long long rows, int dtype, hipStream_t s) {
    const long long total = rows * Cp;
    const int g = nblocks(total, 256) > 16384 ? 16384 : nblocks(total, 256);
    if (is_h16(dtype)) WDM_H16_SWITCH(dtype, hipLaunchKernelGGL(pad_channels2_kernel<H16>, dim3(g), dim3(256), 0, s, (const H16*)x, C, Cp, (H16*)y, total));
    else hipLaunchKernelGGL(pad_channels2_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)x, C, Cp, (float*)y, total);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
__global__ __launch_bounds__(256) void flag_out_of_range_kernel(const float* __restrict__ x, long long n, float limit, int* __restrict__ flag) {
    bool bad = false;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) bad = bad || !(fabsf(x[id]) <= limit);      // NaN compares false
    if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1;          // every writer stores the same value: no atomic needed
}
int k_flag_out_of_range(const float* x, long long n, float limit, int* flag_dev, hipStream_t s) {
    const int g = nblocks(n, 256) > 1024 ? 1024 : nblocks(n, 256);
    hipLaunchKernelGGL(flag_out_of_range_kernel, dim3(g), dim3(256), 0, s, x, n, limit, flag_dev);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
int k_copy_f32(const float* src, float* dst, long long n, hipStream_t s) {
    WDM_HIP(hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return WDM_OK;
}

// ---- AttnBlock operand folding (set-up path: runs when a parameter of an AttnBlock is loaded; common.h: k_attn_fold) -------------------------------------
// out[r][c] = sum_o A(o, r) B[o][c]  (TA: A^T B)   |   sum_m A[r][m] B[m][c]  (A B); C x C fp32 matrices, fp64 sums in a fixed (ascending) order
template <bool TA>
__global__ __launch_bounds__(256) void fold_mm_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ out, int C) {
    __shared__ float sa[16][17], sb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r = blockIdx.y * 16 + ty, c = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int k0 = 0; k0 < C; k0 += 16) {
        // sa[kk][rr] = A(k0 + kk, r0 + rr) (TA) or A[r0 + rr][k0 + kk]
        const int ar = blockIdx.y * 16 + (TA ? tx : ty), ak = k0 + (TA ? ty : tx);
        sa[TA ? ty : tx][TA ? tx : ty] = (ar < C && ak < C) ? (TA ? A[(size_t)ak * C + ar] : A[(size_t)ar * C + ak]) : 0.f;
        sb[ty][tx] = (k0 + ty < C && c < C) ? Bm[(size_t)(k0 + ty) * C + c] : 0.f;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc += (double)sa[kk][ty] * (double)sb[kk][tx];
        __syncthreads();
    }
    if (r < C && c < C) out[(size_t)r * C + c] = (float)acc;
}
// out[r] = sum_o A(o, r) x[o] (TA) | sum_m A[r][m] x[m], + add[r] when given
template <bool TA>
__global__ __launch_bounds__(64) void fold_mv_kernel(const float* __restrict__ A, const float* __restrict__ x, const float* __restrict__ add, float* __restrict__ out, int C) {
    const int r = blockIdx.x, lane = threadIdx.x;
    double acc = 0.0;
    for (int k = lane; k < C; k += 64) acc += (double)(TA ? A[(size_t)k * C + r] : A[(size_t)r * C + k]) * (double)x[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) out[r] = (float)(acc + (add ? (double)add[r] : 0.0));
}
int k_attn_fold(const float* wq, const float* bq, const float* wk, const float* wv, const float* bv, const float* wp, const float* bp, int C, float* M, float* cq, float* Wvp,
                float* bvp, hipStream_t s) {
    const dim3 g((C + 15) / 16, (C + 15) / 16);
    if (M) {
        hipLaunchKernelGGL(fold_mm_kernel<true>, g, dim3(256), 0, s, wk, wq, M, C);                          // M[c'][c] = sum_o Wk[o][c'] Wq[o][c]
        hipLaunchKernelGGL(fold_mv_kernel<true>, dim3(C), dim3(64), 0, s, wk, bq, (const float*)nullptr, cq, C);      // cq[c'] = sum_o Wk[o][c'] bq[o]
    }
    if (Wvp) {
        hipLaunchKernelGGL(fold_mm_kernel<false>, g, dim3(256), 0, s, wp, wv, Wvp, C);                       // Wvp[o][c] = sum_m Wp[o][m] Wv[m][c]
        hipLaunchKernelGGL(fold_mv_kernel<false>, dim3(C), dim3(64), 0, s, wp, bv, bp, bvp, C);              // bvp[o] = sum_m Wp[o][m] bv[m] + bp[o]
    }
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

}  // namespace wdm
