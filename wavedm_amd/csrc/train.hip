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
#include <atomic>

#include "common.h"
#include "conv_wgrad_kernel.h"

namespace wdm {

static inline int nblk(long long n, int bs) { long long g = (n + bs - 1) / bs; return (int)(g > 16384 ? 16384 : g); }

// dst[b][c][k] (row length kp, k = oy*Wo + ox; image stride dst_img elements) = src[b][stride*oy + off_y][stride*ox + off_x][c]
// (0 outside the map / past Ho*Wo), src = the channel concat [src0 (C0 channels) | src1]; rows C <= c < Crows are written as zeros.
// A 64-channel x 64-position tile per workgroup goes through LDS, so the reads run along the channels of a pixel and the writes along the
// positions of a channel: both coalesced.
// Bg > 1: the images are laid out in groups of Bg along the row -- dst[b / Bg][c][b % Bg][k], dst_img = elements per GROUP -- so that a GEMM
// over a row contracts the pixels of Bg images at once (conv_wgrad).  ndx = 3: blockIdx.z also runs over three copies shifted by
// off_x - 1, off_x, off_x + 1 columns, dst_dx elements apart (the dx taps of a 3x3 wgrad).
template <typename T>
__global__ __launch_bounds__(256) void gather_t_kernel(const T* __restrict__ src0, int xs0, int C0, const T* __restrict__ src1, int xs1, int C, int Crows, int H, int W,
                                                       int Ho, int Wo, int stride, int off_y, int off_x, T* __restrict__ dst, long long dst_img, int kp, int Bg, int B,
                                                       long long dst_dx, float* __restrict__ csum) {
    __shared__ float tile[64][65];
    __shared__ float cred[4][64];
    const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int dxi = blockIdx.z / B;
    const long long b = blockIdx.z - dxi * B;
    if (gridDim.z > (unsigned)B) { off_x += dxi - 1; dst += dxi * dst_dx; }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (c0 < C) {
        const int c = c0 + tx;
        for (int j = ty; j < 64; j += 4) {
            const int k = k0 + j;
            float v = 0.f;
            if (c < C && k < Ho * Wo) {
                const int oy = k / Wo, ox = k - oy * Wo;
                const int y = stride * oy + off_y, x = stride * ox + off_x;
                if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                    const long long px = (b * H + y) * W + x;
                    v = c < C0 ? TI<T>::ld(src0, px * xs0 + c) : TI<T>::ld(src1, px * xs1 + (c - C0));
                }
            }
            tile[j][tx] = v;
        }
    }
    __syncthreads();
    if (csum != nullptr && c0 < C) {                  // column sums of this tile: csum[b][k tile][c] (bias / temb gradients, summed later in tile order)
        float a4 = 0.f;
        for (int j = ty; j < 64; j += 4) a4 += tile[j][tx];
        cred[ty][tx] = a4;
        __syncthreads();
        if (ty == 0 && c0 + tx < C) csum[((long long)b * gridDim.x + blockIdx.x) * C + c0 + tx] = (cred[0][tx] + cred[1][tx]) + (cred[2][tx] + cred[3][tx]);
    }
    const int k = k0 + tx;
    if (k >= kp) return;
    for (int j = ty; j < 64; j += 4) {
        const int c = c0 + j;
        if (c >= Crows) break;
        TI<T>::st(dst, (b / Bg) * dst_img + ((long long)c * Bg + (b % Bg)) * kp + k, c0 < C ? tile[tx][j] : 0.f);
    }
}
// The same gather for bf16 tensors whose channel counts and strides are multiples of 8 (every layer of the model): 16-byte loads along the channels of a
// pixel, 16-byte stores along the positions of a channel -- the element-wise form above moves two bytes per memory instruction and was 24 % of a training
// step.  Same tile, same fp32 column sums in the same order.
__global__ __launch_bounds__(256) void gather_t_bf16x8_kernel(const __bf16* __restrict__ src0, int xs0, int C0, const __bf16* __restrict__ src1, int xs1, int C, int Crows,
                                                              int H, int W, int Ho, int Wo, int stride, int off_y, int off_x, __bf16* __restrict__ dst, long long dst_img,
                                                              int kp, int Bg, int B, long long dst_dx, float* __restrict__ csum) {
    __shared__ float tile[64][65];
    __shared__ float cred[4][64];
    const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int dxi = blockIdx.z / B;
    const long long b = blockIdx.z - dxi * B;
    if (gridDim.z > (unsigned)B) { off_x += dxi - 1; dst += dxi * dst_dx; }
    const int tid = threadIdx.x;
    if (c0 < C) {
        const int cv = tid & 7, kl = tid >> 3;                  // 8 channel vectors x 32 positions per pass: bank (kl + 8 cv + e) % 64, conflict-free
        const int c = c0 + cv * 8;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int j = kl + pass * 32, k = k0 + j;
            uint4 u = make_uint4(0u, 0u, 0u, 0u);
            if (c < C && k < Ho * Wo) {
                const int oy = k / Wo, ox = k - oy * Wo;
                const int y = stride * oy + off_y, x = stride * ox + off_x;
                if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                    const long long px = (b * H + y) * W + x;
                    u = c < C0 ? *(const uint4*)(src0 + px * xs0 + c) : *(const uint4*)(src1 + px * xs1 + (c - C0));
                }
            }
            float f[8];
            TI<__bf16>::unpack(u, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[j][cv * 8 + e] = f[e];
        }
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    if (csum != nullptr && c0 < C) {
        float a4 = 0.f;
        for (int j = ty; j < 64; j += 4) a4 += tile[j][tx];
        cred[ty][tx] = a4;
        __syncthreads();
        if (ty == 0 && c0 + tx < C) csum[((long long)b * gridDim.x + blockIdx.x) * C + c0 + tx] = (cred[0][tx] + cred[1][tx]) + (cred[2][tx] + cred[3][tx]);
    }
    const int kv = tid & 7, cl = tid >> 3;                      // 8 position vectors x 32 channel rows per pass: bank (8 kv + e + cl) % 64, conflict-free
    const int k = k0 + kv * 8;
    if (k >= kp) return;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int j = cl + pass * 32, c = c0 + j;
        if (c >= Crows) break;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = c0 < C ? tile[kv * 8 + e][j] : 0.f;
        *(uint4*)(dst + (b / Bg) * dst_img + ((long long)c * Bg + (b % Bg)) * kp + k) = TI<__bf16>::pack(f);
    }
}
// grad[co][ci][tap] (OIHW f32) (+)= sum_b partial[tap][b][co][ci]   (rows_g rows per image in the partial buffer), b ascending.
// A thread owns (tap row, co, ci): reads coalesced along ci, eight partials in flight at a time, three consecutive floats of the OIHW row written per thread
// (the first form, a thread per (tap, co, ci) with one dependent load per partial and 36-byte-strided single stores, was 8 % of a training step).
__global__ __launch_bounds__(256) void reduce_wgrad_kernel(const float* __restrict__ part, int taps, int B, int rows_g, int cout, int cin, float* __restrict__ grad,
                                                           int accumulate) {
    const int tpr = taps == 9 ? 3 : 1, ntr = taps / tpr;           // taps per thread, tap rows
    const long long total = (long long)ntr * cout * cin;
    const long long img = (long long)rows_g * cin;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(id % cin);
        const long long t2 = id / cin;
        const int co = (int)(t2 % cout);
        const int tr = (int)(t2 / cout);
        for (int k = 0; k < tpr; ++k) {
            const int tap = tr * tpr + k;
            const float* p = part + (long long)tap * B * img + (long long)co * cin + ci;
            float s = 0.f;
            int b = 0;
            for (; b + 8 <= B; b += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = p[(long long)(b + u) * img];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; b < B; ++b) s += p[(long long)b * img];
            const long long o = ((long long)co * cin + ci) * taps + tap;
            grad[o] = accumulate ? grad[o] + s : s;
        }
    }
}
// out[g][c] (+)= sum over the rows of group g of x[row][c]; rows_per_group rows per group (bias grad: one group; temb grad: one per image).
// Two stages, both in a fixed order: blocks of 64 channels x 4 row slices sum a chunk of rows each, then the chunks are added up.
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ x, int xs, int C, long long rows_per_group, int chunk_rows, int nchunks,
                                                          float* __restrict__ part) {
    __shared__ float red[256];
    const int g = blockIdx.y, ch = blockIdx.z;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    const long long r0 = (long long)ch * chunk_rows, r1 = r0 + chunk_rows < rows_per_group ? r0 + chunk_rows : rows_per_group;
    float s = 0.f;
    if (c < C)
        for (long long r = r0 + sl; r < r1; r += 4) s += TI<T>::ld(x, (g * rows_per_group + r) * xs + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && c < C) part[((long long)g * nchunks + ch) * C + c] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}
// 64 channels x 4 chunk slices per workgroup; the four slice sums are added in a fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int part_ld, int C, int nchunks, int groups, float* __restrict__ out,
                                                           int out_ld, int accumulate) {
    __shared__ float red[256];
    const int g = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int k = sl; k < nchunks; k += 4) s += part[((long long)g * nchunks + k) * part_ld + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && c < C) {
        const float t = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
        const long long o = (long long)g * out_ld + c;
        out[o] = accumulate ? out[o] + t : t;
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
// dgrad weights: dst[tap'][row = ci][k = co] = w[co][ci][taps-1-tap'] (transposed, taps mirrored), k zero-padded to kpad, rows to rows_total.
// One workgroup per 32 (co) x 32 (ci) block: the OIHW rows are read contiguously (32 ci x taps floats per co) into LDS, each tap is
// written as 32-element runs along co.
template <typename T, int KK>
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ w, int cout, int cin, T* __restrict__ dst, int rows_total, int kpad) {
    constexpr int ROW = 32 * KK;
    __shared__ float sm[32][ROW + 1];
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 32;
    const int nci = min(32, cin - ci0);             // may be <= 0 for the zero-padded rows
    for (int r = threadIdx.x >> 5; r < 32; r += 8) {
        const int co = co0 + r;
        for (int e = threadIdx.x & 31; e < ROW; e += 32)
            sm[r][e] = (co < cout && e < nci * KK) ? w[((long long)co * cin + ci0) * KK + e] : 0.f;
    }
    __syncthreads();
    const int co = co0 + (threadIdx.x & 31);
    if (co >= kpad) return;
    for (int tp = 0; tp < KK; ++tp)
        for (int r = threadIdx.x >> 5; r < 32; r += 8) {
            const int ci = ci0 + r;
            if (ci < rows_total) TI<T>::st(dst, ((long long)tp * rows_total + ci) * kpad + co, sm[threadIdx.x & 31][r * KK + (KK - 1 - tp)]);
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

// ---- GroupNorm (+ SiLU) backward (unet.py:31-37 under autograd), three passes with full-row 16-byte accesses:
//   xh = (x - mean) rstd,  pre = xh g + b,  y = silu(pre) (or pre);   dv = dy * silu'(pre)
//   (1) sums:     per (image, pixel slab, channel)  sum dv xh  and  sum dv                       [gn_bwd_sums_kernel]
//   (2) finalize: per (image, group) add the slabs in a fixed order -> dgamma_c, dbeta_c of the image (summed over the batch afterwards) and the
//                 group means  ma = sum_c g_c dbeta_c / N,  mb = sum_c g_c dgamma_c / N                [gn_bwd_finalize_kernel, gn_bwd_param_kernel]
//   (3) apply:    dx = rstd (dv g - ma - xh mb)                                                    [gn_bwd_apply_kernel]
// x = [x0 | x1] (channel concat), dy dense [B][HW][C]; dx0 / dx1 dense per source, optionally accumulated into.
// grid (nslab, B, column blocks): 256 threads = (pixel rows) x (16-byte channel vectors); partial[b][slab][c] = {sum dv xh, sum dv}
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const T* __restrict__ x0, int xs0, int C0, const T* __restrict__ x1, int xs1, int C, int HW, int nslab,
                                                          const T* __restrict__ dy, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean_rstd, int silu, float2* __restrict__ partial) {
    constexpr int VEC = TI<T>::VEC;
    __shared__ float red[256 * VEC * 2];
    const int cols = C / VEC;
    const int cb = blockIdx.z;
    const int cols_here = min(cols - cb * 256, 256);
    const int rows = 256 / cols_here;
    const int tid = threadIdx.x;
    const int col = tid % cols_here, row = tid / cols_here;
    const int b = blockIdx.y, slab = blockIdx.x;
    const int pps = HW / nslab;
    const int p0 = slab * pps, p1 = (slab == nslab - 1) ? HW : p0 + pps;
    const int c = (cb * 256 + col) * VEC;
    const int gw = C / 32;
    float sg[VEC], sb[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sg[e] = 0.f; sb[e] = 0.f; }
    if (row < rows) {
        // the vector's statistics, gamma and beta are loop constants: in registers, per element (a vector may straddle a group boundary when gw % VEC != 0)
        float gm[VEC], bt[VEC], mean[VEC], rstd[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int ge = (c + e) / gw;
            gm[e] = gamma[c + e]; bt[e] = beta[c + e];
            mean[e] = mean_rstd[((long long)b * 32 + ge) * 2]; rstd[e] = mean_rstd[((long long)b * 32 + ge) * 2 + 1];
        }
        const bool first = c < C0;
        const T* xp = first ? x0 + c : x1 + (c - C0);
        const int xs = first ? xs0 : xs1;
        for (int p = p0 + row; p < p1; p += rows) {
            float xh[VEC], dv[VEC];
            const long long bp = (long long)b * HW + p;
            const uint4 ux = *(const uint4*)(xp + bp * xs);
            const uint4 ud = *(const uint4*)(dy + bp * C + c);
            TI<T>::unpack(ux, xh);
            TI<T>::unpack(ud, dv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                xh[e] = (xh[e] - mean[e]) * rstd[e];
                if (silu) { const float pre = xh[e] * gm[e] + bt[e]; const float sgm = 1.0f / (1.0f + __expf(-pre)); dv[e] *= sgm * (1.0f + pre * (1.0f - sgm)); }
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) { sg[e] += dv[e] * xh[e]; sb[e] += dv[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[(tid * VEC + e) * 2] = sg[e]; red[(tid * VEC + e) * 2 + 1] = sb[e]; }
    __syncthreads();
    if (row == 0) {
        for (int r = 1; r < rows; ++r) {
            const int o = (r * cols_here + col) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { sg[e] += red[(o + e) * 2]; sb[e] += red[(o + e) * 2 + 1]; }
        }
        float2* dst = partial + ((long long)b * nslab + slab) * C + c;
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[e] = make_float2(sg[e], sb[e]);
    }
}
// Finalize in two launches (round 4: ONE workgroup per group walking B x gw channels x nslab slabs serially was 48 us on 32 of 256 CUs, 5 % of a step):
//   (2a) grid (32 groups, B images), 256 threads: a channel's slabs are summed by eight lanes (slab sl goes to lane sl % 8, ascending within a lane; the eight
//        partial sums are joined by a fixed xor tree) -> dgamma_c, dbeta_c of the image; then the image's group means (mab), channels in ascending order;
//   (2b) one thread per channel and parameter: the batch sum over the images in ascending order.
// Every order is fixed: deterministic.
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float2* __restrict__ partial, int nslab, int B, int C, int HW, const float* __restrict__ gamma,
                                                             float* __restrict__ dgam_part, float* __restrict__ dbet_part, float* __restrict__ mab) {
    __shared__ float sgs[64], sbs[64];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int gw = C / 32, cg0 = g * gw;                        // gw <= 64 (C <= 2048)
    for (int i = tid; i < ((gw * 8 + 63) & ~63); i += 256) {    // whole waves take part in the shuffles
        const int ci = i >> 3, part = i & 7;
        float sg = 0.f, sb = 0.f;
        if (ci < gw) {
            const int c = cg0 + ci;
            for (int sl = part; sl < nslab; sl += 8) { const float2 v = partial[((long long)b * nslab + sl) * C + c]; sg += v.x; sb += v.y; }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { sg += __shfl_xor(sg, o); sb += __shfl_xor(sb, o); }
        if (part == 0 && ci < gw) {
            dgam_part[(long long)b * C + cg0 + ci] = sg; dbet_part[(long long)b * C + cg0 + ci] = sb;
            sgs[ci] = sg; sbs[ci] = sb;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const float N = (float)gw * (float)HW;
        float Sa = 0.f, Sb = 0.f;
        for (int ci = 0; ci < gw; ++ci) { Sa += gamma[cg0 + ci] * sbs[ci]; Sb += gamma[cg0 + ci] * sgs[ci]; }
        mab[((long long)b * 32 + g) * 2] = Sa / N; mab[((long long)b * 32 + g) * 2 + 1] = Sb / N;
    }
}
__global__ __launch_bounds__(256) void gn_bwd_param_kernel(const float* __restrict__ dgam_part, const float* __restrict__ dbet_part, int B, int C,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    // eight lanes per (parameter, channel): image b goes to lane b % 8, ascending within a lane, then a fixed xor tree (64 dependent loads in one thread were 16 us)
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 3, part = threadIdx.x & 7;
    const bool live = i < 2 * C;
    const int c = i < C ? i : i - C;
    const float* src = i < C ? dgam_part : dbet_part;
    float t = 0.f;
    if (live) for (int b = part; b < B; b += 8) t += src[(long long)b * C + c];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) t += __shfl_xor(t, o);
    if (live && part == 0) { float* out = i < C ? dgamma : dbeta; out[c] = accumulate ? out[c] + t : t; }
}
// elementwise over (image, pixel, 16-byte channel vector).  grid (pixel chunks, B, column blocks): a thread keeps ONE channel vector -- its group statistics,
// gamma and beta live in registers, and no index in the pixel loop divides by a run-time value (the first form did eleven such divisions per vector and ran
// at a fifth of the memory rate).  Same arithmetic per element.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x0, int xs0, int C0, const T* __restrict__ x1, int xs1, int C, int HW,
                                                           const T* __restrict__ dy, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean_rstd, const float* __restrict__ mab, int silu, T* __restrict__ dx0, int acc0,
                                                           T* __restrict__ dx1, int acc1) {
    constexpr int VEC = TI<T>::VEC;
    const int cols = C / VEC, gw = C / 32, C1 = C - C0;
    const int cb = blockIdx.z;
    const int cols_here = min(cols - cb * 256, 256);
    const int rows = 256 / cols_here;
    const int tid = threadIdx.x;
    const int col = tid % cols_here, row = tid / cols_here;
    if (row >= rows) return;
    const int c = (cb * 256 + col) * VEC;
    const long long b = blockIdx.y;
    float gm[VEC], bt[VEC], mean[VEC], rstd[VEC], ma[VEC], mb[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int g = (c + e) / gw;
        gm[e] = gamma[c + e]; bt[e] = beta[c + e];
        mean[e] = mean_rstd[(b * 32 + g) * 2]; rstd[e] = mean_rstd[(b * 32 + g) * 2 + 1];
        ma[e] = mab[(b * 32 + g) * 2]; mb[e] = mab[(b * 32 + g) * 2 + 1];
    }
    const bool first = c < C0;
    const T* xp = first ? x0 + c : x1 + (c - C0);
    const int xs = first ? xs0 : xs1;
    T* dp = first ? dx0 + c : dx1 + (c - C0);
    const int ds = first ? C0 : C1;
    const int acc = first ? acc0 : acc1;
    for (int p = blockIdx.x * rows + row; p < HW; p += gridDim.x * rows) {
        const long long bp = b * HW + p;
        const uint4 ux = *(const uint4*)(xp + bp * xs);
        const uint4 ud = *(const uint4*)(dy + bp * C + c);
        float xh[VEC], dv[VEC], d[VEC];
        TI<T>::unpack(ux, xh);
        TI<T>::unpack(ud, dv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float h = (xh[e] - mean[e]) * rstd[e];
            float v = dv[e];
            if (silu) { const float pre = h * gm[e] + bt[e]; const float sgm = 1.0f / (1.0f + __expf(-pre)); v *= sgm * (1.0f + pre * (1.0f - sgm)); }
            d[e] = rstd[e] * (v * gm[e] - ma[e] - h * mb[e]);
        }
        T* dst = dp + bp * ds;
        if (acc) {
            float o[VEC];
            TI<T>::unpack(*(const uint4*)dst, o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[e] += o[e];
        }
        *(uint4*)dst = TI<T>::pack(d);
    }
}
// zero two small ranges of 16-bit / 32-bit elements (the margins around the shifted wgrad operand) in one launch
__global__ __launch_bounds__(256) void zero2_kernel(unsigned char* __restrict__ p0, long long n0, unsigned char* __restrict__ p1, long long n1) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1; i += (long long)gridDim.x * blockDim.x) {
        if (i < n0) p0[i] = 0; else p1[i - n0] = 0;
    }
}
template <typename T>
static void l_gn_act_bwd(hipStream_t s, int B, const void* x0, int xs0, int C0, const void* x1, int xs1, int C, int HW, const void* dy, const float* g, const float* bta,
                         const float* mr, int silu, void* dx0, int acc0, void* dx1, int acc1, float* dgp, float* dbp, float2* partial, int nslab, float* mab,
                         float* dgamma, float* dbeta, int acc_param) {
    constexpr int VEC = TI<T>::VEC;
    const int cols = C / VEC;
    hipLaunchKernelGGL(gn_bwd_sums_kernel<T>, dim3(nslab, B, (cols + 255) / 256), dim3(256), 0, s, (const T*)x0, xs0, C0, (const T*)x1, xs1, C, HW, nslab, (const T*)dy, g,
                       bta, mr, silu, partial);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(32, B), dim3(256), 0, s, partial, nslab, B, C, HW, g, dgp, dbp, mab);
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((2 * C * 8 + 255) / 256), dim3(256), 0, s, dgp, dbp, B, C, dgamma, dbeta, acc_param);
    {
        const int cblocks = (cols + 255) / 256;
        const int rows = 256 / (cols < 256 ? cols : 256);
        int chunks = (HW + rows * 4 - 1) / (rows * 4);            // ~four vectors per thread
        if (chunks < 1) chunks = 1;
        hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3(chunks, B, cblocks), dim3(256), 0, s, (const T*)x0, xs0, C0, (const T*)x1, xs1, C, HW, (const T*)dy, g, bta, mr, mab,
                           silu, (T*)dx0, acc0, (T*)dx1, acc1);
    }
}

static int kalign(int dtype) { return dtype == WDM_BF16 ? 32 : 16; }

// typed launch helpers ------------------------------------------------------------------------------------------------
template <typename T>
static void gather_t(hipStream_t s, const void* src0, int xs0, int C0, const void* src1, int xs1, int C1, int B, int H, int W, int Ho, int Wo, int stride, int off_y,
                     int off_x, void* dst, int rows_per_img, int kp, int zero_rows_to, int Bg, int ndx, long long dst_dx, float* csum = nullptr) {
    // dst group stride is rows_per_img * Bg * kp; rows C .. zero_rows_to-1 of every image are zero-filled (the GEMM's padded M rows)
    const int C = C0 + C1;
    const int crows = zero_rows_to > C ? zero_rows_to : C;
    if constexpr (sizeof(T) == 2) {
        if (C0 % 8 == 0 && C1 % 8 == 0 && xs0 % 8 == 0 && (src1 == nullptr || xs1 % 8 == 0) && kp % 8 == 0 && dst_dx % 8 == 0 && ((size_t)dst & 15) == 0) {
            hipLaunchKernelGGL(gather_t_bf16x8_kernel, dim3((kp + 63) / 64, (crows + 63) / 64, B * ndx), dim3(256), 0, s, (const __bf16*)src0, xs0, C0,
                               (const __bf16*)(src1 ? src1 : src0), xs1, C, crows, H, W, Ho, Wo, stride, off_y, off_x, (__bf16*)dst, (long long)rows_per_img * Bg * kp, kp, Bg, B,
                               dst_dx, csum);
            return;
        }
    }
    hipLaunchKernelGGL(gather_t_kernel<T>, dim3((kp + 63) / 64, (crows + 63) / 64, B * ndx), dim3(256), 0, s, (const T*)src0, xs0, C0, (const T*)(src1 ? src1 : src0), xs1, C,
                       crows, H, W, Ho, Wo, stride, off_y, off_x, (T*)dst, (long long)rows_per_img * Bg * kp, kp, Bg, B, dst_dx, csum);
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
// one pass over OIHW: forward layout dstf[tap][rows_total][cin] (rows >= cout zero) and dgrad layout dstd[KK-1-tap][rows_d][kpad] (transposed)
template <typename T, int KK>
__device__ __forceinline__ void pack_both_tile(const float* __restrict__ w, int cout, int cin, T* __restrict__ dstf, int rows_total, T* __restrict__ dstd, int rows_d, int kpad,
                                               int bx, int by, float (*sm)[32 * KK + 1]) {
    constexpr int ROW = 32 * KK;
    const int co0 = bx * 32, ci0 = by * 32;
    const int nci = min(32, cin - ci0);
    for (int r = threadIdx.x >> 5; r < 32; r += 8) {
        const int co = co0 + r;
        for (int e = threadIdx.x & 31; e < ROW; e += 32)
            sm[r][e] = (co < cout && e < nci * KK) ? w[((long long)co * cin + ci0) * KK + e] : 0.f;
    }
    __syncthreads();
    const int l = threadIdx.x & 31;
    for (int tp = 0; tp < KK; ++tp)
        for (int r = threadIdx.x >> 5; r < 32; r += 8) {
            // forward: row co0 + r, 32 consecutive input channels
            if (co0 + r < rows_total && ci0 + l < cin) TI<T>::st(dstf, ((long long)tp * rows_total + co0 + r) * cin + ci0 + l, sm[r][l * KK + tp]);
            // dgrad: row ci0 + r, 32 consecutive output channels, taps mirrored
            if (dstd != nullptr && ci0 + r < rows_d && co0 + l < kpad) TI<T>::st(dstd, ((long long)tp * rows_d + ci0 + r) * kpad + co0 + l, sm[l][r * KK + (KK - 1 - tp)]);
        }
}
template <typename T, int KK>
__global__ __launch_bounds__(256) void pack_both_kernel(const float* __restrict__ w, int cout, int cin, T* __restrict__ dstf, int rows_total, T* __restrict__ dstd,
                                                        int rows_d, int kpad) {
    __shared__ float sm[32][32 * KK + 1];
    pack_both_tile<T, KK>(w, cout, cin, dstf, rows_total, dstd, rows_d, kpad, (int)blockIdx.x, (int)blockIdx.y, sm);
}
// ... for a BATCH of layers in one launch (the training step packs all ~90 convs of the model after every optimiser step: one launch per layer was 4.6 % of
// a 64-sample step, most of it launch latency).  The descriptors travel as kernel arguments; workgroup b belongs to the layer whose block range holds b.
template <typename T, int KK>
__global__ __launch_bounds__(256) void pack_both_batch_kernel(const PackBatch pb) {
    __shared__ float sm[32][32 * KK + 1];
    int j = 0;
#pragma unroll 1
    for (int k = 1; k < pb.n; ++k) if ((int)blockIdx.x >= pb.d[k].blk0) j = k;
    const PackDesc& d = pb.d[j];
    const int local = (int)blockIdx.x - d.blk0;
    pack_both_tile<T, KK>(d.w, d.cout, d.cin, (T*)d.dstf, d.rows_total, (T*)d.dstd, d.rows_d, d.kpad, local % d.gx, local / d.gx, sm);
}
int k_pack_conv_both_batch(const PackDesc* descs, int n, int k, int dtype, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += PackBatch::MAX) {
        PackBatch pb;
        pb.n = std::min(PackBatch::MAX, n - i0);
        int blocks = 0;
        for (int i = 0; i < pb.n; ++i) {
            PackDesc d = descs[i0 + i];
            d.rows_d = conv_rows_pad(d.cin);
            d.kpad = (int)align_up((size_t)d.cout, kalign(dtype));
            d.gx = (std::max(d.rows_total, d.kpad) + 31) / 32;
            const int gy = (std::max(d.cin, d.rows_d) + 31) / 32;
            d.blk0 = blocks;
            blocks += d.gx * gy;
            pb.d[i] = d;
        }
        if (dtype == WDM_BF16) {
            if (k == 3) hipLaunchKernelGGL((pack_both_batch_kernel<__bf16, 9>), dim3(blocks), dim3(256), 0, s, pb);
            else hipLaunchKernelGGL((pack_both_batch_kernel<__bf16, 1>), dim3(blocks), dim3(256), 0, s, pb);
        } else {
            if (k == 3) hipLaunchKernelGGL((pack_both_batch_kernel<float, 9>), dim3(blocks), dim3(256), 0, s, pb);
            else hipLaunchKernelGGL((pack_both_batch_kernel<float, 1>), dim3(blocks), dim3(256), 0, s, pb);
        }
    }
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
template <typename T> static void l_pack_dgrad(hipStream_t s, const float* w, int cout, int cin, int kk, void* dst, int rows, int kpad) {
    const dim3 grid((kpad + 31) / 32, (rows + 31) / 32);
    if (kk == 9) hipLaunchKernelGGL((pack_dgrad_kernel<T, 9>), grid, dim3(256), 0, s, w, cout, cin, (T*)dst, rows, kpad);
    else hipLaunchKernelGGL((pack_dgrad_kernel<T, 1>), grid, dim3(256), 0, s, w, cout, cin, (T*)dst, rows, kpad);
}
template <typename T> static void l_pad_channels(hipStream_t s, const void* x, int C, int Cp, void* y, long long total) {
    hipLaunchKernelGGL(pad_channels_kernel<T>, dim3(nblk(total, 256)), dim3(256), 0, s, (const T*)x, C, Cp, (T*)y, total);
}
// per-image column sums in ONE launch: out[g][c] = sum over the rows of image g of x[g][row][c].  grid (ceil(C / 64), images), 256 threads = (64 / VEC channel
// vectors) x (row lanes); a row lane adds its rows in ascending order, the lanes are joined in ascending order.  (colsum_part + colsum_final: two launches.)
template <typename T>
__global__ __launch_bounds__(256) void colsum_img_kernel(const T* __restrict__ x, int xs, int C, int rows, float* __restrict__ out, int out_ld, int accumulate) {
    constexpr int VEC = TI<T>::VEC, CV = 64 / VEC, RL = 256 / CV;
    __shared__ float red[RL][65];
    const int g = blockIdx.y, cv = threadIdx.x % CV, rl = threadIdx.x / CV;
    const int c = blockIdx.x * 64 + cv * VEC;
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    if (c < C) {
        const T* p = x + (long long)g * rows * xs + c;
        for (int r = rl; r < rows; r += RL) {
            float f[VEC];
            TI<T>::unpack(*(const uint4*)(p + (long long)r * xs), f);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[rl][cv * VEC + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + (int)threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < RL; ++k) t += red[k][threadIdx.x];
        const long long o = (long long)g * out_ld + blockIdx.x * 64 + threadIdx.x;
        out[o] = accumulate ? out[o] + t : t;
    }
}
// needs a scratch buffer of groups * nchunks * C floats
static inline int colsum_chunks(long long rows_per_group) { long long n = (rows_per_group + 255) / 256; return (int)(n < 1 ? 1 : (n > 256 ? 256 : n)); }
template <typename T> static void l_colsum(hipStream_t s, const void* x, int xs, int C, long long rows_per_group, int groups, float* out, int acc, int out_ld,
                                           float* scratch) {
    const int nchunks = colsum_chunks(rows_per_group);
    const int chunk_rows = (int)((rows_per_group + nchunks - 1) / nchunks);
    hipLaunchKernelGGL(colsum_part_kernel<T>, dim3((C + 63) / 64, groups, nchunks), dim3(256), 0, s, (const T*)x, xs, C, rows_per_group, chunk_rows, nchunks, scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64, groups), dim3(256), 0, s, scratch, C, C, nchunks, groups, out, out_ld ? out_ld : C, acc);
}

// ---- dgrad: dx (+)= conv^T(dy).  (H, W) is the forward INPUT map; dy is dense NHWC [B][Ho][Wo][cout]; dx dense [B][H][W][cin].
size_t conv_dgrad_packed_bytes(int cin, int cout, int k, int dtype) {
    return (size_t)k * k * conv_rows_pad(cin) * align_up((size_t)cout, kalign(dtype)) * dsize(dtype);
}
int k_pack_conv_both(const float* w_oihw, int cout, int cin, int k, void* dst_fwd, int rows_total, void* dst_dgrad, int dtype, hipStream_t s) {
    const int rows_d = conv_rows_pad(cin), kpad = (int)align_up((size_t)cout, kalign(dtype));
    const dim3 grid((std::max(rows_total, kpad) + 31) / 32, (std::max(cin, rows_d) + 31) / 32);
    if (dtype == WDM_BF16) {
        if (k == 3) hipLaunchKernelGGL((pack_both_kernel<__bf16, 9>), grid, dim3(256), 0, s, w_oihw, cout, cin, (__bf16*)dst_fwd, rows_total, (__bf16*)dst_dgrad, rows_d, kpad);
        else hipLaunchKernelGGL((pack_both_kernel<__bf16, 1>), grid, dim3(256), 0, s, w_oihw, cout, cin, (__bf16*)dst_fwd, rows_total, (__bf16*)dst_dgrad, rows_d, kpad);
    } else {
        if (k == 3) hipLaunchKernelGGL((pack_both_kernel<float, 9>), grid, dim3(256), 0, s, w_oihw, cout, cin, (float*)dst_fwd, rows_total, (float*)dst_dgrad, rows_d, kpad);
        else hipLaunchKernelGGL((pack_both_kernel<float, 1>), grid, dim3(256), 0, s, w_oihw, cout, cin, (float*)dst_fwd, rows_total, (float*)dst_dgrad, rows_d, kpad);
    }
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
int conv_dgrad(Ctx& c, int mode, const float* w_oihw, int cin, int cout, const Tens& dy, int H, int W, void* dx, bool accumulate, const void* wd_prepacked) {
    const int k = mode == MODE_P1 ? 1 : 3, kk = k * k;
    const size_t es = dsize(c.dtype);
    const int kpad = (int)align_up((size_t)cout, kalign(c.dtype));             // contraction length (forward cout), padded
    const int rows = conv_rows_pad(cin);
    void* wd = wd_prepacked ? const_cast<void*>(wd_prepacked) : c.ar->alloc((size_t)kk * rows * kpad * es);
    if (!wd) WDM_FAIL(WDM_ENOMEM, "workspace too small (dgrad weights)");
    if (!c.dry && !wd_prepacked) BY_DTYPE(c.dtype, l_pack_dgrad, c.s, w_oihw, cout, cin, kk, wd, rows, kpad);
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
    if (!wd_prepacked) c.ar->free(wd);
    return rc;
}

// tile column sums csum[B][nkt][cout] (from the dy gather) -> per-image sums (into dtemb when asked for, else a scratch) -> bias gradient; frees csum
static int finish_colsums(Ctx& c, float* csum, int nkt, int cout, float* db, float* dtemb, int dtemb_ld) {
    if (!csum) return WDM_OK;
    float* per_img = dtemb;
    int ld = dtemb_ld;
    if (!per_img) {
        per_img = (float*)c.ar->alloc((size_t)c.B * cout * sizeof(float));
        if (!per_img) WDM_FAIL(WDM_ENOMEM, "workspace too small (bias gradient)");
        ld = cout;
    }
    hipLaunchKernelGGL(colsum_final_kernel, dim3((cout + 63) / 64, c.B), dim3(256), 0, c.s, csum, cout, cout, nkt, c.B, per_img, ld, 0);
    if (db) hipLaunchKernelGGL(colsum_final_kernel, dim3((cout + 63) / 64, 1), dim3(256), 0, c.s, per_img, ld, cout, c.B, 1, db, cout, 0);
    WDM_HIP(hipGetLastError());
    if (!dtemb) c.ar->free(per_img);
    c.ar->free(csum);
    return WDM_OK;
}

// ---- wgrad: dw[co][ci][tap] (OIHW f32) (+)= sum_pixels dy[p][co] * x[p + tap][ci];  x = [x0 | x1] (the forward input; for MODE_UPS the
// low-resolution map, upsampled here), dy dense [B][Ho][Wo][cout]
int conv_wgrad(Ctx& c, int mode, const Tens& x0, const Tens* x1, const Tens& dy, int cout, float* dw, bool accumulate, float* db, float* dtemb, int dtemb_ld) {
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
    const bool shifted = (mode == MODE_S1 || mode == MODE_UPS);
    // Images are contracted in groups of Bg (K = Bg x pixels per GEMM row): one fp32 partial per GROUP instead of per image.  Bg is the
    // largest divisor of B that still leaves ~512 workgroups per launch (small layers need the split over images for parallelism).
    int Bg = 1;
    {
        const long long wgs = (long long)(shifted ? 9 : 1) * ((rows_g + 127) / 128) * ((cin + 127) / 128);
        const long long s_min = (512 + wgs - 1) / wgs;
        for (int d = 1; d <= c.B; ++d)
            if (c.B % d == 0 && c.B / d >= s_min) Bg = d;
        { const int v = env_cfg().wgrad_bg; if (v >= 1 && c.B % v == 0) Bg = v; }
    }
    const int S = c.B / Bg;
    int rc = WDM_OK;
    // 3x3 stride-1 layers on 16-pixel-wide maps, bf16: the direct kernel (conv_wgrad_kernel.h) -- no transposed copies, one launch + the partial reduction
    const bool map8 = H == 8 && W == 8;
    const bool fits32 = (unsigned long long)c.B * H * W * (unsigned long long)std::max(std::max(cout, s0->xs), s1 ? s1->xs : 0) * 2ull < 4294901760ull;      // buffer offsets are 32-bit, 0xFFFF0000 marks "outside" (conv_dispatch.inc: set_extents)
    if (shifted && fits32 && c.dtype == WDM_BF16 && env_cfg().wgrad_bg == 0 && ((W % 16 == 0 && H % 8 == 0) || map8) && dy.xs == cout && cout % 8 == 0 && s0->xs % 8 == 0 &&
        (!s1 || (s0->C % 64 == 0 && s1->xs % 8 == 0)) && cin % 8 == 0) {
        WgradArgs w{};
        w.dy = dy.p; w.x0 = s0->p; w.x1 = s1 ? s1->p : nullptr;
        w.B = c.B; w.H = H; w.W = W; w.cout = cout; w.C0 = s0->C; w.C1 = s1 ? s1->C : 0; w.xs0 = s0->xs; w.xs1 = s1 ? s1->xs : 0; w.cin = cin; w.rows_g = rows_g;
        w.n_co = (cout + 127) / 128; w.n_ci = (cin + 63) / 64;
        w.nchunk = map8 ? (c.B + 1) / 2 : c.B * (H / 8) * (W / 16);
        const int ntile = w.n_co * w.n_ci;
        int Sd = (256 + ntile - 1) / ntile;                       // one workgroup per CU: the pixels are split as far as the tiles leave CUs idle
        if (Sd > w.nchunk) Sd = w.nchunk;
        w.cps = (w.nchunk + Sd - 1) / Sd;
        w.S = (w.nchunk + w.cps - 1) / w.cps;
        const size_t px = (size_t)c.B * H * W;
        w.dy_bytes = (unsigned)(px * cout * es); w.x0_bytes = (unsigned)(px * s0->xs * es); w.x1_bytes = s1 ? (unsigned)(px * s1->xs * es) : 0u;
        float* part = (float*)c.ar->alloc((size_t)9 * w.S * rows_g * cin * sizeof(float));
        if (!part) WDM_FAIL(WDM_ENOMEM, "workspace too small (wgrad partials)");
        w.part = part;
        if (db || dtemb) {          // (allocations outside the dry test: a sizing pass must see the same arena sequence as the real one)
            float* per_img = dtemb; int ld = dtemb_ld;
            if (!per_img) { per_img = (float*)c.ar->alloc((size_t)c.B * cout * sizeof(float)); ld = cout; if (!per_img) WDM_FAIL(WDM_ENOMEM, "workspace too small (bias gradient)"); }
            WDM_TRY(colsum(c, dy, per_img, true, false, ld));
            if (db && !c.dry) hipLaunchKernelGGL(colsum_final_kernel, dim3((cout + 63) / 64, 1), dim3(256), 0, c.s, per_img, ld, cout, c.B, 1, db, cout, 0);
            if (!dtemb) c.ar->free(per_img);
        }
        if (!c.dry) {
            {   // hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the function: once per (kernel, device)
                static std::atomic<unsigned> devs{0};
                int dev = 0;
                WDM_HIP(hipGetDevice(&dev));
                const unsigned bit = 1u << (dev & 31);
                if (!(devs.load(std::memory_order_acquire) & bit)) {
                    WDM_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WgradCfg<false>::LDS_BYTES));
                    WDM_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WgradCfg<true>::LDS_BYTES));
                    devs.fetch_or(bit, std::memory_order_release);
                }
            }
            if (map8) hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3(w.S * ntile), dim3(512), WgradCfg<true>::LDS_BYTES, c.s, w);
            else hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3(w.S * ntile), dim3(512), WgradCfg<false>::LDS_BYTES, c.s, w);
            const long long total = (long long)3 * cout * cin;
            hipLaunchKernelGGL(reduce_wgrad_kernel, dim3(nblk(total, 256)), dim3(256), 0, c.s, part, 9, w.S, rows_g, cout, cin, dw, accumulate ? 1 : 0);
            WDM_HIP(hipGetLastError());
        }
        c.ar->free(part);
        if (t_up0) c.ar->free(t_up0);
        return WDM_OK;
    }
    if (shifted) {
        // 3x3 stride 1: ONE transposed image per dx column on a grid of (H + 2) rows x Wq columns (Wq = W rounded up to 8, so that a
        // dy tap is a 16-byte aligned shift of +-Wq elements): aT_dx[ci][(y+1) Wq + x] = a[y][x + dx - 1], dyT on the same grid with
        // zero border rows / columns -- 4 gathers + 9 GEMMs instead of 10 gathers.  Where a shifted read leaves its row the dyT factor
        // is a border zero (operands are finite activations), so the neighbouring row's data is harmless; the buffer ends get zeroed margins.
        const int Wq = (int)align_up((size_t)W, 8), Hq = H + 2;
        const int kq = (int)align_up((size_t)Hq * Wq, kalign(c.dtype));
        const size_t a_elems = (size_t)c.B * cin * kq;
        void* dyT = c.ar->alloc((size_t)c.B * rows_g * kq * es);
        char* aT3 = (char*)c.ar->alloc((3 * a_elems + 2 * (size_t)Wq) * es);
        float* part = (float*)c.ar->alloc((size_t)kk * S * rows_g * cin * sizeof(float));
        if (!dyT || !aT3 || !part) WDM_FAIL(WDM_ENOMEM, "workspace too small (wgrad)");
        if (!c.dry) {
            hipLaunchKernelGGL(zero2_kernel, dim3(1), dim3(256), 0, c.s, (unsigned char*)aT3, (long long)((size_t)Wq * es),
                               (unsigned char*)aT3 + ((size_t)Wq + 3 * a_elems) * es, (long long)((size_t)Wq * es));
            const int nkt = (kq + 63) / 64;
            float* csum = (db || dtemb) ? (float*)c.ar->alloc((size_t)c.B * nkt * cout * sizeof(float)) : nullptr;
            if ((db || dtemb) && !csum) WDM_FAIL(WDM_ENOMEM, "workspace too small (wgrad column sums)");
            BY_DTYPE(c.dtype, gather_t, c.s, dy.p, dy.xs, cout, nullptr, 0, 0, c.B, Ho, Wo, Hq, Wq, 1, -1, 0, dyT, rows_g, kq, rows_g, Bg, 1, 0, csum);
            WDM_TRY(finish_colsums(c, csum, nkt, cout, db, dtemb, dtemb_ld));
            // the three dx-shifted copies of [x0 | x1] in one launch
            BY_DTYPE(c.dtype, gather_t, c.s, s0->p, s0->xs, s0->C, s1 ? s1->p : nullptr, s1 ? s1->xs : 0, s1 ? s1->C : 0, c.B, H, W, Hq, Wq, 1, -1, 0,
                     aT3 + (size_t)Wq * es, cin, kq, 0, Bg, 3, (long long)a_elems);
            {   // the nine taps as ONE batched GEMM: "image" i = tap * S + g reads dyT[g] and aT_{tap % 3}[g] shifted by (tap / 3 - 1) grid rows;
                // a row is the concatenation of the group's Bg images (a shift that leaves an image's segment meets that image's zero border in dyT)
                const int kg = Bg * kq;
                ConvArgs a{};
                a.x0 = dyT; a.C0 = kg; a.xs0 = kg; a.C1 = 0;
                a.B = 9 * S; a.Hin = a.Hout = Hg; a.Win = a.Wout = Wg;
                a.Cin = kg; a.Cout = cin;
                a.w = aT3 + (size_t)Wq * es;
                a.w_tap_stride = 0; a.w_img_stride = (long long)cin * kg; a.w_row_stride = kg; a.w_rows = cin;
                a.img_mod = S; a.w_tx_stride = (long long)a_elems; a.w_ty_stride = Wq;
                a.w_bytes = (unsigned)((size_t)cin * kg * es);
                a.alpha = 1.f;
                a.y = part; a.y_mode = Y_NHWC_F32; a.y_s = cin;
                rc = launch_conv(a, MODE_P1, c.dtype, c.s);
            }
            if (rc == WDM_OK) {
                const long long total = (long long)kk * cout * cin;
                hipLaunchKernelGGL(reduce_wgrad_kernel, dim3(nblk(total, 256)), dim3(256), 0, c.s, part, kk, S, rows_g, cout, cin, dw, accumulate ? 1 : 0);
                WDM_HIP(hipGetLastError());
            }
        }
        c.ar->free(part); c.ar->free(aT3); c.ar->free(dyT);
        if (t_up0) c.ar->free(t_up0);
        return rc;
    }
    void* dyT = c.ar->alloc((size_t)c.B * rows_g * kp * es);
    void* aT = c.ar->alloc((size_t)c.B * cin * kp * es);
    float* part = (float*)c.ar->alloc((size_t)kk * S * rows_g * cin * sizeof(float));
    if (!dyT || !aT || !part) WDM_FAIL(WDM_ENOMEM, "workspace too small (wgrad)");
    if (!c.dry) {
        const int kg = Bg * kp;
        const int nkt = (kp + 63) / 64;
        float* csum = (db || dtemb) ? (float*)c.ar->alloc((size_t)c.B * nkt * cout * sizeof(float)) : nullptr;
        if ((db || dtemb) && !csum) WDM_FAIL(WDM_ENOMEM, "workspace too small (wgrad column sums)");
        BY_DTYPE(c.dtype, gather_t, c.s, dy.p, dy.xs, cout, nullptr, 0, 0, c.B, Ho, Wo, Ho, Wo, 1, 0, 0, dyT, rows_g, kp, rows_g, Bg, 1, 0, csum);
        WDM_TRY(finish_colsums(c, csum, nkt, cout, db, dtemb, dtemb_ld));
        const int stride = mode == MODE_S2 ? 2 : 1;
        for (int tap = 0; tap < kk && rc == WDM_OK; ++tap) {
            const int ty = tap / k, tx = tap % k;
            const int oy = mode == MODE_P1 ? 0 : ty, ox = mode == MODE_P1 ? 0 : tx;      // Downsample: pad(0,1,0,1), stride 2
            BY_DTYPE(c.dtype, gather_t, c.s, s0->p, s0->xs, s0->C, s1 ? s1->p : nullptr, s1 ? s1->xs : 0, s1 ? s1->C : 0, c.B, H, W, Ho, Wo, stride, oy, ox, aT, cin, kp,
                     0, Bg, 1, 0);
            ConvArgs a{};
            a.x0 = dyT; a.C0 = kg; a.xs0 = kg; a.C1 = 0;
            a.B = S; a.Hin = a.Hout = Hg; a.Win = a.Wout = Wg;
            a.Cin = kg; a.Cout = cin;
            a.w = aT; a.w_tap_stride = 0; a.w_img_stride = (long long)cin * kg; a.w_row_stride = kg; a.w_rows = cin;
            a.w_bytes = (unsigned)((size_t)cin * kg * es);
            a.alpha = 1.f;
            a.y = part + (size_t)tap * S * rows_g * cin; a.y_mode = Y_NHWC_F32; a.y_s = cin;
            rc = launch_conv(a, MODE_P1, c.dtype, c.s);
        }
        if (rc == WDM_OK) {
            const long long total = (long long)kk * cout * cin;
            hipLaunchKernelGGL(reduce_wgrad_kernel, dim3(nblk(total, 256)), dim3(256), 0, c.s, part, kk, S, rows_g, cout, cin, dw, accumulate ? 1 : 0);
            WDM_HIP(hipGetLastError());
        }
    }
    c.ar->free(part); c.ar->free(aT); c.ar->free(dyT);
    if (t_up0) c.ar->free(t_up0);
    return rc;
}

// db[c] (+)= sum over all rows of dy;  per_image: out[b][c] = sum over the image's rows (temb gradient)
int colsum(Ctx& c, const Tens& dy, float* out, bool per_image, bool accumulate, int out_ld) {
    const long long rows = (long long)dy.H * dy.W * (per_image ? 1 : c.B);
    const int groups = per_image ? c.B : 1;
    float* scratch = (float*)c.ar->alloc((size_t)groups * colsum_chunks(rows) * dy.C * sizeof(float));
    if (!scratch) WDM_FAIL(WDM_ENOMEM, "workspace too small (column sums)");
    if (!c.dry) {
        const int vec = c.dtype == WDM_BF16 ? 8 : 4;
        if (per_image && dy.C % vec == 0 && dy.xs % vec == 0 && rows <= (1 << 20)) {        // one launch per tensor; a thread walks rows / 32 (16) rows
            const dim3 grid((dy.C + 63) / 64, groups);
            if (c.dtype == WDM_BF16) hipLaunchKernelGGL(colsum_img_kernel<__bf16>, grid, dim3(256), 0, c.s, (const __bf16*)dy.p, dy.xs, dy.C, (int)rows, out, out_ld ? out_ld : dy.C, accumulate ? 1 : 0);
            else hipLaunchKernelGGL(colsum_img_kernel<float>, grid, dim3(256), 0, c.s, (const float*)dy.p, dy.xs, dy.C, (int)rows, out, out_ld ? out_ld : dy.C, accumulate ? 1 : 0);
        } else {
            BY_DTYPE(c.dtype, l_colsum, c.s, dy.p, dy.xs, dy.C, rows, groups, out, accumulate ? 1 : 0, out_ld, scratch);
        }
        WDM_HIP(hipGetLastError());
    }
    c.ar->free(scratch);
    return WDM_OK;
}

// GroupNorm (+SiLU) backward over [x0 | x1]; mean_rstd from the forward's finalize.  dgamma / dbeta (+)= batch sums.
int gn_act_backward(Ctx& c, const NormW& nw, const Tens& x0, const Tens* x1, const float* mean_rstd, const Tens& dy, int silu, void* dx0, bool acc0, void* dx1,
                    bool acc1, float* dgamma, float* dbeta, bool acc_param) {
    const int C = x0.C + (x1 ? x1->C : 0), HW = x0.H * x0.W;
    const int vec = c.dtype == WDM_BF16 ? 8 : 4;
    if (C % 32 || x0.C % vec || C % vec) WDM_FAIL(WDM_EINVAL, "GroupNorm backward: %d (+%d) channels unsupported", x0.C, C - x0.C);
    // slabs: at most ~4 pixel iterations per thread (8 x 8 maps with hundreds of channels would otherwise run B long-latency workgroups)
    int nslab;
    {
        const int cols = C / vec, rows = std::max(1, 256 / std::min(cols, 256));
        nslab = std::min(256, std::max(1, HW / (4 * rows)));
    }
    // per-image dgamma / dbeta [2][B][C], slab partials [B][nslab][C] x 2, group means [B][32] x 2
    float* part = (float*)c.ar->alloc(((size_t)2 * c.B * C + (size_t)2 * c.B * nslab * C + (size_t)2 * c.B * 32) * sizeof(float));
    if (!part) WDM_FAIL(WDM_ENOMEM, "workspace too small (GroupNorm backward)");
    if (!c.dry) {
        float* slabs = part + (size_t)2 * c.B * C;
        float* mab = slabs + (size_t)2 * c.B * nslab * C;
        BY_DTYPE(c.dtype, l_gn_act_bwd, c.s, c.B, x0.p, x0.xs, x0.C, x1 ? x1->p : x0.p, x1 ? x1->xs : 0, C, HW, dy.p, nw.g, nw.b, mean_rstd, silu, dx0, acc0 ? 1 : 0,
                 x1 ? dx1 : dx0, acc1 ? 1 : 0, part, part + (size_t)c.B * C, (float2*)slabs, nslab, mab, dgamma, dbeta, acc_param ? 1 : 0);
        WDM_HIP(hipGetLastError());
    }
    c.ar->free(part);
    return WDM_OK;
}

// dst[b][c][n] = src[b][n][c]  (tokens n = 0..N-1, dense rows of C): the attention backward's operand transposes
int transpose_tokens(Ctx& c, const void* src, int N, int Cc, void* dst) {
    if (c.dry) return WDM_OK;
    BY_DTYPE(c.dtype, gather_t, c.s, src, Cc, Cc, nullptr, 0, 0, c.B, 1, N, 1, N, 1, 0, 0, dst, Cc, N, 0, 1, 1, 0);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}
void l_colsum_f32(hipStream_t s, const float* x, int C, int rows, float* out, float* scratch) { l_colsum<float>(s, x, C, C, rows, 1, out, 0, 0, scratch); }   // scratch: colsum_chunks(rows) * C floats

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
    WDM_TRY(conv_wgrad(c, mode, tx, nullptr, tdy, cout, dw, false, db));
    return WDM_OK;
}

extern "C" int wdm_gn_act_backward(wdm_handle* h, const float* x, int C0, int C, const float* gamma, const float* beta, const float* dy, int silu, int B, int H, int W,
                                   float* dx, float* dgamma, float* dbeta, int dtype, void* scratch, size_t scratch_bytes, void* stream) {
    if (!h || !x || !gamma || !beta || !dy || !dx || !dgamma || !dbeta || !scratch) WDM_FAIL(WDM_EINVAL, "wdm_gn_act_backward: null argument");
    if (C % 32 || C0 <= 0 || C0 > C) WDM_FAIL(WDM_EINVAL, "wdm_gn_act_backward: bad channel split");
    Arena ar(scratch, scratch_bytes);
    Ctx c{(hipStream_t)stream, dtype, B, &ar, false};
    const size_t es = dsize(dtype);
    const int HW = H * W, C1 = C - C0;
    // the block input arrives as ONE NCHW tensor; split it into the two NHWC sources the executor would hold
    void* full = ar.alloc((size_t)B * HW * C * es);
    Tens tdy; tdy.p = ar.alloc((size_t)B * HW * C * es); tdy.C = C; tdy.H = H; tdy.W = W; tdy.xs = C;
    void* d0 = ar.alloc((size_t)B * HW * C0 * es);
    void* d1 = C1 ? ar.alloc((size_t)B * HW * C1 * es) : nullptr;
    void* dfull = ar.alloc((size_t)B * HW * C * es);
    float* mr = (float*)ar.alloc((size_t)B * 64 * sizeof(float));
    float *sc = (float*)ar.alloc((size_t)B * C * 4), *sh = (float*)ar.alloc((size_t)B * C * 4);
    if (!full || !tdy.p || !d0 || (C1 && !d1) || !dfull || !mr || !sc || !sh) WDM_FAIL(WDM_ENOMEM, "wdm_gn_act_backward: scratch too small");
    WDM_TRY(k_nchw_to_nhwc(x, full, B, C, H, W, dtype, c.s));
    WDM_TRY(k_nchw_to_nhwc(dy, tdy.p, B, C, H, W, dtype, c.s));
    Tens t0, t1;
    t0.p = full; t0.C = C0; t0.H = H; t0.W = W; t0.xs = C;
    t1.p = (char*)full + (size_t)C0 * es; t1.C = C1; t1.H = H; t1.W = W; t1.xs = C;
    NormW nw; nw.g = gamma; nw.b = beta; nw.c = C;
    const int ns = gn_default_nslab(HW);
    float* st0 = (float*)ar.alloc(gn_stats_bytes(B, ns, C0));
    float* st1 = C1 ? (float*)ar.alloc(gn_stats_bytes(B, ns, C1)) : nullptr;
    if (!st0 || (C1 && !st1)) WDM_FAIL(WDM_ENOMEM, "wdm_gn_act_backward: scratch too small");
    WDM_TRY(k_gn_partial(t0, B, st0, ns, dtype, c.s));
    if (C1) WDM_TRY(k_gn_partial(t1, B, st1, ns, dtype, c.s));
    WDM_TRY(k_gn_finalize(B, HW, st0, ns, C0, st1, ns, C1, nw, 1e-6f, 0, sc, sh, c.s, mr));
    WDM_TRY(gn_act_backward(c, nw, t0, C1 ? &t1 : nullptr, mr, tdy, silu, d0, false, d1, false, dgamma, dbeta, false));
    // reassemble (B, C, H, W): dx = [d0 | d1]
    {
        Tens o0; o0.p = d0; o0.C = C0; o0.H = H; o0.W = W; o0.xs = C0;
        // identity "apply" = copy into the concat layout: scale 1, shift 0 rows
        WDM_HIP(hipMemsetAsync(sh, 0, (size_t)B * C * 4, c.s));
        std::vector<float> ones((size_t)B * C, 1.0f);
        WDM_HIP(hipMemcpyAsync(sc, ones.data(), ones.size() * 4, hipMemcpyHostToDevice, c.s));
        WDM_HIP(hipStreamSynchronize(c.s));
        WDM_TRY(k_gn_apply(o0, B, sc, sh, C, dfull, C, 0, 0, dtype, c.s));
        if (C1) { Tens o1; o1.p = d1; o1.C = C1; o1.H = H; o1.W = W; o1.xs = C1; WDM_TRY(k_gn_apply(o1, B, sc, sh, C, dfull, C, C0, 0, dtype, c.s)); }
    }
    return k_nhwc_to_nchw(dfull, dx, B, C, H, W, dtype, c.s);
}
