// The extra operators of the optional `data.global_attn` model (DiffusionUNet_Global / Attn_Global, models/unet.py:397-636):
//   * direct convolutions of the global branch -- Conv2d k4 s2 p1 (`down_global.conv`), ConvTranspose2d k4 s2 p1 (`up_global.conv`), the
//     patchifying q projection (k = stride = local_patch_size) and the depthwise k / v projections (k = stride = global_patch_size);
//   * GroupNorm(32, 1e-6) on its own (`norm_patch` is applied to both inputs of the attention, :433-434);
//   * the cross attention of every patch pixel block to the <= 64 tokens of the down-convolved whole image (:438-455);
//   * proj_out's nearest upsample + residual (:457-462).
// The flag is off in every YAML the reference ships and the branch is small next to the main path (the whole-image feature map is one
// 64x64 ... 8x8 tensor per patch), so these are plain fp32 NCHW kernels built for exactness, not MFMA throughput; the ResnetBlocks,
// AttnBlocks and 3x3 / 1x1 convolutions of the model run on the same block executors as DiffusionUNet (wavedm_amd/unet_global.py).
#include "common.h"

namespace wdm {
namespace {

// y[b][co][oy][ox] = bias[co] + sum_{ci in group, i, j} w * x        (forward convolution, weight [Cout][Cin/groups][k][k])
__global__ __launch_bounds__(256) void conv2d_direct_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            int B, int Cin, int H, int W, int Cout, int k, int stride, int pad, int groups, int Ho, int Wo,
                                                            float* __restrict__ y) {
    const long long total = (long long)B * Cout * Ho * Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho), co = (int)((idx / ((long long)Wo * Ho)) % Cout), b = (int)(idx / ((long long)Wo * Ho * Cout));
    const int cpg = Cin / groups, opg = Cout / groups;
    const int g = co / opg;
    float acc = bias ? bias[co] : 0.f;
    for (int c = 0; c < cpg; ++c) {
        const float* xp = x + ((long long)b * Cin + g * cpg + c) * H * W;
        const float* wp = w + ((long long)co * cpg + c) * k * k;
        for (int i = 0; i < k; ++i) {
            const int iy = oy * stride - pad + i;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int j = 0; j < k; ++j) {
                const int ix = ox * stride - pad + j;
                if ((unsigned)ix >= (unsigned)W) continue;
                acc = fmaf(wp[i * k + j], xp[(long long)iy * W + ix], acc);
            }
        }
    }
    y[idx] = acc;
}

// ConvTranspose2d (groups = 1), weight [Cin][Cout][k][k]:  y[b][co][oy][ox] = bias[co] + sum_{ci,i,j : oy = iy*stride - pad + i} w[ci][co][i][j] x[b][ci][iy][ix]
__global__ __launch_bounds__(256) void convT2d_direct_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int B,
                                                             int Cin, int H, int W, int Cout, int k, int stride, int pad, int Ho, int Wo, float* __restrict__ y) {
    const long long total = (long long)B * Cout * Ho * Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % Wo), oy = (int)((idx / Wo) % Ho), co = (int)((idx / ((long long)Wo * Ho)) % Cout), b = (int)(idx / ((long long)Wo * Ho * Cout));
    float acc = bias ? bias[co] : 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* xp = x + ((long long)b * Cin + ci) * H * W;
        const float* wp = w + ((long long)ci * Cout + co) * k * k;
        for (int i = 0; i < k; ++i) {
            const int ty = oy + pad - i;
            if (ty < 0 || ty % stride) continue;
            const int iy = ty / stride;
            if (iy >= H) continue;
            for (int j = 0; j < k; ++j) {
                const int tx = ox + pad - j;
                if (tx < 0 || tx % stride) continue;
                const int ix = tx / stride;
                if (ix >= W) continue;
                acc = fmaf(wp[i * k + j], xp[(long long)iy * W + ix], acc);
            }
        }
    }
    y[idx] = acc;
}

// GroupNorm(32, eps) on NCHW f32, one workgroup per (group, image), two passes in fp64 sums (fixed order), optional SiLU
__global__ __launch_bounds__(256) void groupnorm_nchw_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                             int HW, float eps, int silu, float* __restrict__ y) {
    __shared__ double red[512];
    const int g = blockIdx.x, b = blockIdx.y, gw = C / 32;
    const long long base = ((long long)b * C + (long long)g * gw) * HW;
    const long long n = (long long)gw * HW;
    double s1 = 0.0, s2 = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) { const double v = x[base + i]; s1 += v; s2 += v * v; }
    red[threadIdx.x] = s1; red[256 + threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) { red[threadIdx.x] += red[threadIdx.x + o]; red[256 + threadIdx.x] += red[256 + threadIdx.x + o]; }
        __syncthreads();
    }
    const double mean = red[0] / (double)n;
    double var = red[256] / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const int c = g * gw + (int)(i / HW);
        float v = (x[base + i] - mu) * rstd * gamma[c] + beta[c];
        if (silu) v = v / (1.0f + expf(-v));
        y[base + i] = v;
    }
}

// Cross attention, any number of keys: q [B][C][Nq], k / v [B][C][Nk] -> out [B][C][Nq];  w = softmax_j(C^-0.5 sum_c q[c][i] k[c][j]), out[c][i] = sum_j v[c][j] w[i][j]
// One workgroup = 64 queries of one image; thread (qi, part) with 4 parts splitting the keys / channels.  Keys go through in blocks of 64: a first sweep
// keeps the running row maximum and the running sum of exp(s - max) (rescaled when the maximum moves), a second sweep recomputes the scores, normalises
// them as softmax does and continues every output's fma chain over the keys in ascending order (out is the accumulator between blocks).  With one block
// (Nk <= 64: every map the reference's own 64 x 64 `total` produces) the arithmetic is exactly the single-block form.
__global__ __launch_bounds__(256) void cross_attention_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int C, int Nq,
                                                              int Nk, float scale, float* __restrict__ out) {
    __shared__ float sc[64][65];
    __shared__ float row_m[64], row_r[64];
    const int b = blockIdx.y, q0 = blockIdx.x * 64;
    const int qi = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int i = q0 + qi;
    const float* qb = q + (long long)b * C * Nq;
    const float* kb = k + (long long)b * C * Nk;
    const float* vb = v + (long long)b * C * Nk;
    auto scores = [&](int j0, int nj) {        // this thread owns keys j = part, part + 4, ... of the block
        for (int j = part; j < nj; j += 4) {
            float a = 0.f;
            if (i < Nq)
                for (int c = 0; c < C; ++c) a = fmaf(qb[(long long)c * Nq + i], kb[(long long)c * Nk + j0 + j], a);
            sc[qi][j] = a * scale;
        }
    };
    float m = -INFINITY, ssum = 0.f;           // part 0: running maximum and sum of the row (fixed order)
    for (int j0 = 0; j0 < Nk; j0 += 64) {
        const int nj = min(64, Nk - j0);
        scores(j0, nj);
        __syncthreads();
        if (part == 0) {
            float mb = m;
            for (int j = 0; j < nj; ++j) mb = fmaxf(mb, sc[qi][j]);
            if (j0 > 0) ssum *= expf(m - mb);
            m = mb;
            for (int j = 0; j < nj; ++j) ssum += expf(sc[qi][j] - m);
        }
        __syncthreads();
    }
    if (part == 0) { row_m[qi] = m; row_r[qi] = 1.0f / ssum; }
    __syncthreads();
    for (int j0 = 0; j0 < Nk; j0 += 64) {
        const int nj = min(64, Nk - j0);
        if (Nk > 64) scores(j0, nj);           // one block: the scores of the first sweep are still there
        __syncthreads();
        if (part == 0)
            for (int j = 0; j < nj; ++j) sc[qi][j] = expf(sc[qi][j] - row_m[qi]) * row_r[qi];
        __syncthreads();
        if (i < Nq)
            for (int c = part; c < C; c += 4) {
                float a = j0 ? out[((long long)b * C + c) * Nq + i] : 0.f;
                for (int j = 0; j < nj; ++j) a = fmaf(vb[(long long)c * Nk + j0 + j], sc[qi][j], a);
                out[((long long)b * C + c) * Nq + i] = a;
            }
        __syncthreads();
    }
}

// y = x + nearest_upsample(h, s)         x, y: [B][C][H][W], h: [B][C][H/s][W/s]
__global__ __launch_bounds__(256) void upsample_add_kernel(const float* __restrict__ x, const float* __restrict__ h, int H, int W, int s, long long total,
                                                           float* __restrict__ y) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % W), oy = (int)((idx / W) % H);
    const long long bc = idx / ((long long)W * H);
    y[idx] = x[idx] + h[(bc * (H / s) + oy / s) * (W / s) + ox / s];
}

}  // namespace
}  // namespace wdm

using namespace wdm;

extern "C" {

int wdm_conv2d_direct(wdm_handle* h, const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout, int k, int stride, int pad,
                      int groups, int transposed, float* y, void* stream) {
    if (!h || !x || !w || !y) WDM_FAIL(WDM_EINVAL, "wdm_conv2d_direct: null argument");
    if (B <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0 || pad < 0 || groups <= 0 || Cin % groups || Cout % groups)
        WDM_FAIL(WDM_EINVAL, "wdm_conv2d_direct: bad geometry");
    if (transposed && groups != 1) WDM_FAIL(WDM_EINVAL, "wdm_conv2d_direct: transposed convolutions are built for groups = 1");
    const int Ho = transposed ? (H - 1) * stride - 2 * pad + k : (H + 2 * pad - k) / stride + 1;
    const int Wo = transposed ? (W - 1) * stride - 2 * pad + k : (W + 2 * pad - k) / stride + 1;
    if (Ho <= 0 || Wo <= 0) WDM_FAIL(WDM_EINVAL, "wdm_conv2d_direct: empty output");
    const long long total = (long long)B * Cout * Ho * Wo;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (transposed) hipLaunchKernelGGL(convT2d_direct_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, B, Cin, H, W, Cout, k, stride, pad, Ho, Wo, y);
    else hipLaunchKernelGGL(conv2d_direct_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, B, Cin, H, W, Cout, k, stride, pad, groups, Ho, Wo, y);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

int wdm_groupnorm(wdm_handle* h, const float* x, const float* gamma, const float* beta, int B, int C, int H, int W, float eps, int silu, float* y, void* stream) {
    if (!h || !x || !gamma || !beta || !y) WDM_FAIL(WDM_EINVAL, "wdm_groupnorm: null argument");
    if (C % 32) WDM_FAIL(WDM_EINVAL, "wdm_groupnorm: %d channels (32 groups)", C);
    hipLaunchKernelGGL(groupnorm_nchw_kernel, dim3(32, B), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, C, H * W, eps, silu, y);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

int wdm_cross_attention(wdm_handle* h, const float* q, const float* k, const float* v, int B, int C, int Nq, int Nk, float* out, void* stream) {
    if (!h || !q || !k || !v || !out) WDM_FAIL(WDM_EINVAL, "wdm_cross_attention: null argument");
    if (Nk < 1) WDM_FAIL(WDM_EINVAL, "wdm_cross_attention: %d keys", Nk);
    const float scale = 1.0f / sqrtf((float)C);                                   // int(c) ** (-0.5), unet.py:445
    hipLaunchKernelGGL(cross_attention_kernel, dim3((Nq + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, q, k, v, C, Nq, Nk, scale, out);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

int wdm_upsample_add(wdm_handle* h, const float* x, const float* hp, int B, int C, int H, int W, int scale, float* y, void* stream) {
    if (!h || !x || !hp || !y) WDM_FAIL(WDM_EINVAL, "wdm_upsample_add: null argument");
    if (scale < 1 || H % scale || W % scale) WDM_FAIL(WDM_EINVAL, "wdm_upsample_add: %dx%d is not a multiple of %d", H, W, scale);
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(upsample_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, hp, H, W, scale, total, y);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

}  // extern "C"
