// HFRM -- the high-frequency refinement module of the reference (models/arch.py:132-253), SURVEY.md §8(f)-1: it runs once
// per image on the raw degraded image and its wavelet coefficients feed the sampler as `x_other` (restoration.py:94-102).
// NAFNet-style blocks: LayerNorm2d -> 1x1 -> depthwise 3x3 -> gate -> channel attention -> 1x1 (+beta residual) ->
// LayerNorm2d -> 1x1 -> gate -> 1x1 (+gamma residual); 2x2 stride-2 convs down, 1x1 + PixelShuffle up.
//
// Every 1x1 convolution (and the 2x2 stride-2 ones, after a space-to-depth gather) is a plain GEMM over the flattened
// pixels and runs on the fused conv kernel (MODE_P1) with bias and residual in its epilogue; beta / gamma are folded into
// the weights and bias of conv3 / conv5 when the parameters are packed.  The rest are small HBM-bound kernels below.
// Activations: NHWC in the model dtype (bf16 or f32), exactly like the UNet.
#include <string.h>

#include <algorithm>
#include <map>
#include <string>

#include "common.h"

using namespace wdm;

namespace wdm {

static inline int nblk(long long n, int bs) { return (int)((n + bs - 1) / bs); }

// ---- LayerNorm2d (arch.py:7-43): per pixel over channels, biased variance --------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln2d_kernel(const T* __restrict__ x, T* __restrict__ y, long long M, int C, const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps) {
    constexpr int VEC = TI<T>::VEC;
    const int lpp = C / VEC;                          // lanes per pixel (power of two: C in {32..512}, VEC in {4,8})
    const int ppw = 64 / lpp > 0 ? 64 / lpp : 1;      // pixels per wave (lpp <= 64), else one pixel over several passes
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lpp <= 64) {
        const long long pix = wave * ppw + lane / lpp;
        const int c = (lane % lpp) * VEC;
        const bool ok = pix < M;
        float f[VEC];
        if (ok) { TI<T>::unpack(*(const uint4*)(x + pix * C + c), f); } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += f[e];
        for (int o = 1; o < lpp; o <<= 1) s += __shfl_xor(s, o);
        const float mu = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = f[e] - mu; q += d * d; }
        for (int o = 1; o < lpp; o <<= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
        if (ok) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = (f[e] - mu) * rstd * w[c + e] + b[c + e];
            *(uint4*)(y + pix * C + c) = TI<T>::pack(f);
        }
    } else {   // C / VEC == 128 (f32, C = 512): one wave per pixel, two vectors per lane
        const long long pix = wave;
        if (pix >= M) return;
        float f[2][VEC];
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            TI<T>::unpack(*(const uint4*)(x + pix * C + (h * 64 + lane) * VEC), f[h]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += f[h][e];
        }
        for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
        const float mu = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = f[h][e] - mu; q += d * d; }
        for (int o = 1; o < 64; o <<= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = (h * 64 + lane) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[h][e] = (f[h][e] - mu) * rstd * w[c + e] + b[c + e];
            *(uint4*)(y + pix * C + c) = TI<T>::pack(f[h]);
        }
    }
}

// ---- depthwise 3x3 (pad 1) on 2d channels + SimpleGate x[:, :d] * x[:, d:] + per-block partial sums for the global
// average pool of the channel attention (arch.py:146-156, 189-192) ---------------------------------------------
// grid: (pixel blocks, B, channel blocks); a thread owns one 16-byte channel vector of the d output channels and walks
// its share of the block's pixels.  All 18 neighbour loads of a pixel are issued together (clamped address + 0/1 weight
// instead of a branch per tap), so the kernel is bandwidth- and not latency-bound.
template <typename T>
__global__ __launch_bounds__(256) void dw3x3_gate_kernel(const T* __restrict__ x, T* __restrict__ g, int H, int W, int d, const float* __restrict__ wdw,
                                                         const float* __restrict__ bdw, float* __restrict__ pool_partial, int nblocks_per_img,
                                                         int pix_per_block) {
    constexpr int VEC = TI<T>::VEC;
    __shared__ float red[256 * VEC];
    const int cols = d / VEC;                      // channel vectors of the output (d in {32..512})
    const int cb = blockIdx.z;
    const int cols_here = min(cols - cb * 256, 256);
    const int rows = 256 / cols_here;
    const int tid = threadIdx.x;
    const int col = tid % cols_here, row = tid / cols_here;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int HW = H * W;
    const int p0 = blk * pix_per_block, p1 = min(p0 + pix_per_block, HW);
    const int c = (cb * 256 + col) * VEC;
    float wa[9][VEC], wb[9][VEC], ba[VEC], bb[VEC], acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        acc[e] = 0.f;
        ba[e] = bdw[c + e]; bb[e] = bdw[d + c + e];
#pragma unroll
        for (int t = 0; t < 9; ++t) { wa[t][e] = wdw[(c + e) * 9 + t]; wb[t][e] = wdw[(d + c + e) * 9 + t]; }
    }
    if (row < rows) {
        const T* xb = x + (long long)b * HW * 2 * d;
        for (int p = p0 + row; p < p1; p += rows) {
            const int py = p / W, px = p - py * W;
            uint4 va[9], vb[9];
            float m[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int yy = py + dy - 1, xx = px + dx - 1;
                    const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                    const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
                    const T* src = xb + ((long long)yc * W + xc) * 2 * d;
                    va[dy * 3 + dx] = *(const uint4*)(src + c);
                    vb[dy * 3 + dx] = *(const uint4*)(src + d + c);
                    m[dy * 3 + dx] = in ? 1.f : 0.f;
                }
            float sa[VEC], sb[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { sa[e] = ba[e]; sb[e] = bb[e]; }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float fa[VEC], fb[VEC];
                TI<T>::unpack(va[t], fa);
                TI<T>::unpack(vb[t], fb);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { sa[e] += fa[e] * (wa[t][e] * m[t]); sb[e] += fb[e] * (wb[t][e] * m[t]); }
            }
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = sa[e] * sb[e];
            const uint4 pk = TI<T>::pack(o);
            *(uint4*)(g + ((long long)b * HW + p) * d + c) = pk;
            float orr[VEC];
            TI<T>::unpack(pk, orr);                          // pool the values as stored
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += orr[e];
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[tid * VEC + e] = acc[e];
    __syncthreads();
    if (row == 0) {
        for (int r = 1; r < rows; ++r)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += red[(r * cols_here + col) * VEC + e];
#pragma unroll
        for (int e = 0; e < VEC; ++e) pool_partial[((long long)b * nblocks_per_img + blk) * d + c + e] = acc[e];
    }
}

// pooled[b][c] = sum_blk partial / HW   (fixed order: 4 interleaved slices per channel, then the 4 slices)
__global__ __launch_bounds__(256) void pool_reduce_kernel(const float* __restrict__ partial, float* __restrict__ pooled, int B, int nblk_img, int d, float inv_hw) {
    __shared__ float red[256];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < d)
        for (int k = sl; k < nblk_img; k += 4) s += partial[((long long)b * nblk_img + k) * d + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && c < d) pooled[b * d + c] = (red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]) * inv_hw;
}

// g[b][p][c] *= s[b][c]
template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(T* __restrict__ g, const float* __restrict__ s, long long nvec, int d, int HW) {
    constexpr int VEC = TI<T>::VEC;
    const int cols = d / VEC;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nvec; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % cols) * VEC;
        const long long pix = id / cols;
        const long long b = pix / HW;
        float f[VEC];
        TI<T>::unpack(*(const uint4*)(g + pix * d + c), f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) f[e] *= s[b * d + c + e];
        *(uint4*)(g + pix * d + c) = TI<T>::pack(f);
    }
}

// SimpleGate: z[p][c] = x[p][c] * x[p][d + c]
template <typename T>
__global__ __launch_bounds__(256) void gate_kernel(const T* __restrict__ x, T* __restrict__ z, long long nvec, int d) {
    constexpr int VEC = TI<T>::VEC;
    const int cols = d / VEC;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nvec; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % cols) * VEC;
        const long long pix = id / cols;
        float fa[VEC], fb[VEC];
        TI<T>::unpack(*(const uint4*)(x + pix * 2 * d + c), fa);
        TI<T>::unpack(*(const uint4*)(x + pix * 2 * d + d + c), fb);
#pragma unroll
        for (int e = 0; e < VEC; ++e) fa[e] *= fb[e];
        *(uint4*)(z + pix * d + c) = TI<T>::pack(fa);
    }
}

// space-to-depth for the 2x2 stride-2 conv (arch.py:220): u[b][y][x][(i*2+j)*d + c] = x[b][2y+i][2x+j][c]
template <typename T>
__global__ __launch_bounds__(256) void unshuffle2_kernel(const T* __restrict__ x, T* __restrict__ u, int B, int H, int W, int d) {
    constexpr int VEC = TI<T>::VEC;
    const int cols = d / VEC, h2 = H / 2, w2 = W / 2;
    const long long total = (long long)B * h2 * w2 * 4 * cols;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int cv = (int)(id % cols);
        const int ij = (int)((id / cols) % 4);
        const long long op = id / (4 * cols);
        const int ox = (int)(op % w2), oy = (int)((op / w2) % h2);
        const long long b = op / ((long long)w2 * h2);
        const T* src = x + (((b * H + 2 * oy + (ij >> 1)) * W) + 2 * ox + (ij & 1)) * d + cv * VEC;
        *(uint4*)(u + op * 4 * d + ij * d + cv * VEC) = *(const uint4*)src;
    }
}

// PixelShuffle(2) + skip add (arch.py:228, 246-247): out[b][2y+i][2x+j][c] = p[b][y][x][c*4 + i*2 + j] + skip[...]
template <typename T>
__global__ __launch_bounds__(256) void pixel_shuffle_add_kernel(const T* __restrict__ p, const T* __restrict__ skip, T* __restrict__ out, int B, int h, int w,
                                                                int dout) {
    const long long total = (long long)B * (2 * h) * (2 * w) * dout;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % dout);
        const long long op = id / dout;
        const int X = (int)(op % (2 * w)), Y = (int)((op / (2 * w)) % (2 * h));
        const long long b = op / ((long long)4 * w * h);
        const float v = TI<T>::ld(p, ((b * h + (Y >> 1)) * w + (X >> 1)) * (4LL * dout) + c * 4 + (Y & 1) * 2 + (X & 1));
        TI<T>::st(out, id, v + TI<T>::ld(skip, id));
    }
}

// conv_in: 3x3 pad 1 on the NCHW f32 image with Cin = 3 (arch.py:210) -> NHWC [M][dim].  One thread per pixel: the cin*9 taps are
// read once (coalesced along x), the [cin*9][dim] weights sit in LDS (broadcast reads), all dim outputs leave as 16-byte stores.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void conv3x3_cin3_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int H, int W, int cin,
                                                           const float* __restrict__ w, const float* __restrict__ bias) {
    constexpr int VEC = TI<T>::VEC;
    __shared__ float ws[16 * 9 * DIM + DIM];
    for (int i = threadIdx.x; i < cin * 9 * DIM; i += 256) {
        const int co = i % DIM, k = i / DIM;                    // k = ci*9 + tap ; w is [co][ci][3][3]
        ws[i] = w[(long long)co * cin * 9 + k];
    }
    for (int i = threadIdx.x; i < DIM; i += 256) ws[cin * 9 * DIM + i] = bias[i];
    __syncthreads();
    const long long M = (long long)B * H * W;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= M) return;
    const int px = (int)(pix % W), py = (int)((pix / W) % H);
    const long long b = pix / ((long long)W * H);
    float acc[DIM];
#pragma unroll
    for (int o = 0; o < DIM; ++o) acc[o] = ws[cin * 9 * DIM + o];
    for (int ci = 0; ci < cin; ++ci) {
        const float* xp = x + (b * cin + ci) * H * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
            const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const float v = in ? xp[(long long)min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)] : 0.f;
            const float* wr = ws + (ci * 9 + t) * DIM;
#pragma unroll
            for (int o = 0; o < DIM; ++o) acc[o] += v * wr[o];
        }
    }
#pragma unroll
    for (int o = 0; o < DIM; o += VEC) *(uint4*)(y + pix * DIM + o) = TI<T>::pack(acc + o);
}

// parameter packing helpers -----------------------------------------------------------------------------------
// 1x1 / 2x2 conv weight [cout][cin][k][k] f32 -> GEMM matrix [rows_pad][k*k*cin] (row-major, k index (i*k+j)*cin + c), each row
// optionally scaled by rowscale[cout] (beta / gamma folding); bias likewise
template <typename T>
__global__ void pack_gemm_w_kernel(const float* __restrict__ w, const float* __restrict__ rowscale, int cout, int cin, int kk, T* __restrict__ dst,
                                   int rows_pad) {
    const long long total = (long long)rows_pad * cin * kk;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int kc = (int)(id % (cin * kk));
        const int o = (int)(id / (cin * kk));
        const int ij = kc / cin, c = kc % cin;
        float v = 0.f;
        if (o < cout) v = w[((long long)o * cin + c) * kk + ij] * (rowscale ? rowscale[o] : 1.f);
        TI<T>::st(dst, id, v);
    }
}
__global__ void scale_vec_kernel(const float* __restrict__ b, const float* __restrict__ s, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = b[i] * (s ? s[i] : 1.f);
}

}  // namespace wdm

// =================================================================================================
// the HFRM object
// =================================================================================================
namespace {
struct HParam { std::string name; int ndim; int64_t shape[4]; size_t raw_off; bool loaded; int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; } };
struct GemmD { size_t w_off, b_off; int cin, cout, rows_pad; };   // packed GEMM weight [rows_pad][cin] (T) + bias f32
struct BlockD {
    int d;
    int p_beta, p_gamma, p_n1w, p_n1b, p_n2w, p_n2b, p_dww, p_dwb, p_caw, p_cab;   // raw parameter indices
    int p_w[5], p_b[5];                                                            // conv1..conv5 raw indices (conv2 = depthwise)
    GemmD g1, g3, g4, g5;
};
}  // namespace

struct wdm_hfrm {
    wdm_handle* h;
    wdm_hfrm_config cfg;
    std::vector<HParam> params;
    std::map<std::string, int> index;
    size_t raw_bytes = 0, packed_bytes = 0;
    char* packed = nullptr;
    bool finalized = false;
    int p_cin_w, p_cin_b, p_cout_w, p_cout_b;
    GemmD g_out;                    // conv_out runs on the 3x3 conv kernel: slab layout handled by k_pack_conv
    size_t cout_w_off = 0;
    std::vector<std::vector<BlockD>> enc, dec;
    std::vector<BlockD> mid;
    std::vector<int> p_down_w, p_down_b, p_up_w;
    std::vector<GemmD> g_down, g_up;

    size_t take(size_t bytes) { size_t o = packed_bytes; packed_bytes = align_up(packed_bytes + bytes, 256); return o; }
    int add(const std::string& name, std::initializer_list<int64_t> shp) {
        HParam p; p.name = name; p.ndim = (int)shp.size(); int i = 0; for (auto v : shp) p.shape[i++] = v; for (; i < 4; ++i) p.shape[i] = 0;
        p.raw_off = raw_bytes; p.loaded = false;
        raw_bytes = align_up(raw_bytes + (size_t)p.numel() * 4, 256);
        index[name] = (int)params.size();
        params.push_back(p);
        return (int)params.size() - 1;
    }
    GemmD gemm(int cin, int cout) {
        GemmD g; g.cin = cin; g.cout = cout; g.rows_pad = conv_rows_pad(cout);
        g.w_off = take((size_t)g.rows_pad * cin * dsize(cfg.dtype)); g.b_off = take((size_t)cout * 4);
        return g;
    }
    BlockD block(const std::string& n, int d) {
        BlockD b; b.d = d;
        b.p_beta = add(n + ".beta", {1, d, 1, 1}); b.p_gamma = add(n + ".gamma", {1, d, 1, 1});
        const char* cn[5] = {"conv1", "conv2", "conv3", "conv4", "conv5"};
        const int co[5] = {2 * d, 2 * d, d, 2 * d, d}, ci[5] = {d, 1, d, d, d}, kk[5] = {1, 3, 1, 1, 1};
        for (int k = 0; k < 3; ++k) { b.p_w[k] = add(n + "." + cn[k] + ".weight", {co[k], ci[k], kk[k], kk[k]}); b.p_b[k] = add(n + "." + cn[k] + ".bias", {co[k]}); }
        b.p_caw = add(n + ".channel_attn.chan_conv.weight", {d, d, 1, 1}); b.p_cab = add(n + ".channel_attn.chan_conv.bias", {d});
        for (int k = 3; k < 5; ++k) { b.p_w[k] = add(n + "." + cn[k] + ".weight", {co[k], ci[k], kk[k], kk[k]}); b.p_b[k] = add(n + "." + cn[k] + ".bias", {co[k]}); }
        b.p_n1w = add(n + ".norm1.weight", {d}); b.p_n1b = add(n + ".norm1.bias", {d});
        b.p_n2w = add(n + ".norm2.weight", {d}); b.p_n2b = add(n + ".norm2.bias", {d});
        b.p_dww = b.p_w[1]; b.p_dwb = b.p_b[1];
        b.g1 = gemm(d, 2 * d); b.g3 = gemm(d, d); b.g4 = gemm(d, 2 * d); b.g5 = gemm(d, d);
        return b;
    }
    const float* raw(int pi) const { return (const float*)(packed + params[pi].raw_off); }
    ConvW cw(const GemmD& g) const { ConvW w; w.w = packed + raw_bytes + g.w_off; w.b = (const float*)(packed + raw_bytes + g.b_off); w.cin = g.cin; w.cout = g.cout; w.k = 1; w.rows_pad = g.rows_pad; return w; }

    int build();
    int finalize(hipStream_t s);
    int run_block(Ctx& c, const BlockD& b, Tens& t, int B, int H, int W);
    int gemm_rows(Ctx& c, const GemmD& g, const void* x, long long M, const void* res, void* y);
    int forward(Ctx& c, const float* x, int B, int H, int W, float* y);
};

int wdm_hfrm::build() {
    const int dim = cfg.dim;
    p_cin_w = add("conv_in.weight", {dim, cfg.in_channel, 3, 3}); p_cin_b = add("conv_in.bias", {dim});
    int d = dim;
    enc.resize(cfg.n_enc); dec.resize(cfg.n_dec);
    for (int i = 0; i < cfg.n_enc; ++i) {
        for (int j = 0; j < cfg.enc_blk_nums[i]; ++j) enc[i].push_back(block("encoders." + std::to_string(i) + "." + std::to_string(j), d));
        d *= 2;
    }
    const int dmid = d;
    for (int i = 0; i < cfg.n_dec; ++i) {
        d /= 2;
        for (int j = 0; j < cfg.dec_blk_nums[i]; ++j) dec[i].push_back(block("decoders." + std::to_string(i) + "." + std::to_string(j), d));
    }
    for (int j = 0; j < cfg.mid_blk_num; ++j) mid.push_back(block("mid_blks." + std::to_string(j), dmid));
    d = dmid;
    for (int i = 0; i < cfg.n_dec; ++i) { p_up_w.push_back(add("ups." + std::to_string(i) + ".0.weight", {2 * d, d, 1, 1})); g_up.push_back(gemm(d, 2 * d)); d /= 2; }
    d = dim;
    for (int i = 0; i < cfg.n_enc; ++i) {
        p_down_w.push_back(add("downs." + std::to_string(i) + ".weight", {2 * d, d, 2, 2})); p_down_b.push_back(add("downs." + std::to_string(i) + ".bias", {2 * d}));
        g_down.push_back(gemm(4 * d, 2 * d));
        d *= 2;
    }
    p_cout_w = add("conv_out.weight", {cfg.in_channel, dim, 3, 3}); p_cout_b = add("conv_out.bias", {cfg.in_channel});
    cout_w_off = take(conv_packed_bytes(dim, cfg.in_channel, 3, cfg.dtype));
    return WDM_OK;
}

template <typename T>
static void pack_gemm(const float* w, const float* rowscale, const float* b, int cout, int cin, int kk, void* wdst, float* bdst, int rows_pad, hipStream_t s) {
    const long long total = (long long)rows_pad * cin * kk;
    hipLaunchKernelGGL(pack_gemm_w_kernel<T>, dim3(nblk(total, 256) > 4096 ? 4096 : nblk(total, 256)), dim3(256), 0, s, w, rowscale, cout, cin, kk, (T*)wdst, rows_pad);
    if (b) hipLaunchKernelGGL(scale_vec_kernel, dim3(nblk(cout, 256)), dim3(256), 0, s, b, rowscale, bdst, cout);
    else (void)hipMemsetAsync(bdst, 0, (size_t)cout * 4, s);
}

int wdm_hfrm::finalize(hipStream_t s) {
    for (auto& p : params) if (!p.loaded) WDM_FAIL(WDM_ESTATE, "wdm_hfrm_finalize: parameter '%s' not loaded", p.name.c_str());
    char* pk = packed + raw_bytes;
    auto pg = [&](const GemmD& g, int pw, int pb, int prs, int kk) {
        const int cin_raw = g.cin / kk;
        if (cfg.dtype == WDM_BF16) pack_gemm<__bf16>(raw(pw), prs >= 0 ? raw(prs) : nullptr, pb >= 0 ? raw(pb) : nullptr, g.cout, cin_raw, kk, pk + g.w_off, (float*)(pk + g.b_off), g.rows_pad, s);
        else pack_gemm<float>(raw(pw), prs >= 0 ? raw(prs) : nullptr, pb >= 0 ? raw(pb) : nullptr, g.cout, cin_raw, kk, pk + g.w_off, (float*)(pk + g.b_off), g.rows_pad, s);
    };
    auto pblock = [&](const BlockD& b) {
        pg(b.g1, b.p_w[0], b.p_b[0], -1, 1);
        pg(b.g3, b.p_w[2], b.p_b[2], b.p_beta, 1);       // y = x + beta * conv3(.)  ->  beta folded into conv3
        pg(b.g4, b.p_w[3], b.p_b[3], -1, 1);
        pg(b.g5, b.p_w[4], b.p_b[4], b.p_gamma, 1);      // out = y + gamma * conv5(.)
    };
    for (auto& lv : enc) for (auto& b : lv) pblock(b);
    for (auto& lv : dec) for (auto& b : lv) pblock(b);
    for (auto& b : mid) pblock(b);
    for (size_t i = 0; i < g_up.size(); ++i) pg(g_up[i], p_up_w[i], -1, -1, 1);
    for (size_t i = 0; i < g_down.size(); ++i) pg(g_down[i], p_down_w[i], p_down_b[i], -1, 4);
    WDM_TRY(k_pack_conv(raw(p_cout_w), cfg.in_channel, cfg.dim, 3, pk + cout_w_off, conv_rows_pad(cfg.in_channel), 0, 1, cfg.dtype, s));
    WDM_HIP(hipGetLastError());
    finalized = true;
    return WDM_OK;
}

// y[M][cout] = x[M][cin] . W^T + bias (+ res): plain GEMM on the conv kernel, pixels flattened onto a 16-wide grid
int wdm_hfrm::gemm_rows(Ctx& c, const GemmD& g, const void* x, long long M, const void* res, void* y) {
    if (c.dry) return WDM_OK;
    const int Hp = (int)align_up((size_t)((M + 15) / 16), 16);
    ConvArgs a{};
    a.x0 = x; a.C0 = g.cin; a.xs0 = g.cin; a.C1 = 0;
    a.B = 1; a.Hin = a.Hout = Hp; a.Win = a.Wout = 16;
    a.Cin = g.cin; a.Cout = g.cout;
    const ConvW w = cw(g);
    a.w = w.w; a.w_tap_stride = 0; a.w_img_stride = 0; a.w_row_stride = g.cin; a.w_rows = g.rows_pad;
    a.w_bytes = (unsigned)((size_t)g.rows_pad * g.cin * dsize(c.dtype));
    a.bias = w.b; a.alpha = 1.f;
    a.res = res; a.res_s = g.cout;
    a.y = y; a.y_mode = Y_NHWC; a.y_s = g.cout;
    a.m_valid = M;
    // the descriptor extents must describe the real tensor (M rows), not the padded grid
    int rc = launch_conv(a, MODE_P1, c.dtype, c.s);
    return rc;
}

int wdm_hfrm::run_block(Ctx& c, const BlockD& b, Tens& t, int B, int H, int W) {
    const int d = b.d, HW = H * W;
    const long long M = (long long)B * HW;
    const size_t es = dsize(c.dtype);
    const int vec = c.dtype == WDM_BF16 ? 8 : 4;
    auto A = [&](size_t bytes) -> void* { return c.ar->alloc(bytes); };
    void* n1 = A((size_t)M * d * es);
    void* a2 = A((size_t)M * 2 * d * es);
    void* g = A((size_t)M * d * es);
    const int cols_blk = std::min(d / vec, 256);
    const int ppb = (256 / cols_blk) * 8;                   // pixels per pooling block: 8 per thread
    const int nb = (HW + ppb - 1) / ppb;
    float* part = (float*)A((size_t)B * nb * d * 4);
    float* pooled = (float*)A((size_t)B * d * 4);
    float* sc = (float*)A((size_t)B * d * 4);
    void* y = A((size_t)M * d * es);
    if (!n1 || !a2 || !g || !part || !pooled || !sc || !y) WDM_FAIL(WDM_ENOMEM, "workspace too small (HFRM block)");
    if (!c.dry) {
        const float* n1w = raw(b.p_n1w); const float* n1b = raw(b.p_n1b);
        const int lpp = d / vec, ppw = lpp <= 64 ? 64 / lpp : 1;
        const long long waves = (M + ppw - 1) / ppw;
        if (c.dtype == WDM_BF16) hipLaunchKernelGGL(ln2d_kernel<__bf16>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c.s, (const __bf16*)t.p, (__bf16*)n1, M, d, n1w, n1b, 1e-6f);
        else hipLaunchKernelGGL(ln2d_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c.s, (const float*)t.p, (float*)n1, M, d, n1w, n1b, 1e-6f);
        WDM_TRY(gemm_rows(c, b.g1, n1, M, nullptr, a2));
        const int cols = d / vec;
        const dim3 grid(nb, B, (cols + 255) / 256);
        if (c.dtype == WDM_BF16) hipLaunchKernelGGL(dw3x3_gate_kernel<__bf16>, grid, dim3(256), 0, c.s, (const __bf16*)a2, (__bf16*)g, H, W, d, raw(b.p_dww), raw(b.p_dwb), part, nb, ppb);
        else hipLaunchKernelGGL(dw3x3_gate_kernel<float>, grid, dim3(256), 0, c.s, (const float*)a2, (float*)g, H, W, d, raw(b.p_dww), raw(b.p_dwb), part, nb, ppb);
        hipLaunchKernelGGL(pool_reduce_kernel, dim3((d + 63) / 64, B), dim3(256), 0, c.s, part, pooled, B, nb, d, 1.0f / (float)HW);
        WDM_TRY(k_linear(pooled, B, d, raw(b.p_caw), raw(b.p_cab), d, sc, 0, c.s));
        const long long nvec = M * cols;
        const int gg = nblk(nvec, 256) > 16384 ? 16384 : nblk(nvec, 256);
        if (c.dtype == WDM_BF16) hipLaunchKernelGGL(scale_channels_kernel<__bf16>, dim3(gg), dim3(256), 0, c.s, (__bf16*)g, sc, nvec, d, HW);
        else hipLaunchKernelGGL(scale_channels_kernel<float>, dim3(gg), dim3(256), 0, c.s, (float*)g, sc, nvec, d, HW);
        WDM_TRY(gemm_rows(c, b.g3, g, M, t.p, y));                                   // y = x + beta*conv3(.)
        const float* n2w = raw(b.p_n2w); const float* n2b = raw(b.p_n2b);
        if (c.dtype == WDM_BF16) hipLaunchKernelGGL(ln2d_kernel<__bf16>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c.s, (const __bf16*)y, (__bf16*)n1, M, d, n2w, n2b, 1e-6f);
        else hipLaunchKernelGGL(ln2d_kernel<float>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c.s, (const float*)y, (float*)n1, M, d, n2w, n2b, 1e-6f);
        WDM_TRY(gemm_rows(c, b.g4, n1, M, nullptr, a2));
        if (c.dtype == WDM_BF16) hipLaunchKernelGGL(gate_kernel<__bf16>, dim3(gg), dim3(256), 0, c.s, (const __bf16*)a2, (__bf16*)g, nvec, d);
        else hipLaunchKernelGGL(gate_kernel<float>, dim3(gg), dim3(256), 0, c.s, (const float*)a2, (float*)g, nvec, d);
        WDM_TRY(gemm_rows(c, b.g5, g, M, y, t.p));                                   // out = y + gamma*conv5(.), written over the block input
        WDM_HIP(hipGetLastError());
    }
    c.ar->free(n1); c.ar->free(a2); c.ar->free(g); c.ar->free(part); c.ar->free(pooled); c.ar->free(sc); c.ar->free(y);
    return WDM_OK;
}

int wdm_hfrm::forward(Ctx& c, const float* x, int B, int H, int W, float* yout) {
    const size_t es = dsize(c.dtype);
    const int nlev = cfg.n_enc;
    if (H % (1 << nlev) || W % (1 << nlev)) WDM_FAIL(WDM_EINVAL, "HFRM: H=%d W=%d must be multiples of %d", H, W, 1 << nlev);
    if (H % 8 || W % 8) WDM_FAIL(WDM_EINVAL, "HFRM: H and W must be multiples of 8");
    int d = cfg.dim, h = H, w = W;
    Tens t; t.C = d; t.H = h; t.W = w; t.xs = d;
    t.p = c.ar->alloc((size_t)B * h * w * d * es);
    void* xin = c.ar->alloc((size_t)B * h * w * cfg.in_channel * es);          // NHWC copy of the input for the final residual
    if (!t.p || !xin) WDM_FAIL(WDM_ENOMEM, "workspace too small (HFRM)");
    if (!c.dry) {
        const int gg = nblk((long long)B * h * w, 256);
        if (c.dtype == WDM_BF16) hipLaunchKernelGGL((conv3x3_cin3_kernel<__bf16, 32>), dim3(gg), dim3(256), 0, c.s, x, (__bf16*)t.p, B, h, w, cfg.in_channel, raw(p_cin_w), raw(p_cin_b));
        else hipLaunchKernelGGL((conv3x3_cin3_kernel<float, 32>), dim3(gg), dim3(256), 0, c.s, x, (float*)t.p, B, h, w, cfg.in_channel, raw(p_cin_w), raw(p_cin_b));
        WDM_TRY(k_nchw_to_nhwc(x, xin, B, cfg.in_channel, h, w, c.dtype, c.s));
    }
    std::vector<Tens> encs;
    for (int i = 0; i < nlev; ++i) {
        for (auto& b : enc[i]) WDM_TRY(run_block(c, b, t, B, h, w));
        encs.push_back(t);
        // down: space-to-depth + GEMM (4d -> 2d)
        void* u = c.ar->alloc((size_t)B * (h / 2) * (w / 2) * 4 * d * es);
        Tens nt; nt.C = 2 * d; nt.H = h / 2; nt.W = w / 2; nt.xs = 2 * d;
        nt.p = c.ar->alloc((size_t)B * nt.H * nt.W * nt.C * es);
        if (!u || !nt.p) WDM_FAIL(WDM_ENOMEM, "workspace too small (HFRM down)");
        if (!c.dry) {
            const long long total = (long long)B * nt.H * nt.W * 4 * (d / (c.dtype == WDM_BF16 ? 8 : 4));
            const int gg = nblk(total, 256) > 16384 ? 16384 : nblk(total, 256);
            if (c.dtype == WDM_BF16) hipLaunchKernelGGL(unshuffle2_kernel<__bf16>, dim3(gg), dim3(256), 0, c.s, (const __bf16*)t.p, (__bf16*)u, B, h, w, d);
            else hipLaunchKernelGGL(unshuffle2_kernel<float>, dim3(gg), dim3(256), 0, c.s, (const float*)t.p, (float*)u, B, h, w, d);
            WDM_TRY(gemm_rows(c, g_down[i], u, (long long)B * nt.H * nt.W, nullptr, nt.p));
        }
        c.ar->free(u);
        t = nt; d *= 2; h /= 2; w /= 2;
    }
    for (auto& b : mid) WDM_TRY(run_block(c, b, t, B, h, w));
    for (int i = 0; i < cfg.n_dec; ++i) {
        // up: 1x1 (d -> 2d, no bias) + PixelShuffle(2) + skip
        void* p = c.ar->alloc((size_t)B * h * w * 2 * d * es);
        Tens skip = encs.back(); encs.pop_back();
        Tens nt; nt.C = d / 2; nt.H = 2 * h; nt.W = 2 * w; nt.xs = d / 2;
        nt.p = c.ar->alloc((size_t)B * nt.H * nt.W * nt.C * es);
        if (!p || !nt.p) WDM_FAIL(WDM_ENOMEM, "workspace too small (HFRM up)");
        if (!c.dry) {
            WDM_TRY(gemm_rows(c, g_up[i], t.p, (long long)B * h * w, nullptr, p));
            const long long total = (long long)B * nt.H * nt.W * nt.C;
            const int gg = nblk(total, 256) > 16384 ? 16384 : nblk(total, 256);
            if (c.dtype == WDM_BF16) hipLaunchKernelGGL(pixel_shuffle_add_kernel<__bf16>, dim3(gg), dim3(256), 0, c.s, (const __bf16*)p, (const __bf16*)skip.p, (__bf16*)nt.p, B, h, w, nt.C);
            else hipLaunchKernelGGL(pixel_shuffle_add_kernel<float>, dim3(gg), dim3(256), 0, c.s, (const float*)p, (const float*)skip.p, (float*)nt.p, B, h, w, nt.C);
        }
        c.ar->free(p); c.ar->free(t.p); c.ar->free(skip.p);
        t = nt; d /= 2; h *= 2; w *= 2;
        for (auto& b : dec[i]) WDM_TRY(run_block(c, b, t, B, h, w));
    }
    // conv_out 3x3 (dim -> 3) + input, NCHW f32 out
    {
        ConvW cwo; cwo.w = packed + raw_bytes + cout_w_off; cwo.b = raw(p_cout_b); cwo.cin = cfg.dim; cwo.cout = cfg.in_channel; cwo.k = 3; cwo.rows_pad = conv_rows_pad(cfg.in_channel);
        Tens xi; xi.p = xin; xi.C = cfg.in_channel; xi.H = h; xi.W = w; xi.xs = cfg.in_channel;
        Tens dummy;
        Ctx cc = c; cc.B = B;
        WDM_TRY(run_conv(cc, cwo, MODE_S1, t, nullptr, nullptr, nullptr, nullptr, 0, 0, &xi, &dummy, Y_NCHW_F32, yout));
    }
    c.ar->free(t.p); c.ar->free(xin);
    return WDM_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int wdm_hfrm_create(wdm_handle* h, const wdm_hfrm_config* cfg, wdm_hfrm** out) {
    if (!cfg || !out) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_create: null argument");
    if (cfg->n_enc < 1 || cfg->n_enc > 8 || cfg->n_dec != cfg->n_enc) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_create: encoder/decoder level counts must match (1..8)");
    if (cfg->dim != 32 || cfg->in_channel < 1 || cfg->in_channel > 16) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_create: dim must be 32 (the reference's width), in_channel <= 16");
    if (cfg->dtype != WDM_BF16 && cfg->dtype != WDM_F32) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_create: bad dtype");
    wdm_hfrm* m = new wdm_hfrm();
    m->h = h; m->cfg = *cfg;
    m->build();
    *out = m;
    return WDM_OK;
}
int wdm_hfrm_destroy(wdm_hfrm* m) { delete m; return WDM_OK; }
int wdm_hfrm_num_params(const wdm_hfrm* m) { return m ? (int)m->params.size() : 0; }
int wdm_hfrm_param_info(const wdm_hfrm* m, int i, const char** name, int* ndim, int64_t shape[4]) {
    if (!m || i < 0 || i >= (int)m->params.size()) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_param_info: index out of range");
    const HParam& p = m->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = p.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    return WDM_OK;
}
size_t wdm_hfrm_packed_bytes(const wdm_hfrm* m) { return m ? m->raw_bytes + m->packed_bytes : 0; }
int wdm_hfrm_set_packed(wdm_hfrm* m, void* packed, size_t bytes) {
    if (!m || !packed) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_set_packed: null argument");
    if (bytes < m->raw_bytes + m->packed_bytes) WDM_FAIL(WDM_ENOMEM, "wdm_hfrm_set_packed: buffer too small");
    if (((uintptr_t)packed) & 255) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_set_packed: buffer must be 256-byte aligned");
    m->packed = (char*)packed; m->finalized = false;
    for (auto& p : m->params) p.loaded = false;
    return WDM_OK;
}
int wdm_hfrm_load_param(wdm_hfrm* m, const char* name, const float* dev_src, int64_t numel, void* stream) {
    if (!m || !name || !dev_src) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_load_param: null argument");
    if (!m->packed) WDM_FAIL(WDM_ESTATE, "wdm_hfrm_load_param: call wdm_hfrm_set_packed first");
    auto it = m->index.find(name);
    if (it == m->index.end()) WDM_FAIL(WDM_ENOTFOUND, "unknown HFRM parameter '%s'", name);
    HParam& p = m->params[it->second];
    if (numel != p.numel()) WDM_FAIL(WDM_EINVAL, "HFRM parameter '%s': %lld elements given, %lld expected", name, (long long)numel, (long long)p.numel());
    WDM_TRY(k_copy_f32(dev_src, (float*)(m->packed + p.raw_off), numel, (hipStream_t)stream));
    p.loaded = true; m->finalized = false;
    return WDM_OK;
}
int wdm_hfrm_finalize(wdm_hfrm* m, void* stream) {
    if (!m || !m->packed) WDM_FAIL(WDM_ESTATE, "wdm_hfrm_finalize: no packed buffer");
    return m->finalize((hipStream_t)stream);
}
size_t wdm_hfrm_workspace_bytes(const wdm_hfrm* m, int B, int H, int W) {
    if (!m || B <= 0) return 0;
    Arena ar = Arena::dry();
    Ctx c{nullptr, m->cfg.dtype, B, &ar, true};
    if (const_cast<wdm_hfrm*>(m)->forward(c, nullptr, B, H, W, nullptr) != WDM_OK) return 0;
    return ar.peak() + 4096;
}
int wdm_hfrm_forward(wdm_hfrm* m, const float* x, int B, int H, int W, float* y, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !x || !y || !workspace) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_forward: null argument");
    if (!m->finalized) WDM_FAIL(WDM_ESTATE, "wdm_hfrm_forward: parameters not loaded / finalized");
    if (((uintptr_t)workspace) & 255) WDM_FAIL(WDM_EINVAL, "wdm_hfrm_forward: workspace must be 256-byte aligned");
    Arena ar(workspace, workspace_bytes);
    Ctx c{(hipStream_t)stream, m->cfg.dtype, B, &ar, false};
    return m->forward(c, x, B, H, W, y);
}

}  // extern "C"
