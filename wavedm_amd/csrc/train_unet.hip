// Training step of the wavelet-domain DiffusionUNet: noise_estimation_loss + backward + Adam + EMA
// (reference: models/ddm_wavelet.py:108-124 loss, :259-272 optimiser step, :34-60 EMAHelper, utils/optimize.py:5-8).
//
// The forward pass here is the inference graph un-fused: every GroupNorm+SiLU output, every conv output and the attention
// intermediates are materialised and kept (dropout is 0 in raindrop_wavelet.yml, so train() and eval() compute the same function);
// each op pushes a closure on a tape, the backward pass runs the tape in reverse.  Contractions run on the forward conv kernels
// (train.hip: dgrad = conv with transposed weights, wgrad = batched pixel-contraction GEMMs).  Parameters, gradients, Adam moments
// and the EMA shadow are five flat fp32 buffers with one layout (wdm_trainer_param_info), so the optimiser is one elementwise kernel
// and a DDP all-reduce is one collective over the gradient buffer.  Deterministic: no atomics, fixed reduction orders.
#include <deque>
#include <functional>
#include <map>
#include <string>

#include "common.h"

using namespace wdm;

namespace wdm {

static inline int nb(long long n, int bs) { long long g = (n + bs - 1) / bs; return (int)(g > 16384 ? 16384 : g); }      // grid-stride kernels
static inline int nbu(long long n, int bs) { return (int)((n + bs - 1) / bs); }                                            // one element per thread

// ---- small kernels -------------------------------------------------------------------------------------------------
// x96[b][p][c] (NHWC, model dtype): c in [c_t0, c_t0 + 3): x0*sa[b] + e*s1m[b]  (q-sample, ddm_wavelet.py:112), else x0
template <typename T>
__global__ __launch_bounds__(256) void build_input_kernel(const float* __restrict__ x0, const float* __restrict__ e, const float* __restrict__ sa,
                                                          const float* __restrict__ s1m, int C, int HW, int c_t0, int pc, T* __restrict__ x96, long long total) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % C);
        const long long bp = id / C;
        const long long b = bp / HW;
        const int p = (int)(bp - b * HW);
        float v = x0[(b * C + c) * HW + p];
        if (c >= c_t0 && c < c_t0 + pc) v = v * sa[b] + e[(b * pc + (c - c_t0)) * HW + p] * s1m[b];
        TI<T>::st(x96, id, v);
    }
}
// loss = sum_b sum (e - out)^2 / B (ddm_wavelet.py:121, :124); dout = 2 (out - e) / B.  out: NHWC fp32 [B][HW][pc]; e: NCHW.
// sample_w (optional, training.use_mse): the objective that is differentiated is mean_b w_b sum (e - out)^2 with w_b = (1 - abar_t) / abar_t,
// which is the reference's mse_loss = sum (x_tar - x0_pred)^2 (:120, :122; x_tar - x0_pred = (out - e) sqrt((1 - abar) / abar)); the value
// written to *loss stays the unweighted one the reference prints and returns.
// Loss and its gradient in two launches: LOSS_WGS workgroups each take a contiguous range of elements (fp64 partial, fixed tree), one more workgroup adds the partials
// in ascending order.  (One 1 024-thread workgroup over all 786 K elements with three run-time divisions per element took 0.58 ms, 1.8 % of a 64-sample step.)
constexpr int LOSS_WGS = 256;
template <typename T>
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ out, const float* __restrict__ e, int B, int pc, int HW, T* __restrict__ dout,
                                                   double* __restrict__ partial, float* __restrict__ out_nchw, const float* __restrict__ sqrt_a,
                                                   const float* __restrict__ sqrt_1ma) {
    __shared__ double red[256];
    const long long total = (long long)B * HW * pc;
    const long long per = (total + gridDim.x - 1) / gridDim.x;
    const long long i0 = (long long)blockIdx.x * per, i1 = i0 + per < total ? i0 + per : total;
    double s = 0.0;
    for (long long id = i0 + threadIdx.x; id < i1; id += 256) {
        const int c = (int)(id % pc);
        const long long bp = id / pc;
        const long long b = bp / HW;
        const int p = (int)(bp - b * HW);
        const long long en = (b * pc + c) * HW + p;
        const float o = out[id], d = o - e[en];
        s += (double)d * d;
        float w = 1.0f;
        if (sqrt_a) { const float r = sqrt_1ma[b] / sqrt_a[b]; w = r * r; }
        TI<T>::st(dout, id, 2.0f * w * d / (float)B);
        if (out_nchw) out_nchw[en] = o;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(64) void loss_final_kernel(const double* __restrict__ partial, int n, int B, float* __restrict__ loss) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += partial[i];
    *loss = (float)(s / (double)B);
}
template <typename T>
__global__ __launch_bounds__(256) void add_into_kernel(T* __restrict__ dst, const T* __restrict__ src, long long n, int accumulate) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x)
        TI<T>::st(dst, id, TI<T>::ld(src, id) + (accumulate ? TI<T>::ld(dst, id) : 0.f));
}
// src dense [rows][C0 + C1] -> d0 [rows][C0] (+=), d1 [rows][C1] (+=)
template <typename T>
__global__ __launch_bounds__(256) void split_add_kernel(const T* __restrict__ src, int C0, int C1, T* __restrict__ d0, int acc0, T* __restrict__ d1, int acc1, long long total) {
    const int C = C0 + C1;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(id % C);
        const long long r = id / C;
        const float v = TI<T>::ld(src, id);
        if (c < C0) { const long long o = r * C0 + c; TI<T>::st(d0, o, v + (acc0 ? TI<T>::ld(d0, o) : 0.f)); }
        else { const long long o = r * C1 + (c - C0); TI<T>::st(d1, o, v + (acc1 ? TI<T>::ld(d1, o) : 0.f)); }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void f32_to_t_kernel(const float* __restrict__ src, T* __restrict__ dst, long long n) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) TI<T>::st(dst, id, src[id]);
}
// attention softmax backward: dS = P * (dP - sum_j dP P) * scale, one wave per row; dP fp32, P / dS model dtype
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, const float* __restrict__ dP, T* __restrict__ dS, long long rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) acc += dP[row * n + j] * TI<T>::ld(P, row * n + j);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    for (int j = lane; j < n; j += 64) { const float p = TI<T>::ld(P, row * n + j); TI<T>::st(dS, row * n + j, p * (dP[row * n + j] - acc) * scale); }
}
// fp32 elementwise for the temb MLP
__global__ void silu_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) { const float v = x[id]; y[id] = v / (1.0f + expf(-v)); }
}
__global__ void silu_bwd_f32_kernel(const float* __restrict__ pre, const float* __restrict__ dy, float* __restrict__ dx, long long n) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) { const float v = pre[id], s = 1.0f / (1.0f + expf(-v)); dx[id] = dy[id] * s * (1.0f + v * (1.0f - s)); }
}
// dW[o][k] = sum_n dy[n][o] x[n][k];  dx[n][k] = sum_o dy[n][o] W[o][k]   (Linear backward, tiny n)
__global__ void lin_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, int n, int o, int k, float* __restrict__ dW) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)o * k) return;
    const int kk = (int)(id % k), oo = (int)(id / k);
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += dy[(long long)i * o + oo] * x[(long long)i * k + kk];
    dW[id] = s;
}
// two stages (64 slices of the o range, then their sum in slice order) so that the [rows][4ch] projection matrix is read by many blocks
__global__ void lin_bwd_x_part_kernel(const float* __restrict__ dy, const float* __restrict__ W, int n, int o, int k, float* __restrict__ part) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)n * k) return;
    const int kk = (int)(id % k), nn = (int)(id / k);
    const int sl = blockIdx.y, per = (o + 63) / 64;
    const int o0 = sl * per, o1 = o0 + per < o ? o0 + per : o;
    float s = 0.f;
    for (int i = o0; i < o1; ++i) s += dy[(long long)nn * o + i] * W[(long long)i * k + kk];
    part[(long long)sl * n * k + id] = s;
}
__global__ void lin_bwd_x_final_kernel(const float* __restrict__ part, long long nk, float* __restrict__ dx) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nk) return;
    float s = 0.f;
    for (int sl = 0; sl < 64; ++sl) s += part[(long long)sl * nk + id];
    dx[id] = s;
}
// torch.optim.Adam (amsgrad = False) + EMAHelper.update, one pass over the flat buffers
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M, float* __restrict__ V,
                                                       float* __restrict__ E, long long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                       float bc2_sqrt, float mu) {
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (long long)gridDim.x * blockDim.x) {
        float g = G[id];
        const float p = P[id];
        if (wd != 0.f) g += wd * p;
        const float m = b1 * M[id] + (1.f - b1) * g;
        const float v = b2 * V[id] + (1.f - b2) * g * g;
        M[id] = m; V[id] = v;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        const float pn = p - (lr / bc1) * (m / denom);
        P[id] = pn;
        if (E) E[id] = (1.0f - mu) * pn + mu * E[id];
    }
}

template <typename T> static void l_add_into(hipStream_t s, void* dst, const void* src, long long n, int acc) {
    hipLaunchKernelGGL(add_into_kernel<T>, dim3(nb(n, 256)), dim3(256), 0, s, (T*)dst, (const T*)src, n, acc);
}
template <typename T> static void l_split_add(hipStream_t s, const void* src, int C0, int C1, void* d0, int a0, void* d1, int a1, long long total) {
    hipLaunchKernelGGL(split_add_kernel<T>, dim3(nb(total, 256)), dim3(256), 0, s, (const T*)src, C0, C1, (T*)d0, a0, (T*)d1, a1, total);
}
template <typename T> static void l_f32_to_t(hipStream_t s, const float* src, void* dst, long long n) {
    hipLaunchKernelGGL(f32_to_t_kernel<T>, dim3(nb(n, 256)), dim3(256), 0, s, src, (T*)dst, n);
}
template <typename T> static void l_softmax_bwd(hipStream_t s, const void* P, const float* dP, void* dS, long long rows, int n, float scale) {
    hipLaunchKernelGGL(softmax_bwd_kernel<T>, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, (const T*)P, dP, (T*)dS, rows, n, scale);
}
template <typename T> static void l_build_input(hipStream_t s, const float* x0, const float* e, const float* sa, const float* s1m, int C, int HW, int c_t0, int pc, void* x96,
                                                long long total) {
    hipLaunchKernelGGL(build_input_kernel<T>, dim3(nb(total, 256)), dim3(256), 0, s, x0, e, sa, s1m, C, HW, c_t0, pc, (T*)x96, total);
}
template <typename T> static void l_loss(hipStream_t s, const float* out, const float* e, int B, int pc, int HW, void* dout, float* loss, float* out_nchw,
                                         const float* sqrt_a, const float* sqrt_1ma, double* partial) {
    hipLaunchKernelGGL(loss_kernel<T>, dim3(LOSS_WGS), dim3(256), 0, s, out, e, B, pc, HW, (T*)dout, partial, out_nchw, sqrt_a, sqrt_1ma);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, partial, LOSS_WGS, B, loss);
}
// transposed copy used by the attention backward: dst[b][c][n] = src[b][n][c]   (train.hip's gather with stride 1, offset 0)
int transpose_tokens(Ctx& c, const void* src, int N, int Cc, void* dst);
void l_colsum_f32(hipStream_t s, const float* x, int C, int rows, float* out, float* scratch);

#define BYT(DT, FN, ...) do { if ((DT) == WDM_BF16) FN<__bf16>(__VA_ARGS__); else FN<float>(__VA_ARGS__); } while (0)

}  // namespace wdm

// =====================================================================================================================
namespace {
struct PInfo { std::string name; int ndim; int64_t shape[4]; size_t off; };
struct ConvP { size_t w = 0, b = 0; int cin = 0, cout = 0, k = 0; };
struct NormP { size_t g = 0, b = 0; int c = 0; };
struct ResP { int cin, cout; NormP n1, n2; ConvP c1, c2, nin; bool has_nin; int temb_row; };
struct AttnP { int c; NormP n; ConvP q, k, v, proj; };
struct TT { Tens t; void* g = nullptr; bool gset = false; bool needs_grad = true; };
}  // namespace

struct wdm_trainer {
    wdm_unet_config cfg;
    int temb_ch = 0, temb_rows = 0;
    std::vector<PInfo> params;
    size_t nfloats = 0;
    float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr, *E = nullptr;
    bool use_mse = false;     // training.use_mse: differentiate the x0-space loss instead of the noise-space one
    // layers
    size_t d0w, d0b, d1w, d1b, tw, tb;
    ConvP conv_in, conv_out;
    NormP norm_out;
    std::vector<std::vector<ResP>> down_res, up_res;
    std::vector<std::vector<AttnP>> down_attn, up_attn;
    std::vector<ConvP> down_ds, up_us;
    ResP mid1, mid2;
    AttnP mid_attn;
    // per-step state
    Ctx* c = nullptr;
    std::deque<TT> acts;
    std::vector<std::function<int()>> tape;
    std::vector<std::pair<long long, long long>> tape_rng;     // [lo, hi) of the flat gradient buffer a tape entry writes ((-1, -1): none)
    // gradient buckets for an all-reduce that overlaps the backward (wdm_trainer_set_grad_events): events to record, the bounds of the last step
    std::vector<hipEvent_t> gev;
    std::vector<long long> gbounds;
    float* temb_all = nullptr;      // [B][temb_rows] forward values, and its gradient
    float* d_temb_all = nullptr;

    size_t take(const std::string& name, std::initializer_list<int64_t> shp) {
        PInfo p; p.name = name; p.ndim = (int)shp.size(); int i = 0; size_t n = 1;
        for (auto v : shp) { p.shape[i++] = v; n *= (size_t)v; }
        for (; i < 4; ++i) p.shape[i] = 0;
        p.off = nfloats; nfloats += n;
        params.push_back(p);
        return p.off;
    }
    std::vector<ConvP> conv_list;                                   // every conv of the model, in construction order (packed together at the start of a step)
    std::map<size_t, std::pair<void*, void*>> packed;              // weight offset -> (forward layout, dgrad layout) of this step
    ConvP add_conv(const std::string& n, int cin, int cout, int k) { ConvP p; p.cin = cin; p.cout = cout; p.k = k; p.w = take(n + ".weight", {cout, cin, k, k}); p.b = take(n + ".bias", {cout}); conv_list.push_back(p); return p; }
    NormP add_norm(const std::string& n, int cc) { NormP p; p.c = cc; p.g = take(n + ".weight", {cc}); p.b = take(n + ".bias", {cc}); return p; }
    std::vector<std::pair<std::string, int>> temb_list;
    ResP add_res(const std::string& n, int cin, int cout) {
        ResP r; r.cin = cin; r.cout = cout; r.has_nin = cin != cout;
        r.n1 = add_norm(n + ".norm1", cin);
        r.c1 = add_conv(n + ".conv1", cin, cout, 3);
        r.temb_row = temb_rows; temb_rows += cout; temb_list.push_back({n + ".temb_proj", cout});
        r.n2 = add_norm(n + ".norm2", cout);
        r.c2 = add_conv(n + ".conv2", cout, cout, 3);
        if (r.has_nin) r.nin = add_conv(n + ".nin_shortcut", cin, cout, 1);
        return r;
    }
    AttnP add_attn(const std::string& n, int cc) {
        AttnP a; a.c = cc; a.n = add_norm(n + ".norm", cc);
        a.q = add_conv(n + ".q", cc, cc, 1); a.k = add_conv(n + ".k", cc, cc, 1); a.v = add_conv(n + ".v", cc, cc, 1); a.proj = add_conv(n + ".proj_out", cc, cc, 1);
        return a;
    }
    void build();
    // ---- graph ops
    TT* new_act() { acts.emplace_back(); return &acts.back(); }
    int grad_buf(TT* t, bool* first) {
        if (!t->g) {
            t->g = c->ar->alloc((size_t)c->B * t->t.H * t->t.W * t->t.C * dsize(c->dtype));
            if (!t->g) WDM_FAIL(WDM_ENOMEM, "training workspace too small (gradient of a %dx%dx%d map)", t->t.H, t->t.W, t->t.C);
        }
        *first = !t->gset; t->gset = true;
        return WDM_OK;
    }
    Tens gtens(TT* t) { Tens d = t->t; d.p = t->g; d.xs = d.C; d.stats = nullptr; return d; }
    int op_conv(const ConvP& p, int mode, TT* x0, TT* x1, int temb_row, TT* res, TT** out);
    int op_gn_act(const NormP& p, TT* x0, TT* x1, int silu, TT** out);
    int op_resblock(const ResP& r, TT* x0, TT* x1, TT** out);
    int op_attn(const AttnP& a, TT* x, TT** out);
    int step(Ctx& cc, const float* x0, const float* t, const float* sa, const float* s1m, const float* e, int c_t0, float* loss, float* out_nchw);
};

void wdm_trainer::build() {
    const int ch = cfg.ch, nres = cfg.n_levels, nrb = cfg.num_res_blocks;
    temb_ch = ch * 4;
    auto is_attn = [&](int res) { for (int i = 0; i < cfg.n_attn_res; ++i) if (cfg.attn_resolutions[i] == res) return true; return false; };
    d0w = take("temb.dense.0.weight", {temb_ch, ch}); d0b = take("temb.dense.0.bias", {temb_ch});
    d1w = take("temb.dense.1.weight", {temb_ch, temb_ch}); d1b = take("temb.dense.1.bias", {temb_ch});
    conv_in = add_conv("conv_in", cfg.in_channels, ch, 3);
    int res = cfg.resolution, block_in = ch;
    down_res.resize(nres); down_attn.resize(nres); down_ds.assign(nres, ConvP{});
    up_res.resize(nres); up_attn.resize(nres); up_us.assign(nres, ConvP{});
    for (int l = 0; l < nres; ++l) {
        block_in = ch * (l == 0 ? 1 : cfg.ch_mult[l - 1]);
        const int block_out = ch * cfg.ch_mult[l];
        for (int b = 0; b < nrb; ++b) { down_res[l].push_back(add_res("down." + std::to_string(l) + ".block." + std::to_string(b), block_in, block_out)); block_in = block_out; }
        if (is_attn(res)) for (int b = 0; b < nrb; ++b) down_attn[l].push_back(add_attn("down." + std::to_string(l) + ".attn." + std::to_string(b), block_out));
        if (l != nres - 1) { down_ds[l] = add_conv("down." + std::to_string(l) + ".downsample.conv", block_in, block_in, 3); res /= 2; }
    }
    mid1 = add_res("mid.block_1", block_in, block_in);
    mid_attn = add_attn("mid.attn_1", block_in);
    mid2 = add_res("mid.block_2", block_in, block_in);
    for (int l = nres - 1; l >= 0; --l) {
        const int block_out = ch * cfg.ch_mult[l];
        int skip_in = ch * cfg.ch_mult[l];
        for (int b = 0; b <= nrb; ++b) {
            if (b == nrb) skip_in = ch * (l == 0 ? 1 : cfg.ch_mult[l - 1]);
            up_res[l].push_back(add_res("up." + std::to_string(l) + ".block." + std::to_string(b), block_in + skip_in, block_out));
            block_in = block_out;
        }
        if (is_attn(res)) for (int b = 0; b <= nrb; ++b) up_attn[l].push_back(add_attn("up." + std::to_string(l) + ".attn." + std::to_string(b), block_out));
        if (l != 0) { up_us[l] = add_conv("up." + std::to_string(l) + ".upsample.conv", block_in, block_in, 3); res *= 2; }
    }
    norm_out = add_norm("norm_out", block_in);
    conv_out = add_conv("conv_out", block_in, cfg.out_ch, 3);
    // all temb_proj layers as ONE [temb_rows][temb_ch] matrix + [temb_rows] bias (rows in block order), like the inference engine
    tw = nfloats;
    for (auto& e : temb_list) take(e.first + ".weight", {e.second, temb_ch});
    tb = nfloats;
    for (auto& e : temb_list) take(e.first + ".bias", {e.second});
}

// y = conv(x0 | x1) + bias (+ temb[b][row + co]) (+ res)
int wdm_trainer::op_conv(const ConvP& p, int mode, TT* x0, TT* x1, int temb_row, TT* res, TT** out) {
    Ctx& cx = *c;
    ConvW w; w.cin = p.cin; w.cout = p.cout; w.k = p.k; w.rows_pad = conv_rows_pad(p.cout); w.b = P + p.b;
    // forward and transposed (dgrad) layouts of every conv were written at the start of the step (pack_all): one pass over the fp32 parameters
    const auto it = packed.find(p.w);
    if (it == packed.end()) WDM_FAIL(WDM_ESTATE, "training: conv weights were not packed for this step");
    void* pk = it->second.first;
    void* wd = x0->needs_grad ? it->second.second : nullptr;
    w.w = pk;
    TT* o = new_act();
    // want_stats: the conv's epilogue also leaves the GroupNorm partial statistics of its output (most conv outputs feed a GroupNorm)
    WDM_TRY(run_conv(cx, w, mode, x0->t, x1 ? &x1->t : nullptr, nullptr, nullptr, temb_row >= 0 ? temb_all + temb_row : nullptr, temb_rows, 1, res ? &res->t : nullptr, &o->t,
                     Y_NHWC, nullptr, true));
    *out = o;
    const ConvP pp = p;
    tape_rng.push_back({(long long)std::min(pp.w, pp.b), (long long)std::max(pp.w + (size_t)pp.cout * pp.cin * pp.k * pp.k, pp.b + (size_t)pp.cout)});
    tape.push_back([this, pp, mode, x0, x1, temb_row, res, o, wd]() -> int {
        Ctx& cx = *c;
        if (!o->g) WDM_FAIL(WDM_ESTATE, "backward: conv output without gradient");
        const Tens dy = gtens(o);
        // weight, bias and (ResnetBlock conv1) per-image temb gradients: the column sums come out of the pass that transposes dy
        WDM_TRY(conv_wgrad(cx, mode, x0->t, x1 ? &x1->t : nullptr, dy, pp.cout, G + pp.w, false, G + pp.b, temb_row >= 0 ? d_temb_all + temb_row : nullptr, temb_rows));
        const long long n_out = (long long)cx.B * dy.H * dy.W * dy.C;
        if (res && res->needs_grad) { bool first; WDM_TRY(grad_buf(res, &first)); BYT(cx.dtype, l_add_into, cx.s, res->g, o->g, n_out, first ? 0 : 1); }
        if (x0->needs_grad) {
            if (!x1) {
                bool first; WDM_TRY(grad_buf(x0, &first));
                WDM_TRY(conv_dgrad(cx, mode, P + pp.w, pp.cin, pp.cout, dy, x0->t.H, x0->t.W, x0->g, !first, wd));
            } else {
                void* tmp = cx.ar->alloc((size_t)cx.B * x0->t.H * x0->t.W * pp.cin * dsize(cx.dtype));
                if (!tmp) WDM_FAIL(WDM_ENOMEM, "training workspace too small (concat dgrad)");
                WDM_TRY(conv_dgrad(cx, mode, P + pp.w, pp.cin, pp.cout, dy, x0->t.H, x0->t.W, tmp, false, wd));
                bool f0, f1; WDM_TRY(grad_buf(x0, &f0)); WDM_TRY(grad_buf(x1, &f1));
                BYT(cx.dtype, l_split_add, cx.s, tmp, x0->t.C, x1->t.C, x0->g, f0 ? 0 : 1, x1->g, f1 ? 0 : 1, (long long)cx.B * x0->t.H * x0->t.W * pp.cin);
                cx.ar->free(tmp);
            }
        }
        WDM_HIP(hipGetLastError());
        return WDM_OK;
    });
    return WDM_OK;
}

// y = act(GroupNorm([x0 | x1]))  (dense)
int wdm_trainer::op_gn_act(const NormP& p, TT* x0, TT* x1, int silu, TT** out) {
    Ctx& cx = *c;
    const int C = x0->t.C + (x1 ? x1->t.C : 0), HW = x0->t.H * x0->t.W;
    NormW nw; nw.g = P + p.g; nw.b = P + p.b; nw.c = C;
    // partial statistics: taken from the producing conv's epilogue when the tensor carries them, else one pass over the tensor
    const int ns = gn_default_nslab(HW);
    const bool own0 = x0->t.stats == nullptr, own1 = x1 && x1->t.stats == nullptr;
    float* st0 = own0 ? (float*)cx.ar->alloc(gn_stats_bytes(cx.B, ns, x0->t.C)) : x0->t.stats;
    float* st1 = !x1 ? nullptr : own1 ? (float*)cx.ar->alloc(gn_stats_bytes(cx.B, ns, x1->t.C)) : x1->t.stats;
    const int ns0 = own0 ? ns : x0->t.nslab, ns1 = !x1 ? ns : own1 ? ns : x1->t.nslab;
    float* sc = (float*)cx.ar->alloc((size_t)cx.B * C * 4);
    float* sh = (float*)cx.ar->alloc((size_t)cx.B * C * 4);
    float* mr = (float*)cx.ar->alloc((size_t)cx.B * 64 * 4);          // kept for the backward pass
    if (!st0 || (x1 && !st1) || !sc || !sh || !mr) WDM_FAIL(WDM_ENOMEM, "training workspace too small (GroupNorm)");
    TT* o = new_act();
    WDM_TRY(alloc_tens(cx, C, x0->t.H, x0->t.W, &o->t));
    if (own0) WDM_TRY(k_gn_partial(x0->t, cx.B, st0, ns, cx.dtype, cx.s));
    if (own1) WDM_TRY(k_gn_partial(x1->t, cx.B, st1, ns, cx.dtype, cx.s));
    WDM_TRY(k_gn_finalize(cx.B, HW, st0, ns0, x0->t.C, st1, ns1, x1 ? x1->t.C : 0, nw, 1e-6f, 0, sc, sh, cx.s, mr));
    WDM_TRY(k_gn_apply(x0->t, cx.B, sc, sh, C, o->t.p, C, 0, silu, cx.dtype, cx.s));
    if (x1) WDM_TRY(k_gn_apply(x1->t, cx.B, sc + x0->t.C, sh + x0->t.C, C, o->t.p, C, x0->t.C, silu, cx.dtype, cx.s));
    cx.ar->free(sh); cx.ar->free(sc);
    if (own1) cx.ar->free(st1);
    if (own0) cx.ar->free(st0);
    *out = o;
    const NormP pp = p;
    tape_rng.push_back({(long long)std::min(pp.g, pp.b), (long long)std::max(pp.g, pp.b) + C});
    tape.push_back([this, pp, x0, x1, silu, o, mr, C]() -> int {
        Ctx& cx = *c;
        if (!o->g) WDM_FAIL(WDM_ESTATE, "backward: GroupNorm output without gradient");
        NormW nw; nw.g = P + pp.g; nw.b = P + pp.b; nw.c = C;
        bool f0 = true, f1 = true;
        WDM_TRY(grad_buf(x0, &f0));
        if (x1) WDM_TRY(grad_buf(x1, &f1));
        return gn_act_backward(cx, nw, x0->t, x1 ? &x1->t : nullptr, mr, gtens(o), silu, x0->g, !f0, x1 ? x1->g : nullptr, !f1, G + pp.g, G + pp.b, false);
    });
    return WDM_OK;
}

int wdm_trainer::op_resblock(const ResP& r, TT* x0, TT* x1, TT** out) {
    TT *a1, *h1, *a2, *sc = nullptr;
    WDM_TRY(op_gn_act(r.n1, x0, x1, 1, &a1));
    WDM_TRY(op_conv(r.c1, MODE_S1, a1, nullptr, r.temb_row, nullptr, &h1));
    WDM_TRY(op_gn_act(r.n2, h1, nullptr, 1, &a2));
    if (r.has_nin) WDM_TRY(op_conv(r.nin, MODE_P1, x0, x1, -1, nullptr, &sc));
    else if (x1) WDM_FAIL(WDM_EINVAL, "resblock: identity shortcut cannot take a concat input");
    return op_conv(r.c2, MODE_S1, a2, nullptr, -1, r.has_nin ? sc : x0, out);
}

// AttnBlock (unet.py:141-193): out = x + proj(softmax(q k^T C^-1/2) v),  q,k,v = 1x1 convs of GroupNorm(x)
int wdm_trainer::op_attn(const AttnP& a, TT* x, TT** out) {
    Ctx& cx = *c;
    const int C = a.c, N = x->t.H * x->t.W, B = cx.B;
    const size_t es = dsize(cx.dtype);
    const float scale = (float)std::pow((double)C, -0.5);
    TT *hn, *q, *k, *v;
    WDM_TRY(op_gn_act(a.n, x, nullptr, 0, &hn));
    WDM_TRY(op_conv(a.q, MODE_P1, hn, nullptr, -1, nullptr, &q));
    WDM_TRY(op_conv(a.k, MODE_P1, hn, nullptr, -1, nullptr, &k));
    WDM_TRY(op_conv(a.v, MODE_P1, hn, nullptr, -1, nullptr, &v));
    // S = q k^T * scale (fp32), P = softmax(S), O = P v
    float* S = (float*)cx.ar->alloc((size_t)B * N * N * 4);
    void* Pm = cx.ar->alloc((size_t)B * N * N * es);                 // kept
    void* vT = cx.ar->alloc((size_t)B * C * N * es);
    TT* o = new_act();
    WDM_TRY(alloc_tens(cx, C, x->t.H, x->t.W, &o->t));
    if (!S || !Pm || !vT) WDM_FAIL(WDM_ENOMEM, "training workspace too small (attention)");
    auto bgemm = [&](const void* xin, int K, const void* w, int rows, long long w_img, void* y, int y_mode, float alpha) -> int {
        // y[b][i][r] = alpha * sum_k xin[b][i][k] * w[b][r][k]   (i over the N tokens)
        ConvArgs g{};
        g.x0 = xin; g.C0 = K; g.xs0 = K; g.B = B; g.Hin = g.Hout = x->t.H; g.Win = g.Wout = x->t.W; g.Cin = K; g.Cout = rows;
        g.w = w; g.w_img_stride = w_img; g.w_row_stride = K; g.w_rows = rows; g.w_bytes = (unsigned)((size_t)rows * K * es);
        g.alpha = alpha; g.y = y; g.y_mode = y_mode; g.y_s = rows;
        return launch_conv(g, MODE_P1, cx.dtype, cx.s);
    };
    WDM_TRY(bgemm(q->t.p, C, k->t.p, N, (long long)N * C, S, Y_NHWC_F32, scale));
    WDM_TRY(k_softmax_rows(S, Pm, (long long)B * N, N, cx.dtype, cx.s));
    WDM_TRY(transpose_tokens(cx, v->t.p, N, C, vT));
    WDM_TRY(bgemm(Pm, N, vT, C, (long long)C * N, o->t.p, Y_NHWC, 1.f));
    cx.ar->free(vT); cx.ar->free(S);
    tape_rng.push_back({-1, -1});
    tape.push_back([this, q, k, v, o, Pm, C, N, B, es, scale, x]() -> int {
        Ctx& cx = *c;
        if (!o->g) WDM_FAIL(WDM_ESTATE, "backward: attention output without gradient");
        auto bgemm = [&](const void* xin, int K, const void* w, int rows, long long w_img, void* y, int y_mode, float alpha) -> int {
            ConvArgs g{};
            g.x0 = xin; g.C0 = K; g.xs0 = K; g.B = B; g.Hin = g.Hout = x->t.H; g.Win = g.Wout = x->t.W; g.Cin = K; g.Cout = rows;
            g.w = w; g.w_img_stride = w_img; g.w_row_stride = K; g.w_rows = rows; g.w_bytes = (unsigned)((size_t)rows * K * es);
            g.alpha = alpha; g.y = y; g.y_mode = y_mode; g.y_s = rows;
            return launch_conv(g, MODE_P1, cx.dtype, cx.s);
        };
        float* dP = (float*)cx.ar->alloc((size_t)B * N * N * 4);
        void* dS = cx.ar->alloc((size_t)B * N * N * es);
        void* t0 = cx.ar->alloc((size_t)B * N * std::max(N, C) * es);
        void* t1 = cx.ar->alloc((size_t)B * N * std::max(N, C) * es);
        if (!dP || !dS || !t0 || !t1) WDM_FAIL(WDM_ENOMEM, "training workspace too small (attention backward)");
        bool f;
        // dP = dO v^T ;  dS = P * (dP - rowsum(dP P)) * scale
        WDM_TRY(bgemm(o->g, C, v->t.p, N, (long long)N * C, dP, Y_NHWC_F32, 1.f));
        BYT(cx.dtype, l_softmax_bwd, cx.s, Pm, dP, dS, (long long)B * N, N, scale);
        // dV[j][c] = sum_i P[i][j] dO[i][c]
        WDM_TRY(transpose_tokens(cx, Pm, N, N, t0));                 // P^T [j][i]
        WDM_TRY(transpose_tokens(cx, o->g, N, C, t1));               // dO^T [c][i]
        WDM_TRY(grad_buf(v, &f));
        WDM_TRY(bgemm(t0, N, t1, C, (long long)C * N, v->g, Y_NHWC, 1.f));
        // dQ[i][c] = sum_j dS[i][j] K[j][c]
        WDM_TRY(transpose_tokens(cx, k->t.p, N, C, t1));             // K^T [c][j]
        WDM_TRY(grad_buf(q, &f));
        WDM_TRY(bgemm(dS, N, t1, C, (long long)C * N, q->g, Y_NHWC, 1.f));
        // dK[j][c] = sum_i dS[i][j] Q[i][c]
        WDM_TRY(transpose_tokens(cx, dS, N, N, t0));                 // dS^T [j][i]
        WDM_TRY(transpose_tokens(cx, q->t.p, N, C, t1));             // Q^T [c][i]
        WDM_TRY(grad_buf(k, &f));
        WDM_TRY(bgemm(t0, N, t1, C, (long long)C * N, k->g, Y_NHWC, 1.f));
        cx.ar->free(t1); cx.ar->free(t0); cx.ar->free(dS); cx.ar->free(dP);
        WDM_HIP(hipGetLastError());
        return WDM_OK;
    });
    return op_conv(a.proj, MODE_P1, o, nullptr, -1, x, out);
}

int wdm_trainer::step(Ctx& cc, const float* x0, const float* t, const float* sa, const float* s1m, const float* e, int c_t0, float* loss, float* out_nchw) {
    c = &cc;
    acts.clear(); tape.clear(); tape_rng.clear();
    const int nres = cfg.n_levels, nrb = cfg.num_res_blocks, R = cfg.resolution, B = cc.B;
    const size_t es = dsize(cc.dtype);
    auto af = [&](size_t n) -> float* { return (float*)cc.ar->alloc(n * 4); };
    // ---- temb MLP (fp32), values kept for the backward pass
    float *emb = af((size_t)B * cfg.ch), *pre0 = af((size_t)B * temb_ch), *t0 = af((size_t)B * temb_ch), *t1 = af((size_t)B * temb_ch), *s1 = af((size_t)B * temb_ch);
    temb_all = af((size_t)B * temb_rows);
    d_temb_all = af((size_t)B * temb_rows);
    if (!emb || !pre0 || !t0 || !t1 || !s1 || !temb_all || !d_temb_all) WDM_FAIL(WDM_ENOMEM, "training workspace too small (temb)");
    WDM_TRY(k_timestep_embedding(t, B, cfg.ch, emb, cc.s));
    WDM_TRY(k_linear(emb, B, cfg.ch, P + d0w, P + d0b, temb_ch, pre0, 0, cc.s));
    hipLaunchKernelGGL(silu_f32_kernel, dim3(nbu((long long)B * temb_ch, 256)), dim3(256), 0, cc.s, pre0, t0, (long long)B * temb_ch);
    WDM_TRY(k_linear(t0, B, temb_ch, P + d1w, P + d1b, temb_ch, t1, 0, cc.s));
    hipLaunchKernelGGL(silu_f32_kernel, dim3(nbu((long long)B * temb_ch, 256)), dim3(256), 0, cc.s, t1, s1, (long long)B * temb_ch);
    WDM_TRY(k_linear(s1, B, temb_ch, P + tw, P + tb, temb_rows, temb_all, 0, cc.s));
    WDM_HIP(hipMemsetAsync(d_temb_all, 0, (size_t)B * temb_rows * 4, cc.s));
    // ---- forward and dgrad weight layouts of all convs (the parameters changed in the last optimiser step): a dozen launches for ~90 layers
    char* pack_region = nullptr;
    {
        size_t bytes = 0;
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        for (const ConvP& q : conv_list) bytes += up(conv_packed_bytes(q.cin, q.cout, q.k, cc.dtype)) + up(conv_dgrad_packed_bytes(q.cin, q.cout, q.k, cc.dtype));
        pack_region = (char*)cc.ar->alloc(bytes);
        if (!pack_region) WDM_FAIL(WDM_ENOMEM, "training workspace too small (packed weights)");
        packed.clear();
        std::vector<PackDesc> d3, d1;
        size_t off = 0;
        for (const ConvP& q : conv_list) {
            PackDesc d{};
            d.w = P + q.w; d.cout = q.cout; d.cin = q.cin; d.rows_total = conv_rows_pad(q.cout);
            d.dstf = pack_region + off; off += up(conv_packed_bytes(q.cin, q.cout, q.k, cc.dtype));
            d.dstd = pack_region + off; off += up(conv_dgrad_packed_bytes(q.cin, q.cout, q.k, cc.dtype));
            packed[q.w] = {d.dstf, d.dstd};
            (q.k == 3 ? d3 : d1).push_back(d);
        }
        if (!cc.dry) {
            WDM_TRY(k_pack_conv_both_batch(d3.data(), (int)d3.size(), 3, cc.dtype, cc.s));
            WDM_TRY(k_pack_conv_both_batch(d1.data(), (int)d1.size(), 1, cc.dtype, cc.s));
        }
    }
    // ---- network input: [x_cond | x_t | x_other] with x_t = sqrt(a) x_tar + sqrt(1-a) e
    TT* xin = new_act();
    xin->needs_grad = false;
    WDM_TRY(alloc_tens(cc, cfg.in_channels, R, R, &xin->t));
    const int pc = cfg.out_ch;
    BYT(cc.dtype, l_build_input, cc.s, x0, e, sa, s1m, cfg.in_channels, R * R, c_t0, pc, xin->t.p, (long long)B * R * R * cfg.in_channels);
    // ---- forward
    std::vector<TT*> hs;
    TT* h;
    WDM_TRY(op_conv(conv_in, MODE_S1, xin, nullptr, -1, nullptr, &h));
    hs.push_back(h);
    for (int l = 0; l < nres; ++l) {
        for (int b = 0; b < nrb; ++b) {
            TT* o;
            WDM_TRY(op_resblock(down_res[l][b], hs.back(), nullptr, &o));
            if (!down_attn[l].empty()) { TT* o2; WDM_TRY(op_attn(down_attn[l][b], o, &o2)); o = o2; }
            hs.push_back(o);
        }
        if (l != nres - 1) { TT* o; WDM_TRY(op_conv(down_ds[l], MODE_S2, hs.back(), nullptr, -1, nullptr, &o)); hs.push_back(o); }
    }
    TT *m1, *m2;
    WDM_TRY(op_resblock(mid1, hs.back(), nullptr, &m1));
    WDM_TRY(op_attn(mid_attn, m1, &m2));
    WDM_TRY(op_resblock(mid2, m2, nullptr, &h));
    for (int l = nres - 1; l >= 0; --l) {
        for (int b = 0; b <= nrb; ++b) {
            TT* skip = hs.back(); hs.pop_back();
            TT* o;
            WDM_TRY(op_resblock(up_res[l][b], h, skip, &o));
            h = o;
            if (!up_attn[l].empty()) { TT* o2; WDM_TRY(op_attn(up_attn[l][b], h, &o2)); h = o2; }
        }
        if (l != 0) { TT* o; WDM_TRY(op_conv(up_us[l], MODE_UPS, h, nullptr, -1, nullptr, &o)); h = o; }
    }
    TT* an;
    WDM_TRY(op_gn_act(norm_out, h, nullptr, 1, &an));
    // conv_out in fp32 NHWC for the loss
    float* outf = af((size_t)B * R * R * pc);
    if (!outf) WDM_FAIL(WDM_ENOMEM, "training workspace too small (output)");
    {
        ConvW w; w.cin = conv_out.cin; w.cout = pc; w.k = 3; w.rows_pad = conv_rows_pad(pc); w.b = P + conv_out.b;
        w.w = packed.at(conv_out.w).first;
        Tens dummy;
        WDM_TRY(run_conv(cc, w, MODE_S1, an->t, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &dummy, Y_NHWC_F32, outf));
    }
    // ---- loss and its gradient
    void* dout = cc.ar->alloc((size_t)B * R * R * pc * es);
    if (!dout) WDM_FAIL(WDM_ENOMEM, "training workspace too small (loss gradient)");
    {
        double* lpart = (double*)cc.ar->alloc(LOSS_WGS * sizeof(double));
        if (!lpart) WDM_FAIL(WDM_ENOMEM, "training workspace too small (loss partials)");
        if (!cc.dry) BYT(cc.dtype, l_loss, cc.s, outf, e, B, pc, R * R, dout, loss, out_nchw, use_mse ? sa : nullptr, use_mse ? s1m : nullptr, lpart);
        cc.ar->free(lpart);
    }
    // ---- backward: conv_out by hand, then the tape in reverse
    {
        Tens dy; dy.p = dout; dy.C = pc; dy.H = R; dy.W = R; dy.xs = pc;
        WDM_TRY(colsum(cc, dy, G + conv_out.b, false, false));
        WDM_TRY(conv_wgrad(cc, MODE_S1, an->t, nullptr, dy, pc, G + conv_out.w, false));
        bool f; WDM_TRY(grad_buf(an, &f));
        WDM_TRY(conv_dgrad(cc, MODE_S1, P + conv_out.w, conv_out.cin, pc, dy, R, R, an->g, false, packed.at(conv_out.w).second));
    }
    // The tape runs in reverse; the parameters sit in the flat buffers in forward order, so the gradient buffer fills from its END.  With events set
    // (wdm_trainer_set_grad_events) the range finished so far is cut into buckets and an event is recorded behind each: the caller's all-reduce of a bucket
    // can start while the rest of the backward still runs (the reference's DistributedDataParallel does the same with 25 MB buckets, ddm_wavelet.py:168).
    // A cut at entry i is valid when no earlier entry writes at or above the running minimum.
    gbounds.clear();
    {
        std::vector<long long> prefix_hi(tape.size() + 1, 0);
        for (size_t i = 0; i < tape.size(); ++i) prefix_hi[i + 1] = std::max(prefix_hi[i], tape_rng[i].second);
        long long run_lo = (long long)std::min(conv_out.w, conv_out.b);
        const long long body_hi = (long long)std::max(conv_out.w + (size_t)pc * conv_out.cin * 9, conv_out.b + (size_t)pc);
        long long body_lo = run_lo;
        for (size_t i = 0; i < tape.size(); ++i) if (tape_rng[i].first >= 0) body_lo = std::min(body_lo, tape_rng[i].first);
        const size_t nev = gev.size();
        const long long target = nev ? (body_hi - body_lo + (long long)nev - 1) / (long long)nev : 0;
        long long prev = body_hi;
        if (nev) gbounds.push_back(body_hi);
        for (size_t i = tape.size(); i-- > 0;) {
            WDM_TRY(tape[i]());
            if (tape_rng[i].first >= 0) run_lo = std::min(run_lo, tape_rng[i].first);
            if (nev && gbounds.size() < nev && i > 0 && prefix_hi[i] <= run_lo && prev - run_lo >= target) {
                WDM_HIP(hipEventRecord(gev[gbounds.size() - 1], cc.s));
                gbounds.push_back(run_lo);
                prev = run_lo;
            }
        }
        if (nev && prev > run_lo) {
            WDM_HIP(hipEventRecord(gev[gbounds.size() - 1], cc.s));
            gbounds.push_back(run_lo);
        }
    }
    // ---- temb MLP backward
    {
        const long long n4 = (long long)B * temb_ch;
        float *d_s1 = af(n4), *d_t1 = af(n4), *d_t0 = af(n4), *d_pre0 = af(n4), *csc = af((size_t)4 * temb_rows), *xpart = af((size_t)64 * n4);
        if (!d_s1 || !d_t1 || !d_t0 || !d_pre0 || !csc || !xpart) WDM_FAIL(WDM_ENOMEM, "training workspace too small (temb backward)");
        hipLaunchKernelGGL(lin_bwd_w_kernel, dim3(nbu((long long)temb_rows * temb_ch, 256)), dim3(256), 0, cc.s, d_temb_all, s1, B, temb_rows, temb_ch, G + tw);
        l_colsum_f32(cc.s, d_temb_all, temb_rows, B, G + tb, csc);
        hipLaunchKernelGGL(lin_bwd_x_part_kernel, dim3(nbu(n4, 256), 64), dim3(256), 0, cc.s, d_temb_all, P + tw, B, temb_rows, temb_ch, xpart);
        hipLaunchKernelGGL(lin_bwd_x_final_kernel, dim3(nbu(n4, 256)), dim3(256), 0, cc.s, xpart, n4, d_s1);
        hipLaunchKernelGGL(silu_bwd_f32_kernel, dim3(nbu(n4, 256)), dim3(256), 0, cc.s, t1, d_s1, d_t1, n4);
        hipLaunchKernelGGL(lin_bwd_w_kernel, dim3(nbu((long long)temb_ch * temb_ch, 256)), dim3(256), 0, cc.s, d_t1, t0, B, temb_ch, temb_ch, G + d1w);
        l_colsum_f32(cc.s, d_t1, temb_ch, B, G + d1b, csc);
        hipLaunchKernelGGL(lin_bwd_x_part_kernel, dim3(nbu(n4, 256), 64), dim3(256), 0, cc.s, d_t1, P + d1w, B, temb_ch, temb_ch, xpart);
        hipLaunchKernelGGL(lin_bwd_x_final_kernel, dim3(nbu(n4, 256)), dim3(256), 0, cc.s, xpart, n4, d_t0);
        hipLaunchKernelGGL(silu_bwd_f32_kernel, dim3(nbu(n4, 256)), dim3(256), 0, cc.s, pre0, d_t0, d_pre0, n4);
        hipLaunchKernelGGL(lin_bwd_w_kernel, dim3(nbu((long long)temb_ch * cfg.ch, 256)), dim3(256), 0, cc.s, d_pre0, emb, B, temb_ch, cfg.ch, G + d0w);
        l_colsum_f32(cc.s, d_pre0, temb_ch, B, G + d0b, csc);
        WDM_HIP(hipGetLastError());
    }
    acts.clear(); tape.clear(); tape_rng.clear();
    return WDM_OK;
}

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

int wdm_trainer_create(wdm_handle* h, const wdm_unet_config* cfg, wdm_trainer** out) {
    if (!cfg || !out) WDM_FAIL(WDM_EINVAL, "wdm_trainer_create: null argument");
    if (cfg->dtype != WDM_BF16 && cfg->dtype != WDM_F32) WDM_FAIL(WDM_EINVAL, "wdm_trainer_create: bad dtype");
    if (cfg->ch % 32 || cfg->in_channels % 32 || cfg->in_channels < 32)
        WDM_FAIL(WDM_EINVAL, "wdm_trainer_create: ch and in_channels must be multiples of 32 (the training step covers the [x_cond | x_t | x_other] input of raindrop_wavelet.yml)");
    wdm_trainer* t = new wdm_trainer();
    t->cfg = *cfg;
    t->build();
    *out = t;
    (void)h;
    return WDM_OK;
}
int wdm_trainer_destroy(wdm_trainer* t) { delete t; return WDM_OK; }
int wdm_trainer_num_params(const wdm_trainer* t) { return t ? (int)t->params.size() : 0; }
int64_t wdm_trainer_num_floats(const wdm_trainer* t) { return t ? (int64_t)t->nfloats : 0; }
int wdm_trainer_param_info(const wdm_trainer* t, int i, const char** name, int* ndim, int64_t shape[4], int64_t* offset) {
    if (!t || i < 0 || i >= (int)t->params.size()) WDM_FAIL(WDM_EINVAL, "wdm_trainer_param_info: index out of range");
    const PInfo& p = t->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = p.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    if (offset) *offset = (int64_t)p.off;
    return WDM_OK;
}
int wdm_trainer_set_buffers(wdm_trainer* t, float* params, float* grads, float* m, float* v, float* ema) {
    if (!t || !params || !grads) WDM_FAIL(WDM_EINVAL, "wdm_trainer_set_buffers: params and grads are required");
    t->P = params; t->G = grads; t->M = m; t->V = v; t->E = ema;
    return WDM_OK;
}
int wdm_trainer_set_objective(wdm_trainer* t, int use_mse) {
    if (!t) WDM_FAIL(WDM_EINVAL, "wdm_trainer_set_objective: null trainer");
    t->use_mse = use_mse != 0;
    return WDM_OK;
}
// Gradient buckets (include/wavedm.h): events the next steps record as the flat gradient buffer fills from its end; the bounds of the last step.
int wdm_trainer_set_grad_events(wdm_trainer* t, void* const* events, int n) {
    if (!t || n < 0 || n > 64 || (n > 0 && !events)) WDM_FAIL(WDM_EINVAL, "wdm_trainer_set_grad_events: bad argument");
    t->gev.clear();
    for (int i = 0; i < n; ++i) {
        if (!events[i]) WDM_FAIL(WDM_EINVAL, "wdm_trainer_set_grad_events: null event %d", i);
        t->gev.push_back((hipEvent_t)events[i]);
    }
    t->gbounds.clear();
    return WDM_OK;
}
int wdm_trainer_grad_buckets(const wdm_trainer* t, int64_t* bounds, int max_bounds, int* n_buckets) {
    if (!t || !bounds || !n_buckets) WDM_FAIL(WDM_EINVAL, "wdm_trainer_grad_buckets: null argument");
    const int nb = t->gbounds.empty() ? 0 : (int)t->gbounds.size() - 1;
    if ((int)t->gbounds.size() > max_bounds) WDM_FAIL(WDM_EINVAL, "wdm_trainer_grad_buckets: %d bounds do not fit %d", (int)t->gbounds.size(), max_bounds);
    for (size_t i = 0; i < t->gbounds.size(); ++i) bounds[i] = (int64_t)t->gbounds[i];
    *n_buckets = nb;
    return WDM_OK;
}
int wdm_trainer_step(wdm_trainer* t, const float* x0, const float* tt, const float* sqrt_a, const float* sqrt_1ma, const float* e, int B, int c_t0, float* loss,
                     float* out_nchw, void* workspace, size_t workspace_bytes, void* stream) {
    if (!t || !x0 || !tt || !sqrt_a || !sqrt_1ma || !e || !loss || !workspace) WDM_FAIL(WDM_EINVAL, "wdm_trainer_step: null argument");
    if (!t->P || !t->G) WDM_FAIL(WDM_ESTATE, "wdm_trainer_step: call wdm_trainer_set_buffers first");
    if (((uintptr_t)workspace) & 255) WDM_FAIL(WDM_EINVAL, "wdm_trainer_step: workspace must be 256-byte aligned");
    Arena ar(workspace, workspace_bytes);
    Ctx c{(hipStream_t)stream, t->cfg.dtype, B, &ar, false};
    return t->step(c, x0, tt, sqrt_a, sqrt_1ma, e, c_t0, loss, out_nchw);
}
int wdm_trainer_adam_ema(wdm_trainer* t, int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float ema_mu, void* stream) {
    if (!t || !t->P || !t->G || !t->M || !t->V) WDM_FAIL(WDM_ESTATE, "wdm_trainer_adam_ema: buffers not set");
    if (step < 1) WDM_FAIL(WDM_EINVAL, "wdm_trainer_adam_ema: step counts from 1");
    const float bc1 = 1.0f - (float)std::pow((double)beta1, (double)step), bc2 = 1.0f - (float)std::pow((double)beta2, (double)step);
    const long long n = (long long)t->nfloats;
    hipLaunchKernelGGL(adam_ema_kernel, dim3(nb(n, 256)), dim3(256), 0, (hipStream_t)stream, t->P, t->G, t->M, t->V, t->E, n, lr, beta1, beta2, eps, weight_decay, bc1,
                       std::sqrt(bc2), ema_mu);
    WDM_HIP(hipGetLastError());
    return WDM_OK;
}

}  // extern "C"
