// Internal declarations shared by the translation units of libwavedm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/wavedm.h"
#include "conv_kernel.h"

namespace wdm {

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define WDM_FAIL(code, ...)        \
    do {                           \
        wdm::set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)
#define WDM_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) WDM_FAIL(WDM_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)
#define WDM_TRY(expr)              \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != WDM_OK) return rc__; \
    } while (0)

inline bool is_h16(int dtype) { return dtype == WDM_BF16 || dtype == WDM_F16; }      // the 16-bit modes: same kernels, layouts and eligibility rules
inline size_t dsize(int dtype) { return is_h16(dtype) ? 2 : 4; }
inline bool dtype_valid(int dtype) { return dtype == WDM_F32 || dtype == WDM_BF16 || dtype == WDM_F32X3 || dtype == WDM_F16; }
// launch a kernel template on the 16-bit element type of `dtype` (H16 = __bf16 or f16_t inside the statement)
#define WDM_H16_SWITCH(dtype, ...)                                          \
    do {                                                                    \
        if ((dtype) == WDM_F16) { using H16 = wdm::f16_t; __VA_ARGS__; }      \
        else { using H16 = __bf16; __VA_ARGS__; }                                 \
    } while (0)
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- deterministic first-fit arena over a caller-provided buffer -----------------------------
// In "dry" mode (base == nullptr) it only tracks the high-water mark; the real run repeats the same
// alloc/free sequence, so it needs exactly that many bytes.
class Arena {
   public:
    Arena(void* base, size_t cap) : base_((char*)base), cap_(cap) { free_.push_back({0, cap_}); }
    static Arena dry() { return Arena(nullptr, (size_t)1 << 60); }
    void* alloc(size_t bytes);
    void free(void* p);
    size_t peak() const { return peak_; }
    bool failed() const { return failed_; }
    bool is_dry() const { return base_ == nullptr; }

   private:
    struct Blk { size_t off, len; };
    char* base_;
    size_t cap_;
    size_t peak_ = 0;
    bool failed_ = false;
    std::vector<Blk> free_;   // sorted by offset
    std::vector<Blk> used_;
};

// ---- tensors inside the executor: NHWC, model dtype, batch implicit ---------------------------
struct Tens {
    void* p = nullptr;
    int C = 0, H = 0, W = 0;
    int xs = 0;   // pixel stride in elements
    float* stats = nullptr;   // GroupNorm partial statistics float4[B][nslab][C] written by the producing conv, or nullptr
    int nslab = 0;
    float* gst = nullptr;     // group-level partials float[B][nslab][32][3] behind them in the same allocation (C = 128 / 256 / 512), or nullptr: gn_inline.h
    // act(GroupNorm(this tensor)) already written by the producing conv (gn_group.h: gn_out_tail) for the norm whose gamma is nrm_for: materialize_gn
    // takes it instead of launching gn_finalize_apply
    void* nrm = nullptr;
    const float* nrm_for = nullptr;
    int nrm_silu = 0;
    // the scale / shift rows of the norm whose gamma is fin_for over [this tensor | the tensor at fin_other], already finalised by the producing kernel's last
    // workgroups (gn_arrive.h): run_gn hands them to the consumer instead of launching gn_finalize
    float* fin_scale = nullptr;
    float* fin_shift = nullptr;
    const float* fin_for = nullptr;
    const void* fin_other = nullptr;
    int fin_silu = 0;
};

struct Ctx {
    hipStream_t s;
    int dtype;
    int B;
    Arena* ar;
    bool dry;     // no launches, only arena accounting
    // per-image arrival counters of the launches that finalise their consumer's GroupNorm (gn_arrive.h): fin_cap slots of B ints, zeroed at the start of a forward
    // call (wdm_unet::forward); nullptr: nobody asks (the block-level entry points)
    int* fin_cnt = nullptr;
    int fin_cap = 0, fin_used = 0;
};

// ---- packed parameter views (device pointers) -------------------------------------------------
struct ConvW {
    const void* w = nullptr;   // [taps][rows_pad][cin] model dtype
    const void* w_up4 = nullptr;   // Upsample convs, bf16: the 16 pre-summed sub-pixel taps (k_pack_up4), or nullptr
    const void* w_sm = nullptr;    // bf16 3x3 convs: slab-major copy [cin / 32][tap][rows_pad][32] (k_pack_conv_sm), or nullptr; the folded AttnBlock's qf / pf: [cin / 32][rows_pad][32]
    const float* b = nullptr;  // [cout]
    int cin = 0, cout = 0, k = 1, rows_pad = 0;
};
struct NormW {
    const float* g = nullptr;
    const float* b = nullptr;
    int c = 0;
};
// what consumes a conv's output next: the norm (over [output | other], other = the tensor the consumer concatenates behind it, or nullptr) whose scale / shift the
// producer may as well finalise itself (run_conv: fin); silu: the consumer is a conv with the GroupNorm+SiLU prologue
// Largest map (pixels) whose ResnetBlocks normalise in a pass: the 8 x 8 level.  (Measured: 16 x 16 neutral, 32 x 32 and up slower -- the pass is HBM-bound there.)
constexpr int GN_PASS_MAX_HW = 64;
struct FinReq {
    const NormW* n;
    const Tens* other;
    int silu;
};
struct ResW {
    int cin = 0, cout = 0;
    NormW n1, n2;
    ConvW c1, c2, nin;
    bool has_nin = false;
    const float* temb = nullptr;   // [n_t][temb_ld] projected temb rows for this block (already + bias)
    int temb_ld = 0;
    int temb_per_image = 0;
};
struct AttnW {
    int c = 0;
    NormW n;
    ConvW qk;     // fused: rows [0,C) = q, rows [C,2C) = k
    ConvW v, proj;
    // 16-bit modes: the folded operands (k_attn_fold): qf = Wk^T Wq with bias Wk^T bq, pf = Wp Wv with bias Wp bv + bp; w == nullptr: not available
    ConvW qf, pf;
};

inline int conv_rows_pad(int cout) { return cout <= 16 ? 16 : (int)align_up((size_t)cout, 64); }
inline size_t conv_packed_bytes(int cin, int cout, int k, int dtype) {
    return (size_t)k * k * conv_rows_pad(cout) * cin * dsize(dtype);
}

// ---- kernels (elementwise.hip) ---------------------------------------------------------------
int k_dwt_fwd(const float* x, float* y, int B, int H, int W, hipStream_t s, float scale = 1.0f, float shift = 0.0f);      // DWT(scale * x + shift)
int k_dwt_inv(const float* y, float* x, int B, int h, int w, hipStream_t s, const float* y_lo = nullptr, int lo_total = 0, int n_lo = 0, int post = 0);
int k_pack_channels(const float* src, int nch, int H, int W, const int32_t* patches, int n, int p, void* x96,
                    int c_total, int c_off, int dtype, hipStream_t s);
int k_ddim_update(const float* eps, const int32_t* patches, int n, int p, const float* x_t, int nimg, int H, int W,
                  float s1m, float sa, float san, float c2, float* x0, float* xn, hipStream_t s, const float* noise = nullptr, float c1 = 0.f);
int k_patch_accumulate(const float* eps, const int32_t* patches, int n, int p, int nimg, int H, int W, float* acc_cnt, hipStream_t s);
int k_ddim_from_sums(const float* acc_cnt, const float* x_t, int nimg, int H, int W, float s1m, float sa, float san, float c2, float* x0,
                     float* xn, hipStream_t s);
int k_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int dtype, hipStream_t s);
int k_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int dtype, hipStream_t s);
// GroupNorm(32, eps): partial statistics float4[B][nslab][C] = (pivot, sum(x-K), sum((x-K)^2), n) and their finalisation
// over the channel concat of up to two tensors -> per-(image, channel) scale/shift.
// for_silu_conv != 0: scale/shift are pre-multiplied by -log2(e) for the conv prologue (conv_kernel.h: gn_silu_unit)
int gn_default_nslab(int HW);
size_t gn_stats_bytes(int B, int nslab, int C);
int k_gn_partial(const Tens& x, int B, float* stats, int nslab, int dtype, hipStream_t s);
int k_gn_finalize(int B, int HW, const float* st0, int nslab0, int C0, const float* st1, int nslab1, int C1, const NormW& nw,
                  float eps, int for_silu_conv, float* scale, float* shift, hipStream_t s, float* mean_rstd = nullptr);   // mean_rstd: [B][32][2], optional
// finalize + apply (+ SiLU, + concat) in one launch: y dense [B][HW][C0 + C1]; st0 / st1 = the tensors' partial statistics
bool gn_fused_pass_eligible(int C0, int C1, int dtype);
int k_gn_finalize_apply(int B, const Tens& x0, const Tens* x1, const float* st0, int nslab0, const float* st1, int nslab1, const NormW& nw, float eps, int silu, void* y,
                        int dtype, hipStream_t s);
// y[b][p][y_choff + c] = act(x*scale + shift); scale/shift rows are sc_ld long (channel concat: pass scale + C0), silu != 0 applies SiLU
int k_gn_apply(const Tens& x, int B, const float* scale, const float* shift, int sc_ld, void* y, int y_stride, int y_choff, int silu, int dtype,
               hipStream_t s);
int k_softmax_rows(const float* S, void* P, long long rows, int n, int dtype, hipStream_t s);
int k_timestep_embedding(const float* t, int n_t, int dim, float* emb, hipStream_t s);
// out[n][o] = post( W[o][:] . pre(in[n][:]) + b[o] );  act: 0 none, 1 SiLU on input, 2 SiLU on output
int k_linear(const float* in, int n, int k, const float* W, const float* b, int o, float* out, int act, hipStream_t s);
// weight packing: OIHW f32 -> [tap][rows_pad][cin] dtype, written at row offset `row_off` of a `rows_total` matrix
// rows [row_off + cout, rows_total) are zero-filled when zero_tail != 0
// cin_dst > cin: destination rows are cin_dst long, the extra columns zero (0 = cin)
int k_pack_conv(const float* w_oihw, int cout, int cin, int k, void* dst, int rows_total, int row_off, int zero_tail,
                int dtype, hipStream_t s, int cin_dst = 0);
// slab-major copy for the LDS-DMA kernels: OIHW f32 3x3 -> [cin / 32][tap][rows_total][32] bf16 (rows >= cout zero): a weight sub-stage's rows are
// 64 B each, and in the plain matrix a 1 KB DMA piece touches 16 half cache lines (27.6 cycles of the CU's vector-memory path, tools/dma_ubench.hip);
// here it is one contiguous run of 8 whole lines (15 cycles)
// (f32x3 mode: the same slot holds the pre-split copy of conv_dmax3_kernel.h, plain [tap][row][cin] order, 16-channel groups as [hi | hi | lo | lo])
inline bool conv_sm_eligible(int dtype, int k, int cin) { return (is_h16(dtype) && k == 3 && cin % 32 == 0) || (dtype == WDM_F32X3 && k == 3 && cin % 16 == 0); }
int k_pack_conv_sm(const float* w_oihw, int cout, int cin, void* dst, int rows_total, hipStream_t s, int dtype = WDM_BF16, int k = 3);      // k = 1: [cin / 32][rows][32] of a 1x1 matrix
int k_pack_up4(const float* w_oihw, int cout, int cin, void* dst, int rows_total, hipStream_t s, int dtype = WDM_BF16);
bool conv_up4_eligible(int dtype, int H, int W, int cin, int cout);      // sub-pixel Upsample kernel applies to this low-resolution map
int k_pad_channels(const void* x, int C, int Cp, void* y, long long rows, int dtype, hipStream_t s);
int k_copy_f32(const float* src, float* dst, long long n, hipStream_t s);
// *flag_dev (device int, zeroed by the caller) = 1 if any |x[i]| > limit or x[i] is not finite: the fp16 range check of the weight loader
int k_flag_out_of_range(const float* x, long long n, float limit, int* flag_dev, hipStream_t s);

// ---- live kernel timing (prof.hip): HIP events on the launch stream around every conv launch ------
int concurrent_streams();      // prof.hip: wdm_set_concurrent_streams
bool prof_enabled();
void prof_begin(hipStream_t s, const char* kernel, double flops, double bytes);
void prof_end(hipStream_t s);

// shapes whose 3x3 conv can also accumulate a 1x1 shortcut over [sC0 | sC1] channels (conv_dma_kernel.h; bf16 only)
inline bool conv_can_fuse_shortcut(int H, int W, int cin, int cout, int sC0, int sC1) {
    const bool t16 = H % 16 == 0 && W % 16 == 0 && cout >= 128 && cin <= 2048;      // conv_dma_kernel.h
    const bool t8 = H == 8 && W == 8;                                                // conv_dma8_kernel.h (convs without the GroupNorm prologue)
    return (t16 || t8) && cin % 32 == 0 && sC0 % 64 == 0 && (sC0 + sC1) % 64 == 0;
}

// ---- fused attention core (attn.hip / attn_fused_kernel.h): q, k [B][256][ld] token-major, v = V^T [B][C][256] or (v_tok) V [B][256][v_ld] -> o [B][256][C], 16-bit
bool attn_fused_eligible(int dtype, int N, int C);
struct AttnOperands {
    const void* q = nullptr; const void* k = nullptr; const void* v = nullptr;
    int q_ld = 0, k_ld = 0, v_ld = 0;      // elements per token row (v_ld: token-major V only)
    int v_tok = 0;
    // folded AttnBlock with proj_out fused (C <= 512): the query projection inside the kernel (attn_fused_kernel.h: QPROJ) -- q is then null and qw / qbias are the
    // folded [C][qw_ld] 16-bit matrix Wk^T Wq and its fp32 bias Wk^T bq
    const void* qw = nullptr; const float* qbias = nullptr; int qw_ld = 0; size_t qw_bytes = 0;
    int bdiag = 0;       // 1: N = 64 tokens per image (8 x 8 maps): four images share a 256-row "image" of the kernel, attention stays inside each (attn_fused_kernel.h)
    int qw_slab = 0;     // 0: qw is the plain [C][qw_ld] matrix; else it is the slab-major copy [C / 32][rows][32] and this the elements between slabs (rows x 32)
};
// vbias != nullptr: V was computed without the v bias, which is added to the output instead
// proj != nullptr (C <= 512): proj_out fused in as a third phase; *proj = the 1x1 conv's arguments as run_conv builds them (weights, bias, residual, output,
// statistics); o is then unused
int launch_attn_fused(const AttnOperands& in, void* o, int B, int C, hipStream_t s, const float* vbias = nullptr, const ConvArgs* proj = nullptr, int dtype = WDM_BF16);
// AttnBlock operand folding (elementwise.hip), fp32 in / out, fp64 sums: M = Wk^T Wq, cq = Wk^T bq (scores: (Wq h_i + bq).(Wk h_j + bk) = (M h_i + cq).h_j + terms constant
// in j, which the softmax cancels); Wvp = Wp Wv, bvp = Wp bv + bp (proj_out(P.(Wv h + bv)) = Wvp (P.h) + bvp: the rows of P sum to one)        (models/unet.py:176-191)
int k_attn_fold(const float* wq, const float* bq, const float* wk, const float* wv, const float* bv, const float* wp, const float* bp, int C, float* M, float* cq, float* Wvp,
                float* bvp, hipStream_t s);

// ---- experiment switches (environment), read ONCE -- at first use or when wdm_env_refresh() is called (tests and A/B harnesses that change the
// environment inside a running process call it); no launch path calls getenv.  Defaults are the measured best.
struct EnvCfg {
    int conv_dma = 1;     // WDM_CONV_DMA=0: no LDS-DMA 3x3 kernel of any mode or dtype (stride-1, 8 x 8, Downsample, sub-pixel Upsample, the f32x3 family): everything on the
                          // register-staged conv_kernel.h -- the cross-check of the whole kernel family; fused shortcuts, in-prologue and in-tile GroupNorm go with it
    int gemm = 1;         // WDM_GEMM=0: 1x1 convs / batched GEMMs on the register-staged kernel (bf16 and f32x3)
    int bn256 = 1;        // WDM_BN256=0|1|2: 256-column tiles (3x3, sub-pixel upsample, 1x1 GEMM) never / where the grid still fills the chip / wherever the shape allows (same bits)
    int gn_tile = 2;      // WDM_GN_TILE=0: gn_finalize_apply launches instead of the in-tile GroupNorm of the producing conv's output (gn_group.h); 1: only for the consumers
                          // that normalise in a pass anyway; 2: also conv1 -> norm2 of the ResnetBlocks on 16 x 16 maps (conv2 then runs without its prologue)
    int gn_inline = 1;    // WDM_GN_INLINE=0: a gn_finalize launch for every conv with the GroupNorm prologue; 1 (default): finalised in the consumer's own prologue where the producer
                          // left group partials (maps up to 32 x 32, single input: gn_inline.h); 2: everything else finalised by the PRODUCER's last workgroups (gn_arrive.h: built
                          // in round 4, bit-identical to the launches, 3.6 % SLOWER end to end -- one workgroup per image reduces what 2 048 waves of gn_finalize do side by side)
    int attn_fold = 1;    // WDM_ATTN_FOLD=0: the AttnBlock keeps its k and v projections (16-bit modes; blocks.hip: run_attn); 1: folded into q and proj_out at load time
    int attn_fused = 3;   // WDM_ATTN_FUSED=0: attention core as three launches (Q.K^T, softmax, P.V); 1: fused core, proj_out as its own GEMM; 2: proj_out fused in as well;
                          // 3 (default): the folded block's query projection too (C = 128 ... 512 on 16 x 16 maps): the AttnBlock behind its GroupNorm is one launch
    int attn_sm = 1;      // WDM_ATTN_SM=0: the fused attention core streams Wk^T Wq / Wp Wv from the plain [row][cin] matrices (64-byte half lines per row) instead of their slab-major copies
    int up4 = 1;          // WDM_UP4=0: 9-tap Upsample conv everywhere (no sub-pixel form)
    int wgrad_bg = 0;     // WDM_WGRAD_BG=<n>: the batched-GEMM form of the weight gradient everywhere, n images per group (0: direct kernel for 3x3 stride-1 layers, training)
};
const EnvCfg& env_cfg();
void env_cfg_refresh();

// ---- conv dispatch (conv_bf16.hip / conv_f32.hip) ---------------------------------------------
int launch_conv(const ConvArgs& a, int mode, int dtype, hipStream_t s);

// ---- blocks (blocks.hip) ----------------------------------------------------------------------
// want_stats: also emit the GroupNorm partial statistics of the output (out->stats) from the conv epilogue
// gn_inl: GroupNorm(+SiLU) of the single input x0 finalised in the conv's own prologue from x0.gst (gn_inline.h); scale / shift are then null
int run_conv(Ctx& c, const ConvW& w, int mode, const Tens& x0, const Tens* x1, const float* scale, const float* shift,
             const float* temb, int temb_ld, int temb_per_image, const Tens* res, Tens* out, int y_mode, void* y_ext,
             bool want_stats = false, const ConvW* shortcut = nullptr, const Tens* sx0 = nullptr, const Tens* sx1 = nullptr, const NormW* gn_inl = nullptr,
             ConvArgs* defer = nullptr,       // defer: fill *defer instead of launching (the caller hands it to another launcher: attn.hip)
             const NormW* on = nullptr, int on_silu = 0,       // on: the consumer's norm -- out->nrm = act(GroupNorm(out)) from the conv itself where its kernel can
             const FinReq* fin = nullptr);                     // fin: the consumer's norm -- its scale / shift rows from this conv's own launch where its kernel can (gn_arrive.h)
int run_gn(Ctx& c, const NormW& nw, const Tens& x0, const Tens* x1, int for_silu_conv, float** scale, float** shift);
// next_n: the norm of the consumer of *out when that consumer normalises in a pass of its own (an AttnBlock, an 8 x 8 ResnetBlock): conv2 writes it (run_conv: on)
// next_fin: the consumer of *out when it is a conv with the GroupNorm prologue on a map too large for the pass / the in-prologue finalize (see run_conv: fin)
int run_resblock(Ctx& c, const ResW& w, const Tens& x0, const Tens* x1, Tens* out, const NormW* next_n = nullptr, int next_silu = 0, const FinReq* next_fin = nullptr);
int run_attn(Ctx& c, const AttnW& w, const Tens& x, Tens* out, const FinReq* next_fin = nullptr);
// whether the consumer (Cin channels in all, H x W map, single: no concat) takes its GroupNorm from a producer-side finalize rather than from the pass or the in-prologue finalize
bool wants_fin(const Ctx& c, int Cin, int H, int W, bool single);
int alloc_tens(Ctx& c, int C, int H, int W, Tens* t);
void free_tens(Ctx& c, Tens& t);

// ---- training step: backward primitives (train.hip) ---------------------------------------------------------------
// wd_prepacked: the transposed / mirrored weights already packed by k_pack_conv_both (else packed here from w_oihw)
int conv_dgrad(Ctx& c, int mode, const float* w_oihw, int cin, int cout, const Tens& dy, int H, int W, void* dx, bool accumulate, const void* wd_prepacked = nullptr);
size_t conv_dgrad_packed_bytes(int cin, int cout, int k, int dtype);
// forward layout [tap][rows_total][cin] AND dgrad layout [taps-1-tap][conv_rows_pad(cin)][cout padded] from one pass over the OIHW weights
int k_pack_conv_both(const float* w_oihw, int cout, int cin, int k, void* dst_fwd, int rows_total, void* dst_dgrad, int dtype, hipStream_t s);
// the same for a batch of layers of one kernel size in ceil(n / 12) launches (descriptors as kernel arguments); dstd may be nullptr (no dgrad layout wanted);
// rows_d / kpad / gx / blk0 are filled in by the launcher
struct PackDesc { const float* w; void* dstf; void* dstd; int cout, cin, rows_total, rows_d, kpad, gx, blk0; };
struct PackBatch { static constexpr int MAX = 12; PackDesc d[MAX]; int n; };
int k_pack_conv_both_batch(const PackDesc* descs, int n, int k, int dtype, hipStream_t s);
// db / dtemb (optional): bias gradient and per-image column sums of dy (rows of length dtemb_ld), taken from the same pass that transposes dy
int conv_wgrad(Ctx& c, int mode, const Tens& x0, const Tens* x1, const Tens& dy, int cout, float* dw, bool accumulate, float* db = nullptr, float* dtemb = nullptr,
               int dtemb_ld = 0);
int colsum(Ctx& c, const Tens& dy, float* out, bool per_image, bool accumulate, int out_ld = 0);
int gn_act_backward(Ctx& c, const NormW& nw, const Tens& x0, const Tens* x1, const float* mean_rstd, const Tens& dy, int silu, void* dx0, bool acc0, void* dx1,
                    bool acc1, float* dgamma, float* dbeta, bool acc_param);

}  // namespace wdm

struct wdm_handle {
    int device;
};
