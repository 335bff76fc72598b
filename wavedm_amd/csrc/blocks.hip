// Host-side composition of the UNet's building blocks out of the fused kernels:
//   ResnetBlock (unet.py:119-138), AttnBlock (unet.py:168-193), Downsample / Upsample convs.
// The same functions back the whole-UNet executor (unet.hip) and the per-block C entry points used by the
// parity tests (api.hip).  In "dry" mode nothing is launched; only the arena is exercised so that the
// executor can report its exact workspace requirement.
#include <stdarg.h>

#include <algorithm>
#include <cmath>

#include <atomic>
#include <mutex>
#include "common.h"
#include "gn_inline.h"

namespace wdm {

// ---- error message (thread local) ------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// ---- arena -----------------------------------------------------------------------------------
void* Arena::alloc(size_t bytes) {
    bytes = align_up(bytes ? bytes : 1, 256);
    for (size_t i = 0; i < free_.size(); ++i) {
        if (free_[i].len >= bytes) {
            const size_t off = free_[i].off;
            free_[i].off += bytes;
            free_[i].len -= bytes;
            if (free_[i].len == 0) free_.erase(free_.begin() + i);
            used_.push_back({off, bytes});
            peak_ = std::max(peak_, off + bytes);
            // dry mode hands out fake, distinct, non-null addresses
            return base_ ? (void*)(base_ + off) : (void*)(uintptr_t)(off + 4096);
        }
    }
    failed_ = true;
    return nullptr;
}
void Arena::free(void* p) {
    if (!p) return;
    const size_t off = base_ ? (size_t)((char*)p - base_) : (size_t)((uintptr_t)p - 4096);
    for (size_t i = 0; i < used_.size(); ++i) {
        if (used_[i].off == off) {
            Blk b = used_[i];
            used_.erase(used_.begin() + i);
            auto it = std::lower_bound(free_.begin(), free_.end(), b, [](const Blk& a, const Blk& c) { return a.off < c.off; });
            it = free_.insert(it, b);
            // coalesce with neighbours
            if (it + 1 != free_.end() && it->off + it->len == (it + 1)->off) { it->len += (it + 1)->len; free_.erase(it + 1); }
            if (it != free_.begin() && (it - 1)->off + (it - 1)->len == it->off) { (it - 1)->len += it->len; free_.erase(it); }
            return;
        }
    }
}

int alloc_tens(Ctx& c, int C, int H, int W, Tens* t) {
    *t = Tens();                 // a reused variable must not carry the previous tensor's statistics / normalised copy / scale-shift rows (they are freed with the tensor)
    t->C = C; t->H = H; t->W = W; t->xs = C;
    t->p = c.ar->alloc((size_t)c.B * H * W * C * dsize(c.dtype));
    if (!t->p) WDM_FAIL(WDM_ENOMEM, "workspace too small (tensor %dx%dx%dx%d)", c.B, H, W, C);
    return WDM_OK;
}
void free_tens(Ctx& c, Tens& t) {
    c.ar->free(t.p);
    if (t.stats) c.ar->free(t.stats);
    if (t.nrm) c.ar->free(t.nrm);                        // a normalised copy nobody took
    if (t.fin_scale) { c.ar->free(t.fin_scale); c.ar->free(t.fin_shift); }      // scale / shift rows nobody took
    t.p = nullptr; t.stats = nullptr; t.gst = nullptr; t.nslab = 0; t.nrm = nullptr; t.nrm_for = nullptr; t.fin_scale = t.fin_shift = nullptr; t.fin_for = nullptr;
}

static int alloc_f32(Ctx& c, size_t n, float** p) {
    *p = (float*)c.ar->alloc(n * sizeof(float));
    if (!*p) WDM_FAIL(WDM_ENOMEM, "workspace too small (%zu floats)", n);
    return WDM_OK;
}

// The sub-pixel Upsample kernel (conv_up4_kernel.h) takes bf16 maps whose LOW-resolution size is a multiple of its 16 x 16 tile, or 8 x 8 (four images per tile); WDM_UP4=0
// keeps the 9-tap kernel everywhere (A/B runs)
// The switches are read into a fresh EnvCfg and published through an atomic pointer: a launch path on another thread sees either the old or the new
// set, never a half-written one (earlier configurations are kept alive: a reader may still hold a reference; a refresh is a test / A-B harness event).
// The switches change what the activation arena holds: after wdm_env_refresh() a workspace sized before it may be too small (the call then fails with
// WDM_ENOMEM, it never overruns).  Callers of the C ABI re-query wdm_unet_workspace_bytes after a refresh; the Python layer drops its cached workspaces
// (_lib.env_refresh() bumps a generation that DiffusionUNet.workspace() checks).
static std::atomic<const EnvCfg*> g_env{nullptr};
static std::mutex g_env_mu;
void env_cfg_refresh() {
    EnvCfg* c = new EnvCfg();
    auto flag = [](const char* name, int dflt) { const char* e = getenv(name); return e ? (e[0] == '0' ? 0 : 1) : dflt; };
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    c->conv_dma = flag("WDM_CONV_DMA", 1); c->gemm = flag("WDM_GEMM", 1); c->bn256 = num("WDM_BN256", 1);
    c->gn_tile = num("WDM_GN_TILE", 2); c->gn_inline = num("WDM_GN_INLINE", 1); c->attn_fused = num("WDM_ATTN_FUSED", 3); c->attn_fold = num("WDM_ATTN_FOLD", 1);
    c->up4 = flag("WDM_UP4", 1); c->attn_sm = flag("WDM_ATTN_SM", 1); c->wgrad_bg = num("WDM_WGRAD_BG", 0);
    std::lock_guard<std::mutex> lk(g_env_mu);
    g_env.store(c, std::memory_order_release);
}
const EnvCfg& env_cfg() {
    const EnvCfg* c = g_env.load(std::memory_order_acquire);
    if (!c) {
        {
            std::lock_guard<std::mutex> lk(g_env_mu);
            c = g_env.load(std::memory_order_acquire);
        }
        if (!c) { env_cfg_refresh(); c = g_env.load(std::memory_order_acquire); }
    }
    return *c;
}

bool conv_up4_eligible(int dtype, int H, int W, int cin, int cout) {
    return env_cfg().up4 && (is_h16(dtype) || dtype == WDM_F32X3) && ((H % 16 == 0 && W % 16 == 0) || (H == 8 && W == 8)) && cin % 32 == 0 && cout % 8 == 0 && cout >= 128;
}

int launch_conv(const ConvArgs& a0, int mode, int dtype, hipStream_t s) {
    const ConvArgs& a = a0;
    return dtype == WDM_BF16 ? launch_conv_bf16(a, mode, s) : dtype == WDM_F16 ? launch_conv_f16(a, mode, s) : dtype == WDM_F32X3 ? launch_conv_f32x3(a, mode, s) : launch_conv_f32(a, mode, s);
}
// ---- one fused convolution ---------------------------------------------------------------------
// out: allocated here (NHWC model dtype) unless y_ext is given (then y_mode says how y_ext is laid out)
int run_conv(Ctx& c, const ConvW& w, int mode, const Tens& x0, const Tens* x1, const float* scale, const float* shift, const float* temb,
             int temb_ld, int temb_per_image, const Tens* res, Tens* out, int y_mode, void* y_ext, bool want_stats, const ConvW* shortcut,
             const Tens* sx0, const Tens* sx1, const NormW* gn_inl, ConvArgs* defer, const NormW* on, int on_silu, const FinReq* fin) {
    const int Cin = x0.C + (x1 ? x1->C : 0);
    if (Cin != w.cin) WDM_FAIL(WDM_EINVAL, "conv: input has %d channels, weights expect %d", Cin, w.cin);
    if (x1 && (x1->H != x0.H || x1->W != x0.W)) WDM_FAIL(WDM_EINVAL, "conv: concat inputs differ in size");
    int Ho = x0.H, Wo = x0.W;
    if (mode == MODE_S2) { Ho = x0.H / 2; Wo = x0.W / 2; }
    if (mode == MODE_UPS) { Ho = x0.H * 2; Wo = x0.W * 2; }
    void* y = y_ext;
    if (!y_ext) {
        WDM_TRY(alloc_tens(c, w.cout, Ho, Wo, out));
        y = out->p;
        y_mode = Y_NHWC;
    }
    ConvArgs a{};
    a.x0 = x0.p; a.x1 = x1 ? x1->p : nullptr;
    a.C0 = x0.C; a.C1 = x1 ? x1->C : 0;
    a.xs0 = x0.xs; a.xs1 = x1 ? x1->xs : 0;
    a.B = c.B; a.Hin = x0.H; a.Win = x0.W; a.Hout = Ho; a.Wout = Wo;
    a.Cin = Cin; a.Cout = w.cout;
    a.w = w.w; a.w_tap_stride = (long long)w.rows_pad * w.cin; a.w_img_stride = 0; a.w_row_stride = w.cin; a.w_rows = w.rows_pad;
    a.w_bytes = (unsigned)((size_t)w.k * w.k * w.rows_pad * w.cin * dsize(c.dtype));
    a.w_sm = ((mode == MODE_S1 || (mode == MODE_S2 && is_h16(c.dtype))) && w.k == 3) ? w.w_sm : nullptr;
    a.bias = w.b; a.alpha = 1.0f;
    a.pro = (scale || gn_inl) ? 1 : 0; a.scale = scale; a.shift = shift;
    if (gn_inl) {
        if (x1 || !x0.gst || scale || gn_inl->c != Cin) WDM_FAIL(WDM_EINVAL, "conv: in-prologue GroupNorm needs a single input with group partials");
        a.gin = x0.gst; a.gin_nslab = x0.nslab; a.gn_gamma = gn_inl->g; a.gn_beta = gn_inl->b; a.gn_eps = 1e-6f;
    }
    a.temb = temb; a.temb_ld = temb_ld; a.temb_per_image = temb_per_image;
    a.res = res ? res->p : nullptr; a.res_s = res ? res->xs : 0;
    a.y = y; a.y_mode = y_mode; a.y_s = w.cout;
    if (mode == MODE_UPS && w.w_up4 && !x1 && !scale && !temb && !res && !shortcut && y_mode == Y_NHWC && conv_up4_eligible(c.dtype, x0.H, x0.W, Cin, w.cout)) {
        mode = MODE_UP4;                                  // same result from 4 pre-summed taps per output phase on the low-resolution map
        a.Hout = x0.H; a.Wout = x0.W;
        a.w = w.w_up4;
        a.w_bytes = (unsigned)((size_t)16 * w.rows_pad * w.cin * dsize(c.dtype));
    }
    if (shortcut) {      // 1x1 conv over [sx0 | sx1] accumulated into the same tile
        a.sx0 = sx0->p; a.sx1 = sx1 ? sx1->p : nullptr;
        a.sC0 = sx0->C; a.sC1 = sx1 ? sx1->C : 0; a.sxs0 = sx0->xs; a.sxs1 = sx1 ? sx1->xs : 0;
        a.sw = shortcut->w; a.sw_row_stride = shortcut->cin; a.sw_rows = shortcut->rows_pad;
        a.sw_bytes = (unsigned)((size_t)shortcut->rows_pad * shortcut->cin * dsize(c.dtype));
        a.sbias = shortcut->b;
    }
    if (want_stats && !y_ext && w.cout % 8 == 0) {
        // the producing conv also emits the GroupNorm partial statistics of its output (no extra pass over HBM)
        int nslab = 0, yn_ok = 0, fin_ok = 0;
        ConvArgs q = a;
        q.query_nslab = &nslab; q.query_yn = &yn_ok; q.query_fin = &fin_ok;
        WDM_TRY(launch_conv(q, mode, c.dtype, c.s));
        out->nslab = nslab;
        if (defer) fin_ok = 1;                            // (the fused attention core runs this conv as its third phase and arrives: attn_fused_kernel.h)
        if (on && yn_ok && env_cfg().gn_tile && on->c == w.cout) {
            // the kernel this conv runs on holds whole images x whole groups per tile: it also writes act(GroupNorm(out)) for the consumer (gn_group.h)
            out->nrm = c.ar->alloc((size_t)c.B * Ho * Wo * w.cout * dsize(c.dtype));
            if (!out->nrm) WDM_FAIL(WDM_ENOMEM, "workspace too small (normalised copy)");
            out->nrm_for = on->g; out->nrm_silu = on_silu;
            a.yn = out->nrm; a.on_gamma = on->g; a.on_beta = on->b; a.on_eps = 1e-6f; a.on_silu = on_silu;
        }
        // group-level partials ride behind the per-channel ones where a consumer can finalise from them (gn_inline.h): group widths 4 / 8 / 16
        const bool want_gst = is_h16(c.dtype) && env_cfg().gn_inline && gn_inline_shape_ok(w.cout, nslab);
        const size_t sb = gn_stats_bytes(c.B, nslab, w.cout);
        out->stats = (float*)c.ar->alloc(sb + (want_gst ? (size_t)c.B * nslab * 96 * sizeof(float) : 0));
        if (!out->stats) WDM_FAIL(WDM_ENOMEM, "workspace too small (GroupNorm statistics)");
        a.stats = out->stats; a.stats_nslab = nslab;
        if (want_gst) { out->gst = (float*)((char*)out->stats + sb); a.gst = out->gst; }
        // the consumer's GroupNorm finalised by this launch's last workgroups (gn_arrive.h) instead of a gn_finalize launch
        const int Cf = w.cout + ((fin && fin->other) ? fin->other->C : 0);
        if (fin && fin_ok && env_cfg().gn_inline >= 2 && c.fin_cnt && c.fin_used < c.fin_cap && is_h16(c.dtype) && fin->n->c == Cf && (!fin->other || fin->other->stats) &&
            (double)c.B * nslab * w.cout * 16.0 < 2147483000.0) {
            WDM_TRY(alloc_f32(c, (size_t)c.B * Cf, &out->fin_scale));
            WDM_TRY(alloc_f32(c, (size_t)c.B * Cf, &out->fin_shift));
            out->fin_for = fin->n->g; out->fin_other = fin->other ? fin->other->p : nullptr; out->fin_silu = fin->silu;
            a.fin_cnt = c.fin_cnt + (size_t)c.fin_used * c.B;
            ++c.fin_used;
            a.fin_st1 = fin->other ? fin->other->stats : nullptr; a.fin_nslab1 = fin->other ? fin->other->nslab : 0; a.fin_C1 = fin->other ? fin->other->C : 0;
            a.fin_gamma = fin->n->g; a.fin_beta = fin->n->b; a.fin_eps = 1e-6f; a.fin_premul = fin->silu ? -1.4426950408889634f : 1.0f;
            a.fin_scale = out->fin_scale; a.fin_shift = out->fin_shift;          // (fin_total: the launcher, which knows the tiling)
        }
    }
    if (c.dry) return WDM_OK;
    if (defer) { *defer = a; return WDM_OK; }        // the caller hands it to another launcher (the fused attention core's proj_out phase)
    return launch_conv(a, mode, c.dtype, c.s);
}

// GroupNorm statistics of [x0 | x1] -> scale/shift (allocated here, caller frees both).  Tensors that came out of a conv
// carry their partial statistics already (Tens::stats); for the others a partial pass over the tensor runs first.
static bool has_fin(const Tens& x0, const Tens* x1, const NormW& nw, int silu) {
    return x0.fin_scale != nullptr && x0.fin_for == nw.g && x0.fin_silu == silu && x0.fin_other == (x1 ? x1->p : nullptr);
}
bool wants_fin(const Ctx& c, int Cin, int H, int W, bool single) {
    if (env_cfg().gn_inline < 2 || !env_cfg().conv_dma || !is_h16(c.dtype) || !c.fin_cnt || H * W <= GN_PASS_MAX_HW || H % 16 || W % 16) return false;
    const bool inl = single && env_cfg().gn_inline && gn_inline_shape_ok(Cin, (H / 16) * (W / 16) * 4);      // the consumer finalises in its own prologue (gn_inline.h)
    return !inl;
}
int run_gn(Ctx& c, const NormW& nw, const Tens& x0, const Tens* x1, int for_silu_conv, float** scale, float** shift) {
    if (has_fin(x0, x1, nw, for_silu_conv)) {          // the producing kernel finalised it already (gn_arrive.h); the caller owns (and frees) the rows from here
        Tens& t = const_cast<Tens&>(x0);
        *scale = t.fin_scale; *shift = t.fin_shift;
        t.fin_scale = t.fin_shift = nullptr; t.fin_for = nullptr;
        return WDM_OK;
    }
    const int C = x0.C + (x1 ? x1->C : 0);
    const int HW = x0.H * x0.W;
    WDM_TRY(alloc_f32(c, (size_t)c.B * C, scale));
    WDM_TRY(alloc_f32(c, (size_t)c.B * C, shift));
    const Tens* src[2] = {&x0, x1};
    float* st[2] = {nullptr, nullptr};
    float* tmp[2] = {nullptr, nullptr};
    int ns[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        if (!src[k]) continue;
        if (src[k]->stats) { st[k] = src[k]->stats; ns[k] = src[k]->nslab; continue; }
        ns[k] = gn_default_nslab(HW);
        tmp[k] = (float*)c.ar->alloc(gn_stats_bytes(c.B, ns[k], src[k]->C));
        if (!tmp[k]) WDM_FAIL(WDM_ENOMEM, "workspace too small (GroupNorm statistics)");
        st[k] = tmp[k];
        if (!c.dry) WDM_TRY(k_gn_partial(*src[k], c.B, tmp[k], ns[k], c.dtype, c.s));
    }
    int rc = WDM_OK;
    if (!c.dry) rc = k_gn_finalize(c.B, HW, st[0], ns[0], x0.C, st[1], ns[1], x1 ? x1->C : 0, nw, 1e-6f, for_silu_conv, *scale, *shift, c.s);
    for (int k = 0; k < 2; ++k) if (tmp[k]) c.ar->free(tmp[k]);    // stream-ordered reuse
    return rc;
}

// ---- ResnetBlock: GN -> SiLU -> conv3x3 (+temb) -> GN -> SiLU -> conv3x3 -> + (x | nin_shortcut(x)) -------------
// Two ways to feed a conv its normalised + activated input:
//  * prologue: the conv kernel applies GN + SiLU to every staged tile (no extra pass over HBM, but every N tile of the
//    conv repeats the transform: Cout / BN times);
//  * pass: one elementwise kernel writes act(gn(x)) (and the channel concat) once, the conv runs without prologue.
// The pass wins where the tensors are small and Cout / BN is large: the 8x8 level (768 channels: 12 N tiles).

// partial statistics of x (its producer's, or a pass over the tensor): *tmp is what the caller has to free afterwards
static int gn_partials_of(Ctx& c, const Tens& x, float** st, int* ns, float** tmp) {
    *tmp = nullptr;
    if (x.stats) { *st = x.stats; *ns = x.nslab; return WDM_OK; }
    *ns = gn_default_nslab(x.H * x.W);
    *tmp = (float*)c.ar->alloc(gn_stats_bytes(c.B, *ns, x.C));
    if (!*tmp) WDM_FAIL(WDM_ENOMEM, "workspace too small (GroupNorm statistics)");
    *st = *tmp;
    if (!c.dry) WDM_TRY(k_gn_partial(x, c.B, *tmp, *ns, c.dtype, c.s));
    return WDM_OK;
}

// act(gn([x0|x1])) as one dense tensor (silu != 0: with SiLU) -- one launch (k_gn_finalize_apply) where that kernel takes the shape, else finalize + apply per tensor
static int materialize_gn(Ctx& c, const NormW& nw, const Tens& x0, const Tens* x1, int silu, Tens* out) {
    const int C = x0.C + (x1 ? x1->C : 0);
    if (!x1 && x0.nrm && x0.nrm_for == nw.g && x0.nrm_silu == silu) {
        // the producing conv wrote it already (run_conv: on); the caller owns (and frees) it from here
        Tens& src = const_cast<Tens&>(x0);
        *out = Tens();
        out->p = src.nrm; out->C = C; out->H = x0.H; out->W = x0.W; out->xs = C;
        src.nrm = nullptr; src.nrm_for = nullptr;
        return WDM_OK;
    }
    if (gn_fused_pass_eligible(x0.C, x1 ? x1->C : 0, c.dtype)) {
        float *st0 = nullptr, *st1 = nullptr, *tmp0 = nullptr, *tmp1 = nullptr;
        int ns0 = 0, ns1 = 1;
        WDM_TRY(gn_partials_of(c, x0, &st0, &ns0, &tmp0));
        if (x1) WDM_TRY(gn_partials_of(c, *x1, &st1, &ns1, &tmp1));
        WDM_TRY(alloc_tens(c, C, x0.H, x0.W, out));
        int rc = WDM_OK;
        if (!c.dry) rc = k_gn_finalize_apply(c.B, x0, x1, st0, ns0, st1, ns1, nw, 1e-6f, silu, out->p, c.dtype, c.s);
        if (tmp0) c.ar->free(tmp0);
        if (tmp1) c.ar->free(tmp1);
        return rc;
    }
    float *sc, *sh;
    WDM_TRY(run_gn(c, nw, x0, x1, 0, &sc, &sh));
    WDM_TRY(alloc_tens(c, C, x0.H, x0.W, out));
    if (!c.dry) {
        WDM_TRY(k_gn_apply(x0, c.B, sc, sh, C, out->p, C, 0, silu, c.dtype, c.s));
        if (x1) WDM_TRY(k_gn_apply(*x1, c.B, sc + x0.C, sh + x0.C, C, out->p, C, x0.C, silu, c.dtype, c.s));
    }
    c.ar->free(sc); c.ar->free(sh);
    return WDM_OK;
}
static int materialize_gn_silu(Ctx& c, const NormW& nw, const Tens& x0, const Tens* x1, Tens* out) { return materialize_gn(c, nw, x0, x1, 1, out); }

// a 3x3 conv with the GroupNorm+SiLU prologue can finalise the norm itself when it will run on an LDS-DMA 3x3 kernel (conv_dispatch.inc: bf16, 16-pixel
// multiple maps, Cout >= 128) and its single input carries group partials (Cin = 128 / 256 / 512)
// (maps up to 32 x 32: 16 slabs.  At 64 x 64 -- 64 slabs, 24 KB of partials per table -- the finalize costs +2.3 us per table; round 4 built it once per image in the
// persistent kernel, which walks an image's tiles back to back and keeps the table under the packed epilogue: 609.3 / 612.9 -> 609.7 / 614.1 img/s at 20 steps, null --
// the table set-up inside the kernel costs what the gn_finalize launch did.  Those layers keep gn_finalize.)
static bool gn_inline_ok(const Ctx& c, const Tens& x0, const Tens* x1, int cout) {
    return env_cfg().gn_inline && env_cfg().conv_dma && is_h16(c.dtype) && !x1 && x0.gst != nullptr && gn_inline_shape_ok(x0.C, x0.nslab) && x0.H % 16 == 0 &&
           x0.W % 16 == 0 && cout >= 128;
}

int run_resblock(Ctx& c, const ResW& w, const Tens& x0, const Tens* x1, Tens* out, const NormW* next_n, int next_silu, const FinReq* next_fin) {
    const int Cin = x0.C + (x1 ? x1->C : 0);
    if (Cin != w.cin) WDM_FAIL(WDM_EINVAL, "resblock: input has %d channels, block expects %d", Cin, w.cin);
    if (!w.has_nin && x1) WDM_FAIL(WDM_EINVAL, "resblock: identity shortcut cannot take a concat input");
    const bool pass = x0.H * x0.W <= GN_PASS_MAX_HW;
    float *sc1, *sh1, *sc2, *sh2;
    Tens t1, sct;
    const NormW* on12 = env_cfg().gn_tile >= 2 ? &w.n2 : nullptr;      // WDM_GN_TILE=2: conv1 also normalises for conv2 on the larger maps where its kernel can
    // conv2's GroupNorm from conv1's own launch (gn_arrive.h) where neither the pass, nor conv1's in-tile GroupNorm (16 x 16 maps), nor conv2's prologue finalises it
    const FinReq fin12{&w.n2, nullptr, 1};
    const FinReq* f12 = (wants_fin(c, w.cout, x0.H, x0.W, true) && !(on12 && x0.H == 16 && x0.W == 16)) ? &fin12 : nullptr;
    // (a GroupNorm+SiLU pass for the channel-concat inputs of the 16 x 16 up blocks -- whose four N tiles each repeat the transform -- measured null at 16 x 16 and
    // -1.5 % with the 32 x 32 maps included: round 3, EXPERIMENTS.md)
    const bool pass1 = pass;
    if (pass1) {
        Tens a1;
        WDM_TRY(materialize_gn_silu(c, w.n1, x0, x1, &a1));
        WDM_TRY(run_conv(c, w.c1, MODE_S1, a1, nullptr, nullptr, nullptr, w.temb, w.temb_ld, w.temb_per_image, nullptr, &t1, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr,
                         nullptr, pass ? &w.n2 : on12, 1));      // ... and act(norm2(h)) for conv2 where the kernel can (8 x 8 maps: conv_dma8_kernel.h; 16 x 16: on12)
        free_tens(c, a1);
    } else if (!has_fin(x0, x1, w.n1, 1) && gn_inline_ok(c, x0, x1, w.cout)) {
        // conv1's GroupNorm finalised in conv1's own prologue from the producer's group partials: no gn_finalize launch (gn_inline.h)
        WDM_TRY(run_conv(c, w.c1, MODE_S1, x0, nullptr, nullptr, nullptr, w.temb, w.temb_ld, w.temb_per_image, nullptr, &t1, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, &w.n1,
                         nullptr, on12, 1, f12));
    } else {
        WDM_TRY(run_gn(c, w.n1, x0, x1, 1, &sc1, &sh1));
        WDM_TRY(run_conv(c, w.c1, MODE_S1, x0, x1, sc1, sh1, w.temb, w.temb_ld, w.temb_per_image, nullptr, &t1, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr, nullptr, on12, 1, f12));
        c.ar->free(sc1); c.ar->free(sh1);
    }
    // conv1 wrote act(norm2(h)) itself (16 x 16 maps: its tile is the whole image): conv2 then runs WITHOUT the prologue, as on the 8 x 8 maps -- every one of
    // its N tiles would otherwise repeat the GroupNorm+SiLU of the same halo slabs (Cout / 128 = 4 times on these maps)
    const bool pre2 = !pass && t1.nrm != nullptr && t1.nrm_for == w.n2.g && t1.nrm_silu == 1;
    // the 1x1 shortcut either runs as its own GEMM (result added in conv2's epilogue) or, where conv2 runs on the LDS-DMA kernel,
    // as a second K phase of conv2 itself: x_shortcut + h is then one fp32 accumulator and the shortcut tensor never exists
    // (8 x 8 maps: conv2 has no prologue there and runs on conv_dma8_kernel.h; WDM_CONV_DMA=0 takes the LDS-DMA kernels, hence the fusion, away)
    const bool fuse_nin = w.has_nin && env_cfg().conv_dma && (!pass || (x0.H == 8 && x0.W == 8 && is_h16(c.dtype))) &&
                          (is_h16(c.dtype) || (c.dtype == WDM_F32X3 && x0.H % 16 == 0 && x0.W % 16 == 0)) &&
                          conv_can_fuse_shortcut(x0.H, x0.W, w.cout, w.cout, x0.C, x1 ? x1->C : 0);
    const Tens* res = &x0;
    if (w.has_nin && !fuse_nin) {
        WDM_TRY(run_conv(c, w.nin, MODE_P1, x0, x1, nullptr, nullptr, nullptr, 0, 0, nullptr, &sct, Y_NHWC, nullptr));
        res = &sct;
    }
    if (pass || pre2) {
        Tens a2;
        WDM_TRY(materialize_gn_silu(c, w.n2, t1, nullptr, &a2));
        if (fuse_nin) WDM_TRY(run_conv(c, w.c2, MODE_S1, a2, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, out, Y_NHWC, nullptr, true, &w.nin, &x0, x1, nullptr, nullptr, next_n, next_silu, next_fin));
        else WDM_TRY(run_conv(c, w.c2, MODE_S1, a2, nullptr, nullptr, nullptr, nullptr, 0, 0, res, out, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr, nullptr, next_n, next_silu, next_fin));
        free_tens(c, a2);
    } else if (!has_fin(t1, nullptr, w.n2, 1) && gn_inline_ok(c, t1, nullptr, w.cout)) {
        if (fuse_nin) WDM_TRY(run_conv(c, w.c2, MODE_S1, t1, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, out, Y_NHWC, nullptr, true, &w.nin, &x0, x1, &w.n2, nullptr, next_n, next_silu, next_fin));
        else WDM_TRY(run_conv(c, w.c2, MODE_S1, t1, nullptr, nullptr, nullptr, nullptr, 0, 0, res, out, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, &w.n2, nullptr, next_n, next_silu, next_fin));
    } else {
        WDM_TRY(run_gn(c, w.n2, t1, nullptr, 1, &sc2, &sh2));
        if (fuse_nin) WDM_TRY(run_conv(c, w.c2, MODE_S1, t1, nullptr, sc2, sh2, nullptr, 0, 0, nullptr, out, Y_NHWC, nullptr, true, &w.nin, &x0, x1, nullptr, nullptr, next_n, next_silu, next_fin));
        else WDM_TRY(run_conv(c, w.c2, MODE_S1, t1, nullptr, sc2, sh2, nullptr, 0, 0, res, out, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr, nullptr, next_n, next_silu, next_fin));
        c.ar->free(sc2); c.ar->free(sh2);
    }
    free_tens(c, t1);
    if (w.has_nin && !fuse_nin) free_tens(c, sct);
    return WDM_OK;
}

// ---- AttnBlock: GN -> q,k,v 1x1 -> softmax(q^T k * C^-1/2) -> v.w^T -> proj_out 1x1 -> + x ----------------------
// All four contractions run on the conv kernel: Q.K^T and P.V are 1x1 convolutions whose "weights" are the
// image's own K (rows = keys) and V^T (rows = channels; produced by storing the v projection channel-major).
int run_attn(Ctx& c, const AttnW& w, const Tens& x, Tens* out, const FinReq* next_fin) {
    const int C = w.c, N = x.H * x.W;
    if (x.C != C) WDM_FAIL(WDM_EINVAL, "attn: input has %d channels, block expects %d", x.C, C);
    if (N % 64 || N > 512) WDM_FAIL(WDM_EINVAL, "attn: %d tokens unsupported (multiple of 64, <= 512)", N);
    const size_t es = dsize(c.dtype);
    Tens hn;
    WDM_TRY(materialize_gn(c, w.n, x, nullptr, 0, &hn));

    const bool fused = attn_fused_eligible(c.dtype, N, C);
    // the 8 x 8 maps' block (64 tokens): the fused core in its block-diagonal form -- four images per 256-row "image" of the kernel, scores outside an image's own block masked
    // (attn_fused_kernel.h: bdiag) -- on the folded operands; any batch size (a ragged last group is skipped per query block), so an image's bits do not depend on the batch
    const bool bdiag = N == 64 && attn_fused_eligible(c.dtype, 256, C) && x.H == 8 && x.W == 8;
    if ((fused || bdiag) && env_cfg().attn_fold && w.qf.w && w.pf.w && w.qf.cin == C && w.qf.cout == C && w.pf.cin == C && w.pf.cout == C) {
        // Folded form (k_attn_fold; 16-bit modes): softmax_j((Wq h_i + bq).(Wk h_j + bk)) = softmax_j((M h_i + cq).h_j) and proj_out(P.(Wv h + bv)) = Wvp (P.h) + bvp, so
        // the normalised input itself is K and V of the core: ONE projection GEMM (q' = M h + cq) instead of three, no V^T tensor, and proj_out runs on Wvp.
        Tens qf, o;
        const bool proj_in = C <= 512 && env_cfg().attn_fused >= 2 && x.H == 16 && x.W == 16;
        // WDM_ATTN_FUSED=3: q' = Mq h + cq as phase 0 of the core (attn_fused_kernel.h: QPROJ) -- same MFMA sequence and rounding as the GEMM it replaces, hence the same bits
        const bool q_in = proj_in && env_cfg().attn_fused >= 3 && w.qf.b != nullptr;
        AttnOperands in;
        in.k = hn.p; in.k_ld = hn.xs; in.v = hn.p; in.v_ld = hn.xs; in.v_tok = 1; in.bdiag = bdiag ? 1 : 0;
        if (q_in) {
            const bool sm = env_cfg().attn_sm && w.qf.w_sm != nullptr;
            in.qw = sm ? w.qf.w_sm : w.qf.w; in.qw_slab = sm ? w.qf.rows_pad * 32 : 0;
            in.qbias = w.qf.b; in.qw_ld = w.qf.cin; in.qw_bytes = (size_t)w.qf.rows_pad * w.qf.cin * es;
        } else {
            WDM_TRY(run_conv(c, w.qf, MODE_P1, hn, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &qf, Y_NHWC, nullptr));      // [B][N][C]
            in.q = qf.p; in.q_ld = qf.xs;
        }
        if (proj_in) {
            ConvArgs a_proj{};
            Tens odummy;
            odummy.p = hn.p; odummy.C = C; odummy.H = x.H; odummy.W = x.W; odummy.xs = C;           // stands for O in run_conv's shape checks only
            WDM_TRY(run_conv(c, w.pf, MODE_P1, odummy, nullptr, nullptr, nullptr, nullptr, 0, 0, &x, out, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr, &a_proj, nullptr, 0,
                             next_fin));
            if (env_cfg().attn_sm && w.pf.w_sm) a_proj.w_sm = w.pf.w_sm;      // phase 3 streams the slab-major copy of Wp Wv
            if (!c.dry) WDM_TRY(launch_attn_fused(in, nullptr, c.B, C, c.s, nullptr, &a_proj, c.dtype));
            if (!q_in) free_tens(c, qf);
            free_tens(c, hn);
            return WDM_OK;
        }
        WDM_TRY(alloc_tens(c, C, x.H, x.W, &o));
        if (!c.dry) WDM_TRY(launch_attn_fused(in, o.p, c.B, C, c.s, nullptr, nullptr, c.dtype));
        free_tens(c, qf); free_tens(c, hn);
        WDM_TRY(run_conv(c, w.pf, MODE_P1, o, nullptr, nullptr, nullptr, nullptr, 0, 0, &x, out, Y_NHWC, nullptr, true));
        free_tens(c, o);
        return WDM_OK;
    }
    Tens qk;
    WDM_TRY(run_conv(c, w.qk, MODE_P1, hn, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &qk, Y_NHWC, nullptr));      // [B][N][2C]
    AttnOperands in;
    in.q = qk.p; in.k = (const char*)qk.p + (size_t)C * es; in.q_ld = in.k_ld = qk.xs;
    void* vT = c.ar->alloc((size_t)c.B * C * N * es);                                                               // [B][C][N]
    if (!vT) WDM_FAIL(WDM_ENOMEM, "workspace too small (attention V^T)");
    in.v = vT;
    // V^T[b] = W_v . h[b]^T as a batched GEMM whose row operand is the weight matrix (shared by the images) and whose per-image "weights" are the
    // tokens: the output rows are channels, so V^T comes out of the ordinary 16-byte-store epilogue instead of the channel-major scalar one
    // (29 -> 18 us).  Its bias moves behind the softmax (attn_fused_kernel.h).  Other shapes: the conv form with the channel-major epilogue.
    // (f32x3 mode, unfused core: the same form on conv_gemmx3_kernel.h, the bias then rides on the P.V product -- softmax rows sum to one)
    const bool x3_vt = c.dtype == WDM_F32X3 && env_cfg().conv_dma && env_cfg().gemm && N == 256;
    const bool v_as_gemm = (fused || x3_vt) && C % 256 == 0 && w.v.rows_pad == C && w.v.cin == C;
    if (v_as_gemm) {
        if (!c.dry) {
            ConvArgs a{};
            a.x0 = w.v.w; a.C0 = C; a.xs0 = C; a.C1 = 0; a.x_img_shared = 1;
            a.B = c.B; a.Hin = a.Hout = C / 16; a.Win = a.Wout = 16;
            a.Cin = C; a.Cout = N;
            a.w = hn.p; a.w_tap_stride = 0; a.w_img_stride = (long long)N * hn.xs; a.w_row_stride = hn.xs; a.w_rows = N;
            a.w_bytes = (unsigned)((size_t)N * hn.xs * es);
            a.alpha = 1.0f;
            a.y = vT; a.y_mode = Y_NHWC; a.y_s = N;
            WDM_TRY(launch_conv(a, MODE_P1, c.dtype, c.s));
        }
    } else {
        Tens dummy;
        WDM_TRY(run_conv(c, w.v, MODE_P1, hn, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &dummy, Y_NCHW, vT));
    }
    free_tens(c, hn);

    Tens o;
    const bool fuse_proj = fused && C <= 512 && env_cfg().attn_fused >= 2 && x.H == 16 && x.W == 16 && w.proj.cin == C && w.proj.cout == C;
    if (fuse_proj) {
        // ... and proj_out with its residual and the next norm's statistics as a third phase of the same kernel: O never reaches HBM
        ConvArgs a_proj{};
        Tens odummy;
        odummy.p = qk.p; odummy.C = C; odummy.H = x.H; odummy.W = x.W; odummy.xs = C;           // stands for O in run_conv's shape checks only
        WDM_TRY(run_conv(c, w.proj, MODE_P1, odummy, nullptr, nullptr, nullptr, nullptr, 0, 0, &x, out, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr, &a_proj, nullptr, 0,
                         next_fin));
        if (!c.dry) WDM_TRY(launch_attn_fused(in, nullptr, c.B, C, c.s, v_as_gemm ? w.v.b : nullptr, &a_proj, c.dtype));
        c.ar->free(vT);
        free_tens(c, qk);
        return WDM_OK;
    }
    if (fused) {
        // scores, softmax and P.V in one kernel: S and P never leave the CU (attn_fused_kernel.h)
        WDM_TRY(alloc_tens(c, C, x.H, x.W, &o));
        if (!c.dry) WDM_TRY(launch_attn_fused(in, o.p, c.B, C, c.s, v_as_gemm ? w.v.b : nullptr, nullptr, c.dtype));
        c.ar->free(vT);
    } else {
    float* S = nullptr;
    WDM_TRY(alloc_f32(c, (size_t)c.B * N * N, &S));
    void* P = c.ar->alloc((size_t)c.B * N * N * es);
    if (!P) WDM_FAIL(WDM_ENOMEM, "workspace too small (attention P)");
    WDM_TRY(alloc_tens(c, C, x.H, x.W, &o));
    if (!c.dry) {
        ConvArgs a{};
        // S[b][i][j] = C^-1/2 * sum_c q[b][i][c] k[b][j][c]
        a.x0 = qk.p; a.C0 = C; a.xs0 = 2 * C; a.C1 = 0;
        a.B = c.B; a.Hin = a.Hout = x.H; a.Win = a.Wout = x.W;
        a.Cin = C; a.Cout = N;
        a.w = (const char*)qk.p + (size_t)C * es; a.w_tap_stride = 0; a.w_img_stride = (long long)N * 2 * C; a.w_row_stride = 2 * C; a.w_rows = N;
        a.w_bytes = (unsigned)(((size_t)N * 2 * C - C) * es);     // this image's K rows (descriptor base moves per image)
        a.alpha = (float)std::pow((double)C, -0.5);
        a.y = S; a.y_mode = Y_NHWC_F32; a.y_s = N;
        WDM_TRY(launch_conv(a, MODE_P1, c.dtype, c.s));
        WDM_TRY(k_softmax_rows(S, P, (long long)c.B * N, N, c.dtype, c.s));
        // O[b][i][c] = sum_j P[b][i][j] V^T[b][c][j]
        ConvArgs p{};
        p.x0 = P; p.C0 = N; p.xs0 = N; p.C1 = 0;
        p.B = c.B; p.Hin = p.Hout = x.H; p.Win = p.Wout = x.W;
        p.Cin = N; p.Cout = C;
        p.w = vT; p.w_tap_stride = 0; p.w_img_stride = (long long)C * N; p.w_row_stride = N; p.w_rows = C;
        p.w_bytes = (unsigned)((size_t)C * N * es);
        p.alpha = 1.0f;
        p.bias = v_as_gemm ? w.v.b : nullptr;            // V^T came without its bias (GEMM form): sum_j P[i][j] (v[j][c] + b[c]) = (P v)[i][c] + b[c]
        p.y = o.p; p.y_mode = Y_NHWC; p.y_s = C;
        WDM_TRY(launch_conv(p, MODE_P1, c.dtype, c.s));
    }
    c.ar->free(S); c.ar->free(P); c.ar->free(vT);
    }
    free_tens(c, qk);
    WDM_TRY(run_conv(c, w.proj, MODE_P1, o, nullptr, nullptr, nullptr, nullptr, 0, 0, &x, out, Y_NHWC, nullptr, true));
    free_tens(c, o);
    return WDM_OK;
}

}  // namespace wdm
