// C ABI entry points other than the UNet object (unet.hip): DWT/IDWT, boundary layout helpers, the DDIM update,
// and the per-block entry points the parity tests call (they run the very same run_resblock / run_attn / run_conv
// the UNet executor runs, on weights given in the reference layout).
#include "common.h"

using namespace wdm;

namespace {

// bump-allocate from the test scratch buffer
struct Scratch {
    Arena ar;
    Scratch(void* p, size_t n) : ar(p, n) {}
    template <typename T> int get(size_t count, T** out) {
        *out = (T*)ar.alloc(count * sizeof(T));
        if (!*out) WDM_FAIL(WDM_ENOMEM, "scratch too small");
        return WDM_OK;
    }
};

int pack_conv_w(Scratch& sc, const float* w, const float* b, int cin, int cout, int k, int dtype, hipStream_t s, ConvW* out, bool sm_1x1 = false) {
    char* dst;
    WDM_TRY(sc.get<char>(conv_packed_bytes(cin, cout, k, dtype), &dst));
    out->cin = cin; out->cout = cout; out->k = k; out->rows_pad = conv_rows_pad(cout);
    WDM_TRY(k_pack_conv(w, cout, cin, k, dst, out->rows_pad, 0, 1, dtype, s));
    out->w = dst; out->b = b;
    if (conv_sm_eligible(dtype, k, cin)) {              // same selection as the UNet executor: the slab-major copy next to the plain matrix
        char* sm;
        WDM_TRY(sc.get<char>(conv_packed_bytes(cin, cout, k, dtype), &sm));
        WDM_TRY(k_pack_conv_sm(w, cout, cin, sm, out->rows_pad, s, dtype));
        out->w_sm = sm;
    } else if (sm_1x1 && k == 1 && is_h16(dtype) && cin % 32 == 0) {      // the folded AttnBlock's matrices: [cin / 32][rows][32] for the fused core (unet.hip: refold)
        char* sm;
        WDM_TRY(sc.get<char>(conv_packed_bytes(cin, cout, k, dtype), &sm));
        WDM_TRY(k_pack_conv_sm(w, cout, cin, sm, out->rows_pad, s, dtype, 1));
        out->w_sm = sm;
    }
    return WDM_OK;
}

int to_nhwc(Scratch& sc, Ctx& c, const float* x, int C, int H, int W, Tens* t) {
    char* p;
    WDM_TRY(sc.get<char>((size_t)c.B * C * H * W * dsize(c.dtype), &p));
    WDM_TRY(k_nchw_to_nhwc(x, p, c.B, C, H, W, c.dtype, c.s));
    t->p = p; t->C = C; t->H = H; t->W = W; t->xs = C;
    return WDM_OK;
}

}  // namespace

extern "C" {

int wdm_dwt_fwd(wdm_handle* h, const float* x, float* y, int B, int H, int W, void* stream) {
    if (!h || !x || !y) WDM_FAIL(WDM_EINVAL, "wdm_dwt_fwd: null argument");
    return k_dwt_fwd(x, y, B, H, W, (hipStream_t)stream);
}
int wdm_dwt_inv(wdm_handle* h, const float* y, float* x, int B, int hh, int ww, void* stream) {
    if (!h || !x || !y) WDM_FAIL(WDM_EINVAL, "wdm_dwt_inv: null argument");
    return k_dwt_inv(y, x, B, hh, ww, (hipStream_t)stream);
}
int wdm_dwt_fwd_affine(wdm_handle* h, const float* x, float scale, float shift, float* y, int B, int H, int W, void* stream) {
    if (!h || !x || !y) WDM_FAIL(WDM_EINVAL, "wdm_dwt_fwd_affine: null argument");
    return k_dwt_fwd(x, y, B, H, W, (hipStream_t)stream, scale, shift);
}
int wdm_dwt_inv_compose(wdm_handle* h, const float* y_lo, int lo_channels, int n_lo, const float* y_hi, float* x, int B, int hh, int ww, int to_unit_range,
                        void* stream) {
    if (!h || !x || !y_hi) WDM_FAIL(WDM_EINVAL, "wdm_dwt_inv_compose: null argument");
    return k_dwt_inv(y_hi, x, B, hh, ww, (hipStream_t)stream, y_lo, lo_channels, n_lo, to_unit_range);
}
int wdm_pack_channels(wdm_handle* h, const float* src, int nch, int H, int W, const int32_t* patches, int n, int p, void* x96, int c_total, int c_off,
                      int dtype, void* stream) {
    if (!h || !src || !x96) WDM_FAIL(WDM_EINVAL, "wdm_pack_channels: null argument");
    return k_pack_channels(src, nch, H, W, patches, n, p, x96, c_total, c_off, dtype, (hipStream_t)stream);
}
int wdm_ddim_update(wdm_handle* h, const float* eps, const int32_t* patches, int n, int p, const float* x_t, int nimg, int H, int W, float sqrt_1m_at,
                    float sqrt_at, float sqrt_at_next, float c2, float* x0_out, float* x_next_out, void* stream) {
    if (!h || !eps || !x_t || !x0_out || !x_next_out) WDM_FAIL(WDM_EINVAL, "wdm_ddim_update: null argument");
    return k_ddim_update(eps, patches, n, p, x_t, nimg, H, W, sqrt_1m_at, sqrt_at, sqrt_at_next, c2, x0_out, x_next_out, (hipStream_t)stream);
}
int wdm_ddim_update_eta(wdm_handle* h, const float* eps, const int32_t* patches, int n, int p, const float* x_t, int nimg, int H, int W, float sqrt_1m_at,
                        float sqrt_at, float sqrt_at_next, float c1, float c2, const float* noise, float* x0_out, float* x_next_out, void* stream) {
    if (!h || !eps || !x_t || !x0_out || !x_next_out || !noise) WDM_FAIL(WDM_EINVAL, "wdm_ddim_update_eta: null argument");
    return k_ddim_update(eps, patches, n, p, x_t, nimg, H, W, sqrt_1m_at, sqrt_at, sqrt_at_next, c2, x0_out, x_next_out, (hipStream_t)stream, noise, c1);
}
int wdm_patch_accumulate(wdm_handle* h, const float* eps, const int32_t* patches, int n, int p, int nimg, int H, int W, float* acc_cnt, void* stream) {
    if (!h || !acc_cnt || (n > 0 && (!eps || !patches))) WDM_FAIL(WDM_EINVAL, "wdm_patch_accumulate: null argument");
    if (n == 0) {      // a rank that owns no patch of this step contributes zeros
        WDM_HIP(hipMemsetAsync(acc_cnt, 0, (size_t)nimg * 3 * H * W * 2 * sizeof(float), (hipStream_t)stream));
        return WDM_OK;
    }
    return k_patch_accumulate(eps, patches, n, p, nimg, H, W, acc_cnt, (hipStream_t)stream);
}
int wdm_ddim_from_sums(wdm_handle* h, const float* acc_cnt, const float* x_t, int nimg, int H, int W, float sqrt_1m_at, float sqrt_at, float sqrt_at_next,
                       float c2, float* x0_out, float* x_next_out, void* stream) {
    if (!h || !acc_cnt || !x_t || !x0_out || !x_next_out) WDM_FAIL(WDM_EINVAL, "wdm_ddim_from_sums: null argument");
    return k_ddim_from_sums(acc_cnt, x_t, nimg, H, W, sqrt_1m_at, sqrt_at, sqrt_at_next, c2, x0_out, x_next_out, (hipStream_t)stream);
}
int wdm_nchw_to_nhwc(wdm_handle* h, const float* src, void* dst, int B, int C, int H, int W, int dtype, void* stream) {
    if (!h || !src || !dst) WDM_FAIL(WDM_EINVAL, "wdm_nchw_to_nhwc: null argument");
    return k_nchw_to_nhwc(src, dst, B, C, H, W, dtype, (hipStream_t)stream);
}
int wdm_nhwc_to_nchw(wdm_handle* h, const void* src, float* dst, int B, int C, int H, int W, int dtype, void* stream) {
    if (!h || !src || !dst) WDM_FAIL(WDM_EINVAL, "wdm_nhwc_to_nchw: null argument");
    return k_nhwc_to_nchw(src, dst, B, C, H, W, dtype, (hipStream_t)stream);
}

// ---- per-block entry points -----------------------------------------------------------------------------------
int wdm_resblock_forward(wdm_handle* h, const wdm_resblock_params* p, const float* x0, int c0, const float* x1, int c1, const float* temb, int n_t,
                         int temb_ch, int B, int H, int W, float* y, int dtype, void* scratch, size_t scratch_bytes, void* stream) {
    if (!h || !p || !x0 || !temb || !y || !scratch) WDM_FAIL(WDM_EINVAL, "wdm_resblock_forward: null argument");
    if (c0 + c1 != p->cin) WDM_FAIL(WDM_EINVAL, "wdm_resblock_forward: c0+c1 != cin");
    if (n_t != 1 && n_t != B) WDM_FAIL(WDM_EINVAL, "wdm_resblock_forward: n_t must be 1 or B");
    Scratch sc(scratch, scratch_bytes);
    Ctx c{(hipStream_t)stream, dtype, B, &sc.ar, false};
    ResW w;
    w.cin = p->cin; w.cout = p->cout;
    w.n1 = NormW{p->norm1_w, p->norm1_b, p->cin};
    w.n2 = NormW{p->norm2_w, p->norm2_b, p->cout};
    WDM_TRY(pack_conv_w(sc, p->conv1_w, p->conv1_b, p->cin, p->cout, 3, dtype, c.s, &w.c1));
    WDM_TRY(pack_conv_w(sc, p->conv2_w, p->conv2_b, p->cout, p->cout, 3, dtype, c.s, &w.c2));
    w.has_nin = p->nin_w != nullptr;
    if (w.has_nin) WDM_TRY(pack_conv_w(sc, p->nin_w, p->nin_b, p->cin, p->cout, 1, dtype, c.s, &w.nin));
    float* tp;
    WDM_TRY(sc.get<float>((size_t)n_t * p->cout, &tp));
    WDM_TRY(k_linear(temb, n_t, temb_ch, p->temb_w, p->temb_b, p->cout, tp, 1, c.s));   // temb_proj(SiLU(temb)), unet.py:125
    w.temb = tp; w.temb_ld = p->cout; w.temb_per_image = n_t > 1;
    Tens t0, t1, out;
    WDM_TRY(to_nhwc(sc, c, x0, c0, H, W, &t0));
    if (c1) WDM_TRY(to_nhwc(sc, c, x1, c1, H, W, &t1));
    WDM_TRY(run_resblock(c, w, t0, c1 ? &t1 : nullptr, &out));
    return k_nhwc_to_nchw(out.p, y, B, p->cout, H, W, dtype, c.s);
}

int wdm_attn_forward(wdm_handle* h, const wdm_attn_params* p, const float* x, int B, int H, int W, float* y, int dtype, void* scratch,
                     size_t scratch_bytes, void* stream) {
    if (!h || !p || !x || !y || !scratch) WDM_FAIL(WDM_EINVAL, "wdm_attn_forward: null argument");
    Scratch sc(scratch, scratch_bytes);
    Ctx c{(hipStream_t)stream, dtype, B, &sc.ar, false};
    const int C = p->c;
    AttnW w;
    w.c = C;
    w.n = NormW{p->norm_w, p->norm_b, C};
    // fused q|k matrix + bias
    char* qkw; float* qkb;
    WDM_TRY(sc.get<char>(conv_packed_bytes(C, 2 * C, 1, dtype), &qkw));
    WDM_TRY(sc.get<float>((size_t)2 * C, &qkb));
    const int rows = conv_rows_pad(2 * C);
    WDM_TRY(k_pack_conv(p->q_w, C, C, 1, qkw, rows, 0, 0, dtype, c.s));
    WDM_TRY(k_pack_conv(p->k_w, C, C, 1, qkw, rows, C, 1, dtype, c.s));
    WDM_TRY(k_copy_f32(p->q_b, qkb, C, c.s));
    WDM_TRY(k_copy_f32(p->k_b, qkb + C, C, c.s));
    w.qk.w = qkw; w.qk.b = qkb; w.qk.cin = C; w.qk.cout = 2 * C; w.qk.k = 1; w.qk.rows_pad = rows;
    WDM_TRY(pack_conv_w(sc, p->v_w, p->v_b, C, C, 1, dtype, c.s, &w.v));
    WDM_TRY(pack_conv_w(sc, p->proj_w, p->proj_b, C, C, 1, dtype, c.s, &w.proj));
    if (is_h16(dtype)) {      // the folded operands, as the UNet's weight loader makes them (unet.hip: wdm_unet_load_param)
        float *M, *cq, *Wvp, *bvp;
        WDM_TRY(sc.get<float>((size_t)C * C, &M));
        WDM_TRY(sc.get<float>((size_t)C, &cq));
        WDM_TRY(sc.get<float>((size_t)C * C, &Wvp));
        WDM_TRY(sc.get<float>((size_t)C, &bvp));
        WDM_TRY(k_attn_fold(p->q_w, p->q_b, p->k_w, p->v_w, p->v_b, p->proj_w, p->proj_b, C, M, cq, Wvp, bvp, c.s));
        WDM_TRY(pack_conv_w(sc, M, cq, C, C, 1, dtype, c.s, &w.qf, true));
        WDM_TRY(pack_conv_w(sc, Wvp, bvp, C, C, 1, dtype, c.s, &w.pf, true));
    }
    Tens t0, out;
    WDM_TRY(to_nhwc(sc, c, x, C, H, W, &t0));
    WDM_TRY(run_attn(c, w, t0, &out));
    return k_nhwc_to_nchw(out.p, y, B, C, H, W, dtype, c.s);
}

int wdm_conv_forward(wdm_handle* h, const float* w, const float* b, int cin, int cout, int mode, const float* x, int B, int H, int W, float* y, int dtype,
                     void* scratch, size_t scratch_bytes, void* stream) {
    if (!h || !w || !x || !y || !scratch) WDM_FAIL(WDM_EINVAL, "wdm_conv_forward: null argument");
    if (mode < 0 || mode > 3) WDM_FAIL(WDM_EINVAL, "wdm_conv_forward: bad mode");
    Scratch sc(scratch, scratch_bytes);
    Ctx c{(hipStream_t)stream, dtype, B, &sc.ar, false};
    ConvW cw;
    WDM_TRY(pack_conv_w(sc, w, b, cin, cout, mode == MODE_P1 ? 1 : 3, dtype, c.s, &cw));
    if (mode == MODE_UPS && conv_up4_eligible(dtype, H, W, cin, cout)) {       // same selection as the UNet executor: sub-pixel taps next to the 3x3 ones
        char* up4;
        WDM_TRY(sc.get<char>((size_t)16 * cw.rows_pad * cin * dsize(dtype), &up4));
        WDM_TRY(k_pack_up4(w, cout, cin, up4, cw.rows_pad, c.s, dtype));
        cw.w_up4 = up4;
    }
    Tens t0, out;
    WDM_TRY(to_nhwc(sc, c, x, cin, H, W, &t0));
    WDM_TRY(run_conv(c, cw, mode, t0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &out, Y_NHWC, nullptr));
    return k_nhwc_to_nchw(out.p, y, B, cout, out.H, out.W, dtype, c.s);
}

int wdm_temb_forward(wdm_handle* h, const float* t, int n_t, int ch, const float* w0, const float* b0, const float* w1, const float* b1, float* temb_out,
                     void* scratch, size_t scratch_bytes, void* stream) {
    if (!h || !t || !w0 || !w1 || !temb_out || !scratch) WDM_FAIL(WDM_EINVAL, "wdm_temb_forward: null argument");
    Scratch sc(scratch, scratch_bytes);
    hipStream_t s = (hipStream_t)stream;
    float *emb, *t0;
    WDM_TRY(sc.get<float>((size_t)n_t * ch, &emb));
    WDM_TRY(sc.get<float>((size_t)n_t * ch * 4, &t0));
    WDM_TRY(k_timestep_embedding(t, n_t, ch, emb, s));
    WDM_TRY(k_linear(emb, n_t, ch, w0, b0, ch * 4, t0, 2, s));
    return k_linear(t0, n_t, ch * 4, w1, b1, ch * 4, temb_out, 0, s);
}

}  // extern "C"
