// Whole-UNet executor: builds the layer list of the reference's DiffusionUNet (models/unet.py:197-307) from the
// config, owns the layout of the single packed-weight buffer, and runs forward (unet.py:346-395) as a fixed sequence
// of fused-kernel launches on one stream.  Parameter names / shapes are exactly the reference's state_dict keys.
#include <string.h>

#include <map>
#include <mutex>
#include <string>

#include "common.h"

namespace wdm {
const char* get_error();
}
using namespace wdm;

namespace {

enum ParamKind { PK_CONV, PK_F32 };

struct ParamSlot {
    std::string name;
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    ParamKind kind = PK_F32;
    size_t off = 0;       // byte offset of the destination matrix / vector in the packed buffer
    int rows_total = 0;   // PK_CONV: rows of the destination matrix (fused qk: 2C)
    int row_off = 0;      // PK_CONV: first destination row;  PK_F32: element offset inside the destination vector
    bool zero_tail = true;   // PK_CONV: this slot also zero-fills the padding rows behind it
    int cin_dst = 0;         // PK_CONV: row length of the destination when it is padded beyond shape[1] (conv_in), else 0
    size_t sm_off = 0;       // PK_CONV of a bf16 3x3 conv: second destination, the slab-major copy (k_pack_conv_sm); 0 = none
    size_t up4_off = 0;      // PK_CONV of an Upsample conv (bf16): second destination, the 16 sub-pixel taps (k_pack_up4); 0 = none
    bool loaded = false;
    int fold = -1, fold_role = -1;   // tensor of an AttnBlock whose folded operands are rebuilt when it is loaded (16-bit modes): index into wdm_unet::folds, FR_* role
    int64_t numel() const { int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i]; return n; }
};

struct ConvD { size_t w_off, b_off; int cin, cout, k, rows_pad; size_t up4_off, sm_off; };
struct NormD { size_t g_off, b_off; int c; };
struct ResD { int cin, cout; NormD n1, n2; ConvD c1, c2, nin; bool has_nin; int temb_row; };
struct AttnD { int c; NormD n; ConvD qk, v, proj, qf, pf; int fold; };
// folded AttnBlock operands (common.h: k_attn_fold): the fp32 originals of the four matrices are kept ([Wq | Wk | Wv | Wp] at stage_off) so that any one tensor can be
// reloaded; the biases' fp32 copies are the ones the unfolded convs use
enum { FR_QW = 0, FR_KW, FR_VW, FR_PW, FR_QB, FR_VB, FR_PB, FR_N };
struct FoldD { int c; size_t stage_off; ConvD qk, v, proj, qf, pf; int slot[FR_N]; };

}  // namespace

struct wdm_unet {
    wdm_handle* h;
    wdm_unet_config cfg;
    std::vector<ParamSlot> params;
    std::map<std::string, int> index;
    size_t packed_bytes = 0;
    size_t range_flag_off = 0;      // WDM_F16 only
    std::vector<FoldD> folds;       // 16-bit modes: one per AttnBlock
    size_t fold_tmp_floats = 0, fold_tmp_off = 0;
    char* packed = nullptr;
    bool all_loaded = false;
    int temb_ch = 0, temb_rows = 0;   // rows of the concatenated temb_proj matrix
    size_t temb_w_off = 0, temb_b_off = 0, d0w = 0, d0b = 0, d1w = 0, d1b = 0;
    ConvD conv_in, conv_out;
    NormD norm_out;
    std::vector<std::vector<ResD>> down_res, up_res;
    std::vector<std::vector<AttnD>> down_attn, up_attn;
    std::vector<ConvD> down_ds, up_us;   // per level (unused entries have cin == 0)
    ResD mid1, mid2;
    AttnD mid_attn;

    // ---- construction helpers
    size_t take(size_t bytes) { size_t o = packed_bytes; packed_bytes = align_up(packed_bytes + bytes, 256); return o; }
    void add_param(const std::string& name, std::initializer_list<int64_t> shp, ParamKind kind, size_t off, int rows_total, int row_off) {
        ParamSlot p;
        p.name = name; p.kind = kind; p.off = off; p.rows_total = rows_total; p.row_off = row_off;
        p.ndim = (int)shp.size();
        int i = 0;
        for (auto v : shp) p.shape[i++] = v;
        index[name] = (int)params.size();
        params.push_back(p);
    }
    // cin_pad > cin: the kernels' K dimension is padded with zero columns (conv_in of a model whose input width is not a multiple of 32)
    // s1: the conv gets the slab-major weight copy: stride-1 convs, and since round 5 the Downsample convs of the 16-bit modes (conv_s2_kernel.h); Upsample convs
    // never read theirs (their sub-pixel taps are slab-major themselves: k_pack_up4): less to pack after every trainer sync, a smaller weight broadcast
    ConvD add_conv(const std::string& name, int cin, int cout, int k, int cin_pad = 0, bool s1 = true) {
        ConvD d{};
        if (cin_pad <= 0) cin_pad = cin;
        d.cin = cin_pad; d.cout = cout; d.k = k; d.rows_pad = conv_rows_pad(cout);
        d.w_off = take(conv_packed_bytes(cin_pad, cout, k, cfg.dtype));
        d.b_off = take((size_t)cout * 4);
        add_param(name + ".weight", {cout, cin, k, k}, PK_CONV, d.w_off, d.rows_pad, 0);
        params.back().cin_dst = cin_pad != cin ? cin_pad : 0;
        if (s1 && cin_pad == cin && conv_sm_eligible(cfg.dtype, k, cin)) {
            d.sm_off = take(conv_packed_bytes(cin, cout, k, cfg.dtype));
            params.back().sm_off = d.sm_off;
        }
        add_param(name + ".bias", {cout}, PK_F32, d.b_off, 0, 0);
        return d;
    }
    NormD add_norm(const std::string& name, int c) {
        NormD d{};
        d.c = c;
        d.g_off = take((size_t)c * 4);
        d.b_off = take((size_t)c * 4);
        add_param(name + ".weight", {c}, PK_F32, d.g_off, 0, 0);
        add_param(name + ".bias", {c}, PK_F32, d.b_off, 0, 0);
        return d;
    }
    ResD add_res(const std::string& name, int cin, int cout, std::vector<std::pair<std::string, int>>& temb_list) {
        ResD r{};
        r.cin = cin; r.cout = cout;
        r.n1 = add_norm(name + ".norm1", cin);
        r.c1 = add_conv(name + ".conv1", cin, cout, 3);
        r.temb_row = temb_rows;
        temb_list.push_back({name + ".temb_proj", cout});
        temb_rows += cout;
        r.n2 = add_norm(name + ".norm2", cout);
        r.c2 = add_conv(name + ".conv2", cout, cout, 3);
        r.has_nin = cin != cout;
        if (r.has_nin) r.nin = add_conv(name + ".nin_shortcut", cin, cout, 1);
        return r;
    }
    AttnD add_attn(const std::string& name, int c) {
        AttnD a{};
        a.c = c;
        a.n = add_norm(name + ".norm", c);
        // q and k are fused into one [2C][C] matrix: one pass over the normalised input produces both
        ConvD qk{};
        qk.cin = c; qk.cout = 2 * c; qk.k = 1; qk.rows_pad = conv_rows_pad(2 * c);
        qk.w_off = take(conv_packed_bytes(c, 2 * c, 1, cfg.dtype));
        qk.b_off = take((size_t)2 * c * 4);
        add_param(name + ".q.weight", {c, c, 1, 1}, PK_CONV, qk.w_off, qk.rows_pad, 0);
        params.back().zero_tail = false;   // the k slot below owns the rows behind q
        add_param(name + ".q.bias", {c}, PK_F32, qk.b_off, 0, 0);
        add_param(name + ".k.weight", {c, c, 1, 1}, PK_CONV, qk.w_off, qk.rows_pad, c);
        add_param(name + ".k.bias", {c}, PK_F32, qk.b_off, 0, c);
        a.qk = qk;
        a.v = add_conv(name + ".v", c, c, 1);
        a.proj = add_conv(name + ".proj_out", c, c, 1);
        a.fold = -1;
        if (is_h16(cfg.dtype)) {
            // the folded operands the fused core runs on (blocks.hip: run_attn), rebuilt by wdm_unet_load_param whenever one of the seven tensors behind them arrives
            FoldD f{};
            f.c = c;
            auto plain = [&](ConvD& d) { d = ConvD{}; d.cin = c; d.cout = c; d.k = 1; d.rows_pad = conv_rows_pad(c); d.w_off = take(conv_packed_bytes(c, c, 1, cfg.dtype)); d.b_off = take((size_t)c * 4);
                                        if (c % 32 == 0) d.sm_off = take(conv_packed_bytes(c, c, 1, cfg.dtype)); };      // + the slab-major copy the fused core streams (attn_fused_kernel.h)
            plain(a.qf); plain(a.pf);
            f.stage_off = take((size_t)4 * c * c * 4);
            f.qk = a.qk; f.v = a.v; f.proj = a.proj; f.qf = a.qf; f.pf = a.pf;
            const char* names[FR_N] = {".q.weight", ".k.weight", ".v.weight", ".proj_out.weight", ".q.bias", ".v.bias", ".proj_out.bias"};
            for (int r = 0; r < FR_N; ++r) {
                f.slot[r] = index.at(name + names[r]);
                params[f.slot[r]].fold = (int)folds.size();
                params[f.slot[r]].fold_role = r;
            }
            fold_tmp_floats = std::max(fold_tmp_floats, (size_t)c * c);
            a.fold = (int)folds.size();
            folds.push_back(f);
        }
        return a;
    }

    int build();
    ConvW cw(const ConvD& d) const { ConvW w; w.w = packed + d.w_off; w.w_up4 = d.up4_off ? packed + d.up4_off : nullptr; w.w_sm = d.sm_off ? packed + d.sm_off : nullptr; w.b = (const float*)(packed + d.b_off); w.cin = d.cin; w.cout = d.cout; w.k = d.k; w.rows_pad = d.rows_pad; return w; }
    NormW nw(const NormD& d) const { NormW n; n.g = (const float*)(packed + d.g_off); n.b = (const float*)(packed + d.b_off); n.c = d.c; return n; }
    ResW rw(const ResD& d, const float* temb_all, int n_t) const {
        ResW r;
        r.cin = d.cin; r.cout = d.cout; r.n1 = nw(d.n1); r.n2 = nw(d.n2); r.c1 = cw(d.c1); r.c2 = cw(d.c2);
        r.has_nin = d.has_nin;
        if (d.has_nin) r.nin = cw(d.nin);
        r.temb = temb_all ? temb_all + d.temb_row : nullptr;
        r.temb_ld = temb_rows;
        r.temb_per_image = n_t > 1;
        return r;
    }
    AttnW aw(const AttnD& d) const { AttnW a; a.c = d.c; a.n = nw(d.n); a.qk = cw(d.qk); a.v = cw(d.v); a.proj = cw(d.proj); if (d.fold >= 0) { a.qf = cw(d.qf); a.pf = cw(d.pf); } return a; }

    int refold(const ParamSlot& p, const float* dev_src, hipStream_t s);
    int check_f16_range(const float* dev, int64_t n, const char* what, hipStream_t s, float limit = 65504.0f);
    int forward(Ctx& c, const void* x96, const float* t, int n_t, float* eps_out, const float* temb_pre = nullptr);
    int temb_table(Ctx& c, const float* t, int n_t, float* temb_all);
};

int wdm_unet::build() {
    const int ch = cfg.ch, nres = cfg.n_levels, nrb = cfg.num_res_blocks;
    temb_ch = ch * 4;
    auto is_attn = [&](int res) { for (int i = 0; i < cfg.n_attn_res; ++i) if (cfg.attn_resolutions[i] == res) return true; return false; };
    std::vector<std::pair<std::string, int>> temb_list;

    d0w = take((size_t)temb_ch * ch * 4); d0b = take((size_t)temb_ch * 4);
    add_param("temb.dense.0.weight", {temb_ch, ch}, PK_F32, d0w, 0, 0);
    add_param("temb.dense.0.bias", {temb_ch}, PK_F32, d0b, 0, 0);
    d1w = take((size_t)temb_ch * temb_ch * 4); d1b = take((size_t)temb_ch * 4);
    add_param("temb.dense.1.weight", {temb_ch, temb_ch}, PK_F32, d1w, 0, 0);
    add_param("temb.dense.1.bias", {temb_ch}, PK_F32, d1b, 0, 0);
    conv_in = add_conv("conv_in", cfg.in_channels, ch, 3, (int)align_up((size_t)cfg.in_channels, 32));

    int res = cfg.resolution;
    int block_in = ch;
    down_res.resize(nres); down_attn.resize(nres); down_ds.assign(nres, ConvD{});
    up_res.resize(nres); up_attn.resize(nres); up_us.assign(nres, ConvD{});
    for (int l = 0; l < nres; ++l) {
        block_in = ch * (l == 0 ? 1 : cfg.ch_mult[l - 1]);
        const int block_out = ch * cfg.ch_mult[l];
        for (int b = 0; b < nrb; ++b) {
            down_res[l].push_back(add_res("down." + std::to_string(l) + ".block." + std::to_string(b), block_in, block_out, temb_list));
            block_in = block_out;
        }
        if (is_attn(res))
            for (int b = 0; b < nrb; ++b) down_attn[l].push_back(add_attn("down." + std::to_string(l) + ".attn." + std::to_string(b), block_out));
        if (l != nres - 1) {
            down_ds[l] = add_conv("down." + std::to_string(l) + ".downsample.conv", block_in, block_in, 3, 0, is_h16(cfg.dtype));      // (16-bit: conv_s2_kernel.h streams the slab-major copy)
            res /= 2;
        }
    }
    mid1 = add_res("mid.block_1", block_in, block_in, temb_list);
    mid_attn = add_attn("mid.attn_1", block_in);
    mid2 = add_res("mid.block_2", block_in, block_in, temb_list);
    for (int l = nres - 1; l >= 0; --l) {
        const int block_out = ch * cfg.ch_mult[l];
        int skip_in = ch * cfg.ch_mult[l];
        for (int b = 0; b <= nrb; ++b) {
            if (b == nrb) skip_in = ch * (l == 0 ? 1 : cfg.ch_mult[l - 1]);
            up_res[l].push_back(add_res("up." + std::to_string(l) + ".block." + std::to_string(b), block_in + skip_in, block_out, temb_list));
            block_in = block_out;
        }
        if (is_attn(res))
            for (int b = 0; b <= nrb; ++b) up_attn[l].push_back(add_attn("up." + std::to_string(l) + ".attn." + std::to_string(b), block_out));
        if (l != 0) {
            up_us[l] = add_conv("up." + std::to_string(l) + ".upsample.conv", block_in, block_in, 3, 0, false);
            if ((is_h16(cfg.dtype) || cfg.dtype == WDM_F32X3) && block_in % 32 == 0 && block_in >= 128) {       // sub-pixel taps for conv_up4_kernel.h / conv_up4x3_kernel.h, next to the 3x3 ones
                up_us[l].up4_off = take((size_t)16 * up_us[l].rows_pad * block_in * dsize(cfg.dtype));
                params[index["up." + std::to_string(l) + ".upsample.conv.weight"]].up4_off = up_us[l].up4_off;
            }
            res *= 2;
        }
    }
    norm_out = add_norm("norm_out", block_in);
    conv_out = add_conv("conv_out", block_in, cfg.out_ch, 3);

    // all temb_proj Linear layers concatenated into one [sum(cout)][temb_ch] fp32 matrix: one GEMV launch per step
    if (fold_tmp_floats) fold_tmp_off = take(fold_tmp_floats * 4);      // one C x C fp32 product of the AttnBlock folding, before it is packed
    if (cfg.dtype == WDM_F16) range_flag_off = take(256);      // device int the weight loader's fp16 range check writes (wdm_unet_load_param)
    temb_w_off = take((size_t)temb_rows * temb_ch * 4);
    temb_b_off = take((size_t)temb_rows * 4);
    int row = 0;
    for (auto& e : temb_list) {
        add_param(e.first + ".weight", {e.second, temb_ch}, PK_F32, temb_w_off, 0, row * temb_ch);
        add_param(e.first + ".bias", {e.second}, PK_F32, temb_b_off, 0, row);
        row += e.second;
    }
    return WDM_OK;
}

// fp16 operands: a weight outside +-65504 (or not finite) would become inf in the packed matrix -- refuse it, loudly.  The one place the weight loader waits for the
// stream (set-up path; the sampling path never does)
int wdm_unet::check_f16_range(const float* dev, int64_t n, const char* what, hipStream_t s, float limit) {
    int* flag = (int*)(packed + range_flag_off);
    int host_flag = 0;
    WDM_HIP(hipMemsetAsync(flag, 0, sizeof(int), s));
    WDM_TRY(k_flag_out_of_range(dev, n, limit, flag, s));
    WDM_HIP(hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    WDM_HIP(hipStreamSynchronize(s));
    if (host_flag) WDM_FAIL(WDM_EINVAL, "parameter '%s': a value lies outside the fp16 range (|w| <= %.0f, finite) -- use WDM_BF16 or WDM_F32X3 for this checkpoint", what, (double)limit);
    return WDM_OK;
}

// A tensor of an AttnBlock arrived: keep the matrix's fp32 original and rebuild the folded operand it belongs to once all of that operand's tensors are there
// (q.weight, k.weight, q.bias -> qf;  v.weight, proj_out.weight, v.bias, proj_out.bias -> pf;  k.bias cancels in the softmax)
int wdm_unet::refold(const ParamSlot& p, const float* dev_src, hipStream_t s) {
    const FoldD& f = folds[p.fold];
    const size_t cc = (size_t)f.c * f.c;
    float* stage = (float*)(packed + f.stage_off);
    if (p.fold_role <= FR_PW) WDM_TRY(k_copy_f32(dev_src, stage + p.fold_role * cc, (long long)cc, s));
    const bool qk_side = p.fold_role == FR_QW || p.fold_role == FR_KW || p.fold_role == FR_QB;
    auto have = [&](std::initializer_list<int> roles) { for (int r : roles) if (!params[f.slot[r]].loaded) return false; return true; };
    float* tmp = (float*)(packed + fold_tmp_off);
    const ConvD& dst = qk_side ? f.qf : f.pf;
    if (qk_side) {
        if (!have({FR_QW, FR_KW, FR_QB})) return WDM_OK;
        WDM_TRY(k_attn_fold(stage + FR_QW * cc, (const float*)(packed + f.qk.b_off), stage + FR_KW * cc, nullptr, nullptr, nullptr, nullptr, f.c, tmp, (float*)(packed + dst.b_off), nullptr,
                            nullptr, s));
    } else {
        if (!have({FR_VW, FR_PW, FR_VB, FR_PB})) return WDM_OK;
        WDM_TRY(k_attn_fold(nullptr, nullptr, nullptr, stage + FR_VW * cc, (const float*)(packed + f.v.b_off), stage + FR_PW * cc, (const float*)(packed + f.proj.b_off), f.c, nullptr, nullptr,
                            tmp, (float*)(packed + dst.b_off), s));
    }
    if (cfg.dtype == WDM_F16) WDM_TRY(check_f16_range(tmp, (int64_t)cc, qk_side ? "Wk^T Wq of an AttnBlock" : "Wp Wv of an AttnBlock", s));
    WDM_TRY(k_pack_conv(tmp, f.c, f.c, 1, packed + dst.w_off, dst.rows_pad, 0, 1, cfg.dtype, s));
    if (dst.sm_off) WDM_TRY(k_pack_conv_sm(tmp, f.c, f.c, packed + dst.sm_off, dst.rows_pad, s, cfg.dtype, 1));
    return WDM_OK;
}

int wdm_unet::temb_table(Ctx& c, const float* t, int n_t, float* temb_all) {
    // timestep embedding MLP + every block's temb projection: depends only on t (unet.py:10-28, 225-230, 354-357, 125)
    float *emb, *t0, *t1;
    auto af = [&](size_t n, float** p) -> int { *p = (float*)c.ar->alloc(n * 4); if (!*p) WDM_FAIL(WDM_ENOMEM, "workspace too small"); return WDM_OK; };
    WDM_TRY(af((size_t)n_t * cfg.ch, &emb));
    WDM_TRY(af((size_t)n_t * temb_ch, &t0));
    WDM_TRY(af((size_t)n_t * temb_ch, &t1));
    if (!c.dry) {
        WDM_TRY(k_timestep_embedding(t, n_t, cfg.ch, emb, c.s));
        WDM_TRY(k_linear(emb, n_t, cfg.ch, (const float*)(packed + d0w), (const float*)(packed + d0b), temb_ch, t0, 2, c.s));
        WDM_TRY(k_linear(t0, n_t, temb_ch, (const float*)(packed + d1w), (const float*)(packed + d1b), temb_ch, t1, 0, c.s));
        WDM_TRY(k_linear(t1, n_t, temb_ch, (const float*)(packed + temb_w_off), (const float*)(packed + temb_b_off), temb_rows, temb_all, 1, c.s));
    }
    c.ar->free(emb); c.ar->free(t0); c.ar->free(t1);
    return WDM_OK;
}

int wdm_unet::forward(Ctx& c, const void* x96, const float* t, int n_t, float* eps_out, const float* temb_pre) {
    const int nres = cfg.n_levels, nrb = cfg.num_res_blocks, R = cfg.resolution;
    // ---- timestep embedding MLP + every block's temb projection (depends only on t): computed here, or taken from a table the caller made for
    // all the timesteps of a sampling run at once (temb_pre: one row, shared by the images)
    float* temb_all = nullptr;
    if (temb_pre) { temb_all = const_cast<float*>(temb_pre); n_t = 1; }
    else {
        temb_all = (float*)c.ar->alloc((size_t)n_t * temb_rows * 4);
        if (!temb_all) WDM_FAIL(WDM_ENOMEM, "workspace too small");
        WDM_TRY(temb_table(c, t, n_t, temb_all));
    }

    Tens x;
    x.p = (void*)x96; x.C = cfg.in_channels; x.H = R; x.W = R; x.xs = cfg.in_channels;
    void* xpad = nullptr;
    if (conv_in.cin != cfg.in_channels) {      // input width not a multiple of the K slab: zero-padded copy (the weights carry zero columns)
        xpad = c.ar->alloc((size_t)c.B * R * R * conv_in.cin * dsize(c.dtype));
        if (!xpad) WDM_FAIL(WDM_ENOMEM, "workspace too small (padded input)");
        if (!c.dry) WDM_TRY(k_pad_channels(x96, cfg.in_channels, conv_in.cin, xpad, (long long)c.B * R * R, c.dtype, c.s));
        x.p = xpad; x.C = conv_in.cin; x.xs = conv_in.cin;
    }
    // arrival counters of the launches that finalise their consumer's GroupNorm themselves (gn_arrive.h; ~17 of them per call): one slot of B ints each, zeroed here
    // (a last arriver resets its counter, but the arena hands out whatever the previous call left at these addresses)
    // Only with WDM_GN_INLINE=2 (opt-in, slower: gn_arrive.h): the default path neither allocates nor zeroes them -- one launch fewer on the serial chain; the dry
    // sizing run follows the same switch (a refresh of the switches invalidates workspace sizes anyway: blocks.hip).
    constexpr int FIN_SLOTS = 48;
    c.fin_cap = FIN_SLOTS; c.fin_used = 0; c.fin_cnt = nullptr;
    if (env_cfg().gn_inline >= 2) {
        c.fin_cnt = (int*)c.ar->alloc((size_t)FIN_SLOTS * c.B * sizeof(int));
        if (!c.fin_cnt) WDM_FAIL(WDM_ENOMEM, "workspace too small (arrival counters)");
        if (!c.dry) WDM_HIP(hipMemsetAsync(c.fin_cnt, 0, (size_t)FIN_SLOTS * c.B * sizeof(int), c.s));
    }
    // the norm a tensor meets next, for the producer to finalise (run_conv: fin) -- only where wants_fin says the consumer would otherwise launch gn_finalize
    NormW fn;
    FinReq fr{&fn, nullptr, 1};
    auto fin_for = [&](const NormW& n, const Tens* other, int C0, int H, int W) -> const FinReq* {
        if (!wants_fin(c, C0 + (other ? other->C : 0), H, W, other == nullptr)) return nullptr;
        fn = n; fr.other = other; fr.silu = 1;
        return &fr;
    };
    std::vector<Tens> hs;
    Tens h;
    WDM_TRY(run_conv(c, cw(conv_in), MODE_S1, x, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &h, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                     fin_for(rw(down_res[0][0], temb_all, n_t).n1, nullptr, conv_in.cout, R, R)));
    if (xpad) c.ar->free(xpad);
    hs.push_back(h);
    for (int l = 0; l < nres; ++l) {
        for (int b = 0; b < nrb; ++b) {
            Tens o;
            // the consumer of the block's output, when it normalises in a pass of its own (an AttnBlock; the next ResnetBlock on the <= 8 x 8 maps): the block's
            // conv2 writes that norm too where its kernel holds whole images x groups per tile (run_conv: on) -- no gn_finalize_apply launch then
            NormW nn; int nn_silu = 0; bool have_nn = false;
            const int hw_l = (R >> l) * (R >> l);
            if (!down_attn[l].empty()) { nn = aw(down_attn[l][b]).n; have_nn = true; }
            else if (hw_l <= 64) {           // the levels whose ResnetBlocks normalise in a pass (blocks.hip: GN_PASS_MAX_HW)
                if (b + 1 < nrb) { nn = rw(down_res[l][b + 1], temb_all, n_t).n1; nn_silu = 1; have_nn = true; }
                else if (l == nres - 1) { nn = rw(mid1, temb_all, n_t).n1; nn_silu = 1; have_nn = true; }
            }
            // ... or the next ResnetBlock's conv1 on the larger maps: its norm1 from this block's conv2 launch (gn_arrive.h)
            const FinReq* nf = nullptr;
            if (down_attn[l].empty() && b + 1 < nrb) { const ResW nxt = rw(down_res[l][b + 1], temb_all, n_t); nf = fin_for(nxt.n1, nullptr, nxt.cin, R >> l, R >> l); }
            WDM_TRY(run_resblock(c, rw(down_res[l][b], temb_all, n_t), hs.back(), nullptr, &o, have_nn ? &nn : nullptr, nn_silu, nf));
            if (!down_attn[l].empty()) {
                Tens o2;
                WDM_TRY(run_attn(c, aw(down_attn[l][b]), o, &o2));
                free_tens(c, o);
                o = o2;
            }
            hs.push_back(o);
        }
        if (l != nres - 1) {
            Tens o;
            WDM_TRY(run_conv(c, cw(down_ds[l]), MODE_S2, hs.back(), nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &o, Y_NHWC, nullptr, true));
            hs.push_back(o);
        }
    }
    // ---- middle (input stays on the skip stack)
    Tens m1, m2;
    const NormW mid_n = aw(mid_attn).n;
    WDM_TRY(run_resblock(c, rw(mid1, temb_all, n_t), hs.back(), nullptr, &m1, &mid_n, 0));
    WDM_TRY(run_attn(c, aw(mid_attn), m1, &m2));
    free_tens(c, m1);
    WDM_TRY(run_resblock(c, rw(mid2, temb_all, n_t), m2, nullptr, &h));
    free_tens(c, m2);
    // ---- up path: ResnetBlock on cat([h, skip]) without materialising the concat
    for (int l = nres - 1; l >= 0; --l) {
        for (int b = 0; b <= nrb; ++b) {
            Tens skip = hs.back();
            hs.pop_back();
            Tens o;
            NormW nn;
            if (!up_attn[l].empty()) nn = aw(up_attn[l][b]).n;
            // what meets this block's (or its AttnBlock's) output next: the next block's norm1 over [output | the next skip], or norm_out at the very end
            const FinReq* nf = nullptr;
            const int Rl = R >> l;
            if (b < nrb) { const ResW nxt = rw(up_res[l][b + 1], temb_all, n_t); nf = fin_for(nxt.n1, &hs.back(), nxt.cin - hs.back().C, Rl, Rl); }
            else if (l == 0) nf = fin_for(nw(norm_out), nullptr, cfg.ch * cfg.ch_mult[0], Rl, Rl);
            WDM_TRY(run_resblock(c, rw(up_res[l][b], temb_all, n_t), h, &skip, &o, up_attn[l].empty() ? nullptr : &nn, 0, up_attn[l].empty() ? nf : nullptr));
            free_tens(c, h);
            free_tens(c, skip);
            h = o;
            if (!up_attn[l].empty()) {
                Tens o2;
                WDM_TRY(run_attn(c, aw(up_attn[l][b]), h, &o2, nf));
                free_tens(c, h);
                h = o2;
            }
        }
        if (l != 0) {
            Tens o;
            const ResW nxt = rw(up_res[l - 1][0], temb_all, n_t);       // the first block of the next level: norm1 over [upsampled | skip]
            WDM_TRY(run_conv(c, cw(up_us[l]), MODE_UPS, h, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, &o, Y_NHWC, nullptr, true, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                             fin_for(nxt.n1, &hs.back(), nxt.cin - hs.back().C, 2 * (R >> l), 2 * (R >> l))));
            free_tens(c, h);
            h = o;
        }
    }
    // ---- norm_out -> SiLU -> conv_out, written as NCHW fp32 (the reference's output layout)
    {
        float *sc, *sh;
        WDM_TRY(run_gn(c, nw(norm_out), h, nullptr, 1, &sc, &sh));
        Tens dummy;
        WDM_TRY(run_conv(c, cw(conv_out), MODE_S1, h, nullptr, sc, sh, nullptr, 0, 0, nullptr, &dummy, Y_NCHW_F32, eps_out));
        c.ar->free(sc); c.ar->free(sh);
    }
    free_tens(c, h);
    c.ar->free(c.fin_cnt);
    c.fin_cnt = nullptr;
    if (!temb_pre) c.ar->free(temb_all);
    if (!hs.empty()) WDM_FAIL(WDM_ESTATE, "internal: skip stack not empty (%zu)", hs.size());
    return WDM_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================

extern "C" {

int wdm_abi_version(void) { return WDM_ABI_VERSION; }
const char* wdm_last_error(void) { return wdm::get_error(); }

int wdm_create(int device, wdm_handle** out) {
    if (!out) WDM_FAIL(WDM_EINVAL, "wdm_create: null out");
    int n = 0;
    WDM_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) WDM_FAIL(WDM_EINVAL, "wdm_create: device %d of %d", device, n);
    hipDeviceProp_t prop;
    WDM_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) WDM_FAIL(WDM_EINVAL, "wdm_create: device is %s, this library is built for gfx950 only", prop.gcnArchName);
    *out = new wdm_handle{device};
    return WDM_OK;
}
int wdm_destroy(wdm_handle* h) { delete h; return WDM_OK; }

int wdm_unet_create(wdm_handle* h, const wdm_unet_config* cfg, wdm_unet** out) {
    if (!cfg || !out) WDM_FAIL(WDM_EINVAL, "wdm_unet_create: null argument");   // h may be NULL for host-only layout queries
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->n_attn_res < 0 || cfg->n_attn_res > 8) WDM_FAIL(WDM_EINVAL, "wdm_unet_create: bad level count");
    if (!dtype_valid(cfg->dtype)) WDM_FAIL(WDM_EINVAL, "wdm_unet_create: bad dtype");
    if (!cfg->resamp_with_conv) WDM_FAIL(WDM_EINVAL, "wdm_unet_create: resamp_with_conv=False is not supported");
    if (cfg->ch % 32 || cfg->in_channels < 1) WDM_FAIL(WDM_EINVAL, "wdm_unet_create: ch must be a multiple of 32");
    if (cfg->resolution % (8 << (cfg->n_levels - 1))) WDM_FAIL(WDM_EINVAL, "wdm_unet_create: resolution %d too small for %d levels (coarsest level must be a multiple of 8)", cfg->resolution, cfg->n_levels);
    if (cfg->out_ch < 1) WDM_FAIL(WDM_EINVAL, "wdm_unet_create: out_ch must be positive");
    wdm_unet* u = new wdm_unet();
    u->h = h;
    u->cfg = *cfg;
    int rc = u->build();
    if (rc != WDM_OK) { delete u; return rc; }
    *out = u;
    return WDM_OK;
}
int wdm_unet_destroy(wdm_unet* u) { delete u; return WDM_OK; }
int wdm_unet_num_params(const wdm_unet* u) { return u ? (int)u->params.size() : 0; }
int wdm_unet_param_info(const wdm_unet* u, int i, const char** name, int* ndim, int64_t shape[4]) {
    if (!u || i < 0 || i >= (int)u->params.size()) WDM_FAIL(WDM_EINVAL, "wdm_unet_param_info: index %d out of range", i);
    const ParamSlot& p = u->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = p.ndim;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    return WDM_OK;
}
size_t wdm_unet_packed_bytes(const wdm_unet* u) { return u ? u->packed_bytes : 0; }
int wdm_unet_set_packed(wdm_unet* u, void* packed, size_t bytes) {
    if (!u || !packed) WDM_FAIL(WDM_EINVAL, "wdm_unet_set_packed: null argument");
    if (bytes < u->packed_bytes) WDM_FAIL(WDM_ENOMEM, "wdm_unet_set_packed: %zu bytes given, %zu needed", bytes, u->packed_bytes);
    if (((uintptr_t)packed) & 255) WDM_FAIL(WDM_EINVAL, "wdm_unet_set_packed: buffer must be 256-byte aligned");
    u->packed = (char*)packed;
    for (auto& p : u->params) p.loaded = false;
    u->all_loaded = false;
    return WDM_OK;
}
int wdm_unet_load_param(wdm_unet* u, const char* name, const float* dev_src, int64_t numel, void* stream) {
    if (!u || !name || !dev_src) WDM_FAIL(WDM_EINVAL, "wdm_unet_load_param: null argument");
    if (!u->packed) WDM_FAIL(WDM_ESTATE, "wdm_unet_load_param: call wdm_unet_set_packed first");
    auto it = u->index.find(name);
    if (it == u->index.end()) WDM_FAIL(WDM_ENOTFOUND, "unknown parameter '%s'", name);
    ParamSlot& p = u->params[it->second];
    if (numel != p.numel()) WDM_FAIL(WDM_EINVAL, "parameter '%s': %lld elements given, %lld expected", name, (long long)numel, (long long)p.numel());
    hipStream_t s = (hipStream_t)stream;
    if (p.kind == PK_CONV) {
        const int cout = (int)p.shape[0], cin = (int)p.shape[1], k = (int)p.shape[2];
        // (an Upsample conv is ALSO packed as pre-summed sub-pixel taps, up to four weights each, k_pack_up4: a quarter of the range keeps every sum inside it -- ADVICE r5)
        if (u->cfg.dtype == WDM_F16) WDM_TRY(u->check_f16_range(dev_src, numel, name, s, p.up4_off ? 65504.0f / 4 : 65504.0f));
        WDM_TRY(k_pack_conv(dev_src, cout, cin, k, u->packed + p.off, p.rows_total, p.row_off, p.zero_tail ? 1 : 0, u->cfg.dtype, s, p.cin_dst));
        if (p.sm_off) WDM_TRY(k_pack_conv_sm(dev_src, cout, cin, u->packed + p.sm_off, p.rows_total, s, u->cfg.dtype));
        if (p.up4_off) WDM_TRY(k_pack_up4(dev_src, cout, cin, u->packed + p.up4_off, p.rows_total, s, u->cfg.dtype));
    } else {
        WDM_TRY(k_copy_f32(dev_src, (float*)(u->packed + p.off) + p.row_off, numel, s));
    }
    p.loaded = true;
    if (p.fold >= 0) {
        const int rc = u->refold(p, dev_src, s);
        if (rc != WDM_OK) {                   // the folded operand was not rebuilt: this tensor does not count as loaded, and nothing runs until it is
            p.loaded = false;
            u->all_loaded = false;
            return rc;
        }
    }
    bool all = true;
    for (auto& q : u->params) all = all && q.loaded;
    u->all_loaded = all;
    return WDM_OK;
}
int wdm_unet_mark_loaded(wdm_unet* u) {
    if (!u || !u->packed) WDM_FAIL(WDM_ESTATE, "wdm_unet_mark_loaded: no packed buffer");
    for (auto& p : u->params) p.loaded = true;
    u->all_loaded = true;
    return WDM_OK;
}
size_t wdm_unet_workspace_bytes(const wdm_unet* u, int B) {
    if (!u || B <= 0) return 0;
    Arena ar = Arena::dry();
    Ctx c{nullptr, u->cfg.dtype, B, &ar, true};
    int rc = const_cast<wdm_unet*>(u)->forward(c, (const void*)(uintptr_t)4096, nullptr, 1, nullptr);
    // n_t = B needs B-1 more rows of temb scratch than n_t = 1: account for the larger case
    if (rc != WDM_OK) return 0;
    const size_t extra = (size_t)(B - 1) * (u->cfg.ch + 2 * (size_t)u->temb_ch + u->temb_rows) * 4 + 4096;
    return ar.peak() + align_up(extra, 256);
}
int wdm_unet_temb_rows(const wdm_unet* u) { return u ? u->temb_rows : 0; }
int wdm_unet_temb_table(wdm_unet* u, const float* t, int n, float* temb_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!u || !t || !temb_out || !workspace || n <= 0) WDM_FAIL(WDM_EINVAL, "wdm_unet_temb_table: null argument");
    if (!u->all_loaded) WDM_FAIL(WDM_ESTATE, "wdm_unet_temb_table: parameters not loaded");
    if (((uintptr_t)workspace) & 255) WDM_FAIL(WDM_EINVAL, "wdm_unet_temb_table: workspace must be 256-byte aligned");
    Arena ar(workspace, workspace_bytes);
    Ctx c{(hipStream_t)stream, u->cfg.dtype, 1, &ar, false};
    return u->temb_table(c, t, n, temb_out);
}
int wdm_unet_forward_temb(wdm_unet* u, const void* x96, const float* temb_row, int B, float* eps_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!u || !x96 || !temb_row || !eps_out || !workspace) WDM_FAIL(WDM_EINVAL, "wdm_unet_forward_temb: null argument");
    if (!u->all_loaded) WDM_FAIL(WDM_ESTATE, "wdm_unet_forward_temb: parameters not loaded");
    if (B <= 0) WDM_FAIL(WDM_EINVAL, "wdm_unet_forward_temb: B=%d", B);
    if (((uintptr_t)workspace) & 255) WDM_FAIL(WDM_EINVAL, "wdm_unet_forward_temb: workspace must be 256-byte aligned");
    // (Round 3 could replay this call as a captured hipGraph -- WDM_GRAPH=1 --: 115.8 / 116.0 img/s against 116.2 / 116.1 eager at the full 100 steps.  The host
    // is never the bound (one C call per UNet forward) and a graph node pays the same kernel-to-kernel gap as an eager launch; removed in round 4.)
    Arena ar(workspace, workspace_bytes);
    Ctx c{(hipStream_t)stream, u->cfg.dtype, B, &ar, false};
    return u->forward(c, x96, nullptr, 1, eps_out, temb_row);
}
int wdm_unet_forward(wdm_unet* u, const void* x96, const float* t, int n_t, int B, float* eps_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!u || !x96 || !t || !eps_out || !workspace) WDM_FAIL(WDM_EINVAL, "wdm_unet_forward: null argument");
    if (!u->all_loaded) WDM_FAIL(WDM_ESTATE, "wdm_unet_forward: parameters not loaded");
    if (B <= 0 || (n_t != 1 && n_t != B)) WDM_FAIL(WDM_EINVAL, "wdm_unet_forward: n_t=%d must be 1 or B=%d", n_t, B);
    if (((uintptr_t)workspace) & 255) WDM_FAIL(WDM_EINVAL, "wdm_unet_forward: workspace must be 256-byte aligned");
    Arena ar(workspace, workspace_bytes);
    Ctx c{(hipStream_t)stream, u->cfg.dtype, B, &ar, false};
    return u->forward(c, x96, t, n_t, eps_out);
}

}  // extern "C"
