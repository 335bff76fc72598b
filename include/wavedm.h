/*
 * wavedm.h -- C ABI of libwavedm_hip.so: the MI355X (gfx950) implementation of WaveDM's sampling
 * hot path: 2-level Haar wavelet-packet DWT/IDWT, the wavelet-domain diffusion
 * UNet forward, and the per-step DDIM patch gather / scatter-mean / update.
 *
 * The reference has no FFI layer (it is pure Python on torch); each entry point
 * below names the reference function whose device work it replaces.  The reference-side binding
 * (ctypes) is shown in INTEGRATION.md and implemented in wavedm_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 (WDM_OK) or a negative WDM_E* code and never throws;
 *     wdm_last_error() gives the message of the last failure on the calling thread;
 *   - the library never allocates device memory: the caller (torch) owns every buffer including
 *     the packed-weight buffer and the workspace, and passes raw device pointers;
 *   - all kernels are enqueued on the `stream` argument (a hipStream_t; pass torch's current
 *     stream); no call synchronises the device;
 *   - one wdm_handle per device, one wdm_unet per model; a handle/unet is not thread-safe,
 *     distinct handles are independent;
 *   - "NCHW f32" tensors are the reference's layout at the boundary; inside the UNet
 *     activations are NHWC (channels-last) in the model dtype (WDM_BF16 / WDM_F16: 2 bytes, WDM_F32 / WDM_F32X3: 4).
 */
#ifndef WAVEDM_H
#define WAVEDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WDM_ABI_VERSION 1

enum {
    WDM_OK = 0,
    WDM_EINVAL = -1,    /* bad argument / unsupported shape */
    WDM_ENOMEM = -2,    /* caller-provided buffer too small */
    WDM_EHIP = -3,      /* a HIP runtime call failed */
    WDM_ESTATE = -4,    /* call order violated (e.g. forward before all params loaded) */
    WDM_ENOTFOUND = -5  /* unknown parameter name */
};

/* WDM_F32X3: fp32 tensors exactly as WDM_F32 (same buffers, same elementwise kernels); only the contractions differ -- each product is three
 * bf16 MFMAs on operands split hi + lo in registers (~1e-5 end to end, several times the rate of the exact-fp32 MFMA chain).  UNet / block /
 * conv entry points take it; the HFRM and the trainer take WDM_F32 or WDM_BF16.
 * WDM_F16: the bf16 path on IEEE half operands -- fp16 tensors and packed weights, v_mfma_f32_16x16x32_f16 (the bf16 MFMA's rate), fp32 accumulators,
 * GroupNorm statistics, softmax and DDIM state exactly as in WDM_BF16.  Three more mantissa bits than bf16 (unit round-off 2^-11): the mode meant to
 * meet the 1e-3 parity bound at the bf16 mode's speed.  The price is range (|x| <= 65504): wdm_unet_load_param fails with WDM_EINVAL on a weight outside
 * it.  UNet / block / conv / layout entry points take it; the HFRM and the trainer do not. */
enum { WDM_F32 = 0, WDM_BF16 = 1, WDM_F32X3 = 2, WDM_F16 = 3 };

typedef struct wdm_handle wdm_handle;
typedef struct wdm_unet wdm_unet;

/* ---- library / handle ------------------------------------------------------------------- */
int wdm_abi_version(void);
const char* wdm_last_error(void);
int wdm_create(int device, wdm_handle** out);
int wdm_destroy(wdm_handle* h);

/* ---- Haar wavelet-packet transform ---------------------------------------------------------
 * Replaces WaveletTransform.forward, models/wavelet.py:37-49 (scale=2, transpose=True):
 *   fwd: x (B,3,H,W) NCHW f32  ->  y (B,48,H/4,W/4) NCHW f32, channel = subband*3 + rgb
 *   inv: y (B,48,h,w)          ->  x (B,3,4h,4w)
 * H and W must be multiples of 4. */
int wdm_dwt_fwd(wdm_handle* h, const float* x, float* y, int B, int H, int W, void* stream);
int wdm_dwt_inv(wdm_handle* h, const float* y, float* x, int B, int hh, int ww, void* stream);
/* The same transforms with the element-wise steps the reference runs around them folded in (no extra pass, no ATen kernel in the loop):
 *   wdm_dwt_fwd_affine:   y = DWT(scale * x + shift)            -- data_transform, 2x - 1 (restoration.py:8-9, ddm_wavelet.py:345-348)
 *   wdm_dwt_inv_compose:  coefficient channels [0, n_lo) are read from y_lo (B, lo_channels, h, w), the others from y_hi (B,48,h,w) -- the
 *                         torch.cat([x0[:, :pc], hf_wav[:, pc:]]) of restoration.py:114 -- and, if to_unit_range, the output is
 *                         clamp((x + 1) / 2, 0, 1) = inverse_data_transform (restoration.py:12-13).  n_lo = 0: y_lo may be NULL. */
int wdm_dwt_fwd_affine(wdm_handle* h, const float* x, float scale, float shift, float* y, int B, int H, int W, void* stream);
int wdm_dwt_inv_compose(wdm_handle* h, const float* y_lo, int lo_channels, int n_lo, const float* y_hi, float* x, int B, int hh, int ww,
                        int to_unit_range, void* stream);

/* ---- layout helpers at the UNet boundary ---------------------------------------------------
 * A "patch list" is n triples (img, hi, wi) of int32 on the DEVICE: patch k is the p x p window
 * at (hi, wi) of image `img` of a (NIMG, C, H, W) NCHW f32 tensor.  It restates the reference's
 * `corners` list (models/ddm_wavelet.py:419) with an explicit image index so that B independent
 * 64x64 crops (corners = [(0,0)] each) and one stitched image (45 corners) use the same kernels.
 *
 * wdm_pack_channels: gather `nch` channels of src (NIMG, nch, H, W) NCHW f32 for every patch into
 * channels [c_off, c_off+nch) of the NHWC UNet input x96 (n, p, p, 96) of dtype `dtype`.
 * Replaces crop()+cat at models/ddm_wavelet.py:467-478 (channel order [x_cond 0:48 | x_t 48:51 |
 * x_other 51:96]). */
int wdm_pack_channels(wdm_handle* h, const float* src, int nch, int H, int W, const int32_t* patches, int n,
                      int p, void* x96, int c_total, int c_off, int dtype, void* stream);

/* wdm_ddim_update: scatter-add of the predicted noise patches into the full image in patch-list
 * order, division by the overlap count, and the eta=0 DDIM update; replaces
 * models/ddm_wavelet.py:485-502.
 *   eps      (n, 3, p, p) NCHW f32 : UNet output per patch
 *   x_t      (NIMG, 3, H, W)       : current sample;  x0_out / x_next_out same shape
 *   coefficients (host floats, computed by the caller in fp32 like utils/sampling.py:10-13):
 *     sqrt_1m_at = sqrt(1-abar_t), sqrt_at = sqrt(abar_t), sqrt_at_next, c2 = sqrt(1-abar_next)
 *   x0 = (x_t - eps*sqrt_1m_at)/sqrt_at ;  x_next = sqrt_at_next*x0 + c2*eps
 * Pixels covered by no patch get eps = 0/0 = NaN exactly like the reference's division. */
int wdm_ddim_update(wdm_handle* h, const float* eps, const int32_t* patches, int n, int p, const float* x_t,
                    int nimg, int H, int W, float sqrt_1m_at, float sqrt_at, float sqrt_at_next, float c2,
                    float* x0_out, float* x_next_out, void* stream);

/* wdm_ddim_update_eta: the same scatter-mean and x0, with the stochastic term of models/ddm_wavelet.py:500-502 (eta != 0; the reference's
 * own callers pass eta = 0, ddm_wavelet.py:302):  c1 = eta*sqrt((1 - at/at_next)(1 - at_next)/(1 - at)), c2 = sqrt((1 - at_next) - c1^2) -- both
 * computed by the caller in fp32 --  x_next = sqrt_at_next*x0 + c1*noise + c2*eps, summed left to right like the reference's expression.
 *   noise (NIMG, 3, H, W) f32 : the caller's randn_like(x) draw of this step */
int wdm_ddim_update_eta(wdm_handle* h, const float* eps, const int32_t* patches, int n, int p, const float* x_t,
                        int nimg, int H, int W, float sqrt_1m_at, float sqrt_at, float sqrt_at_next, float c1, float c2,
                        const float* noise, float* x0_out, float* x_next_out, void* stream);

/* Patch-sharded single image (no reference counterpart -- its eval is single-GPU): every rank runs the UNet
 * on ITS patches only.  wdm_patch_accumulate writes the rank's partial sums (NIMG*3*H*W floats) followed by its partial overlap
 * counts (same size) into acc_cnt, the caller all-reduces (sum) that ONE buffer over RCCL, and wdm_ddim_from_sums divides and
 * applies the DDIM update of wdm_ddim_update on every rank.  n = 0 writes zeros. */
int wdm_patch_accumulate(wdm_handle* h, const float* eps, const int32_t* patches, int n, int p, int nimg, int H, int W,
                         float* acc_cnt, void* stream);
int wdm_ddim_from_sums(wdm_handle* h, const float* acc_cnt, const float* x_t, int nimg, int H, int W, float sqrt_1m_at,
                       float sqrt_at, float sqrt_at_next, float c2, float* x0_out, float* x_next_out, void* stream);

/* NCHW f32 (B,C,H,W) -> NHWC dtype (B,H,W,C) and back (used by the drop-in model(x, t) call). */
int wdm_nchw_to_nhwc(wdm_handle* h, const float* src, void* dst, int B, int C, int H, int W, int dtype,
                     void* stream);
int wdm_nhwc_to_nchw(wdm_handle* h, const void* src, float* dst, int B, int C, int H, int W, int dtype,
                     void* stream);

/* ---- UNet ----------------------------------------------------------------------------------
 * Replaces DiffusionUNet.__init__/forward, models/unet.py:197-307, 346-395 (use_window and
 * wavelet_in_unet off, as in configs/raindrop_wavelet.yml). */
typedef struct wdm_unet_config {
    int ch;                  /* model.ch */
    int n_levels;            /* len(model.ch_mult), <= 8 */
    int ch_mult[8];
    int num_res_blocks;
    int n_attn_res;          /* len(model.attn_resolutions), <= 8 */
    int attn_resolutions[8];
    int in_channels;         /* UNet input channels (96 for raindrop_wavelet.yml, unet.py:212); any width >= 1: conv_in is zero-padded to a multiple of 32 */
    int out_ch;              /* 3 (12 / 48 with data.use_window / data.wavelet_in_unet) */
    int resolution;          /* data.image_size */
    int resamp_with_conv;    /* must be 1 */
    int dtype;               /* WDM_BF16 (throughput), WDM_F16 (throughput, fp16 operands), WDM_F32 (exact parity mode) or WDM_F32X3 (fast parity mode) */
} wdm_unet_config;

int wdm_unet_create(wdm_handle* h, const wdm_unet_config* cfg, wdm_unet** out);
int wdm_unet_destroy(wdm_unet* u);

/* state_dict enumeration: names/shapes are exactly the reference's state_dict keys */
int wdm_unet_num_params(const wdm_unet* u);
int wdm_unet_param_info(const wdm_unet* u, int i, const char** name, int* ndim, int64_t shape[4]);

/* packed weights live in ONE caller-allocated device buffer (so a rank-0 RCCL broadcast of that
 * buffer is the whole weight distribution step) */
size_t wdm_unet_packed_bytes(const wdm_unet* u);
int wdm_unet_set_packed(wdm_unet* u, void* packed, size_t bytes);
/* repack one fp32 parameter (device pointer, reference layout: conv OIHW, Linear [out,in]) into
 * the packed buffer, converting to the model dtype */
int wdm_unet_load_param(wdm_unet* u, const char* name, const float* dev_src, int64_t numel, void* stream);
/* mark the packed buffer as complete without per-parameter loads (after a broadcast) */
int wdm_unet_mark_loaded(wdm_unet* u);

size_t wdm_unet_workspace_bytes(const wdm_unet* u, int B);
/* x96: (B, R, R, in_channels) NHWC in the model dtype; t: n_t device floats, n_t in {1, B};
 * eps_out: (B, out_ch, R, R) NCHW f32 */
int wdm_unet_forward(wdm_unet* u, const void* x96, const float* t, int n_t, int B, float* eps_out,
                     void* workspace, size_t workspace_bytes, void* stream);
/* The timestep-dependent part of the network -- sinusoidal embedding, temb MLP and every ResnetBlock's temb_proj (models/unet.py:10-28, 225-230,
 * 354-357, 125) -- depends on t alone.  A sampler knows its whole timestep sequence up front (models/ddm_wavelet.py:296-297): wdm_unet_temb_table
 * computes the rows for all n timesteps in four launches, temb_out[n][wdm_unet_temb_rows(u)] (f32, device), and wdm_unet_forward_temb runs one UNet
 * call from one row of it (shared by the B images) -- the same bits as wdm_unet_forward with that timestep, four launches fewer per call.
 * The table is a function of the weights: rebuild it after a weight update.  workspace: the forward workspace serves. */
int wdm_unet_temb_rows(const wdm_unet* u);
int wdm_unet_temb_table(wdm_unet* u, const float* t, int n, float* temb_out, void* workspace, size_t workspace_bytes, void* stream);
int wdm_unet_forward_temb(wdm_unet* u, const void* x96, const float* temb_row, int B, float* eps_out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-block entry points (unit parity tests; same code the UNet executor runs) -----------
 * Weights are given in the reference layout as fp32 device pointers and packed on the fly into
 * `scratch`; x / y are NCHW f32 at this test boundary.  See wavedm_amd/csrc/api.hip. */
typedef struct wdm_resblock_params {  /* ResnetBlock, models/unet.py:81-138 */
    int cin, cout;
    const float *norm1_w, *norm1_b, *conv1_w, *conv1_b, *temb_w, *temb_b;
    const float *norm2_w, *norm2_b, *conv2_w, *conv2_b, *nin_w, *nin_b; /* nin_* NULL if cin==cout */
} wdm_resblock_params;
typedef struct wdm_attn_params {      /* AttnBlock, models/unet.py:141-193 */
    int c;
    const float *norm_w, *norm_b, *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *proj_w, *proj_b;
} wdm_attn_params;

/* x may be the channel concat of two tensors (x0: c0 channels, x1: c1 channels, c1 may be 0) */
int wdm_resblock_forward(wdm_handle* h, const wdm_resblock_params* p, const float* x0, int c0, const float* x1,
                         int c1, const float* temb /* (n_t, 512) raw temb, SiLU applied inside */, int n_t,
                         int temb_ch, int B, int H, int W, float* y, int dtype, void* scratch,
                         size_t scratch_bytes, void* stream);
int wdm_attn_forward(wdm_handle* h, const wdm_attn_params* p, const float* x, int B, int H, int W, float* y,
                     int dtype, void* scratch, size_t scratch_bytes, void* stream);
/* mode: 0 = conv3x3 s1 p1, 1 = Downsample (pad(0,1,0,1) + conv3x3 s2), 2 = Upsample (nearest x2 +
 * conv3x3 p1), 3 = conv1x1 */
int wdm_conv_forward(wdm_handle* h, const float* w, const float* b, int cin, int cout, int mode, const float* x,
                     int B, int H, int W, float* y, int dtype, void* scratch, size_t scratch_bytes,
                     void* stream);
/* timestep embedding + temb MLP (unet.py:10-28, 354-357): t (n_t) -> temb (n_t, 4*ch) */
int wdm_temb_forward(wdm_handle* h, const float* t, int n_t, int ch, const float* w0, const float* b0,
                     const float* w1, const float* b1, float* temb_out, void* scratch, size_t scratch_bytes,
                     void* stream);

/* ---- operators of the optional `data.global_attn` model (DiffusionUNet_Global / Attn_Global, models/unet.py:397-636) ----
 * The ResnetBlocks, AttnBlocks and 3x3 / 1x1 convolutions of that model run on the block entry points above (wavedm_amd/unet_global.py);
 * these are the operators only it has, as plain fp32 NCHW kernels (wavedm_amd/csrc/global_attn.hip).
 * wdm_conv2d_direct: torch.nn.Conv2d(Cin, Cout, k, stride, pad, groups) (weight [Cout][Cin/groups][k][k]) or, transposed != 0,
 *   torch.nn.ConvTranspose2d(Cin, Cout, k, stride, pad) (weight [Cin][Cout][k][k]) -- unet.py:407-424 (q, k, v), :522, :567.
 * wdm_groupnorm: GroupNorm(32, eps) [+ SiLU] (unet.py:36-37; Attn_Global applies norm_patch to both inputs, :433-434).
 * wdm_cross_attention: q (B,C,Nq), k and v (B,C,Nk), any Nk >= 1 -> out (B,C,Nq), softmax(C^-0.5 q^T k) over the keys (unet.py:438-455).
 * wdm_upsample_add: y = x + nearest_upsample(hp, scale) (unet.py:459-462). */
int wdm_conv2d_direct(wdm_handle* h, const float* x, const float* w, const float* bias, int B, int Cin, int H, int W, int Cout, int k,
                      int stride, int pad, int groups, int transposed, float* y, void* stream);
int wdm_groupnorm(wdm_handle* h, const float* x, const float* gamma, const float* beta, int B, int C, int H, int W, float eps, int silu,
                  float* y, void* stream);
int wdm_cross_attention(wdm_handle* h, const float* q, const float* k, const float* v, int B, int C, int Nq, int Nk, float* out,
                        void* stream);
int wdm_upsample_add(wdm_handle* h, const float* x, const float* hp, int B, int C, int H, int W, int scale, float* y, void* stream);

/* ---- HFRM -------------------------------------------------------------------
 * Replaces HFRM.__init__/forward, models/arch.py:206-253 (the module restoration.py:94 runs once per image to
 * produce the 45 "other" wavelet channels).  Same protocol as the UNet object: enumerate the state_dict keys,
 * give one caller-allocated buffer, load fp32 parameters, finalize (packs the GEMM weights, folds beta/gamma),
 * forward on NCHW f32 images whose H and W are multiples of 16. */
typedef struct wdm_hfrm wdm_hfrm;
typedef struct wdm_hfrm_config {
    int in_channel;          /* 3 */
    int dim;                 /* 32 */
    int mid_blk_num;         /* 6 */
    int n_enc;               /* 4 */
    int enc_blk_nums[8];     /* 2,2,2,4 (models/ddm_wavelet.py:139) */
    int n_dec;               /* 4 */
    int dec_blk_nums[8];     /* 2,2,2,2 */
    int dtype;
} wdm_hfrm_config;
int wdm_hfrm_create(wdm_handle* h, const wdm_hfrm_config* cfg, wdm_hfrm** out);
int wdm_hfrm_destroy(wdm_hfrm* m);
int wdm_hfrm_num_params(const wdm_hfrm* m);
int wdm_hfrm_param_info(const wdm_hfrm* m, int i, const char** name, int* ndim, int64_t shape[4]);
size_t wdm_hfrm_packed_bytes(const wdm_hfrm* m);
int wdm_hfrm_set_packed(wdm_hfrm* m, void* packed, size_t bytes);
int wdm_hfrm_load_param(wdm_hfrm* m, const char* name, const float* dev_src, int64_t numel, void* stream);
int wdm_hfrm_finalize(wdm_hfrm* m, void* stream);
size_t wdm_hfrm_workspace_bytes(const wdm_hfrm* m, int B, int H, int W);
int wdm_hfrm_forward(wdm_hfrm* m, const float* x, int B, int H, int W, float* y, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---- output side of DiffusiveRestoration.restore ---------------------------------
 * wdm_image_sqdiff: a, b (B,3,H,W) f32 on the device -> sums[B][2] (device, fp64):
 *   [0] = sum over 3 channels and pixels of (clamp01(a) - clamp01(b))^2   -> torchPSNR, utils/metrics.py:7-11
 *   [1] = sum over pixels of (Y(a) - Y(b))^2, Y = (24.966 c0 + 128.553 c1 + 65.481 c2 + 16)/255
 *                                                                          -> calculate_psnr_in_GPU(.., True), :30-51
 * wdm_to_u8_hwc: (B,C,H,W) f32 -> (B,H,W,C) u8 with torchvision.utils.save_image's rounding
 *   (x*255 + 0.5, clamp to [0,255], truncate), replacing utils/logging.py:9-12's device->host float copy. */
int wdm_image_sqdiff(wdm_handle* h, const float* a, const float* b, int B, int H, int W, double* sums, void* stream);
int wdm_to_u8_hwc(wdm_handle* h, const float* x, int B, int C, int H, int W, uint8_t* y, void* stream);

/* ---- training step: backward primitives, test entry points --------------------------
 * wdm_conv_backward: autograd of one convolution of models/unet.py (mode as wdm_conv_forward: 0 conv3x3 s1 p1, 1 Downsample,
 * 2 Upsample, 3 conv1x1).  x (B,cin,H,W), dy (B,cout,Ho,Wo), w OIHW f32 -> dx (B,cin,H,W) (optional), dw OIHW f32, db (cout) (optional). */
int wdm_conv_backward(wdm_handle* h, const float* w, int cin, int cout, int mode, const float* x, const float* dy, int B,
                      int H, int W, float* dx, float* dw, float* db, int dtype, void* scratch, size_t scratch_bytes,
                      void* stream);

/* wdm_gn_act_backward: autograd of GroupNorm(32, 1e-6)(+SiLU) (unet.py:31-37) over a channel concat [C0 | C - C0] of x (B,C,H,W);
 * dy (B,C,H,W) -> dx (B,C,H,W), dgamma (C), dbeta (C). */
int wdm_gn_act_backward(wdm_handle* h, const float* x, int C0, int C, const float* gamma, const float* beta, const float* dy,
                        int silu, int B, int H, int W, float* dx, float* dgamma, float* dbeta, int dtype, void* scratch,
                        size_t scratch_bytes, void* stream);

/* ---- training step object -----------------------------------------------------------
 * Replaces the body of DenoisingDiffusion_Wavelet.train's inner loop (models/ddm_wavelet.py:259-272) for the raindrop_wavelet.yml
 * branch: noise_estimation_loss (:108-124) forward + backward, torch.optim.Adam step (utils/optimize.py:5-8) and EMAHelper.update
 * (:48-53).  Parameters, gradients, Adam moments and the EMA shadow are five caller-allocated flat fp32 DEVICE buffers sharing one
 * layout (wdm_trainer_param_info gives name, shape and float offset of every state_dict entry; the temb_proj layers sit at the end as
 * one [rows][4ch] matrix).  A data-parallel job all-reduces the gradient buffer between _step and _adam_ema.
 *   x0 (B, in_channels, R, R) f32: [x_cond | x_tar | x_other] in the wavelet domain; e (B, out_ch, R, R) noise; t (B) timesteps as
 *   float; sqrt_a / sqrt_1ma (B): sqrt(abar_t), sqrt(1 - abar_t) (device); c_t0: first channel of x_tar.
 *   loss (device float) = mean_b sum (e - out)^2; out_nchw (optional): the network output. */
typedef struct wdm_trainer wdm_trainer;
int wdm_trainer_create(wdm_handle* h, const wdm_unet_config* cfg, wdm_trainer** out);
int wdm_trainer_destroy(wdm_trainer* t);
int wdm_trainer_num_params(const wdm_trainer* t);
int64_t wdm_trainer_num_floats(const wdm_trainer* t);
int wdm_trainer_param_info(const wdm_trainer* t, int i, const char** name, int* ndim, int64_t shape[4], int64_t* offset);
int wdm_trainer_set_buffers(wdm_trainer* t, float* params, float* grads, float* m, float* v, float* ema);
/* training.use_mse (ddm_wavelet.py:263-266): back-propagate mse_loss = mean_b sum (x_tar - x0_pred)^2 instead of the noise-space loss
 * (default 0).  *loss of wdm_trainer_step stays the noise-space value either way. */
int wdm_trainer_set_objective(wdm_trainer* t, int use_mse);
int wdm_trainer_step(wdm_trainer* t, const float* x0, const float* tt, const float* sqrt_a, const float* sqrt_1ma, const float* e,
                     int B, int c_t0, float* loss, float* out_nchw, void* workspace, size_t workspace_bytes, void* stream);
int wdm_trainer_adam_ema(wdm_trainer* t, int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                         float ema_mu, void* stream);
/* Gradient buckets for a data-parallel all-reduce that overlaps the backward (the reference wraps the model in DistributedDataParallel, ddm_wavelet.py:168,
 * which all-reduces 25 MB buckets while the backward runs).  The parameters sit in the flat buffers in forward order and the backward runs in reverse, so the
 * gradient buffer fills from its end: with n events set (hipEvent_t handles owned by the caller; n = 0 switches the feature off) every following
 * wdm_trainer_step cuts the filled range into at most n buckets of about equal size and records event k on the step's stream behind the last launch that
 * writes bucket k.  wdm_trainer_grad_buckets returns the bounds of the last step in float elements of the gradient buffer, descending:
 * bucket k = [bounds[k + 1], bounds[k]), 0 <= k < *n_buckets <= n (bounds needs room for n + 1 values).  What lies outside [bounds[n_buckets], bounds[0]) --
 * the timestep-embedding MLP and the temb_proj matrix -- is final only when the whole step is. */
int wdm_trainer_set_grad_events(wdm_trainer* t, void* const* events, int n);
int wdm_trainer_grad_buckets(const wdm_trainer* t, int64_t* bounds, int max_bounds, int* n_buckets);

/* ---- live kernel timing (bench.py roofline leg) ----------------------------------------------
 * While enabled, every convolution launch is bracketed by two HIP events on its own stream and
 * tagged with its algorithmic flops (2*M*N*K) and bytes (input + weights + output once).
 * wdm_prof_report synchronises those events, aggregates per kernel configuration and clears. */
typedef struct wdm_prof_entry {
    char kernel[96];
    long long launches;
    double total_ms, total_flops, total_bytes;
} wdm_prof_entry;
int wdm_prof_enable(int on);
int wdm_prof_report(wdm_prof_entry* out, int max_entries, int* n_entries);

/* ---- experiment switches ------------------------------------------------------------------------
 * The WDM_* environment variables (alternative kernels kept for A/B runs; defaults are the measured best) are read once, at
 * first use -- no launch path calls getenv.  A harness that changes them inside a running process calls this to have them read again.
 * (No counterpart in the reference: csrc/common.h EnvCfg.) */
int wdm_env_refresh(void);

/* ---- concurrent streams -------------------------------------------------------------------------
 * A caller that keeps n independent forward calls in flight on n HIP streams (wavedm_amd/sampling.py: chunks of independent crops) says so here: the tile
 * choices that follow the workgroup count of ONE launch ("256-column tiles where they still fill the chip") then count n launches side by side.  Those
 * alternatives write the same bits, so this changes speed only.  n = 1 (default): one launch owns the chip.  (No counterpart in the reference.)
 *
 * HAZARD, measured and not root-caused (EXPERIMENTS.md, round 3 "chunks of independent crops on separate HIP streams"): with four chunks on four streams the
 * sampler gains 0 ... +5 % when every pass takes streams it has not used before, and LOSES 35 % (126 -> 83 img/s, every box) when the same four stream
 * objects carry pass after pass -- also on every eighth pass of a caller that draws from a pool of 32.  GPU_MAX_HW_QUEUES = 1 / 2 are worse still, so the
 * mapping of streams onto the hardware queues is involved; what exactly is not understood.  The library itself creates no streams and keeps no per-stream
 * state (each call's scratch is the caller's workspace), so nothing here depends on which stream a call arrives on: the effect is in the runtime's queue
 * assignment.  Until it is understood, run ONE stream per device (the default everywhere in wavedm_amd; sampling.ddim_sample(streams=) is opt-in) and do
 * not cache side streams across passes. */
int wdm_set_concurrent_streams(int n);

#ifdef __cplusplus
}
#endif
#endif /* WAVEDM_H */
