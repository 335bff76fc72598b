"""ctypes binding of libwavedm_hip.so (include/wavedm.h).

The HIP library is THE product path: nothing in this package falls back to PyTorch or to the CPU
oracle.  If the shared object is missing `lib()` raises with the build command."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WAVEDM_LIB") or os.path.join(_HERE, "csrc", "libwavedm_hip.so")     # WAVEDM_LIB: another build of the same ABI (A/B runs)

WDM_F32, WDM_BF16, WDM_F32X3, WDM_F16 = 0, 1, 2, 3
WDM_OK, WDM_EINVAL, WDM_ENOMEM, WDM_EHIP, WDM_ESTATE, WDM_ENOTFOUND = 0, -1, -2, -3, -4, -5
DTYPES = {"f32": WDM_F32, "fp32": WDM_F32, "float32": WDM_F32, "bf16": WDM_BF16, "bfloat16": WDM_BF16, "f32x3": WDM_F32X3, "f16": WDM_F16, "fp16": WDM_F16, "float16": WDM_F16, "half": WDM_F16}

_lib = None
_handles = {}


class UNetConfig(C.Structure):
    _fields_ = [("ch", C.c_int), ("n_levels", C.c_int), ("ch_mult", C.c_int * 8), ("num_res_blocks", C.c_int),
                ("n_attn_res", C.c_int), ("attn_resolutions", C.c_int * 8), ("in_channels", C.c_int),
                ("out_ch", C.c_int), ("resolution", C.c_int), ("resamp_with_conv", C.c_int), ("dtype", C.c_int)]


class HFRMConfig(C.Structure):
    _fields_ = [("in_channel", C.c_int), ("dim", C.c_int), ("mid_blk_num", C.c_int), ("n_enc", C.c_int),
                ("enc_blk_nums", C.c_int * 8), ("n_dec", C.c_int), ("dec_blk_nums", C.c_int * 8), ("dtype", C.c_int)]


class ResblockParams(C.Structure):
    _fields_ = [("cin", C.c_int), ("cout", C.c_int)] + [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "conv1_w", "conv1_b", "temb_w", "temb_b", "norm2_w", "norm2_b", "conv2_w", "conv2_b",
        "nin_w", "nin_b")]


class ProfEntry(C.Structure):
    _fields_ = [("kernel", C.c_char * 96), ("launches", C.c_longlong), ("total_ms", C.c_double),
                ("total_flops", C.c_double), ("total_bytes", C.c_double)]


class AttnParams(C.Structure):
    _fields_ = [("c", C.c_int)] + [(n, C.c_void_p) for n in (
        "norm_w", "norm_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "proj_w", "proj_b")]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"wavedm_amd: HIP library not built ({LIB_PATH} missing). Build it with "
            f"`make -C {os.path.join(_HERE, 'csrc')}` or `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU / PyTorch fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i, f, sz, i64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64
    sig = {
        "wdm_abi_version": (i, []),
        "wdm_last_error": (C.c_char_p, []),
        "wdm_create": (i, [i, C.POINTER(vp)]),
        "wdm_destroy": (i, [vp]),
        "wdm_dwt_fwd": (i, [vp, vp, vp, i, i, i, vp]),
        "wdm_dwt_inv": (i, [vp, vp, vp, i, i, i, vp]),
        "wdm_pack_channels": (i, [vp, vp, i, i, i, vp, i, i, vp, i, i, i, vp]),
        "wdm_ddim_update": (i, [vp, vp, vp, i, i, vp, i, i, i, f, f, f, f, vp, vp, vp]),
        "wdm_ddim_update_eta": (i, [vp, vp, vp, i, i, vp, i, i, i, f, f, f, f, f, vp, vp, vp, vp]),
        "wdm_patch_accumulate": (i, [vp, vp, vp, i, i, i, i, i, vp, vp]),
        "wdm_ddim_from_sums": (i, [vp, vp, vp, i, i, i, f, f, f, f, vp, vp, vp]),
        "wdm_nchw_to_nhwc": (i, [vp, vp, vp, i, i, i, i, i, vp]),
        "wdm_nhwc_to_nchw": (i, [vp, vp, vp, i, i, i, i, i, vp]),
        "wdm_unet_create": (i, [vp, C.POINTER(UNetConfig), C.POINTER(vp)]),
        "wdm_unet_destroy": (i, [vp]),
        "wdm_unet_num_params": (i, [vp]),
        "wdm_unet_param_info": (i, [vp, i, C.POINTER(C.c_char_p), C.POINTER(i), C.POINTER(i64 * 4)]),
        "wdm_unet_packed_bytes": (sz, [vp]),
        "wdm_unet_set_packed": (i, [vp, vp, sz]),
        "wdm_unet_load_param": (i, [vp, C.c_char_p, vp, i64, vp]),
        "wdm_unet_mark_loaded": (i, [vp]),
        "wdm_unet_workspace_bytes": (sz, [vp, i]),
        "wdm_unet_forward": (i, [vp, vp, vp, i, i, vp, vp, sz, vp]),
        "wdm_unet_temb_rows": (i, [vp]),
        "wdm_unet_temb_table": (i, [vp, vp, i, vp, vp, sz, vp]),
        "wdm_unet_forward_temb": (i, [vp, vp, vp, i, vp, vp, sz, vp]),
        "wdm_resblock_forward": (i, [vp, C.POINTER(ResblockParams), vp, i, vp, i, vp, i, i, i, i, i, vp, i, vp, sz, vp]),
        "wdm_attn_forward": (i, [vp, C.POINTER(AttnParams), vp, i, i, i, vp, i, vp, sz, vp]),
        "wdm_conv_forward": (i, [vp, vp, vp, i, i, i, vp, i, i, i, vp, i, vp, sz, vp]),
        "wdm_temb_forward": (i, [vp, vp, i, i, vp, vp, vp, vp, vp, vp, sz, vp]),
        "wdm_hfrm_create": (i, [vp, C.POINTER(HFRMConfig), C.POINTER(vp)]),
        "wdm_hfrm_destroy": (i, [vp]),
        "wdm_hfrm_num_params": (i, [vp]),
        "wdm_hfrm_param_info": (i, [vp, i, C.POINTER(C.c_char_p), C.POINTER(i), C.POINTER(i64 * 4)]),
        "wdm_hfrm_packed_bytes": (sz, [vp]),
        "wdm_hfrm_set_packed": (i, [vp, vp, sz]),
        "wdm_hfrm_load_param": (i, [vp, C.c_char_p, vp, i64, vp]),
        "wdm_hfrm_finalize": (i, [vp, vp]),
        "wdm_hfrm_workspace_bytes": (sz, [vp, i, i, i]),
        "wdm_hfrm_forward": (i, [vp, vp, i, i, i, vp, vp, sz, vp]),
        "wdm_image_sqdiff": (i, [vp, vp, vp, i, i, i, vp, vp]),
        "wdm_to_u8_hwc": (i, [vp, vp, i, i, i, i, vp, vp]),
        "wdm_conv_backward": (i, [vp, vp, i, i, i, vp, vp, i, i, i, vp, vp, vp, i, vp, sz, vp]),
        "wdm_gn_act_backward": (i, [vp, vp, i, i, vp, vp, vp, i, i, i, i, vp, vp, vp, i, vp, sz, vp]),
        "wdm_trainer_create": (i, [vp, C.POINTER(UNetConfig), C.POINTER(vp)]),
        "wdm_trainer_destroy": (i, [vp]),
        "wdm_trainer_num_params": (i, [vp]),
        "wdm_trainer_num_floats": (i64, [vp]),
        "wdm_trainer_param_info": (i, [vp, i, C.POINTER(C.c_char_p), C.POINTER(i), C.POINTER(i64 * 4), C.POINTER(i64)]),
        "wdm_trainer_set_buffers": (i, [vp, vp, vp, vp, vp, vp]),
        "wdm_trainer_set_objective": (i, [vp, i]),
        "wdm_trainer_step": (i, [vp, vp, vp, vp, vp, vp, i, i, vp, vp, vp, sz, vp]),
        "wdm_trainer_adam_ema": (i, [vp, i64, f, f, f, f, f, f, vp]),
        "wdm_trainer_set_grad_events": (i, [vp, C.POINTER(vp), i]),
        "wdm_trainer_grad_buckets": (i, [vp, C.POINTER(i64), i, C.POINTER(i)]),
        "wdm_dwt_fwd_affine": (i, [vp, vp, f, f, vp, i, i, i, vp]),
        "wdm_dwt_inv_compose": (i, [vp, vp, i, i, vp, vp, i, i, i, i, vp]),
        "wdm_conv2d_direct": (i, [vp, vp, vp, vp, i, i, i, i, i, i, i, i, i, i, vp, vp]),
        "wdm_groupnorm": (i, [vp, vp, vp, vp, i, i, i, i, f, i, vp, vp]),
        "wdm_cross_attention": (i, [vp, vp, vp, vp, i, i, i, i, vp, vp]),
        "wdm_upsample_add": (i, [vp, vp, vp, i, i, i, i, i, vp, vp]),
        "wdm_prof_enable": (i, [i]),
        "wdm_env_refresh": (i, []),
        "wdm_set_concurrent_streams": (i, [i]),
        "wdm_prof_report": (i, [C.POINTER(ProfEntry), i, C.POINTER(i)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here = header / library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    if L.wdm_abi_version() != 1:
        raise RuntimeError("wavedm_amd: libwavedm_hip.so ABI version mismatch")
    _lib = L
    return L


EXPORTED = ["wdm_abi_version", "wdm_last_error", "wdm_create", "wdm_destroy", "wdm_dwt_fwd", "wdm_dwt_inv",
            "wdm_pack_channels", "wdm_ddim_update", "wdm_ddim_update_eta", "wdm_patch_accumulate", "wdm_ddim_from_sums", "wdm_nchw_to_nhwc", "wdm_nhwc_to_nchw", "wdm_unet_create",
            "wdm_unet_destroy", "wdm_unet_num_params", "wdm_unet_param_info", "wdm_unet_packed_bytes",
            "wdm_unet_set_packed", "wdm_unet_load_param", "wdm_unet_mark_loaded", "wdm_unet_workspace_bytes",
            "wdm_unet_forward", "wdm_unet_temb_rows", "wdm_unet_temb_table", "wdm_unet_forward_temb", "wdm_resblock_forward", "wdm_attn_forward", "wdm_conv_forward", "wdm_temb_forward",
            "wdm_hfrm_create", "wdm_hfrm_destroy", "wdm_hfrm_num_params", "wdm_hfrm_param_info", "wdm_hfrm_packed_bytes",
            "wdm_hfrm_set_packed", "wdm_hfrm_load_param", "wdm_hfrm_finalize", "wdm_hfrm_workspace_bytes",
            "wdm_hfrm_forward", "wdm_image_sqdiff", "wdm_to_u8_hwc", "wdm_conv_backward", "wdm_gn_act_backward", "wdm_trainer_create", "wdm_trainer_destroy", "wdm_trainer_num_params",
            "wdm_trainer_num_floats", "wdm_trainer_param_info", "wdm_trainer_set_buffers", "wdm_trainer_set_objective", "wdm_trainer_step", "wdm_trainer_adam_ema", "wdm_trainer_set_grad_events", "wdm_trainer_grad_buckets", "wdm_dwt_fwd_affine", "wdm_dwt_inv_compose", "wdm_conv2d_direct", "wdm_groupnorm", "wdm_cross_attention", "wdm_upsample_add",
            "wdm_prof_enable", "wdm_prof_report", "wdm_env_refresh", "wdm_set_concurrent_streams"]


_PROF_ON = False


def prof_enable(on: bool):
    global _PROF_ON
    check(lib().wdm_prof_enable(1 if on else 0))
    _PROF_ON = bool(on)


def prof_on() -> bool:
    """Per-launch timing is on (events around every launch on the launch stream): samplers then keep to one stream."""
    return _PROF_ON


def set_concurrent_streams(n: int):
    """Tell the library how many forward calls the caller keeps in flight side by side (include/wavedm.h)."""
    check(lib().wdm_set_concurrent_streams(int(n)))


_ENV_GEN = 0


def env_refresh():
    """Have the library re-read its WDM_* experiment switches (they are read once, at first use).  Switches change the size of the activation arena,
    so every cached workspace is stale afterwards: the generation counter makes DiffusionUNet.workspace() re-query and re-allocate."""
    global _ENV_GEN
    check(lib().wdm_env_refresh())
    _ENV_GEN += 1


def env_generation() -> int:
    return _ENV_GEN


def prof_report():
    """-> list of dicts {kernel, launches, ms, flops, bytes} aggregated since the last report."""
    arr = (ProfEntry * 512)()
    n = C.c_int()
    check(lib().wdm_prof_report(arr, 512, C.byref(n)))
    return [dict(kernel=arr[k].kernel.decode(), launches=int(arr[k].launches), ms=float(arr[k].total_ms),
                 flops=float(arr[k].total_flops), bytes=float(arr[k].total_bytes)) for k in range(n.value)]


def check(rc: int):
    if rc != 0:
        raise RuntimeError(f"libwavedm_hip: error {rc}: {lib().wdm_last_error().decode(errors='replace')}")


def handle(device_index: int):
    """One wdm_handle per device (created on first use)."""
    if device_index not in _handles:
        h = C.c_void_p()
        check(lib().wdm_create(int(device_index), C.byref(h)))
        _handles[device_index] = h
    return _handles[device_index]


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr())


def require_cuda_f32(t, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{name}: expected a float32 tensor on the GPU (got {getattr(t, 'dtype', type(t))} on "
                        f"{getattr(t, 'device', '?')}); wavedm_amd has no CPU path")
    return t.contiguous()


_LIBC = None


def pinned_dontfork(t):
    """madvise(MADV_DONTFORK) on a pinned host tensor's bytes; returns t.  Why (round 6, scripts/fork_probe.py, profiles/r06_fork_probe.log): when a process that holds a HIP
    context fork()s -- a DataLoader starting its workers -- its GPU queue does not run new work until the driver has sorted out every page of PINNED host memory the fork
    made copy-on-write: ~21 ms per MB (0.2 s for a bare process, 5.7 s with 512 MB pinned, on an MI355X box).  Pinned staging buffers are of no use to a child; kept out of
    fork() they cost nothing.  Page-granular and idempotent; a failure (non-Linux libc, odd mapping) is ignored -- it only costs the stall back."""
    global _LIBC
    try:
        if t is None or not t.is_pinned():
            return t
        if _LIBC is None:
            _LIBC = C.CDLL("libc.so.6", use_errno=True)
            _LIBC.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
            _LIBC.madvise.restype = C.c_int
        a0 = t.data_ptr() & ~4095
        n = ((t.data_ptr() + t.numel() * t.element_size() + 4095) & ~4095) - a0
        if n > 0:
            _LIBC.madvise(a0, n, 10)                    # MADV_DONTFORK
    except Exception:                                   # noqa: BLE001
        pass
    return t
