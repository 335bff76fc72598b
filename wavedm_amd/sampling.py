"""Device-resident DDIM sampler with overlapping-patch stitching.

Mirrors `utils/sampling.py:10-13` (compute_alpha) and `models/ddm_wavelet.py:413-506`
(overlapping_grid_indices, generalized_steps_overlapping, eta = 0): per timestep the patches are
gathered straight into the NHWC UNet input, the UNet runs on all patches, and one kernel does the
scatter-add in corner order, the division by the overlap count and the DDIM update.  Unlike the
reference nothing leaves the GPU inside the loop (it does `.to('cpu')` twice and four `.item()`
syncs per step, ddm_wavelet.py:498-504)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """ddm_wavelet.py:87-105 (float64 numpy array; the caller casts to fp32 like ddm_wavelet.py:177)."""
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_diffusion_timesteps, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, num_diffusion_timesteps, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(num_diffusion_timesteps, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(num_diffusion_timesteps, 1, num_diffusion_timesteps, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        x = np.linspace(-6, 6, num_diffusion_timesteps)
        betas = 1 / (np.exp(-x) + 1) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (num_diffusion_timesteps,)
    return betas


DEFAULT_MAX_BATCH = 384      # patches per UNet call of the stitched sampler when args.max_batch is not set (workspace ~ 40 MB per patch: 15 GB of the 288)

_TRACE = None       # scripts/restore_trace.py: a callable(label) for host-side timeline marks inside the loop

_ABAR = {}          # id(betas tensor) -> (weak reference to it, its version counter, (abar table on the host, the betas' bytes))
_DEVCONST = {}      # small read-only device tensors the launches read through raw pointers (patch lists, timestep rows)


def _betas_entry(betas: torch.Tensor):
    """The host-side view of a betas tensor, read back ONCE per tensor object (and per in-place update of it): `betas.cpu()` is a blocking copy on the current
    stream, i.e. a wait for everything queued before it -- restore() keeps the next group of images queued behind the running one (restoration.py), so the
    sampler must not synchronise on entry.  Keyed by object identity, checked through a weak reference (an id can be reused after its tensor is gone)."""
    import weakref
    ent = _ABAR.get(id(betas))
    if ent is not None and ent[0]() is betas and ent[1] == betas._version:
        return ent[2]
    host = betas.detach().float().cpu()
    b = torch.cat([torch.zeros(1, dtype=torch.float32), host], dim=0)
    val = ((1 - b).cumprod(dim=0), host.numpy().tobytes())
    for k in [k for k, e in _ABAR.items() if e[0]() is None]:
        del _ABAR[k]
    _ABAR[id(betas)] = (weakref.ref(betas), betas._version, val)
    return val


def alpha_bar_table(betas: torch.Tensor) -> torch.Tensor:
    """fp32 table T[t+1] = abar(t), T[0] = abar(-1) = 1 -- the cumprod of utils/sampling.py:11-12, on the CPU."""
    return _betas_entry(betas)[0]


def _device_const(dev, kind, values, dtype):
    """A small constant on the device, uploaded once per distinct content: a pageable H2D copy waits for the stream (see _betas_entry), and the sampler is called
    with the same patch list / timestep sequence for every group of same-sized images.  The tensors are never written."""
    key = (str(dev), kind, values)
    t = _DEVCONST.get(key)
    if t is None:
        t = torch.tensor(list(values), dtype=dtype).to(dev)
        if len(_DEVCONST) >= 64:
            _DEVCONST.pop(next(iter(_DEVCONST)))
        _DEVCONST[key] = t
    return t


def compute_alpha(beta, t):
    """utils/sampling.py:10-13 (kept for API compatibility; t is a LongTensor)."""
    beta = torch.cat([torch.zeros(1).to(beta.device), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def overlapping_grid_indices(h, w, output_size, r=None):
    """ddm_wavelet.py:426-435 on plain ints."""
    r = 16 if r is None else r
    h_list = [i for i in range(0, h - output_size + 1, r)]
    w_list = [i for i in range(0, w - output_size + 1, r)]
    if h_list[-1] + output_size < h:
        h_list.append(h - output_size)
    if w_list[-1] + output_size < w:
        w_list.append(w - output_size)
    return h_list, w_list


def _chunk_streams(dev, n):
    """Side streams of the chunked sampler: fresh ones from torch's pool per call.
    Root cause of the "-35 % on reused streams" of rounds 3-4 (round 5, profiles/r05_streams_hw_queues.log): it is the HIP runtime's stream -> HARDWARE QUEUE
    mapping (GPU_MAX_HW_QUEUES, default 4), not the streams' age.  Whenever two chunks' streams sit on DIFFERENT hardware queues their kernels really run side by
    side -- and that is what costs 35 % (2 streams on 4 or 8 queues: 400-423 img/s against 615-619 on one stream; 4 streams on 8 queues 371; 8 on 16: 247): every
    workgroup of these kernels takes a whole CU's LDS, so two launches only split the CUs between them while each streams its own weights and halo tiles through
    the caches, and the tile rules that count one launch's workgroups per round no longer describe what a CU sees.  When the chunks' streams SHARE a queue (2 streams
    with GPU_MAX_HW_QUEUES=2: 627.6; 4 streams on the default 4 queues, which the main stream and torch's pool also use: 628.4) the launches serialise on it and the
    result is the single-stream rate +- 2 %.  So there is nothing to win here, and the mode stays opt-in."""
    return [torch.cuda.Stream(device=dev) for _ in range(n)]


# ---- the whole sampling loop as ONE hipGraph (OPT-IN: WAVEDM_GRAPH=1) -----------------------------------------------------------------------------------
# Back-to-back launches of an EMPTY kernel cost 3.4 ... 3.6 us each on a stream and 1.6 us replayed from a hipGraph (tools/launch_ubench.hip, round 5) -- which
# suggested ~4 % for the ~10 000 launches of a trajectory.  Built and measured (scripts/graph_probe.py, 64 crops x 20 steps, one process): 103.9 / 104.0 ms
# direct, 104.0 / 103.8 ms replayed -- nothing.  The 3.5 us was the HOST's submission rate, which real kernels (15 ... 200 us each) hide completely; the device-side
# boundary is the same either way.  What the graph does buy is a free host thread during sampling and launch-latency immunity at tiny batches, so it stays as
# an option: the loop is a fixed launch sequence for given shapes (no host decision depends on device data), captured once per (model buffers, shapes, timestep
# sequence, patch list) and replayed; inputs are copied into the graph's static tensors, kept results cloned out of its memory.  Same kernels, same arguments,
# same order: the same bits (tests/test_gpu_unet.py).  Profiling with per-launch events, several streams and the patch-sharded mode always launch directly.
_GRAPHS = None
_GRAPH_CACHE = int(os.environ.get("WAVEDM_GRAPH_CACHE", "6"))        # captured loops kept (least recently used first out): each holds its activations' memory pool


def _replay_graph(unet, run_loop, x, x_cond, x_other, n_run, key_tail, keepalive=()):
    global _GRAPHS
    from collections import OrderedDict
    if _GRAPHS is None:
        _GRAPHS = OrderedDict()
    dev = x.device
    packed = unet.pack_weights()
    key = (id(unet), packed.data_ptr(), unet._dtype_code, _lib.env_generation(), str(dev), tuple(x.shape), tuple(x_cond.shape),
           None if x_other is None else tuple(x_other.shape)) + tuple(key_tail)
    ent = _GRAPHS.get(key)
    if ent is None:
        xs_, xc_, xo_ = x.clone(), x_cond.clone(), (None if x_other is None else x_other.clone())
        # one step directly, on a side stream, before the capture: first-use initialisation (hipFuncSetAttribute per kernel, workspace and switch set-up) is not capturable
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run_loop(xs_, xc_, xo_, 1, False)
        torch.cuda.current_stream(dev).wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):      # (other threads -- the PNG writer's -- keep using the device meanwhile)
            out_xs, out_x0 = run_loop(xs_, xc_, xo_, n_run, False)
        # the entry keeps alive what the captured launches point at: the packed weights and the workspace of this moment (the model may replace either later)
        # ... and the device tensors of THIS call the launches read through raw pointers (patch list, timesteps): a later call makes its own
        ent = (g, xs_, xc_, xo_, out_xs, out_x0, packed, dict(unet._ws), run_loop, tuple(keepalive))
        _GRAPHS[key] = ent
        while len(_GRAPHS) > max(1, _GRAPH_CACHE):
            _GRAPHS.popitem(last=False)
    else:
        _GRAPHS.move_to_end(key)
        g, xs_, xc_, xo_, out_xs, out_x0 = ent[:6]
        xs_.copy_(x)
        xc_.copy_(x_cond)
        if xo_ is not None:
            xo_.copy_(x_other)
    g.replay()
    xs = [x] + [None if t is None else t.clone() for t in out_xs[1:]]          # out of the graph's memory: the next replay overwrites it
    x0_preds = [None if t is None else t.clone() for t in out_x0]
    return xs, x0_preds


def graph_cache_clear():
    """Drop every captured sampling loop (and the memory its activations live in)."""
    global _GRAPHS
    _GRAPHS = None


def ddim_sample(unet, x, x_cond, x_other, seq, betas, corners=None, p_size=None, max_batch=64, keep="all", stop_at=None,
                patch_group=None, streams=None, eta=0.0):
    """DDIM over `seq` (ascending list of timesteps) for NIMG images; eta = 0 (what every caller in the reference passes) is the deterministic sampler.

    x (NIMG,3,H,W) start noise, x_cond (NIMG,48,H,W), x_other (NIMG,45,H,W): fp32 on the GPU.
    corners: None -> every image is one p x p patch at (0,0) (p == H == W; the batched 64x64 case),
             or a list of (hi, wi) applied to EVERY image of the batch, like the reference's crops
             `x_cond[:, :, hi:hi+p, wi:wi+p]` (ddm_wavelet.py:467-478) -- image-major patch order, so each image's
             overlap sums run in the same order as when it is restored alone,
             or an explicit list of (img, hi, wi).
    keep: "all" -> (xs, x0_preds) lists as the reference returns; or a set of negative indices into
          x0_preds / xs to retain, e.g. {-5, -1} (saves nothing but list bookkeeping).
    stop_at: opt-in early stop: a negative index k means "x0_preds[k] is all the caller needs", so the
          |k|-1 steps after it -- which the reference computes and discards (restoration.py:108 uses [-5]) -- are skipped;
          the lists are padded with None so indices keep their meaning.  Default None = run every step like the reference.
    streams: OPT-IN (default WAVEDM_STREAMS or 1).  Independent crops (corners=None) are split into this many chunks, each walking its whole trajectory on a
          HIP stream of its own (own UNet workspace), so that one chunk's kernels can fill the launch boundaries and tails of the others'.  Per-image results do
          not depend on the batch an image sits in: the same bits as one stream (tests/test_gpu_unet.py).  Measured: +-2 % when the chunks' streams share a
          hardware queue, -35 % when they do not (_chunk_streams: kernels that take a whole CU each do not gain from running side by side).  Patch lists (overlap
          sums couple an image's patches every step) and per-launch profiling always run on one stream.
    eta: the stochastic term of ddm_wavelet.py:500-502: c1 = eta*sqrt((1 - at/at_next)(1 - at_next)/(1 - at)), c2 = sqrt((1 - at_next) - c1^2), and
          x_next = sqrt(at_next)*x0 + c1*randn_like(x) + c2*et -- one draw of x's shape per step from torch's device generator, like the reference.
          Stochastic runs launch directly (no captured graph), on one stream, and not patch-sharded (every rank would need the same draw).
    patch_group: a torch.distributed process group (or True for the default group) = patch-sharded latency mode
          (SURVEY.md §8e-ii): the patch list is split contiguously over the ranks, each rank runs the UNet on its patches,
          and ONE all-reduce(sum) per step (RCCL) combines partial sums and overlap counts before the DDIM update, which every
          rank then applies identically.  All ranks must pass the same x / x_cond / x_other.  The fp32 sum order differs from
          the sequential scatter by at most the association of the per-rank partial sums.
    """
    x = _lib.require_cuda_f32(x, "x")
    x_cond = _lib.require_cuda_f32(x_cond, "x_cond")
    if x_other is not None:
        x_other = _lib.require_cuda_f32(x_other, "x_other")
    dev = x.device
    L, h = _lib.lib(), _lib.handle(dev.index or 0)
    if hasattr(unet, "pack_weights"):
        unet.pack_weights()                   # before the compute dtype is read: the automatic f16 mode may fall back while it packs (DiffusionUNet.pack_weights)
    nimg, pc, H, W = x.shape
    ncond, nother = x_cond.shape[1], (x_other.shape[1] if x_other is not None else 0)      # x_other None: model.use_other_channels False
    cin = unet.in_channels
    assert ncond + pc + nother == cin, f"channel split {ncond}+{pc}+{nother} != UNet in_channels {cin}"
    if pc != 3:        # wdm_ddim_update / wdm_patch_accumulate / wdm_ddim_from_sums scatter exactly 3 prediction channels per patch
        raise NotImplementedError(f"ddim_sample: model.pred_channels = {pc}; the DDIM update kernels are built for 3 (raindrop_wavelet.yml)")
    with torch.cuda.device(dev):
        if corners is None:
            p = H
            assert H == W == unet.resolution
            n, patches, pptr = nimg, None, None
        else:
            p = int(p_size)
            if all(len(c) == 2 for c in corners):
                tri = [(im, int(c[0]), int(c[1])) for im in range(nimg) for c in corners]
            else:
                tri = [tuple(int(v) for v in c) for c in corners]
            for (im, hi, wi) in tri:
                if not (0 <= im < nimg and 0 <= hi and hi + p <= H and 0 <= wi and wi + p <= W):
                    raise ValueError(f"patch {(im, hi, wi)} of size {p} outside the {nimg}x{H}x{W} image")
            sharded = patch_group is not None
            if sharded:
                import torch.distributed as dist
                from .parallel import shard_range
                grp = None if patch_group is True else patch_group
                lo, hi = shard_range(len(tri), dist.get_rank(grp), dist.get_world_size(grp))
                tri = tri[lo:hi]
            n = len(tri)
            patches = _device_const(dev, "patches", tuple(tri) if tri else ((0, 0, 0),), torch.int32)
            pptr = _lib.ptr(patches)
        assert p == unet.resolution, "patch size must equal config.data.image_size (unet.py:351)"
        sharded = corners is not None and patch_group is not None
        if patch_group is not None and corners is None:
            raise ValueError("patch_group needs a corner list: independent crops shard by image (parallel.restore_sharded)")
        seq = list(seq)
        seq_next = [-1] + seq[:-1]
        abar = alpha_bar_table(betas)
        t_dev = _device_const(dev, "timesteps", tuple(float(v) for v in reversed(seq)), torch.float32)
        n_run = len(seq) if stop_at is None else len(seq) + int(stop_at) + 1
        assert 1 <= n_run <= len(seq), f"stop_at={stop_at} out of range for {len(seq)} steps"
        keep_set = None if keep == "all" else frozenset(int(v) for v in keep)
        eta = float(eta)
        if eta != 0.0 and sharded:
            raise NotImplementedError("ddim_sample: eta != 0 with patch_group (every rank would have to draw the same noise)")
        ns = int(os.environ.get("WAVEDM_STREAMS", "1")) if streams is None else int(streams)
        multi = corners is None and not sharded and ns > 1 and n >= 2 * ns and not _lib.prof_on() and eta == 0.0
        # UNet calls per step: ceil(n / max_batch) calls of (nearly) EQUAL size -- 360 patches under a cap of 128 run as 3 x 120, not 128 + 128 + 104
        # (per-image results do not depend on the batch an image sits in, tests/test_gpu_unet.py)
        n_calls = max(1, -(-n // max_batch))
        call_b = -(-n // n_calls) if os.environ.get("WAVEDM_EVEN_CALLS", "1") != "0" else max_batch

        def run_loop(x, x_cond, x_other, n_steps, use_chunks):
            """The sampling loop proper: every launch goes to the CURRENT stream (or the chunks' streams) -- also the body a hipGraph is captured from."""
            st = _lib.stream_ptr()
            x96 = torch.empty(max(n, 1), p, p, cin, device=dev, dtype=unet._torch_dtype)
            if n:
                _lib.check(L.wdm_pack_channels(h, _lib.ptr(x_cond), ncond, H, W, pptr, n, p, _lib.ptr(x96), cin, 0, unet._dtype_code, st))
                if nother:
                    _lib.check(L.wdm_pack_channels(h, _lib.ptr(x_other), nother, H, W, pptr, n, p, _lib.ptr(x96), cin, ncond + pc, unet._dtype_code, st))
            eps = torch.empty(max(n, 1), pc, p, p, device=dev, dtype=torch.float32)
            acc_cnt = torch.empty(2 * x.numel(), device=dev, dtype=torch.float32) if sharded else None
            # the timestep-dependent part of the UNet (embedding MLP, every temb_proj) for the WHOLE sequence at once: four launches per run instead of per step
            temb = unet.temb_table(t_dev, B=min(max(n, 1), call_b)) if os.environ.get("WAVEDM_TEMB_TABLE", "1") != "0" else None
            S = len(seq)
            xs, x0_preds = [x], []
            xt = x
            # chunks of independent crops, one HIP stream each (see `streams`)
            chunks = None
            if use_chunks:
                per = -(-n // ns)
                main = torch.cuda.current_stream()
                chunks = []
                pool = _chunk_streams(dev, ns)
                for ci, lo in enumerate(range(0, n, per)):
                    sc = pool[ci]
                    sc.wait_stream(main)                                  # inputs, x96's constant channels and the temb table are ready
                    chunks.append((lo, min(lo + per, n), sc))
                for ci, (lo, hi, _) in enumerate(chunks):                # the chunks' workspaces, allocated HERE, on the caller's stream, each for the largest
                    unet.workspace(min(max_batch, hi - lo), dev, ci)     # call its slot will run (smaller calls reuse it: DiffusionUNet.workspace)
            if chunks is not None:
                _lib.set_concurrent_streams(len(chunks))                  # tile rules that count one launch's workgroups count the chunks' together (whole loop)
            try:
                for k, (i_t, j_t) in enumerate(zip(reversed(seq), reversed(seq_next))):
                    if _TRACE is not None:
                        _TRACE(f"sampler: step {k}")
                    if k >= n_steps:
                        x0_preds.append(None)
                        xs.append(None)
                        continue
                    at, at_next = abar[i_t + 1], abar[j_t + 1]                       # fp32 scalars, as compute_alpha returns
                    s1m, sa = float((1 - at).sqrt()), float(at.sqrt())
                    san, c2 = float(at_next.sqrt()), float((1 - at_next).sqrt())      # c1 = 0 (eta = 0)
                    noise, c1 = None, 0.0
                    if eta != 0.0:                                                    # ddm_wavelet.py:500-501, in fp32 tensors like the reference
                        c1t = eta * ((1 - at / at_next) * (1 - at_next) / (1 - at)).sqrt()
                        c1, c2 = float(c1t), float(((1 - at_next) - c1t ** 2).sqrt())
                        noise = torch.randn_like(x)
                    x0 = torch.empty_like(x)
                    xn = torch.empty_like(x)
                    if chunks is not None:
                        # independent crops: every chunk's step on its own stream (same kernels, same per-image bits)
                        for ci, (lo, hi, sc) in enumerate(chunks):
                            with torch.cuda.stream(sc):
                                stc = sc.cuda_stream
                                _lib.check(L.wdm_pack_channels(h, _lib.ptr(xt[lo:hi]), pc, H, W, None, hi - lo, p, _lib.ptr(x96[lo:hi]), cin, ncond, unet._dtype_code, stc))
                                for i in range(lo, hi, max_batch):
                                    j = min(i + max_batch, hi)
                                    unet.forward_nhwc(x96[i:j], t_dev[k:k + 1], eps[i:j], temb_row=None if temb is None else temb[k], ws_slot=ci)
                                _lib.check(L.wdm_ddim_update(h, _lib.ptr(eps[lo:hi]), None, hi - lo, p, _lib.ptr(xt[lo:hi]), hi - lo, H, W, s1m, sa, san, c2,
                                                             _lib.ptr(x0[lo:hi]), _lib.ptr(xn[lo:hi]), stc))
                    else:
                        if n:
                            _lib.check(L.wdm_pack_channels(h, _lib.ptr(xt), pc, H, W, pptr, n, p, _lib.ptr(x96), cin, ncond, unet._dtype_code, st))
                        for i in range(0, n, call_b):
                            unet.forward_nhwc(x96[i:i + call_b], t_dev[k:k + 1], eps[i:i + call_b], temb_row=None if temb is None else temb[k])
                        if sharded:
                            _lib.check(L.wdm_patch_accumulate(h, _lib.ptr(eps), pptr, n, p, nimg, H, W, _lib.ptr(acc_cnt), st))
                            dist.all_reduce(acc_cnt, op=dist.ReduceOp.SUM, group=grp)
                            _lib.check(L.wdm_ddim_from_sums(h, _lib.ptr(acc_cnt), _lib.ptr(xt), nimg, H, W, s1m, sa, san, c2, _lib.ptr(x0), _lib.ptr(xn), st))
                        elif noise is not None:
                            _lib.check(L.wdm_ddim_update_eta(h, _lib.ptr(eps), pptr, n, p, _lib.ptr(xt), nimg, H, W, s1m, sa, san, c1, c2, _lib.ptr(noise),
                                                             _lib.ptr(x0), _lib.ptr(xn), st))
                        else:
                            _lib.check(L.wdm_ddim_update(h, _lib.ptr(eps), pptr, n, p, _lib.ptr(xt), nimg, H, W, s1m, sa, san, c2,
                                                         _lib.ptr(x0), _lib.ptr(xn), st))
                    # lists as the reference returns them; with `keep` given, what nobody asked for is dropped at once (inside a captured graph its memory is reused)
                    x0_preds.append(x0 if keep_set is None or (k - S) in keep_set else None)
                    xs.append(xn)
                    if keep_set is not None and len(xs) >= 3 and chunks is None:     # (with side streams still reading them the tensors stay until the loop's join)
                        xs[-2] = xs[-2] if (len(xs) - 2 - (S + 1)) in keep_set else None
                    xt = xn
            finally:
                if chunks is not None:
                    # also on an exception: x96, eps, xn and the temb table are released on the caller's stream, which must not happen while a side
                    # stream's kernels may still touch them
                    _lib.set_concurrent_streams(1)
                    for (_, _, sc) in chunks:
                        torch.cuda.current_stream().wait_stream(sc)        # the caller's stream sees every chunk's results
                    if keep_set is not None:                               # ... and only now lets go of what nobody asked for
                        for q in range(1, len(xs) - 1):
                            if (q - (S + 1)) not in keep_set:
                                xs[q] = None
            return xs, x0_preds

        graphed = (os.environ.get("WAVEDM_GRAPH", "0") == "1" and not sharded and not multi and n > 0 and not _lib.prof_on() and eta == 0.0 and
                   not torch.cuda.is_current_stream_capturing())
        if graphed:
            # everything the captured launch sequence depends on is in the key: shapes and the model's buffers (in _replay_graph), the timestep
            # sequence, the patch list, the call size, what is kept, the schedule's bytes (cached per betas tensor: no read-back per call), the temb-table switch
            xs, x0_preds = _replay_graph(unet, run_loop, x, x_cond, x_other, n_run,
                                         (tuple(seq), None if corners is None else tuple(map(tuple, tri)), p, max_batch, call_b, keep_set, n_run, _betas_entry(betas)[1],
                                          os.environ.get("WAVEDM_TEMB_TABLE", "1"), str(x_cond.dtype), None if x_other is None else str(x_other.dtype)),
                                         keepalive=(patches, t_dev))
        else:
            xs, x0_preds = run_loop(x, x_cond, x_other, n_run, multi)
        if keep != "all":
            S = len(x0_preds)
            x0_preds = [t if (i - S) in keep else None for i, t in enumerate(x0_preds)]
        return xs, x0_preds
