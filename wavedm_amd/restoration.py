"""DiffusiveRestoration -- the reference's evaluation wrapper (`models/restoration.py:16-196`).

`restore(val_loader, validation, r)` consumes the loader contract `(x[B,6,H,W] in [0,1], img_id, total)`: DWT of the
degraded image, HFRM -> DWT -> `x_other`, stitched DDIM sampling, `x0_preds[-5]` (restoration.py:108), concat with the
HFRM high-frequency bands, IDWT, clamp, PSNR, PNG dumps.  Everything between the H2D copy of the batch and the PSNR
numbers stays on the GPU:

* metrics: one device reduction per image pair (imageio.sqdiff) gives the three PSNRs the reference prints
  (torchPSNR on a CPU copy, calculate_psnr_in_GPU, and the numpy calculate_psnr after two float D2H copies);
* PNGs: quantised on the device, copied on a side stream, encoded by a worker thread (imageio.AsyncImageWriter) --
  `save_images=False` switches them off.

What makes this call surface as fast as the sampler underneath it -- all of it bit-identical per image to the plain one-image-at-a-time loop
(tests/test_gpu_io.py):

* `args.early_stop` (default True HERE; `sample_image` called directly runs every step): restore() reads `x0_preds[-5]` and nothing behind it
  (restoration.py:108), so the four steps the reference computes and throws away are not run -- 4 of 25 steps at the reference's default
  `--sampling_timesteps 25` (eval_diffusion.py:26);
* `args.images_per_call` (default "auto"): consecutive loader items of the same size are restored in ONE sampler call, so a 480x720 image's 45 patches
  per step become 45 x N and the UNet calls fill up (auto: the smallest N whose patches fill whole 64-patch units of the calls to >= 97 %: 7 images at
  45 patches under the default cap of 384 patches per UNet call -- images_per_call_for).  The start noise is still drawn image by image in loader order, and every kernel is batch-composition
  independent, so each image's result is the one-at-a-time result bit for bit.  `images_per_call=1` is the reference's loop shape;
* a pipeline of depth two over the groups: a feeder thread pulls the loader, pins the group and copies it to the device on a copy stream while the
  previous group samples; the main thread queues a group's whole device work (HFRM, DWTs, sampler, IDWT, metric sums, 8-bit conversion) WITHOUT waiting
  for it, and only then reads the metrics of the group before -- the GPU's queue never runs dry between groups, PNG encoding runs behind in the writer's
  threads.  Console lines come out in loader order, as before."""
from __future__ import annotations

import os

import numpy as np
import torch

from .ddm_wavelet import data_transform, inverse_data_transform
from . import _lib, imageio, sampling


def torchPSNR(tar_img, prd_img):
    """utils/metrics.py:7-11."""
    imdff = torch.clamp(prd_img, 0, 1) - torch.clamp(tar_img, 0, 1)
    rmse = (imdff ** 2).mean().sqrt()
    return 20 * torch.log10(1 / rmse)


def save_image(img, path):
    """utils/logging.py:9-12 without torchvision, synchronous: (1,3,H,W) or (3,H,W) in [0,1] -> PNG."""
    from PIL import Image
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    a = img.detach().float().cpu()
    if a.dim() == 4:
        a = a[0]
    a = (a * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    Image.fromarray(a).save(path)


class _StageBudget:
    """Bytes of staged input (pinned host + device copies) the feeder may hold ahead of the sampler."""

    def __init__(self, limit, stop):
        import threading
        self.limit, self.used, self.stop, self.cv = max(1, limit), 0, stop, threading.Condition()

    def acquire(self, n):
        with self.cv:
            while self.used > 0 and self.used + n > self.limit and not self.stop.is_set():      # (one group always fits)
                self.cv.wait(0.25)
            self.used += n

    def release(self, n):
        with self.cv:
            self.used -= n
            self.cv.notify_all()

    def wake(self):
        with self.cv:
            self.cv.notify_all()


class DiffusiveRestoration:
    def _mark(self, what):
        """WAVEDM_RESTORE_TRACE=1: host-side timeline of restore() in self.trace [(seconds since the call, thread, label)] (scripts/restore_trace.py)."""
        if self.trace is not None:
            import threading
            import time
            ev = None
            if threading.current_thread() is threading.main_thread() and os.environ.get("WAVEDM_RESTORE_TRACE_GPU", "0") == "1":
                ev = torch.cuda.Event(enable_timing=True)                       # when the GPU gets to this point of the main stream
                ev.record(torch.cuda.current_stream(self.diffusion.device))
            self.trace.append((time.perf_counter() - self._t0, threading.current_thread().name, what, ev))

    def __init__(self, diffusion, args, config, save_images=True):
        self.trace = None
        self.args = args
        self.config = config
        self.diffusion = diffusion
        self.save_images = save_images
        self.writer = None
        if os.path.isfile(getattr(args, "resume", "") or ""):
            self.diffusion.model.eval()                                        # restoration.py:23-25
        else:
            print("Pre-trained diffusion model path is missing!")

    # ---- grouping ---------------------------------------------------------------------------------------------------
    def _max_batch(self):
        mb = getattr(self.diffusion.args, "max_batch", None) or getattr(self.args, "max_batch", None)
        return int(mb) if mb else sampling.DEFAULT_MAX_BATCH

    def images_per_call_for(self, h, w, r=None):
        """How many loader items of wavelet-domain size h x w go into one sampler call: `args.images_per_call` when it is a number; otherwise ("auto", None, 0)
        the smallest count (up to 16) whose patches fill the UNet calls to >= 97 %, else the best-filling one.  A UNet call works in units of 64 patches -- every
        level's launch then holds whole rounds of workgroups on the 256 CUs (64 patches = 256 tiles of a 16x16 or 32x32 map, 512 of a 64x64 map or an 8x8 map's
        two-image tiles) --, and the sampler splits n patches into ceil(n / max_batch) equal calls (sampling.ddim_sample).  45 patches per 480x720 image under the
        default cap of 384: 7 images = 315 patches = one call, 98 % of five units (profiles/r06_restore_sweep.log: 8 x 45 = 360 in calls of 120 6.32 img/s,
        in one call 6.47; a call of 144 = 2.25 units 5.70)."""
        v = getattr(self.args, "images_per_call", None)
        if v not in (None, 0, "auto", "Auto", "AUTO"):
            return max(1, int(v))
        p_size = self.config.data.patch_size if self.config.data.wavelet_in_unet else self.config.data.image_size
        hl, wl = sampling.overlapping_grid_indices(h, w, p_size, r)
        P, mb = len(hl) * len(wl), self._max_batch()

        def fill(n_img):
            n = n_img * P
            calls = -(-n // mb)
            per = -(-n // calls)
            return n / (calls * (-(-per // 64) * 64))
        cap = max(16, mb // max(P, 1))
        return max(range(1, cap + 1), key=lambda n: (min(fill(n), 0.97), -n))        # the smallest count that reaches 97 %, else the best fill (then the smallest)

    def _feed(self, val_loader, r, q, stop, budget, hungry):
        """Feeder thread: loader items -> groups of same-sized images -> pinned -> device (copy stream).  Puts (x_dev, names, copy_done_event, pinned) on q,
        an exception if one happened, then None."""
        dev = self.diffusion.device
        try:
            torch.cuda.set_device(dev)
            copy_stream = torch.cuda.Stream(device=dev)
            group, limit, n_emitted = [], 1, 0
            # partial groups are a timing decision of THIS process: never in the patch-sharded mode, where every rank must form the same groups (one all-reduce per step)
            auto = (getattr(self.args, "images_per_call", None) in (None, 0, "auto", "Auto", "AUTO")) and getattr(self.diffusion, "patch_group", None) is None

            def emit(group):
                nonlocal n_emitted
                n_emitted += 1
                self._mark(f"feeder: group of {len(group)} read")
                names = [it[1] for it in group]
                if group[0][0].is_cuda:
                    x = torch.cat([it[0] for it in group], dim=0).float().contiguous()
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    q.put((x, names, ev, None))
                    return
                # gathered straight into pinned memory (the host allocator recycles these blocks once their copies are done)
                xp = _lib.pinned_dontfork(torch.empty((len(group),) + tuple(group[0][0].shape[1:]), dtype=torch.float32, pin_memory=True))
                for k, it in enumerate(group):
                    xp[k].copy_(_lib.pinned_dontfork(it[0])[0])                # (a pin_memory=True loader's own pinned batch: kept out of the NEXT fork as well)
                with torch.cuda.stream(copy_stream):
                    xd = xp.to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                self._mark("feeder: group pinned, copy queued")
                budget.acquire(xp.numel() * 4)
                q.put((xd, names, ev, xp))

            for i, (x, y, total) in enumerate(val_loader):
                if stop.is_set():
                    return
                x = x.flatten(start_dim=0, end_dim=1) if x.ndim == 5 else x    # restoration.py:72
                for k in range(x.shape[0]):                                    # loader batches are split into images
                    name = y[k] if isinstance(y, (list, tuple)) and len(y) == x.shape[0] else y
                    item = (x[k:k + 1], name)
                    if group and group[0][0].shape != item[0].shape:           # a different size closes the group
                        emit(group)
                        group = []
                    if not group:
                        limit = self.images_per_call_for(item[0].shape[-2] // 4, item[0].shape[-1] // 4, r)
                    group.append(item)
                    # full -- or, with automatic grouping, the sampler is WAITING for input (the start of a run, a loader slower than the GPU): what is there goes
                    # at once, the GPU starts on image 1 while the loader still decodes image 2 (per-image results do not depend on the grouping)
                    if len(group) >= limit or (auto and hungry.is_set()):
                        emit(group)
                        group = []
            if group:
                emit(group)
        except BaseException as e:                                             # surfaced by restore()
            q.put(e)
        finally:
            q.put(None)

    # ---- one sampler call over a group of same-sized loader items: everything QUEUED here, nothing waited for ---------
    def _launch_group(self, staged, r, image_folder):
        cfg, d = self.config, self.diffusion
        pc, ob = cfg.model.pred_channels, cfg.model.other_channels_begin
        x, names, copied, pinned = staged
        cur = torch.cuda.current_stream(d.device)
        cur.wait_event(copied)
        x.record_stream(cur)                                                   # (allocated on the feeder's copy stream)
        inp, gt = x[:, :3].contiguous(), x[:, 3:].contiguous()
        x_cond = d.wavelet_dec.forward_affine(inp)                             # restoration.py:79, :88: DWT(2x - 1) in one kernel
        x_gt = d.wavelet_dec.forward_affine(gt)                                # :89
        self._mark("main: DWTs queued")
        hf = d.generator(inp)                                                  # :94 (HFRM)
        self._mark("main: HFRM queued")
        hf_wav = d.wavelet_dec.forward_affine(hf.contiguous())                 # :95-96
        x_other = hf_wav[:, ob:].contiguous()                                  # :102
        early = bool(getattr(self.args, "early_stop", True))
        if int(self.diffusion.args.sampling_timesteps) < 5:
            raise IndexError("x0_preds[-5] needs at least 5 sampling steps (restoration.py:108)")
        rec = lambda lo, hi: d.wavelet_rec.compose(lo, hi, pc)                    # IDWT(cat([lo[:, :pc], hi[:, pc:]])) -> clamp((x + 1) / 2), one kernel
        names = [n[0] if isinstance(n, (list, tuple)) else n for n in names]
        w = self.writer if self.save_images else None
        if w is not None:
            # five of the reference's seven PNGs per image do not depend on the sampler: converted and handed to the writer BEFORE the sampler is queued, so that
            # they are encoded while it runs and only two per image are left for the end of the group (the order of the files on disk is nobody's contract)
            for k, name in enumerate(names):
                sl = slice(k, k + 1)
                w.save(rec(x_gt[sl], hf_wav[sl]), os.path.join(image_folder, f"{name}_lrgt_hrwdnet.png"))      # :118-120, :158
                w.save(hf[sl], os.path.join(image_folder, f"{name}_all_wdnet.png"))
                w.save(rec(x_gt[sl], x_cond[sl]), os.path.join(image_folder, f"{name}_lrgt_hrcond.png"))       # :121-123
                w.save(inp[sl], os.path.join(image_folder, f"{name}_cond.png"))
                w.save(gt[sl], os.path.join(image_folder, f"{name}_gt.png"))
        xs, x0_preds = self.diffusive_restoration(x_cond, x_other=x_other, r=r, last=False, total=None,
                                                  use_global=False, use_other=True, stop_at=-5 if early else None)
        pred = x0_preds[-5]                                                    # :108
        x_output = rec(pred, hf_wav)                                           # :114-115, :124, :134
        H, W = x_output.shape[-2:]
        # the three pairs the reference prints: output, "cond" (IDWT(DWT(x)) == x: the input), HFRM image (restoration.py:146 clamps x_output_wdnet first)
        self._mark("main: sampler queued")
        sums = torch.stack([imageio.sqdiff(gt, x_output), imageio.sqdiff(gt, inp), imageio.sqdiff(gt, hf.clamp(0.0, 1.0))])
        sums_host = _lib.pinned_dontfork(torch.empty(sums.shape, dtype=sums.dtype, pin_memory=True))
        sums_host.copy_(sums, non_blocking=True)
        done = torch.cuda.Event()
        done.record(cur)
        if w is not None:
            for k, name in enumerate(names):
                sl = slice(k, k + 1)
                w.save(x_output[sl], os.path.join(image_folder, f"{name}_output.png"))
                w.save(rec(pred[sl], x_gt[sl]), os.path.join(image_folder, f"{name}_lrdiff_hrgt.png"))         # :112-113
        return dict(names=names, out=x_output, sums=sums_host, done=done, HW=(H, W), keep=(pinned, sums))

    def _finish_group(self, g, acc):
        """Wait for a queued group's metric sums and print what the reference prints per image."""
        g["done"].synchronize()
        H, W = g["HW"]
        m_out, m_cond, m_hf = (imageio.psnr_from_sums(g["sums"][j], H, W) for j in range(3))
        for k, name in enumerate(g["names"]):
            acc["torch"].append(m_out[k][0]); acc["y"].append(m_out[k][1]); acc["wdnet"].append(m_hf[k][1])
            print("psnr this", m_out[k][0])
            print("psnr cond", m_cond[k][0])
        return [g["out"][k:k + 1] for k in range(len(g["names"]))]

    def restore(self, val_loader, validation="snow", r=None):
        import queue
        import threading
        cfg, d = self.config, self.diffusion
        if not (cfg.data.wavelet and not cfg.data.wavelet_in_unet and cfg.model.use_other_channels):
            raise NotImplementedError("DiffusiveRestoration.restore: only the raindrop_wavelet.yml branch is accelerated")
        image_folder = os.path.join(self.args.image_folder, cfg.data.dataset, validation)
        if os.environ.get("WAVEDM_RESTORE_TRACE", "0") == "1":
            import time
            self.trace, self._t0 = [], time.perf_counter()
        if self.save_images and self.writer is None:
            self.writer = imageio.AsyncImageWriter()
        acc = {"torch": [], "y": [], "wdnet": []}
        outputs, pending = [], None
        # groups staged ahead of the sampler: bounded by BYTES (args.prefetch_bytes, default 2 GB of device + pinned memory each), not by count -- a short
        # validation set is read to its end at once, which also lets a fork-based DataLoader's worker processes exit early (see _StageBudget)
        q, stop = queue.Queue(), threading.Event()
        budget = _StageBudget(int(getattr(self.args, "prefetch_bytes", 2 << 30)), stop)
        hungry = threading.Event()                                              # set while the main thread waits for a group
        feeder = threading.Thread(target=self._feed, args=(val_loader, r, q, stop, budget, hungry), name="wavedm-restore-feeder", daemon=True)
        feeder.start()
        try:
            with torch.no_grad(), torch.cuda.device(d.device):
                while True:
                    try:
                        staged = q.get_nowait()
                    except queue.Empty:
                        hungry.set()
                        staged = q.get()
                        hungry.clear()
                    if staged is None:
                        break
                    if isinstance(staged, BaseException):
                        raise staged
                    self._mark("main: group taken")
                    budget.release(staged[0].numel() * 4 if staged[3] is not None else 0)
                    cur = self._launch_group(staged, r, image_folder)          # group k queued behind group k - 1 ...
                    self._mark("main: group queued")
                    if pending is not None:
                        outputs += self._finish_group(pending, acc)            # ... before the host waits for k - 1's numbers
                        self._mark("main: previous group's numbers in")
                    pending = cur
                if pending is not None:
                    outputs += self._finish_group(pending, acc)
        finally:
            stop.set()
            budget.wake()                                                      # (an error above: the feeder runs into its stop flag instead of waiting for room)
            feeder.join()
        self._mark("main: last group done")
        if self.writer is not None:
            self.writer.flush()
        self._mark("main: PNGs flushed")
        if acc["torch"]:
            print("psnr all torch", float(np.mean(acc["torch"])))
            print("psnr all np", float(np.mean(acc["y"])))
            print("psnr all GPU", float(np.mean(acc["y"])))       # deliberately the same accumulator: the reference's numpy and torch Y-PSNR agree
            print("psnr all wdnet", float(np.mean(acc["wdnet"])))
        self.last_outputs, self.last_psnrs, self.last_psnrs_y = outputs, acc["torch"], acc["y"]
        return outputs, acc["torch"]

    def diffusive_restoration(self, x_cond, x_other=None, r=None, last=True, total=None, use_global=False, use_other=False, stop_at=None):
        """restoration.py:170-185.  The start noise is drawn image by image (the reference sees one image per call), so a
        batched call consumes the generator exactly like the same images restored one after the other.  `stop_at`: see sample_image."""
        p_size = self.config.data.patch_size if self.config.data.wavelet_in_unet else self.config.data.image_size
        h_list, w_list = self.overlapping_grid_indices(x_cond, output_size=p_size, r=r)
        corners = [(i, j) for i in h_list for j in w_list]
        shp = (1, self.config.model.pred_channels, x_cond.shape[2], x_cond.shape[3])
        x = torch.cat([torch.randn(shp, device=self.diffusion.device) for _ in range(x_cond.shape[0])], dim=0)
        return self.diffusion.sample_image(x_cond, x, x_other=x_other, last=last, patch_locs=corners, patch_size=p_size,
                                           total=total, use_global=use_global, use_other=use_other, stop_at=stop_at)

    def overlapping_grid_indices(self, x_cond, output_size, r=None):
        """restoration.py:187-196."""
        _, c, h, w = x_cond.shape
        return sampling.overlapping_grid_indices(h, w, output_size, r)
