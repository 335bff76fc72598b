"""Image output off the critical path (reference: `utils/logging.py:9-12`, called ~7x per image from
`models/restoration.py:158-166` with a synchronous float D2H copy + PNG encode each).

`AsyncImageWriter.save(img, path)`: quantise on the GPU (wdm_to_u8_hwc: one byte per sample crosses PCIe instead of four),
copy to a pinned buffer on a side stream, hand the buffer to a worker thread that waits for the copy's event and encodes
the PNG with PIL.  The sampler's stream never waits for any of it.  `metrics(gt, out)` returns the three PSNRs the
reference prints, from one device reduction (wdm_image_sqdiff)."""
from __future__ import annotations

import math
import os
import queue
import threading

import torch

from . import _lib


def to_u8_hwc(img: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) or (C,H,W) f32 on the GPU -> (B,H,W,C) uint8 with torchvision.utils.save_image's rounding."""
    img = _lib.require_cuda_f32(img if img.dim() == 4 else img[None], "to_u8_hwc input")
    B, C, H, W = img.shape
    out = torch.empty(B, H, W, C, dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.lib().wdm_to_u8_hwc(_lib.handle(img.device.index or 0), _lib.ptr(img), B, C, H, W, _lib.ptr(out), _lib.stream_ptr()))
    return out


def make_grid(tensor: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """torchvision.utils.make_grid's layout (torchvision is not a dependency here): (N,C,H,W) -> (C, rows*(H+pad)+pad, cols*(W+pad)+pad)
    with cols = min(nrow, N), image k at row k // cols, column k % cols; one image is returned unframed; 1-channel input is repeated
    to 3.  Used by the training loop's validation sheet (ddm_wavelet.py:407-410).  Works on any device."""
    if tensor.dim() == 3:
        tensor = tensor[None]
    if tensor.shape[1] == 1:
        tensor = tensor.expand(-1, 3, -1, -1)
    N, Cc, H, W = tensor.shape
    if N == 1:
        return tensor[0]
    cols = min(int(nrow), N)
    rows = (N + cols - 1) // cols
    h, w = H + padding, W + padding
    grid = tensor.new_full((Cc, h * rows + padding, w * cols + padding), float(pad_value))
    for k in range(N):
        r, c = divmod(k, cols)
        grid[:, r * h + padding:r * h + padding + H, c * w + padding:c * w + padding + W] = tensor[k]
    return grid


def sqdiff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Per-image [sum (clamp a - clamp b)^2 over RGB, sum (Y(a)-Y(b))^2] in fp64 on the device: (B,2)."""
    a, b = _lib.require_cuda_f32(a, "metrics input"), _lib.require_cuda_f32(b, "metrics input")
    if a.shape != b.shape or a.dim() != 4 or a.shape[1] != 3:
        raise ValueError(f"metrics: expected two (B,3,H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
    B, _, H, W = a.shape
    out = torch.empty(B, 2, dtype=torch.float64, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().wdm_image_sqdiff(_lib.handle(a.device.index or 0), _lib.ptr(a), _lib.ptr(b), B, H, W, _lib.ptr(out), _lib.stream_ptr()))
    return out


def psnr_from_sums(sums, H, W):
    """-> list of (psnr_torch, psnr_y) per image; utils/metrics.py:7-11 and :43-51 / :53-77 (the numpy Y-PSNR on 0..255 data is the
    same number as the GPU one: the 255s cancel)."""
    res = []
    for s_rgb, s_y in sums.tolist():
        mse_rgb, mse_y = s_rgb / (3.0 * H * W), s_y / (H * W)
        res.append((20.0 * math.log10(1.0 / math.sqrt(mse_rgb)) if mse_rgb > 0 else float("inf"),
                    20.0 * math.log10(1.0 / math.sqrt(mse_y)) if mse_y > 0 else float("inf")))
    return res


class AsyncImageWriter:
    """PNG output off the critical path: 8-bit conversion on the device, D2H copy on a side stream into pinned memory, encoding in worker threads.
    `workers` threads share one queue (zlib releases the GIL): with one thread the seven PNGs per 480x720 image of restore() took 3 s of an 8-image call
    whose sampling takes 1.5 s (bench.py: configs[4] whole pipeline); files are PIL's default encoding, as torchvision.utils.save_image writes them.
    Several threads take the FIFO order away, so saves to the SAME path are ordered explicitly: every save gets a per-path sequence number, a worker holds the path's
    lock while it writes and skips its item when a later save to that path has been queued -- the last save wins, as with one thread."""

    def __init__(self, max_pending: int = 512, workers: int = 0):
        # max_pending: restore() hands over a whole group's files (7 per image) without waiting; workers: one round of encoding for the ~50 files a group of seven
        # 480x720 images leaves at its end (a 480x720 PNG of incompressible data takes PIL ~55 ms)
        self._q: "queue.Queue" = queue.Queue(maxsize=max_pending)
        n = workers if workers > 0 else max(1, min(64, (os.cpu_count() or 2) // 4))
        self._threads = [threading.Thread(target=self._run, name=f"wavedm-png-writer-{k}", daemon=True) for k in range(n)]
        self._stream = None
        self._errors = []
        self._seq = {}                                  # path -> (sequence number of the latest save queued, lock held while a worker writes the file)
        self._seq_mu = threading.Lock()
        for t in self._threads:
            t.start()

    def _run(self):
        from PIL import Image
        while True:
            item = self._q.get()
            try:
                if item is None:
                    return
                host, event, path, seq = item
                event.synchronize()
                os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
                arr = host.numpy()
                with self._seq_mu:
                    lock = self._seq[path][1]
                with lock:
                    with self._seq_mu:
                        latest = self._seq[path][0]
                    if seq == latest:                       # (an older save to a path that has been saved again since: the newer one stands)
                        Image.fromarray(arr[..., 0] if arr.shape[-1] == 1 else arr).save(path)
            except Exception as e:                      # surfaced by flush()
                self._errors.append(e)
            finally:
                self._q.task_done()

    def save(self, img: torch.Tensor, path: str):
        """img: (1,C,H,W) or (C,H,W) f32 in [0,1] on the GPU (utils/logging.save_image's first image semantics)."""
        if img.dim() == 4:
            img = img[:1]
        u8 = to_u8_hwc(img)[0]
        dev = u8.device
        if self._stream is None or self._stream.device != dev:
            self._stream = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream(dev)
        host = _lib.pinned_dontfork(torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True))      # (pinned pages a later fork() need not make copy-on-write: _lib.pinned_dontfork)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            host.copy_(u8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        u8.record_stream(self._stream)
        with self._seq_mu:
            n, lock = self._seq.get(path, (0, None))
            self._seq[path] = (n + 1, lock or threading.Lock())
        self._q.put((host, ev, path, n + 1))

    def flush(self):
        self._q.join()
        with self._seq_mu:
            self._seq.clear()
        if self._errors:
            e, self._errors = self._errors[0], []
            raise e

    def close(self):
        self.flush()
        for _ in self._threads:
            self._q.put(None)
        for t in self._threads:
            t.join(timeout=10)
