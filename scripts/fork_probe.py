#!/usr/bin/env python3
"""Round 6: does a process with a HIP context launch kernels more slowly while fork()ed children of it are alive?  (restore() found: yes, tenfold, profiles/r06_restore_fork_interference.log.)
Times the same loops -- torch elementwise launches, library launches (wdm_dwt_fwd on a small batch), small allocations -- before the fork, with 8 sleeping children alive, and after
they are gone.  Variants: PIN_MB=<n> pinned host memory held by the parent, WS_GB=<n> device memory held, THP=never (madvise off for new mappings is not possible from here: informational)."""
import os
import signal
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wavedm_amd.wavelet import WaveletTransform

dev = torch.device("cuda", 0)
x = torch.zeros(1 << 16, device=dev)
img = torch.rand(2, 3, 64, 64, device=dev)
dwt = WaveletTransform(scale=2, dec=True)
pin = torch.empty(int(os.environ.get("PIN_MB", "0")) << 20, dtype=torch.uint8, pin_memory=True) if os.environ.get("PIN_MB") else None
ws = torch.empty(int(os.environ.get("WS_GB", "0")) << 30, dtype=torch.uint8, device=dev) if os.environ.get("WS_GB") else None
host = [bytearray(1 << 20) for _ in range(int(os.environ.get("HEAP_MB", "0")))]      # private, written heap pages of the parent
if pin is not None and os.environ.get("DONTFORK") == "1":                              # keep the pinned pages out of the children: no copy-on-write on them at fork()
    import ctypes
    libc = ctypes.CDLL("libc.so.6", use_errno=True)
    a0 = pin.data_ptr() & ~4095
    n = ((pin.data_ptr() + pin.numel() + 4095) & ~4095) - a0
    rc = libc.madvise(ctypes.c_void_p(a0), ctypes.c_size_t(n), 10)                    # MADV_DONTFORK
    print("madvise(MADV_DONTFORK) on the pinned buffer:", rc, ctypes.get_errno(), flush=True)


def loops(tag):
    # completion latency first: one tiny kernel, queued and waited for (a stalled GPU queue shows here, not in the host-side enqueue times below)
    c0 = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    lat = time.perf_counter() - c0
    print(f"{tag:<34s} first kernel queued -> complete: {1e3 * lat:9.2f} ms", flush=True)
    t0 = time.perf_counter()
    for _ in range(2000):
        x.add_(1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(2000):
        dwt(img)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    for _ in range(2000):
        torch.empty(1 << 20, device=dev)
    t5 = time.perf_counter()
    for h in host:
        h[0] = 1; h[4096] = 1
    t6 = time.perf_counter()
    extra = ""
    if pin is not None:                                      # write the pinned buffer (allocated BEFORE the fork) on the host, then DMA it to the device
        dst = torch.empty(pin.numel(), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        t7 = time.perf_counter()
        pin.fill_(3)
        t8 = time.perf_counter()
        dst.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        t9 = time.perf_counter()
        fresh = torch.empty(pin.numel(), dtype=torch.uint8, pin_memory=True)       # ... and the same with a pinned buffer allocated NOW
        fresh.fill_(3)
        t10 = time.perf_counter()
        dst.copy_(fresh, non_blocking=True)
        torch.cuda.synchronize()
        t11 = time.perf_counter()
        del fresh
        extra = f"   pinned {pin.numel() >> 20} MB: host write {1e3 * (t8 - t7):7.1f} ms, H2D {1e3 * (t9 - t8):7.1f} ms | fresh block: alloc + write {1e3 * (t10 - t9):7.1f} ms, H2D {1e3 * (t11 - t10):7.1f} ms"
    print(f"{tag:<34s} torch add_ {1e3 * (t1 - t0) / 2:7.1f} us/launch   wdm_dwt_fwd {1e3 * (t3 - t2) / 2:7.1f} us/call   torch.empty {1e3 * (t5 - t4) / 2:7.1f} us   heap touch {1e3 * (t6 - t5):7.2f} ms{extra}", flush=True)


loops("before any fork")
loops("before any fork (again)")
kids = []
for _ in range(8):
    pid = os.fork()
    if pid == 0:
        time.sleep(60)
        os._exit(0)
    kids.append(pid)
loops("8 forked children alive")
loops("8 forked children alive (again)")
loops("8 forked children alive (third)")
for pid in kids:
    os.kill(pid, signal.SIGKILL)
    os.waitpid(pid, 0)
loops("children gone")
loops("children gone (again)")
