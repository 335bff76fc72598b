"""Multi-GPU sampling: independent images shard across ranks (one process per GPU), no collective
inside the DDIM loop.  Two collectives exist, both outside the timed hot loop:

  * `broadcast_weights`: rank 0 packs the checkpoint once; the single packed buffer (313 MB bf16) is
    broadcast over RCCL/xGMI and the other ranks adopt it (`wdm_unet_mark_loaded`);
  * `all_gather_shards`: the restored outputs of every rank are gathered at the end.

The reference has no such code (its eval is single-GPU, eval_diffusion.py:72-73).  The functions
take a process group and work with `gloo` on CPU tensors too, which is how tests cover N > 1."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced split of n items: the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(unet, src: int = 0, group=None):
    """Rank `src` holds loaded parameters; everyone ends with an identical packed buffer."""
    rank = dist.get_rank(group)
    dev = next(unet.parameters()).device
    if rank == src:
        buf = unet.pack_weights()
    else:
        buf = unet.alloc_packed(dev)
    dist.broadcast(buf, src=src, group=group)
    if rank != src:
        unet.adopt_packed()
    return buf


def all_gather_shards(local: torch.Tensor, n_total: int, group=None):
    """Gather per-rank shards (made with `shard_range`) of a batch back into (n_total, ...) on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def restore_sharded(restore_fn, inputs, n_total: int, group=None):
    """Run `restore_fn(*[t[lo:hi] for t in inputs]) -> tensor (hi-lo, ...)` on this rank's shard and all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_total, rank, world)
    out = restore_fn(*[t[lo:hi] for t in inputs])
    if world == 1:
        return out
    return all_gather_shards(out, n_total, group=group)
