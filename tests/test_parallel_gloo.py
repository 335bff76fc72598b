"""N > 1 path on CPU: world-size-2 gloo processes exercise shard_range / all_gather_shards / restore_sharded and
the broadcast-then-adopt protocol (with a stand-in object for the packed buffer: the HIP kernels cannot run here)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavedm_amd import parallel


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class FakeUNet:
    """Mimics the three methods broadcast_weights uses; the 'packed buffer' is a CPU tensor."""
    def __init__(self, rank):
        self.rank, self.adopted = rank, False
        self._w = torch.nn.Parameter(torch.zeros(1))

    def parameters(self):
        return iter([self._w])

    def pack_weights(self):
        self.buf = torch.arange(1000, dtype=torch.uint8)
        return self.buf

    def alloc_packed(self, device):
        self.buf = torch.zeros(1000, dtype=torch.uint8)
        return self.buf

    def adopt_packed(self):
        self.adopted = True


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 7                                                     # ragged: shards of 4 and 3
        g = torch.Generator().manual_seed(0)
        x = torch.rand(n, 3, 4, 4, generator=g)
        restore = lambda t: t * 2 + 1                             # stands in for the per-image restoration
        out = parallel.restore_sharded(restore, [x], n)
        ok = torch.equal(out, x * 2 + 1)
        u = FakeUNet(rank)
        buf = parallel.broadcast_weights(u, src=0)
        ok = ok and torch.equal(buf, torch.arange(1000, dtype=torch.uint8)) and (u.adopted == (rank != 0))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        # the 45-patch list of one 480x720 image over 8 ranks: 6/6/6/6/6/5/5/5, gathered back in patch order
        for n in (45, 512, 13, 5):                                # ragged, even (configs[3]'s 512 images), and fewer items than ranks (empty shards)
            lo, hi = parallel.shard_range(n, rank, world)
            local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(-1, 2, 3).contiguous() * 2 + 1
            got = parallel.all_gather_shards(local, n)
            want = (torch.arange(n, dtype=torch.float32).view(-1, 1, 1).expand(-1, 2, 3) * 2 + 1)
            ok = ok and got.shape == (n, 2, 3) and torch.equal(got, want)
        sizes = [parallel.shard_range(45, r, world)[1] - parallel.shard_range(45, r, world)[0] for r in range(world)]
        ok = ok and sizes == [6, 6, 6, 6, 6, 5, 5, 5]
        u = FakeUNet(rank)
        buf = parallel.broadcast_weights(u, src=0)
        ok = ok and torch.equal(buf, torch.arange(1000, dtype=torch.uint8)) and (u.adopted == (rank != 0))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_eight_rank_gloo_ragged_all_gather():
    """World size 8 (the node the reference is launched on, train_weather_script.py:3): ragged, even and partly empty shards through all_gather_shards."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(8)]


def test_two_rank_gloo_sharded_restore_and_broadcast():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
