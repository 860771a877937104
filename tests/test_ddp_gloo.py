"""CPU, world_size 2, gloo: the data-parallel gradient reduction used by bench.py --gpus N (GradAllReducer)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Arena:
    """Stand-in for FusedAdamW's flat arenas (the real one needs the GPU)."""

    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.g = torch.randn(n, generator=g)
        self.p = torch.full((n,), float(rank + 1))
        self.grad_scale = 1.0


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semivl_amd.train import GradAllReducer
    arena = _Arena(n, rank)
    red = GradAllReducer(arena, bucket_mb=0.01)
    assert len(red.buckets) > 3 and red.buckets[0][0] == 0 and red.buckets[-1][1] == n
    red.broadcast_params()
    red.reduce()
    q.put((rank, arena.g.numpy().copy(), arena.p.numpy().copy(), arena.grad_scale))  # by value, not shm handles
    dist.destroy_process_group()


def test_grad_allreduce_world2():
    world, n = 2, 10007
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    expect = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
    for rank, g, p, scale in res:
        assert torch.allclose(torch.from_numpy(g), expect, atol=1e-6), "SUM over ranks (1/W is folded into AdamW)"
        assert scale == 0.5
        assert (p == 1.0).all(), "parameters broadcast from rank 0"
