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
    red = GradAllReducer(arena, bucket_mb=0.01, profile=True)
    assert len(red.buckets) > 3 and red.buckets[0][0] == 0 and red.buckets[-1][1] == n
    red.broadcast_params()
    red.reduce()
    rep = red.timing_report()          # bench.py's "allreduce" object: one entry per bucket, blocking backend = all exposed
    assert rep is not None and len(rep["buckets"]) == len(red.buckets) and rep["exposed_ms"] > 0
    assert abs(sum(b["mbytes"] for b in rep["buckets"]) - n * 4 / 2 ** 20) < 0.1 * len(red.buckets)
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


class _Opt:
    """FusedAdamW's arena layout on the CPU: per-parameter segments of one flat gradient buffer, `main_grad` views."""

    def __init__(self, sizes, rank):
        self.params = [torch.nn.Parameter(torch.zeros(s)) for s in sizes]
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + (s + 3) // 4 * 4)
        self.seg_off = torch.tensor(offs)
        self.g = torch.zeros(offs[-1])
        self.p = torch.zeros(offs[-1])
        self.groups = [dict(param=p) for p in self.params]
        for p, o, s in zip(self.params, offs, sizes):
            p.main_grad = self.g[o:o + s]
        self.grad_scale = 1.0
        self.rank = rank

    def _fold_autograd_grads(self):   # a gradient that arrived through torch autograd (pos_embed behind its resize)
        for p in self.params:
            if p.grad is not None:
                p.main_grad += p.grad
                p.grad = None


def _overlap_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semivl_amd import gradsync
    from semivl_amd.train import GradAllReducer
    sizes = [1000, 3000, 10, 5000, 7, 2500, 4000]
    opt = _Opt(sizes, rank)
    red = GradAllReducer(opt, bucket_mb=0.01, overlap=True)      # ~2600 floats per bucket
    assert len(red.buckets) >= 3 and red.buckets[-1][1] == opt.g.numel()
    log = []
    orig = red._fire
    red._fire = lambda b_: (log.append(b_) if b_ not in red._fired else None, orig(b_))
    for step in range(2):                                        # counters must reset between steps
        opt.g.zero_()
        log.clear()
        # two graphs reach every parameter except #0, which only gets an autograd-delivered gradient
        for _ in range(2):
            gradsync.expect(opt.params[1:])
        for graph in range(2):
            for i in range(len(sizes) - 1, 0, -1):               # backward order: last parameter first
                opt.params[i].main_grad += float((rank + 1) * (graph + 1))
                gradsync.ready([opt.params[i]])
                if graph == 0:
                    assert not log, "a bucket fired before its second contribution"
        early = list(log)
        opt.params[0].grad = torch.full((sizes[0],), 10.0 * (rank + 1))
        red.finish()
        q.put((rank, step, early, list(log), opt.g.clone().numpy()))
    dist.destroy_process_group()


def test_overlapped_bucket_scheduling_world2():
    """Buckets fire from inside 'backward' exactly when all contributions of all their parameters are in, in
    backward-completion order, never early; stragglers (autograd-delivered gradients) are folded and flushed by finish()."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 7) % 1000
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sizes = [1000, 3000, 10, 5000, 7, 2500, 4000]
    for rank, step, early, all_fired, g in res:
        assert early and early == sorted(early, reverse=True), early       # last buckets first
        assert 0 not in early and sorted(all_fired) == list(range(max(all_fired) + 1)) and all_fired[-1] == 0
        o = 0
        for i, s in enumerate(sizes):
            want = 10.0 * 3 if i == 0 else (1 + 2) * (1 + 2)               # sum over ranks (and graphs)
            assert (g[o:o + s] == want).all(), (i, g[o], want)
            o += (s + 3) // 4 * 4
