"""Multi-process checks on the GPU box (two gloo processes sharing its one GPU).  (1) SyncBN exchange of the Cityscapes
side encoder across data-parallel ranks: the two processes each push half of a batch through ResNetV1c; outputs, running statistics and the rank-summed parameter
gradients must equal a single-process run on the whole batch (= what torch.nn.SyncBatchNorm guarantees).  (2) One full
training step through GradAllReducer + FusedAdamW on two ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, sd, img, dy, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semivl_amd.model.resnet import ResNetV1c
    dev = torch.device("cuda:0")
    m = ResNetV1c()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    half = img.shape[0] // world
    x = torch.from_numpy(img[rank * half:(rank + 1) * half]).to(dev)
    tok, (H, W) = m.forward_tokens(x)
    tok.backward(torch.from_numpy(dy[rank * half:(rank + 1) * half]).to(dev))
    torch.cuda.synchronize()
    q.put((rank, tok.detach().cpu().numpy(), {n: p.grad.cpu().numpy() for n, p in m.named_parameters()},
           {n: b.cpu().numpy() for n, b in m.named_buffers()}))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_two_ranks_equal_one_big_batch(dev):
    from semivl_amd.model.resnet import ResNetV1c
    torch.manual_seed(3)
    ref = ResNetV1c()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.copy_(1.0 + 0.2 * torch.randn_like(p) if n.endswith("weight") else 0.1 * torch.randn_like(p))
    sd = {k: v.numpy().copy() for k, v in ref.state_dict().items()}
    B, S = 4, 48
    img = torch.randn(B, 3, S, S).numpy()
    ref = ref.to(dev).train()
    tok, (H, W) = ref.forward_tokens(torch.from_numpy(img).to(dev))
    dy = torch.randn(B, H * W, 256).numpy()
    tok.backward(torch.from_numpy(dy).to(dev))
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, world, port, sd, img, dy, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    out = np.concatenate([r[1] for r in res], 0)
    assert np.abs(out - tok.detach().cpu().numpy()).max() < 2e-4
    for n, b in ref.named_buffers():
        for r in res:
            assert np.abs(r[3][n].astype(np.float64) - b.cpu().numpy()).max() < 1e-4 * max(1.0, float(b.abs().max())), n
    for n, p in ref.named_parameters():
        g = sum(r[2][n] for r in res)                     # DDP SUM (the 1/W is folded into the optimizer)
        e = np.linalg.norm(g - p.grad.cpu().numpy()) / (np.linalg.norm(p.grad.cpu().numpy()) + 1e-12)
        assert e < 3e-2, (n, e)                           # ReLU sign flips at rounding distance, see test_model_gpu


def _step_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, _one_step(torch.device("cuda:0"), distributed=True)))
    dist.barrier()
    dist.destroy_process_group()


def _one_step(dev, distributed):
    from golden_util import build_hip, fixture_batch, fixture_fp_masks, fixture_state, load_fixture
    from semivl_amd.synthetic import exp40_cfg
    from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step
    z, c = load_fixture("tiny")
    hip = build_hip(c)
    hip.load_state_dict(fixture_state(z, c, hip), strict=True)
    hip.to(dev)
    opt = FusedAdamW(hip, exp40_cfg()["optimizer"])
    red = GradAllReducer(opt, bucket_mb=0.05) if distributed else None
    if red is not None:
        red.broadcast_params()
    batch = {k: v.to(dev) for k, v in fixture_batch(z, c).items()}
    cfg = dict(conf_thresh=0.05, conf_mode="pixelwise", mcc_conf_thresh=0.9, mcc_loss_reduce="mean_all",
               maskclip_consistency_lambda=[0.1, 0])
    semivl_train_step(hip, batch, 3, 50, cfg, optimizer=opt, reducer=red, fp_masks=[m.to(dev) for m in fixture_fp_masks(z, c)])
    torch.cuda.synchronize()
    return opt.p.cpu().numpy()


def test_two_rank_step_with_identical_shards_equals_single_process(dev):
    """GradAllReducer (bucketed SUM on the side stream) + grad_scale 1/W inside FusedAdamW on GPU tensors: with the same
    shard on both ranks the mean gradient is the single-process gradient, so the updated arenas must match."""
    single = _one_step(dev, distributed=False)
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]), "ranks diverged"
    # (a + a) / 2 == a exactly, so only the reduction's summation order could differ: none here
    assert np.abs(res[0][1] - single).max() < 1e-7
