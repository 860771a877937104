import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from semivl_amd import ops
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, semivl_train_step
dev = torch.device("cuda:0")
B_, S_, N_, D_ = (int(os.environ.get("BATCH", 16)), int(os.environ.get("CROP", 512)), int(os.environ.get("NCLASS", 21)),
                  os.environ.get("DATASET", "pascal"))
cfg = exp40_cfg(B_, S_, N_, D_)
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"])
batch = synthetic_batch(B_, S_, N_, seed=1234, device=dev)
def run(tag, **kw):
    c = dict(cfg, **kw)
    for i in range(2): l = semivl_train_step(model, batch, i, 10000, c, optimizer=opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = int(os.environ.get("STEPS", 6))
    for i in range(n): l = semivl_train_step(model, batch, i, 10000, c, optimizer=opt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{tag:28s} {dt*1e3:7.1f} ms/step  {2*B_/dt:6.2f} img/s  loss {l[0].item():.5f}", flush=True)
mode = int(os.environ.get("EMU", "0"))
ops.set_gemm_emulation(mode)
run(f"emu{mode} overlap", overlap_streams=True)
run(f"emu{mode} sequential", overlap_streams=False)
run(f"emu{mode} overlap", overlap_streams=True)
run(f"emu{mode} sequential", overlap_streams=False)
