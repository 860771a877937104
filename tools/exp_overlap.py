import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from semivl_amd import ops
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, semivl_train_step
dev = torch.device("cuda:0")
cfg = exp40_cfg(16, 512, 21, "pascal")
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"])
batch = synthetic_batch(16, 512, 21, seed=1234, device=dev)
def run(tag, **kw):
    c = dict(cfg, **kw)
    for i in range(2): l = semivl_train_step(model, batch, i, 10000, c, optimizer=opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(6): l = semivl_train_step(model, batch, i, 10000, c, optimizer=opt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(f"{tag:28s} {dt*1e3:7.1f} ms/step  {32/dt:6.2f} img/s  loss {l[0].item():.5f}", flush=True)
mode = int(os.environ.get("EMU", "0"))
ops.set_gemm_emulation(mode)
run(f"emu{mode} overlap", overlap_streams=True)
run(f"emu{mode} sequential", overlap_streams=False)
run(f"emu{mode} overlap", overlap_streams=True)
run(f"emu{mode} sequential", overlap_streams=False)
