"""Step time and peak memory vs cfg['act_mem_fraction'] (the decoder's keep / recompute budget) at N = 81 / 150.
Env: NCLASS (150), DATASET (ade), BATCH (16), STEPS (3), FRACS ("0.62,0.72,0.8")."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from semivl_amd import ops
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, semivl_train_step
dev = torch.device("cuda:0")
B_, N_, D_ = int(os.environ.get("BATCH", 16)), int(os.environ.get("NCLASS", 150)), os.environ.get("DATASET", "ade")
cfg = exp40_cfg(B_, 512, N_, D_)
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"])
batch = synthetic_batch(B_, 512, N_, seed=1234, device=dev)
ops.set_gemm_emulation(6)
n = int(os.environ.get("STEPS", 3))
for frac in [float(f) for f in os.environ.get("FRACS", "0.62,0.72,0.8").split(",")]:
    c = dict(cfg, act_mem_fraction=frac)
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    semivl_train_step(model, batch, 0, 10000, c, optimizer=opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        l = semivl_train_step(model, batch, i, 10000, c, optimizer=opt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"N={N_} act_mem_fraction {frac}: {dt * 1e3:7.1f} ms/step  {2 * B_ / dt:6.2f} img/s  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB "
          f"reserved {torch.cuda.max_memory_reserved() / 2**30:.1f} GB", flush=True)
