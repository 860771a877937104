"""Wall-clock of the phases of one training step under torch.distributed (dry-run aid for the N>1 path)."""
import os, sys, time, cProfile, pstats, io, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group(os.environ.get("SVL_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step
dev = torch.device("cuda:0")
cfg = exp40_cfg(2, 512, 21, "pascal")
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"]); red = GradAllReducer(opt); red.broadcast_params()
batch = synthetic_batch(2, 512, 21, seed=1234 + rank, device=dev)
for i in range(2):
    semivl_train_step(model, batch, i, 100, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
semivl_train_step(model, batch, 2, 100, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize(); dt = time.time() - t0; pr.disable()
if rank == 0:
    print(f"step {dt:.3f} s, OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}", flush=True)
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500], flush=True)
if world > 1:
    dist.destroy_process_group()
