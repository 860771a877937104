"""Times GradAllReducer.reduce() on a 125 MB GPU arena under SVL_DIST_BACKEND (2 processes on one GPU: gloo)."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("SVL_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
from semivl_amd.train import GradAllReducer
class A: pass
a = A(); a.g = torch.ones(31_350_000, device="cuda"); a.p = torch.ones(31_350_000, device="cuda"); a.grad_scale = 1.0
red = GradAllReducer(a)
for tag, fn in (("broadcast", red.broadcast_params), ("reduce", red.reduce), ("reduce", red.reduce), ("barrier", dist.barrier)):
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize()
    if rank == 0: print(f"{tag}: {time.time() - t0:.3f} s", flush=True)
dist.destroy_process_group()
