"""Times a 125 MB all-reduce of a GPU tensor under the backend in SVL_DIST_BACKEND (dry-run aid: gloo is the only
backend two processes can use on a ONE-GPU box; RCCL needs one GPU per rank)."""
import os, time, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("SVL_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
g = torch.ones(31_350_000, device="cuda")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    dist.all_reduce(g)
    torch.cuda.synchronize(); dt = time.time() - t0
    if rank == 0:
        print(f"all_reduce 125 MB: {dt:.2f} s", flush=True)
dist.destroy_process_group()
