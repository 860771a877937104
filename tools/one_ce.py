"""One fused CE forward+backward launch at the bench shape (B=16, N=21, 512^2) for the PMC traffic passes (tools/pmc_ce.sh)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
B, N, S = int(os.environ.get("ONE_B", 16)), int(os.environ.get("ONE_NCLS", 21)), int(os.environ.get("ONE_S", 512))
logits = torch.randn(B, N, S, S, device=dev)
tgt = torch.randint(0, N, (B, S, S), device=dev)
conf = torch.rand(B, S, S, device=dev)
ign = torch.zeros(B, S, S, dtype=torch.int64, device=dev)
mc = torch.randint(0, N, (B, S, S), device=dev)
dl = torch.empty_like(logits)
gs = torch.ones(2, device=dev)
sums = torch.zeros(4, dtype=torch.float64, device=dev)
for _ in range(3):
    ops.ce_fused(logits, tgt, False, conf=conf, ign=ign, conf_thresh=0.5, mc=mc, dlogits=dl, gscale=gs, sums_out=sums)
torch.cuda.synchronize()
