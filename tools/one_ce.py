import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
B, N, S = 16, 21, 512
logits = torch.randn(B, N, S, S, device=dev)
lab = torch.randint(0, N, (B, S, S), device=dev)
conf = torch.rand(B, S, S, device=dev)
ign = torch.zeros(B, S, S, dtype=torch.int64, device=dev)
mc = torch.randint(0, N, (B, S, S), device=dev)
dl = torch.empty_like(logits)
gs = torch.tensor([1e-6, 1e-7], device=dev)
for _ in range(3):
    ops.ce_fused(logits, lab, False, conf=conf, ign=ign, conf_thresh=0.5, mc=mc, dlogits=dl, gscale=gs)
    ops.softmax_max(logits)
torch.cuda.synchronize()
