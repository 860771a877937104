"""Three launches of the resize-fused cross entropy (svl_ce_up_fused_f32) + softmax-max at the bench shape (B=16, N=21,
128^2 -> 512^2) for the SQ-counter record:  bash tools/pmc_sq.sh tools/one_ce_up.py ce_up_kernel"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
B, N, S = int(os.environ.get("ONE_B", 16)), int(os.environ.get("ONE_NCLS", 21)), int(os.environ.get("ONE_S", 512))
h = 4 * ((S + 15) // 16)
lg = torch.randn(B, N, h, h, device=dev) * 3
tgt = torch.randint(0, N, (B, S, S), device=dev)
conf = torch.rand(B, S, S, device=dev)
ign = torch.zeros(B, S, S, dtype=torch.int64, device=dev)
mc = torch.randint(0, N, (B, S, S), device=dev)
dl = torch.empty_like(lg)
gs = torch.ones(2, device=dev)
sums = torch.zeros(4, dtype=torch.float64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 1:
        e0.record()
    ops.ce_up_fused(lg, S, S, False, tgt, False, conf=conf, ign=ign, conf_thresh=0.5, mc=mc, dlogits=dl, gscale=gs, sums_out=sums)
e1.record()
ops.softmax_max_up(lg, S, S, False)
torch.cuda.synchronize()
print(f"ce_up_fused B={B} N={N} {h}^2 -> {S}^2: {e0.elapsed_time(e1) / 3:.3f} ms per launch")
