import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from semivl_amd import ops
from test_ops_gpu import rnd
dev = torch.device("cuda:0")
ops.set_gemm_emulation(6)
for (Bn, T, H) in [(1, 256, 1), (1, 128, 2), (2, 129, 3), (1, 516, 2), (1, 133, 2)]:
    E = 64 * H
    qkv = rnd(Bn * T, 3 * E, dev=dev, seed=50)
    qkv[:, :2 * E] *= 2.0
    if T > 40:
        qkv.view(Bn, T, 3 * E)[0, T - 3, E:E + 64] = 6.0 * qkv.view(Bn, T, 3 * E)[0, 5, 0:64]
    for it in range(2):
        out, lse = ops.attention_fwd(qkv, Bn, T, H)
        bad = torch.isnan(out).any(dim=1).nonzero().flatten()
        badc = torch.isnan(out).any(dim=0).nonzero().flatten()
        print(Bn, T, H, "it", it, "nan rows", bad.numel(), bad[:8].tolist(), bad[-4:].tolist(), "cols", badc.numel(), badc[:4].tolist(), "lse nan", int(torch.isnan(lse).sum()), "lse inf", int(torch.isinf(lse).sum()))
