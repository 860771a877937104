import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
M, N, K = 32800, int(os.environ.get("ONE_N", 3072)), int(os.environ.get("ONE_K", 768))
x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
out = torch.empty(M, N, device=dev)
for _ in range(5):
    ops.linear(x, w, out=out)
torch.cuda.synchronize()
