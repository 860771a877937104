import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
Ms = [int(v) for v in os.environ.get("ONE_M", "32800").split(",")]
for M in Ms:
    for (N, K) in [(3072, 768), (768, 3072), (2304, 768), (768, 768)]:
        x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        for _ in range(3): ops.linear(x, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.linear(x, w, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(os.environ.get("SVL_GEMM_EMU"), os.environ.get("SVL_EMU_DBG"), M, N, K, f"{ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TF", flush=True)
