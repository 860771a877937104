"""LayerNorm forward / backward (+ planes) timing at the ViT shape: python tools/bench_ln.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
rows, C = 32800, 768
x, dy, add = (torch.randn(rows, C, device=dev) for _ in range(3))
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
_, st = ops.layernorm_fwd(x, g, b, 1e-6)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
MB = rows * C * 4 / 1e6
for name, f, mb in (("fwd fp32", lambda: ops.layernorm_fwd(x, g, b, 1e-6), 2 * MB),
                    ("fwd fp32+planes", lambda: ops.layernorm_fwd(x, g, b, 1e-6, planes=True), 3.5 * MB),
                    ("fwd planes only", lambda: ops.layernorm_fwd(x, g, b, 1e-6, planes=True, want_y=False), 2.5 * MB),
                    ("bwd", lambda: ops.layernorm_bwd(dy, x, st, g, dx_add=add), 4 * MB),
                    ("bwd+planes", lambda: ops.layernorm_bwd(dy, x, st, g, dx_add=add, planes=True), 5.5 * MB),
                    ("bwd+wgrad", lambda: ops.layernorm_bwd(dy, x, st, g, dx_add=add, want_wgrad=True), 4 * MB),
                    ("split_planes", lambda: ops.split_planes(x), 2.5 * MB)):
    t = timeit(f)
    print(f"{name:18s} {t:8.1f} us  {mb / t * 1e-3 * 1e3:6.2f} TB/s")
