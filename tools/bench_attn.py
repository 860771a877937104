"""Fused attention timing at T = 1024 vs 1025 (cost of the ragged cls-token row)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


Bn, H, D = 32, 12, 64
for T in (1024, 1025, 1088, 1152):
    qkv = torch.randn(Bn * T, 3 * H * D, device=dev)
    o, lse = ops.attention_fwd(qkv, Bn, T, H)
    do = torch.randn_like(o)
    f = timeit(lambda: ops.attention_fwd(qkv, Bn, T, H))
    b = timeit(lambda: ops.attention_bwd(do, qkv, o, lse, Bn, T, H))
    fl = 4.0 * Bn * H * T * T * D
    print(f"T={T}: fwd {f:.3f} ms ({fl / f / 1e9:.1f} TF)  bwd {b:.3f} ms ({2 * fl / b / 1e9:.1f} TF algorithmic, {3.5 * fl / b / 1e9:.1f} executed)", flush=True)
