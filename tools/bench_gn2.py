"""GroupNorm forward / backward at the decoder's shapes (us, effective TB/s): python tools/bench_gn2.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for imgs, HW, C, G in ((1008, 1024, 128, 8), (672, 4096, 64, 4), (672, 16384, 32, 2), (456, 2601, 128, 8), (304, 10404, 64, 4), (304, 41616, 32, 2)):
    x = torch.randn(imgs * HW, C, device=dev); dy = torch.randn_like(x); y = torch.empty_like(x); dx = torch.empty_like(x)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    st = ops.groupnorm_fwd(x, C, g, b, 1e-5, imgs, HW, C, G, True, y, C)
    mb = x.numel() * 4 / 1e6
    tf = timeit(lambda: ops.groupnorm_fwd(x, C, g, b, 1e-5, imgs, HW, C, G, True, y, C))
    tb = timeit(lambda: ops.groupnorm_bwd(dy, C, x, C, y, C, st, g, imgs, HW, C, G, True, dx, C, beta=b))
    print(f"imgs {imgs:5d} HW {HW:6d} C {C:4d}: fwd {tf:8.1f} us ({3 * mb / tf:5.2f} TB/s over 3 passes)  bwd {tb:8.1f} us ({5 * mb / tb:5.2f} TB/s over 5 passes)")
    del x, dy, y, dx
