"""GroupNorm fwd / bwd at the VLG-head shapes vs the HBM bound (fwd: read x twice + write y; bwd: read dy, x twice + write dx)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
for imgs, HW, C, G in [(168, 4096, 128, 8), (168, 4096, 64, 4), (672, 4096, 64, 4), (672, 16384, 32, 2), (1008, 4096, 128, 8)]:
    x = torch.randn(imgs * HW, C, device=dev)
    dy = torch.randn_like(x)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    def t(fn, n=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    st = ops.groupnorm_fwd(x, C, g, b, 1e-5, imgs, HW, C, G, True, y, C)
    tf = t(lambda: ops.groupnorm_fwd(x, C, g, b, 1e-5, imgs, HW, C, G, True, y, C))
    tb = t(lambda: ops.groupnorm_bwd(dy, C, x, C, y, C, st, g, imgs, HW, C, G, True, dx, C))
    by = x.numel() * 4.0
    print(f"imgs={imgs} HW={HW} C={C}: tensor {by / 1e6:.0f} MB  fwd {tf:.3f} ms = {3 * by / tf / 1e6:.0f} GB/s (3 passes)   "
          f"bwd {tb:.3f} ms = {5 * by / tb / 1e6:.0f} GB/s (5 passes)")
