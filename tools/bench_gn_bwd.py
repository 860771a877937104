"""GroupNorm backward (sums + apply) at the decoder's class-image counts: HIP-event timed, bytes per pass."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops

dev = torch.device("cuda:0")
imgs = int(os.environ.get("IMGS", 1200))
for (C, Hh) in [(32, 128), (64, 64), (128, 32)]:
    HW = Hh * Hh
    x = torch.randn(imgs * HW, C, device=dev)
    dy = torch.randn(imgs * HW, C, device=dev)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    y = ops.empty(imgs * HW, C, device=dev)
    st = ops.groupnorm_fwd(x, C, gamma, beta, 1e-5, imgs, HW, C, C // 16, True, y, C)
    dx = ops.empty(imgs * HW, C, device=dev)
    f = lambda: ops.groupnorm_bwd(dy, C, x, C, None, C, st, gamma, imgs, HW, C, C // 16, True, dx, C, beta=beta)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = imgs * HW * C * 4 * 5 / 1e9
    print(f"groupnorm_bwd C={C} {Hh}x{Hh} x {imgs}: {ms:.3f} ms  ({gb:.2f} GB of passes: {gb / ms:.2f} TB/s)", flush=True)
