"""Tiled 3x3 weight gradient at the VLG head's shapes: fp32 kernel vs the bf16x6 kernel (ms, TF fp32-equivalent)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")


def t(fn, n=5):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for imgs, HW, Ct, Co in [(672, 64, 128, 64), (672, 128, 64, 32), (672, 64, 64, 64), (672, 128, 32, 32)]:
    x = torch.randn(imgs * HW * HW, Ct, device=dev)
    dy = torch.randn(imgs * HW * HW, Co, device=dev)
    fl = 2.0 * Co * 9 * Ct * imgs * HW * HW
    line = f"imgs={imgs} {HW}x{HW} Ct={Ct} Co={Co}:"
    for mode in (0, 6):
        ops.set_gemm_emulation(mode)
        ms = t(lambda: ops.conv_wgrad(dy, Co, x, Ct, imgs, HW, HW, Ct, Co, 3, 3, 1, 1))
        line += f"  mode {mode}: {ms:.3f} ms {fl / ms / 1e9:.1f} TF"
    print(line, flush=True)
ops.set_gemm_emulation(0)
