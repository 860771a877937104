"""Dilated 3x3 convolution of the ASPP module ALONE (C = 128 -> 128 on 32 x 32 maps, ONE_IMGS class-images): solo time of the
whole-image fp16 x 2 kernel (conv_dil.hip) and of the implicit-GEMM path it replaces (the same pack without its planes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops, lib as L
dev = torch.device("cuda:0")
ops.set_gemm_emulation(int(os.environ.get("EMU", "6")))
n, H, W, C = int(os.environ.get("ONE_IMGS", "960")), 32, 32, 128
torch.manual_seed(0)
x = torch.randn(n * H * W, C, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.1
wf, wd = ops.pack_conv_w(w)
bare = wf.clone()
def timed(f, it=10):
    for _ in range(3):
        f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(it):
        f()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / it
for d in (6, 12, 18):
    fl = 2.0 * n * H * W * C * 9 * C
    for name, pk in (("whole-image h2", wf), ("implicit GEMM", bare)):
        ms = timed(lambda: ops.conv_fwd(x, C, n, H, W, C, pk, C, 3, 3, d, d))
        print(f"d={d:2d} {name:15s} {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF (fp32-equivalent)  path {L.load().svl_last_gemm_path()}")
