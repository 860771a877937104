"""One launch family for PMC passes: the dilated ASPP convolution (implicit-GEMM split kernel, gemm_bf16x_kernel<3, 2, 0>)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
ops.set_gemm_emulation(int(os.environ.get("EMU", "6")))
imgs, C, Hh, dil = int(os.environ.get("ONE_IMGS", 1200)), 128, 32, int(os.environ.get("ONE_DIL", 6))
x = torch.randn(imgs * Hh * Hh, C, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.1
wf, wd = ops.pack_conv_w(w)
for _ in range(4):
    ops.conv_fwd(x, C, imgs, Hh, Hh, C, wf, C, 3, 3, dil, dil)
torch.cuda.synchronize()
