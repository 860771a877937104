import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
ops.set_gemm_emulation(int(os.environ.get("EMU", "0")))   # 6: the bf16 x 6 kernels
C, Co, Hh, imgs = int(os.environ.get("ONE_C", 32)), int(os.environ.get("ONE_CO", 32)), int(os.environ.get("ONE_HW", 128)), int(os.environ.get("ONE_IMGS", 336))
x = torch.randn(imgs * Hh * Hh, C, device=dev)
w = torch.randn(Co, C, 3, 3, device=dev)
wf, wd = ops.pack_conv_w(w)
dy = torch.randn(imgs * Hh * Hh, Co, device=dev)
mode = os.environ.get("ONE_MODE", "fwd")
for _ in range(4):
    if mode == "fwd":
        ops.conv_fwd(x, C, imgs, Hh, Hh, C, wf, Co, 3, 3, 1, 1)
    else:
        ops.conv_wgrad(dy, Co, x, C, imgs, Hh, Hh, C, Co, 3, 3, 1, 1)
torch.cuda.synchronize()
