"""ConvTranspose2d(k 2, s 2) input gradient of the last Up block ALONE at the VOC step's shape (672 class-images, 64^2 -> 128^2,
48 upsampled channels in a 64-channel pixel stride, 64 input channels): solo time of whichever kernel svl_gemm_f32 picks."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
ops.set_gemm_emulation(int(os.environ.get("EMU", "6")))
n, H, W, Co, Ci, ld = int(os.environ.get("ONE_IMGS", "672")), 64, 64, 48, 64, 64
torch.manual_seed(0)
du = torch.randn(n * 4 * H * W, ld, device=dev)
wb = torch.randn(Ci, 4 * Co, device=dev) * 0.1
for _ in range(3):
    dx = ops.convT2x_dgrad(du, ld, n, H, W, Co, wb, Ci)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
it = 10
ev[0].record()
for _ in range(it):
    dx = ops.convT2x_dgrad(du, ld, n, H, W, Co, wb, Ci)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / it
gb = (du.numel() * 4 + dx.numel() * 4) / 1e9
print(f"convT2x_dgrad solo: {ms:.3f} ms  ({gb / ms:.2f} TB/s over {gb:.2f} GB, {2.0 * n * H * W * Ci * 4 * Co / ms / 1e9:.1f} TF)  path {__import__('semivl_amd.lib', fromlist=['x']).load().svl_last_gemm_path()}")
