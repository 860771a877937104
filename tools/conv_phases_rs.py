"""Measurement build (-DSVL_CONV_PHASE_TIMING) of the role-split tiled convolution: cycles per interval of wave 0 of each
group in its MEM phases, MFMA phases and at the barriers."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["SVL_CONV_TILED_RS"] = "1"
import torch
from semivl_amd import ops, lib as L

dev = torch.device("cuda:0")
ops.set_gemm_emulation(6)
lib = L.load()
fn = lib.svl_debug_conv_phases
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
imgs = 300
for (C, Co, Hh) in [(128, 64, 64), (64, 64, 64), (64, 32, 128), (32, 32, 128)]:
    x = torch.randn(imgs * Hh * Hh, C, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.1
    wf, wd = ops.pack_conv_w(w)
    ops.conv_fwd(x, C, imgs, Hh, Hh, C, wf, Co, 3, 3, 1, 1)
    torch.cuda.synchronize()
    fn(None, 1)
    ops.conv_fwd(x, C, imgs, Hh, Hh, C, wf, Co, 3, 3, 1, 1)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    fn(ctypes.cast(buf, ctypes.c_void_p), 0)
    nb, ni = max(1, buf[7]), max(1, buf[6])
    per = ni / nb
    names = ["epilogue", "wstore", "wload", "xstore", "xload", "reads+wait"]
    print(f"{C:3d}->{Co:3d} {Hh}x{Hh}: blocks {nb}, intervals/block {per:.0f}; group 0, cycles per interval PAIR: " +
          "  ".join(f"{n} {2 * buf[i] / ni:6.0f}" for i, n in enumerate(names)), flush=True)
