"""Measurement build only (conv_tiled.hip compiled with -DSVL_CONV_PHASE_TIMING): where wave 0 of every block of the tiled
bf16x6 convolution spends its cycles -- s_memtime deltas summed over blocks: prologue | MFMA phase | barrier 1 | staging |
barrier 2 | epilogue."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops, lib as L

dev = torch.device("cuda:0")
ops.set_gemm_emulation(6)
lib = L.load()
fn = lib.svl_debug_conv_phases
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["prologue", "mfma", "barrier1", "stage", "barrier2", "epilogue"]
imgs = 300
for (C, Co, Hh) in [(128, 64, 64), (64, 64, 64), (64, 32, 128), (32, 32, 128)]:
    x = torch.randn(imgs * Hh * Hh, C, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.1
    wf, wd = ops.pack_conv_w(w)
    for tag, pk in (("planes", wf), ("split ", wf.clone())):
        ops.conv_fwd(x, C, imgs, Hh, Hh, C, pk, Co, 3, 3, 1, 1)
        torch.cuda.synchronize()
        fn(None, 1)
        ops.conv_fwd(x, C, imgs, Hh, Hh, C, pk, Co, 3, 3, 1, 1)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        fn(ctypes.cast(buf, ctypes.c_void_p), 0)
        nb = max(1, buf[7])
        tot = sum(buf[i] for i in range(6))
        print(f"{C:3d}->{Co:3d} {Hh}x{Hh} {tag}: blocks {nb}  cycles/block {tot / nb:9.0f}  " +
              "  ".join(f"{n} {buf[i] / nb:7.0f} ({100.0 * buf[i] / tot:4.1f}%)" for i, n in enumerate(names)), flush=True)
