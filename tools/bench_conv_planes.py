"""Tiled 3x3 convolutions of the Up blocks (ADE class-image counts) with and without the pre-split weight planes:
HIP-event timed, same process, bit-equality checked."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops

dev = torch.device("cuda:0")
ops.set_gemm_emulation(6)


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


imgs = int(os.environ.get("IMGS", 300))
for (C, Co, Hh) in [(128, 64, 64), (64, 64, 64), (64, 32, 128), (32, 32, 128)]:
    x = torch.randn(imgs * Hh * Hh, C, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * 0.1
    wf, wd = ops.pack_conv_w(w)
    bare_f, bare_d = wf.clone(), wd.clone()
    dy = torch.randn(imgs * Hh * Hh, Co, device=dev)
    fl = 2.0 * imgs * Hh * Hh * Co * C * 9
    for name, a, b in (("fwd", wf, bare_f),):
        y1 = ops.conv_fwd(x, C, imgs, Hh, Hh, C, a, Co, 3, 3, 1, 1)
        y0 = ops.conv_fwd(x, C, imgs, Hh, Hh, C, b, Co, 3, 3, 1, 1)
        t1 = timeit(lambda: ops.conv_fwd(x, C, imgs, Hh, Hh, C, a, Co, 3, 3, 1, 1))
        t0 = timeit(lambda: ops.conv_fwd(x, C, imgs, Hh, Hh, C, b, Co, 3, 3, 1, 1))
        print(f"conv3x3 {name} {C:3d}->{Co:3d} {Hh}x{Hh}: split in kernel {t0:7.3f} ms {fl / t0 / 1e9:6.1f} TF | planes {t1:7.3f} ms "
              f"{fl / t1 / 1e9:6.1f} TF | equal {torch.equal(y0, y1)}", flush=True)
    if C in (32, 64):
        d1 = ops.conv_dgrad(dy, Co, imgs, Hh, Hh, Co, wd, C, 3, 3, 1, 1)
        d0 = ops.conv_dgrad(dy, Co, imgs, Hh, Hh, Co, bare_d, C, 3, 3, 1, 1)
        t1 = timeit(lambda: ops.conv_dgrad(dy, Co, imgs, Hh, Hh, Co, wd, C, 3, 3, 1, 1))
        t0 = timeit(lambda: ops.conv_dgrad(dy, Co, imgs, Hh, Hh, Co, bare_d, C, 3, 3, 1, 1))
        print(f"conv3x3 dgrad {Co:3d}->{C:3d} {Hh}x{Hh}: split in kernel {t0:7.3f} ms {fl / t0 / 1e9:6.1f} TF | planes {t1:7.3f} ms "
              f"{fl / t1 / 1e9:6.1f} TF | equal {torch.equal(d0, d1)}", flush=True)
