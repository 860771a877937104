"""Accuracy (vs fp64) and rate of the dense GEMM in its three arithmetic modes: exact fp32 MFMA, bf16x6, bf16x3."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def relerr(y, ref):
    return float(((y.double() - ref).norm() / ref.norm()))


torch.manual_seed(0)
M = 32800
cases = []
for (N, K) in [(768, 3072), (3072, 768), (2304, 768), (768, 768)]:
    x, w = torch.randn(M, K, device=dev), torch.nn.Parameter(torch.randn(N, K, device=dev) * 0.02)
    cases.append((f"linear KC/KC M={M} N={N} K={K}", lambda x=x, w=w: ops.linear(x, w), lambda x=x, w=w: x.double() @ w.double().t(), 2.0 * M * N * K))
for (N, K) in [(3072, 768), (768, 3072)]:
    dy, w = torch.randn(M, K, device=dev), torch.nn.Parameter(torch.randn(K, N, device=dev) * 0.02)
    cases.append((f"dgrad  KC/NC M={M} N={N} K={K}", lambda dy=dy, w=w: ops.matmul_nn(dy, w), lambda dy=dy, w=w: dy.double() @ w.double(), 2.0 * M * N * K))
for (Mo, N) in [(768, 768), (2304, 768), (768, 3072)]:
    dy, x = torch.randn(M, Mo, device=dev), torch.randn(M, N, device=dev)
    cases.append((f"wgrad  MC/NC M={Mo} N={N} K={M}", lambda dy=dy, x=x: ops.matmul_tn(dy, x), lambda dy=dy, x=x: dy.double().t() @ x.double(), 2.0 * M * N * Mo))
# ragged: M, N, K not multiples of the tile
x, w = torch.randn(1000, 333, device=dev), torch.randn(200, 333, device=dev)
cases.append(("linear ragged M=1000 N=200 K=333", lambda x=x, w=w: ops.linear(x, w), lambda x=x, w=w: x.double() @ w.double().t(), 2.0 * 1000 * 200 * 333))

for name, fn, ref_fn, fl in cases:
    ref = ref_fn()
    line = f"{name:40s}"
    for mode, planes in ((0, True), (6, True), (6, False), (3, True)):   # 6 + planes: pre-split operands (gemm_planes.hip)
        ops.set_gemm_emulation(mode)
        ops.PLANES_PATH = planes
        y = fn()
        ms = timeit(fn)
        tag = f"m{mode}" + ("" if mode != 6 else ("p" if planes else "r"))
        line += f" | {tag}: {ms:7.3f} ms {fl / ms / 1e9:6.1f} TF err {relerr(y, ref):.2e}"
    ops.PLANES_PATH = False
    ops.set_gemm_emulation(0)
    print(line, flush=True)

# the pre-split pipeline pieces alone: split pass, GEMM on ready planes, GEMM that emits planes
ops.set_gemm_emulation(6)
ops.PLANES_PATH = True
with torch.no_grad():
    x = torch.randn(M, 768, device=dev)
    w1 = torch.nn.Parameter(torch.randn(3072, 768, device=dev) * 0.02)
    w2 = torch.nn.Parameter(torch.randn(768, 3072, device=dev) * 0.02)
    b1 = torch.zeros(3072, device=dev)
    xp = ops.split_planes(x)
    pre = torch.empty(M, 3072, device=dev)
    print("split pass   [M,768]           %7.3f ms" % timeit(lambda: ops.split_planes(x)))
    print("FFN-1 planes in, fp32 out      %7.3f ms  %6.1f TF" % ((lambda t: (t, 2.0 * M * 3072 * 768 / t / 1e9))(timeit(lambda: ops.linear(xp, w1, b1)))))
    print("FFN-1 planes in, GELU+preact, planes out %7.3f ms" % timeit(lambda: ops.linear(xp, w1, b1, act=ops.ACT_GELU, preact=pre, planes_only=True)))
    hp = ops.linear(xp, w1, b1, act=ops.ACT_GELU, planes_only=True)
    print("FFN-2 planes in, fp32 out      %7.3f ms  %6.1f TF" % ((lambda t: (t, 2.0 * M * 3072 * 768 / t / 1e9))(timeit(lambda: ops.linear(hp, w2)))))
ops.set_gemm_emulation(0)
