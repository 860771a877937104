"""Micro-benchmark of svl_gemm_f32 on the step's dominant shapes (HIP-event timed)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, flops, name, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:44s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TF/s", flush=True)


M = 32800
for (N, K) in [(768, 3072), (3072, 768), (2304, 768), (768, 768), (512, 768)]:
    x, w = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    timeit(lambda: ops.linear(x, w), 2.0 * M * N * K, f"linear KC/KC M={M} N={N} K={K}")
for (N, K) in [(3072, 768), (768, 3072)]:
    dy, w = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    timeit(lambda: ops.matmul_nn(dy, w), 2.0 * M * N * K, f"dgrad  KC/NC M={M} N={N} K={K}")
for (Mo, N) in [(768, 768), (2304, 768)]:
    dy, x = torch.randn(M, Mo, device=dev), torch.randn(M, N, device=dev)
    timeit(lambda: ops.matmul_tn(dy, x), 2.0 * M * N * Mo, f"wgrad  MC/NC M={Mo} N={N} K={M} (split-K)")
Bn, T, H, D = 32, 1025, 12, 64
qkv = torch.randn(Bn * T, 3 * H * D, device=dev)
timeit(lambda: ops.vit_attention_fwd(qkv, Bn, T, H, D), 4.0 * Bn * H * T * T * D, "attention fwd (QK^T, softmax, PV) b=32")
# decoder convs (b'=64 -> imgs=1344 is big; use imgs=336 = 16*21)
imgs = 336
for (C, Co, Hh, dil) in [(128, 128, 32, 6), (128, 64, 64, 1), (64, 64, 64, 1), (64, 32, 128, 1), (32, 32, 128, 1)]:
    x = torch.randn(imgs * Hh * Hh, C, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev)
    wf, wd = ops.pack_conv_w(w)
    fl = 2.0 * imgs * Hh * Hh * Co * C * 9
    timeit(lambda: ops.conv_fwd(x, C, imgs, Hh, Hh, C, wf, Co, 3, 3, dil, dil), fl, f"conv3x3 fwd C={C}->{Co} {Hh}x{Hh} d={dil}")
    dy = torch.randn(imgs * Hh * Hh, Co, device=dev)
    timeit(lambda: ops.conv_dgrad(dy, Co, imgs, Hh, Hh, Co, wd, C, 3, 3, dil, dil), fl, f"conv3x3 dgrad")
    timeit(lambda: ops.conv_wgrad(dy, Co, x, C, imgs, Hh, Hh, C, Co, 3, 3, dil, dil), fl, f"conv3x3 wgrad")
timeit(lambda: ops.attention_fwd(qkv, Bn, T, H), 4.0 * Bn * H * T * T * D, "FUSED attention fwd b=32")
o_, lse_ = ops.attention_fwd(qkv, Bn, T, H)
do_ = torch.randn_like(o_)
timeit(lambda: ops.attention_bwd(do_, qkv, o_, lse_, Bn, T, H), 8.0 * Bn * H * T * T * D, "FUSED attention bwd b=32 (alg. 8*B*H*T^2*D)")
