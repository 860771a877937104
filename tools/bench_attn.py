"""Fused ViT attention: exact fp32 MFMA kernels vs the bf16x6 split emulation (time, TF/s, error vs fp64).
Usage: python tools/bench_attn.py [B T H]"""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from semivl_amd import ops


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B, T, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 1025, 12)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    E = 64 * H
    qkv = torch.randn(B * T, 3 * E, device=dev, generator=g)
    do = torch.randn(B * T, E, device=dev, generator=g)
    # fp64 reference on one image
    q, k, v = [t.reshape(1, T, H, 64).transpose(1, 2).double() for t in qkv[:T].view(1, T, 3 * E).split(E, dim=2)]
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    ref = ((q * 0.125) @ k.transpose(-1, -2)).softmax(-1) @ v
    ref2 = ref.transpose(1, 2).reshape(T, E)
    gq, gk, gv = torch.autograd.grad(ref2, (q, k, v), do[:T].double())
    gref = torch.cat([t.transpose(1, 2).reshape(T, E) for t in (gq, gk, gv)], 1)
    fl_f = 4.0 * T * T * 64 * B * H
    for mode in (0, 6):
        ops.set_gemm_emulation(mode)
        out, lse = ops.attention_fwd(qkv, B, T, H)
        dqkv = ops.attention_bwd(do, qkv, out, lse, B, T, H)
        ef = float((out[:T].double() - ref2).abs().max())
        eb = float((dqkv[:T].double() - gref).abs().max())
        tf = timeit(lambda: ops.attention_fwd(qkv, B, T, H))
        tb = timeit(lambda: ops.attention_bwd(do, qkv, out, lse, B, T, H))
        print(f"mode {mode}: fwd {tf:.3f} ms ({fl_f / tf / 1e9:.1f} TF)  bwd {tb:.3f} ms ({2.5 * fl_f / tb / 1e9:.1f} TF)  "
              f"err vs fp64: fwd {ef:.3e} bwd {eb:.3e}", flush=True)
    ops.set_gemm_emulation(0)


if __name__ == "__main__":
    main()
