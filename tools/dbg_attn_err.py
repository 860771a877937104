"""Per-output error of the fused attention against fp64 on the spiky test case (tests/test_ops_gpu.py::test_fused_attention_split_emulation)."""
import os, sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from semivl_amd import ops


def case(Bn, T, H, seed=51):
    dev = torch.device("cuda:0")
    D, E = 64, 64 * H
    g = torch.Generator(device="cpu").manual_seed(seed)
    qkv = torch.randn(Bn * T, 3 * E, generator=g).to(dev)
    qkv[:, :2 * E] *= 2.0
    if T > 40:
        qkv.view(Bn, T, 3 * E)[0, T - 3, E:E + 64] = 6.0 * qkv.view(Bn, T, 3 * E)[0, 5, 0:64]
    do = torch.randn(Bn * T, E, generator=g).to(dev)
    qd = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(Bn, T, H, D).transpose(1, 2) for t in qd.view(Bn, T, 3 * E).split(E, dim=2)]
    sc = (q * D ** -0.5) @ k.transpose(-1, -2)
    ref = (sc.softmax(-1) @ v).transpose(1, 2).reshape(Bn * T, E)
    (gr,) = torch.autograd.grad(ref, qd, do.double())
    return qkv, do, ref.detach(), gr, float(sc.abs().max())


for (Bn, T, H) in [(2, 1025, 12), (1, 2602, 2), (3, 17, 4), (2, 129, 3), (1, 161, 1), (1, 97, 2)]:
    qkv, do, ref, gr, smax = case(Bn, T, H)
    E = 64 * H
    row = []
    for mode in (0, 6):
        ops.set_gemm_emulation(mode)
        out, lse = ops.attention_fwd(qkv, Bn, T, H)
        dqkv = ops.attention_bwd(do, qkv, out, lse, Bn, T, H)
        e = [float((out.double() - ref).abs().max())] + [float((dqkv[:, i * E:(i + 1) * E].double() - gr[:, i * E:(i + 1) * E]).abs().max()) for i in range(3)]
        row.append(e)
    ops.set_gemm_emulation(0)
    print(f"B{Bn} T{T} H{H} max|s| {smax:.0f}: fp32 out/dq/dk/dv " + " ".join(f"{x:.2e}" for x in row[0]) + " | emu " +
          " ".join(f"{x:.2e}" for x in row[1]) + " | ratio " + " ".join(f"{b / a:.2f}" for a, b in zip(row[0], row[1])), flush=True)
