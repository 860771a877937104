"""Attention kernels alone at the bench shape (32 images x 1025 tokens x 12 heads): times fwd / bwd, or runs once under
rocprofv3 (tools/pmc_attn.sh)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
ops.set_gemm_emulation(int(os.environ.get("EMU", "0")))   # 6: the bf16 x 6 kernels
Bn, T, H = 32, 1025, 12
torch.manual_seed(0)
qkv = torch.randn(Bn * T, 3 * H * 64, device=dev)
do = torch.randn(Bn * T, H * 64, device=dev)
n = int(os.environ.get("ITERS", "3"))
for _ in range(3):
    o, lse = ops.attention_fwd(qkv, Bn, T, H)
    ops.attention_bwd(do, qkv, o, lse, Bn, T, H)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ev[0].record()
for _ in range(n):
    o, lse = ops.attention_fwd(qkv, Bn, T, H)
ev[1].record()
for _ in range(n):
    dqkv = ops.attention_bwd(do, qkv, o, lse, Bn, T, H)
ev[2].record()
torch.cuda.synchronize()
fl = 4.0 * Bn * H * T * T * 64
print(f"fwd {ev[0].elapsed_time(ev[1]) / n:.3f} ms ({fl / ev[0].elapsed_time(ev[1]) * n / 1e9:.1f} TF)  "
      f"bwd {ev[1].elapsed_time(ev[2]) / n:.3f} ms ({2.5 * fl / ev[1].elapsed_time(ev[2]) * n / 1e9:.1f} TF)")
if os.environ.get("CHECK"):
    q, k, v = (qkv.view(Bn, T, 3, H, 64)[:, :, i].permute(0, 2, 1, 3).double() for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.125
    pr = torch.softmax(s, -1)
    ref = (pr @ v).permute(0, 2, 1, 3).reshape(Bn * T, H * 64)
    print("fwd max err", (o.double() - ref).abs().max().item(), "lse err",
          (lse.view(Bn, H, T).double() - torch.logsumexp(s, -1)).abs().max().item())
    dO = do.view(Bn, T, H, 64).permute(0, 2, 1, 3).double()
    dv = pr.transpose(-1, -2) @ dO
    dp = dO @ v.transpose(-1, -2)
    ds = pr * (dp - (dp * pr).sum(-1, keepdim=True))
    dq = ds @ k * 0.125
    dk = ds.transpose(-1, -2) @ q * 0.125
    refg = torch.stack([dq, dk, dv], 2).permute(0, 3, 2, 1, 4).reshape(Bn * T, 3 * H * 64)
    print("bwd max err", (dqkv.double() - refg).abs().max().item(), "scale", refg.abs().max().item())
