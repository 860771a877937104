"""Debug aid: fused conv3x3 + GroupNorm statistics against the two-pass path at the full-size head shapes (both arithmetics)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from semivl_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def nhwc(x): return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()
cases = [(48, 16, 32, 128, 128, 63, 21), (32, 0, 32, 128, 128, 63, 1), (96, 32, 64, 64, 64, 63, 21), (64, 0, 64, 64, 64, 63, 1),
         (48, 16, 32, 128, 128, 672, 21)]
for (C1, C2, Co, H, W, n, rep) in cases:
    a = torch.randn(n, C1, H, W, device=dev).relu() + 0.3
    b2 = torch.randn(n // rep, C2, H, W, device=dev) if C2 else None
    w = torch.randn(Co, C1 + C2, 3, 3, device=dev) * 0.05
    wf, _ = ops.pack_conv_w(w)
    kw = dict(src2=nhwc(b2), ld2=C2, C2=C2, rep=rep) if C2 else {}
    gamma, beta = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    for mode in (0, 6):
        ops.set_gemm_emulation(mode)
        pre, st = ops.conv3x3_gn(nhwc(a), C1, n, H, W, C1, wf, Co, 1e-5, **kw)
        plain = ops.conv_fwd(nhwc(a), C1, n, H, W, C1, wf, Co, 3, 3, 1, 1, **kw)
        y0 = ops.empty(n * H * W, Co, device=dev)
        st0 = ops.groupnorm_fwd(plain, Co, gamma, beta, 1e-5, n, H * W, Co, Co // 16, True, y0, Co)
        g = plain.double().view(n, H * W, Co // 16, 16)
        m64, v64 = g.mean(dim=(1, 3)), g.var(dim=(1, 3), unbiased=False)
        r64 = (v64 + 1e-5).rsqrt()
        print(f"C1={C1} C2={C2} Co={Co} {H}x{W} n={n} mode {mode}: pre equal {torch.equal(pre, plain)}; stats differ in {int((st != st0).sum())} of {st.numel()}; "
              f"mean err fused {((st[..., 0].double() - m64).abs() / (m64.abs() + v64.sqrt())).max().item():.2e} two-pass {((st0[..., 0].double() - m64).abs() / (m64.abs() + v64.sqrt())).max().item():.2e}; "
              f"rstd rel err fused {((st[..., 1].double() - r64).abs() / r64).max().item():.2e} two-pass {((st0[..., 1].double() - r64).abs() / r64).max().item():.2e}; "
              f"mean/std max {(m64.abs() / v64.sqrt()).max().item():.2f}")
    ops.set_gemm_emulation(0)
