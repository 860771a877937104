"""Short-K linears of the VLG head (K = 64 / 128, millions of rows): time vs the HBM bound."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from semivl_amd import ops
dev = torch.device("cuda:0")
shapes = [(4128768, 192, 64), (2752512, 192, 64), (1032192, 384, 128), (688128, 384, 128), (688128, 128, 128),
          (688128, 640, 128), (2752512, 64, 192), (4128768, 64, 64), (1000003, 200, 64)]
for M, N, K in shapes:
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(2):
        ops.linear(x, w, b, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.linear(x, w, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    by = 4.0 * M * (N + K)
    err = 0.0
    for sl in (slice(0, 4096), slice(M - 4099, M)):
        ref = (x[sl].double() @ w.double().t() + b.double())
        err = max(err, (out[sl].double() - ref).abs().max().item())
    print(f"M={M:8d} N={N:4d} K={K:4d}  {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:6.1f} TF  {by / ms / 1e6:7.0f} GB/s  "
          f"(HBM bound {by / 6.0e9:6.3f} ms at 6 TB/s)  err {err:.2e}")
    del x, out
