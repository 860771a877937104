"""Shader clock sustained over a whole training step (svl_clock_probe waves on a second stream), and the wall time of the
step with / without the probe resident.  usage: python tools/clock_step.py"""
import os, sys, time, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from semivl_amd import ops, lib as L
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step
dev = torch.device("cuda:0")
ops.set_gemm_emulation(6)
cfg = exp40_cfg(16, 512, 21, "pascal")
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"]); red = GradAllReducer(opt)
batch = synthetic_batch(16, 512, 21, seed=1234, device=dev)
def step(i): semivl_train_step(model, batch, i, 100, cfg, optimizer=opt, reducer=red)
for i in range(3): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter(); step(3); torch.cuda.synchronize(); t_plain = time.perf_counter() - t0
n = 64
out = torch.zeros(2 * n, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(dev)
for frac in (0.8, 0.3):
    out.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    L.check(L.load().svl_clock_probe(ctypes.c_void_p(out.data_ptr()), n, int(t_plain * frac * 1e8), ctypes.c_void_p(side.cuda_stream)), "probe")
    step(4); torch.cuda.current_stream().synchronize(); t_step = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    o = out.cpu().double().view(n, 2); mhz = o[:, 0] / o[:, 1] * 100
    print(f"probe over {frac:.1f} of the step: step alone {t_plain*1e3:.1f} ms, with probe {t_step*1e3:.1f} ms (all streams {t_all*1e3:.1f} ms); "
          f"clock mean {mhz.mean():.0f} min {mhz.min():.0f} max {mhz.max():.0f} MHz; ticks {o[:,1].mean()/1e5:.1f} ms")
