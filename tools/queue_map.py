"""Which of the step's streams share a hardware queue (the runtime maps streams onto GPU_MAX_HW_QUEUES = 4 queues in creation
order)?  A 3 ms single-wave kernel on X followed by a tiny one on Y: if Y's kernel only finishes after X's, they share a queue.
usage: [SVL_DUMMY_STREAMS=n] python tools/queue_map.py"""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from semivl_amd import ops, lib as L
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step
import semivl_amd.train as T
dev = torch.device("cuda:0")
dummy = []
for _ in range(int(os.environ.get("SVL_DUMMY_STREAMS", "0"))):
    s_ = torch.cuda.Stream()
    with torch.cuda.stream(s_):
        torch.zeros(1, device=dev).add_(1)
    dummy.append(s_)
ops.set_gemm_emulation(6)
cfg = exp40_cfg(4, 512, 21, "pascal")
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"]); red = GradAllReducer(opt)
batch = synthetic_batch(4, 512, 21, seed=1234, device=dev)
for i in range(2):
    semivl_train_step(model, batch, i, 100, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize()
streams = {"main": torch.cuda.current_stream(), "side": T._SIDE[dev], "wgrad": ops._WG[dev.index]}
for i in range(4):
    streams[f"new{i}"] = torch.cuda.Stream()
buf = torch.zeros(4, dtype=torch.int64, device=dev)
lib = L.load()
def shares(x, y):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(x)
    L.check(lib.svl_clock_probe(ctypes.c_void_p(buf.data_ptr()), 1, 300000, ctypes.c_void_p(x.cuda_stream)), "probe")   # 3 ms
    y.wait_event(e0)
    L.check(lib.svl_clock_probe(ctypes.c_void_p(buf[2:].data_ptr()), 1, 100, ctypes.c_void_p(y.cuda_stream)), "probe")
    e1.record(y)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) > 2.0
names = list(streams)
print("dummy streams:", len(dummy))
for a in names:
    print(f"{a:6s}", " ".join(("X" if (a != b and shares(streams[a], streams[b])) else ".") for b in names), "  <-", " ".join(names))
