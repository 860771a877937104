"""Three SemiVL steps on one GPU with GradAllReducer's STREAM-ORDERED branch live (world = 2 stand-in): the injected
collective launches REAL kernels on the communication stream -- a bucket-sized copy (permute_rows_kernel: at N = 21 classes no
other launch of the step uses it) followed by the x2 that "SUM over two ranks with identical gradients" amounts to (cancelled exactly by
AdamW's 1 / W).  Run under rocprofv3 --kernel-trace by tools/comm_overlap_trace.sh; tools/comm_overlap_parse.py then checks
from the trace that those kernels (1) sit on a hardware queue none of the step's other kernels use and (2) run while
backward kernels of the step are in flight."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")        # what every rank of an N > 1 run gets (bench.py)
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from semivl_amd import ops
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step

B = int(os.environ.get("SVL_COMM_B", "8"))
dev = torch.device("cuda", 0)
cfg = exp40_cfg(B, 512, 21, "pascal")
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"])


class _Done:
    def wait(self):
        return True


def collective(g):
    flat = g.view(-1)
    n4 = flat.numel() // 4 * 4
    ops.permute_rows(flat[:n4], 1, 1, n4 // 4, 4)                    # the "wire": one bucket-sized pass on the communication stream
    ops.add(flat, flat, out=flat)                                    # SUM over two identical ranks
    return _Done()


red = GradAllReducer(opt, world=2, collective=collective, profile=True)
ops.set_gemm_emulation(6)
batch = synthetic_batch(B, 512, 21, seed=1234, device=dev)
for i in range(3):
    losses = semivl_train_step(model, batch, i, 1000, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize()
rep = red.timing_report()
print("loss", float(losses[0].item()), "buckets launched inside backward:", rep["launched_inside_backward"], "of", len(rep["buckets"]),
      "exposed_ms", rep["exposed_ms"], "queues", rep["queues"])
