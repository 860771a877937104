"""One-rank RCCL dry run of the data-parallel step on a one-GPU box (run under `rocprofv3 --kernel-trace`, tools/rccl1_timeline.sh):
the REAL ProcessGroupNCCL path of GradAllReducer -- told world=2 so that its multi-rank branch runs; a one-rank SUM is the
identity -- with a marker kernel on the communication stream next to every bucket, so that the trace shows which stream /
hardware queue the RCCL kernels land on and what they overlap."""
import os
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("SVL_NO_WGRAD_STREAM", "1")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29655"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    B = int(os.environ.get("RCCL1_BATCH", "4"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from semivl_amd import ops
    from semivl_amd.model.builder import build_model
    from semivl_amd.synthetic import exp40_cfg, synthetic_batch
    from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step
    cfg = exp40_cfg(B, 512, 21, "pascal")
    torch.manual_seed(1234)
    model = build_model(cfg).to(dev)
    opt = FusedAdamW(model, cfg["optimizer"])
    red = GradAllReducer(opt, world=2, profile=True)
    opt.grad_scale = 1.0
    batch = synthetic_batch(B, 512, 21, seed=1, device=dev)
    ops.set_gemm_emulation(6)
    marker = torch.zeros(1 << 20, device=dev)
    real = dist.all_reduce

    def marked(t, *a, **kw):         # a 4 MB fill on whatever stream is current when the collective is enqueued
        ops.fill(marker, 1.0)
        return real(t, *a, **kw)
    dist.all_reduce = marked
    for it in range(3):
        semivl_train_step(model, batch, it, 100, cfg, optimizer=opt, reducer=red)
    torch.cuda.synchronize()
    rep = red.timing_report()
    print("RCCL1_REPORT", rep)
    print("RCCL1_STREAMS main=%#x comm=%#x" % (torch.cuda.current_stream().cuda_stream, red._comm.cuda_stream))
    dist.all_reduce = real
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
