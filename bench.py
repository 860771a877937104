#!/usr/bin/env python
"""SemiVL training-step benchmark on MI355X (BASELINE.json metric: train images/sec, 512^2, ViT-B/16).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full iteration of semivl.py:223-345 on one synthetic batch per rank: CutMix, pseudo-label
forward, MaskCLIP guidance on the frozen CLIP encoder, the two grad-carrying forwards (need_fp + strong), the fused
per-pixel losses, backward through the VLG head and the ViT, gradient all-reduce (RCCL) and the fused AdamW step.
images/s := (B_labeled + B_unlabeled) * world / t_step = 2*B*W / t_step, inputs resident in HBM.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: must be in the environment before the HIP/HSA runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
AS_MULTI = "--as-multi" in sys.argv     # one GPU, but with the stream / queue settings every rank of an N > 1 run gets
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or AS_MULTI:
    # (an explicit SVL_WGRAD_STREAM=1 keeps the weight-gradient stream on there: A/B runs)
    # The runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES = 4 hardware queues in creation order, and two
    # streams on one queue are serialised (tools/queue_map.py).  A rank of a multi-GPU job owns main + second +
    # helper + weight-gradient + communication streams (+ the communicator's own): with 4 queues the bucketed all-reduce can
    # end up BEHIND the backward kernels it is meant to overlap.  Eight queues give every stream its own; the weight-gradient
    # stream is switched off there, because truly concurrent with the chain's GEMMs it costs 20 ms (measured on one GPU).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if not os.environ.get("SVL_WGRAD_STREAM"):
        os.environ.setdefault("SVL_NO_WGRAD_STREAM", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per step per unit batch, VOC N=21, 512^2
VIT_GF_PER_UNIT_B = 2866.6      # 5 ViT fwd (231.0) + 2 frozen fwd (209.2) + 4 bwd (323.3) GFLOP
DEC_GF_PER_UNIT_B = 1322.0      # 7 decoder fwd + 6 x 2 fwd-equivalents bwd, 69.6 GFLOP/img
# (nclass, crop) -> (ViT, decoder) algorithmic GFLOP per step per unit batch
ALGO_GF = {(21, 512): (2866.6, 1322.0), (81, 512): (2866.6, 5075.0), (150, 512): (2866.6, 9412.0),
           (19, 801): (5 * 725.0 + 2 * 669.8 + 4 * 1098.1, 3027.0)}
# Of SURVEY's 19 decoder forward-equivalents per image pair (7 forwards + 6 backwards x 2) the step EXECUTES 14: the
# feature-perturbed copy of the labeled half is never decoded (semivl.py:247 discards it) and the detached pred_w is not
# back-propagated (semivl.py:251): 6 forwards + 4 backwards x 2.  The ViT figure is what is executed.
DEC_EXECUTED_SHARE = 14.0 / 19.0
PEAK_F32_MFMA_TF = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TF = 2500.0      # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
PEAK_HBM_GBS = 8000.0
PEAK_CLOCK_MHZ = 2400.0         # the clock the peaks above are quoted at


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="labeled (= unlabeled) images per GPU per step")
    ap.add_argument("--crop", type=int, default=512)
    ap.add_argument("--nclass", type=int, default=21)
    ap.add_argument("--config", choices=["voc", "cityscapes", "ade", "coco"], default=None,
                    help="BASELINE.json config presets: voc = configs[1] (N=21, 512, bs 16); cityscapes = configs[2] "
                         "(N=19, 801, bs 8, skr04); ade = configs[3] (N=150, bs 16); coco = configs[4] (N=81, bs 16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, nargs=2, default=(3, 5), metavar=("WARM", "TIMED"),
                    help="oracle steps on ALL physical host cores: SURVEY §8(d)'s protocol, 3 warm-up + 5 timed, median "
                         "(~7.5 min on the 128-core host of the pool); `1 1` for a quick line")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--as-multi", action="store_true",
                    help="single GPU with the settings of a multi-GPU rank (GPU_MAX_HW_QUEUES=8, no weight-gradient stream): "
                         "the N=1 anchor a scaling curve should be divided by")
    ap.add_argument("--no-multi-anchor", action="store_true",
                    help="skip the `n1_same_settings` leg of the default N=1 run (a child process re-measuring the step "
                         "with --as-multi)")
    ap.add_argument("--gemm-arith", choices=["f32", "bf16x6", "bf16x3"], default="bf16x6",
                    help="arithmetic of the large dense GEMMs in the timed region (include/semivl_hip.h, "
                         "svl_set_gemm_emulation); 'value' is always measured in this mode.  bf16x6 (default): fp32 "
                         "in / out / accumulate, every operand split into 3 bf16 terms, 6 cross products -- error vs fp64 "
                         "at the level of the plain fp32 MFMA chain's on every operand layout (tests/test_ops_gpu.py: <= 1.2x, measured 0.85-1.15x); f32: "
                         "v_mfma_f32_32x32x2_f32 everywhere (reported next to `value` as `exact_f32`); bf16x3 is a "
                         "16-bit-product mode for experiments and is never the default")
    ap.add_argument("--no-throughput-mode", "--no-second-mode", dest="no_throughput_mode", action="store_true",
                    help="skip the extra measurement of the other arithmetic (exact f32 next to a bf16x6 value and vice versa)")
    a = ap.parse_args()
    if a.config is not None:
        a.nclass, a.crop, a.batch = {"voc": (21, 512, 16), "cityscapes": (19, 801, 8), "ade": (150, 512, 16),
                                     "coco": (81, 512, 16)}[a.config]
    return a


def respawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) through torch.distributed.run, the same
    way the driver does, and relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.exit(subprocess.call(cmd, env=env))


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(crop, nclass, warm, timed):
    """the oracle restatement (kind 'port': the reference's Python cannot travel to the GPU box) on BASELINE
    configs[0] -- VOC N=21, 512^2, bs=2, world_size 1: full steps (forward, losses, backward, AdamW), `warm` warm-up +
    `timed` timed, median.  Run with torch.set_num_threads(all physical cores) as the protocol says AND with 32 threads
    (PyTorch's CPU kernels on this many-class-image workload scale negatively on a 2-socket 128-core host); `value` is
    the FASTER of the two, i.e. the figure most favourable to the CPU; both are reported."""
    from oracle import semivl_oracle as O
    torch.manual_seed(0)
    bs = 2
    text, mcc = O.synthetic_text(nclass), O.synthetic_text(nclass, seed=8)
    model = O.build_vlm(dict(nclass=nclass, crop=crop), text, mcc)
    model.backbone.init_weights_()
    model.clip_encoder.init_weights_()
    batch = O.synthetic_batch(bs, crop, nclass, seed=1234)
    params = [p for n, p in model.named_parameters() if p.requires_grad and not n.startswith("clip_encoder")]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)

    def run(threads, nwarm, ntimed):
        torch.set_num_threads(threads)
        times = []
        for i in range(nwarm + ntimed):
            t0 = time.time()
            loss, _ = O.semivl_step(model, batch, i, 100)
            opt.zero_grad()
            loss.backward()
            opt.step()
            times.append(time.time() - t0)
        tt = sorted(times[nwarm:])
        med = tt[len(tt) // 2] if len(tt) % 2 else 0.5 * (tt[len(tt) // 2 - 1] + tt[len(tt) // 2])
        return med, sum(times)

    cores = physical_cores()
    runs, total = {}, 0.0
    # the protocol run at all physical cores (`value`), and a short run at 32 threads beside it (PyTorch's CPU kernels scale
    # negatively on this workload: the figure most favourable to the CPU is reported as `best_thread_count`)
    for threads, (w_, t_) in ((cores, (warm, timed)), (min(32, cores), (1, 2))):
        if threads in runs:
            continue
        med, spent = run(threads, w_, t_)
        runs[threads] = dict(threads=threads, s_per_step=round(med, 2), images_per_s=round(2.0 * bs / med, 4),
                             protocol=f"{w_} warm-up + {t_} timed, median")
        total += spent
    prim = runs[cores]                                   # torch.set_num_threads(all physical cores)
    best = min(runs.values(), key=lambda r: r["s_per_step"])
    return dict(value=prim["images_per_s"], unit="images/s", cores=prim["threads"], kind="port", cpu=cpu_model(),
                physical_cores=cores, s_per_step=prim["s_per_step"], runs=list(runs.values()),
                best_thread_count=dict(threads=best["threads"], images_per_s=best["images_per_s"], s_per_step=best["s_per_step"]),
                sample=f"protocol: {prim['protocol']} at {prim['threads']} threads (= `value`); "
                       f"full SemiVL steps of oracle/semivl_oracle.py (PyTorch CPU fp32) at VOC N={nclass}, {crop}x{crop}, "
                       f"bs={bs} ({2 * bs} images/step, BASELINE configs[0]) on {cpu_model()} ({cores} physical cores): `value` = "
                       f"all {cores} physical cores, {prim['protocol']} (SURVEY §8(d)'s thread setting and step counts by default); "
                       f"PyTorch's CPU kernels scale negatively on this "
                       f"workload, the best thread count tried is reported beside it: "
                       f"{', '.join(str(r['threads']) + ' thr -> ' + str(r['s_per_step']) + ' s/step' for r in runs.values())}; "
                       f"{total:.0f} s of CPU work")


def clock_probe_mhz(fn, est_ms, dev, n_waves=64):
    """Mean shader clock (MHz) while fn() runs on the current stream: svl_clock_probe waves on a second stream sample the
    shader-clock counter against the 100 MHz counter over 80 % of the estimated duration."""
    import ctypes
    import torch
    from semivl_amd import lib as L
    out = torch.zeros(2 * n_waves, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    L.check(L.load().svl_clock_probe(ctypes.c_void_p(out.data_ptr()), n_waves, int(est_ms * 0.8 * 1e5),
                                     ctypes.c_void_p(side.cuda_stream)), "svl_clock_probe")
    fn()
    torch.cuda.synchronize()
    o = out.cpu().double().view(n_waves, 2)
    if (o[:, 1] <= 0).any():
        return None
    return round(float((o[:, 0] / o[:, 1]).mean()) * 100.0, 0)


def pmc_traffic_record(batch, name="pmc_gemm_traffic.json", src="gemm.hip"):
    """Fabric-side bytes of the dominant launch from a committed rocprofv3 --pmc record (profiles/pmc_x6p_traffic.json for
    the packed-planes bf16x6 kernel, written by tools/pmc_x6p.sh; pmc_gemm_traffic.json for the exact fp32 kernel, written
    by tools/pmc_traffic.sh): used only when it was measured on the SAME kernel source (sha256 of the .hip file)."""
    import hashlib
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        sha = hashlib.sha256(open(os.path.join(ROOT, "semivl_amd", "csrc", src), "rb").read()).hexdigest()[:16]
        if rec.get("src_sha16", rec.get("gemm_hip_sha16")) == sha and rec.get("M") == 32 * 1025 * batch // 16:
            return rec["traffic_bytes"], rec.get("note", "")
    except (OSError, ValueError, KeyError):
        pass
    return None, f"no PMC record for this kernel source (profiles/{name} absent or from another {src})"


def rocprof_dispatch_record(batch, src="gemm_planes_impl.h"):
    """Per-dispatch durations of the dominant launch as rocprofv3 --kernel-trace timed them inside the step of the same
    command (profiles/dominant_dispatches.json, written by tools/round_end_run.sh from the committed per-dispatch rows):
    used only when it was measured on the SAME kernel source."""
    import hashlib
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "dominant_dispatches.json")))
        sha = hashlib.sha256(open(os.path.join(ROOT, "semivl_amd", "csrc", src), "rb").read()).hexdigest()[:16]
        if rec.get("src_sha16") == sha and rec.get("M") == 32 * 1025 * batch // 16:
            return rec
    except (OSError, ValueError, KeyError):
        pass
    return None


def inloop_dispatch_record():
    """Mean in-step kernel durations of the attention launches from the committed rocprofv3 --kernel-trace of this command
    (profiles/inloop_dispatches.json, written by tools/inloop_record.py): used only when measured on the same sources."""
    import hashlib
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "inloop_dispatches.json")))
        sha = hashlib.sha256(open(os.path.join(ROOT, "semivl_amd", "csrc", "attn_h2.hip"), "rb").read()).hexdigest()[:16]
        if rec.get("attn_h2_sha16") == sha:
            return rec
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        respawn_ranks(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; the product path has no CPU fallback")
    local = local % torch.cuda.device_count()  # (lets a 1-GPU box exercise the N>1 code path with SVL_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("SVL_DIST_BACKEND", "nccl")  # 'nccl' == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from semivl_amd import ops
    from semivl_amd.model.builder import build_model
    from semivl_amd.synthetic import exp40_cfg, synthetic_batch
    from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step

    dataset = {21: "pascal", 81: "coco", 150: "ade", 19: "cityscapes"}.get(a.nclass, "pascal")
    cfg = exp40_cfg(a.batch, a.crop, a.nclass, dataset)
    torch.manual_seed(1234)  # identical random init on every rank (no pretrained CLIP offline)
    model = build_model(cfg).to(dev)
    opt = FusedAdamW(model, cfg["optimizer"])
    red = GradAllReducer(opt)
    red.profile = world > 1       # per-bucket all-reduce time + the un-overlapped remainder go into the JSON line
    red.broadcast_params()
    batch = synthetic_batch(a.batch, a.crop, a.nclass, seed=1234 + rank, device=dev)
    total_iters = 10000

    def step(i, **kw):
        return semivl_train_step(model, batch, i, total_iters, dict(cfg, **kw) if kw else cfg, optimizer=opt, reducer=red)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nwarm, nsteps, first):
        for i in range(nwarm):
            step(first + i)
        sync()
        t0 = time.perf_counter()
        for i in range(nsteps):
            losses = step(first + nwarm + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, losses

    EMU = {"f32": 0, "bf16x6": 6, "bf16x3": 3}
    ops.set_gemm_emulation(EMU[a.gemm_arith])
    dt, losses = timed(a.warmup, a.steps, 0)
    loss_val = float(losses[0].item())
    ms = dt / a.steps * 1e3
    ips = 2.0 * a.batch * world / (dt / a.steps)

    out = dict(metric="train images/sec (512^2, ViT-B/16)", value=round(ips, 3), unit="images/s", n_gpus=world,
               steps=a.steps, warmup=a.warmup, ms_per_step=round(ms, 2), higher_is_better=True, scaling="weak",
               vs_baseline=None,
               dtype="f32" if a.gemm_arith == "f32" else
               "f32 storage and accumulation; the large GEMMs' products from TWO bf16 terms per operand (three 16-bit products per "
               "fp32 MAC, ~4e-6 relative per GEMM): a reduced-precision build-only mode, see mode_label" if a.gemm_arith == "bf16x3" else
               f"f32 (fp32 in / out / accumulate everywhere; the MFMA products as split 16-bit terms with fp32 accumulation: the ViT's "
               f"linears, in_proj weight gradients and fused attention and the decoder's 3x3 convolutions (tiled: forward, input and "
               f"weight gradient; dilated: forward, input gradient) on fp16 x 2 operands with power-of-two scales (3 products per fp32 "
               f"MAC), the other weight gradients, the ConvTranspose streams and narrow GEMMs on bf16 x 3 terms (6 products: --gemm-arith "
               f"{a.gemm_arith}), the remaining MFMA kernels on the fp32 pipe.  Accuracy: "
               f"per kernel error vs fp64 <= 1.2 x the exact fp32 MFMA chain's (tests/test_ops_gpu.py); full-size step vs a FLOAT64 oracle "
               f"(`numerics`): every tensor family <= 2.5 x the fp32 oracle's own distance, while the EXACT fp32 mode's worst family sits at "
               f"3.65 x -- summation order of its split-K chains, not operand precision)",
               data="synthetic",
               **({"mode_label": "build-only mode, no reference counterpart: the reference has no reduced-precision path (SURVEY D2: "
                                 "no autocast / GradScaler / .half()); --gemm-arith bf16x3 = every operand of the large GEMMs as two "
                                 "bf16 terms, three 16-bit products, fp32 accumulate (svl_set_gemm_emulation(3)); parity bounds: "
                                 "tests/test_fullsize_gpu.py::reduced_precision_mode_check"} if a.gemm_arith == "bf16x3" else {}),
               config=dict(workload=f"SemiVL step, {dataset} N={a.nclass}, {cfg['model'].replace('mmseg.', '')}, {a.crop}x{a.crop}, "
                                    f"bs={a.batch}/GPU labeled + {a.batch}/GPU unlabeled" +
                                    {(21, 512, 16): " (BASELINE configs[1])", (19, 801, 8): " (BASELINE configs[2])",
                                     (150, 512, 16): " (BASELINE configs[3])", (81, 512, 16): " (BASELINE configs[4])"}.get(
                                        (a.nclass, a.crop, a.batch), ""),
                           global_batch=2 * a.batch * world, parallelism=f"dp{world}", loss=round(loss_val, 5),
                           peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)))

    # ---- roofline of the dominant kernel family (fp32-MFMA GEMM / implicit-GEMM), measured live with HIP events ----
    if not a.no_profile:   # every rank runs the step (it contains the gradient all-reduce); rank 0 records the events
        if rank == 0:
            ops.PROFILE = {}
        # per-kernel HIP-event durations are taken with the step's two streams run back to back: under the overlap used
        # for `value` a kernel's interval includes the time it shares the chip with the other stream's kernels
        step(a.warmup + a.steps, overlap_streams=False)
        torch.cuda.synchronize()
        # ... and once more in the configuration `value` is measured in (streams overlapped): the same HIP-event brackets
        # then give every launch's duration INSIDE the real step (it shares the chip with the other streams' kernels)
        prof_solo = ops.PROFILE if rank == 0 else None
        if rank == 0:
            ops.PROFILE = {}
        step(a.warmup + a.steps + 1)
        torch.cuda.synchronize()
    if rank == 0 and not a.no_profile:
        g_arith_exact = a.gemm_arith == "f32"
        prof_in, prof, ops.PROFILE = ops.PROFILE, prof_solo, None
        in_step = {}        # launch shape -> [summed ms, launches] with the streams overlapped
        for fam_ in ("gemm", "gemm_bf16x", "attention", "attention_bf16x"):
            for e0, e1, w, tag, _sc in prof_in.get(fam_, []):
                r_ = in_step.setdefault(tag, [0.0, 0])
                r_[0] += e0.elapsed_time(e1)
                r_[1] += 1
        gx = prof.get("gemm_bf16x", []) + prof.get("attention_bf16x", [])   # launches served by the bf16 pipe (split emulation)
        g = prof.get("gemm", []) + gx + prof.get("attention", [])   # every MFMA kernel family
        t_gemm = sum(e0.elapsed_time(e1) for e0, e1, *_ in g) * 1e-3
        executed = sum(w for _, _, w, *_ in g)
        issued = executed       # sum of 2MNK over the launches
        if (a.nclass, a.crop) in ALGO_GF:
            vit_gf, dec_gf = ALGO_GF[(a.nclass, a.crop)]
            algo_contract = (vit_gf + dec_gf) * 1e9 * a.batch                         # 19 decoder fwd-equivalents
            algo = (vit_gf + dec_gf * DEC_EXECUTED_SHARE) * 1e9 * a.batch            # what the step executes: 14
        else:
            algo_contract = algo = issued
        ach = algo / t_gemm / 1e12
        step_tf = algo / (ms * 1e-3) / 1e12
        # (kept for continuity with rounds 1-2 under its own key: ALL MFMA kernel time of the step against the fp32-MFMA
        # peak.  It is NOT a roofline fraction in bf16x6 mode -- most launches run on the bf16 pipe; `roofline` below
        # prices the dominant kernel against the pipe it runs on.)
        out["mfma_step_vs_f32_pipe"] = dict(bound="mfma", achieved=round(ach, 2), peak=PEAK_F32_MFMA_TF, unit="TFLOP/s",
                               frac=round(ach / PEAK_F32_MFMA_TF, 4),
                               frac_executed=round(issued / t_gemm / 1e12 / PEAK_F32_MFMA_TF, 4),
                               frac_whole_step=round(step_tf / PEAK_F32_MFMA_TF, 4), whole_step_tflops=round(step_tf, 2),
                               algorithmic_gflop_per_step=round(algo / 1e9, 1),
                               contract=dict(algorithmic_gflop_per_step=round(algo_contract / 1e9, 1),
                                             achieved=round(algo_contract / t_gemm / 1e12, 2),
                                             whole_step_tflops=round(algo_contract / (ms * 1e-3) / 1e12, 2),
                                             note="SURVEY §8(d)'s per-step figure: 19 decoder forward-equivalents per image "
                                                  "pair, of which the step (like the reference's useful work) executes 14"),
                               kernel="all MFMA launches of one step: gemm_kernel / conv kernels + attn_{fwd,bwd}_kernel "
                                      "(v_mfma_f32_32x32x2_f32)" + (" + gemm_bf16x_kernel (v_mfma_f32_32x32x16_bf16, 6 products) / gemm_x6p_kernel<2> + attn_*_h2_kernel (v_mfma_f32_32x32x16_f16, 3 products)"
                                                                       if gx else ""), launches=len(g),
                               kernel_time_ms=round(t_gemm * 1e3, 2),
                               executed_tflops=round(issued / t_gemm / 1e12, 2),
                               note="achieved = EXECUTED algorithmic FLOPs of one step (ViT 2866.6 GF x B + decoder 1322.0 GF x "
                                    "14/19 x B: the unused pred_x_fp decode and the backward of the detached pred_w are not run) / "
                                    "summed duration of all svl_gemm_f32 + svl_attention_* launches of one step (HIP events on the launch stream); "
                                    "frac_executed = 2MNK actually issued by those launches / the same time; frac_whole_step = "
                                    "executed algorithmic FLOPs / the WHOLE step time (every non-MFMA pass counted against the MFMA "
                                    "peak); `contract` = the same ratios with SURVEY §8(d)'s 19-forward-equivalent figure")
        def nprod_of(tag_):
            """16-bit products a launch issues per fp32 MAC: 3 on fp16 x 2 operands (packed-planes GEMM, fused attention on
            pre-packed operands), else the mode's 6 (3 in bf16x3 mode)."""
            return 3 if (tag_ and tag_[0] in ("planes_h2", "fwd_h2", "bwd_h2", "split_h2")) or a.gemm_arith == "bf16x3" else 6

        if gx:   # the split-emulation GEMM family against ITS pipe: 16-bit dense peak vs the products actually issued
            nprod = 6 if a.gemm_arith == "bf16x6" else 3
            t_x = sum(e0.elapsed_time(e1) for e0, e1, *_ in gx) * 1e-3
            f_x = sum(w for _, _, w, *_ in gx)
            iss_x = sum(w * nprod_of(tag) for _, _, w, tag, *_ in gx)
            out["mfma_step_vs_f32_pipe"]["bf16_pipe"] = dict(
                kernel=f"gemm_bf16x_kernel<{nprod // 2 if nprod == 6 else 2},...> (svl_gemm_f32 in emulation mode {nprod})"
                       + (" + gemm_x6p_kernel<2, ...> (fp16 x 2 planes) + attn_{fwd,dq,dkv}_h2_kernel (fp16 x 2 pre-packed operands)"
                          if prof.get("attention_bf16x") else ""),
                launches=len(gx), kernel_time_ms=round(t_x * 1e3, 2), achieved=round(f_x / t_x / 1e12, 2),
                issued_16bit_tflops=round(iss_x / t_x / 1e12, 1), peak=PEAK_BF16_MFMA_TF, unit="TFLOP/s",
                frac=round(iss_x / t_x / 1e12 / PEAK_BF16_MFMA_TF, 4),
                note=f"achieved = 2MNK of the launches / their summed duration (fp32-equivalent); frac = the 16-bit products "
                     f"they issue (3 per fp32 MAC on fp16 x 2 planes, 6 on bf16 x 3 and the in-register split kernels) / the "
                     f"same time vs {PEAK_BF16_MFMA_TF:.0f} TF dense")
        # the ViT encoder alone (north_star: ">= 60 % MFMA peak on the ViT encoder"): launches issued inside the encoder's
        # forward / backward regions (5 trainable + 2 frozen forwards, 4 backwards per step)
        gv = [e for e in g if e[4] == "vit"]
        if gv and (a.nclass, a.crop) == (21, 512):
            t_v = sum(e0.elapsed_time(e1) for e0, e1, *_ in gv) * 1e-3
            ach_v = VIT_GF_PER_UNIT_B * 1e9 * a.batch / t_v / 1e12
            # priced against the pipe the encoder's launches run on: bf16 dense / products per fp32 MAC in the split
            # modes (every ViT GEMM and the fused attention), the fp32 MFMA peak in exact mode
            split_n = {"bf16x6": 6, "bf16x3": 3}.get(a.gemm_arith)
            if split_n:   # products per fp32 MAC, weighted by the executed FLOPs of the launches (3 on fp16 x 2 planes, else 6)
                split_n = sum(e[2] * nprod_of(e[3]) for e in gv) / max(sum(e[2] for e in gv), 1.0)
            peak_v = PEAK_BF16_MFMA_TF / split_n if split_n else PEAK_F32_MFMA_TF
            out["roofline_vit_encoder"] = dict(bound="mfma", achieved=round(ach_v, 2), peak=round(peak_v, 1),
                                               products_per_fp32_mac=round(split_n, 2) if split_n else None,
                                               unit="TFLOP/s (fp32-equivalent)" if split_n else "TFLOP/s",
                                               frac=round(ach_v / peak_v, 4),
                                               frac_vs_f32_mfma_pipe=round(ach_v / PEAK_F32_MFMA_TF, 4), launches=len(gv),
                                               kernel_time_ms=round(t_v * 1e3, 2),
                                               note="algorithmic ViT FLOPs (2866.6 GF x B) / summed duration of the "
                                                    "GEMM + svl_attention_* launches of the encoder regions; peak = the pipe they "
                                                    "run on (bf16 dense / products per fp32 MAC in the split modes); "
                                                    "frac_vs_f32_mfma_pipe = the same rate against the 157.3 TF fp32 MFMA peak the "
                                                    "exact mode is bound by (north_star's >= 60 % was stated for that pipe)")
        # phase split of the MFMA time: launches inside the ViT regions / the VLG head regions / elsewhere
        gh = [e for e in g if e[4] == "head"]
        t_h = sum(e0.elapsed_time(e1) for e0, e1, *_ in gh)
        t_vv = sum(e0.elapsed_time(e1) for e0, e1, *_ in gv)
        out["phase_ms"] = dict(step=round(ms, 1), vit_encoder_mfma=round(t_vv, 1), vlg_head_mfma=round(t_h, 1),
                               other_mfma=round(t_gemm * 1e3 - t_vv - t_h, 1),
                               pixel_loss=round(sum(e0.elapsed_time(e1) for e0, e1, *_ in prof.get("ce_fused", []) +
                                                    prof.get("softmax_max", []) + prof.get("ce_up_fused", []) +
                                                    prof.get("softmax_max_up", [])), 2),
                               note="HIP-event durations of the timed kernel families in one step; the rest of the step is "
                                    "normalisation / elementwise / resampling / optimizer passes (profiles/)")
        # largest single launch shapes
        by, fam_of = {}, {}
        for fam_ in ("gemm", "gemm_bf16x", "attention", "attention_bf16x"):
            for e0, e1, w, tag, _scope in prof.get(fam_, []):
                r = by.setdefault(tag, [0.0, 0.0, 0])
                r[0] += e0.elapsed_time(e1) * 1e-3
                r[1] += w
                r[2] += 1
                fam_of[tag] = fam_
        if os.environ.get("SVL_BENCH_SHAPES"):  # every launch shape of the step, for tuning (tools/README.md)
            with open(os.environ["SVL_BENCH_SHAPES"], "w") as f:
                for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
                    f.write("%-44s n=%3d  %8.2f ms  %6.1f TF\n" % (k, v[2], v[0] * 1e3, v[1] / v[0] / 1e12))
        top = sorted(by.items(), key=lambda kv: -kv[1][0])[:6]
        out["mfma_step_vs_f32_pipe"]["top_shapes"] = [dict(mode_MNKb=list(k), ms=round(v[0] * 1e3, 2), n=v[2],
                                              tflops=round(v[1] / v[0] / 1e12, 1)) for k, v in top]
        # ---- `roofline`: the single dominant kernel (FFN-1 of the ViT blocks, M = images x 1025, N = 3072, K = 768: the
        # largest share of the step's kernel time), ALGORITHMIC FLOPs per launch / its average launch duration (HIP events
        # on the launch stream), priced against the peak of the pipe it runs on: bf16 dense / 6 products per fp32 MAC in
        # bf16x6 mode, the fp32-MFMA peak in exact mode.  Traffic: committed PMC record of the same kernel source.
        ntok = ((a.crop + 15) // 16) ** 2 + 1           # 1025 at 512^2, 2602 at 801^2
        Md = 2 * a.batch * ntok

        def in_step_fields(tag, flops, peak_):
            """The same launches inside the real (stream-overlapped) step: mean duration and the fraction it gives."""
            r_ = in_step.get(tag)
            if not r_ or not r_[1]:
                return {}
            avg = r_[0] / r_[1]
            return dict(avg_ms_in_step=round(avg, 4), frac_in_step=round(flops / (avg * 1e-3) / 1e12 / peak_, 4),
                        in_step_note="avg_ms = HIP events with the step's streams run back to back (a solo duration); "
                                     "avg_ms_in_step = the same brackets in the step as `value` measures it (streams "
                                     "overlapped: a launch shares the chip with the other streams' kernels)")

        voc_cfg = (a.nclass, a.crop) == (21, 512)
        if not voc_cfg and by:
            # Other configs: the launch SHAPE with the largest summed duration of THIS run (at N = 81 / 150 the decoder's
            # convolutions, at 801^2 the attention), priced against the pipe its kernel family runs on.
            tag, v = max(by.items(), key=lambda kv: kv[1][0])
            fam = fam_of[tag]
            on_bf16 = fam.endswith("bf16x")
            nprod = nprod_of(tag)
            peak = PEAK_BF16_MFMA_TF / nprod if on_bf16 else PEAK_F32_MFMA_TF
            d_tf = v[1] / v[0] / 1e12
            fl = v[1] / v[2]
            kern = {"gemm": "svl_gemm_f32 on v_mfma_f32_32x32x2_f32 (gemm_kernel / tiled 3x3 / short-K stream)",
                    "gemm_bf16x": "svl_gemm_f32 / svl_gemm_planes_f32 / tiled 3x3 in the split arithmetic (gemm_bf16x_kernel, "
                                  "gemm_x6p_kernel, conv3x3_*_bf16x_kernel: v_mfma_f32_32x32x16_bf16)",
                    "attention": "svl_attention_* (attn_*_kernel, fp32 MFMA)",
                    "attention_bf16x": "svl_attention_*_h2 (attn_{fwd,dq,dkv}_h2_kernel on fp16 x 2 pre-packed operands, "
                                       "v_mfma_f32_32x32x16_f16; pack pass included in the launch's duration)"
                                       if tag[0] in ("fwd_h2", "bwd_h2") else
                                       "svl_attention_* (attn_*_x6_kernel, v_mfma_f32_32x32x16_bf16)"}[fam]
            out["roofline"] = dict(
                bound="mfma", achieved=round(d_tf, 1), peak=round(peak, 1),
                unit="TFLOP/s (fp32-equivalent)" if on_bf16 else "TFLOP/s", frac=round(d_tf / peak, 4), traffic=None,
                kernel=f"{kern}, launch shape (a_mode, b_mode, M, N, K, batch | name, ...) = {tag}",
                launches=v[2], avg_ms=round(v[0] * 1e3 / v[2], 4), **in_step_fields(tag, fl, peak), flops_per_launch=fl,
                share_of_mfma_time=round(v[0] / t_gemm, 4),
                traffic_note="no PMC record for this launch shape (profiles/ holds the records of the VOC line's kernels)",
                note="the launch shape with the largest summed duration of this run; achieved = its 2MNK / mean launch "
                     "duration (HIP events, streams back to back); peak = the pipe its kernel family runs on (2500 TF bf16 "
                     f"dense / {nprod} products per fp32 MAC, or the 157.3 TF fp32 MFMA peak)")
        elif not g_arith_exact:
            h2 = ("planes_h2", Md, 3072, 768, ops.ACT_GELU) in by      # fp16 x 2 planes: three products per fp32 MAC
            planes = h2 or ("planes", Md, 3072, 768, ops.ACT_GELU) in by
            nprod = 3 if (h2 or a.gemm_arith == "bf16x3") else 6
            dom_tag = (("planes_h2" if h2 else "planes"), Md, 3072, 768, ops.ACT_GELU) if planes else (0, 0, Md, 3072, 768, 1)
            dom = by.get(dom_tag)
            clock = None
            if dom is not None and planes:
                # the clock the chip sustains under this kernel (svl_clock_probe waves on a second stream next to 24
                # back-to-back launches of the same shape and epilogue)
                xa = ops.split_planes(torch.randn(Md, 768, device=dev))
                wb = ops.split_planes(torch.randn(3072, 768, device=dev) * 0.05)
                bias_, pre_ = torch.zeros(3072, device=dev), ops.empty(Md, 3072, device=dev)
                po_ = ops.Planes(Md, 3072, device=dev)
                avg_ = dom[0] * 1e3 / dom[2]
                clock = dict(
                    under_dominant_kernel=clock_probe_mhz(
                        lambda: [ops.pgemm(xa, wb, Md, 3072, None, po_, bias_, ops.ACT_GELU, pre_) for _ in range(24)], 24 * avg_, dev),
                    nominal=PEAK_CLOCK_MHZ,
                    note="mean of 64 probe waves: shader-clock counter / 100 MHz counter over 80 % of the 24 launches; the "
                         "MFMA peaks are quoted at the nominal clock (the whole step cannot be probed this way: with the "
                         "runtime's 4 hardware queues the probe's stream shares a queue with one of the step's streams "
                         "and serialises it -- tools/clock_step.py)")
                del xa, wb, bias_, pre_, po_
            if dom is not None:
                peak = PEAK_BF16_MFMA_TF / nprod
                fl_ = 2.0 * Md * 3072 * 768
                solo_ms = dom[0] * 1e3 / dom[2]
                ins = in_step.get(dom_tag)
                # `achieved` / `frac` = the launches inside the step as `value` measures it (streams overlapped); the solo
                # duration (the step's streams run back to back) is kept beside it
                avg_ms = ins[0] / ins[1] if ins and ins[1] else solo_ms
                d_tf = fl_ / (avg_ms * 1e-3) / 1e12
                opb = 4 if h2 else 6        # bytes per operand element
                traffic, tnote = (pmc_traffic_record(a.batch, "pmc_x6p_traffic.json", "gemm_planes_impl.h") if planes
                                  else (None, "no PMC record for this kernel"))
                rr = rocprof_dispatch_record(a.batch) if planes else None
                rocprof_in_step = None if rr is None else dict(
                    mean_ms=round(rr["mean_us"] * 1e-3, 4), min_ms=round(rr["min_us"] * 1e-3, 4), dispatches=rr["n"],
                    frac=round(fl_ / (rr["mean_us"] * 1e-6) / 1e12 / peak, 4), source=rr.get("source"),
                    note="the same launches as rocprofv3 --kernel-trace timed them inside the overlapped step (kernel begin -> "
                         "end); `avg_ms` brackets the launch with HIP events on its stream, which also contain the time the "
                         "dispatch waits for CUs another stream's kernel holds -- the event figure is the larger one")
                out["roofline"] = dict(
                    bound="mfma", achieved=round(d_tf, 1), peak=round(peak, 1), unit="TFLOP/s (fp32-equivalent)",
                    frac=round(d_tf / peak, 4), traffic=traffic,
                    kernel=((f"gemm_x6p_kernel<{2 if h2 else 3}, 256, EPI_GELU> (svl_gemm_planes_f32: packed "
                             + ("fp16 x 2 planes with row scales, 3 products on v_mfma_f32_32x32x16_f16"
                                if h2 else "bf16 x 3 planes, 6 products on v_mfma_f32_32x32x16_bf16")
                             + "; bias + erf-GELU + saved pre-activation + result as planes)") if planes
                            else "gemm_bf16x_kernel (svl_gemm_f32, in-register split)") + ", M=%d N=3072 K=768" % Md,
                    launches=dom[2], avg_ms=round(avg_ms, 4), avg_ms_solo=round(solo_ms, 4),
                    achieved_solo=round(fl_ / (solo_ms * 1e-3) / 1e12, 1),
                    frac_solo=round(fl_ / (solo_ms * 1e-3) / 1e12 / peak, 4),
                    flops_per_launch=fl_, mfma_issued_tflops=round(d_tf * nprod, 1),
                    frac_of_16bit_dense_peak=round(d_tf * nprod / PEAK_BF16_MFMA_TF, 4),
                    algorithmic_bytes=(Md * 768 * opb + 3072 * 768 * opb + Md * 3072 * (4 + opb)) if planes else 513.0e6 * a.batch / 16,
                    traffic_note=tnote, clock_mhz=clock, rocprof_in_step=rocprof_in_step,
                    frac_solo_at_sustained_clock=(round(fl_ / (solo_ms * 1e-3) / 1e12 / peak * PEAK_CLOCK_MHZ / clock["under_dominant_kernel"], 4)
                                                  if clock and clock["under_dominant_kernel"] else None),
                    note=f"achieved = 2MNK / mean launch duration INSIDE the timed step (HIP events on the launch stream, the "
                         f"step's streams overlapped as `value` runs them); *_solo = the same launches with the step's streams "
                         f"run back to back; peak = {PEAK_BF16_MFMA_TF:.0f} TF 16-bit dense / {nprod} products per fp32 MAC "
                         f"(MI355X_MICROARCH.md); algorithmic bytes = A and B planes ({opb} B/element) read once + "
                         f"pre-activation (4 B) and result planes ({opb} B) written once")
            if "roofline" in out:
                # The dominant launch is the most frequent large one (40 per step); two shapes hold MORE of the step's time
                # and are reported beside it, priced against the same pipe (fp16 x 2 operands: 3 products per fp32 MAC):
                # the fused attention backward (pack + D + dK/dV + dQ kernels of one call) and the 32800 x 768 x 3072 GEMMs
                # (FFN-2 forward and the FFN-1 input gradient).
                rr2 = inloop_dispatch_record()
                extra = []
                for kind_, label_ in (("bwd_h2", "fused attention backward, svl_attention_bwd_h2 (operand pack + D + dK/dV + dQ "
                                                 "grids + leftover-row tail; 14 T^2 d FLOP per (image, head): S and dP recomputed)"),
                                      ("fwd_h2", "fused attention forward, svl_attention_fwd_h2 (operand pack + grid + tail; 4 T^2 d)"),
                                      ("ffn2", f"gemm_x6p_kernel<2, 192> M={Md} N=768 K=3072 (FFN-2 forward / FFN-1 input gradient)")):
                    if kind_ == "ffn2":
                        tags_ = [t_ for t_ in by if t_ and t_[0] == "planes_h2" and tuple(t_[1:4]) == (Md, 768, 3072)]
                    else:
                        tags_ = [t_ for t_ in by if t_ and t_[0] == kind_]
                    if not tags_:
                        continue
                    tg_ = max(tags_, key=lambda t_: by[t_][0])
                    v_ = by[tg_]
                    fl2 = v_[1] / v_[2]
                    solo2 = v_[0] * 1e3 / v_[2]
                    ins2 = in_step.get(tg_)
                    avg2 = ins2[0] / ins2[1] if ins2 and ins2[1] else solo2
                    pk2 = PEAK_BF16_MFMA_TF / 3
                    e_ = dict(shape=label_, tag=list(tg_), launches=v_[2], flops_per_launch=fl2, avg_ms=round(avg2, 4),
                              avg_ms_solo=round(solo2, 4), achieved=round(fl2 / (avg2 * 1e-3) / 1e12, 1), peak=round(pk2, 1),
                              frac=round(fl2 / (avg2 * 1e-3) / 1e12 / pk2, 4), frac_solo=round(fl2 / (solo2 * 1e-3) / 1e12 / pk2, 4),
                              ms_per_step=round(avg2 * (ins2[1] if ins2 else v_[2]), 2))
                    if rr2 and kind_ in rr2.get("kernels", {}):
                        e_["rocprof_in_step"] = rr2["kernels"][kind_]
                    extra.append(e_)
                out["roofline"]["larger_shapes"] = extra
                out["roofline"]["larger_shapes_note"] = (
                    "avg_ms = HIP events around the whole entry point inside the timed step (streams overlapped); rocprof_in_step "
                    "(when present) = mean kernel durations of the same launches from the committed rocprofv3 --kernel-trace of "
                    "this command (profiles/inloop_dispatches.json, same kernel sources by sha-256)")
        else:
            dom_tag = (0, 0, Md, 3072, 768, 1)
            dom = by.get(dom_tag)
            if dom is not None:
                d_tf = dom[1] / dom[0] / 1e12
                traffic, tnote = pmc_traffic_record(a.batch)
                out["roofline"] = dict(
                    bound="mfma", achieved=round(d_tf, 1), peak=PEAK_F32_MFMA_TF, unit="TFLOP/s",
                    frac=round(d_tf / PEAK_F32_MFMA_TF, 4), frac_is="solo (streams back to back); frac_in_step beside it",
                    traffic=traffic,
                    kernel="gemm_kernel<128,128,2,2,KCONTIG,KCONTIG> (svl_gemm_f32, v_mfma_f32_32x32x2_f32), M=%d N=3072 K=768" % Md,
                    launches=dom[2], avg_ms=round(dom[0] * 1e3 / dom[2], 4),
                    **in_step_fields(dom_tag, 2.0 * Md * 3072 * 768, PEAK_F32_MFMA_TF), flops_per_launch=2.0 * Md * 3072 * 768,
                    algorithmic_bytes=513.0e6 * a.batch / 16, traffic_note=tnote)
        c = prof.get("ce_fused", [])
        cu = prof.get("ce_up_fused", [])
        if cu and not c:
            # The step's cross entropy runs on head-resolution logits (ce_up_kernel evaluates the x4 resize itself, forward and
            # backward): `roofline_hbm` describes THAT kernel as the timed step launches it.  `frac` is priced on SURVEY §8(d)'s
            # contract bytes, (12 N + 40) B per full-resolution pixel and branch -- the traffic of the unfused reference the
            # kernel replaces -- because that is the figure north_star's ">= 70 % HBM" was stated on; the bytes the kernel
            # really moves (head-resolution logits in, gradient out, 28 B of maps per pixel) are 7.6 x fewer and are printed
            # beside it: by its own traffic the kernel is LDS / VALU-bound, not HBM-bound (profiles/*_pmc_sq_ce_up.txt).
            cin = prof_in.get("ce_up_fused", []) or cu
            N_, S_ = a.nclass, a.crop
            contract = float(a.batch * S_ * S_) * (12 * N_ + 40)
            t_in = sum(e0.elapsed_time(e1) for e0, e1, *_ in cin) * 1e-3 / len(cin)
            t_solo = sum(e0.elapsed_time(e1) for e0, e1, *_ in cu) * 1e-3 / len(cu)
            real = sum(w for _, _, w, *_ in cu) / len(cu)
            sm_in = prof_in.get("softmax_max_up", []) or prof.get("softmax_max_up", [])
            t_su = sum(e0.elapsed_time(e1) for e0, e1, *_ in sm_in) * 1e-3 / max(len(sm_in), 1)
            out["roofline_hbm"] = dict(
                bound="hbm", kernel="ce_up_kernel (svl_ce_up_fused_f32): bilinear x4 resize + CE + confidence weighting + "
                                    "guidance term, forward and backward, on [B, N, h, w] logits; gradient at [h, w]",
                in_step=True, achieved=round(contract / t_in / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                frac=round(contract / t_in / 1e9 / PEAK_HBM_GBS, 4), frac_is="SURVEY §8(d) contract bytes / in-step launch duration",
                launches=len(cin), avg_ms=round(t_in * 1e3, 4), avg_ms_solo=round(t_solo * 1e3, 4),
                frac_solo=round(contract / t_solo / 1e9 / PEAK_HBM_GBS, 4),
                algorithmic_bytes=contract, traffic=None,
                real_bytes=dict(per_launch=real, achieved_gbs=round(real / t_in / 1e9, 1),
                                frac_of_hbm_peak=round(real / t_in / 1e9 / PEAK_HBM_GBS, 4),
                                note="what the kernel moves: 8 N h w + 28 H W bytes per image -- the [B, N, H, W] logits, their "
                                     "gradient and the two resize passes are never written (bytes NOT moved per launch: "
                                     f"{float(a.batch * S_ * S_) * ((8 * N_ + 28) + 8.5 * N_) - real:.3e})"),
                softmax_max_up_ms=round(t_su * 1e3, 4),
                traffic_note="no PMC byte record for this kernel; its SQ record (LDS bank conflicts / LDS waits) is "
                             "profiles/*_pmc_sq_ce_up.txt",
                note="achieved / frac = (12 N + 40) B x B H W per branch (the unfused reference's API-boundary traffic, SURVEY "
                     "§8(d)) / mean launch duration inside the timed step (HIP events on the launch stream); real_bytes = the "
                     "kernel's own traffic: by it the kernel is LDS / VALU-bound (DESIGN §3)")
        elif c:
            t_ce = sum(e0.elapsed_time(e1) for e0, e1, *_ in c) * 1e-3
            by_ce = sum(w for _, _, w, *_ in c)
            ce_traffic, ce_note = None, "no PMC record (profiles/pmc_ce_traffic.json)"
            try:
                rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_ce_traffic.json")))
                if rec.get("N") == a.nclass and rec.get("B") == a.batch and rec.get("HW") == a.crop * a.crop:
                    ce_traffic, ce_note = rec["traffic_bytes"], rec.get("note", "")
            except (OSError, ValueError, KeyError):
                pass
            avg_s = t_ce / len(c)
            contract_gbs = by_ce / t_ce / 1e9
            out["roofline_hbm"] = dict(bound="hbm", kernel="ce_fused_kernel (svl_ce_fused_f32; the step resizes its logits: "
                                                            "geometry the resize-fused kernel refuses, or SVL_NO_UP_LOSS)",
                                       in_step=True, achieved=round(contract_gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                       frac=round(contract_gbs / PEAK_HBM_GBS, 4), frac_is="SURVEY §8(d) contract bytes / launch duration",
                                       traffic=ce_traffic, launches=len(c), avg_ms=round(avg_s * 1e3, 4),
                                       algorithmic_bytes=by_ce / len(c),
                                       frac_real_traffic=round(ce_traffic / avg_s / 1e9 / PEAK_HBM_GBS, 4) if ce_traffic else None,
                                       traffic_note=ce_note,
                                       note="(12N+40) B/px per fwd+bwd branch (SURVEY §8(d)) / mean launch duration; the kernel "
                                            "itself moves (8N+28) B/px (`traffic`: PMC FETCH + WRITE of one launch)")
    # ---- the same step in the OTHER arithmetic (exact fp32 MFMA next to a bf16x6 value, bf16x6 next to an f32 value) ----
    if not a.no_throughput_mode and a.gemm_arith in ("f32", "bf16x6"):
        other = "f32" if a.gemm_arith == "bf16x6" else "bf16x6"
        ops.set_gemm_emulation(EMU[other])
        n2 = min(a.steps, 5)
        dt2, losses2 = timed(1, n2, a.warmup + a.steps + 1)
        ops.set_gemm_emulation(EMU[a.gemm_arith])
        key = "exact_f32" if other == "f32" else "throughput_mode"
        out[key] = dict(
            gemm_arith=other, value=round(2.0 * a.batch * world / (dt2 / n2), 3), unit="images/s", steps=n2,
            ms_per_step=round(dt2 / n2 * 1e3, 2), loss=round(float(losses2[0].item()), 5),
            whole_step_frac_of_f32_mfma_peak=(round(sum(ALGO_GF[(a.nclass, a.crop)]) * 1e9 * a.batch / (dt2 / n2) / 1e12 /
                                                    PEAK_F32_MFMA_TF, 4) if (a.nclass, a.crop) in ALGO_GF else None),
            note="the same step with every GEMM on v_mfma_f32_32x32x2_f32 (bit-for-bit a k-ordered fp32 fma chain)"
                 if other == "f32" else
                 "svl_set_gemm_emulation(6): dense GEMMs with M>=256 split every fp32 operand element into 3 bf16 terms and "
                 "accumulate the 6 leading cross products in fp32 (v_mfma_f32_32x32x16_bf16)")
    # ---- both arithmetics on the SAME weights and the SAME step (no optimizer step in between; the dropout2d masks of the
    # feature perturbation are re-seeded): cross-mode consistency evidence next to the two throughput numbers
    if world == 1 and not a.no_throughput_mode and a.gemm_arith in ("f32", "bf16x6"):
        cm = {}
        for name_, mode_ in (("bf16x6", 6), ("f32", 0)):
            ops.set_gemm_emulation(mode_)
            torch.manual_seed(4321)
            l_ = semivl_train_step(model, batch, a.warmup + a.steps + 9, total_iters, cfg, optimizer=None, reducer=None)
            cm[name_] = [round(float(v), 6) for v in l_.tolist()]
        ops.set_gemm_emulation(EMU[a.gemm_arith])
        out["cross_mode_same_weights"] = dict(
            losses_bf16x6=cm["bf16x6"], losses_f32=cm["f32"],
            max_abs_diff=round(max(abs(x - y) for x, y in zip(cm["bf16x6"], cm["f32"])), 7),
            names=["loss", "loss_x", "loss_s1", "loss_s2", "loss_fp", "loss_mc_s1", "loss_mc_s2", "loss_mc_fp"],
            note="one forward + loss evaluation of the step in each arithmetic from identical weights, inputs and dropout masks")
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(512, 21, *a.cpu_baseline_steps)
        except Exception as e:  # the baseline leg must never take the GPU number down with it
            out["cpu_baseline"] = dict(value=None, unit="images/s", cores=physical_cores(), kind="port",
                                       sample=f"failed: {type(e).__name__}: {e}")
    if rank == 0:
        # accuracy record of the arithmetic `value` runs in: worst distance from a FLOAT64 evaluation of the same step, per mode
        # and tensor family, written by tests/test_fullsize_gpu.py::test_fullsize_gradient_error_against_fp64 on the GPU box
        # and committed (profiles/numerics_fp64.json) -- bench.py itself has no oracle at B = 16
        try:
            out["numerics"] = dict(json.load(open(os.path.join(ROOT, "profiles", "numerics_fp64.json"))),
                                   source="profiles/numerics_fp64.json (committed record of the GPU test; its FP64_RATCHET asserts "
                                          "these figures + 20 %)")
        except (OSError, ValueError):
            out["numerics"] = None
    if AS_MULTI:
        out["settings"] = dict(as_multi=True, gpu_max_hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"),
                               weight_gradient_stream=bool(ops.WGRAD_STREAM),
                               note="one GPU with the stream / hardware-queue settings of a multi-GPU rank")
    if rank == 0 and world == 1 and not AS_MULTI and not a.no_multi_anchor:
        # The ranks of an N > 1 run use 8 hardware queues and no weight-gradient stream (top of this file): re-measure
        # N = 1 in a child process with exactly those settings, so that a scaling curve has an apples-to-apples anchor.
        import subprocess
        torch.cuda.empty_cache()
        n3 = min(a.steps, 5)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--as-multi", "--steps", str(n3), "--warmup", "2",
               "--batch", str(a.batch), "--crop", str(a.crop), "--nclass", str(a.nclass), "--gemm-arith", a.gemm_arith,
               "--no-profile", "--no-cpu-baseline", "--no-throughput-mode"]
        try:
            r_ = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            child = json.loads(r_.stdout.strip().splitlines()[-1])
            out["n1_same_settings"] = dict(value=child["value"], unit="images/s", ms_per_step=child["ms_per_step"], steps=n3,
                                           settings=child.get("settings"),
                                           note="python bench.py --gpus 1 --as-multi: what every rank of `--gpus N` runs with; "
                                                "divide a multi-GPU `value` by N x THIS for scaling efficiency of the "
                                                "communication alone")
            # kept for the N > 1 runs the driver starts next on this node: they print their efficiency against it themselves
            try:
                import socket
                json.dump(dict(host=socket.gethostname(), batch=a.batch, crop=a.crop, nclass=a.nclass, gemm_arith=a.gemm_arith,
                               value=child["value"], ms_per_step=child["ms_per_step"], default_line_value=round(ips, 3)),
                          open(os.path.join(ROOT, ".bench_n1_anchor.json"), "w"))
            except OSError:
                pass
        except Exception as e:
            out["n1_same_settings"] = dict(value=None, note=f"child run failed: {type(e).__name__}: {e}")
    if world > 1:      # the communication side of the last timed step, so that a scaling curve explains itself
        torch.cuda.synchronize()
        rep = red.timing_report()
        if rank == 0 and rep is not None:
            out["allreduce"] = dict(rep, payload_mb=round(opt.g.numel() * 4 / 2 ** 20, 1), backend=dist.get_backend(),
                                    world=world)
        if rank == 0:
            # efficiency against the N = 1 anchor with THESE settings (8 hardware queues, no weight-gradient stream), left by
            # the N = 1 run of this node (`n1_same_settings`); the driver computes its own figure from the per-N values
            try:
                import socket
                anc = json.load(open(os.path.join(ROOT, ".bench_n1_anchor.json")))
                same = (anc.get("host") == socket.gethostname() and (anc.get("batch"), anc.get("crop"), anc.get("nclass"),
                        anc.get("gemm_arith")) == (a.batch, a.crop, a.nclass, a.gemm_arith))
                out["scaling_vs_n1_same_settings"] = dict(
                    n1_same_settings=anc["value"], efficiency=round(ips / (world * anc["value"]), 4) if same else None,
                    efficiency_vs_default_n1=round(ips / (world * anc["default_line_value"]), 4) if same else None,
                    note="value / (N x the N = 1 run with a rank's stream settings) -- the communication's share alone; "
                         "efficiency_vs_default_n1 divides by the default N = 1 line (weight-gradient stream on, 4 queues)" if same
                         else "anchor file is from another host / config: not used")
            except (OSError, ValueError, KeyError):
                out["scaling_vs_n1_same_settings"] = dict(efficiency=None, note="no N = 1 anchor on this node yet (run `python bench.py` first)")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
