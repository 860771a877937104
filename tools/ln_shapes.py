"""Shapes and solo durations of every LayerNorm forward / backward call of one step: python tools/ln_shapes.py [config]"""
import os, sys, collections, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from semivl_amd import ops
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step
name = sys.argv[1] if len(sys.argv) > 1 else "pascal"
B, crop, ncls = {"pascal": (16, 512, 21), "cityscapes": (8, 801, 19), "ade": (16, 512, 150), "coco": (16, 512, 81)}[name]
dev = torch.device("cuda:0")
ops.set_gemm_emulation(6)
cfg = dict(exp40_cfg(B, crop, ncls, name), overlap_streams=False)
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"]); red = GradAllReducer(opt)
batch = synthetic_batch(B, crop, ncls, seed=1234, device=dev)
for i in range(2):
    semivl_train_step(model, batch, i, 100, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize()
rec = []
def wrap(fn, tag):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        x = a[1] if tag == "bwd" else a[0]
        rec.append((tag, tuple(x.shape), bool(k.get("planes")), k.get("want_y", True), bool(k.get("want_wgrad")), e0, e1))
        return r
    return f
ops.layernorm_fwd, ops.layernorm_bwd = wrap(ops.layernorm_fwd, "fwd"), wrap(ops.layernorm_bwd, "bwd")
import semivl_amd.model.vit as V, semivl_amd.model.vlg_head as H
semivl_train_step(model, batch, 2, 100, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for tag, shp, pl, wy, wg, e0, e1 in rec:
    a = agg[(tag, shp, pl, wy, wg)]; a[0] += e0.elapsed_time(e1); a[1] += 1
for k, (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    tag, shp, pl, wy, wg = k
    mb = shp[0] * shp[1] * 4 / 1e6 * ((2 if not pl else (3.5 if wy else 2.5)) if tag == "fwd" else 4)
    print(f"{ms:7.2f} ms n={n:3d} {ms / n * 1e3:7.1f} us  {mb * n / ms * 1e-3:5.2f} TB/s  {tag} {shp} planes={pl} want_y={wy} wgrad={wg}")
