"""Every elementwise / copy / add launch of one step by (op, mode, elements): count, solo us.  python tools/elt_shapes.py [config]"""
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
import traceback
def wrap(fn, tag, key):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(*a, **k); e1.record()
        fr = traceback.extract_stack(limit=3)[0]
        rec.append((tag, key(a, k), f"{os.path.basename(fr.filename)}:{fr.lineno}", e0, e1))
        return r
    return f
ops.eltwise = wrap(ops.eltwise, "eltwise", lambda a, k: (a[0], a[1].numel()))
ops.add = wrap(ops.add, "add", lambda a, k: (a[0].numel(),))
ops.copy2d = wrap(ops.copy2d, "copy2d", lambda a, k: (a[10] * a[11],))
ops.fill = wrap(ops.fill, "fill", lambda a, k: (a[0].numel(),))
ops.colsum = wrap(ops.colsum, "colsum", lambda a, k: (a[0].numel(),))
semivl_train_step(model, batch, 2, 100, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for tag, key, where, e0, e1 in rec:
    a = agg[(tag, key, where)]; a[0] += e0.elapsed_time(e1); a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"# {name}: {tot:.2f} ms in {len(rec)} launches")
for (tag, key, where), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{ms:7.2f} ms n={n:3d} {ms / n * 1e3:7.1f} us  {tag} {key} @ {where}")
