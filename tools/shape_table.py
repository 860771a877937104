"""Per-shape table of the MFMA launches of one bench step (HIP-event brackets of ops.PROFILE; launches serialised by the
brackets' own ordering are NOT -- the weight-gradient stream is switched off so that every duration is a solo duration).
usage: python tools/shape_table.py [config] [batch]"""
import os, sys, collections, torch
os.environ.setdefault("SVL_NO_WGRAD_STREAM", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from semivl_amd import ops
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
from semivl_amd.train import FusedAdamW, GradAllReducer, semivl_train_step
name = sys.argv[1] if len(sys.argv) > 1 else "pascal"
name = {"voc": "pascal"}.get(name, name)
shapes = {"pascal": (16, 512, 21), "cityscapes": (8, 801, 19), "ade": (16, 512, 150), "coco": (16, 512, 81)}
B, crop, ncls = shapes[name]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
dev = torch.device("cuda:0")
ops.set_gemm_emulation(int(os.environ.get("SVL_SHAPE_EMU", "6")))
cfg = exp40_cfg(B, crop, ncls, name)
cfg_solo = dict(cfg, overlap_streams=False)     # the two streams of the step back to back: solo durations
torch.manual_seed(1234)
model = build_model(cfg).to(dev)
opt = FusedAdamW(model, cfg["optimizer"]); red = GradAllReducer(opt)
batch = synthetic_batch(B, crop, ncls, seed=1234, device=dev)
for i in range(2):
    semivl_train_step(model, batch, i, 100, cfg, optimizer=opt, reducer=red)
torch.cuda.synchronize()
ops.PROFILE = {}
semivl_train_step(model, batch, 2, 100, cfg_solo, optimizer=opt, reducer=red)
torch.cuda.synchronize()
prof, ops.PROFILE = ops.PROFILE, None
rows = collections.defaultdict(lambda: [0.0, 0, 0.0])
for fam, recs in prof.items():
    for e0, e1, work, tag, scope in recs:
        r = rows[(fam, tag, scope)]
        r[0] += e0.elapsed_time(e1); r[1] += 1; r[2] += work
tot = sum(r[0] for r in rows.values())
print(f"# {name} B={B}: {tot:.1f} ms in bracketed launches")
for (fam, tag, scope), (ms, n, work) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"{ms:8.2f} ms  n={n:4d}  {work / ms / 1e9 if ms else 0:7.1f} T/s  {fam:12s} {scope or '-':6s} {tag}")
