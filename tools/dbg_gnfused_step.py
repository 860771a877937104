"""Debug aid: the full-size VOC step (B=2) in mode 6 with the GroupNorm statistics fused into the conv epilogue vs the two-pass
path: parameter gradients of the two runs against each other."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_fullsize_gpu import build_pair, fp_masks_for
from oracle import semivl_oracle as O
from semivl_amd import ops
from semivl_amd.train import semivl_train_step
dev = torch.device("cuda:0")
cfg, hip, orc = build_pair(dev)
cfg = dict(cfg, conf_thresh=0.0, head_chunk_class_images=2 * 21)
batch = O.synthetic_batch(2, 512, 21, seed=99)
masks = fp_masks_for((768, 768, 512), b=4)
def run(fused, mode):
    ops.CONV_GN_FUSED = fused
    ops.set_gemm_emulation(mode)
    for p_ in hip.parameters():
        p_.grad = None
    losses, aux = semivl_train_step(hip, {k: v.to(dev) for k, v in batch.items()}, 100, 1000, cfg,
                                    fp_masks=[m.to(dev) for m in masks], return_aux=True)
    ops.set_gemm_emulation(0)
    return losses.cpu(), {n: p.grad.clone() for n, p in hip.named_parameters() if p.grad is not None}, aux
for mode in (0, 6):
    la, ga, aa = run(True, mode)
    lb, gb, ab = run(False, mode)
    lc, gc, ac = run(True, mode)
    print(f"mode {mode}: losses fused {la.tolist()[:3]} two-pass {lb.tolist()[:3]}; repeat identical: {all(torch.equal(ga[n], gc[n]) for n in ga)}")
    print("  label diffs:", {k: int((aa[k] != ab[k]).sum()) for k in ("mask_w", "mask_w_other", "mclip", "mclip_other")},
          " pred_x max diff", (aa["pred_x"] - ab["pred_x"]).abs().max().item())
    rows = sorted((((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-20)).item(), n) for n in ga)
    print("  largest fused-vs-two-pass rel-L2:", [(f"{v:.1e}", n) for v, n in rows[-12:]])
    print("  smallest:", [(f"{v:.1e}", n) for v, n in rows[:4]])
