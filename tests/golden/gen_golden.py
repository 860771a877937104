"""Golden-vector generator.  Runs ONLY in the build container: it imports the real reference modules from
/root/reference (through tests/golden/_ref_shim.py for the un-vendored deps), drives them with a restatement of the
semivl.py:223-328 loop body, checks the oracle restatement (oracle/semivl_oracle.py) against them, and writes small
fixtures (inputs + expected outputs, never reference source) to tests/golden/*.npz.

    python tests/golden/gen_golden.py            # regenerates fixtures, prints oracle-vs-reference deltas

Fixture 'tiny'  : everything small; weights, inputs and all outputs are stored.
Fixture 'vlgdim': real VLG decoder dims (channels 128, 4 heads of 64) on a tiny ViT; weights/inputs are regenerated
                  from seeds (checksums stored), outputs stored.
"""
import os
import runpy
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from golden_util import seeded_state, smooth_batch, tie_masks  # noqa: E402  (shared with the tests: one weight / input stream)

TEXT = "configs/_base_/datasets/text_embedding/voc12_wbg_single.npy"
MCC_TEXT = "configs/_base_/datasets/text_embedding/voc12_wbg_concept4_single.npy"

CONFIGS = {
    "tiny": dict(S=128, B=1, embed=64, layers=3, heads=4, out_indices=[0, 1, 3], channels=32, text_channels=32,
                 dec_heads=1, up=(32, 16), skip=(16, 16), seed=11, conf_thresh=0.95),
    # live confidence gate WITH A MARGIN: peaked predictions spread the confidences over (0.05, 1); 'auto' places the
    # threshold in the middle of the widest gap between any two pixel confidences of (conf_w, conf_w_other) inside their
    # 25 % ... 75 % quantile range and requires that gap to be >= 2e-4, i.e. every pixel is >= 1e-4 away from the gate: a
    # legitimate 1e-7 change of the forward rounding (a fused normalisation, a different reduction order) cannot move a
    # pixel across it.  (Round 2's fixed 0.058 sat inside a dense cloud of confidences: one flipped pixel shifted every
    # decoder gradient by 0.5 % and froze the forward kernels bit for bit.)
    "vlgdim": dict(S=128, B=1, embed=64, layers=3, heads=1, out_indices=[0, 1, 3], channels=128, text_channels=128,
                   dec_heads=4, up=(64, 32), skip=(32, 16), seed=12, conf_thresh="auto", logit_gain=150.0,
                   smooth_inputs=True),
    # off-size crop (72 -> corner-padded to 80 -> 5x5 patches vs a 4x4 trained pos grid: per-forward bicubic pos resize,
    # AvgPool floor 5 -> 1) with the Cityscapes recipe's conf_mode 'pixelavg' and batch 2
    "offsize": dict(S=72, B=2, embed=64, layers=3, heads=1, out_indices=[0, 1, 3], channels=32, text_channels=32,
                    dec_heads=1, up=(32, 16), skip=(16, 16), seed=13, conf_thresh=0.95, conf_mode="pixelavg"),
    # Cityscapes-recipe wiring (vlm-vlg-aspp-s2p4-skr04-ftap-mcvitb): ViT out_indices [k, L], conv_encoder (stand-in
    # ResNetV1c, see _ref_shim.py) as the second skip source, renorm_clip_img, pixelavg
    "skr": dict(S=96, B=2, embed=64, layers=3, heads=1, out_indices=[1, 3], channels=32, text_channels=32,
                dec_heads=1, up=(32, 16), skip=(16, 16), seed=14, conf_thresh=0.05, conf_mode="pixelavg",
                conv_encoder=True),
    # peaked predictions (decoder's last conv x150, low-passed inputs): with conf_mode 'pixelwise' at the shipped threshold
    # 0.95 a large share of the pixels passes the confidence gate, so loss_s1 / loss_s2 / loss_fp are non-zero against
    # the reference's own confidence_weighted_loss (train_utils.py:30-49), which the reference run below really calls
    "conf": dict(S=128, B=2, embed=64, layers=3, heads=1, out_indices=[0, 1, 3], channels=32, text_channels=32,
                 dec_heads=1, up=(32, 16), skip=(16, 16), seed=15, conf_thresh=0.95, logit_gain=150.0,
                 smooth_inputs=True),
}


def build_reference(c):
    import model.vlm as ref_vlm  # noqa: registers VLM
    import model.decode_heads.vlg_head  # noqa: registers VLGHead
    import third_party.maskclip.models.backbones.maskclip_vit  # noqa: registers the ViT
    from model.builder import forward_wrapper
    import types
    skr = bool(c.get("conv_encoder"))
    mcfg = runpy.run_path("configs/_base_/models/vlm-vlg-aspp-s2p4-%s-ftap-mcvitb.py" % ("skr04" if skr else "sk04"))["model"]
    ccfg = runpy.run_path("configs/_base_/models/mcvit16.py")["backbone"]
    S = c["S"]
    for bb in (mcfg["backbone"], ccfg):
        bb.update(img_size=(S, S), embed_dims=c["embed"], num_layers=c["layers"], num_heads=c["heads"])
        bb.pop("pretrained", None)
    mcfg["backbone"]["out_indices"] = c["out_indices"]
    mcfg["decode_head"].update(img_size=S, num_classes=21, text_channels=c["text_channels"], up_channels=c["up"],
                               skip_in_channels=(c["embed"], 256 if skr else c["embed"]), skip_channels=c["skip"],
                               num_heads=c["dec_heads"], channels=c["channels"])
    if skr:
        mcfg["conv_encoder"].pop("pretrained", None)
        mcfg["renorm_clip_img"] = True          # experiments.py:219-220 (exp 44)
    mcfg.pop("type")
    mcfg.pop("pretrained", None)
    for bb in (mcfg["backbone"], ccfg):
        bb.pop("type", None)
    mcfg["backbone"]["type"] = "MaskClipVisionTransformer"
    ccfg["type"] = "MaskClipVisionTransformer"
    mcfg["decode_head"]["type"] = "VLGHead"
    m = ref_vlm.VLM(load_text_embedding=TEXT, load_mcc_text_embedding=MCC_TEXT, load_pl_text_embedding=TEXT,
                    clip_encoder=ccfg, maskclip_class_filter=None, **mcfg)
    m.disable_dropout, m.fp_rate = True, 0.5
    m.forward = types.MethodType(forward_wrapper, m)
    return m


class MaskFeeder:
    """Replaces F.dropout2d's RNG by injected {0,1} channel masks (same scaling 1/(1-p))."""

    def __init__(self, masks):
        self.masks, self.i = masks, 0

    def __call__(self, f, p=0.5, training=True, inplace=False):
        m = self.masks[self.i % len(self.masks)]
        self.i += 1
        return f * m[:, :, None, None] / (1.0 - p)


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    import _ref_shim
    _ref_shim.install()
    from oracle import semivl_oracle as O
    from model.text_embeddings import get_class_to_concept_idxs
    cls2con = get_class_to_concept_idxs(MCC_TEXT)
    # The reference's OWN loss helpers drive the reference run: utils/train_utils.py:19-49 imports as it is, semivl.py
    # (whose top-level imports resolve through the shim) provides compute_mc_loss (:52-58), bound to its module globals
    # `criterion_mc` / `mcc_loss_reduce` exactly as its __main__ block sets them (:156-162).
    import semivl as ref_semivl
    import utils.train_utils as ref_tu

    def ref_cwl(loss, conf, ign, conf_mode, conf_thresh):
        return ref_tu.confidence_weighted_loss(loss, conf, ign, dict(conf_mode=conf_mode, conf_thresh=conf_thresh))

    def ref_mc(pred, mask, ign, reduce):
        ref_semivl.mcc_loss_reduce = reduce
        ref_semivl.criterion_mc = (torch.nn.CrossEntropyLoss(ignore_index=255) if reduce == "mean" else
                                   torch.nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
        return ref_semivl.compute_mc_loss(pred, mask, ign)

    REF_HELPERS = (ref_tu.cutmix_img_, ref_tu.cutmix_mask, ref_cwl, ref_mc)
    text = torch.from_numpy(np.load(TEXT))
    mcc = torch.from_numpy(np.load(MCC_TEXT))

    only = sys.argv[1:]
    for name, c in CONFIGS.items():
        if only and name not in only:
            continue
        torch.manual_seed(c["seed"])
        ref = build_reference(c)
        shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
        sd = seeded_state(shapes, c["seed"], c.get("logit_gain"))
        ref.load_state_dict(sd, strict=True)

        orc = O.build_vlm(dict(nclass=21, crop=c["S"], embed=c["embed"], layers=c["layers"], heads=c["heads"],
                               out_indices=tuple(c["out_indices"]), channels=c["channels"],
                               text_channels=c["text_channels"], up=c["up"],
                               skip_in=(c["embed"], 256 if c.get("conv_encoder") else c["embed"]),
                               skip=c["skip"], conv_encoder=bool(c.get("conv_encoder")),
                               renorm_clip_img=bool(c.get("conv_encoder"))), text, mcc, cls2con)
        if c["dec_heads"] != 4:
            for lyr in orc.decode_head.layers:
                lyr.transformer.attn.attn.num_heads = c["dec_heads"]
        missing = orc.load_state_dict(sd, strict=True)

        B, S = c["B"], c["S"]
        batch = O.synthetic_batch(B, S, 21, seed=1234 + c["seed"])
        if c.get("smooth_inputs"):
            batch = smooth_batch(batch)
        g = torch.Generator().manual_seed(c["seed"] + 100)
        fp_ch = (c["embed"], 512, 256) if c.get("conv_encoder") else (c["embed"], c["embed"], 512)  # dropout2d call order
        fp_masks = [(torch.rand(2 * B, ch, generator=g) > 0.5).float() for ch in fp_ch]
        total_iters, iters = 100, 10
        if c["conf_thresh"] == "auto":
            with torch.no_grad():
                _, a0 = O.semivl_step(orc, batch, iters, total_iters, conf_thresh=0.0, fp_masks=fp_masks)
            vals = torch.cat([a0["conf_w"].flatten(), a0["pred_w_other"].softmax(1).max(1).values.flatten()]).double().sort().values
            lo, hi = vals[int(0.25 * len(vals))], vals[int(0.75 * len(vals))]
            gaps = vals[1:] - vals[:-1]
            mid = 0.5 * (vals[1:] + vals[:-1])
            gaps = torch.where((mid > lo) & (mid < hi), gaps, torch.zeros_like(gaps))
            j = int(gaps.argmax())
            assert gaps[j].item() >= 2e-4, f"no confidence gap >= 2e-4 in the middle half ({gaps[j].item():.2e})"
            c = dict(c, conf_thresh=round(float(mid[j]), 6))
            assert abs(c["conf_thresh"] - float(mid[j])) < 1e-6
            print(f"[{name}] auto conf_thresh {c['conf_thresh']} (gap {gaps[j].item():.2e}; {float((vals >= c['conf_thresh']).double().mean()):.2f} of the pixels pass)")

        # ---- reference run (its modules, the restated loop) ---------------------------------------------
        def run(model, is_ref):
            model.zero_grad()
            if is_ref:
                feeder = MaskFeeder(fp_masks)
                orig = F.dropout2d
                F.dropout2d = feeder
                try:
                    class Adapter:
                        def __init__(s, m):
                            s.m = m
                        def eval(s):
                            s.m.eval()
                        def train(s):
                            s.m.train()
                        def __call__(s, img, need_fp=False, fp_masks=None):
                            return s.m(img, need_fp=need_fp)
                        def forward_maskclip(s, img, t):
                            return s.m.forward_maskclip(img, t)
                    loss, aux = O.semivl_step(Adapter(model), batch, iters, total_iters, conf_thresh=c["conf_thresh"],
                                              conf_mode=c.get("conf_mode", "pixelwise"), fp_masks=fp_masks,
                                              helpers=REF_HELPERS)
                finally:
                    F.dropout2d = orig
            else:
                loss, aux = O.semivl_step(model, batch, iters, total_iters, conf_thresh=c["conf_thresh"],
                                          conf_mode=c.get("conf_mode", "pixelwise"), fp_masks=fp_masks)
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
            return loss.detach(), aux, grads

        # eval forward + maskclip on img_x (before the step: a train-mode pass updates BatchNorm running statistics)
        ref.eval()
        with torch.no_grad():
            logits_eval = ref(batch["img_x"])
            mclip_x = ref.forward_maskclip(batch["img_x"], 0.9)
            orc.eval()
            o_mx, o_top2 = orc.forward_maskclip(batch["img_x"], 0.9, True)
            assert torch.equal(o_mx, mclip_x)
            tie_mx = ((o_top2[:, 0] - o_top2[:, 1]) < 1e-6) | ((o_top2[:, 0] - 0.9).abs() < 1e-6)

        rl, raux, rg = run(ref, True)
        ol, oaux, og = run(orc, False)
        print(f"[{name}] loss ref {rl.item():.8f} oracle {ol.item():.8f}  |d| {abs(rl.item() - ol.item()):.2e}")
        for k in ("loss_x", "loss_s1", "loss_s2", "loss_fp", "loss_mc_s1", "loss_mc_s2", "loss_mc_fp"):
            print(f"    {k:11s} ref {raux[k].item():.8f}  |d| {abs(raux[k].item() - oaux[k].item()):.2e}")
            # reference modules + reference loss helpers vs the oracle's restatement of both
            assert abs(raux[k].item() - oaux[k].item()) <= 1e-6 * max(1.0, abs(raux[k].item())), k
        assert abs(rl.item() - ol.item()) <= 1e-6 * max(1.0, abs(rl.item()))
        # the other reductions of the guidance loss and the other confidence modes, helper against helper on this
        # fixture's own tensors (semivl.py:52-58,156-162; train_utils.py:30-49)
        with torch.no_grad():
            px, cw, ig = raux["pred_s1"].detach(), raux["conf_w"], batch["ignore_mask"]
            ce = F.cross_entropy(px, raux["mask_w"], reduction="none")
            for mode in ("pixelwise", "pixelratio", "pixelavg"):
                a_, b_ = ref_cwl(ce, cw, ig, mode, c["conf_thresh"]), O.confidence_weighted_loss(ce, cw, ig, mode, c["conf_thresh"])
                assert torch.equal(a_, b_), (mode, a_, b_)
            for red in ("mean", "mean_valid", "mean_all"):
                a_, b_ = ref_mc(px, raux["mclip"], ig, red), O.compute_mc_loss(px, raux["mclip"], ig, red)
                assert torch.equal(a_, b_), (red, a_, b_)
            m1 = ref_tu.cutmix_mask(raux["mask_w"], raux["mask_w_other"], batch["mix1"])
            assert torch.equal(m1, O.cutmix_mask(raux["mask_w"], raux["mask_w_other"], batch["mix1"]))
            i1, i2 = batch["img_s1"].clone(), batch["img_s1"].clone()
            ref_tu.cutmix_img_(i1, batch["img_s1_other"], batch["mix1"])
            O.cutmix_img_(i2, batch["img_s1_other"], batch["mix1"])
            assert torch.equal(i1, i2)
        for k in ("mask_w", "mask_w_other", "mclip", "mclip_other"):
            assert torch.equal(raux[k], oaux[k]), f"{k} differs"
        # (the pseudo-label tie budget is a LOGIT gap: fixtures whose last conv is scaled by logit_gain scale it alike)
        ties = tie_masks(oaux, B, eps=1e-6 * c.get("logit_gain", 1.0))
        print("    fp-tie pixels:", {k: int(v.sum()) for k, v in ties.items()})
        print("    frac conf_w >= thr:", (raux["conf_w"] >= c["conf_thresh"]).float().mean().item())
        print("    label maps bit-exact; conf_w |d|", (raux["conf_w"] - oaux["conf_w"]).abs().max().item(),
              " pred_x |d|", (raux["pred_x"] - oaux["pred_x"]).abs().max().item())
        assert set(rg) == set(og), (set(rg) ^ set(og))
        worst = max(((rg[k] - og[k]).abs().max() / (rg[k].abs().max() + 1e-12)).item() for k in rg)
        print(f"    {len(rg)} grads, worst rel max-err {worst:.2e}; trainable ref={sum(p.requires_grad for p in ref.parameters())}")
        assert worst < 1e-4

        # one AdamW step with the reference hyper-parameters (exp 40): lr 1e-4, wd 0.01, backbone x0.01, head x10
        ck = dict(backbone=dict(lr_mult=0.01), text_encoder=dict(lr_mult=0.0), conv_encoder=dict(lr_mult=1.0),
                  norm=dict(decay_mult=0.0), ln=dict(decay_mult=0.0), head=dict(lr_mult=10.0))
        groups = [gr for gr in O.param_groups(ref, 1e-4, 0.01, ck) if gr["params"][0].grad is not None]
        names = [gr.pop("name") for gr in groups]
        opt = torch.optim.AdamW(groups, lr=1e-4, weight_decay=0.01)
        opt.step()
        after = {k: v.detach().clone() for k, v in ref.state_dict().items() if k in rg}

        out = dict(
            cfg=np.array(repr(c)), iters=np.array([iters, total_iters]),
            loss=rl.numpy(), **{k: raux[k].detach().numpy() for k in ("loss_x", "loss_s1", "loss_s2", "loss_fp",
                                                                       "loss_mc_s1", "loss_mc_s2", "loss_mc_fp")},
            mask_w=raux["mask_w"].numpy().astype(np.uint8), mask_w_other=raux["mask_w_other"].numpy().astype(np.uint8),
            mclip=raux["mclip"].numpy().astype(np.uint8), mclip_other=raux["mclip_other"].numpy().astype(np.uint8),
            conf_w=raux["conf_w"].numpy().astype(np.float32),
            pred_x_s4=raux["pred_x"].detach()[:, :, ::4, ::4].numpy(), logits_eval_s4=logits_eval[:, :, ::4, ::4].numpy(),
            mclip_x=mclip_x.numpy().astype(np.uint8),
            fp_masks=np.concatenate([m.numpy().ravel() for m in fp_masks]).astype(np.uint8),
            **{"tie/" + k: np.packbits(v.numpy()) for k, v in ties.items()}, **{"tie/mclip_x": np.packbits(tie_mx.numpy())},
            grad_names=np.array(sorted(rg)), opt_group_lr=np.array([gr["lr"] for gr in groups]),
            opt_group_wd=np.array([gr["weight_decay"] for gr in groups]), opt_group_names=np.array(names),
        )
        for k in sorted(rg):
            out["gnorm/" + k] = np.array([rg[k].norm().item(), rg[k].flatten()[0].item(), rg[k].flatten()[-1].item()])
            out["after/" + k] = np.array([after[k].double().sum().item(), after[k].flatten()[0].item(),
                                          after[k].flatten()[-1].item()])
        if name == "tiny":
            for k, v in sd.items():
                out["w/" + k] = v.numpy()
            for k, v in batch.items():
                out["in/" + k] = v.numpy() if v.dtype != torch.int64 else v.numpy().astype(np.uint8)
            for k in sorted(rg):
                out["grad/" + k] = rg[k].numpy()
        else:
            out["w_checksum"] = np.array([sum(v.double().sum().item() for v in sd.values()),
                                          sum(v.double().abs().sum().item() for v in sd.values())])
            out["in_checksum"] = np.array([sum(v.double().sum().item() for v in batch.values())])
        path = os.path.join(HERE, f"semivl_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"    wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


if __name__ == "__main__":
    main()
