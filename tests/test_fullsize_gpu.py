"""Full-size parity (BASELINE config dims: CLIP ViT-B/16 + VLG head at every class count of BASELINE.json's configs --
VOC N=21, COCO N=81, ADE N=150 at 512x512 and the Cityscapes recipe N=19 at 801x801 with the ResNetV1c side encoder) at
batch 1: the product on MI355X against the oracle restatement on the host CPU, same seeded weights and inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TEXT_DIR = "configs/_base_/datasets/text_embedding/"


def build_pair(dev, nclass=21, dataset="pascal", crop=512, seed=4242):
    from golden_util import PKG, seeded_state
    from oracle import semivl_oracle as O
    from semivl_amd.model.builder import build_model
    from semivl_amd.model.text_embeddings import get_class_to_concept_idxs
    from semivl_amd.synthetic import exp40_cfg
    cfg = exp40_cfg(1, crop, nclass, dataset)
    hip = build_model(cfg)
    sd = seeded_state([(k, tuple(v.shape)) for k, v in hip.state_dict().items()], seed)
    hip.load_state_dict(sd, strict=True)
    prefix = {"pascal": "voc12_wbg", "cityscapes": "cityscapes", "coco": "coco", "ade": "ade"}[dataset]
    tpath = f"{TEXT_DIR}{prefix}_{cfg['text_embedding_variant']}.npy"
    mpath = f"{TEXT_DIR}{prefix}_{cfg['mcc_text']}.npy"
    t = torch.from_numpy(np.load(os.path.join(PKG, tpath)))
    m = torch.from_numpy(np.load(os.path.join(PKG, mpath)))
    cls2con = get_class_to_concept_idxs(mpath) if m.shape[0] != nclass else None
    ocfg = dict(nclass=nclass, crop=crop)
    kw = {}
    if dataset == "cityscapes":    # exp 44: skr04 model, CLIP renorm; the frozen CLIP keeps its 512^2 position grid
        ocfg.update(out_indices=(4, 12), skip_in=(768, 256), skip=(32, 32), conv_encoder=True, renorm_clip_img=True)
        kw["clip_img_size"] = 512
    orc = O.build_vlm(ocfg, t, m, cls2con, **kw)
    orc.load_state_dict(sd, strict=True)
    return cfg, hip.to(dev), orc


def oracle_step(orc, cfg, batch, masks, conf_thresh, label_override=None):
    from oracle import semivl_oracle as O
    for p_ in orc.parameters():
        p_.grad = None
    loss, aux = O.semivl_step(orc, batch, 100, 1000, conf_thresh=conf_thresh, conf_mode=cfg["conf_mode"],
                              fp_masks=masks, label_override=label_override)
    loss.backward()
    return loss, aux


LOGIT_ERR_BOUND = 1.5e-5     # asserted on every logit tensor below (measured: 4.5e-6 ... 9.5e-6 over all configs and modes)
TIE_EPS = 2 * LOGIT_ERR_BOUND   # a pseudo-label may differ only where the ORACLE's top-2 logit gap is below this constant
GRAD_REL_L2 = 4e-3            # per-tensor ||g - g_ref|| / ||g_ref|| (measured <= 3.0e-3: skip_proj.1 / layer-0 attention / pos_embed,
GRAD_MAX_ERR = 6e-3           # the far ends of the chains); element-wise bound in units of the tensor's largest entry (<= 4.1e-3)
# The tensors BEHIND the class-shared features at the far end of the backward chain: the gradient of a skip feature is the
# sum of N class-images' input gradients that nearly cancel at random init, so the ~1e-6 rounding differences between two
# valid fp32 evaluations come out multiplied by ~3e3 in these four (and only these) tensors.  Measured on this step, same
# data, the product's own arithmetic variants against the fp32 oracle: exact fp32 MFMA 2.1 - 2.7e-3; split products with
# the K = 128 row streams on the fp32 pipe 2.8 - 3.5e-3, with them on the split pipe 5.1 - 7.0e-3; one GroupNorm statistic
# moved by 1 ulp 2.9 -> 5.2e-3 (round 4) -- a draw per rounding pattern, not an accuracy ranking: against float64
# (test_fullsize_gradient_error_against_fp64, its own seed) both arithmetic modes stay within the asserted 3e-3 on every
# tensor with either routing of those launches, and the fp32 oracle itself is 1.5e-3 away on skip_proj.1.0.weight.  They
# get their own bound; every other tensor keeps GRAD_REL_L2.
SHARED_FAR_END = ("backbone.pos_embed", "backbone.layers.0.attn.attn.in_proj_weight", "backbone.layers.0.attn.attn.out_proj.weight",
                  "decode_head.skip_proj.1.0.weight")
GRAD_REL_L2_SHARED = 1.2e-2     # split-product modes only; the exact fp32 mode keeps GRAD_REL_L2 on these tensors too
MCLIP_TIE = 1e-5                # MaskCLIP label ties: top-2 probability gap / distance to the threshold (round 4: 1e-4)
# Ratchets (round-4 review): what this tree MEASURES on each full-size case -- (most label flips in one map, worst rel-L2 over
# the ordinary tensors, worst rel-L2 over SHARED_FAR_END) -- asserted with 30 % headroom (+2 flips), so that a regression
# inside the generous fixed bounds above still fails.  The kernels are deterministic: the numbers only move when the
# arithmetic of a kernel changes, and then they are re-measured (the test prints `RATCHET measured`).
# Mode 6 re-measured when the fused attention moved to fp16 x 2 operands (round 5, csrc/attn_h2.hip; the exact-mode rows did
# not move): pascal (14, 2.1e-3, 4.3e-3) -> (13, 2.0e-3, 4.2e-3), coco (2, 4.8e-4, 7.4e-4) -> (3, 3.0e-4, 6.7e-4),
# ade (5, 5.4e-4, 8.9e-4) -> (6, 6.4e-4, 1.27e-3) -- on coco / ade still less than half of the exact fp32 mode's distance.
# Round 6 (slab sums in double, fp16 x 2 ASPP forward + in_proj weight gradient, MFMA tail kernels for the leftover attention
# rows, GroupNorm-backward sums from the dgrad epilogue): mode 6 re-measured -- pascal (13, 2.0e-3, 4.2e-3) -> (21, 2.0e-3, 4.2e-3)
# (the flip count is now the exact mode's: WHICH near-ties flip is a draw per rounding pattern), coco (3, 3.0e-4, 6.7e-4) ->
# (2, 4.0e-4, 7.2e-4), ade (6, 6.4e-4, 1.27e-3) -> (5, 5.7e-4, 8.2e-4); the exact-mode rows did not move.
RATCHET = {("pascal", 0): (21, 1.3e-3, 2.7e-3), ("pascal", 6): (21, 1.9e-3, 2.8e-3),
           ("coco", 0): (2, 8.2e-4, 1.1e-3), ("coco", 6): (1, 3.8e-4, 7.1e-4),
           ("ade", 0): (6, 1.4e-3, 2.8e-3), ("ade", 6): (3, 5.0e-4, 7.6e-4)}


def check_step(dev, cfg, hip, orc, batch, masks, loss, aux, gemm_mode=0, bn_stack=(), after_bn=(), after_bn_tol=None,
               ratchet=None):
    """Losses / logits within north_star's 1e-3 (logits in fact within LOGIT_ERR_BOUND); label maps bit-exact except at fp
    ties of the oracle, with a FIXED budget: a flip needs the oracle's top-2 logit gap to be below TIE_EPS (MaskCLIP
    labels: top-2 probability gap or distance to the 0.9 threshold below MCLIP_TIE -- its softmax runs at temperature 100);
    every parameter gradient against the oracle's: rel-L2 <= GRAD_REL_L2 and every element within GRAD_MAX_ERR of the
    tensor's scale.  `bn_stack`: parameter-name fragments of ReLU-on-BatchNorm stacks (3e-2: pre-activations within
    rounding of 0 flip, test_model_gpu.py::test_conv_encoder_matches_oracle); `after_bn`: tensors fed by such a stack's
    features, held to `after_bn_tol`."""
    from golden_util import assert_labels
    from semivl_amd import ops
    from semivl_amd.train import LOSS_NAMES, semivl_train_step
    for p_ in hip.parameters():
        p_.grad = None
        if hasattr(p_, "main_grad"):
            p_.main_grad = None
    ops.set_gemm_emulation(gemm_mode)
    try:
        losses, haux = semivl_train_step(hip, {k: v.to(dev) for k, v in batch.items()}, 100, 1000, cfg,
                                         fp_masks=[m.to(dev) for m in masks], return_aux=True)
    finally:
        ops.set_gemm_emulation(0)
    got = dict(zip(LOSS_NAMES, losses.cpu().tolist()))
    assert abs(got["loss"] - loss.item()) < 1e-3, (got["loss"], loss.item())
    for k in LOSS_NAMES[1:]:
        assert abs(got[k] - aux[k].item()) < 1e-3, (k, got[k], aux[k].item())
    errs = {}
    for k in ("pred_x", "pred_s1", "pred_w", "pred_w_other"):
        errs[k] = (haux[k].cpu() - aux[k].detach()).abs().max().item()
        assert errs[k] < LOGIT_ERR_BOUND, (k, errs[k])   # (north_star's logits tolerance is 1e-3)
    B = batch["img_w"].shape[0]

    def gap(logits):
        t = logits.detach().topk(2, dim=1).values
        return t[:, 0] - t[:, 1]
    t2 = aux["mclip_top2"]
    mtie = ((t2[:, 0] - t2[:, 1]) < MCLIP_TIE) | ((t2[:, 0] - cfg["mcc_conf_thresh"]).abs() < MCLIP_TIE)
    ties = dict(mask_w=gap(aux["pred_w"]) <= TIE_EPS, mask_w_other=gap(aux["pred_w_other"]) <= TIE_EPS,
                mclip=mtie[:B], mclip_other=mtie[B:])
    flips = {k: assert_labels(haux[k].cpu().numpy(), aux[k].numpy(), ties[k].numpy(), k)
             for k in ("mask_w", "mask_w_other", "mclip", "mclip_other")}
    for k, n in flips.items():
        assert n <= 5e-4 * aux[k].numel() and n <= int(ties[k].sum()), (k, n)   # (only tie pixels may flip: the tie budget)
    assert (haux["conf_w"].cpu() - aux["conf_w"]).abs().max().item() < 1e-4
    og = {n: p.grad for n, p in orc.named_parameters() if p.grad is not None}
    hg = {n: p.grad for n, p in hip.named_parameters() if p.grad is not None}
    assert sorted(og) == sorted(hg) and len(og) > 100, sorted(set(og) ^ set(hg))

    def worst_rel(ref):
        return max(((hg[n].cpu() - ref[n]).norm() / (ref[n].norm() + 1e-20)).item() for n in ref
                   if n != "decode_head.head.bias" and not any(s_ in n for s_ in bn_stack) and
                   not any(s_ in n for s_ in after_bn) and n not in SHARED_FAR_END)
    # Pseudo-label ties.  The label maps above differ from the oracle's at a handful of pixels whose top-2 logit gap is
    # below either implementation's rounding error -- legitimately, but at random init a gradient tensor is an incoherent
    # sum over ~5e5 pixels, so ONE flipped target moves it by ~1/sqrt(#pixels) ~ 1.4e-3 of its norm, and WHICH ties flip
    # changes with any reordering of a sum (round 4: statistics from a conv epilogue moved backbone.pos_embed from 2.9e-3
    # to 5.2e-3 with 12 instead of 14 flips).  When the plain comparison is out of bounds and ties did flip, the gradients
    # are compared under the SAME tie decisions: the oracle's step is repeated with the product's label maps.
    if sum(flips.values()) > 0 and worst_rel(og) >= GRAD_REL_L2:
        keep = {n: g.clone() for n, g in og.items()}
        override = {k: haux[k].cpu() for k in ("mask_w", "mask_w_other", "mclip", "mclip_other")}
        oracle_step(orc, cfg, batch, masks, cfg["conf_thresh"], label_override=override)
        og = {n: p.grad.clone() for n, p in orc.named_parameters() if p.grad is not None}
        for n, p_ in orc.named_parameters():      # (the module-scoped oracle serves the other arithmetic mode next)
            if n in keep:
                p_.grad = keep[n]
        print(f"[gemm_mode {gemm_mode}] gradients compared under the product's tie decisions ({flips})")
    table = sorted(((hg[n].cpu() - og[n]).abs().max().item() / max(og[n].abs().max().item(), 1e-12), n) for n in og)
    t2_ = sorted((((hg[n].cpu() - og[n]).norm() / (og[n].norm() + 1e-20)).item(), n) for n in og)
    print(f"[gemm_mode {gemm_mode}] logit errs {errs}, label flips at ties {flips}; largest grad max-err / scale:",
          [(f"{v:.1e}", n) for v, n in table[-6:]], "; largest rel-L2:", [(f"{v:.1e}", n) for v, n in t2_[-10:]],
          "; head rel-L2:", [(f"{v:.1e}", n) for v, n in t2_ if ("up2" in n or "skip_proj" in n or "up1" in n)][-8:])
    plain = [v for v, n in t2_ if n != "decode_head.head.bias" and not any(s_ in n for s_ in bn_stack) and
             not any(s_ in n for s_ in after_bn) and n not in SHARED_FAR_END]
    shared = [v for v, n in t2_ if n in SHARED_FAR_END]
    meas = (max(flips.values()), max(plain), max(shared) if shared else 0.0)
    print(f"RATCHET measured [gemm_mode {gemm_mode}]: flips {meas[0]}, rel-L2 {meas[1]:.2e}, shared far end {meas[2]:.2e}; asserted: {ratchet}")
    if ratchet is not None:
        assert meas[0] <= int(1.3 * ratchet[0]) + 2, ("label flips ratchet", meas, ratchet)
        assert meas[1] <= 1.3 * ratchet[1] and meas[2] <= 1.3 * ratchet[2], ("gradient rel-L2 ratchet", meas, ratchet)
    for n in og:
        ref = og[n]
        err = (hg[n].cpu() - ref).abs().max().item()
        scale = ref.abs().max().item()
        rel2 = ((hg[n].cpu() - ref).norm() / (ref.norm() + 1e-20)).item()
        if n == "decode_head.head.bias":   # sum(softmax - onehot) over all pixels: exactly 0 in exact arithmetic, rounding noise here
            assert err < 1e-5, (n, err)     # (the noise grows with the number of class planes summed)
            continue
        if any(s_ in n for s_ in bn_stack):
            assert rel2 < 3e-2 or ref.norm().item() < 1e-7, f"{n}: rel L2 {rel2}"
            continue
        tol2 = after_bn_tol if (after_bn_tol is not None and any(s_ in n for s_ in after_bn)) else GRAD_REL_L2
        if n in SHARED_FAR_END and gemm_mode != 0:
            tol2 = max(tol2, GRAD_REL_L2_SHARED)
        assert rel2 < tol2, f"{n}: grad rel L2 {rel2} (bound {tol2})"
        assert err < (GRAD_MAX_ERR if tol2 == GRAD_REL_L2 else 1.5 * tol2) * scale + 1e-8, f"{n}: grad max err {err} vs scale {scale}"
    return haux


def fp_masks_for(chans, seed=5, b=2):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(b, c, generator=g) > 0.5).float() for c in chans]


@pytest.fixture(scope="module")
def fullsize_case():
    """Oracle side of the full-size VOC step, computed once for both GEMM arithmetic modes."""
    from oracle import semivl_oracle as O
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, torch.get_num_threads()))
    cfg, hip, orc = build_pair(dev)
    batch = O.synthetic_batch(2, 512, 21, seed=99)      # B = 2: the memory plan's sample chunks are crossed at full size
    masks = fp_masks_for((768, 768, 512), b=4)
    # random-init confidences are ~1/21: conf_thresh = 0 keeps the whole unsupervised CE term alive WITHOUT putting
    # thousands of pixels within rounding distance of the threshold (at 0.06 a 2e-6 logit perturbation moved pixels in
    # and out of the loss and the gradients by several per cent -- thresholding itself is covered by the fixtures)
    cfg = dict(cfg, conf_thresh=0.0, head_chunk_class_images=2 * 21)   # two samples per chunk: three chunks per decode
    loss, aux = oracle_step(orc, cfg, batch, masks, 0.0)
    return cfg, hip, orc, batch, masks, loss, aux


@pytest.mark.parametrize("gemm_mode", [0, 6])
def test_fullsize_step_matches_oracle(dev, fullsize_case, gemm_mode):
    """gemm_mode 0: exact fp32 MFMA; 6: the ViT linears on the bf16 pipe (3-way split, 6 products) -- same tolerances."""
    cfg, hip, orc, batch, masks, loss, aux = fullsize_case
    check_step(dev, cfg, hip, orc, batch, masks, loss, aux, gemm_mode, ratchet=RATCHET[("pascal", gemm_mode)])


# Ratchet of the float64 comparison below (round 6): worst ratio (product's distance from float64 / the fp32 oracle's own
# distance + 1e-4) per arithmetic mode and tensor family, as THIS tree measures it, asserted with 20 % headroom on top of
# the fixed 4 x cap -- a regression inside the cap still fails.  Re-measured when a kernel's arithmetic changes (the test
# prints `FP64 RATCHET measured`, and writes the record bench.py quotes as `numerics`).  The largest entry belongs to the
# EXACT fp32 mode: the decoder's ASPP weight gradients, a k-ordered fp32 chain per split-K slab against the host library's
# blocked summation -- summation order, not operand precision; the split-product modes sit at or below it on those tensors.
FP64_FAMILIES = (("vit", "backbone."), ("aspp", "decode_head.aspp"), ("up", "decode_head.up"), ("head_other", "decode_head."))
FP64_RATCHET = {0: dict(vit=1.83, aspp=3.65, up=1.43, head_other=2.63), 6: dict(vit=1.39, aspp=1.17, up=2.48, head_other=1.37)}


def _fp64_family(name):
    return next(f for f, pre in FP64_FAMILIES if name.startswith(pre))


def test_fullsize_gradient_error_against_fp64(dev):
    """Where the 4e-3 of GRAD_REL_L2 comes from: the same step with the ORACLE in float64.  The fp32 oracle itself is up
    to 1.5e-3 (rel-L2) away from it on the tensors at the far ends of the chains (the first block's attention projections,
    the skip projection of v0); the product -- in BOTH arithmetic modes alike -- must stay within 3e-3 of the float64
    gradients (5e-3 on the four SHARED_FAR_END tensors, whose sums over class-images nearly cancel) and within 4x the fp32
    oracle's own distance + 1e-4, and within 1.2 x the per-family ratios this tree measured (FP64_RATCHET).
    Pseudo-label ties: ONE target that flips against the float64 run moves every gradient tensor of this random-init step
    by ~1e-3 of its norm, and WHICH near-ties flip changes with any change of a kernel's rounding.  Whenever label maps
    differ, every differing pixel must be a near-tie of the float64 run itself (top-2 logit gap <= TIE_EPS; MaskCLIP:
    MCLIP_TIE), there may be at most 5e-4 of the map of them, and the float64 step is repeated with the PRODUCT's label maps:
    the gradients are always compared under the same decisions."""
    import copy
    import json
    from oracle import semivl_oracle as O
    from semivl_amd import ops
    from semivl_amd.train import semivl_train_step
    torch.set_num_threads(min(32, torch.get_num_threads()))
    cfg, hip, orc = build_pair(dev, seed=4243)
    cfg = dict(cfg, conf_thresh=0.0)
    batch = O.synthetic_batch(1, 512, 21, seed=98)
    masks = fp_masks_for((768, 768, 512), seed=6)
    oracle_step(orc, cfg, batch, masks, 0.0)
    g32 = {n: p.grad.clone() for n, p in orc.named_parameters() if p.grad is not None}
    orc64 = copy.deepcopy(orc).double()
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    _, aux64 = oracle_step(orc64, cfg, b64, [m.double() for m in masks], 0.0)
    g64 = {n: p.grad.clone() for n, p in orc64.named_parameters() if p.grad is not None}
    rel = lambda a, b: ((a.double() - b).norm() / (b.norm() + 1e-30)).item()
    own = {n: rel(g32[n], g64[n]) for n in g32}
    cap = lambda n: 5e-3 if n in SHARED_FAR_END else 3e-3
    within = lambda n, e_, o_: e_ <= cap(n) and e_ <= 4 * (o_ + 1e-4)
    maps = ("mask_w", "mask_w_other", "mclip", "mclip_other")
    B = batch["img_w"].shape[0]

    def gap(logits):
        t = logits.detach().topk(2, dim=1).values
        return t[:, 0] - t[:, 1]
    t2 = aux64["mclip_top2"]
    mtie = ((t2[:, 0] - t2[:, 1]) < MCLIP_TIE) | ((t2[:, 0] - cfg["mcc_conf_thresh"]).abs() < MCLIP_TIE)
    ties = dict(mask_w=gap(aux64["pred_w"]) <= TIE_EPS, mask_w_other=gap(aux64["pred_w_other"]) <= TIE_EPS,
                mclip=mtie[:B], mclip_other=mtie[B:])
    rows, record = [], {}
    for mode in (0, 6):
        for p_ in hip.parameters():
            p_.grad = None
        ops.set_gemm_emulation(mode)
        try:
            _, haux = semivl_train_step(hip, {k: v.to(dev) for k, v in batch.items()}, 100, 1000, cfg,
                                        fp_masks=[m.to(dev) for m in masks], return_aux=True)
        finally:
            ops.set_gemm_emulation(0)
        hg = {n: p_.grad.cpu() for n, p_ in hip.named_parameters() if p_.grad is not None and n != "decode_head.head.bias"}
        diff = {k: haux[k].cpu() != aux64[k] for k in maps}
        flips = {k: int(d_.sum()) for k, d_ in diff.items()}
        ref64 = g64
        if sum(flips.values()) > 0:
            for k, d_ in diff.items():     # every flipped pixel is a near-tie of the float64 run itself
                assert not bool((d_ & ~ties[k]).any()), (k, flips[k], int((d_ & ~ties[k]).sum()))
            assert sum(flips.values()) <= 5e-4 * aux64["mask_w"].numel(), flips       # (a handful of near-ties, not a drift)
            oracle_step(orc64, cfg, b64, [m.double() for m in masks], 0.0, label_override={k: haux[k].cpu() for k in maps})
            ref64 = {n: p.grad.clone() for n, p in orc64.named_parameters() if p.grad is not None}
            print(f"[gemm_mode {mode}] float64 gradients recomputed under the product's tie decisions ({flips})")
        mine = [(rel(g, ref64[n]) / (own[n] + 1e-4), mode, n, rel(g, ref64[n]), own[n]) for n, g in hg.items()]
        rows += mine
        fam = {}
        for r_, _, n_, e_, _ in mine:
            f_ = _fp64_family(n_)
            if r_ > fam.get(f_, (0.0,))[0]:
                fam[f_] = (r_, n_, e_)
        record[mode] = dict(flips=flips, worst_ratio={f_: round(v[0], 2) for f_, v in fam.items()},
                            worst_tensor={f_: v[1] for f_, v in fam.items()},
                            rel_l2_vs_fp64={f_: float(f"{v[2]:.3e}") for f_, v in fam.items()})
    top = sorted(((v, n) for n, v in own.items() if n != "decode_head.head.bias"))[-4:]
    rows.sort()
    print("fp32 oracle vs fp64 (rel-L2), largest:", [(f"{v:.1e}", n) for v, n in top], "; product vs fp64, worst ratios to the oracle's own error:",
          [(f"{r_:.1f}x", m_, n_, f"{e_:.1e}", f"{o_:.1e}") for r_, m_, n_, e_, o_ in rows[-14:]])
    print("FP64 RATCHET measured:", json.dumps({str(m_): r_["worst_ratio"] for m_, r_ in record.items()}), "; asserted:", FP64_RATCHET)
    try:     # the record bench.py quotes as `numerics` (copied to profiles/ by the round-end script)
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out_dir, exist_ok=True)
        json.dump(dict(test="tests/test_fullsize_gpu.py::test_fullsize_gradient_error_against_fp64",
                       meaning="per arithmetic mode (0 = exact fp32 MFMA, 6 = split products) and tensor family: worst "
                               "ratio of the product's rel-L2 distance from the FLOAT64 oracle's gradients to the fp32 oracle's "
                               "own distance (+ 1e-4), full-size VOC step at B = 1, same tie decisions",
                       fp32_oracle_own_largest=[(float(f"{v:.2e}"), n) for v, n in top],
                       modes={str(m_): r_ for m_, r_ in record.items()}, cap="4 x and 3e-3 (5e-3: SHARED_FAR_END)"),
                  open(os.path.join(out_dir, "numerics_fp64.json"), "w"), indent=1)
    except OSError:
        pass
    for r_, m_, n_, e_, o_ in rows:
        assert within(n_, e_, o_), f"mode {m_} {n_}: {e_:.2e} vs the fp32 oracle's own {o_:.2e}"
    for m_, r_ in record.items():
        for f_, v_ in r_["worst_ratio"].items():
            lim = FP64_RATCHET[m_].get(f_)
            assert lim is None or v_ <= 1.2 * lim, ("fp64 ratchet", m_, f_, v_, lim, r_["worst_tensor"][f_])


@pytest.mark.parametrize("nclass,dataset", [(81, "coco"), (150, "ade")])
def test_fullsize_step_large_class_counts(dev, nclass, dataset):
    """BASELINE configs 4 / 5 (experiments.py:373-424): the whole training step at N = 81 / 150 class-images per image
    (SemanticTransformer sequence length N, N-wide similarity GEMM, CE over N planes, chunked decoder) against the oracle."""
    from oracle import semivl_oracle as O
    torch.set_num_threads(min(64, torch.get_num_threads()))
    cfg, hip, orc = build_pair(dev, nclass, dataset, seed=4300 + nclass)
    cfg = dict(cfg, conf_thresh=0.0, head_chunk_class_images=2 * nclass)   # two samples per chunk: several chunks at B = 1
    batch = O.synthetic_batch(1, 512, nclass, seed=100 + nclass)
    masks = fp_masks_for((768, 768, 512), seed=nclass)
    loss, aux = oracle_step(orc, cfg, batch, masks, 0.0)
    check_step(dev, cfg, hip, orc, batch, masks, loss, aux, ratchet=RATCHET[(dataset, 0)])
    # the arithmetic bench.py measures by default (GEMMs, attention, tiled / ASPP convolutions as split products)
    check_step(dev, cfg, hip, orc, batch, masks, loss, aux, gemm_mode=6, ratchet=RATCHET[(dataset, 6)])
    if dataset == "coco":
        reduced_precision_mode_check(dev, cfg, hip, batch, masks, loss, aux, {n: p.grad for n, p in orc.named_parameters()
                                                                              if p.grad is not None})
    hip.decode_head.chunk_class_images = 1344


BF16X3_GRAD_REL_L2 = 1e-2      # (measured 3.4e-3 on skip_proj.1.0.weight, median 6e-4: printed as `BF16X3 measured`)


def reduced_precision_mode_check(dev, cfg, hip, batch, masks, loss, aux, og):
    """BASELINE configs[4] names "COCO ... + fp16 mixed precision".  The reference has NO reduced-precision path (SURVEY D2:
    no autocast / GradScaler / .half() anywhere), so there is nothing to be in parity with: `--gemm-arith bf16x3`
    (svl_set_gemm_emulation(3): every operand of the large GEMMs as TWO bf16 terms, three 16-bit products, fp32 accumulate)
    is a BUILD-ONLY MODE, NO REFERENCE COUNTERPART.  What is asserted is that the mode is a usable approximation of the
    fp32 step on the COCO-size problem: the 8 loss terms and every logit within north_star's 1e-3 of the fp32 oracle, every
    parameter gradient within BF16X3_GRAD_REL_L2 (rel-L2).  The default arithmetic (fp16 x 2 / bf16 x 3 split products at
    fp32-chain accuracy) is what `value` measures and what the tight bounds of check_step hold."""
    from semivl_amd import ops
    from semivl_amd.train import LOSS_NAMES, semivl_train_step
    for p_ in hip.parameters():
        p_.grad = None
    ops.set_gemm_emulation(3)
    try:
        losses, haux = semivl_train_step(hip, {k: v.to(dev) for k, v in batch.items()}, 100, 1000, cfg,
                                         fp_masks=[m.to(dev) for m in masks], return_aux=True)
    finally:
        ops.set_gemm_emulation(0)
    got = dict(zip(LOSS_NAMES, losses.cpu().tolist()))
    assert abs(got["loss"] - loss.item()) < 1e-3, (got["loss"], loss.item())
    for k in LOSS_NAMES[1:]:
        assert abs(got[k] - aux[k].item()) < 1e-3, (k, got[k], aux[k].item())
    lerr = max((haux[k].cpu() - aux[k].detach()).abs().max().item() for k in ("pred_x", "pred_s1", "pred_w", "pred_w_other"))
    assert lerr < 1e-3, lerr
    hg = {n: p.grad.cpu() for n, p in hip.named_parameters() if p.grad is not None}
    rel = sorted((((hg[n] - og[n]).norm() / (og[n].norm() + 1e-20)).item(), n) for n in og if n != "decode_head.head.bias")
    print(f"BF16X3 measured (build-only mode, no reference counterpart): loss {got['loss']:.6f} vs {loss.item():.6f}, "
          f"max logit err {lerr:.2e}, worst grad rel-L2 {rel[-1][0]:.2e} ({rel[-1][1]}), median {rel[len(rel) // 2][0]:.2e}")
    assert rel[-1][0] < BF16X3_GRAD_REL_L2, rel[-4:]


def test_fullsize_cityscapes_recipe(dev):
    """BASELINE config 3 as the reference defines it (experiments.py:428-456): N=19 at 801x801 (51x51 ragged patch grid,
    per-forward bicubic pos-embed resize, AvgPool floor 51 -> 12), skr04 model with the ResNetV1c side encoder as the
    second skip source, CLIP re-normalisation, cityscapes_conceptavg3 text / concept3 (54 -> 19) MaskCLIP aggregation,
    conf_mode 'pixelavg'."""
    from oracle import semivl_oracle as O
    torch.set_num_threads(min(64, torch.get_num_threads()))
    cfg, hip, orc = build_pair(dev, 19, "cityscapes", crop=801, seed=4319)
    with torch.no_grad():   # zero-init bn3 of the torchvision-style bottlenecks would silence the residual branches
        sd = orc.state_dict()
        g = torch.Generator().manual_seed(9)
        for n, p in orc.named_parameters():
            if "conv_encoder" in n and p.dim() == 1:
                r = torch.randn(p.shape, generator=g)
                p.copy_(1.0 + 0.2 * r if n.endswith("weight") else 0.1 * r)
        hip.load_state_dict({k: v.clone() for k, v in orc.state_dict().items()}, strict=True)
    assert cfg["conf_mode"] == "pixelavg" and cfg["model"].endswith("skr04-ftap-mcvitb")
    batch = O.synthetic_batch(1, 801, 19, seed=119)
    masks = fp_masks_for((768, 512, 256), seed=19)     # dropout2d call order: [v4, emb, conv feature]
    loss, aux = oracle_step(orc, cfg, batch, masks, cfg["conf_thresh"])
    start = {k: v.clone() for k, v in hip.state_dict().items()}
    for gemm_mode in (0, 6):   # 6 = the arithmetic bench.py measures by default; T = 2602 tokens: a ragged last attention block
        hip.load_state_dict(start, strict=True)
        # the side encoder itself is a ReLU-on-BatchNorm stack (sign flips at rounding level); the decoder blocks its features
        # feed (skip_proj.1 -> up2, and up1 through the shared upstream gradient) are held to the standard bound in exact
        # arithmetic (measured <= 1.3e-3) and to 1.2e-2 in mode 6, where the side encoder's convolutions run as split
        # products and flip a different set of pre-activations than the fp32 oracle (measured <= 8.0e-3)
        check_step(dev, cfg, hip, orc, batch, masks, loss, aux, gemm_mode=gemm_mode, bn_stack=("conv_encoder",),
                   after_bn=("skip_proj.1", "up2", "up1"), after_bn_tol=1.2e-2 if gemm_mode == 6 else None)
        for (n, bo), (_, bh) in zip(sorted(orc.named_buffers()), sorted(hip.named_buffers())):
            if "conv_encoder" in n:   # SyncBN running statistics after the train-mode passes
                assert (bh.cpu().float() - bo.float()).abs().max().item() < 1e-4 * max(1.0, bo.float().abs().max().item()), n


@pytest.mark.parametrize("nclass,dataset,crop,B", [(21, "pascal", 512, 16), (81, "coco", 512, 16), (19, "cityscapes", 801, 8)])
def test_forward_is_batch_invariant_at_baseline_batch(dev, nclass, dataset, crop, B):
    """Oracle parity runs at B <= 2; BASELINE.json's configs run B = 16 (8 at 801^2) per GPU.  The bridge is a property of
    the forward in the arithmetic bench.py measures (mode 6): every op is per sample, so
      * the ViT encoder's features (planes GEMMs with per-row scales, fused attention, LayerNorm) of sample i from the
        full-batch forward equal the two-sample forward's BIT FOR BIT -- a tile map, a split or a scale that depended on the
        batch would show;
      * the decoder's logits agree to rounding (measured 2.1e-6, asserted 1e-5 -- north_star's tolerance is 1e-3): its
        kernel SELECTION depends on the row count (the short-K row stream serves launches of >= 32768 rows, split-K plans
        follow the tile count), i.e. the same sums in another order, not another computation;
      * MaskCLIP labels are identical, pseudo labels wherever the top-2 logit gap exceeds that rounding."""
    from golden_util import seeded_state
    from semivl_amd import ops
    from semivl_amd.model.builder import build_model
    from semivl_amd.synthetic import exp40_cfg, synthetic_batch
    cfg = exp40_cfg(B, crop, nclass, dataset)
    hip = build_model(cfg)
    hip.load_state_dict(seeded_state([(k, tuple(v.shape)) for k, v in hip.state_dict().items()], 5150 + nclass), strict=True)
    hip.to(dev).eval()
    img = synthetic_batch(B, crop, nclass, seed=77, device=dev)["img_w"]
    ops.set_gemm_emulation(6)
    worst = 0.0
    try:
        with torch.no_grad():
            full = hip(img)
            lab = hip.forward_maskclip(img, 0.9)
            feats, _ = hip.backbone.forward_tokens(hip.renormalize_img_for_clip(img), need_global=False)
            for s0 in (0, B - 2):
                sub = img[s0:s0 + 2].contiguous()
                fsub, _ = hip.backbone.forward_tokens(hip.renormalize_img_for_clip(sub), need_global=False)
                for fa, fb in zip(feats, fsub):
                    assert torch.equal(fa[s0:s0 + 2], fb), f"encoder features of samples {s0}, {s0 + 1} depend on the batch ({dataset}, B = {B})"
                part = hip(sub)
                worst = max(worst, float((full[s0:s0 + 2] - part).abs().max()))
                assert torch.equal(lab[s0:s0 + 2], hip.forward_maskclip(sub, 0.9))
                t2 = part.topk(2, dim=1).values
                clear = (t2[:, 0] - t2[:, 1]) > 1e-5
                assert torch.equal(full[s0:s0 + 2].argmax(1)[clear], part.argmax(1)[clear])
    finally:
        ops.set_gemm_emulation(0)
    print(f"[{dataset} B={B}] logits of the full batch vs two-sample forwards: max |diff| {worst:.2e}")
    assert worst < 1e-5 and full.shape == (B, nclass, crop, crop) and torch.isfinite(full).all()


def test_fullsize_ade150_head_forward(dev):
    """N = 150 classes (ADE config): eval forward of the full model vs the oracle (exercises the seq=150 attention,
    N=150 similarity GEMM and class-batched decoder at 150 class-images per image)."""
    from golden_util import PKG, assert_labels, seeded_state
    from oracle import semivl_oracle as O
    from semivl_amd.model.builder import build_model
    from semivl_amd.synthetic import exp40_cfg
    cfg = exp40_cfg(1, 512, 150, "ade")
    hip = build_model(cfg)
    sd = seeded_state([(k, tuple(v.shape)) for k, v in hip.state_dict().items()], 777)
    hip.load_state_dict(sd, strict=True)
    t = torch.from_numpy(np.load(os.path.join(PKG, "configs/_base_/datasets/text_embedding/ade_single.npy")))
    orc = O.build_vlm(dict(nclass=150, crop=512), t, t)
    orc.load_state_dict(sd, strict=True)
    img = O.synthetic_batch(1, 512, 150, seed=3)["img_x"]
    hip.to(dev).eval()
    orc.eval()
    with torch.no_grad():
        ref = orc(img)
        out = hip(img.to(dev))
        rm, top2 = orc.forward_maskclip(img, 0.9, True)
        hm = hip.forward_maskclip(img.to(dev), 0.9)
    assert (out.cpu() - ref).abs().max().item() < 1e-3
    tie = ((top2[:, 0] - top2[:, 1]) < MCLIP_TIE) | ((top2[:, 0] - 0.9).abs() < MCLIP_TIE)
    assert_labels(hm.cpu().numpy(), rm.numpy(), tie.numpy(), "maskclip labels")
