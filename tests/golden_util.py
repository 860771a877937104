"""Shared helpers for the golden-fixture tests (no reference access: fixtures + seeds only)."""
import ast
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TEXT = "configs/_base_/datasets/text_embedding/voc12_wbg_single.npy"
MCC_TEXT = "configs/_base_/datasets/text_embedding/voc12_wbg_concept4_single.npy"
PKG = os.path.join(os.path.dirname(HERE), "semivl_amd")


def load_fixture(name):
    z = np.load(os.path.join(HERE, "golden", f"semivl_{name}.npz"), allow_pickle=False)
    cfg = ast.literal_eval(str(z["cfg"]))
    return z, cfg


def seeded_state(named_shapes, seed, logit_gain=None):
    """Deterministic non-trivial parameters (shared with tests/golden/gen_golden.py): LN/GN gains 1+0.1n, biases 0.02n,
    cls/pos/weights 0.05n, conv/linear weights fan-in scaled so activations stay O(1).  `logit_gain` multiplies the
    decoder's last conv so that the softmax is peaked (confidences reach the 0.95 threshold, like a trained model's)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in named_shapes:
        r = torch.randn(*shape, generator=g) if len(shape) else torch.zeros(())
        leaf = name.split(".")[-1]
        parent = name.split(".")[-2] if "." in name else ""
        is_norm = parent.startswith("ln") or (leaf in ("weight", "bias") and len(shape) == 1 and
                                              any(s in name for s in (".1.weight", ".1.bias", ".2.weight", ".2.bias",
                                                                      ".4.weight", ".4.bias")))
        if leaf == "num_batches_tracked":          # BatchNorm buffers of the conv_encoder fixtures
            out[name] = torch.zeros(shape, dtype=torch.long)
        elif leaf == "running_var":
            out[name] = 1.0 + 0.1 * r.abs()
        elif leaf == "running_mean":
            out[name] = 0.05 * r
        elif (is_norm or parent.startswith("bn")) and leaf == "weight":
            out[name] = 1.0 + 0.1 * r
        elif leaf == "bias" or name.endswith("in_proj_bias"):
            out[name] = 0.02 * r
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            out[name] = r * (1.0 / np.sqrt(fan_in))
        else:
            out[name] = 0.05 * r
        if logit_gain is not None and name == "decode_head.head.weight":
            out[name] = out[name] * logit_gain
    return out


def tie_masks(aux, B, mcc_thresh=0.9, eps=1e-6):
    """Pixels where a label decision of the ORACLE sits on a floating-point tie: top-2 logit gap < eps for the pseudo
    labels, top-2 probability gap < eps or |certainty - threshold| < eps for the MaskCLIP guidance.  Label maps must be
    bit-exact everywhere else.  `aux` = the dict returned by oracle.semivl_step."""
    def gap(logits):
        t = logits.detach().topk(2, dim=1).values
        return (t[:, 0] - t[:, 1]) < eps
    out = dict(mask_w=gap(aux["pred_w"]), mask_w_other=gap(aux["pred_w_other"]))
    t2 = aux["mclip_top2"]
    eps = min(eps, 1e-6)     # (MaskCLIP certainties are probabilities of the frozen encoder: never scaled)
    tie = ((t2[:, 0] - t2[:, 1]) < eps) | ((t2[:, 0] - mcc_thresh).abs() < eps)
    out["mclip"], out["mclip_other"] = tie[:B], tie[B:]
    return out


def fixture_tie(z, key, shape):
    """Unpack the stored fp-tie mask of label map `key` (tests/golden/gen_golden.py)."""
    n = int(np.prod(shape))
    return np.unpackbits(z["tie/" + key])[:n].astype(bool).reshape(shape)


def assert_labels(got, ref, tie=None, what="labels"):
    """Bit-exact label maps; a mismatch is tolerated only where `tie` (see tie_masks) marks an fp tie of the checker."""
    got, ref = np.asarray(got), np.asarray(ref)
    bad = got != ref
    if not bad.any():
        return 0
    n = int(bad.sum())
    assert tie is not None, f"{what}: {n} mismatching pixels"
    off = int((bad & ~np.asarray(tie)).sum())
    assert off == 0, f"{what}: {n} mismatching pixels, {off} of them away from any fp tie"
    return n


def smooth_batch(batch, k=9):
    """Low-pass the synthetic images (box filter k x k, renormalised to unit variance): spatially coherent inputs give
    spatially coherent predictions, like natural crops do."""
    import torch.nn.functional as F
    out = dict(batch)
    for key, v in batch.items():
        if key.startswith("img_"):
            s_ = F.avg_pool2d(F.pad(v, (k // 2,) * 4, mode="reflect"), k, stride=1)
            out[key] = (s_ / s_.std()).contiguous()
    return out


def text_feats():
    t = torch.from_numpy(np.load(os.path.join(PKG, TEXT)))
    m = torch.from_numpy(np.load(os.path.join(PKG, MCC_TEXT)))
    return t, m


def build_oracle(c):
    from oracle import semivl_oracle as O
    from semivl_amd.model.text_embeddings import get_class_to_concept_idxs
    t, m = text_feats()
    skr = bool(c.get("conv_encoder"))   # Cityscapes-recipe wiring: side conv encoder as the second skip, CLIP renorm
    orc = O.build_vlm(dict(nclass=21, crop=c["S"], embed=c["embed"], layers=c["layers"], heads=c["heads"],
                           out_indices=tuple(c["out_indices"]), channels=c["channels"],
                           text_channels=c["text_channels"], up=c["up"],
                           skip_in=(c["embed"], 256 if skr else c["embed"]), skip=c["skip"], conv_encoder=skr,
                           renorm_clip_img=skr), t, m, get_class_to_concept_idxs(MCC_TEXT))
    for lyr in orc.decode_head.layers:
        lyr.transformer.attn.attn.num_heads = c["dec_heads"]
    return orc


def build_hip(c):
    """The product model with the fixture's (tiny) dimensions, built through the same cfg-dict constructors."""
    from semivl_amd.model.builder import VLM, builtin_model_cfg
    import copy
    mcfg = copy.deepcopy(builtin_model_cfg("vlm-vlg-aspp-s2p4-sk04-ftap-mcvitb"))["model"]
    ccfg = copy.deepcopy(builtin_model_cfg("mcvit16"))["backbone"]
    S = c["S"]
    for bb in (mcfg["backbone"], ccfg):
        bb.update(img_size=(S, S), embed_dims=c["embed"], num_layers=c["layers"], num_heads=c["heads"])
        bb.pop("pretrained", None)
    mcfg["backbone"]["out_indices"] = c["out_indices"]
    mcfg["decode_head"].update(img_size=S, num_classes=21, text_channels=c["text_channels"], up_channels=c["up"],
                               skip_in_channels=(c["embed"], c["embed"]), skip_channels=c["skip"],
                               num_heads=c["dec_heads"], channels=c["channels"])
    if c.get("conv_encoder"):
        skr = copy.deepcopy(builtin_model_cfg("vlm-vlg-aspp-s2p4-skr04-ftap-mcvitb"))["model"]
        skr["conv_encoder"].pop("pretrained", None)
        mcfg.update(conv_encoder=skr["conv_encoder"], renorm_clip_img=True)
        mcfg["decode_head"].update(skip_in_channels=(c["embed"], 256), skip_from_conv_feat=True)
    mcfg.pop("type")
    mcfg.pop("pretrained", None)
    return VLM(load_text_embedding=TEXT, load_mcc_text_embedding=MCC_TEXT, load_pl_text_embedding=TEXT,
               clip_encoder=ccfg, maskclip_class_filter=None, **mcfg)


def fixture_state(z, c, model):
    if "w_checksum" in z.files:
        shapes = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        sd = seeded_state(shapes, c["seed"], c.get("logit_gain"))
        chk = np.array([sum(v.double().sum().item() for v in sd.values()),
                        sum(v.double().abs().sum().item() for v in sd.values())])
        assert np.allclose(chk, z["w_checksum"], rtol=0, atol=1e-6), "seeded weight stream differs from the fixture's"
        return sd
    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w/")}


def fixture_batch(z, c):
    from oracle import semivl_oracle as O
    if "in_checksum" in z.files:
        batch = O.synthetic_batch(c["B"], c["S"], 21, seed=1234 + c["seed"])
        if c.get("smooth_inputs"):
            batch = smooth_batch(batch)
        chk = sum(v.double().sum().item() for v in batch.values())
        assert abs(chk - float(z["in_checksum"][0])) < 1e-6, "seeded input stream differs from the fixture's"
        return batch
    out = {}
    for k in z.files:
        if k.startswith("in/"):
            v = torch.from_numpy(z[k])
            out[k[3:]] = v.long() if v.dtype == torch.uint8 else v
    return out


def fixture_fp_masks(z, c):
    flat = torch.from_numpy(z["fp_masks"]).float()
    B = c["B"]
    chans = (c["embed"], 512, 256) if c.get("conv_encoder") else (c["embed"], c["embed"], 512)
    sizes = [2 * B * ch for ch in chans]
    parts = flat.split(sizes)
    return [p.view(2 * B, -1) for p in parts]
