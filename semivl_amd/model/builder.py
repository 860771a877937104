"""`build_model(cfg)` and the `VLM` segmentor — the drop-in model API of the reference
(/root/reference model/builder.py:56-159, model/vlm.py:27-127) on the HIP kernel library.

`build_model` consumes the SAME flat experiment dict the reference's `experiments.py` generates:
`model`, `nclass`, `crop_size`, `dataset`, `text_embedding_variant`, `mcc_text`, `pl_text`, `clip_encoder`,
`model_args`, `disable_dropout`, `fp_rate`.  Model hyper-parameter files `configs/_base_/models/<name>.py` are read
from the working directory when present (the reference resolves them relative to cwd, builder.py:110), otherwise the
package's own restatement of the VLG configs is used.
"""
import copy
import os
import runpy

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .text_embeddings import aggregate_concept_predictions, get_class_to_concept_idxs
from .resnet import ResNetV1c
from .vit import MaskClipVisionTransformer
from .vlg_head import VLGHead

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BACKBONES = {"MaskClipVisionTransformer": MaskClipVisionTransformer, "ResNetV1c": ResNetV1c}
HEADS = {"VLGHead": VLGHead}
SEGMENTORS = {}


def _resolve(path):
    """cwd-relative first (reference behaviour), then the package copy of configs/_base_/..."""
    if os.path.exists(path):
        return path
    alt = os.path.join(_PKG, path)
    if os.path.exists(alt):
        return alt
    raise FileNotFoundError(path)


def _vit_cfg(img_size, out_indices):
    return dict(type="MaskClipVisionTransformer", img_size=(img_size, img_size), patch_size=16, patch_bias=False,
                in_channels=3, embed_dims=768, num_layers=12, num_heads=12, mlp_ratio=4, out_indices=out_indices,
                qkv_bias=True, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, with_cls_token=True,
                output_cls_token=False, norm_cfg=dict(type="LN", eps=1e-6), act_cfg=dict(type="GELU"),
                patch_norm=False, pre_norm=True, final_norm=True, return_clip_embed=True, return_qkv=True,
                interpolate_mode="bicubic", num_fcs=2, norm_eval=False)


def builtin_model_cfg(name):
    """Own restatement of the hyper-parameters of configs/_base_/models/{vlm-vlg-aspp-s2p4-sk04-ftap-mcvitb,mcvit16}.py."""
    if name == "vlm-vlg-aspp-s2p4-sk04-ftap-mcvitb":
        return dict(img_size=512, model=dict(
            type="VLM", pretrained="pretrained/clip2mmseg_ViT16_clip_backbone.pth",
            backbone=_vit_cfg(512, [0, 4, 12]),
            decode_head=dict(type="VLGHead", img_size=512, num_classes=19, text_in_channels=512, text_channels=128,
                             up_channels=(64, 32), skip_in_channels=(768, 768), skip_channels=(32, 16),
                             skip_from_conv_feat=False, num_layers=2, num_heads=4, channels=128, pool_size=(4, 4),
                             conv1_ksize=7, align_corners=False, loss_decode=None),
            freeze_backbone=True, exclude_keys=["attn", "pos_embed"]))
    if name == "vlm-vlg-aspp-s2p4-skr04-ftap-mcvitb":   # Cityscapes recipe: side conv encoder as the second skip source
        c = builtin_model_cfg("vlm-vlg-aspp-s2p4-sk04-ftap-mcvitb")
        c["model"]["backbone"] = _vit_cfg(512, [4, 12])
        c["model"]["conv_encoder"] = dict(type="ResNetV1c", pretrained="pretrained/resnet101_v1c-e67eebb6.pth",
                                          depth=101, num_stages=1, out_indices=[0], dilations=[1], strides=[1],
                                          norm_cfg=dict(type="SyncBN", requires_grad=True), style="pytorch",
                                          contract_dilation=True)
        c["model"]["decode_head"].update(skip_in_channels=(768, 256), skip_channels=(32, 32), skip_from_conv_feat=True)
        return c
    if name == "mcvit16":
        bb = _vit_cfg(512, None)
        bb["pretrained"] = "pretrained/clip2mmseg_ViT16_clip_backbone.pth"
        return dict(img_size=512, backbone=bb)
    raise ValueError(f"no built-in model config '{name}' (SURVEY §8(f): skr04 / dlv3p / zegclip variants are next-row or out of scope)")


def load_model_cfg(name):
    path = f"configs/_base_/models/{name}.py"
    if os.path.exists(path):
        ns = runpy.run_path(path)
        return {k: v for k, v in ns.items() if not k.startswith("_")}
    return builtin_model_cfg(name)


def build_backbone(cfg):
    cfg = dict(cfg)
    return BACKBONES[cfg.pop("type")](**cfg)


def build_head(cfg):
    cfg = dict(cfg)
    return HEADS[cfg.pop("type")](**cfg)


class VLM(nn.Module):
    """model/vlm.py:27-127 (+ mmseg EncoderDecoder's attribute surface: backbone, decode_head, align_corners, num_classes)."""

    def __init__(self, backbone, decode_head, freeze_backbone=False, exclude_keys=None, load_text_embedding=None,
                 load_mcc_text_embedding=None, load_pl_text_embedding=None, clip_encoder=None, conv_encoder=None,
                 maskclip_class_filter=None, maskclip_trust_head=None, renorm_clip_img=False, pretrained=None,
                 train_cfg=None, test_cfg=None, neck=None, auxiliary_head=None, init_cfg=None, type=None):
        super().__init__()
        assert load_text_embedding == load_pl_text_embedding
        assert maskclip_class_filter is None and maskclip_trust_head is None
        backbone = dict(backbone)
        if pretrained is not None and backbone.get("pretrained") is None:
            backbone["pretrained"] = pretrained  # EncoderDecoder forwards `pretrained` to the backbone cfg
        self.backbone = build_backbone(backbone)
        self.decode_head = build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.local_iter = 0
        self.clip_encoder = build_backbone(clip_encoder) if clip_encoder is not None else None
        self.conv_encoder = build_backbone(conv_encoder) if conv_encoder is not None else None  # vlm.py:50-53
        self.load_text_embedding = load_text_embedding
        self.decode_head.load_text_embedding = load_text_embedding
        self.load_mcc_text_embedding = load_mcc_text_embedding
        self.renorm_clip_img = renorm_clip_img
        if not self.load_mcc_text_embedding:
            raise NotImplementedError
        self.loaded_mcc_text_feat = torch.from_numpy(np.load(_resolve(self.load_mcc_text_embedding))).float()
        # the reference re-reads the .npy on every extract_feat (vlm.py:116): cached, same values
        self._text_feat = torch.from_numpy(np.load(_resolve(self.load_text_embedding)))
        self._dev_cache = {}
        self.disable_dropout, self.fp_rate = True, 0.5
        if freeze_backbone:
            self.freeze(self.backbone, exclude_keys=exclude_keys)

    def renormalize_img_for_clip(self, img):  # vlm.py:69-78: loader (ImageNet) statistics -> CLIP statistics
        if not self.renorm_clip_img:
            return img
        k = ("renorm", str(img.device))
        if k not in self._dev_cache:
            lm, ls = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
            cm, cs = (torch.tensor([0.48145466, 0.4578275, 0.40821073]),
                      torch.tensor([0.26862954, 0.26130258, 0.27577711]))
            self._dev_cache[k] = ops.StreamCached(torch.stack((ls, lm, cm, cs)).contiguous().to(img.device))   # [4, 3]
        return ops.affine_planes(img, self._dev_cache[k].get())      # (img * ls + lm - cm) / cs, one pass

    def freeze(self, model, exclude_keys=None):  # vlm.py:80-88
        for n, m in model.named_parameters():
            m.requires_grad = False
            if exclude_keys is not None:
                assert isinstance(exclude_keys, list)
                for k in exclude_keys:
                    if str(k) in n:
                        m.requires_grad = True

    def init_weights(self):
        pass  # sub-modules initialise themselves at construction (pretrained file when present)

    def _on(self, name, t, device):
        key = (name, str(device))
        if key not in self._dev_cache:
            self._dev_cache[key] = ops.StreamCached(t.to(device))   # built on whichever stream asks first (train.py)
        return self._dev_cache[key].get()

    def text_feat(self, device):
        return self._on("text", self._text_feat, device)

    def text_feat_f32(self, device):
        """fp32 copy of the (fp16 on disk, vlm.py:116-117) text embedding, converted once per device instead of once per
        forward inside the head (`text.float()`, vlg_head.py:215)."""
        key = ("text_f32", str(device))
        if key not in self._dev_cache:
            self._dev_cache[key] = ops.StreamCached(self._text_feat.float().contiguous().to(device))
        return self._dev_cache[key].get()

    # -- MaskCLIP guidance (vlm.py:90-110) ---------------------------------------------------------------------
    def forward_maskclip(self, img, conf_tresh, ignore_mask=None):
        """int64 [b, H, W] in {0..N-1, 255}.  `ignore_mask` (optional, fused form of semivl.py:239-240): pixels where
        it equals 255 are set to 255."""
        with torch.no_grad():
            img = self.renormalize_img_for_clip(img)
            feats, _ = self.clip_encoder.forward_tokens(img, need_global=False)
            emb = feats[-1]  # [b, hw, 512]
            b, HW, Ce = emb.shape
            ps = self.clip_encoder.patch_size
            hp, wp = (img.shape[2] + ps - 1) // ps, (img.shape[3] + ps - 1) // ps
            text = self._on("mcc", self.loaded_mcc_text_feat, img.device)
            NC = text.shape[0]
            dense = ops.empty(b, NC, hp, wp, device=img.device)  # F.conv2d(visual_feat, text[:, :, None, None])
            ops.gemm(ops.A_KC, ops.B_KC, HW, NC, Ce, ops.Op(emb, Ce, 0, HW * Ce, 0), ops.Op(text, Ce), dense, ldc_m=1,
                     ldc_n=HW, batch=b, c_bso=NC * HW)
            if NC != self.num_classes:
                dense = aggregate_concept_predictions(dense, get_class_to_concept_idxs(self.load_mcc_text_embedding))
            assert dense.shape[1] == self.num_classes
            return ops.maskclip_labels(dense, img.shape[2], img.shape[3], 100.0, conf_tresh, ignore_mask)

    # -- features ------------------------------------------------------------------------------------------------
    def extract_feat(self, img):  # vlm.py:112-123 (reference return format)
        visual_feat = self.backbone(self.renormalize_img_for_clip(img))
        self.decode_head.load_text_embedding = self.load_text_embedding
        conv_feat = self.conv_encoder(img) if self.conv_encoder is not None else None
        return [visual_feat, self.text_feat(img.device), conv_feat]

    def _decode_head_forward_test(self, x, img_metas=None):  # vlm.py:125-127
        return self.decode_head.forward(x, force_output_pred_masks=True)["pred_masks"]

    # -- forward_wrapper (builder.py:56-102) ---------------------------------------------------------------------
    def head_res_size(self, in_size):
        """(h, w) of the decode head's own logit map for an input of `in_size`, when forward(head_res=True) would return
        it un-resized (the input is the training crop: ONE resize separates it from the loss), else None."""
        S_ = self.decode_head.image_size
        if tuple(in_size) != (S_, S_):
            return None
        ps = self.backbone.patch_size
        return (4 * ((in_size[0] + ps - 1) // ps), 4 * ((in_size[1] + ps - 1) // ps))

    def forward(self, img, gt=None, need_fp=False, only_fp=False, forward_mode="default", fp_masks=None,
                split_fp=True, fp_range=None, head_res=False):
        """Logits [b, N, H, W] at input resolution; with need_fp a 2-tuple (plain, feature-perturbed) halves.
        `fp_masks` (list of three {0,1} tensors [b, C_i]) injects the F.dropout2d channel masks for parity tests;
        `split_fp=False` returns the un-chunked [2b, ...] tensor; `fp_range=(s0, s1)` (with split_fp=False) perturbs
        and decodes only samples [s0, s1): output [b + s1 - s0, ...].  `head_res=True` (training crops only, see
        head_res_size): the logits are returned at the head's resolution [b, N, 4 hp, 4 wp] and the caller evaluates the
        resize of vlg_head.py:247 / builder.py:93-97 inside its loss kernels (ops.ce_up_fused / ops.softmax_max_up)."""
        if forward_mode == "maskclip_trust":    # builder.py:57-58 calls a method the reference never defines
            raise AttributeError("'VLM' object has no attribute 'train_maskclip_trust'")
        if forward_mode != "default":
            raise ValueError(forward_mode)
        S_ = self.decode_head.image_size
        in_size = tuple(img.shape[2:])
        feats, _ = self.backbone.forward_tokens(self.renormalize_img_for_clip(img), need_global=False)
        skip0_hw = None
        if self.conv_encoder is not None:   # vlm.py:119-121: the side encoder sees the loader-normalised image
            if len(feats) != 2:
                raise NotImplementedError("conv_encoder expects backbone.out_indices = [k, num_layers] (skr04)")
            ctok, skip0_hw = self.conv_encoder.forward_tokens(img)
        ps = self.backbone.patch_size
        hp, wp = (img.shape[2] + ps - 1) // ps, (img.shape[3] + ps - 1) // ps
        masks = None
        if only_fp or need_fp:
            masks = fp_masks
            drop_order = list(feats) + ([ctok] if self.conv_encoder is not None else [])  # builder.py:80-85
            if masks is None:  # F.dropout2d: one Bernoulli(1-p) draw per (sample, channel); always stochastic (App. E.7)
                masks = [ops.bernoulli((f.shape[0], f.shape[2]), 1.0 - self.fp_rate, img.device) for f in drop_order]
            assert len(masks) == len(drop_order)
            if only_fp:     # builder.py:65-77: every feature REPLACED by its channel-dropout copy (no doubling), then decoded
                sc = 1.0 / (1.0 - self.fp_rate)
                drop_order = [_ChanMaskFn.apply(f, mk.contiguous(), sc) for f, mk in zip(drop_order, masks)]
                feats = drop_order[:len(feats)]
                if self.conv_encoder is not None:
                    ctok = drop_order[-1]
                masks = None
            elif fp_range is not None:
                assert not split_fp, "fp_range returns the un-chunked tensor"
                masks = [mk[fp_range[0]:fp_range[1]] for mk in masks]
        if self.conv_encoder is not None:   # head slots [second-Up skip, first-Up skip, embedding]
            feats = [ctok, feats[0], feats[1]]
            if masks is not None:
                masks = [masks[2], masks[0], masks[1]]
        # the head resizes its 4x map to (img_size, img_size) (vlg_head.py:247), forward_wrapper then resizes to the input
        # size (builder.py:93-97): one and the same interpolation when the input IS img_size (every training crop), two
        # chained ones for evaluation windows of another shape (supervised.py:104-133)
        at_head = head_res and self.head_res_size(in_size) is not None
        out = self.decode_head.forward_tokens(feats, self.text_feat_f32(img.device), (hp, wp), masks, self.fp_rate,
                                              out_size=(4 * hp, 4 * wp) if at_head else (S_, S_),
                                              fp_range=fp_range if need_fp else None, skip0_hw=skip0_hw)
        if in_size != (S_, S_):
            out = _PlanesResizeFn.apply(out, in_size, self.align_corners)
        if need_fp and split_fp:
            return out.chunk(2)
        return out


class _ChanMaskFn(torch.autograd.Function):
    """F.dropout2d with a given {0,1} channel mask [b, C] on a token tensor [b, P, C]: y = x * mask / (1 - p); the same
    product backward (builder.py:68-72)."""

    @staticmethod
    def forward(ctx, f, mask, scale):
        ctx.save_for_backward(mask)
        ctx.scale = scale
        b, P, Cc = f.shape
        return ops.chanmask(f.contiguous().view(b * P, Cc), mask, scale, P).view(b, P, Cc)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        b, P, Cc = dy.shape
        return ops.chanmask(dy.contiguous().view(b * P, Cc), mask, ctx.scale, P).view(b, P, Cc), None, None


class _PlanesResizeFn(torch.autograd.Function):
    """F.interpolate(x [b, N, h, w], size, mode='bilinear', align_corners) on the library (mmseg.ops.resize, builder.py:93)."""

    @staticmethod
    def forward(ctx, x, size, align):
        ctx.geo = (x.shape[2], x.shape[3], size, align)
        return ops.bilinear_planes_fwd(x.contiguous(), x.shape[2], x.shape[3], align, size[0], size[1])

    @staticmethod
    def backward(ctx, dy):
        h, w, size, align = ctx.geo
        return ops.bilinear_planes_bwd(dy.contiguous(), h, w, align, size[0], size[1]), None, None


SEGMENTORS["VLM"] = VLM


def nested_set(dic, key, value):
    keys = key.split(".")
    for k in keys[:-1]:
        dic = dic.setdefault(k, {})
    dic[keys[-1]] = value


def build_model(cfg):
    """builder.py:104-159."""
    model_type = cfg["model"]
    if "mmseg." not in model_type:
        raise ValueError(model_type)  # 'deeplabv3plus' (UniMatch CNN baseline) is out of scope
    model_type = model_type.replace("mmseg.", "")
    mcfg = copy.deepcopy(load_model_cfg(model_type))
    mcfg["model"]["decode_head"]["num_classes"] = cfg["nclass"]
    if "zegclip" in model_type or "vlm" in model_type:
        if mcfg["img_size"] != cfg["crop_size"]:
            nested_set(mcfg, "img_size", cfg["crop_size"])
            nested_set(mcfg, "model.backbone.img_size", (cfg["crop_size"], cfg["crop_size"]))
            nested_set(mcfg, "model.decode_head.img_size", cfg["crop_size"])
        prefix = {"pascal": "voc12_wbg", "cityscapes": "cityscapes", "coco": "coco", "ade": "ade"}[cfg["dataset"]]
        base = "configs/_base_/datasets/text_embedding/"
        nested_set(mcfg, "model.load_text_embedding", f"{base}{prefix}_{cfg['text_embedding_variant']}.npy")
        nested_set(mcfg, "model.load_mcc_text_embedding", f"{base}{prefix}_{cfg['mcc_text']}.npy")
        nested_set(mcfg, "model.load_pl_text_embedding", f"{base}{prefix}_{cfg['pl_text']}.npy")
    if cfg.get("clip_encoder") is not None:
        ccfg = copy.deepcopy(load_model_cfg(cfg["clip_encoder"]))
        ccfg["img_size"] = mcfg["img_size"]
        if cfg.get("mcc_fix_resize_pos"):
            ccfg["backbone"]["img_size"] = mcfg["img_size"]
        mcfg["model"]["clip_encoder"] = ccfg["backbone"]
    if "model_args" in cfg:
        mcfg["model"].update(cfg["model_args"])
    if cfg.get("allow_random_init"):   # (not a reference key) synthetic-weight benchmarks / tests: no CLIP file on disk
        for sub in ("backbone", "clip_encoder", "conv_encoder"):
            if isinstance(mcfg["model"].get(sub), dict):
                mcfg["model"][sub]["allow_random_init"] = True
    mdict = dict(mcfg["model"])
    model = SEGMENTORS[mdict.pop("type")](**mdict)
    model.disable_dropout = cfg["disable_dropout"]
    model.fp_rate = cfg["fp_rate"]
    model.init_weights()
    return model
