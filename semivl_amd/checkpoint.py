"""Checkpoint I/O with the reference's key schema.

* `convert_clip_visual` -- OpenAI CLIP (`ViT-B-16.pt`) visual-tower state dict -> the mmseg-style file the reference's
  backbone loads (`third_party/maskclip/convert_clip_weights.py:21-62`, ViT branch; the ResNet branch is off the SemiVL
  path).  `MaskClipVisionTransformer.init_weights` (model/vit.py) consumes the result, bicubic pos-embed resize included
  (`maskclip_vit.py:378-410`).
* `save_checkpoint` / `load_checkpoint` -- the `{'model', 'optimizer', 'epoch'}` file of `semivl.py:426-433` and the
  loading rules of `third_party/unimatch/eval.py:131-139` (strip the DDP `module.` prefix, drop `clip_encoder.*`,
  optional `ema_model`), so the 'model' entry of files written here loads in the reference's eval script and vice versa
  (the 'optimizer' entry indexes only the parameters that can receive a gradient and is NOT interchangeable with the
  reference's per-parameter mmcv groups; `FusedAdamW.load_state_dict` verifies the stored parameter names).

Host-side plumbing only: nothing here runs inside a training step.
"""
import os

import torch


def load_clip_archive(path):
    """State dict of an OpenAI CLIP file: a TorchScript archive (the official `ViT-B-16.pt`) or a plain state dict."""
    try:
        return torch.jit.load(path, map_location="cpu").state_dict()
    except RuntimeError:
        sd = torch.load(path, map_location="cpu")
        return sd.get("state_dict", sd) if isinstance(sd, dict) else sd.state_dict()


def convert_clip_visual(clip_state_dict, backbone=True):
    """convert_clip_weights.py:21-62 and :80-84 for the ViT models.

    backbone=True  -> {'meta': {}, 'state_dict': {'backbone.<key>': fp32 tensor}}   (clip2mmseg_*_clip_backbone.pth)
    backbone=False -> {'proj': {'weight': W^T}, 'clip': {<key>: tensor}}              (clip2mmseg_*_clip_weights.pth)
    Only keys containing 'visual' are taken; `visual.proj` [width, embed] becomes `proj.weight` [embed, width] and is
    NOT prefixed."""
    out, proj = {}, None
    for key, val in clip_state_dict.items():
        if "visual" not in key:
            continue
        k = key[len("visual."):]
        val = val.float()
        if k == "proj":
            proj = val.t()
            out["proj.weight"] = proj
            continue
        if k == "class_embedding":
            k, val = "cls_token", val[None, None, :]
        elif k == "positional_embedding":
            k, val = "pos_embed", val[None, :, :]
        elif k == "conv1.weight":
            k = "patch_embed.projection.weight"
        elif "ln_pre" in k:
            k = "ln0." + k.split(".")[-1]
        elif "ln_post" in k:
            k = "ln1." + k.split(".")[-1]
        elif "transformer" in k:
            k = "layers." + k[len("transformer.resblocks."):]
            if "mlp" in k:
                k = k.replace("mlp", "ffn.layers")
            if "c_fc" in k:
                k = k.replace("c_fc", "0.0")
            if "c_proj" in k:
                k = k.replace("c_proj", "1")
            if "attn" in k:
                k = k.replace("attn", "attn.attn")
            elif "ln_" in k:
                k = k.replace("ln_", "ln")
        if backbone:
            k = "backbone." + k
        out[k] = val
    if backbone:
        return {"meta": {}, "state_dict": out}
    out.pop("proj.weight", None)
    res = {"clip": dict(out, **({"proj.weight": proj} if proj is not None else {}))}
    if proj is not None:
        res["proj"] = {"weight": proj}
    return res


def save_checkpoint(path, model, optimizer, epoch, ddp_prefix=True, ema_model=None):
    """semivl.py:426-433.  The reference saves the DDP-wrapped model, hence the `module.` prefix on every key."""
    pre = "module." if ddp_prefix else ""
    ck = {"model": {pre + k: v.detach().cpu() for k, v in model.state_dict().items()},
          "optimizer": optimizer.state_dict() if optimizer is not None else None,
          "epoch": epoch}
    if ema_model is not None:
        ck["ema_model"] = {pre + k: v.detach().cpu() for k, v in ema_model.state_dict().items()}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ck, path)
    return ck


def load_checkpoint(path_or_dict, model, ema=False, strict=True):
    """third_party/unimatch/eval.py:131-139.  Returns the stored epoch.  `clip_encoder.*` entries of the file are
    dropped (the frozen guidance encoder is rebuilt from the pretrained CLIP file, never from a checkpoint); with
    strict=True the model's own `clip_encoder.*` keys are likewise not required."""
    ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    src = ck["ema_model"] if ema else ck["model"]
    sd = {k.replace("module.", ""): v for k, v in src.items()}
    sd = {k: v for k, v in sd.items() if "clip_encoder" not in k}
    res = model.load_state_dict(sd, strict=False)
    if strict:
        missing = [k for k in res.missing_keys if "clip_encoder" not in k]
        if missing or res.unexpected_keys:
            raise RuntimeError(f"load_checkpoint: missing keys {missing[:8]}, unexpected keys {res.unexpected_keys[:8]}")
    return ck.get("epoch", -1)
