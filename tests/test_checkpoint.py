"""Checkpoint I/O: the CLIP weight converter against golden vectors produced by the reference's own
converter script (tests/golden/gen_golden_ckpt.py), the pretrained-file loading path of the backbone (pos-embed resize
included) and the {'model','optimizer','epoch'} checkpoint round trip with the reference's eval.py loading rules."""
import os

import numpy as np
import pytest
import torch

from golden_util import build_hip, load_fixture

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    z = np.load(os.path.join(HERE, "golden", "clip_convert.npz"))
    take = lambda pre: {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
    return z, take("in::"), take("backbone::"), take("clip::")


def test_convert_clip_visual_matches_reference_script():
    from semivl_amd.checkpoint import convert_clip_visual
    z, src, ref_bb, ref_clip = _golden()
    src = {k: v.half() if v.dim() > 0 else v for k, v in src.items()}       # OpenAI CLIP weights are fp16
    bb = convert_clip_visual(src, backbone=True)
    assert sorted(bb.keys()) == ["meta", "state_dict"] and bb["meta"] == {}
    assert sorted(bb["state_dict"]) == sorted(ref_bb)
    for k, v in ref_bb.items():
        got = bb["state_dict"][k]
        assert got.dtype == torch.float32 and got.shape == v.shape and torch.equal(got, v), k
    assert "proj.weight" in bb["state_dict"] and not any(k.startswith("backbone.proj") for k in bb["state_dict"])
    al = convert_clip_visual(src, backbone=False)
    assert sorted(al.keys()) == ["clip", "proj"]
    assert sorted(al["clip"]) == sorted(ref_clip)
    for k, v in ref_clip.items():
        assert torch.equal(al["clip"][k], v), k
    assert torch.equal(al["proj"]["weight"], torch.from_numpy(z["proj::weight"]))


def test_backbone_loads_converted_file_with_pos_embed_resize(tmp_path):
    """maskclip_vit.py:378-410: 'backbone.' prefix stripped, 3x3 pos-embed grid resized (bicubic) to the model's 4x4,
    proj.weight [E, W] -> conv weight [E, W, 1, 1]."""
    from semivl_amd.checkpoint import convert_clip_visual
    from semivl_amd.model.vit import MaskClipVisionTransformer
    _, src, ref_bb, _ = _golden()
    f = tmp_path / "clip2mmseg_ViT16_clip_backbone.pth"
    torch.save(convert_clip_visual({k: v.half() if v.dim() > 0 else v for k, v in src.items()}, backbone=True), f)
    m = MaskClipVisionTransformer(img_size=(16, 16), patch_size=4, embed_dims=32, num_layers=2, num_heads=4,
                                  patch_bias=False, out_indices=[0, 2], pre_norm=True, final_norm=True, return_qkv=True,
                                  return_clip_embed=True, norm_cfg=dict(type="LN", eps=1e-6), pretrained=str(f))
    sd = m.state_dict()
    assert torch.equal(sd["layers.1.attn.attn.in_proj_weight"], ref_bb["backbone.layers.1.attn.attn.in_proj_weight"])
    assert torch.equal(sd["layers.0.ffn.layers.0.0.weight"], ref_bb["backbone.layers.0.ffn.layers.0.0.weight"])
    assert torch.equal(sd["ln0.bias"], ref_bb["backbone.ln0.bias"]) and torch.equal(sd["ln1.weight"], ref_bb["backbone.ln1.weight"])
    assert torch.equal(sd["patch_embed.projection.weight"], ref_bb["backbone.patch_embed.projection.weight"])
    assert torch.equal(sd["proj.weight"][:, :, 0, 0], ref_bb["proj.weight"])
    pe = ref_bb["backbone.pos_embed"]
    assert sd["pos_embed"].shape == (1, 17, 32) and torch.equal(sd["pos_embed"][:, 0], pe[:, 0])
    want = torch.nn.functional.interpolate(pe[:, 1:].reshape(1, 3, 3, 32).permute(0, 3, 1, 2), size=(4, 4),
                                           mode="bicubic", align_corners=False).flatten(2).transpose(1, 2)
    assert torch.allclose(sd["pos_embed"][:, 1:], want, atol=1e-6)


def test_checkpoint_roundtrip_follows_eval_py_rules(tmp_path):
    """semivl.py:426-433 / eval.py:131-139: 'module.' prefix on save, stripped on load; clip_encoder.* never restored."""
    from semivl_amd.checkpoint import load_checkpoint, save_checkpoint
    _, c = load_fixture("tiny")
    a, b = build_hip(c), build_hip(c)
    with torch.no_grad():
        for p in a.parameters():
            p.add_(torch.randn_like(p) * 0.01)
    ck = save_checkpoint(tmp_path / "best.pth", a, None, epoch=7)
    assert all(k.startswith("module.") for k in ck["model"]) and ck["epoch"] == 7
    assert sorted(k[len("module."):] for k in ck["model"]) == sorted(a.state_dict())
    clip_before = {k: v.clone() for k, v in b.state_dict().items() if "clip_encoder" in k}
    assert load_checkpoint(tmp_path / "best.pth", b) == 7
    for k, v in a.state_dict().items():
        if "clip_encoder" in k:
            assert torch.equal(b.state_dict()[k], clip_before[k]), k
        else:
            assert torch.equal(b.state_dict()[k], v), k
    ck["model"]["module.decode_head.bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        load_checkpoint(ck, b)
    ck["ema_model"] = {k: v + 1 for k, v in ck["model"].items() if "bogus" not in k}
    load_checkpoint(ck, b, ema=True)
    k0 = next(k for k in a.state_dict() if k.startswith("decode_head"))
    assert torch.equal(b.state_dict()[k0], a.state_dict()[k0] + 1)


@pytest.mark.gpu
def test_fused_adamw_state_dict_roundtrip(dev):
    """torch.optim.AdamW-shaped optimizer state: a restored optimizer continues bit-identically."""
    from semivl_amd.train import FusedAdamW
    _, c = load_fixture("tiny")
    ocfg = dict(type="AdamW", lr=1e-3, weight_decay=0.01,
                paramwise_cfg=dict(custom_keys=dict(backbone=dict(lr_mult=0.01), head=dict(lr_mult=10.0))))
    ma, mb = build_hip(c).to(dev), build_hip(c).to(dev)
    mb.load_state_dict(ma.state_dict())
    oa, ob = FusedAdamW(ma, ocfg), FusedAdamW(mb, ocfg)
    g = torch.Generator(device="cpu").manual_seed(0)
    grads = [torch.randn(oa.total, generator=g).to(dev) for _ in range(3)]
    for i in range(2):
        oa.g.copy_(grads[i]); oa.step(); oa.poly_lr(i + 1, 100)
    sd = oa.state_dict()
    assert len(sd["param_groups"]) == len(oa.groups) and len(sd["state"]) == len(oa.groups)
    assert sd["state"][0]["exp_avg"].shape == oa.groups[0]["param"].shape and float(sd["state"][0]["step"]) == 2.0
    tgroups = [dict(params=[torch.nn.Parameter(torch.zeros(g_["param"].shape))]) for g_ in oa.groups]
    topt = torch.optim.AdamW(tgroups, lr=1e-3)
    topt.load_state_dict({k: sd[k] for k in ("state", "param_groups")})      # the layout torch itself accepts
    mb.load_state_dict(ma.state_dict())
    ob.p.copy_(oa.p)
    ob.load_state_dict(sd)
    for o in (oa, ob):
        o.g.copy_(grads[2]); o.step()
    sa, sb = oa.state_dict(), ob.state_dict()          # per-tensor views (the arena's alignment padding is not state)
    for i, (ga, gb) in enumerate(zip(oa.groups, ob.groups)):
        assert torch.equal(ga["param"], gb["param"]), ga["name"]
        assert torch.equal(sa["state"][i]["exp_avg"], sb["state"][i]["exp_avg"]), ga["name"]
        assert torch.equal(sa["state"][i]["exp_avg_sq"], sb["state"][i]["exp_avg_sq"]), ga["name"]
        assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"]) == 3.0
