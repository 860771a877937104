"""Checkpoint I/O: the CLIP weight converter against golden vectors produced by the reference's own
converter script (tests/golden/gen_golden_ckpt.py), the pretrained-file loading path of the backbone (pos-embed resize
included) and the {'model','optimizer','epoch'} checkpoint round trip with the reference's eval.py loading rules."""
import os

import numpy as np
import pytest
import torch

from golden_util import build_hip, build_oracle, load_fixture

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
    # the reference's layout (semivl.py:428 + mmcv's constructor): one group per model parameter, frozen ones included
    nall = len(list(ma.named_parameters()))
    assert len(sd["param_groups"]) == nall > len(oa.groups) and len(sd["state"]) == len(oa.groups)
    slot = {ai: j for j, (_, ai) in enumerate(oa.all_params) if ai is not None}
    assert sd["state"][slot[0]]["exp_avg"].shape == oa.groups[0]["param"].shape and float(sd["state"][slot[0]]["step"]) == 2.0
    assert [g_["params"] for g_ in sd["param_groups"]] == [[j] for j in range(nall)]
    mb.load_state_dict(ma.state_dict())
    ob.p.copy_(oa.p)
    ob.load_state_dict(sd)
    for o in (oa, ob):
        o.g.copy_(grads[2]); o.step()
    sa, sb = oa.state_dict(), ob.state_dict()          # per-tensor views (the arena's alignment padding is not state)
    for i, (ga, gb) in enumerate(zip(oa.groups, ob.groups)):
        j = slot[i]
        assert torch.equal(ga["param"], gb["param"]), ga["name"]
        assert torch.equal(sa["state"][j]["exp_avg"], sb["state"][j]["exp_avg"]), ga["name"]
        assert torch.equal(sa["state"][j]["exp_avg_sq"], sb["state"][j]["exp_avg_sq"]), ga["name"]
        assert float(sa["state"][j]["step"]) == float(sb["state"][j]["step"]) == 3.0


@pytest.mark.gpu
def test_optimizer_state_dict_is_index_compatible_with_the_reference(dev):
    """The 'optimizer' entry of a checkpoint interchanges with the reference's (semivl.py:426-433): a torch.optim.AdamW over
    the oracle's mmcv-style param groups (one per named parameter) and the fused optimizer, stepped on the same gradients,
    load each other's state_dict and land on the same moments / step / learning rates."""
    from oracle import semivl_oracle as O
    from semivl_amd.train import FusedAdamW
    z, c = load_fixture("tiny")
    ck = dict(backbone=dict(lr_mult=0.01), head=dict(lr_mult=10.0))
    ocfg = dict(type="AdamW", lr=1e-3, weight_decay=0.01, paramwise_cfg=dict(custom_keys=ck))
    orc = build_oracle(c)
    hip = build_hip(c)
    hip.load_state_dict(orc.state_dict(), strict=True)
    hip.to(dev)
    fo = FusedAdamW(hip, ocfg)
    to = torch.optim.AdamW(O.param_groups(orc, 1e-3, 0.01, ck), lr=1e-3, weight_decay=0.01)
    names = [n for n, _ in orc.named_parameters()]
    assert names == [n for n, _ in fo.all_params]
    trainable = {g_["name"] for g_ in fo.groups}
    gen = torch.Generator().manual_seed(5)
    for _ in range(2):
        for (n, p) in orc.named_parameters():
            p.grad = torch.randn(p.shape, generator=gen) * 0.1 if n in trainable else None
        for g_ in fo.groups:
            g_["param"].main_grad.copy_(dict(orc.named_parameters())[g_["name"]].grad)
        to.step(); fo.step()
    rs, fs = to.state_dict(), fo.state_dict()
    assert sorted(rs["state"]) == sorted(fs["state"]) and len(rs["param_groups"]) == len(fs["param_groups"])
    for j in rs["state"]:
        assert float(rs["state"][j]["step"]) == float(fs["state"][j]["step"]) == 2.0
        for k in ("exp_avg", "exp_avg_sq"):
            a, b_ = rs["state"][j][k], fs["state"][j][k]
            assert a.shape == b_.shape and (a - b_).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item()), (names[j], k)
    for j, (gr, gf) in enumerate(zip(rs["param_groups"], fs["param_groups"])):
        assert gr["params"] == gf["params"] == [j]
        if names[j] in trainable:
            assert abs(gr["lr"] - gf["lr"]) < 1e-12 and abs(gr["weight_decay"] - gf["weight_decay"]) < 1e-12, names[j]
    # reference -> fused: moments land in the arena; fused -> reference: torch accepts the dict as it is
    f2 = FusedAdamW(build_hip(c).to(dev), ocfg)
    f2.load_state_dict({k: rs[k] for k in ("state", "param_groups")})
    assert f2.step_count == 2
    for j, (n, ai) in enumerate(f2.all_params):
        if ai is not None:
            off, cnt = int(f2.seg_off[ai]), f2.groups[ai]["param"].numel()
            assert torch.equal(f2.m[off:off + cnt].cpu().view(rs["state"][j]["exp_avg"].shape), rs["state"][j]["exp_avg"]), n
    t2 = torch.optim.AdamW(O.param_groups(build_oracle(c), 1e-3, 0.01, ck), lr=1e-3, weight_decay=0.01)
    t2.load_state_dict({k: fs[k] for k in ("state", "param_groups")})
    assert torch.equal(t2.state_dict()["state"][min(fs["state"])]["exp_avg"], fs["state"][min(fs["state"])]["exp_avg"])
    # EVERY group (frozen tensors and clip_encoder.* included) carries `initial_lr` and a scheduled `lr`, so the reference's
    # own loop (semivl.py:124-125 setdefault, :341-345 re-schedule of every group) runs on the loaded dict as it is
    fo.poly_lr(3, 50)
    fs2 = fo.state_dict()
    f_ = (1 - 3 / 50) ** 0.9
    for j, gf in enumerate(fs2["param_groups"]):
        assert set(gf) >= {"lr", "initial_lr", "weight_decay", "betas", "eps", "amsgrad", "params"}, (names[j], sorted(gf))
        assert abs(gf["lr"] - gf["initial_lr"] * f_) < 1e-15, names[j]
    t2.load_state_dict({k: fs2[k] for k in ("state", "param_groups")})
    for group in t2.param_groups:                               # the reference's poly schedule, verbatim in spirit
        group["lr"] = group["initial_lr"] * (1 - 4 / 50) ** 0.9
    f3 = FusedAdamW(build_hip(c).to(dev), ocfg)
    f3.load_state_dict({k: fs2[k] for k in ("state", "param_groups")})
    assert abs(f3._lr_factor - f_) < 1e-12 and f3.state_dict()["param_groups"][0]["lr"] == fs2["param_groups"][0]["lr"]
