"""CPU: host-side logic of the product path (no kernels run): model construction / state_dict schema / freeze policy,
optimizer param grouping + poly LR, concept maps, synthetic inputs, loud failure without a GPU."""
import math

import numpy as np
import pytest
import torch

from semivl_amd.synthetic import exp40_cfg, synthetic_batch


def test_build_model_schema_and_freeze_policy():
    from semivl_amd.model.builder import build_model
    m = build_model(exp40_cfg())
    sd = m.state_dict()
    assert len(sd) == 372
    total = sum(p.numel() for p in m.parameters())
    assert abs(total / 1e6 - 175.87) < 0.01
    bb_tr = [n for n, p in m.backbone.named_parameters() if p.requires_grad]
    assert len(bb_tr) == 49 and all(("attn" in n) or ("pos_embed" in n) for n in bb_tr)   # vlm.py:66-67,80-88
    assert sum(p.numel() for n, p in m.backbone.named_parameters() if p.requires_grad) == 29_135_616
    assert sum(p.numel() for p in m.decode_head.parameters()) == 2_217_185
    for k in ("backbone.cls_token", "backbone.pos_embed", "backbone.patch_embed.projection.weight", "backbone.ln0.weight",
              "backbone.proj.weight", "backbone.layers.11.attn.attn.in_proj_weight",
              "backbone.layers.0.ffn.layers.0.0.weight", "backbone.layers.0.ffn.layers.1.bias",
              "clip_encoder.layers.3.ln2.bias", "decode_head.conv1.weight", "decode_head.aspp.aspp_convs.4.gap.1.weight",
              "decode_head.aspp.project.0.weight", "decode_head.layers.1.transformer.attn.attn.out_proj.weight",
              "decode_head.text_proj.0.weight", "decode_head.skip_proj.1.0.weight", "decode_head.up1.up.weight",
              "decode_head.up2.conv.4.bias", "decode_head.head.weight"):
        assert k in sd, k
    assert sd["backbone.pos_embed"].shape == (1, 1025, 768) and sd["backbone.proj.weight"].shape == (512, 768, 1, 1)
    assert sd["decode_head.up1.up.weight"].shape == (128, 96, 2, 2) and sd["decode_head.conv1.weight"].shape == (128, 1, 7, 7)
    assert m.num_classes == 21 and m.align_corners is False and m.fp_rate == 0.5
    assert tuple(m.loaded_mcc_text_feat.shape) == (98, 512)


def test_unsupported_configs_fail_loudly():
    from semivl_amd.model.builder import build_model
    cfg = exp40_cfg()
    cfg["model"] = "mmseg.vlm-dlv3p-bn11-sk4-ft-tvit-in1k"      # UniMatch-with-ViT ablation (experiments.py:433)
    with pytest.raises((ValueError, NotImplementedError)):
        build_model(cfg)
    cfg["model"] = "deeplabv3plus"
    with pytest.raises(ValueError):
        build_model(cfg)


def test_skr04_config_builds_with_reference_key_schema():
    """Cityscapes recipe (experiments.py:428-456): ViT out_indices [4, 12], ResNetV1c side encoder, (768, 256) skips."""
    from semivl_amd.model.builder import build_model
    cfg = exp40_cfg(batch_size=1, crop=64, nclass=19, dataset="cityscapes")
    cfg["model"] = "mmseg.vlm-vlg-aspp-s2p4-skr04-ftap-mcvitb"
    cfg["model_args"] = dict(maskclip_class_filter=None, renorm_clip_img=True)
    m = build_model(cfg)
    keys = list(m.state_dict())
    assert "conv_encoder.stem.0.weight" in keys and "conv_encoder.layer1.0.downsample.1.running_var" in keys
    assert "conv_encoder.layer1.2.bn3.num_batches_tracked" in keys
    assert tuple(m.decode_head.skip_proj[1][0].weight.shape) == (32, 256, 3, 3)
    assert m.renorm_clip_img and m.backbone.out_indices == [4, 12]
    assert all(p.requires_grad for p in m.conv_encoder.parameters())


def test_no_cpu_fallback():
    """The product path refuses to run without the GPU/extension instead of silently computing on the CPU."""
    from semivl_amd.model.builder import build_model
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = build_model(exp40_cfg(crop=64))
    with pytest.raises((AssertionError, RuntimeError)):
        m(torch.randn(1, 3, 64, 64))


def test_param_groups_and_poly_lr():
    from semivl_amd.train import mmcv_param_groups
    ck = exp40_cfg()["optimizer"]["paramwise_cfg"]["custom_keys"]
    names = ["backbone.pos_embed", "backbone.layers.0.attn.attn.in_proj_weight", "decode_head.conv1.weight",
             "decode_head.layers.0.transformer.ln1.weight", "decode_head.aspp.aspp_convs.0.1.bias", "conv_encoder.x",
             "something.norm.weight", "plain.weight"]
    g = {d["name"]: d for d in mmcv_param_groups([(n, None) for n in names], 1e-4, 0.01, ck)}
    assert math.isclose(g["backbone.pos_embed"]["lr"], 1e-6) and g["backbone.pos_embed"]["weight_decay"] == 0.01
    assert math.isclose(g["decode_head.conv1.weight"]["lr"], 1e-3)
    # 'head' outranks 'ln' (longest key first): decoder LN gets lr x10 AND keeps weight decay
    assert math.isclose(g["decode_head.layers.0.transformer.ln1.weight"]["lr"], 1e-3)
    assert g["decode_head.layers.0.transformer.ln1.weight"]["weight_decay"] == 0.01
    assert math.isclose(g["conv_encoder.x"]["lr"], 1e-4)
    assert g["something.norm.weight"]["weight_decay"] == 0.0 and g["plain.weight"]["lr"] == 1e-4
    from oracle import semivl_oracle as O
    assert math.isclose(O.poly_lr(1e-3, 10, 100), 1e-3 * 0.9 ** 0.9)


def test_concept_maps():
    from semivl_amd.model.text_embeddings import concept_offsets, get_class_to_concept_idxs
    voc = get_class_to_concept_idxs("configs/_base_/datasets/text_embedding/voc12_wbg_concept4_single.npy")
    assert len(voc) == 21 and len(voc[0]) == 45 and voc[20][-1] == 97
    cs = get_class_to_concept_idxs("x/cityscapes_concept3_single.npy")
    assert len(cs) == 19 and sum(len(v) for v in cs.values()) == 54
    assert concept_offsets(voc, "cpu").tolist()[-1] == 98
    with pytest.raises(ValueError):
        get_class_to_concept_idxs("voc12_wbg_single.npy")


def test_synthetic_batch_spec():
    b = synthetic_batch(4, 64, 21, seed=1)
    assert set(b) == {"img_x", "mask_x", "img_w", "img_s1", "img_s2", "ignore_mask", "mix1", "mix2", "img_w_other",
                      "img_s1_other", "img_s2_other", "ignore_mask_other"}
    assert b["mask_x"].dtype == torch.int64 and set(b["mask_x"].unique().tolist()) <= set(range(21)) | {255}
    assert (b["ignore_mask"][1, -8:] == 255).all() and (b["ignore_mask"][0] == 0).all()
    assert set(b["mix1"].unique().tolist()) <= {0.0, 1.0}
    from oracle import semivl_oracle as O
    o = O.synthetic_batch(4, 64, 21, seed=1)
    assert all(torch.equal(b[k], o[k]) for k in b)   # product and oracle feed the same synthetic stream


def test_fresh_model_has_no_uninitialised_parameters():
    """Every parameter of a freshly built model (no pretrained file, no fixture state) must come from an initialiser:
    finite and O(1) (the decoder's attention in-projection once was torch.empty)."""
    import torch
    from golden_util import build_hip, load_fixture
    _, c = load_fixture("tiny")
    for _ in range(3):   # allocator reuse makes garbage show up on later builds
        m = build_hip(c)
        junk = torch.empty(1 << 20).fill_(float("nan"))
        del junk
        for n, p in m.named_parameters():
            assert torch.isfinite(p).all() and p.abs().max() < 50, n


def test_reference_experiment_cfgs_drop_in():
    """`build_model` accepts, unchanged, the flat dicts the reference's own experiments.generate_experiment_cfgs(40|42|43|44)
    emits (tests/golden/experiment_cfgs.json, dumped by tests/golden/gen_golden_cfgs.py), and the package's built-in model
    hyper-parameters equal the evaluated contents of the reference's configs/_base_/models/*.py."""
    import json
    import os
    from semivl_amd.model.builder import build_model, builtin_model_cfg
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "experiment_cfgs.json")))

    def norm(o):   # JSON turned tuples into lists
        if isinstance(o, dict):
            return {k: norm(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [norm(v) for v in o]
        return o
    for name, ref in d["model_cfgs"].items():
        mine = norm(builtin_model_cfg(name))     # (the reference files may hold extra module-level helpers, e.g. norm_cfg)
        assert mine == {k: norm(ref[k]) for k in mine}, name
    want = {"40": (21, 512, False), "42": (81, 512, False), "43": (150, 512, False), "44": (19, 801, True)}
    for e, cfg in d["experiments"].items():
        nclass, crop, conv = want[e]
        assert (cfg["nclass"], cfg["crop_size"]) == (nclass, crop)
        # a configured-but-missing pretrained file must fail loudly (mmcv load_checkpoint behaviour) ...
        with pytest.raises(FileNotFoundError):
            build_model(cfg)
        # ... and builds with the explicit synthetic-weight switch
        with torch.device("meta"):
            m = build_model(dict(cfg, allow_random_init=True))
        assert m.num_classes == nclass and m.decode_head.image_size == crop and (m.conv_encoder is not None) == conv
        assert m.backbone.pos_embed.shape[1] == (crop // 16) ** 2 + 1
        assert m.clip_encoder.pos_embed.shape[1] == 32 * 32 + 1          # mcc_fix_resize_pos unset: frozen CLIP keeps 512^2
        assert m._text_feat.shape[0] == nclass
        trainable = [n for n, p in m.backbone.named_parameters() if p.requires_grad]
        assert len(trainable) == 49 and all(("attn" in n or "pos_embed" in n) for n in trainable)


def test_warmup_then_poly_lr():
    from semivl_amd.train import FusedAdamW
    class _O(FusedAdamW):
        def __init__(self):
            self.groups = [dict(initial_lr=1e-3, lr=1e-3), dict(initial_lr=1e-5, lr=1e-5)]
            self._lr_host, self.seg_lr, self._lr_evt = torch.zeros(2), torch.zeros(2), None
    o = _O()
    o.poly_lr(5, 100, warmup_iters=10, warmup_ratio=1e-6)        # semivl.py:339-342
    k = (1 - 5 / 10) * (1 - 1e-6)
    assert math.isclose(o.groups[0]["lr"], 1e-3 * (1 - k)) and math.isclose(o.groups[1]["lr"], 1e-5 * (1 - k))
    o.poly_lr(50, 100, warmup_iters=10)                          # :343-345
    assert math.isclose(o.groups[0]["lr"], 1e-3 * 0.5 ** 0.9)
    assert torch.allclose(o.seg_lr, torch.tensor([g["lr"] for g in o.groups]))


def test_head_chunk_plan():
    """Sample chunks of the class-batched decoder (memory plan): live ranges announced by the step are kept apart from the
    dead samples, chunks never exceed the class-image budget, and the plan tiles the batch exactly."""
    from types import SimpleNamespace
    from semivl_amd.model.vlg_head import _chunk_plan
    m = SimpleNamespace(chunk_class_images=1344, _bwd_ranges={48: [(16, 48)]})
    assert _chunk_plan(m, 48, 21) == [(0, 16, False), (16, 48, True)]            # VOC B=16: exactly the round-1 launches
    plan = _chunk_plan(m, 48, 150)                                               # ADE: 8 samples (1200 class-images) per chunk
    assert plan == [(0, 8, False), (8, 16, False), (16, 24, True), (24, 32, True), (32, 40, True), (40, 48, True)]
    assert _chunk_plan(m, 32, 150) == [(0, 8, True), (8, 16, True), (16, 24, True), (24, 32, True)]   # no entry: all live
    m2 = SimpleNamespace(chunk_class_images=100, _bwd_ranges={10: [(2, 5), (7, 10)]})
    plan = _chunk_plan(m2, 10, 81)                                               # budget below one sample: one sample per chunk
    assert [c[:2] for c in plan] == [(i, i + 1) for i in range(10)]
    assert [c[2] for c in plan] == [False, False, True, True, True, False, False, True, True, True]
    m3 = SimpleNamespace(chunk_class_images=1344, _bwd_ranges=None)
    plan = _chunk_plan(m3, 7, 300)                                               # 4 samples fit: balanced 4 + 3, not 4 + 4 - 1
    assert plan == [(0, 4, True), (4, 7, True)]
