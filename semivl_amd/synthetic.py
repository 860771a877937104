"""Synthetic step inputs for benchmarking: same tensor set / dtypes / value ranges as one iteration of
semivl.py:205-221 delivers (ImageNet-normalised crops ~ N(0,1), block-constant labels with 5 % ignore, pad-crop ignore
strips on odd samples, CutMix boxes per third_party/unimatch/dataset/transform.py:66-84)."""
import math

import torch


def synthetic_batch(B, S, nclass, seed=1234, device="cpu"):
    g = torch.Generator().manual_seed(seed)

    def img():
        return torch.randn(B, 3, S, S, generator=g)

    blk = max(S // 16, 1)
    nb = (S + blk - 1) // blk
    lab = torch.randint(0, nclass, (B, nb, nb), generator=g)
    lab[torch.rand(B, nb, nb, generator=g) < 0.05] = 255
    mask_x = lab.repeat_interleave(blk, 1).repeat_interleave(blk, 2)[:, :S, :S].contiguous()

    def ign():
        m = torch.zeros(B, S, S, dtype=torch.int64)
        m[1::2, S - S // 8:] = 255
        return m

    def box():
        m = torch.zeros(B, S, S)
        for i in range(B):
            if torch.rand(1, generator=g).item() < 0.5:
                area = (0.02 + 0.38 * torch.rand(1, generator=g).item()) * S * S
                ratio = 0.3 + (1 / 0.3 - 0.3) * torch.rand(1, generator=g).item()
                w, h = min(int(math.sqrt(area / ratio)), S), min(int(math.sqrt(area * ratio)), S)
                x = int(torch.randint(0, S - w + 1, (1,), generator=g).item())
                y = int(torch.randint(0, S - h + 1, (1,), generator=g).item())
                m[i, y:y + h, x:x + w] = 1
        return m

    b = dict(img_x=img(), mask_x=mask_x, img_w=img(), img_s1=img(), img_s2=img(), ignore_mask=ign(), mix1=box(),
             mix2=box(), img_w_other=img(), img_s1_other=img(), img_s2_other=img(), ignore_mask_other=ign())
    return {k: v.to(device) for k, v in b.items()}


def exp40_cfg(batch_size=16, crop=512, nclass=21, dataset="pascal"):
    """The flat experiment dict of experiments.py exp 40 restricted to the keys the hot path reads; for
    dataset='cityscapes' the exp 44 recipe (experiments.py:428-456): skr04 model (ResNetV1c conv_encoder), CLIP
    re-normalisation, concept-averaged text, pixelavg, lr 5e-5, backbone / conv_encoder lr_mult 0.1."""
    cs = dataset == "cityscapes"
    tv = "conceptavg3_single" if cs else "single"
    margs = dict(maskclip_class_filter=None)
    if cs:
        margs["renorm_clip_img"] = True
    return dict(
        dataset=dataset, nclass=nclass, crop_size=crop,
        model="mmseg.vlm-vlg-aspp-s2p4-%s-ftap-mcvitb" % ("skr04" if cs else "sk04"),
        model_args=margs, text_embedding_variant=tv,
        mcc_text="concept4_single" if dataset == "pascal" else ("concept3_single" if cs else "single"), pl_text=tv,
        method="semivl", use_fp=True, conf_mode="pixelavg" if cs else "pixelwise", conf_thresh=0.95, disable_dropout=True, pleval=True, fp_rate=0.5,
        maskclip_consistency_lambda=[0.1, 0], clip_encoder="mcvit16", mcc_conf_thresh=0.9, mcc_loss_reduce="mean_all",
        criterion=dict(name="CELoss", kwargs=dict(ignore_index=255)), criterion_u="CELoss",
        optimizer=dict(type="AdamW", lr=5e-5 if cs else 1e-4, weight_decay=0.01, paramwise_cfg=dict(custom_keys=dict(
            backbone=dict(lr_mult=0.1 if cs else 0.01), text_encoder=dict(lr_mult=0.0),
            conv_encoder=dict(lr_mult=0.1 if cs else 1.0), norm=dict(decay_mult=0.0), ln=dict(decay_mult=0.0),
            head=dict(lr_mult=10.0)))),
        warmup_iters=0, warmup_ratio=1e-6, batch_size=batch_size, epochs=80,
        allow_random_init=True)   # synthetic weights: pretrained/clip2mmseg_ViT16_clip_backbone.pth is not in the container
