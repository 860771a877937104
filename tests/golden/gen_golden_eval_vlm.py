"""Golden vectors for the evaluation path THROUGH THE REAL MODEL: the reference's own `predict`
(third_party/unimatch/supervised.py:40-133) driving the reference's own VLM (model/vlm.py + forward_wrapper,
model/builder.py:56-102) on an image that is neither square nor a multiple of the crop -- so the model sees windows
smaller than / differently shaped from `img_size` (non-square token grids, per-forward pos-embed resize, the two chained
output resizes of vlg_head.py:247 + builder.py:93-97).  Build container only (needs /root/reference).
Writes tests/golden/eval_vlm.npz; also checks the oracle restatement against the reference."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

CFG = dict(S=128, B=1, embed=64, layers=3, heads=1, out_indices=[0, 1, 3], channels=32, text_channels=32, dec_heads=1,
           up=(32, 16), skip=(16, 16), seed=21, logit_gain=150.0, H=160, W=150, stride=85)


def eval_image(c):
    g = torch.Generator().manual_seed(c["seed"])
    img = torch.randn(2, 3, c["H"], c["W"], generator=g)
    img = torch.nn.functional.avg_pool2d(img, 9, stride=1, padding=4)
    img = img / img.std()
    mask = torch.randint(0, 21, (2, c["H"] // 8, c["W"] // 8 + 1), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
    mask = mask[:, :c["H"], :c["W"]].contiguous()
    mask[torch.rand(mask.shape, generator=g) < 0.03] = 255
    return img, mask


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    import _ref_shim
    _ref_shim.install()
    import gen_golden as G
    from golden_util import build_oracle, seeded_state
    from oracle import eval_oracle as E
    from third_party.unimatch.supervised import predict as ref_predict
    from third_party.unimatch.util.utils import intersectionAndUnion as ref_iau
    c = CFG
    ref = G.build_reference(c)
    sd = seeded_state([(k, tuple(v.shape)) for k, v in ref.state_dict().items()], c["seed"], c["logit_gain"])
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    orc = build_oracle(c)
    orc.load_state_dict(sd, strict=True)
    orc.eval()
    img, mask = eval_image(c)
    cfg = dict(crop_size=c["S"], stride=c["stride"], nclass=21)
    torch.Tensor.cuda = lambda self, *a, **k: self      # supervised.py calls .cuda() on fresh tensors
    out = dict(cfg=np.array(repr(c)), img_checksum=np.array([img.double().sum().item(), img.double().abs().sum().item()]),
               mask=mask.numpy().astype(np.uint8),
               w_checksum=np.array([sum(v.double().sum().item() for v in sd.values()),
                                    sum(v.double().abs().sum().item() for v in sd.values())]))
    with torch.no_grad():
        for mode in ("sliding_window", "zegclip_sliding_window", "original"):
            pred, final = ref_predict(ref, img, mask, mode, cfg, return_logits=True)
            if mode == "sliding_window":
                opred, ofinal = E.predict_sliding_window(orc, img, c["S"], 21)
            elif mode == "zegclip_sliding_window":
                opred, ofinal = E.predict_zegclip_sliding_window(orc, img, mask.shape[-2:], c["S"], c["stride"], 21)
            else:
                ofinal = orc(img)
                opred = ofinal.argmax(dim=1)
            d = (final - ofinal).abs().max().item()
            assert torch.equal(pred, opred) and d < 1e-4, (mode, d)
            t2 = final.topk(2, dim=1).values
            gap = t2[:, 0] - t2[:, 1]
            tie = gap < 1e-6
            inter, union, target = ref_iau(pred.numpy(), mask.numpy(), 21, 255)
            print(f"[{mode}] reference == oracle (pred maps identical, logits |d| {d:.1e}); labels used "
                  f"{pred.unique().numel()}, fp ties {int(tie.sum())}, mIoU {E.miou(inter.astype(float), union.astype(float))[0]:.4f}")
            out[f"pred/{mode}"] = pred.numpy().astype(np.uint8)
            out[f"final_s4/{mode}"] = final[:, :, ::4, ::4].numpy()
            out[f"gap/{mode}"] = gap.numpy().astype(np.float32)    # top-2 gap of the reference's own decision, per pixel
            out[f"inter/{mode}"], out[f"union/{mode}"], out[f"target/{mode}"] = inter, union, target
    path = os.path.join(HERE, "eval_vlm.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
