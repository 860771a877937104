"""Golden vectors for the evaluation path, captured from the reference's OWN `predict` / `intersectionAndUnion`
(build container only; needs /root/reference).  Writes tests/golden/eval_zegclip.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    os.chdir(REF)
    sys.path.insert(0, REF)
    import _ref_shim
    _ref_shim.install()
    from third_party.unimatch.supervised import predict as ref_predict
    from third_party.unimatch.util.utils import intersectionAndUnion as ref_iau
    from oracle import eval_oracle as E
    K, crop, stride = 21, 512, 426
    cfg = dict(crop_size=crop, stride=stride, nclass=K)
    g = torch.Generator().manual_seed(77)
    img = torch.randn(2, 3, 600, 700, generator=g)
    # smooth the image a little so that argmax regions are not pure noise
    img = torch.nn.functional.avg_pool2d(img, 9, stride=1, padding=4)
    mask = torch.randint(0, K, (2, 75, 88), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :590, :690]
    mask[torch.rand(mask.shape, generator=g) < 0.03] = 255
    model = E.ToyModel(K)
    with torch.no_grad():
        pred, final = ref_predict(model, img, mask, "zegclip_sliding_window", cfg, return_logits=True)
        opred, ofinal = E.predict_zegclip_sliding_window(model, img, mask.shape[-2:], crop, stride, K)
    assert torch.equal(pred, opred) and torch.equal(final, ofinal), "oracle restatement differs from the reference"
    # the probability-averaging modes call .cuda() on fresh tensors: run them on the CPU by neutralising it
    torch.Tensor.cuda = lambda self, *a, **k: self
    sw_mask = mask[:, :, :] if False else torch.zeros(2, 600, 700, dtype=torch.long)
    with torch.no_grad():
        p_sw, f_sw = ref_predict(model, img, sw_mask, "sliding_window", cfg, return_logits=True)
        p_pd, f_pd = ref_predict(model, img, sw_mask, "padded_sliding_window", cfg, return_logits=True)
        o_sw = E.predict_sliding_window(model, img, crop, K)
        o_pd = E.predict_padded_sliding_window(model, img, crop, stride, K)
    assert torch.equal(p_sw, o_sw[0]) and torch.equal(f_sw, o_sw[1]) and torch.equal(p_pd, o_pd[0]) and torch.equal(f_pd, o_pd[1])
    print("sliding_window / padded_sliding_window: reference == oracle")
    ri = ref_iau(pred.numpy(), mask.numpy(), K, 255)
    oi = E.intersection_and_union(pred.numpy(), mask.numpy(), K, 255)
    assert all(np.array_equal(a, b) for a, b in zip(ri, oi))
    print("reference == oracle: pred maps identical, counts identical; mIoU", E.miou(ri[0].astype(float), ri[1].astype(float))[0])
    np.savez_compressed(os.path.join(HERE, "eval_zegclip.npz"), img_checksum=np.array([img.double().sum().item(), img.double().abs().sum().item()]),
                        mask=mask.numpy().astype(np.uint8), pred=pred.numpy().astype(np.uint8),
                        final_s8=final[:, :, ::8, ::8].numpy(), inter=ri[0], union=ri[1], target=ri[2],
                        pred_sw=p_sw.numpy().astype(np.uint8), final_sw_s8=f_sw[:, :, ::8, ::8].numpy(),
                        pred_pd=p_pd.numpy().astype(np.uint8), final_pd_s8=f_pd[:, :, ::8, ::8].numpy(),
                        cfg=np.array([K, crop, stride]))
    print("wrote eval_zegclip.npz", os.path.getsize(os.path.join(HERE, "eval_zegclip.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
