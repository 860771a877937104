"""ORACLE — test infrastructure only.  CPU restatement of the reference's evaluation path:
`predict(..., mode='zegclip_sliding_window')` of third_party/unimatch/supervised.py:70-102 and `intersectionAndUnion`
of third_party/unimatch/util/utils.py:91-103.  Pinned against the reference's own functions by
tests/golden/gen_golden_eval.py (fixture tests/golden/eval_zegclip.npz)."""
import numpy as np
import torch
import torch.nn.functional as F


def predict_zegclip_sliding_window(model, img, mask_hw, crop, stride, nclass):
    b, _, h, w = img.shape
    hg = max(h - crop + stride - 1, 0) // stride + 1
    wg = max(w - crop + stride - 1, 0) // stride + 1
    preds = img.new_zeros((b, nclass, h, w))
    count = img.new_zeros((b, 1, h, w))
    for hi in range(hg):
        for wi in range(wg):
            y1, x1 = hi * stride, wi * stride
            y2, x2 = min(y1 + crop, h), min(x1 + crop, w)
            y1, x1 = max(y2 - crop, 0), max(x2 - crop, 0)
            logit = model(img[:, :, y1:y2, x1:x2])
            preds += F.pad(logit, (int(x1), int(w - x2), int(y1), int(h - y2)))
            count[:, :, y1:y2, x1:x2] += 1
    assert (count == 0).sum() == 0
    preds = preds / count
    final = F.interpolate(preds, size=tuple(mask_hw), mode="bilinear", align_corners=True)
    return final.argmax(dim=1), final


def predict_sliding_window(model, img, crop, nclass):
    """supervised.py:104-117: probability accumulation over windows with stride int(2/3 crop)."""
    b, _, h, w = img.shape
    final = torch.zeros(b, nclass, h, w)
    step = int(crop * 2 / 3)
    row = 0
    while row < h:
        col = 0
        while col < w:
            pred = model(img[:, :, row:min(h, row + crop), col:min(w, col + crop)])
            final[:, :, row:min(h, row + crop), col:min(w, col + crop)] += pred.softmax(dim=1)
            col += step
        row += step
    return final.argmax(dim=1), final


def predict_padded_sliding_window(model, img, crop, stride, nclass):
    """supervised.py:41-68: zero-padded crops."""
    if stride < 1:
        stride = int(crop * stride)
    b, _, h, w = img.shape
    final = torch.zeros(b, nclass, h, w)
    row = 0
    while row < h:
        col = 0
        while col < w:
            y2, x2 = min(h, row + crop), min(w, col + crop)
            ch, cw = y2 - row, x2 - col
            cropped = torch.zeros((b, 3, crop, crop))
            cropped[:, :, :ch, :cw] = img[:, :, row:y2, col:x2]
            pred = model(cropped)
            final[:, :, row:y2, col:x2] += pred.softmax(dim=1)[:, :, :ch, :cw]
            col += stride
        row += stride
    return final.argmax(dim=1), final


def intersection_and_union(output, target, K, ignore_index=255):
    output = np.asarray(output).reshape(-1).copy()
    target = np.asarray(target).reshape(-1)
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    ai, _ = np.histogram(inter, bins=np.arange(K + 1))
    ao, _ = np.histogram(output, bins=np.arange(K + 1))
    at, _ = np.histogram(target, bins=np.arange(K + 1))
    return ai, ao + at - ai, at


def miou(inter_sum, union_sum):
    iou = inter_sum / (union_sum + 1e-10) * 100.0
    return float(np.mean(iou)), iou


class ToyModel(torch.nn.Module):
    """Deterministic stand-in segmentor for the eval fixtures: logits = 1x1 conv of the (smoothed) image."""

    def __init__(self, nclass, seed=5):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = torch.nn.Parameter(torch.randn(nclass, 3, generator=g), requires_grad=False)

    def forward(self, x):
        return torch.einsum("nk,bkhw->bnhw", self.w, x)
