"""Evaluation path: sliding-window inference + mIoU, mirroring
third_party/unimatch/supervised.py:40-164 (`predict`, `evaluate`) and util/utils.py:91-103 (`intersectionAndUnion`).

Same `predict(model, img, mask, mode, cfg)` signature and modes.  Window logits are accumulated on the GPU with the
library's strided copy kernel, the confusion counts are integer histograms on the device, and the per-image
3 x all_reduce of the reference is replaced by ONE all_reduce of the summed int64 counts (integer sums commute, so the
result is identical).
"""
import numpy as np
import torch
import torch.distributed as dist

from . import ops


def _crop(img, y1, y2, x1, x2):
    """img[:, :, y1:y2, x1:x2] as a contiguous tensor (strided row copy)."""
    b, c, h, w = img.shape
    ch, cw = y2 - y1, x2 - x1
    out = ops.empty(b, c, ch, cw, device=img.device)
    ops.copy2d(img, y1 * w + x1, ch, h * w, w, out, 0, ch, ch * cw, cw, b * c * ch, cw)
    return out


def _window_add(canvas, win, y1, x1):
    """canvas[:, :, y1:y1+ch, x1:x1+cw] += win."""
    b, n, h, w = canvas.shape
    ch, cw = win.shape[2:]
    ops.copy2d(win.contiguous(), 0, ch, ch * cw, cw, canvas, y1 * w + x1, ch, h * w, w, b * n * ch, cw, accumulate=True)


def predict(model, img, mask, mode, cfg, return_logits=False):
    img = img.contiguous()
    if mode == "zegclip_sliding_window":  # supervised.py:70-102
        hs = ws = cfg["stride"]
        hc = wc = cfg["crop_size"]
        b, _, h, w = img.shape
        K = cfg["nclass"]
        hg = max(h - hc + hs - 1, 0) // hs + 1
        wg = max(w - wc + ws - 1, 0) // ws + 1
        preds = ops.zeros(b, K, h, w, device=img.device)
        count = ops.zeros(b, 1, h, w, device=img.device)
        for hi in range(hg):
            for wi in range(wg):
                y1, x1 = hi * hs, wi * ws
                y2, x2 = min(y1 + hc, h), min(x1 + wc, w)
                y1, x1 = max(y2 - hc, 0), max(x2 - wc, 0)
                logit = model(_crop(img, y1, y2, x1, x2))
                _window_add(preds, logit, y1, x1)
                ones = ops.fill(ops.empty(b, 1, y2 - y1, x2 - x1, device=img.device), 1.0)
                _window_add(count, ones, y1, x1)
        cnt = count.expand(b, K, h, w).contiguous() if K > 1 else count
        preds = ops.eltwise(7, preds.view(-1), cnt.view(-1)).view(b, K, h, w)
        H, W = mask.shape[-2:]
        final = ops.bilinear_planes_fwd(preds, h, w, True, H, W) if (H, W) != (h, w) else preds
    elif mode == "sliding_window":  # supervised.py:104-117: softmax-probability accumulation, stride int(2/3 crop)
        grid = cfg["crop_size"]
        b, _, h, w = img.shape
        final = ops.zeros(b, cfg["nclass"], h, w, device=img.device)
        step = int(grid * 2 / 3)
        row = 0
        while row < h:
            col = 0
            while col < w:
                y2, x2 = min(h, row + grid), min(w, col + grid)
                logit = model(_crop(img, row, y2, col, x2))
                _window_add(final, ops.softmax_planes(logit.contiguous()), row, col)
                col += step
            row += step
    elif mode == "padded_sliding_window":  # supervised.py:41-68: zero-padded crops, configurable stride
        grid, stride = cfg["crop_size"], cfg["stride"]
        if stride < 1:
            stride = int(grid * stride)
        b, c, h, w = img.shape
        final = ops.zeros(b, cfg["nclass"], h, w, device=img.device)
        row = 0
        while row < h:
            col = 0
            while col < w:
                y2, x2 = min(h, row + grid), min(w, col + grid)
                ch, cw = y2 - row, x2 - col
                padded = ops.zeros(b, c, grid, grid, device=img.device)
                ops.copy2d(img, row * w + col, ch, h * w, w, padded, 0, ch, grid * grid, grid, b * c * ch, cw)
                prob = ops.softmax_planes(model(padded).contiguous())
                _window_add(final, _crop(prob, 0, ch, 0, cw), row, col)
                col += stride
            row += stride
    elif mode in ("original", "center_crop"):
        if mode == "center_crop":
            h, w = img.shape[-2:]
            sh, sw = (h - cfg["crop_size"]) // 2, (w - cfg["crop_size"]) // 2
            img = _crop(img, sh, sh + cfg["crop_size"], sw, sw + cfg["crop_size"])
        final = model(img)
    else:
        raise ValueError(mode)
    _, pred = ops.softmax_max(final.contiguous())  # argmax over classes (first max wins, like torch.argmax)
    return (pred, final) if return_logits else pred


def intersection_and_union(pred, target, K, ignore_index=255, hist=None):
    """Device version of intersectionAndUnion: returns (area_intersection, area_union, area_target) int64 [K]."""
    own = hist is None
    if own:
        hist = ops.zeros(3 * K, dtype=torch.int64, device=pred.device)
    ops.iou_hist(pred.contiguous(), target.contiguous(), K, ignore_index, hist)
    if not own:
        return None
    inter, out, tgt = hist[:K], hist[K:2 * K], hist[2 * K:]
    return inter, out + tgt - inter, tgt


def evaluate(model, loader, mode, cfg):
    """supervised.py:135-164.  Returns (mIoU, iou_class) with the reference's formula (x100, +1e-10)."""
    model.eval()
    assert mode in ["original", "center_crop", "padded_sliding_window", "zegclip_sliding_window", "sliding_window"]
    K = cfg["nclass"]
    hist = None
    with torch.no_grad():
        for img, mask, _ in loader:
            img = img.cuda(non_blocking=True)
            mask = mask.cuda(non_blocking=True)
            if hist is None:
                hist = ops.zeros(3 * K, dtype=torch.int64, device=img.device)
            pred = predict(model, img, mask, mode, cfg)
            intersection_and_union(pred, mask, K, 255, hist)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(hist)
    h = hist.cpu().numpy().astype(np.float64)
    inter, union = h[:K], h[K:2 * K] + h[2 * K:] - h[:K]
    iou_class = inter / (union + 1e-10) * 100.0
    return float(np.mean(iou_class)), iou_class
