"""GPU-side augmentation for the SemiVL loaders.

`GpuAugmenter` restates `SemiDataset.__getitem__` for the train modes
(third_party/unimatch/dataset/semi.py:61-127) with the random parameters drawn on the host exactly as `transform.py`
draws them (same distributions, Python `random` / numpy) and every pixel operation on the device
(`csrc/augment.hip`): decode -> uint8 HWC tensor on the GPU -> resize / pad / crop / flip -> weak view + two strong
views (ColorJitter p=0.8, RandomGrayscale p=0.2, blur p=0.5) + ignore mask + CutMix boxes, already normalised and in
the layout `semivl_train_step` consumes.  Image decoding and the split files stay on the host (PIL), as in the
reference.  Parity with the reference is statistical: its PIL chain is reproduced op by op (tests compare each op
against Pillow itself), the random streams are not.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import lib as L
from .ops import _p, _st

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # transform.py:35
_MEAN3, _STD3 = (C.c_float * 3)(*MEAN), (C.c_float * 3)(*STD)


# ------------------------------------------------------------------------------------------------ kernels
def resample(src, rh, rw, x0, y0, S, flip, nearest=False, fill=0):
    """src uint8 [H, W, C] (C = 3) or [H, W] -> uint8 [S, S, C] / [S, S]."""
    H, W = src.shape[:2]
    Cc = src.shape[2] if src.dim() == 3 else 1
    dst = torch.empty((S, S, Cc) if src.dim() == 3 else (S, S), dtype=torch.uint8, device=src.device)
    L.check(L.load().svl_aug_resample_u8(_p(src), H, W, Cc, rh, rw, x0, y0, S, 1 if flip else 0, 1 if nearest else 0,
                                         fill, _p(dst), _st()), "svl_aug_resample_u8")
    return dst


def to_float(img_u8):
    """ToTensor + Normalize: uint8 [S, S, 3] -> float32 [3, S, S]."""
    S0, S1 = img_u8.shape[:2]
    out = torch.empty(3, S0, S1, dtype=torch.float32, device=img_u8.device)
    L.check(L.load().svl_aug_to_float(_p(img_u8), S0 * S1, _MEAN3, _STD3, _p(out), _st()), "svl_aug_to_float")
    return out


def mask_i64(mask_u8, frm=-1, to=-1):
    out = torch.empty(mask_u8.shape, dtype=torch.int64, device=mask_u8.device)
    L.check(L.load().svl_aug_mask_i64(_p(mask_u8), mask_u8.numel(), frm, to, _p(out), _st()), "svl_aug_mask_i64")
    return out


BRIGHTNESS, CONTRAST, SATURATION, HUE, GRAYSCALE = range(5)


def photometric_(img_u8, op, factor=1.0, scratch=None):
    if scratch is None:
        scratch = torch.zeros(1, dtype=torch.int64, device=img_u8.device)
    L.check(L.load().svl_aug_photometric_u8(_p(img_u8), img_u8.shape[0] * img_u8.shape[1], op, float(factor), _p(scratch),
                                            _st()), "svl_aug_photometric_u8")
    return img_u8


def gaussian_blur(img_u8, sigma):
    tmp, out = torch.empty_like(img_u8), torch.empty_like(img_u8)
    L.check(L.load().svl_aug_gaussian_blur_u8(_p(img_u8), img_u8.shape[0], img_u8.shape[1], float(sigma), _p(tmp),
                                              _p(out), _st()), "svl_aug_gaussian_blur_u8")
    return out


# ------------------------------------------------------------------------------------------------ parameter draws
def draw_resize(h, w, ratio_range):
    """transform.py:43-57 -> (oh, ow)."""
    long_side = random.randint(int(max(h, w) * ratio_range[0]), int(max(h, w) * ratio_range[1]))
    if h > w:
        return long_side, int(1.0 * w * long_side / h + 0.5)
    return int(1.0 * h * long_side / w + 0.5), long_side


def draw_crop(oh, ow, size):
    """transform.py:9-20 -> (x0, y0) in the padded image."""
    pw, ph = max(ow, size), max(oh, size)
    return random.randint(0, pw - size), random.randint(0, ph - size)


def draw_cutmix_box(img_size, p=0.5, size_min=0.02, size_max=0.4, ratio_1=0.3, ratio_2=1 / 0.3):
    """transform.py:66-84 -> (x, y, w, h) or None."""
    if random.random() > p:
        return None
    size = np.random.uniform(size_min, size_max) * img_size * img_size
    while True:
        ratio = np.random.uniform(ratio_1, ratio_2)
        cw, ch = int(np.sqrt(size / ratio)), int(np.sqrt(size * ratio))
        x, y = np.random.randint(0, img_size), np.random.randint(0, img_size)
        if x + cw <= img_size and y + ch <= img_size:
            return x, y, cw, ch


def draw_color_jitter(brightness=0.5, contrast=0.5, saturation=0.5, hue=0.25):
    """torchvision ColorJitter.get_params: a random order of the four ops and their factors."""
    order = list(np.random.permutation(4))
    f = {BRIGHTNESS: random.uniform(max(0, 1 - brightness), 1 + brightness),
         CONTRAST: random.uniform(max(0, 1 - contrast), 1 + contrast),
         SATURATION: random.uniform(max(0, 1 - saturation), 1 + saturation), HUE: random.uniform(-hue, hue)}
    return [(int(o), f[int(o)]) for o in order]


class GpuAugmenter:
    """semi.py `SemiDataset` train modes on the device.  `size` = crop size, `scale_ratio_range` = cfg['scale_ratio_range']
    (VOC (0.5, 2.0))."""

    def __init__(self, size, scale_ratio_range=(0.5, 2.0), device="cuda"):
        self.size, self.ratio, self.device = size, scale_ratio_range, torch.device(device)
        self._scratch = None

    def _geom(self, img, mask, ignore_value):
        H, W = img.shape[:2]
        oh, ow = draw_resize(H, W, self.ratio)
        x0, y0 = draw_crop(oh, ow, self.size)
        flip = random.random() < 0.5
        im = resample(img, oh, ow, x0, y0, self.size, flip, nearest=False, fill=0)
        mk = resample(mask, oh, ow, x0, y0, self.size, flip, nearest=True, fill=ignore_value)
        return im, mk

    def _strong(self, im):
        """semi.py:98-103: ColorJitter(0.5, 0.5, 0.5, 0.25) with p = 0.8, RandomGrayscale(0.2), blur(0.5)."""
        s = im.clone()
        if self._scratch is None:
            self._scratch = torch.zeros(1, dtype=torch.int64, device=im.device)
        if random.random() < 0.8:
            for op, f in draw_color_jitter():
                photometric_(s, op, f, self._scratch)
        if random.random() < 0.2:
            photometric_(s, GRAYSCALE)
        if random.random() < 0.5:
            s = gaussian_blur(s, np.random.uniform(0.1, 2.0))
        return s

    def _box(self):
        m = torch.zeros(self.size, self.size, device=self.device)
        b = draw_cutmix_box(self.size)
        if b is not None:
            x, y, w, h = b
            m[y:y + h, x:x + w] = 1
        return m

    def train_l(self, img_u8, mask_u8):
        """Labeled sample (semi.py:88-95): (img float [3,S,S], mask int64 [S,S], padding = 255)."""
        im, mk = self._geom(img_u8.to(self.device), mask_u8.to(self.device), 255)
        return to_float(im), mask_i64(mk)

    def train_u(self, img_u8, mask_u8=None):
        """Unlabeled sample (semi.py:97-125): img_w, img_s1, img_s2, ignore_mask, cutmix_box1, cutmix_box2."""
        img = img_u8.to(self.device)
        mask = mask_u8.to(self.device) if mask_u8 is not None else torch.zeros(img.shape[:2], dtype=torch.uint8,
                                                                              device=self.device)
        im, mk = self._geom(img, mask, 254)
        s1, s2 = self._strong(im), self._strong(im)
        box1, box2 = self._box(), self._box()
        ign = mask_i64(torch.where(mk == 254, mk, torch.zeros_like(mk)), 254, 255)   # 255 on the padding, 0 elsewhere
        return to_float(im), to_float(s1), to_float(s2), ign, box1, box2

    def batch(self, labeled, unlabeled, unlabeled_other):
        """Lists of (img_u8 [H,W,3], mask_u8 [H,W]) -> the 12-tensor dict of one SemiVL step (semivl.py:205-221)."""
        xs = [self.train_l(i, m) for i, m in labeled]
        us = [self.train_u(i, m) for i, m in unlabeled]
        uo = [self.train_u(i, m) for i, m in unlabeled_other]
        st = lambda seq, k: torch.stack([s[k] for s in seq])
        return dict(img_x=st(xs, 0), mask_x=st(xs, 1), img_w=st(us, 0), img_s1=st(us, 1), img_s2=st(us, 2),
                    ignore_mask=st(us, 3), mix1=st(us, 4), mix2=st(us, 5), img_w_other=st(uo, 0),
                    img_s1_other=st(uo, 1), img_s2_other=st(uo, 2), ignore_mask_other=st(uo, 3))
