"""Input pipeline of the SemiVL loaders with the pixel work on the GPU.

`SemiDataset` is the host half of third_party/unimatch/dataset/semi.py:16-52: split files, PIL decode,
`reduce_zero_label`; it hands out uint8 arrays.  `GpuAugmenter` is the device half of `SemiDataset.__getitem__`
(semi.py:53-127): the random parameters are drawn on the host exactly as `transform.py` / mmseg `Resize` draw them (same
distributions, Python `random` / numpy) and every pixel operation runs on the device (`csrc/augment.hip`): resize (both
branches: `img_scale=None` -> transform.py::resize with Pillow arithmetic; `img_scale=[2048, 512]` -> mmseg
`Resize(img_scale, ratio_range)`, the keep-ratio rescale exp 40 / VOC and ADE use, experiments.py:71) / pad / crop /
flip -> weak view + two strong views (ColorJitter p=0.8, RandomGrayscale p=0.2, blur p=0.5) + ignore mask + CutMix
boxes, normalised and in the layout `semivl_train_step` consumes; `val()` is the validation transform (semi.py:53-60:
short side to 512 for img_scale recipes).  `StepLoader` = `zip(loader_l, loader_u, loader_u)` (semivl.py:200-203) with
host decoding in worker threads and the augmentation of the NEXT batch on a side stream under the current step.
Parity with the reference is statistical: the PIL chain is reproduced op by op (tests compare each op against Pillow
itself), mmseg/mmcv/cv2 are un-vendored (parity unpinned: the keep-ratio rule and OpenCV's published INTER_LINEAR /
INTER_NEAREST sampling rules are restated; cv2's 11-bit fixed-point coefficients are not, +-1 level), the random
streams are not reproduced.
"""
import ctypes as C
import math
import os
import queue
import random

import numpy as np
import torch

from . import lib as L
from .ops import _p, _st

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # transform.py:35
_MEAN3, _STD3 = (C.c_float * 3)(*MEAN), (C.c_float * 3)(*STD)


# ------------------------------------------------------------------------------------------------ kernels
def resample(src, rh, rw, x0, y0, S, flip, nearest=False, fill=0, cv2=False):
    """src uint8 [H, W, C] (C = 3) or [H, W] -> uint8 [OH, OW, C] / [OH, OW]; S = crop size (int) or (OH, OW).
    cv2=False: Pillow BILINEAR (antialiased) / NEAREST; cv2=True: OpenCV INTER_LINEAR / INTER_NEAREST sampling rules."""
    H, W = src.shape[:2]
    OH, OW = (S, S) if isinstance(S, int) else S
    Cc = src.shape[2] if src.dim() == 3 else 1
    dst = torch.empty((OH, OW, Cc) if src.dim() == 3 else (OH, OW), dtype=torch.uint8, device=src.device)
    mode = (3 if nearest else 2) if cv2 else (1 if nearest else 0)
    L.check(L.load().svl_aug_resample_u8(_p(src), H, W, Cc, rh, rw, x0, y0, OH, OW, 1 if flip else 0, mode, fill, _p(dst),
                                         _st()), "svl_aug_resample_u8")
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


def rescale_size(h, w, scale):
    """mmcv.rescale_size for a (long, short) scale tuple: the largest keep-ratio size inside it, rounded half up."""
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(h * f + 0.5), int(w * f + 0.5)


def draw_img_scale(h, w, img_scale, ratio_range):
    """mmseg 0.24 `Resize(img_scale, ratio_range)` with keep_ratio (semi.py:61-71; mmseg is un-vendored: restated from
    its published `random_sample_ratio` + `mmcv.imrescale`): ratio ~ U[min, max) scales BOTH entries of img_scale, the
    image is then rescaled to the largest size that fits (long side <= max(scale), short side <= min(scale))."""
    lo, hi = ratio_range
    ratio = np.random.random_sample() * (hi - lo) + lo
    scale = (int(img_scale[0] * ratio), int(img_scale[1] * ratio))
    return rescale_size(h, w, scale)


def val_size(h, w, img_scale, min_size=512):
    """mmseg `Resize(img_scale, min_size=512)` (semi.py:53-58): short side to max(min(img_scale), min_size) when
    min(img_scale) < min_size else min(img_scale); aspect kept."""
    new_short = min_size if min(img_scale) < min_size else min(img_scale)
    scale = (new_short * h / w, new_short) if h > w else (new_short, new_short * w / h)
    return rescale_size(h, w, scale)


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

    def __init__(self, size, scale_ratio_range=(0.5, 2.0), device="cuda", img_scale=None,
                 labeled_photometric_distortion=False):
        if labeled_photometric_distortion:
            raise NotImplementedError("labeled_photometric_distortion (mmseg PhotoMetricDistortion, semi.py:90-93) is "
                                      "False in every shipped SemiVL recipe (experiments.py:73)")
        self.size, self.ratio, self.device = size, tuple(scale_ratio_range), torch.device(device)
        self.img_scale = tuple(img_scale) if img_scale is not None else None
        self._scratch = None

    @classmethod
    def from_cfg(cls, cfg, device="cuda"):
        """The keys SemiDataset.__init__ reads (semi.py:17-28)."""
        return cls(cfg["crop_size"], cfg.get("scale_ratio_range", (0.5, 2.0)), device, cfg.get("img_scale"),
                   cfg.get("labeled_photometric_distortion", False))

    def _geom(self, img, mask, ignore_value):
        H, W = img.shape[:2]
        cv2 = self.img_scale is not None
        oh, ow = draw_img_scale(H, W, self.img_scale, self.ratio) if cv2 else draw_resize(H, W, self.ratio)
        x0, y0 = draw_crop(oh, ow, self.size)
        flip = random.random() < 0.5
        im = resample(img, oh, ow, x0, y0, self.size, flip, nearest=False, fill=0, cv2=cv2)
        mk = resample(mask, oh, ow, x0, y0, self.size, flip, nearest=True, fill=ignore_value, cv2=cv2)
        return im, mk

    def val(self, img_u8, mask_u8):
        """Validation sample (semi.py:53-60): only the IMAGE is rescaled (short side 512 for img_scale recipes); the mask
        keeps its size -- `predict` resizes the logits to it.  Returns (img float [3, h', w'], mask int64 [h, w])."""
        img = img_u8.to(self.device)
        if self.img_scale is not None:
            oh, ow = val_size(img.shape[0], img.shape[1], self.img_scale)
            img = resample(img, oh, ow, 0, 0, (oh, ow), False, cv2=True)
        return to_float(img), mask_i64(mask_u8.to(self.device))

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


# ------------------------------------------------------------------------------------------------ host side: decode + splits
class SemiDataset:
    """semi.py:16-52 without the pixel work: ids from the split file (labeled ids repeated up to `nsample`), PIL decode,
    `reduce_zero_label` (ADE: 0 -> 255, k -> k-1).  __getitem__ -> (img uint8 [H, W, 3], mask uint8 [H, W], id)."""

    def __init__(self, cfg, mode, id_path=None, nsample=None):
        self.name, self.mode = cfg["dataset"], mode
        self.root = os.path.expandvars(os.path.expanduser(cfg["data_root"]))
        self.reduce_zero_label = cfg.get("reduce_zero_label", False)
        if mode not in ("train_l", "train_u"):
            id_path = id_path or "splits/%s/val.txt" % self.name
        with open(id_path) as f:
            self.ids = f.read().splitlines()
        if mode == "train_l" and nsample is not None:
            self.ids = (self.ids * math.ceil(nsample / len(self.ids)))[:nsample]

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, item):
        from PIL import Image
        id_ = self.ids[item]
        img = np.array(Image.open(os.path.join(self.root, id_.split(" ")[0])).convert("RGB"))
        mask = np.array(Image.open(os.path.join(self.root, id_.split(" ")[1])))
        if self.reduce_zero_label:
            mask = mask.copy()
            mask[mask == 0] = 255
            mask = mask - 1
            mask[mask == 254] = 255
        return torch.from_numpy(img), torch.from_numpy(mask.astype(np.uint8)), id_


def epoch_order(n, epoch, rank=0, world=1, seed=0):
    """torch DistributedSampler(shuffle=True).set_epoch(epoch) index order for this rank (semivl.py:170-178,206-207):
    permutation seeded by seed + epoch, padded by wrap-around to a multiple of world, strided by rank."""
    g = torch.Generator().manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    total = math.ceil(n / world) * world
    idx += idx[:total - n]
    return idx[rank:total:world]


class StepLoader:
    """One epoch of `zip(loader_l, loader_u, loader_u)` (semivl.py:200-203) as ready step-input dicts on the GPU.

    Both `loader_u` iterators of the reference share ONE sampler and epoch seed: the "other" unlabeled
    batch holds the same image ids as the first, re-augmented -- reproduced here by augmenting every unlabeled sample
    twice.  Decoding runs in `workers` host threads (PIL releases the GIL), at most `prefetch` batches ahead; the GPU
    augmentation of batch k+1 is issued on a side stream while step k runs, and the consumer's stream waits on it
    (event-ordered, no host sync) when the batch is handed over."""

    def __init__(self, labeled, unlabeled, augmenter, batch_size, epoch=0, rank=0, world=1, workers=4, prefetch=2):
        self.l, self.u, self.aug, self.bs = labeled, unlabeled, augmenter, batch_size
        self.order_l = epoch_order(len(labeled), epoch, rank, world)
        self.order_u = epoch_order(len(unlabeled), epoch, rank, world)
        self.steps = min(len(self.order_l), len(self.order_u)) // batch_size     # drop_last=True (semivl.py:171-178)
        self.workers, self.prefetch = workers, prefetch

    def __len__(self):
        return self.steps

    def _decode(self, k):
        s = slice(k * self.bs, (k + 1) * self.bs)
        return ([self.l[i][:2] for i in self.order_l[s]], [self.u[i][:2] for i in self.order_u[s]])

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        use_gpu = self.aug.device.type == "cuda"
        side = torch.cuda.Stream(self.aug.device) if use_gpu else None
        pool = ThreadPoolExecutor(max(1, self.workers))
        futs = queue.Queue()
        nxt = 0

        def submit():
            nonlocal nxt
            if nxt < self.steps:
                futs.put(pool.submit(self._decode, nxt))
                nxt += 1

        def augment(host):
            lab, unl = host
            if side is None:
                return self.aug.batch(lab, unl, unl), None
            with torch.cuda.stream(side):
                b = self.aug.batch(lab, unl, unl)
                ev = torch.cuda.Event()
                ev.record(side)
            return b, ev

        try:
            for _ in range(self.prefetch):
                submit()
            ready = augment(futs.get().result()) if self.steps else None
            for k in range(self.steps):
                submit()
                batch, ev = ready
                if k + 1 < self.steps:        # next batch's kernels go to the side stream before this one is consumed
                    ready = augment(futs.get().result())
                if ev is not None:
                    torch.cuda.current_stream().wait_event(ev)
                    for t in batch.values():
                        t.record_stream(torch.cuda.current_stream())
                yield batch
        finally:
            pool.shutdown(wait=False, cancel_futures=True)
