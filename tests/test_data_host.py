"""CPU: host half of the input pipeline -- the scale rules of mmseg `Resize` as semi.py:53-71 uses them,
split files / decoding / reduce_zero_label (semi.py:16-52), sampler order and the zip(loader_l, loader_u, loader_u) loop
(semivl.py:170-178,200-207)."""
import os

import numpy as np
import torch
from PIL import Image


def test_img_scale_rules():
    from semivl_amd import data
    # keep-ratio rescale: the largest size with long side <= max(scale) and short side <= min(scale), rounded half up
    assert data.rescale_size(375, 500, (2048, 512)) == (512, 683)       # short side binds (VOC landscape)
    assert data.rescale_size(500, 375, (2048, 512)) == (683, 512)
    assert data.rescale_size(100, 1000, (2048, 512)) == (205, 2048)     # long side binds (extreme panorama)
    np.random.seed(0)
    sizes = [data.draw_img_scale(375, 500, (2048, 512), (0.5, 2.0)) for _ in range(4000)]
    shorts = np.array([min(s) for s in sizes])
    # ratio ~ U[0.5, 2) scales (2048, 512): the short side of a VOC image lands uniformly in [256, 1024)
    assert shorts.min() >= 256 and shorts.max() <= 1024 and abs(shorts.mean() - 640) < 12
    assert np.all(np.abs(np.array([s[1] / s[0] for s in sizes]) - 500 / 375) < 0.01)
    hist, _ = np.histogram(shorts, bins=4, range=(256, 1024))
    assert hist.min() > 0.2 * len(sizes)                                # uniform, unlike transform.py::resize's long-side rule
    # validation: short side -> 512, aspect kept (semi.py:53-58: Resize(img_scale, min_size=512))
    assert data.val_size(375, 500, (2048, 512)) == (512, 683)
    assert data.val_size(500, 334, (2048, 512)) == (766, 512)
    assert data.val_size(1024, 2048, (2048, 1024)) == (1024, 2048)


def _make_split(tmp, n, zero_label=False):
    os.makedirs(os.path.join(tmp, "img")); os.makedirs(os.path.join(tmp, "lab"))
    rng = np.random.RandomState(0)
    lines = []
    for i in range(n):
        Image.fromarray(rng.randint(0, 256, (40 + i, 50, 3)).astype(np.uint8)).save(os.path.join(tmp, f"img/{i}.png"))
        Image.fromarray(rng.randint(0, 5, (40 + i, 50)).astype(np.uint8)).save(os.path.join(tmp, f"lab/{i}.png"))
        lines.append(f"img/{i}.png lab/{i}.png")
    path = os.path.join(tmp, "split.txt")
    open(path, "w").write("\n".join(lines))
    return path


def test_dataset_decode_and_reduce_zero_label(tmp_path):
    from semivl_amd.data import SemiDataset
    split = _make_split(str(tmp_path), 3)
    cfg = dict(dataset="pascal", data_root=str(tmp_path))
    ds = SemiDataset(cfg, "train_l", split, nsample=8)
    assert len(ds) == 8 and ds.ids[3] == ds.ids[0]                      # labeled ids repeated up to nsample
    img, mask, id_ = ds[1]
    assert img.dtype == torch.uint8 and img.shape == (41, 50, 3) and mask.shape == (41, 50) and id_.startswith("img/1")
    raw = np.array(Image.open(os.path.join(str(tmp_path), "lab/1.png")))
    assert np.array_equal(mask.numpy(), raw)
    ade = SemiDataset(dict(cfg, reduce_zero_label=True), "val", split)
    m2 = ade[1][1].numpy()
    assert np.array_equal(m2 == 255, raw == 0) and np.array_equal(m2[raw > 0], raw[raw > 0] - 1)


def test_epoch_order_is_distributed_sampler_order():
    from torch.utils.data.distributed import DistributedSampler
    from semivl_amd.data import epoch_order
    ds = list(range(23))
    for world in (1, 2, 4):
        for rank in range(world):
            s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True)
            s.set_epoch(5)
            assert list(s) == epoch_order(23, 5, rank, world)


class _HostAug:
    """Stand-in augmenter (CPU): records what it was handed, returns one marker tensor."""
    device = torch.device("cpu")

    def batch(self, lab, unl, unl_other):
        assert unl is unl_other or all(a[0].data_ptr() == b[0].data_ptr() for a, b in zip(unl, unl_other))
        return dict(n_l=torch.tensor([int(i[0, 0, 0]) for i, _ in lab]), n_u=torch.tensor([int(i[0, 0, 0]) for i, _ in unl]))


def test_step_loader_zip_semantics():
    from semivl_amd.data import StepLoader, epoch_order
    mk = lambda n, off: [(torch.full((4, 4, 3), off + i, dtype=torch.uint8), torch.zeros(4, 4, dtype=torch.uint8), str(i))
                         for i in range(n)]
    lab, unl = mk(9, 0), mk(14, 100)
    ld = StepLoader(lab, unl, _HostAug(), batch_size=2, epoch=3, rank=1, world=2, workers=2)
    ol, ou = epoch_order(9, 3, 1, 2), epoch_order(14, 3, 1, 2)
    assert len(ld) == min(len(ol), len(ou)) // 2 == 2                    # zip stops at the shorter loader, drop_last
    got = list(ld)
    assert len(got) == 2
    for k, b in enumerate(got):
        assert b["n_l"].tolist() == [ol[2 * k], ol[2 * k + 1]]
        assert b["n_u"].tolist() == [100 + ou[2 * k], 100 + ou[2 * k + 1]]
