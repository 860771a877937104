"""Pixel losses on head-resolution logits (svl_softmax_max_up_f32 / svl_ce_up_fused_f32): the kernels evaluate the
bilinear resize of vlg_head.py:247 / builder.py:93-97 themselves.  Checked against F.interpolate + the plain PyTorch
loss of semivl.py:232,252,267-323 (autograd through the resize gives the low-resolution gradient), against the unfused
kernels of the same library, and at step level against the step that writes the resized tensors."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, h, w, H, W): the training geometries (VOC / COCO / ADE 128 -> 512, Cityscapes 204 -> 801), tiny fixtures (8 -> 32),
# ragged tiles, non-integer ratios, ratio 1 and 2
GEOMS = [(21, 128, 128, 512, 512), (19, 204, 204, 801, 801), (5, 8, 8, 32, 32), (150, 24, 40, 96, 160),
         (7, 13, 20, 50, 79), (3, 9, 9, 9, 9), (4, 17, 11, 34, 22), (81, 16, 16, 64, 64)]


def _rnd(*shape, dev, seed, scale=2.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("N,h,w,H,W", GEOMS)
def test_softmax_max_up(dev, N, h, w, H, W, align):
    from semivl_amd import ops
    B = 2
    lg = _rnd(B, N, h, w, dev=dev, seed=31)
    assert ops.ce_up_ok(B, N, h, w, H, W, align)
    conf, lab = ops.softmax_max_up(lg, H, W, align)
    up = F.interpolate(lg, size=(H, W), mode="bilinear", align_corners=align)
    rc, rl = up.softmax(1).max(1)
    assert (conf - rc).abs().max().item() < 2e-6
    top2 = up.topk(2, dim=1).values
    tie = (top2[:, 0] - top2[:, 1]) < 1e-5            # (the resize's fma contraction may differ by an ulp)
    assert torch.equal(lab[~tie], rl[~tie]), "labels must agree off ties"
    # the library's own resize + softmax-max: same expression, bit-equal maps
    c2, l2 = ops.softmax_max(ops.bilinear_planes_fwd(lg, h, w, align, H, W))
    assert (lab != l2).sum().item() <= int(tie.sum().item())
    assert (conf - c2).abs().max().item() < 2e-6
    # a constant map: first maximum wins
    _, l0 = ops.softmax_max_up(torch.zeros(1, N, h, w, device=dev), H, W, align)
    assert (l0 == 0).all()


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("N,h,w,H,W", GEOMS)
def test_ce_up_fused(dev, N, h, w, H, W, align):
    from semivl_amd import ops
    B = 2
    lg = _rnd(B, N, h, w, dev=dev, seed=32).requires_grad_(True)
    up = F.interpolate(lg, size=(H, W), mode="bilinear", align_corners=align)
    g = torch.Generator(device="cpu").manual_seed(33)
    # supervised branch: CE(ignore_index = 255, mean)
    tgt = torch.randint(0, N, (B, H, W), generator=g)
    tgt[torch.rand(B, H, W, generator=g) < 0.1] = 255
    tgt = tgt.to(dev)
    ref = F.cross_entropy(up, tgt, ignore_index=255)
    (gr,) = torch.autograd.grad(ref, lg, retain_graph=True)
    nval = int((tgt != 255).sum().item())
    gs = torch.tensor([1.0 / nval, 0.0], device=dev)
    dl = torch.full_like(lg, float("nan"))
    sums = ops.ce_up_fused(lg.detach(), H, W, align, tgt, True, dlogits=dl, gscale=gs)
    assert abs((sums[0] / sums[3]).item() - ref.item()) < 1e-5 and sums[3].item() == nval
    assert torch.allclose(dl, gr, atol=2e-8, rtol=2e-4), (dl - gr).abs().max().item()
    # unsupervised branch: pixelwise confidence weighting + guidance term (semivl.py:274-284)
    lab = torch.randint(0, N, (B, H, W), generator=g).to(dev)
    conf = torch.rand(B, H, W, generator=g).to(dev)
    ign = torch.zeros(B, H, W, dtype=torch.int64)
    ign[:, -3:] = 255
    ign = ign.to(dev)
    mc = torch.randint(0, N, (B, H, W), generator=g)
    mc[torch.rand(B, H, W, generator=g) < 0.5] = 255
    mc = mc.to(dev)
    valid = ign != 255
    lu = F.cross_entropy(up, lab, reduction="none")
    lu = (lu * ((conf >= 0.7) & valid)).sum() / valid.sum().item()
    lm = F.cross_entropy(up, mc, ignore_index=255, reduction="none").sum() / ign.numel()
    (g2,) = torch.autograd.grad(0.125 * lu + 0.03 * lm, lg, retain_graph=True)
    gs2 = torch.tensor([0.125 / valid.sum().item(), 0.03 / ign.numel()], device=dev)
    dl2 = torch.full_like(lg, float("nan"))
    s2 = ops.ce_up_fused(lg.detach(), H, W, align, lab, False, conf=conf, ign=ign, conf_thresh=0.7, mc=mc, dlogits=dl2,
                         gscale=gs2)
    assert abs((s2[0] / s2[3]).item() - lu.item()) < 1e-5
    assert abs((s2[1] / ign.numel()).item() - lm.item()) < 1e-5
    assert abs(s2[2].item() - (conf * valid).sum().item()) < 1e-2 + 1e-5 * conf.numel()
    assert torch.allclose(dl2, g2, atol=2e-9, rtol=2e-4), (dl2 - g2).abs().max().item()
    # deterministic, and the forward-only call returns the same sums
    dl3 = torch.empty_like(lg)
    s3 = ops.ce_up_fused(lg.detach(), H, W, align, lab, False, conf=conf, ign=ign, conf_thresh=0.7, mc=mc, dlogits=dl3,
                         gscale=gs2)
    assert torch.equal(s3, s2) and torch.equal(dl3, dl2)
    s4 = ops.ce_up_fused(lg.detach(), H, W, align, lab, False, conf=conf, ign=ign, conf_thresh=0.7, mc=mc)
    assert torch.equal(s4, s2)
    # 'pixelavg' / 'pixelratio' (train_utils.py:39-46): the whole map, a per-image factor
    iw = torch.tensor([0.25, 0.75], device=dev)
    lw = (F.cross_entropy(up, lab, reduction="none") * iw[:, None, None]).sum() / valid.sum().item()
    (g5,) = torch.autograd.grad(lw, lg)
    gs5 = torch.tensor([1.0 / valid.sum().item(), 0.0], device=dev)
    dl5 = torch.empty_like(lg)
    s5 = ops.ce_up_fused(lg.detach(), H, W, align, lab, False, conf=conf, ign=ign, conf_thresh=0.7, dlogits=dl5, gscale=gs5,
                         all_pixels=True, img_weight=iw)
    assert abs((s5[0] / s5[3]).item() - lw.item()) < 1e-5
    assert torch.allclose(dl5, g5, atol=2e-9, rtol=2e-4)
    # the unfused kernels of the library on the resized tensor: same sums, and their gradient through the resize backward
    upk = ops.bilinear_planes_fwd(lg.detach().contiguous(), h, w, align, H, W)
    dlf = torch.empty_like(upk)
    sf = ops.ce_fused(upk, lab, False, conf=conf, ign=ign, conf_thresh=0.7, mc=mc, dlogits=dlf, gscale=gs2)
    assert torch.allclose(sf, s2, rtol=1e-6, atol=1e-6)
    assert torch.allclose(ops.bilinear_planes_bwd(dlf, h, w, align, H, W), dl2, atol=2e-9, rtol=2e-4)


@pytest.mark.parametrize("B,N,h,H", [(16, 21, 128, 512), (8, 19, 204, 801), (4, 150, 128, 512), (4, 81, 128, 512)])
def test_ce_up_invariants_at_baseline_sizes(dev, B, N, h, H):
    """Size-independent properties at BASELINE.json's batch and crop (no reference tensor of that size needed): every
    head-resolution cell's gradient sums to zero over the classes (each pixel contributes g (p - onehot_t) + g_m (p - onehot_m),
    both sum to zero, and the resize's weights do not depend on the class); the per-image loss sums are those of the same
    images evaluated alone (a block never reads another image); two runs are bit-identical; softmax-max labels are the
    argmax of the library's own resized logits on a sample of the images."""
    from semivl_amd import ops
    g = torch.Generator(device="cpu").manual_seed(51)
    lg = (torch.randn(B, N, h, h, generator=g) * 3).to(dev)
    lab = torch.randint(0, N, (B, H, H), generator=g).to(dev)
    conf = torch.rand(B, H, H, generator=g).to(dev)
    ign = torch.zeros(B, H, H, dtype=torch.int64)
    ign[:, :, :7] = 255
    ign = ign.to(dev)
    mc = torch.randint(0, N, (B, H, H), generator=g)
    mc[torch.rand(B, H, H, generator=g) < 0.7] = 255
    mc = mc.to(dev)
    gs = torch.tensor([1.0 / (B * H * H), 0.1 / (B * H * H)], device=dev)
    dl = torch.empty_like(lg)
    kw = dict(conf=conf, ign=ign, conf_thresh=0.4, mc=mc, gscale=gs)
    s = ops.ce_up_fused(lg, H, H, False, lab, False, dlogits=dl, **kw)
    scale = dl.abs().max().item()
    assert scale > 0 and dl.sum(1).abs().max().item() < 2e-5 * scale * N ** 0.5
    dl2 = torch.empty_like(lg)
    s2 = ops.ce_up_fused(lg, H, H, False, lab, False, dlogits=dl2, **kw)
    assert torch.equal(s, s2) and torch.equal(dl, dl2)
    # images evaluated alone: same gradient bits (same gscale), loss sums add up
    tot = torch.zeros(4, dtype=torch.float64, device=dev)
    for b in (0, B - 1):
        d1 = torch.empty_like(lg[b:b + 1])
        s1 = ops.ce_up_fused(lg[b:b + 1].contiguous(), H, H, False, lab[b:b + 1].contiguous(), False, conf=conf[b:b + 1].contiguous(),
                             ign=ign[b:b + 1].contiguous(), conf_thresh=0.4, mc=mc[b:b + 1].contiguous(), gscale=gs, dlogits=d1)
        assert torch.equal(d1[0], dl[b])
        tot += s1
    if B == 2:
        assert torch.allclose(tot, s, rtol=1e-6)
    conf_, lab_ = ops.softmax_max_up(lg, H, H, False)
    up = ops.bilinear_planes_fwd(lg[:2].contiguous(), h, h, False, H, H)
    top2 = up.topk(2, dim=1).values
    tie = (top2[:, 0] - top2[:, 1]) < 1e-5
    assert torch.equal(lab_[:2][~tie], up.argmax(1)[~tie])
    assert (conf_[:2] - up.softmax(1).amax(1)).abs().max().item() < 5e-6


def test_unsupported_geometries_are_refused(dev):
    from semivl_amd import ops
    assert not ops.ce_up_ok(2, 21, 64, 64, 512, 512, False)       # ratio 8: the region of a block would not fit
    assert not ops.ce_up_ok(2, 21, 128, 128, 64, 64, False)       # a reduction
    assert not ops.ce_up_ok(2, 300, 128, 128, 512, 512, False)    # the staged tile would not fit
    with pytest.raises(RuntimeError):
        ops.softmax_max_up(torch.zeros(1, 21, 64, 64, device=dev), 512, 512, False)


@pytest.mark.parametrize("name", ["tiny", "conf"])
def test_step_with_the_resize_inside_the_loss_equals_the_step_on_resized_logits(dev, name):
    """Same weights, inputs and dropout masks: the eight losses, the label maps and every parameter gradient of the step
    that keeps the logits at the head's resolution against the step that writes the resized tensors."""
    from golden_util import build_hip, fixture_batch, fixture_fp_masks, fixture_state, load_fixture
    from semivl_amd.train import semivl_train_step
    z, c = load_fixture(name)
    outs = []
    for fuse in (True, False):
        hip = build_hip(c)
        hip.load_state_dict(fixture_state(z, c, hip), strict=True)
        hip.to(dev)
        batch = {k: v.to(dev) for k, v in fixture_batch(z, c).items()}
        masks = [m.to(dev) for m in fixture_fp_masks(z, c)]
        cfg = dict(conf_thresh=0.05, conf_mode="pixelwise", mcc_conf_thresh=0.9, mcc_loss_reduce="mean_all",
                   maskclip_consistency_lambda=[0.1, 0], fuse_upsample_loss=fuse)
        for p in hip.parameters():
            p.grad = None
        losses, aux = semivl_train_step(hip, batch, 1, 10, cfg, fp_masks=masks, return_aux=True)
        assert aux["upsample_in_loss"] == fuse
        grads = {n: p.grad.detach().clone() for n, p in hip.named_parameters() if p.grad is not None}
        outs.append((losses.clone(), aux, grads))
    (la, aa, ga), (lb, ab, gb) = outs
    assert (la - lb).abs().max().item() < 1e-5, (la, lb)
    for k in ("mask_w", "mask_w_other"):
        assert (aa[k] != ab[k]).float().mean().item() < 1e-3, k
    # (the two forms contract the resize's fma chain differently: an ulp of the largest logit)
    ulp = 2.0 ** -22 * max(ab["pred_x"].abs().max().item(), ab["pred_w"].abs().max().item(), 1.0)
    assert (aa["conf_w"] - ab["conf_w"]).abs().max().item() < 4 * ulp + 1e-6
    assert (aa["pred_x"] - ab["pred_x"]).abs().max().item() < 4 * ulp
    assert ga.keys() == gb.keys() and len(ga) > 10
    scale = max(g_.abs().max().item() for g_ in gb.values())
    for n in ga:
        if n == "decode_head.head.bias":      # sum(softmax - onehot) = 0: rounding noise in either form
            assert ga[n].abs().max().item() < 1e-4 * scale
            continue
        num = (ga[n] - gb[n]).norm().item()
        den = gb[n].norm().item()
        # (absolute floor: the head conv's bias gradient is sum(dlogits) = 0 up to rounding in either form)
        assert num <= 2e-4 * den + 1e-5 * scale * ga[n].numel() ** 0.5, (n, num, den, scale)
