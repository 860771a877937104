"""CPU: COMPUTE through the C-ABI without a GPU -- the CPU reference backend (oracle/cabi_cpu.c, same header, same
argument conventions as libsemivl_hip.so) against PyTorch / the oracle on the host; the `-m gpu` half holds the HIP
library against this second implementation of the same entry points (integer outputs bit-exact)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cabi_cpu as K


def _case(B=2, N=21, H=24, W=20, seed=0):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(B, N, H, W, generator=g) * 3).contiguous()
    target = torch.randint(0, N, (B, H, W), generator=g)
    target[torch.rand(B, H, W, generator=g) < 0.1] = 255
    conf = torch.rand(B, H, W, generator=g)
    ign = torch.zeros(B, H, W, dtype=torch.int64)
    ign[:, -5:] = 255
    mc = torch.randint(0, N, (B, H, W), generator=g)
    mc[torch.rand(B, H, W, generator=g) < 0.5] = 255
    return logits, target, conf, ign, mc


def test_cpu_backend_exports_the_header_signatures():
    lib = K.load()
    assert lib.svl_version() >= 300
    assert lib.svl_fill_f32(None, 0.0, 0, None) == -1        # same error convention as the HIP library
    buf = C.create_string_buffer(64)
    lib.svl_last_error(buf, 64)
    assert b"svl_fill_f32" in buf.value


def test_cpu_softmax_max_and_cutmix():
    lib = K.load()
    logits, target, conf, ign, _ = _case()
    B, N, H, W = logits.shape
    x = logits.numpy()
    cf, lab = np.zeros((B, H, W), np.float32), np.zeros((B, H, W), np.int64)
    K.check(lib.svl_softmax_max_f32(K.ptr(x), B, N, H * W, K.ptr(cf), K.ptr(lab), None))
    rc, rl = logits.softmax(1).max(1)
    assert np.array_equal(lab, rl.numpy()) and np.abs(cf - rc.numpy()).max() < 1e-6
    t = np.zeros((1, N, 8), np.float32)                     # ties -> lowest index
    l2, c2 = np.zeros((1, 8), np.int64), np.zeros((1, 8), np.float32)
    K.check(lib.svl_softmax_max_f32(K.ptr(t), 1, N, 8, K.ptr(c2), K.ptr(l2), None))
    assert (l2 == 0).all()
    box = (torch.rand(B, H, W) < 0.3).float().numpy()
    a, b_ = target.numpy(), (target.numpy() + 1) % 7
    out = np.zeros_like(a)
    K.check(lib.svl_cutmix_i64(K.ptr(out), K.ptr(a), K.ptr(b_), K.ptr(box), B, H * W, None))
    assert np.array_equal(out, np.where(box == 1, b_, a))
    cnt = np.zeros(1, np.int64)
    K.check(lib.svl_count_valid_i64(K.ptr(a), a.size, K.ptr(cnt), None))
    assert cnt[0] == (a != 255).sum()


@pytest.mark.parametrize("all_pixels", [False, True])
def test_cpu_ce_fused_matches_autograd(all_pixels):
    """Forward sums and d(loss)/d(logits) of the fused CE entry point == torch autograd of the reference expressions
    (semivl.py:275-310: CE(reduction='none') * confidence gate + mc CE(ignore 255))."""
    logits, target, conf, ign, mc = _case(seed=3)
    gs = np.array([0.37, 0.11], np.float32)
    sums, dl = K.ce_fused(logits.numpy(), target.clamp(max=20).numpy(), False, conf.numpy(), ign.numpy(), 0.6, mc.numpy(),
                          gs, all_pixels)
    lt = logits.clone().requires_grad_(True)
    tgt = target.clamp(max=20)
    ce = F.cross_entropy(lt, tgt, reduction="none")
    valid = ign != 255
    w = torch.ones_like(conf) if all_pixels else ((conf >= 0.6) & valid).float()
    ce_m = F.cross_entropy(lt, mc, ignore_index=255, reduction="none")
    (gs[0] * (w * ce).sum() + gs[1] * ce_m.sum()).backward()
    assert abs(sums[0] - (w * ce).sum().item()) < 1e-3 * max(1.0, abs(sums[0]))
    assert abs(sums[1] - ce_m.sum().item()) < 1e-3 * max(1.0, abs(sums[1]))
    assert abs(sums[2] - (conf * valid).sum().item()) < 1e-3 and sums[3] == valid.sum().item()
    assert np.abs(dl - lt.grad.numpy()).max() < 1e-5
    # labeled branch: ignore_index 255 on the target, no confidence maps
    sums_x, dl_x = K.ce_fused(logits.numpy(), target.numpy(), True, gscale=np.array([0.5, 0.0], np.float32))
    lt2 = logits.clone().requires_grad_(True)
    lx = F.cross_entropy(lt2, target, ignore_index=255, reduction="sum")
    (0.5 * lx).backward()
    assert abs(sums_x[0] - lx.item()) < 1e-3 * abs(lx.item()) and sums_x[3] == (target != 255).sum().item()
    assert np.abs(dl_x - lt2.grad.numpy()).max() < 1e-5


def test_cpu_loss_assembly_matches_oracle_formula():
    from oracle import semivl_oracle as O
    lib = K.load()
    logits, target, conf, ign, mc = _case(seed=5)
    lam = 0.07
    numel = float(ign.numel())
    counts = np.array([(target != 255).sum(), (ign != 255).sum(), (ign != 255).sum(), (ign != 255).sum()], np.int64)
    gs = np.zeros((4, 2), np.float32)
    K.check(lib.svl_semivl_gscale(K.ptr(counts), numel, lam, None, None, K.ptr(gs), None))
    assert math.isclose(gs[0, 0], 0.5 / counts[0], rel_tol=1e-6) and math.isclose(gs[3, 1], 0.5 * lam / numel, rel_tol=1e-6)
    sums = np.zeros((4, 4), np.float64)
    sums[0], _ = K.ce_fused(logits.numpy(), target.numpy(), True)
    tg = target.clamp(max=20)
    for i in (1, 2, 3):
        sums[i], _ = K.ce_fused(logits.numpy(), tg.numpy(), False, conf.numpy(), ign.numpy(), 0.6, mc.numpy())
    out = np.zeros(8, np.float32)
    K.check(lib.svl_semivl_loss(K.ptr(sums), numel, lam, None, None, K.ptr(out), None))
    ce = F.cross_entropy(logits, tg, reduction="none")
    lu = O.confidence_weighted_loss(ce, conf, ign, "pixelwise", 0.6)
    lmc = O.compute_mc_loss(logits, mc, ign)
    lx = F.cross_entropy(logits, target, ignore_index=255)
    ref = (lx + lu * 0.25 + lu * 0.25 + lu * 0.5) / 2.0 + lmc * 0.25 * lam + lmc * 0.25 * lam + lmc * 0.5 * lam
    assert abs(out[0] - ref.item()) < 1e-5 and abs(out[1] - lx.item()) < 1e-5 and abs(out[5] - lmc.item()) < 1e-6
    # pixelavg factor (train_utils.py:43-46)
    f = np.zeros(1, np.float64)
    ws = np.zeros(int(lib.svl_conf_avg_ws_doubles(2)), np.float64)
    K.check(lib.svl_conf_avg_factor(K.ptr(conf.numpy()), K.ptr(ign.numpy()), 2, ign[0].numel(), K.ptr(f), K.ptr(ws), None))
    v = (ign != 255)
    assert abs(f[0] - ((conf * v).sum((1, 2)) / v.sum((1, 2))).sum().item()) < 1e-6


@pytest.mark.parametrize("reduce", ["mean_valid", "mean"])
def test_cpu_pixelratio_and_mc_reduce_match_oracle_formula(reduce):
    """conf_mode 'pixelratio' (train_utils.py:39-42: per-image share of confident valid pixels on the WHOLE CE map) and the
    guidance loss's other two normalisers (semivl.py:52-58) through the ABI: ratios, forward sums, d(loss)/d(logits) and
    the assembled loss against torch autograd of the oracle's expressions."""
    from oracle import semivl_oracle as O
    lib = K.load()
    logits, target, conf, ign, mc = _case(seed=11)
    tg = target.clamp(max=20)
    B = logits.shape[0]
    ratio = np.zeros(B, np.float32)
    ws = np.zeros(int(lib.svl_conf_avg_ws_doubles(B)), np.float64)
    K.check(lib.svl_conf_ratio_f32(K.ptr(conf.numpy()), K.ptr(ign.numpy()), B, ign[0].numel(), 0.6, K.ptr(ratio), K.ptr(ws), None))
    v = ign != 255
    ref_ratio = ((conf >= 0.6) & v).sum((1, 2)) / v.sum((1, 2))
    assert np.array_equal(ratio, ref_ratio.numpy())
    lam, numel = 0.07, float(ign.numel())
    counts = np.array([(target != 255).sum(), v.sum(), v.sum(), v.sum()], np.int64)
    mcn = counts[1:].copy() if reduce == "mean_valid" else np.array([(mc != 255).sum()] * 3, np.int64)
    gs = np.zeros((4, 2), np.float32)
    K.check(lib.svl_semivl_gscale(K.ptr(counts), numel, lam, None, K.ptr(mcn), K.ptr(gs), None))
    sums = np.zeros((4, 4), np.float64)
    sums[0], _ = K.ce_fused(logits.numpy(), target.numpy(), True)
    dls = []
    for i in (1, 2, 3):
        sums[i], dl = K.ce_fused(logits.numpy(), tg.numpy(), False, conf.numpy(), ign.numpy(), 0.6, mc.numpy(), gs[i],
                                 all_pixels=True, img_weight=ratio)
        dls.append(dl)
    out = np.zeros(8, np.float32)
    K.check(lib.svl_semivl_loss(K.ptr(sums), numel, lam, None, K.ptr(mcn), K.ptr(out), None))
    lt = logits.clone().requires_grad_(True)
    lu = O.confidence_weighted_loss(F.cross_entropy(lt, tg, reduction="none"), conf, ign, "pixelratio", 0.6)
    lmc = O.compute_mc_loss(lt, mc, ign, reduce)
    lx = F.cross_entropy(logits, target, ignore_index=255)
    ref = (lx + lu * 0.25 + lu * 0.25 + lu * 0.5) / 2.0 + lmc * 0.25 * lam + lmc * 0.25 * lam + lmc * 0.5 * lam
    assert abs(out[0] - ref.item()) < 1e-5 and abs(out[2] - lu.item()) < 1e-5 and abs(out[5] - lmc.item()) < 1e-6
    (lu * 0.25 / 2.0 + lmc * 0.25 * lam).backward()        # the s1 branch's share of the total loss
    assert np.abs(dls[0] - lt.grad.numpy()).max() < 1e-6 * max(1.0, float(lt.grad.abs().max()))


def test_cpu_maskclip_labels_iou_hist_adamw():
    from oracle import eval_oracle as E
    lib = K.load()
    g = torch.Generator().manual_seed(9)
    B, NC, h, S = 2, 12, 8, 64
    emb = F.normalize(torch.randn(B, 16, h, h, generator=g), dim=1)
    text = F.normalize(torch.randn(NC, 16, generator=g), dim=1)
    dense = F.conv2d(emb, text[:, :, None, None]).contiguous()
    offs = np.array([0, 5, 7, 12], np.int32)
    agg = np.zeros((B, 3, h, h), np.float32)
    K.check(lib.svl_concept_max_f32(K.ptr(dense.numpy()), B, NC, h * h, K.ptr(offs), 3, K.ptr(agg), None))
    ref_agg = torch.stack([dense[:, offs[i]:offs[i + 1]].max(1).values for i in range(3)], 1)
    assert np.array_equal(agg, ref_agg.numpy())
    prob = (100.0 * F.interpolate(ref_agg, size=(S, S), mode="bilinear", align_corners=False)).softmax(1)
    cert, pred = prob.max(1)
    ref = pred.clone()
    ref[cert < 0.9] = 255
    out = np.zeros((B, S, S), np.int64)
    K.check(lib.svl_maskclip_labels(K.ptr(agg), B, 3, h, h, S, S, 100.0, 0.9, None, K.ptr(out), None))
    t2 = prob.topk(2, 1).values
    near = (((cert - 0.9).abs() < 1e-5) | ((t2[:, 0] - t2[:, 1]) < 1e-5)).numpy()
    assert not ((out != ref.numpy()) & ~near).any()
    # intersectionAndUnion
    pr, tg = torch.randint(0, 5, (3000,), generator=g), torch.randint(0, 5, (3000,), generator=g)
    tg[::7] = 255
    hist = np.zeros(15, np.int64)
    K.check(lib.svl_iou_hist_i64(K.ptr(pr.numpy()), K.ptr(tg.numpy()), 3000, 5, 255, K.ptr(hist), None))
    i, u, t = E.intersection_and_union(pr.numpy(), tg.numpy(), 5, 255)
    assert np.array_equal(hist[:5], i) and np.array_equal(hist[5:10] + hist[10:] - hist[:5], u) and np.array_equal(hist[10:], t)
    # AdamW on a two-segment arena
    n = 500
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([dict(params=[ref_p], lr=1e-3, weight_decay=0.01)])
    p, m, v = p0.numpy().copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    off, lr, wd = np.array([0, 200, n], np.int64), np.full(2, 1e-3, np.float32), np.full(2, 0.01, np.float32)
    for step in (1, 2, 3):
        gr = torch.randn(n, generator=g)
        ref_p.grad = gr.clone()
        opt.step()
        K.check(lib.svl_adamw_step(K.ptr(p), K.ptr(gr.numpy()), K.ptr(m), K.ptr(v), K.ptr(off), K.ptr(lr), K.ptr(wd), 2, n,
                                   0.9, 0.999, 1e-8, step, 1.0, None, 0.0, None))
        assert np.abs(p - ref_p.detach().numpy()).max() < 1e-6


def _up_case(N, h, w, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    B = 2
    lg = (torch.randn(B, N, h, w, generator=g) * 2).contiguous()
    lab = torch.randint(0, N, (B, H, W), generator=g)
    conf = torch.rand(B, H, W, generator=g)
    ign = torch.zeros(B, H, W, dtype=torch.int64)
    ign[:, -3:] = 255
    mc = torch.randint(0, N, (B, H, W), generator=g)
    mc[torch.rand(B, H, W, generator=g) < 0.5] = 255
    return lg, lab, conf, ign, mc


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("N,h,w,H,W", [(5, 8, 8, 32, 32), (7, 13, 20, 50, 79), (21, 16, 16, 64, 64), (3, 9, 9, 9, 9)])
def test_cpu_resize_fused_losses_match_interpolate_and_autograd(N, h, w, H, W, align):
    """svl_softmax_max_up_f32 / svl_ce_up_fused_f32 of the CPU backend (the plain-C restatement of F.interpolate -> loss ->
    its backward) against PyTorch itself: vlg_head.py:247 / builder.py:93-97 + semivl.py:232,252,267-323."""
    lib = K.load()
    lg, lab, conf, ign, mc = _up_case(N, h, w, H, W, seed=41)
    B = lg.shape[0]
    assert lib.svl_ce_up_num_blocks(B, N, h, w, H, W, int(align)) == B
    up = F.interpolate(lg, size=(H, W), mode="bilinear", align_corners=align)
    cf, lb = np.zeros((B, H, W), np.float32), np.zeros((B, H, W), np.int64)
    K.check(lib.svl_softmax_max_up_f32(K.ptr(lg.numpy()), B, N, h, w, H, W, int(align), K.ptr(cf), K.ptr(lb), None))
    rc, rl = up.softmax(1).max(1)
    top2 = up.topk(2, dim=1).values
    tie = ((top2[:, 0] - top2[:, 1]) < 1e-5).numpy()
    assert np.array_equal(lb[~tie], rl.numpy()[~tie]) and np.abs(cf - rc.numpy()).max() < 5e-6
    lgr = lg.clone().requires_grad_(True)
    upr = F.interpolate(lgr, size=(H, W), mode="bilinear", align_corners=align)
    valid = ign != 255
    lu = (F.cross_entropy(upr, lab, reduction="none") * ((conf >= 0.7) & valid)).sum() / valid.sum().item()
    lm = F.cross_entropy(upr, mc, ignore_index=255, reduction="none").sum() / ign.numel()
    (g,) = torch.autograd.grad(0.125 * lu + 0.03 * lm, lgr)
    gs = np.array([0.125 / valid.sum().item(), 0.03 / ign.numel()], np.float32)
    sums, dl = K.ce_up_fused(lg.numpy(), H, W, align, lab.numpy(), False, conf.numpy(), ign.numpy(), 0.7, mc.numpy(), gs)
    assert abs(sums[0] / sums[3] - lu.item()) < 1e-5 and abs(sums[1] / ign.numel() - lm.item()) < 1e-5
    assert sums[3] == valid.sum().item() and abs(sums[2] - (conf * valid).sum().item()) < 1e-2
    assert np.abs(dl - g.numpy()).max() < 2e-9 + 2e-4 * np.abs(g.numpy()).max()
    # geometries the entry points refuse (callers resize and use svl_ce_fused_f32)
    assert lib.svl_ce_up_num_blocks(B, N, 8, 8, 64, 64, int(align)) == -1 and lib.svl_ce_up_num_blocks(B, N, 16, 16, 8, 8, 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("align", [False, True])
def test_hip_resize_fused_losses_agree_with_the_cpu_backend(dev, align):
    """The HIP kernels (tiled, gather-form backward) against the plain-C restatement (per-pixel scatter) of the same entry
    points: labels identical off ties, sums / confidences / head-resolution gradients to rounding."""
    from semivl_amd import ops
    lib = K.load()
    for (N, h, w, H, W) in ((21, 32, 32, 128, 128), (19, 26, 20, 101, 79), (150, 8, 8, 32, 32)):
        lg, lab, conf, ign, mc = _up_case(N, h, w, H, W, seed=43)
        B = lg.shape[0]
        cf, lb = np.zeros((B, H, W), np.float32), np.zeros((B, H, W), np.int64)
        K.check(lib.svl_softmax_max_up_f32(K.ptr(lg.numpy()), B, N, h, w, H, W, int(align), K.ptr(cf), K.ptr(lb), None))
        gcf, glb = ops.softmax_max_up(lg.to(dev), H, W, align)
        up = F.interpolate(lg, size=(H, W), mode="bilinear", align_corners=align)
        top2 = up.topk(2, dim=1).values
        tie = ((top2[:, 0] - top2[:, 1]) < 1e-5).numpy()
        assert np.array_equal(glb.cpu().numpy()[~tie], lb[~tie]) and np.abs(gcf.cpu().numpy() - cf).max() < 5e-6
        gs = torch.tensor([0.37, 0.11])
        sums, dl = K.ce_up_fused(lg.numpy(), H, W, align, lab.numpy(), False, conf.numpy(), ign.numpy(), 0.6, mc.numpy(), gs.numpy())
        gdl = torch.empty_like(lg, device=dev)
        gsums = ops.ce_up_fused(lg.to(dev), H, W, align, lab.to(dev), False, conf=conf.to(dev), ign=ign.to(dev), conf_thresh=0.6,
                                mc=mc.to(dev), dlogits=gdl, gscale=gs.to(dev))
        assert np.abs(gsums.cpu().numpy() - sums).max() < 1e-3 * max(1.0, np.abs(sums).max()) and gsums[3].item() == sums[3]
        assert np.abs(gdl.cpu().numpy() - dl).max() < 1e-5 * max(1.0, np.abs(dl).max())


@pytest.mark.gpu
def test_hip_library_agrees_with_cpu_backend_on_the_same_inputs(dev):
    """Two implementations of the same entry points (HIP kernels vs plain C): integer outputs identical, fp32 outputs to
    rounding."""
    from semivl_amd import ops
    lib = K.load()
    logits, target, conf, ign, mc = _case(B=2, N=21, H=64, W=48, seed=11)
    B, N, H, W = logits.shape
    cf, lab = np.zeros((B, H, W), np.float32), np.zeros((B, H, W), np.int64)
    K.check(lib.svl_softmax_max_f32(K.ptr(logits.numpy()), B, N, H * W, K.ptr(cf), K.ptr(lab), None))
    gcf, glab = ops.softmax_max(logits.to(dev))
    assert np.array_equal(glab.cpu().numpy(), lab) and np.abs(gcf.cpu().numpy() - cf).max() < 1e-6
    gs = torch.tensor([0.37, 0.11])
    tg = target.clamp(max=20)
    sums, dl = K.ce_fused(logits.numpy(), tg.numpy(), False, conf.numpy(), ign.numpy(), 0.6, mc.numpy(), gs.numpy())
    gdl = torch.empty_like(logits, device=dev)
    gsums = ops.ce_fused(logits.to(dev), tg.to(dev), False, conf=conf.to(dev), ign=ign.to(dev), conf_thresh=0.6,
                         mc=mc.to(dev), dlogits=gdl, gscale=gs.to(dev))
    assert np.abs(gsums.cpu().numpy() - sums).max() < 1e-3 * max(1.0, np.abs(sums).max()) and gsums[3].item() == sums[3]
    assert np.abs(gdl.cpu().numpy() - dl).max() < 1e-6
    pr, tt = torch.randint(0, 21, (B, H, W)), target
    hist = np.zeros(63, np.int64)
    K.check(lib.svl_iou_hist_i64(K.ptr(pr.numpy()), K.ptr(tt.numpy()), pr.numel(), 21, 255, K.ptr(hist), None))
    gh = ops.zeros(63, dtype=torch.int64, device=dev)
    ops.iou_hist(pr.to(dev), tt.to(dev), 21, 255, gh)
    assert np.array_equal(gh.cpu().numpy(), hist)
