"""The SemiVL training step on MI355X: loss helpers, the two-branch step, fused AdamW and the gradient all-reduce.

Mirrors (same names / argument meaning):
  utils/train_utils.py:19-49   cutmix_img_, cutmix_mask, confidence_weighted_loss
  semivl.py:52-58              compute_mc_loss
  semivl.py:223-345            the loop body  -> semivl_train_step()
  semivl.py:123-125,339-345    mmcv param-wise AdamW + poly LR -> FusedAdamW
  semivl.py:139-140            DistributedDataParallel -> GradAllReducer (RCCL all-reduce of the flat grad arena)
"""

import contextlib
import os
import time

import torch
import torch.distributed as dist

from . import ops


# ------------------------------------------------------------------------------------------------ reference-named helpers
def cutmix_img_(img, img_mix, cutmix_box):
    """In place: img[box == 1] = img_mix[box == 1] (train_utils.py:19-21), one select kernel, no host sync."""
    ops.cutmix_f32(img, img_mix, cutmix_box, out=img)


def cutmix_mask(mask, mask_mix, cutmix_box):
    """train_utils.py:24-27 for int64 label-like maps and fp32 confidence maps."""
    if mask.dtype == torch.int64:
        return ops.cutmix_i64(mask, mask_mix, cutmix_box)
    return ops.cutmix_f32(mask, mask_mix, cutmix_box)


def confidence_weighted_loss(loss, conf_map, ignore_mask, cfg):
    """train_utils.py:30-49 on an autograd-tracked per-pixel loss map.  API-compatibility helper for callers that
    compute the per-pixel CE themselves; `semivl_train_step` uses the fused CE kernel instead."""
    assert loss.dim() == 3 and conf_map.dim() == 3 and ignore_mask.dim() == 3
    valid_mask = ignore_mask != 255
    sum_pixels = dict(dim=(1, 2), keepdim=True)
    if cfg["conf_mode"] == "pixelwise":
        loss = loss * ((conf_map >= cfg["conf_thresh"]) & valid_mask)
        return loss.sum() / valid_mask.sum().item()
    if cfg["conf_mode"] == "pixelratio":
        r = ((conf_map >= cfg["conf_thresh"]) & valid_mask).sum(**sum_pixels) / valid_mask.sum(**sum_pixels)
        return (loss * r).sum() / valid_mask.sum().item()
    if cfg["conf_mode"] == "pixelavg":
        avg_conf = (conf_map * valid_mask).sum(**sum_pixels) / valid_mask.sum(**sum_pixels)
        return (loss.sum() * avg_conf).sum() / valid_mask.sum().item()
    raise ValueError(cfg["conf_mode"])


def compute_mc_loss(pred, mask, ign, mcc_loss_reduce="mean_all", criterion_mc=None):
    """semivl.py:52-58 (MaskCLIP guidance loss).  The reference reads `criterion_mc` / `mcc_loss_reduce` from module
    globals set in main() (semivl.py:158-164); here they are arguments with the same meaning: reduce 'mean' ->
    nn.CrossEntropyLoss(ignore_index=255), 'mean_valid' / 'mean_all' -> the per-pixel map (reduction='none') summed and
    divided by the number of non-ignored pixels / by all pixels.  API-compatibility helper: an autograd-tracked scalar for
    callers that write the loop body themselves; `semivl_train_step` uses the fused CE kernel instead."""
    if criterion_mc is None:
        criterion_mc = (torch.nn.CrossEntropyLoss(ignore_index=255) if mcc_loss_reduce == "mean" else
                        torch.nn.CrossEntropyLoss(ignore_index=255, reduction="none"))
    l_mc = criterion_mc(pred, mask)
    if mcc_loss_reduce == "mean_valid":
        l_mc = l_mc.sum() / (ign != 255).sum()
    elif mcc_loss_reduce == "mean_all":
        l_mc = l_mc.sum() / ign.numel()
    elif mcc_loss_reduce != "mean":
        raise ValueError(mcc_loss_reduce)
    return l_mc


def _cat2(a, b):
    """torch.cat((a, b)) along dim 0 with the library's copy kernel."""
    out = ops.empty(a.shape[0] + b.shape[0], *a.shape[1:], device=a.device)
    flat = out.view(-1)
    ops.eltwise(4, a.contiguous().view(-1), None, out=flat[:a.numel()])
    ops.eltwise(4, b.contiguous().view(-1), None, out=flat[a.numel():])
    return out


_SIDE = {}     # device -> second stream for the gradient-free passes of the step


def create_step_streams(dev):
    """Create every stream a training step uses on `dev` NOW (the gradient-free passes' second stream, the weight-gradient
    stream when it is enabled, and the library's helper streams of both and of the current stream) instead of lazily inside
    the first step.  The HIP runtime assigns streams to its GPU_MAX_HW_QUEUES hardware queues in creation order and
    serialises streams that share a queue: creating the step's streams first, and only then a communicator's,
    makes that map the same in every process -- GradAllReducer calls this before it creates its communication stream."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        return []
    from . import lib as L
    out = [torch.cuda.current_stream(dev)]
    if dev not in _SIDE:
        _SIDE[dev] = torch.cuda.Stream(dev)
    out.append(_SIDE[dev])
    if ops.WGRAD_STREAM:
        out.append(ops._wg_stream(dev))
    with torch.cuda.device(dev):
        for s_ in out:
            L.check(L.load().svl_stream_prepare(s_.cuda_stream), "svl_stream_prepare")
    return out


def step_helper_streams(streams):
    """The library's helper streams of `streams` (ragged-row GEMM parts, the attention's leftover-row kernels run there) as
    torch ExternalStreams: the communication stream must not share a hardware queue with them either."""
    import ctypes
    from . import lib as L
    out = []
    for s_ in streams:
        h_ = ctypes.c_void_p()
        with torch.cuda.device(s_.device):
            L.check(L.load().svl_stream_helper(s_.cuda_stream, ctypes.byref(h_)), "svl_stream_helper")
        if h_.value:
            out.append(torch.cuda.ExternalStream(h_.value, device=s_.device))
    return out

LOSS_NAMES = ("loss", "loss_x", "loss_s1", "loss_s2", "loss_fp", "loss_mc_s1", "loss_mc_s2", "loss_mc_fp")


def semivl_train_step(model, batch, iters, total_iters, cfg, optimizer=None, reducer=None, fp_masks=None,
                      return_aux=False):
    """One iteration of semivl.py:223-328 (see _semivl_train_step for the body).  Per-step overrides taken from `cfg`
    (`wgrad_stream`, `overlap_streams`, `head_remat`, `head_chunk_class_images`) and the decode head's per-step memory
    plan are undone when the step ends -- also when it raises -- so that a later grad-enabled decode outside a training
    step (validation with gradients, a test, another batch size) never inherits a stale decision."""
    head = getattr(model, "decode_head", None)
    keep_wg = ops.WGRAD_STREAM
    keep_head = {a_: getattr(head, a_) for a_ in ("remat", "chunk_class_images") if head is not None and hasattr(head, a_)}
    try:
        return _semivl_train_step(model, batch, iters, total_iters, cfg, optimizer, reducer, fp_masks, return_aux)
    finally:
        ops.WGRAD_STREAM = keep_wg
        if head is not None:
            head._bwd_ranges = None
            # (a record for tests / logs: the level this step decided on; getattr: a step that raised before the memory plan
            # was set up -- or a decode head that is not a VLGHead -- must not replace the original exception)
            head.last_remat_step = getattr(head, "_remat_step", None)
            head._remat_step = None
            head._live_class_images = None
            for a_, v_ in keep_head.items():
                setattr(head, a_, v_)


def _semivl_train_step(model, batch, iters, total_iters, cfg, optimizer=None, reducer=None, fp_masks=None,
                       return_aux=False):
    """One iteration of semivl.py:223-328 (method 'semivl', CELoss(ignore 255) / CELoss; conf_mode 'pixelwise' /
    'pixelavg' / 'pixelratio', train_utils.py:30-49; mcc_loss_reduce 'mean_all' / 'mean_valid' / 'mean', semivl.py:52-58).  `batch`: the 12 step tensors on the GPU.  No host syncs: the
    returned `losses` is a device float[8] (LOSS_NAMES order).
    """
    conf_mode = cfg.get("conf_mode", "pixelwise")
    mcc_reduce = cfg.get("mcc_loss_reduce", "mean_all")
    if conf_mode not in ("pixelwise", "pixelavg", "pixelratio"):
        raise ValueError(conf_mode)                      # train_utils.py:47-48
    if mcc_reduce not in ("mean_all", "mean_valid", "mean"):
        raise ValueError(mcc_reduce)                     # semivl.py:161-162
    pixelavg = conf_mode == "pixelavg"
    whole_map = conf_mode in ("pixelavg", "pixelratio")  # both weight the WHOLE CE map, not the confident valid pixels
    lam_cfg = cfg.get("maskclip_consistency_lambda", [0.1, 0])
    if isinstance(lam_cfg, (list, tuple)):
        prog = iters / total_iters
        lam = lam_cfg[0] * (1 - prog) + lam_cfg[1] * prog
    else:
        lam = lam_cfg
    b = batch
    img_x, mask_x = b["img_x"], b["mask_x"]
    img_w, img_s1, img_s2 = b["img_w"], b["img_s1"], b["img_s2"]
    ign, mix1, mix2 = b["ignore_mask"], b["mix1"], b["mix2"]
    ign_o = b["ignore_mask_other"]
    B = img_x.shape[0]
    dev = img_x.device
    # CutMix images (in place, like the reference)
    cutmix_img_(img_s1, b["img_s1_other"], mix1)
    cutmix_img_(img_s2, b["img_s2_other"], mix2)
    if getattr(model, "decode_head", None) is not None:
        model.decode_head._bwd_ranges = None
    if reducer is not None:
        reducer.begin()
    # the weight-gradient stream is a process-wide switch (ops.WGRAD_STREAM: environment, multi_rank_defaults() or the
    # caller); the step only overrides it when its cfg says so explicitly
    if "wgrad_stream" in cfg:
        ops.WGRAD_STREAM = bool(cfg["wgrad_stream"])
    elif not cfg.get("overlap_streams", True):
        ops.WGRAD_STREAM = False
    # pseudo labels + MaskCLIP guidance (model.eval(): the side encoder's BatchNorm uses its running statistics here,
    # semivl.py:228-244; nothing else on the path depends on the mode).  Both passes are gradient-free and independent of
    # the two student forwards below: they are enqueued on a second stream (event-forked from / joined back into the
    # caller's stream), so their kernels fill the partial last rounds of the student's grids instead of queueing behind
    # them.  With a conv_encoder only the frozen-CLIP guidance moves over: the pseudo-label pass reads, in eval mode, the
    # BatchNorm running statistics that the train-mode forwards update, so it stays in program order on the main stream.
    side = None
    if img_x.is_cuda and cfg.get("overlap_streams", True) and not os.environ.get("SVL_NO_SIDE_STREAM"):
        side = _SIDE.get(dev)
        if side is None:
            side = _SIDE[dev] = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev) if img_x.is_cuda else None
    pl_side = side if getattr(model, "conv_encoder", None) is None else None
    # The logits stay at the head's resolution (round 5): the bilinear resize to the crop (vlg_head.py:247, builder.py:93-97)
    # is evaluated inside the softmax-max / cross-entropy kernels and the gradient comes back at the head's resolution --
    # the [B, N, H, W] tensors and the two resize passes do not exist.  `up` = (H, W, align_corners) or None (models without
    # the option, geometries the kernels do not take, cfg / SVL_NO_UP_LOSS: the resized tensors as before).
    up = None
    if img_x.is_cuda and cfg.get("fuse_upsample_loss", ops.UP_LOSS) and hasattr(model, "head_res_size"):
        hs = model.head_res_size(tuple(img_x.shape[2:]))
        if hs is not None and ops.ce_up_ok(B, model.num_classes, hs[0], hs[1], img_x.shape[2], img_x.shape[3],
                                           model.align_corners):
            up = (int(img_x.shape[2]), int(img_x.shape[3]), bool(model.align_corners))
    fwd_kw = dict(head_res=True) if up is not None else {}

    def smax(pred):
        return ops.softmax_max_up(pred, *up) if up is not None else ops.softmax_max(pred)

    def full_res(pred):
        """(return_aux only) the resized logits the fused kernels never write"""
        if up is None:
            return pred
        pred = pred.contiguous()
        return ops.bilinear_planes_fwd(pred, pred.shape[2], pred.shape[3], up[2], up[0], up[1])

    model.eval()
    if side is not None:
        side.wait_stream(main)
    with torch.no_grad(), (torch.cuda.stream(pl_side) if pl_side is not None else contextlib.nullcontext()):
        pred_w_other = model(b["img_w_other"], **fwd_kw)
        conf_w_other, mask_w_other = smax(pred_w_other)
        if not return_aux:
            del pred_w_other
    with torch.no_grad(), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        mclip_all = model.forward_maskclip(_cat2(img_w, b["img_w_other"]), cfg.get("mcc_conf_thresh", 0.9),
                                           ignore_mask=_cat2i(ign, ign_o))
        mclip, mclip_other = mclip_all[:B], mclip_all[B:]
    model.train()          # semivl.py:246: unconditionally back to train mode
    # predictions
    # the feature-perturbed copy of the labeled half (pred_x_fp) is never read by the step (semivl.py:247): only the
    # unlabeled half is perturbed and decoded.  The batch is ordered [w, x] (the reference: [x, w]; every op is
    # per-sample or a permutation-invariant batch statistic) so that the decoded batch is [w, x, w_fp]: the two
    # gradient-carrying thirds are CONTIGUOUS and the head's backward runs once on 2B samples (full grid rounds)
    # instead of twice on B.
    if fp_masks is not None:   # injected masks come in the reference's [x, w] row order
        fp_masks = [torch.cat((m_[B:2 * B], m_[:B])) for m_ in fp_masks]
    # pred_w is detached (semivl.py:251) -> exactly-zero dlogits: the head decodes those samples without keeping
    # activations and its backward skips them (results identical).  Memory plan: activations of the remaining chunks are
    # kept while the allocation stays under `act_mem_fraction` of the device, the others are recomputed in backward.
    head = getattr(model, "decode_head", None)
    if head is not None:
        head._bwd_ranges = {3 * B: [(B, 3 * B)]}
        # both grad-carrying decodes of the step ([x, w_fp] here, [s1, s2] below) keep activations until backward: the
        # head decides ONCE per step whether they fit or are re-materialised (model/vlg_head.py::_remat_decision)
        head._live_class_images = 4 * B * head.num_classes
        head._remat_step = {}
        if "head_remat" in cfg:
            head.remat = cfg["head_remat"]
        if cfg.get("head_chunk_class_images"):
            head.chunk_class_images = int(cfg["head_chunk_class_images"])
        frac = cfg.get("act_mem_fraction", 0.70)
        head.act_limit_bytes = (None if frac is None or not img_x.is_cuda else
                                int(frac * torch.cuda.get_device_properties(dev).total_memory))
    preds4 = model(_cat2(img_w, img_x), need_fp=True, fp_masks=fp_masks, split_fp=False, fp_range=(0, B), **fwd_kw)
    preds_s = model(_cat2(img_s1, img_s2), **fwd_kw)                                       # [s1, s2]
    pred_w, pred_x, pred_w_fp = preds4[:B], preds4[B:2 * B], preds4[2 * B:]
    pred_s1, pred_s2 = preds_s[:B], preds_s[B:]
    conf_w, mask_w = smax(pred_w.detach())
    if side is not None:        # join: the label maps of the side stream are consumed from here on
        main.wait_stream(side)
        for t_ in ((conf_w_other, mask_w_other, mclip_all) if pl_side is not None else (mclip_all,)):
            t_.record_stream(main)
    # CutMix labels
    mw1, mw2 = cutmix_mask(mask_w, mask_w_other, mix1), cutmix_mask(mask_w, mask_w_other, mix2)
    cw1, cw2 = cutmix_mask(conf_w, conf_w_other, mix1), cutmix_mask(conf_w, conf_w_other, mix2)
    ig1, ig2 = cutmix_mask(ign, ign_o, mix1), cutmix_mask(ign, ign_o, mix2)
    mc1, mc2 = cutmix_mask(mclip, mclip_other, mix1), cutmix_mask(mclip, mclip_other, mix2)
    # normalisers -> per-branch gradient scales, on the device
    counts = ops.zeros(4, dtype=torch.int64, device=dev)
    for i, m_ in enumerate((mask_x, ig1, ig2, ign)):
        ops.count_valid(m_, counts[i:i + 1])
    numel_u = float(ign.numel())
    gscale = ops.empty(4, 2, device=dev)
    factors = None
    if pixelavg:  # train_utils.py:43-46: (sum of the WHOLE CE map) * sum_b mean_valid(conf_b) / #valid
        factors = ops.empty(3, dtype=torch.float64, device=dev)
        for i, (c_, g_) in enumerate(((cw1, ig1), (cw2, ig2), (conf_w, ign))):
            ops.conf_avg_factor(c_, g_, factors[i:i + 1])
    ratios = (None, None, None)
    if conf_mode == "pixelratio":   # train_utils.py:39-42: image b's CE map times its share of confident valid pixels
        ratios = tuple(ops.conf_ratio(c_, g_, cfg["conf_thresh"]) for c_, g_ in ((cw1, ig1), (cw2, ig2), (conf_w, ign)))
    mc_counts = None                # semivl.py:52-58: the guidance loss's normaliser
    if mcc_reduce == "mean_valid":
        mc_counts = counts[1:]
    elif mcc_reduce == "mean":      # nn.CrossEntropyLoss(ignore_index=255): mean over the labelled guidance pixels
        mc_counts = ops.zeros(3, dtype=torch.int64, device=dev)
        for i, m_ in enumerate((mc1, mc2, mclip)):
            ops.count_valid(m_, mc_counts[i:i + 1])
    ops.semivl_gscale(counts, numel_u, lam, gscale, factors, mc_counts)
    # fused CE forward + backward per branch
    thr = cfg["conf_thresh"]
    dl4 = torch.empty_like(preds4)
    dls = torch.empty_like(preds_s)
    ops.fill(dl4[:B], 0.0)  # pred_w is detached (semivl.py:251)
    sums = ops.empty(4, 4, dtype=torch.float64, device=dev)
    def ce(pred, *a_, **kw):
        if up is not None:
            return ops.ce_up_fused(pred, up[0], up[1], up[2], *a_, **kw)
        return ops.ce_fused(pred, *a_, **kw)

    ce(pred_x.detach(), mask_x, True, dlogits=dl4[B:2 * B], gscale=gscale[0], sums_out=sums[0])
    ce(pred_s1.detach(), mw1, False, conf=cw1, ign=ig1, conf_thresh=thr, mc=mc1, dlogits=dls[:B],
       gscale=gscale[1], sums_out=sums[1], all_pixels=whole_map, img_weight=ratios[0])
    ce(pred_s2.detach(), mw2, False, conf=cw2, ign=ig2, conf_thresh=thr, mc=mc2, dlogits=dls[B:],
       gscale=gscale[2], sums_out=sums[2], all_pixels=whole_map, img_weight=ratios[1])
    ce(pred_w_fp.detach(), mask_w, False, conf=conf_w, ign=ign, conf_thresh=thr, mc=mclip,
       dlogits=dl4[2 * B:], gscale=gscale[3], sums_out=sums[3], all_pixels=whole_map, img_weight=ratios[2])
    losses = ops.empty(8, device=dev)
    ops.semivl_loss(sums, numel_u, lam, losses, factors, mc_counts)
    # backward (+ all-reduce) + optimizer
    if cfg.get("step_barrier", False) and dist.is_initialized() and dist.get_world_size() > 1:
        # semivl.py:325: the reference synchronises all ranks before every backward.  It orders nothing the gradient
        # all-reduce does not order already, so it is off by default; the flag restores the reference's
        # lock-step pacing (useful when per-rank logging / timing is compared against the reference's).
        dist.barrier()
    if optimizer is not None:
        optimizer.zero_grad()
    try:
        torch.autograd.backward([preds4, preds_s], [dl4, dls])
    finally:
        if head is not None:
            head._bwd_ranges = None
    if reducer is not None:
        reducer.finish()    # folds autograd-delivered grads, flushes / waits for the overlapped all-reduce buckets
    if optimizer is not None:
        optimizer.step()
        optimizer.poly_lr(iters, cfg.get("scheduler_max_iters", total_iters), warmup_iters=cfg.get("warmup_iters", 0),
                          warmup_ratio=cfg.get("warmup_ratio", 1e-6))
    if return_aux:
        aux = dict(mask_w=mask_w, mask_w_other=mask_w_other, mclip=mclip, mclip_other=mclip_other, conf_w=conf_w,
                   pred_x=full_res(pred_x.detach()), pred_s1=full_res(pred_s1.detach()), pred_w=full_res(pred_w.detach()),
                   pred_w_other=full_res(pred_w_other), dl4=dl4, dls=dls, upsample_in_loss=up is not None)
        return losses, aux
    return losses


def _cat2i(a, b):
    out = ops.empty(a.shape[0] + b.shape[0], *a.shape[1:], dtype=torch.int64, device=a.device)
    # int64 maps: copy as pairs of fp32 words with the fp32 copy kernel
    fa, fb = a.contiguous().view(torch.float32), b.contiguous().view(torch.float32)
    fo = out.view(torch.float32).view(-1)
    ops.eltwise(4, fa.view(-1), None, out=fo[:fa.numel()])
    ops.eltwise(4, fb.view(-1), None, out=fo[fa.numel():])
    return out


# ------------------------------------------------------------------------------------------------ optimizer
def mmcv_param_groups(named_params, lr, weight_decay, custom_keys):
    """mmcv 1.4.4 DefaultOptimizerConstructor semantics (recalled): one group per parameter; custom keys
    sorted alphabetically then longest-first; the FIRST key contained in the parameter name sets lr_mult/decay_mult."""
    keys = sorted(sorted(custom_keys.keys()), key=len, reverse=True)
    out = []
    for name, p in named_params:
        g = dict(name=name, param=p, lr=lr, weight_decay=weight_decay)
        for k in keys:
            if k in name:
                g["lr"] = lr * custom_keys[k].get("lr_mult", 1.0)
                g["weight_decay"] = weight_decay * custom_keys[k].get("decay_mult", 1.0)
                break
        out.append(g)
    return out


class FusedAdamW:
    """torch.optim.AdamW semantics over ONE flat fp32 arena: parameters, gradients (`main_grad` views the model's
    backward writes into), exp_avg, exp_avg_sq; one svl_adamw_step launch per step (28 B/param of HBM traffic).

    Only parameters that can receive a gradient are placed in the arena: `clip_encoder.*` (registered with
    requires_grad=True in the reference, never given a grad — SURVEY App. E.2) and frozen backbone tensors are left
    untouched, which is also what torch's AdamW does for params whose .grad is None.
    """

    def __init__(self, model, optimizer_cfg, ema_decay=None):
        assert optimizer_cfg.get("type", "AdamW") == "AdamW"
        self.lr, self.wd = optimizer_cfg["lr"], optimizer_cfg.get("weight_decay", 0.01)
        self.betas, self.eps = optimizer_cfg.get("betas", (0.9, 0.999)), optimizer_cfg.get("eps", 1e-8)
        ck = optimizer_cfg.get("paramwise_cfg", {}).get("custom_keys", {})
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and not n.startswith("clip_encoder.")]
        self.groups = mmcv_param_groups(named, self.lr, self.wd, ck)
        arena_index = {id(p): i for i, (_, p) in enumerate(named)}
        self.all_params = [(n, arena_index.get(id(p))) for n, p in model.named_parameters()]   # (name, arena slot | None)
        dev = named[0][1].device
        sizes = [g["param"].numel() for g in self.groups]
        # 16-byte aligned segments
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += (s + 3) // 4 * 4
        self.total = o
        self.p = ops.zeros(self.total, device=dev)
        self.g = ops.zeros(self.total, device=dev)
        self.m = ops.zeros(self.total, device=dev)
        self.v = ops.zeros(self.total, device=dev)
        self.ema = None
        for g_, off, s in zip(self.groups, offs, sizes):
            prm = g_["param"]
            view = self.p[off:off + s].view(prm.shape)
            ops.eltwise(4, prm.data.contiguous().view(-1), None, out=view.view(-1))
            prm.data = view
            prm.main_grad = self.g[off:off + s].view(prm.shape)
            g_["initial_lr"] = g_["lr"]
        if ema_decay is not None:
            self.ema = self.p.clone()
        self.ema_decay = ema_decay or 0.0
        self.seg_off = torch.tensor(offs + [self.total], dtype=torch.int64, device=dev)
        self.seg_wd = torch.tensor([g_["weight_decay"] for g_ in self.groups], dtype=torch.float32, device=dev)
        self._lr_host = torch.tensor([g_["lr"] for g_ in self.groups], dtype=torch.float32).pin_memory() \
            if torch.cuda.is_available() else torch.tensor([g_["lr"] for g_ in self.groups], dtype=torch.float32)
        self.seg_lr = self._lr_host.to(dev)
        self._lr_evt = None
        self.step_count = 0
        self.grad_scale = 1.0
        self._lr_factor = 1.0     # the schedule's current factor (poly_lr): lr of the groups outside the arena

    @property
    def param_groups(self):
        return self.groups

    def zero_grad(self):
        ops.fill(self.g, 0.0)

    def _fold_autograd_grads(self):
        """Gradients that reached a parameter through torch autograd instead of the main_grad sink (e.g. pos_embed
        behind its bicubic resize at 801x801) are added to the arena."""
        for g_ in self.groups:
            prm = g_["param"]
            if prm.grad is not None:
                ops.add(prm.main_grad.view(-1), prm.grad.contiguous().view(-1), out=prm.main_grad.view(-1))
                prm.grad = None

    def step(self):
        self._fold_autograd_grads()
        self.step_count += 1
        ops.adamw_step(self.p, self.g, self.m, self.v, self.seg_off, self.seg_lr, self.seg_wd, len(self.groups),
                       self.betas[0], self.betas[1], self.eps, self.step_count, self.grad_scale, self.ema,
                       self.ema_decay)
        ops.weights_changed()    # cached bf16 planes of the trainable weights are stale now (ops.weight_planes)

    def state_dict(self):
        """The layout of the reference's checkpoint entry (`semivl.py:428` stores `optimizer.state_dict()` of a
        torch.optim.AdamW built by mmcv's DefaultOptimizerConstructor): ONE param group per tensor of
        `model.named_parameters()`, in that order, frozen tensors and `clip_encoder.*` included (mmcv lists them with the
        base lr / weight decay; they never receive a gradient, so they have no `state` entry); state[i] = step / exp_avg /
        exp_avg_sq for the tensors of the arena.  Index-compatible with the reference in both directions
        (tests/test_model_gpu.py::test_optimizer_state_dict_is_index_compatible).  `names` (all parameters, same order) is
        stored in addition and verified on load."""
        state, groups = {}, []
        off = self.seg_off.tolist()
        for j, (name, ai) in enumerate(self.all_params):
            if ai is None:      # mmcv lists them with the base lr; semivl.py:124-125 gives EVERY group an initial_lr and
                # :341-345 re-schedules every group from it, so a reference-style loop can load this dict as it is
                groups.append(dict(lr=self.lr * self._lr_factor, initial_lr=self.lr, weight_decay=self.wd,
                                   betas=tuple(self.betas), eps=self.eps, amsgrad=False, params=[j]))
                continue
            g_ = self.groups[ai]
            shp, n = g_["param"].shape, g_["param"].numel()
            if self.step_count > 0:
                state[j] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.m[off[ai]:off[ai] + n].view(shp).detach().cpu().clone(),
                                exp_avg_sq=self.v[off[ai]:off[ai] + n].view(shp).detach().cpu().clone())
            groups.append(dict(lr=g_["lr"], initial_lr=g_["initial_lr"], weight_decay=g_["weight_decay"],
                               betas=tuple(self.betas), eps=self.eps, amsgrad=False, params=[j]))
        return dict(state=state, param_groups=groups, names=[n for n, _ in self.all_params])

    def load_state_dict(self, sd):
        """Accepts the reference layout (one group per model parameter, above) and the compact round-1/2 layout of this
        package (one group per arena tensor)."""
        pg = sd["param_groups"]
        if len(pg) == len(self.all_params):
            index = [(j, ai) for j, (_, ai) in enumerate(self.all_params) if ai is not None]
            mine = [n for n, _ in self.all_params]
        else:
            assert len(pg) == len(self.groups), "optimizer state does not match this model"
            index = [(i, i) for i in range(len(self.groups))]
            mine = [g_.get("name", "") for g_ in self.groups]
        if "names" in sd:
            assert list(sd["names"]) == mine, "optimizer state was saved for different parameters: %s" % (
                sorted(set(sd["names"]) ^ set(mine))[:6],)
        off = self.seg_off.tolist()
        steps = set()
        for j, ai in index:
            g_, sg = self.groups[ai], pg[j]
            assert [int(k) for k in sg["params"]] == [j], "one parameter per group expected (mmcv constructor layout)"
            g_["lr"], g_["initial_lr"] = sg["lr"], sg.get("initial_lr", g_["initial_lr"])
            self._lr_host[ai] = g_["lr"]
            st = sd["state"].get(j)
            if st is not None:
                n = g_["param"].numel()
                assert tuple(st["exp_avg"].shape) == tuple(g_["param"].shape), (g_.get("name"), st["exp_avg"].shape)
                self.m[off[ai]:off[ai] + n].copy_(st["exp_avg"].reshape(-1))
                self.v[off[ai]:off[ai] + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
        if len(pg) == len(self.all_params):
            for j, (_, ai) in enumerate(self.all_params):
                if ai is None and pg[j].get("initial_lr"):
                    self._lr_factor = pg[j]["lr"] / pg[j]["initial_lr"]
                    break
        stray = set(sd["state"]) - {j for j, _ in index}
        assert not stray, "state for parameters this model never trains: %s" % sorted(stray)[:6]
        assert len(steps) <= 1, "per-tensor step counts differ"
        self.step_count = steps.pop() if steps else 0
        self.seg_lr.copy_(self._lr_host)

    def poly_lr(self, iters, max_iters, power=0.9, warmup_iters=0, warmup_ratio=1e-6):
        """semivl.py:339-345: applied after the step, for the next one; linear warm-up while iters < warmup_iters
        (semivl.py:339-342: lr = initial_lr * (1 - (1 - iters / warmup_iters) * (1 - warmup_ratio)))."""
        if iters < warmup_iters:
            f = 1 - (1 - iters / warmup_iters) * (1 - warmup_ratio)
        else:
            f = (1 - iters / max_iters) ** power
        # the pinned staging buffer may still be the source of the previous call's queued copy: wait for THAT copy
        # (issued a whole step ago, so this never stalls in the training loop) before overwriting it
        if self._lr_evt is not None:
            self._lr_evt.synchronize()
        self._lr_factor = f
        for i, g_ in enumerate(self.groups):
            g_["lr"] = g_["initial_lr"] * f
            self._lr_host[i] = g_["lr"]
        self.seg_lr.copy_(self._lr_host, non_blocking=True)
        if self.seg_lr.is_cuda:
            self._lr_evt = torch.cuda.Event()
            self._lr_evt.record()


def build_optimizer(model, optimizer_cfg):
    return FusedAdamW(model, optimizer_cfg)


# ------------------------------------------------------------------------------------------------ data parallel
class GradAllReducer:
    """Data-parallel gradient mean over the flat grad arena, overlapped with backward (replaces DDP's reducer,
    semivl.py:139-140: the reference's all-reduce buckets fire from autograd hooks inside `loss.backward()`, :327).

    One process per GPU, `torch.distributed` backend 'nccl' (= RCCL over xGMI on ROCm; 'gloo' for the CPU tests).
    The arena holds only the 31.4 M live gradients (125 MB) -- the reference's DDP also reduces the 86.8 M never-updated
    clip_encoder gradients.  It is cut at parameter boundaries into contiguous buckets of ~`bucket_mb`.
    The backward regions announce contributions through `gradsync.expect / ready`; a bucket whose parameters have all
    received every expected contribution (two per step: the [w, x, w_fp] graph and the [s1, s2] graph both reach every
    trainable tensor) is all-reduced (SUM) at once on a dedicated communication stream that is event-ordered after the
    last gradient write, so the collective runs under the remaining backward compute: the decoder's bucket under the
    second ViT backward, layers 11..3 under the layers below them.  `finish()` (after backward) folds gradients that
    arrived through torch autograd (pos_embed behind its resize), flushes whatever is left in bucket order (same order on
    every rank) and makes the compute stream wait for all collectives before AdamW.  xGMI is point-to-point (~153 GB/s per
    link): a 25 MB bucket is ~0.3 ms on the ring, so few, large buckets; the 1/W of the mean is folded into the AdamW
    kernel (grad_scale).  Loss normalisers stay per-rank as in the reference.
    With `overlap=False`, or on backends that cannot run stream-ordered collectives on GPU tensors (gloo: it stages
    through the host and stalls the launching thread), the same buckets are reduced back to back in finish()."""

    def __init__(self, optimizer, bucket_mb=25, group=None, overlap=None, world=None, collective=None, profile=False):
        """`world` / `collective`: test hooks -- a stand-in for the process group (world size and a callable
        (tensor) -> work object with .wait(), run inside the communication stream's context) so that the stream-ordered
        branch (side stream, event ordering, join in finish()) executes on a single GPU, where RCCL cannot be brought up
        with two ranks.  `profile`: HIP events around every bucket's collective and around finish()'s join."""
        self.opt, self.group = optimizer, group
        self._collective = collective
        self.profile, self._ev, self.timing = bool(profile), [], None
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        optimizer.grad_scale = 1.0 / self.world
        n = optimizer.g.numel()
        per = max(1, int(bucket_mb * 1024 * 1024 // 4))
        groups = getattr(optimizer, "groups", None)
        self._slot = {}          # id(param) -> bucket index
        self._expected, self._done = {}, {}
        if groups:               # cut at parameter boundaries (arena order = model.named_parameters() order)
            offs = optimizer.seg_off.tolist()
            self.buckets, start, members = [], 0, []
            for i, g_ in enumerate(groups):
                members.append(g_["param"])
                if offs[i + 1] - start >= per or i == len(groups) - 1:
                    self.buckets.append((start, offs[i + 1]))
                    for prm in members:
                        self._slot[id(prm)] = len(self.buckets) - 1
                    start, members = offs[i + 1], []
        else:                    # bare arena (tests)
            self.buckets = [(s, min(n, s + per)) for s in range(0, n, per)]
        self._members = [0] * len(self.buckets)
        for b_ in self._slot.values():
            self._members[b_] += 1
        backend = dist.get_backend(group) if (dist.is_initialized() and self.world > 1) else None
        if collective is not None:
            backend = "nccl"          # the injected collective is stream-ordered like RCCL's
        self.overlap = (backend == "nccl") if overlap is None else bool(overlap)
        self._async = backend == "nccl"
        # Where the collective's kernel runs.  ProcessGroupNCCL issues an `async_op=True` collective on a stream of ITS
        # OWN (ordered after the caller's current stream), so the hardware queue that matters is one this class never
        # sees.  Since torch 2.8 a blocking-style call (`async_op=False`) is enqueued on the CURRENT stream instead: issued
        # inside `torch.cuda.stream(self._comm)` the RCCL kernel runs on the communication stream picked below -- the one
        # whose queue was probed against the step's streams.  (`on_comm_stream=False`, or an older torch, keeps the
        # async form + a stream-level wait.)
        tv = tuple(int(x) for x in torch.__version__.split("+")[0].split(".")[:2])
        self.on_comm_stream = tv >= (2, 8) and not os.environ.get("SVL_RCCL_ASYNC_OP")
        # stream -> hardware-queue map: the step's own streams are created FIRST (deterministic map, see
        # create_step_streams), then the communication stream, and the result is probed once and kept for the bench line
        self.queue_info = None
        on_gpu = optimizer.g.is_cuda
        if on_gpu and self.world > 1:
            step_streams = create_step_streams(optimizer.g.device)
        self._comm = None
        if on_gpu and self.world > 1 and not os.environ.get("SVL_NO_QUEUE_PROBE"):
            # The communication stream is PICKED, not taken as it comes: up to six fresh streams are probed against the
            # step's streams and the first one that shares a hardware queue with none of them is kept (the runtime's
            # stream -> queue assignment is not round-robin once library helper streams exist: with 8 queues the first
            # fresh stream landed on the main stream's queue, measured).  Blocking backends have no communication stream:
            # the same search reports what one created at this point WOULD get.
            names = ["main", "second"] + (["weight_gradient"] if len(step_streams) > 2 else [])
            helpers = step_helper_streams(step_streams)          # (round 6: the leftover-row / ragged-row launches' streams too)
            names = names + [n_ + "_helper" for n_ in names[:len(helpers)]]
            step_streams = list(step_streams) + helpers
            tried, pick, shared = [], None, None
            for _ in range(8):
                cand = torch.cuda.Stream(optimizer.g.device)
                sh_ = [n_ for n_, s_ in zip(names, step_streams) if ops.streams_share_queue(s_, cand)]
                tried.append(cand)               # (kept alive: a destroyed stream's queue slot would be handed out again)
                if pick is None or len(sh_) < len(shared):
                    pick, shared = cand, sh_
                if not sh_:
                    break
            # (the unpicked candidates are kept only until the pick is made: a destroyed stream's queue slot is handed out
            # again, which is fine once nothing else is created by this class)
            self._probed_streams = [pick]
            n_tried = len(tried)
            del tried
            if self._async:
                self._comm = pick
            self.queue_info = dict(gpu_max_hw_queues=int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                                   comm_stream_shares_queue=bool(shared), shares_queue_with=shared,
                                   streams_probed=n_tried, weight_gradient_stream=bool(ops.WGRAD_STREAM),
                                   collective_runs_on=("the communication stream (async_op=False under torch >= 2.8)"
                                                       if self.on_comm_stream else "ProcessGroupNCCL's internal stream"),
                                   comm_stream="dedicated" if self._async else "none (blocking backend; probed a stand-in)")
        elif self._async and on_gpu:
            self._comm = torch.cuda.Stream(optimizer.g.device)
        self._works, self._fired, self._complete = [], set(), [0] * len(self.buckets)
        self.early_fires = 0
        if self.world > 1 and groups:
            for g_ in groups:
                g_["param"]._svl_reducer = self

    # ---- region hooks (gradsync.expect / gradsync.ready) ---------------------------------------------------------
    def _expect(self, prm):
        self._expected[id(prm)] = self._expected.get(id(prm), 0) + 1

    def _ready(self, prm):
        k = id(prm)
        self._done[k] = self._done.get(k, 0) + 1
        if self._done[k] == self._expected.get(k, 0):
            b_ = self._slot[k]
            self._complete[b_] += 1
            if self.overlap and self._complete[b_] == self._members[b_]:
                self.early_fires += 1           # (statistics: buckets launched from inside backward)
                self._fire(b_)

    def _fire(self, b_):
        if b_ in self._fired:
            return
        self._fired.add(b_)
        s, e = self.buckets[b_]
        g = self.opt.g[s:e]
        if self._comm is not None:
            self._comm.wait_stream(torch.cuda.current_stream())     # ordered after the last gradient write
            with torch.cuda.stream(self._comm):
                if self.profile:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self._comm)
                if self._collective is not None:
                    w = self._collective(g)
                elif self.on_comm_stream:
                    # enqueued on the current stream = self._comm (torch >= 2.8): no work object, nothing to wait for
                    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=False)
                    w = None
                else:
                    w = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                # (async form: ProcessGroupNCCL runs the collective on its OWN stream, which only waits for the caller's; the
                # stream-level wait below orders the communication stream after the collective itself, so the events
                # around it bracket its duration, not its enqueue, and joining `_comm` joins every collective.  Only for
                # works whose wait() is stream-level: a blocking wait here would stall backward on the host.)
                if w is not None and not os.environ.get("TORCH_NCCL_BLOCKING_WAIT"):
                    w.wait()
                self._works.append(w)
                if self.profile:
                    e1.record(self._comm)
                    self._ev.append((b_, (e - s) * 4, e0, e1))
        else:
            t0 = time.perf_counter() if self.profile else 0.0
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            if self.profile:       # blocking collective on the compute thread (gloo dry runs): host wall time, all of it exposed
                self._ev.append((b_, (e - s) * 4, None, time.perf_counter() - t0))

    def broadcast_params(self, src=0):
        if self.world > 1 and self._collective is None:
            dist.broadcast(self.opt.p, src, group=self.group)

    def begin(self):
        """Start of a step: forget whatever a previous, aborted step left behind -- a
        backward that raised, or a grad-enabled forward that never ran its backward, would otherwise carry `expected` /
        `done` counts into this step and fire a bucket before its last contribution (or never fire it early).
        semivl_train_step calls it before the first grad-enabled forward (the forwards register their `expect`s)."""
        if self._works and self._comm is not None:
            # collectives of an aborted step may still be in flight on arena slices: the next zero_grad / gradient
            # writes on the compute stream must not overtake them
            torch.cuda.current_stream().wait_stream(self._comm)
        self._works, self._fired = [], set()
        self._complete = [0] * len(self.buckets)
        self._expected, self._done = {}, {}

    def finish(self):
        """After backward: every gradient in the arena, every bucket reduced, compute stream ordered after them."""
        if hasattr(self.opt, "_fold_autograd_grads"):
            self.opt._fold_autograd_grads()     # e.g. pos_embed behind its bicubic resize: BEFORE its bucket is reduced
        if self.world > 1:
            for b_ in range(len(self.buckets)):
                self._fire(b_)
            if self.profile and self._comm is not None:
                j0, j1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                j0.record(torch.cuda.current_stream())
            for w in self._works:
                if w is not None:
                    w.wait()                     # stream-level wait (no host sync) for NCCL/RCCL works
            if self._comm is not None:
                torch.cuda.current_stream().wait_stream(self._comm)
            if self.profile and self._comm is not None:
                j1.record(torch.cuda.current_stream())
                self.timing = dict(events=self._ev, join=(j0, j1), early_fires=self.early_fires)
                self._ev = []
            elif self.profile:
                self.timing = dict(events=self._ev, join=None, early_fires=self.early_fires)
                self._ev = []
        self._works, self._fired = [], set()
        self._complete = [0] * len(self.buckets)
        self._expected, self._done = {}, {}

    reduce = finish

    def timing_report(self):
        """(after a synchronize) per-bucket collective time on the communication stream and the part of the step the
        compute stream spent waiting for them in finish() -- what backward did NOT hide."""
        t = self.timing
        if not t:
            return None
        if t["join"] is None:      # synchronous backend: every bucket blocks the compute thread for its whole duration
            bk = [dict(bucket=b_, mbytes=round(nb / 2 ** 20, 1), ms=round(dt * 1e3, 3)) for b_, nb, _, dt in t["events"]]
            return dict(buckets=bk, exposed_ms=round(sum(b_["ms"] for b_ in bk), 3), launched_inside_backward=t["early_fires"],
                        queues=self.queue_info,
                        note="blocking all-reduce on the compute thread (backend without stream-ordered collectives): host "
                             "wall time per bucket, all of it exposed")
        return dict(buckets=[dict(bucket=b_, mbytes=round(nb / 2 ** 20, 1), ms=round(e0.elapsed_time(e1), 3))
                             for b_, nb, e0, e1 in t["events"]],
                    exposed_ms=round(t["join"][0].elapsed_time(t["join"][1]), 3), launched_inside_backward=t["early_fires"],
                    queues=self.queue_info,
                    note="ms = HIP events around each bucket's all-reduce on the communication stream; exposed_ms = time "
                         "the compute stream waited in finish() for the collectives backward did not cover")
