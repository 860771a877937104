"""Language-guided decode head (VLGHead) on the HIP kernel library.

Mirror of /root/reference model/decode_heads/vlg_head.py:27-251 (SemanticTransformer, ASPPModule, ASPPPooling, Up,
VLGHead): same constructor kwargs, same parameter names / shapes, same math.  The B x num_classes
"class-images" live channels-last ([(b n), h, w, C] row-major) so every conv is an implicit GEMM of the fp32-MFMA core
(K = taps x channels contiguous), the einops permutes of the reference become stride arithmetic, `repeat`+`cat` of the
skip features is a two-source operand, and ConvTranspose2d is a GEMM with a pixel-shuffle epilogue.
Forward and backward of the whole head are explicit kernel sequences inside one autograd.Function.
"""
import torch
import torch.nn as nn

from .. import gradsync, ops
from .vit import TransformerEncoderLayer


import os as _os
_MEM_DEBUG = bool(_os.environ.get("SVL_MEM_DEBUG"))


# ------------------------------------------------------------------------------------------------ parameter containers
class SemanticTransformer(nn.Module):
    def __init__(self, channels, text_channels, num_heads, pool_size):
        super().__init__()
        assert pool_size is not None and pool_size[0] == pool_size[1]
        self.pool_size = pool_size[0]
        self.pool = nn.AvgPool2d(pool_size)  # parameter-free; kept for module-tree parity
        self.transformer = TransformerEncoderLayer(channels + text_channels, num_heads, 4 * channels)


class ASPPPooling(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.gap = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False),
                                 nn.GroupNorm(cout // 16, cout), nn.ReLU(True))


class ASPPModule(nn.Module):
    def __init__(self, cin, atrous_rates=(1, 6, 12, 18)):
        super().__init__()
        cout = cin
        self.rates = tuple(atrous_rates)
        self.aspp_convs = nn.ModuleList()
        for d in atrous_rates:
            k, pad = (1, 0) if d == 1 else (3, d)
            self.aspp_convs.append(nn.Sequential(nn.Conv2d(cin, cout, k, padding=pad, dilation=d, bias=False),
                                                 nn.GroupNorm(cout // 16, cout), nn.ReLU(True)))
        self.aspp_convs.append(ASPPPooling(cin, cout))
        self.project = nn.Sequential(nn.Conv2d(5 * cout, cout, 1, bias=False), nn.GroupNorm(cout // 16, cout),
                                     nn.ReLU(True))


class Up(nn.Module):
    def __init__(self, cin, cout, cskip):
        super().__init__()
        self.up = nn.ConvTranspose2d(cin, cin - cskip, kernel_size=2, stride=2)
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.GroupNorm(cout // 16, cout),
                                  nn.ReLU(inplace=True),
                                  nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.GroupNorm(cout // 16, cout),
                                  nn.ReLU(inplace=True))


class _GradCollector:
    """Collects parameter gradients produced inside the head's backward; routes them to main_grad or to autograd."""

    def __init__(self):
        self.out = {}

    def put(self, param, fn):
        """fn(dst, accumulate) writes the gradient."""
        prev = self.out.get(id(param))
        if getattr(param, "main_grad", None) is not None:
            fn(param.main_grad, True)
            self.out[id(param)] = None
        elif isinstance(prev, torch.Tensor):
            fn(prev, True)
        else:
            t = ops.empty(*param.shape, device=param.device)
            fn(t, False)
            self.out[id(param)] = t

    def put_tensor(self, param, t):
        self.put(param, lambda dst, acc: (ops.add(dst.view(-1), t.reshape(-1), out=dst.view(-1)) if acc
                                          else ops.eltwise(4, t.reshape(-1), None, out=dst.view(-1))))


# ------------------------------------------------------------------------------------------------ re-materialised activations
# Memory plan, second level (round 4).  Of the ~25 MB a class-image keeps for backward at 512^2, ~8 MB are GroupNorm+ReLU
# outputs y = relu(gn(pre)) whose `pre` is kept anyway (GroupNorm backward needs it) and ~4.5 MB are ConvTranspose2d outputs
# whose input is kept anyway.  With `remat` on they are NOT kept: backward re-creates them right before their one use (the
# consumer's weight gradient) by the forward's own apply pass (`svl_groupnorm_apply`: same kernel, same statistics) or
# ConvTranspose launch -- bit-identical tensors, 8-12 B per element of extra traffic instead of a re-run of the whole
# chunk's forward.  ADE N = 150 at B = 16: every live chunk then fits and none is recomputed (`_head_forward`).
class _LazyGN:
    """relu(groupnorm(pre)) of one conv + GN unit, described by what backward keeps anyway."""

    def __init__(self, sv, gn):
        # (the fields, not `sv` itself: sv["lazy"] -> handle -> sv would be a reference cycle, and the tensors of a finished
        # step would live until the cycle collector runs -- 90 GB at ADE)
        self.pre, self.st, self.geom, self.gn = sv["pre"], sv["st"], sv["geom"], gn

    def get(self, out=None, ldo=None):
        imgs, H, W, _C1, Co, _k, _dil, _pad = self.geom
        pre = self.pre
        if out is None:
            out, ldo = ops.empty(imgs * H * W, Co, device=pre.device), Co
        return ops.groupnorm_apply(pre, Co, self.gn.weight, self.gn.bias, imgs, H * W, Co, self.gn.num_groups, True,
                                   self.st, out, ldo)


class _LazyCat:
    """The ASPP concat buffer [pix, 5 Ch]: four GroupNorm outputs and the upsampled pooling branch."""

    def __init__(self, branches, gy, imgs, h, w, Ch):
        self.branches, self.gy, self.dims = branches, gy, (imgs, h, w, Ch)

    def get(self):
        imgs, h, w, Ch = self.dims
        cat = ops.empty(imgs * h * w, 5 * Ch, device=self.gy.device)
        for j, lz in enumerate(self.branches):
            lz.get(cat[:, j * Ch:], 5 * Ch)
        ops.bilinear_nhwc_fwd(self.gy, Ch, imgs, 1, 1, Ch, True, 1, h, w, cat[:, 4 * Ch:], 5 * Ch)
        return cat


class _LazyConvT:
    """ConvTranspose2d(k 2, s 2) output of an Up block, from the block's (kept or itself lazy) input."""

    def __init__(self, x, Cin, imgs, h, w, wp, Cu, bias):
        self.x, self.args = x, (Cin, imgs, h, w, wp, Cu, bias)

    def get(self, xin=None):
        Cin, imgs, h, w, wp, Cu, bias = self.args
        xin = _mat(self.x) if xin is None else xin
        u = ops.empty(imgs * 4 * h * w, Cu, device=xin.device)
        ops.convT2x_fwd(xin, Cin, imgs, h, w, Cin, wp, Cu, bias, u, Cu)
        return u


class _GNDeferred:
    """Output of a conv + GroupNorm + ReLU unit that is NOT written (round 4): the pre-normalisation tensor plus the
    [imgs, 2, C] (scale, shift) table.  The next 3x3 convolution -- forward and weight gradient -- forms relu(gn(pre)) while
    it stages its operand tiles (`gn_in` of the tiled kernels); .get() materialises the tensor for anything else."""

    def __init__(self, pre, st, gn, table, imgs, HW, Co):
        self.pre, self.st, self.gn, self.table, self.dims = pre, st, gn, table, (imgs, HW, Co)

    def get(self, out=None, ldo=None):
        imgs, HW, Co = self.dims
        if out is None:
            out, ldo = ops.empty(imgs * HW, Co, device=self.pre.device), Co
        return ops.groupnorm_apply(self.pre, Co, self.gn.weight, self.gn.bias, imgs, HW, Co, self.gn.num_groups, True,
                                   self.st, out, ldo)


def _mat(x):
    return x if (x is None or isinstance(x, torch.Tensor)) else x.get()


# ------------------------------------------------------------------------------------------------ conv + GN (+ReLU) unit
def _conv_gn_fwd(x, ldx, imgs, H, W, C1, conv, gn, k, dil, sv, src2=None, ld2=0, C2=0, rep=1, y=None, ldy=None,
                 remat=False, x_keep=None, defer_for=None):
    """`remat`: y is not kept in `sv` (callers keep `sv["lazy"]`, a _LazyGN, instead of the tensor); `x_keep`: what to
    remember as this unit's input in place of the tensor x (a lazy handle of the producer).  `x` may be a _GNDeferred (the
    previous unit's unwritten output).  `defer_for` = output channels of the 3x3 conv + GN unit that is this unit's ONLY
    consumer (or "cout1": the head's Conv2d(C -> 1)): when that consumer's kernels can apply GroupNorm + ReLU to their
    operand themselves, y is not written and a _GNDeferred is returned in its place."""
    Co = conv.weight.shape[0]
    wf, wd = ops.pack_conv_w(conv.weight)
    pad = dil * (k - 1) // 2
    fusable = k == 3 and dil == 1 and gn.num_groups * 16 == Co
    gn_in, dev = None, (x.pre.device if isinstance(x, _GNDeferred) else x.device)
    fused = None
    if isinstance(x, _GNDeferred):     # the operand is relu(gn(x.pre)), applied by the convolution's own staging
        fused = ops.conv3x3_gn(x.pre, ldx, imgs, H, W, C1, wf, Co, gn.eps, gn_in=x.table) if (fusable and C2 == 0) else None
        if fused is not None:
            gn_in = x.table
        else:                           # (a consumer the tiled kernel does not take after all: write the tensor)
            x = x.get()
    if fused is None and fusable:
        # the narrow 3x3 layers: GroupNorm statistics come out of the convolution's epilogue (one tensor pass less)
        fused = ops.conv3x3_gn(x, ldx, imgs, H, W, C1, wf, Co, gn.eps, src2=src2, ld2=ld2, C2=C2, rep=rep)
    deferred = None
    can_defer = False
    if fused is not None and defer_for is not None and y is None and ops.GN_DEFER:
        if defer_for == "cout1":        # the head's Conv2d(C -> 1): LDS-tiled forward + channel-lane weight gradient
            can_defer = ops.conv_cout1_gn_ok(H, W, Co)
        else:                           # a 3x3 conv + GN unit with `defer_for` output channels (tiled forward and wgrad)
            can_defer = defer_for in (32, 64) and ops.conv_wgrad_tiled_ok(imgs, H, W, Co, 0, defer_for, defer_for, Co)
    if can_defer:
        pre, st = fused
        deferred = _GNDeferred(pre, st, gn, ops.groupnorm_scale_shift(st, gn.weight, gn.bias, imgs, Co, gn.num_groups),
                               imgs, H * W, Co)
        ldy = Co
    else:
        if y is None:
            y = ops.empty(imgs * H * W, Co, device=dev)
            ldy = Co
        if fused is not None:
            pre, st = fused
            ops.groupnorm_apply(pre, Co, gn.weight, gn.bias, imgs, H * W, Co, gn.num_groups, True, st, y, ldy)
        else:
            pre = ops.conv_fwd(x, ldx, imgs, H, W, C1, wf, Co, k, k, dil, pad, src2=src2, ld2=ld2, C2=C2, rep=rep)
            st = ops.groupnorm_fwd(pre, Co, gn.weight, gn.bias, gn.eps, imgs, H * W, Co, gn.num_groups, True, y, ldy)
    if sv is not None:
        # the unit's input as backward will read it: the deferred producer's `pre` + table when the forward consumed it
        # that way, else the tensor / the lazy handle of the memory plan
        xs = x if gn_in is not None else (x if x_keep is None else x_keep)
        sv.update(x=xs, gn_in=gn_in, ldx=ldx, pre=pre, y=None if (remat or deferred is not None) else y, ldy=ldy, st=st,
                  wd=wd, geom=(imgs, H, W, C1, Co, k, dil, pad), src2=src2, ld2=ld2, C2=C2, rep=rep)
        if remat:
            sv["lazy"] = _LazyGN(sv, gn)
    return deferred if deferred is not None else y


_WGRAD_AFTER_DGRAD = True     # (launched beside the input gradient instead: two matrix-bound kernels share one pipe, measured slower)


def _conv_gn_bwd(dy, lddy, conv, gn, sv, gc, need_dx=True, dx_acc=None, x=None, chan_sums=None, below=None):
    """`dx_acc`: an existing input-gradient buffer the convolution's dgrad is ADDED to in the GEMM epilogue (the branches of a
    residual / multi-branch node) instead of returned as a new tensor and added by a separate pass.  `x`: the unit's input
    when the caller has already re-materialised it (else `sv["x"]`, tensor or lazy handle).
    Round 6: `below` = (sv, gn) of the conv + GN + ReLU unit whose OUTPUT is this unit's input: this unit's input gradient is
    that GroupNorm's dy, and the tiled dgrad kernel's epilogue then leaves that GroupNorm's backward channel sums -- the call
    returns (dx, sums or None) and the unit below is handed them as `chan_sums` (its statistics pass over dy and x is skipped)."""
    imgs, H, W, C1, Co, k, dil, pad = sv["geom"]
    dpre = ops.empty(imgs * H * W, Co, device=dy.device)
    if chan_sums is not None:
        dg, db = ops.groupnorm_bwd_from_sums(dy, lddy, sv["pre"], Co, sv["st"], gn.weight, gn.bias, imgs, H * W, Co,
                                             gn.num_groups, True, chan_sums, dpre, Co)
    else:
        dg, db = ops.groupnorm_bwd(dy, lddy, sv["pre"], Co, sv["y"], sv["ldy"], sv["st"], gn.weight, imgs, H * W, Co,
                                   gn.num_groups, True, dpre, Co, beta=gn.bias)   # ReLU mask re-derived from `pre`: y is not read
    gc.put_tensor(gn.weight, dg)
    gc.put_tensor(gn.bias, db)
    C2 = sv["C2"]
    gn_in = sv.get("gn_in")
    if gn_in is not None and x is None:    # the input was never written: the weight gradient normalises `pre` on the fly
        x = sv["x"].pre
    else:
        gn_in = None
        x = _mat(sv["x"]) if x is None else x
    def wgrad():
        with ops.wgrad_side(dpre, x, sv["src2"], gn_in):     # off the dependency chain: weight-gradient stream (ops.wgrad_side)
            dwf = ops.conv_wgrad(dpre, Co, x, sv["ldx"], imgs, H, W, C1, Co, k, k, dil, pad, src2=sv["src2"],
                                 ld2=sv["ld2"], C2=C2, rep=sv["rep"], gn_in=gn_in)
            gc.put_tensor(conv.weight, ops.unpack_conv_wgrad(dwf, Co, C1 + C2, k, k))
    # WGRAD_AFTER_DGRAD: the weight gradient is ordered behind the input gradient -- two matrix-bound kernels side by side
    # share one pipe and finish no earlier, while the NEXT unit's GroupNorm backward (bandwidth-bound, on the chain) then has
    # a matrix-bound partner
    if not (_WGRAD_AFTER_DGRAD and need_dx):
        wgrad()
    if not need_dx:
        return None
    below_sums = None
    if dx_acc is not None:
        dx = ops.conv_dgrad(dpre, Co, imgs, H, W, Co, sv["wd"], C1 + C2, k, k, dil, pad, out=dx_acc, ldo=dx_acc.stride(0),
                            accumulate=True)
    else:
        fused = None
        if below is not None and C2 == 0 and (k, dil, pad) == (3, 1, 1):
            bsv, bgn = below
            if bsv["geom"][:3] == (imgs, H, W) and bsv["geom"][4] == C1 and bsv["pre"].stride(0) == C1:
                fused = ops.conv3x3_dgrad_gnb(dpre, Co, imgs, H, W, Co, sv["wd"], C1, bsv["pre"], bsv["st"], bgn.weight, bgn.bias,
                                              bgn.num_groups)
        if fused is not None:
            dx, below_sums = fused
        else:
            dx = ops.conv_dgrad(dpre, Co, imgs, H, W, Co, sv["wd"], C1 + C2, k, k, dil, pad)  # [pix, C1+C2]
    if _WGRAD_AFTER_DGRAD:
        wgrad()
    return (dx, below_sums) if below is not None else dx


# ------------------------------------------------------------------------------------------------ head
class VLGHead(nn.Module):
    def __init__(self, img_size, num_classes, text_in_channels, text_channels, up_channels, skip_in_channels,
                 skip_channels, skip_from_conv_feat, num_layers, num_heads, channels, pool_size, conv1_ksize,
                 loss_decode, align_corners, type=None):
        super().__init__()
        assert loss_decode is None
        self.image_size, self.num_classes, self.align_corners = img_size, num_classes, align_corners
        self.text_in_channels, self.num_layers, self.channels = text_in_channels, num_layers, channels
        self.skip_from_conv_feat = skip_from_conv_feat
        self.num_heads, self.text_channels = num_heads, text_channels
        self.conv1_ksize = conv1_ksize
        self.conv1 = nn.Conv2d(1, channels, kernel_size=conv1_ksize, stride=1, padding=(conv1_ksize - 1) // 2)
        self.aspp = ASPPModule(channels)
        self.layers = nn.ModuleList([SemanticTransformer(channels, text_channels, num_heads, pool_size)
                                     for _ in range(num_layers)])
        self.text_proj = nn.Sequential(nn.Linear(text_in_channels, text_channels), nn.ReLU())
        self.skip_proj = nn.ModuleList([nn.Sequential(nn.Conv2d(sic, sc, kernel_size=3, stride=1, padding=1), nn.ReLU())
                                        for sic, sc in zip(skip_in_channels, skip_channels)])
        self.up1 = Up(channels, up_channels[0], skip_channels[0])
        self.up2 = Up(up_channels[0], up_channels[1], skip_channels[1])
        self.head = nn.Conv2d(up_channels[1], 1, kernel_size=3, stride=1, padding=1)
        self.load_text_embedding = None
        # memory plan (see _chunk_plan / _head_forward): class-images per chunk, allocated-bytes ceiling above which a
        # chunk's activations are not kept but recomputed in backward (None: keep everything), live sample ranges
        self.chunk_class_images = int(_os.environ.get("SVL_HEAD_CHUNK", "1344"))
        self.act_limit_bytes = None
        self.remat = None              # None: decide per step from the activation budget (_remat_decision)
        self._bwd_ranges = None
        if (channels + text_channels) % num_heads or (channels + text_channels) // num_heads != 64:
            raise NotImplementedError("SemanticTransformer head dim must be 64")

    # ---------------------------------------------------------------------------------------------------
    def forward_tokens(self, feats, text, hw, fp_masks=None, fp_rate=0.5, out_size=None, fp_range=None, skip0_hw=None):
        """feats: [v0, v4, emb] token tensors [b, hw, C]; text [N, 512] (any float dtype).  With
        `skip_from_conv_feat` (vlg_head.py:196-205) slot v0 -- the skip of the second Up block -- is the conv_encoder's
        feature [b, h0*w0, C0] on its own grid `skip0_hw` = (h0, w0).
        fp_masks: None, or list of three {0,1} masks [b, C_i]: the batch is doubled with the channel-dropped copy
        (builder.py:78-89).  `fp_range=(s0, s1)` perturbs only samples [s0, s1) (masks [s1-s0, C_i]): the step never
        reads the perturbed copy of the labeled half (semivl.py:247), so it need not be decoded.
        Returns logits [b', N, S, S]."""
        params = [p for p in self.parameters() if p.requires_grad]
        need_grad = torch.is_grad_enabled() and (bool(params) or any(f.requires_grad for f in feats))
        out_size = out_size or (self.image_size, self.image_size)
        if need_grad:
            return _HeadFn.apply(self, (hw, skip0_hw or hw), fp_masks, (fp_rate, fp_range), out_size, text, feats[0],
                                 feats[1], feats[2], *params)
        with ops.prof_scope("head"):
            return _head_forward(self, (hw, skip0_hw or hw), fp_masks, (fp_rate, fp_range), out_size, text, feats, None)[0]

    def forward(self, inputs, force_output_pred_masks=False):
        """Reference signature (vlg_head.py:192-251): inputs = [[feature_pyramid, global], text_feats, conv_feats]."""
        pyramid = inputs[0][0]
        toks, hw = [], None
        for f in pyramid:  # NCHW (possibly channels-last views) -> tokens
            b, c, h, w = f.shape
            hw = (h, w)
            toks.append(f.permute(0, 2, 3, 1).contiguous().view(b, h * w, c))
        skip0_hw = None
        if self.skip_from_conv_feat:  # skip_feats = [*pyramid[:-1][::-1], *conv_feats[::-1]] (vlg_head.py:196-205)
            cf = inputs[2][0]
            skip0_hw = tuple(cf.shape[2:])
            toks = [cf.permute(0, 2, 3, 1).contiguous().view(cf.shape[0], -1, cf.shape[1])] + toks
        size = (self.image_size, self.image_size) if force_output_pred_masks else (4 * hw[0], 4 * hw[1])
        x = self.forward_tokens(toks, inputs[1], hw, out_size=size, skip0_hw=skip0_hw)
        return {"pred_masks": x} if force_output_pred_masks else x


def _chunk_plan(m, b, N):
    """Sample chunks [(s0, s1, live)] of a decoded batch of b samples x N class-images.  Every op of the head is
    per-sample (the SemanticTransformer couples the N classes of ONE sample, vlg_head.py:44-62), so the batch can be cut
    anywhere along b.  `m._bwd_ranges = {b: [(s0, s1), ...]}` (set by the training step) names the sample ranges whose
    dlogits are not identically zero; everything else is 'dead': decoded, never saved, never back-propagated.
    Chunks hold at most `m.chunk_class_images` class-images (the head's transient + saved activations scale with it:
    ~33 MB per class-image at 512^2)."""
    ranges = (getattr(m, "_bwd_ranges", None) or {}).get(b)
    segs, pos = [], 0
    for s0, s1 in (ranges if ranges is not None else [(0, b)]):
        if s0 > pos:
            segs.append((pos, s0, False))
        segs.append((s0, s1, True))
        pos = s1
    if pos < b:
        segs.append((pos, b, False))
    per = max(1, int(getattr(m, "chunk_class_images", 1344)) // N)
    out = []
    for s0, s1, live in segs:
        n = -(-(s1 - s0) // per)
        size = -(-(s1 - s0) // n)
        for c0 in range(s0, s1, size):
            out.append((c0, min(s1, c0 + size), live))
    return out


def _kept_bytes_per_class_image(m, HW, level=0):
    """What backward keeps per class-image (fp32) at re-materialisation level 0 / 1 / 2: conv1 out, 4 ASPP pre + the concat,
    project pre, the residual sum (12 Ch maps at h x w), ~1.8 MB of SemanticTransformer tokens per 1024 pixels, and per Up
    block at 4x / 16x the pixels: ConvTranspose out + 2 x pre + the second unit's output (the first unit's output -- and
    up2's second, consumed by the head conv -- are never written: _GNDeferred; counted as kept when those kernels cannot
    take the layer would be pessimistic by < 15 %).  Level 1 drops the ConvTranspose outputs and up1's output, level 2 the
    ASPP concat as well."""
    Ch = m.channels
    c_up1, c_up2 = m.up1.conv[0].weight.shape[0], m.up2.conv[0].weight.shape[0]
    cu1, cu2 = m.up1.up.weight.shape[1], m.up2.up.weight.shape[1]
    maps = 12 * Ch + 450 + 4 * (cu1 + 3 * c_up1) + 16 * (cu2 + 2 * c_up2)
    if level >= 1:
        maps -= 4 * (cu1 + c_up1) + 16 * cu2
    if level >= 2:
        maps -= 5 * Ch
    return 4 * HW * maps


def _remat_decision(m, plan, N, HW, dev):
    """Re-materialisation level (see _LazyGN): 0 keep everything, 1 re-create the Up blocks' ConvTranspose / GroupNorm outputs
    in backward, 2 the ASPP concat too.  `m.remat`: False / True (= 2) / a level, or None = decide: the LOWEST level at which
    everything the step's grad-carrying decodes keep fits under the activation budget (then no chunk has to be re-run as a
    whole; ADE N = 150 at B = 16: level 1).  The training step announces the class-image count of ALL its live decodes
    (`m._live_class_images`) and the decision of the step's first decode holds for the others (`m._remat_step`)."""
    mode = getattr(m, "remat", None)
    if mode is not None:
        return 2 if mode is True else int(mode)
    step = getattr(m, "_remat_step", None)
    if isinstance(step, dict) and "on" in step:
        return step["on"]
    limit = getattr(m, "act_limit_bytes", None)
    on = 0
    if limit is not None and dev.type == "cuda":
        live = getattr(m, "_live_class_images", None) or sum(s1 - s0 for s0, s1, lv in plan if lv) * N
        room = 0.9 * (limit - torch.cuda.memory_allocated(dev))
        while on < 2 and live * _kept_bytes_per_class_image(m, HW, on) > room:
            on += 1
    if isinstance(step, dict):
        step["on"] = on
    return on


def _head_forward(m, hw, fp_masks, fp_cfg, out_size, text, feats, chunks_out):
    """Whole-batch forward, executed chunk by chunk.  chunks_out: None (inference) or a list that receives
    (s0, s1, live, saved-or-None) per chunk; `saved` is None for dead chunks and for chunks whose activations did not
    fit under `m.act_limit_bytes` (they are recomputed in backward).  Returns (logits [b, N, S, S], shared dict)."""
    fp_rate, fp_range = fp_cfg
    (h, w), (h0, w0) = hw
    HW, HW0 = h * w, h0 * w0
    v0, v4, emb = [f.contiguous() for f in feats]
    b0 = emb.shape[0]
    dev = emb.device
    N = text.shape[0]
    # (the reference asserts the same, vlg_head.py:212: `list(text_feats.shape) == [B, self.num_classes, C]` -- its concept
    #  aggregation behind the decoder, vlg_head.py:242-244, is therefore unreachable; concepts enter through MaskCLIP's guidance)
    assert N == m.num_classes, f"VLGHead: {N} text embeddings for num_classes = {m.num_classes} (vlg_head.py:212)"
    Ce, Cv, C0 = emb.shape[2], v4.shape[2], v0.shape[2]
    # ---- feature perturbation: cat(f, dropout2d(f)) ------------------------------------------------------
    if fp_masks is not None:
        r0, r1 = fp_range if fp_range is not None else (0, b0)
        b = b0 + (r1 - r0)
        sc = 1.0 / (1.0 - fp_rate)

        def dbl(f, mk, Cc, hw_):
            out = ops.empty(b * hw_, Cc, device=dev)
            ops.eltwise(4, f.view(-1), None, out=out.view(-1)[:b0 * hw_ * Cc])
            ops.chanmask(f.view(b0 * hw_, Cc)[r0 * hw_:r1 * hw_], mk.contiguous(), sc, hw_, out=out[b0 * hw_:])
            return out
        v0, v4, emb = dbl(v0, fp_masks[0], C0, HW0), dbl(v4, fp_masks[1], Cv, HW), dbl(emb, fp_masks[2], Ce, HW)
    else:
        b = b0
        v0, v4, emb = v0.view(b * HW0, C0), v4.view(b * HW, Cv), emb.view(b * HW, Ce)
    # ---- class-side tensors shared by every chunk ------------------------------------------------------------
    textf = text.float().contiguous()
    textn, _ = ops.l2norm_fwd(textf, 1e-12)
    tp = ops.linear(textn, m.text_proj[0].weight, m.text_proj[0].bias, act=ops.ACT_RELU)
    shared = dict(textn=textn, tp=tp, b0=b0, b=b, N=N, hw=hw, out_size=out_size, fp=(fp_masks, fp_rate, fp_range),
                  feats=(v0, v4, emb))
    # ---- skip projections (order [v4, v0], vlg_head.py:207): per SAMPLE, not per class-image -> once for the whole decoded
    # batch (per chunk of 8 samples they were 15 launches of an M = 8192, N = 16 / 32, K = 6912 GEMM at 4-9 TF each)
    skips, skip_sv = [], []
    for proj, f, (Cf, fh, fw) in zip(m.skip_proj, (v4, v0), ((Cv, h, w), (C0, h0, w0))):
        wf, wd = ops.pack_conv_w(proj[0].weight)
        Cs = proj[0].weight.shape[0]
        sk = ops.conv_fwd(f, Cf, b, fh, fw, Cf, wf, Cs, 3, 3, 1, 1, bias=proj[0].bias, act=ops.ACT_RELU)
        skips.append(sk)
        skip_sv.append(dict(x=f, wd=wd, y=sk, Cs=Cs, geo=(Cf, fh, fw)))
    shared.update(skips=skips, skip_sv=skip_sv)
    logits = ops.empty(b, N, out_size[0], out_size[1], device=dev)
    limit = getattr(m, "act_limit_bytes", None)
    plan = _chunk_plan(m, b, N)
    shared["remat"] = chunks_out is not None and _remat_decision(m, plan, N, HW, dev)
    over = False
    for s0, s1, live in plan:
        sv = {} if (chunks_out is not None and live and not over) else None
        _head_core_forward(m, shared, s0, s1, sv, logits[s0:s1])
        if sv is not None and limit is not None and torch.cuda.memory_allocated(dev) > limit:
            sv, over = None, True    # does not fit: drop, recompute this chunk (and the following ones) in backward
        if chunks_out is not None:
            chunks_out.append((s0, s1, live, sv))
    if _MEM_DEBUG and chunks_out is not None:
        print(f"[head fwd] b={b} N={N} remat={shared['remat']} chunks={[(c[0], c[1], c[2], c[3] is not None) for c in chunks_out]} "
              f"allocated {torch.cuda.memory_allocated(dev) / 2**30:.1f} GB (limit {None if limit is None else round(limit / 2**30, 1)})", flush=True)
    return logits, shared


def _head_core_forward(m, shared, s0, s1, sv, logits_out):
    """Forward of samples [s0, s1) of the (doubled) batch.  sv: dict receiving what backward needs, or None.
    logits_out: contiguous [s1-s0, N, S, S] slice to write, or None (backward-time recompute: activations only)."""
    (h, w), (h0, w0) = shared["hw"]
    HW, HW0 = h * w, h0 * w0
    N, out_size, textn, tp = shared["N"], shared["out_size"], shared["textn"], shared["tp"]
    b = s1 - s0
    v0, v4, emb = shared["feats"]
    v0, v4, emb = v0[s0 * HW0:s1 * HW0], v4[s0 * HW:s1 * HW], emb[s0 * HW:s1 * HW]
    dev = emb.device
    Ch, Ct = m.channels, m.text_channels
    Ce, Cv, C0 = emb.shape[1], v4.shape[1], v0.shape[1]
    imgs = b * N
    # ---- cosine similarity map (vlg_head.py:214-217) ------------------------------------------------------
    embn, inv_e = ops.l2norm_fwd(emb, 1e-12)
    sim = ops.empty(imgs * HW, 1, device=dev)  # [(b n), h, w, 1]
    ops.gemm(ops.A_KC, ops.B_KC, HW, N, Ce, ops.Op(embn, Ce, 0, HW * Ce, 0), ops.Op(textn, Ce), sim, ldc_m=1, ldc_n=HW,
             batch=b, c_bso=N * HW)
    # ---- conv1 7x7 -----------------------------------------------------------------------------------------
    k1 = m.conv1_ksize
    w1f, w1d = ops.pack_conv_w(m.conv1.weight)
    x1 = ops.conv_fwd(sim, 1, imgs, h, w, 1, w1f, Ch, k1, k1, 1, (k1 - 1) // 2, bias=m.conv1.bias)
    # ---- ASPP ----------------------------------------------------------------------------------------------
    level = int(shared.get("remat") or 0) if sv is not None else 0
    remat, remat_aspp = level >= 1, level >= 2
    cat = ops.empty(imgs * HW, 5 * Ch, device=dev)
    aspp_sv = []
    for j, d in enumerate(m.aspp.rates):
        seq = m.aspp.aspp_convs[j]
        s_ = {} if sv is not None else None
        _conv_gn_fwd(x1, Ch, imgs, h, w, Ch, seq[0], seq[1], 1 if d == 1 else 3, d, s_, y=cat[:, j * Ch:], ldy=5 * Ch,
                     remat=remat_aspp)
        aspp_sv.append(s_)
    gap = m.aspp.aspp_convs[4].gap
    pooled = ops.avgpool_cat_fwd(x1, imgs, h, w, Ch, (h, w), None, 1)      # AdaptiveAvgPool2d(1) on any map shape
    s_gap = {} if sv is not None else None
    gy = _conv_gn_fwd(pooled, Ch, imgs, 1, 1, Ch, gap[1], gap[2], 1, 1, s_gap)
    ops.bilinear_nhwc_fwd(gy, Ch, imgs, 1, 1, Ch, True, 1, h, w, cat[:, 4 * Ch:], 5 * Ch)
    s_proj = {} if sv is not None else None
    x2 = _conv_gn_fwd(cat, 5 * Ch, imgs, h, w, 5 * Ch, m.aspp.project[0], m.aspp.project[1], 1, 1, s_proj, remat=remat,
                      x_keep=_LazyCat([a_["lazy"] for a_ in aspp_sv], gy, imgs, h, w, Ch) if remat_aspp else None)
    # y = x + project(cat).  (x2 is not needed again: the project GN's ReLU mask is re-derived from its `pre`; without
    # remat a new buffer is used all the same so that the unit's kept `y` stays what it was.)
    x = ops.add(x2, x1, out=x2) if remat else ops.add(x2, x1)
    del cat
    # ---- semantic reasoning ----------------------------------------------------------------------------------
    tr_sv = []
    for lyr in m.layers:
        s_ = {} if sv is not None else None
        x = _semtr_forward(lyr, x, tp, imgs, b, N, h, w, Ch, Ct, s_)
        tr_sv.append(s_)
    # ---- skip projections: computed once per decoded batch (_head_forward); this chunk's samples ---------------------
    skips = [shared["skips"][0][s0 * HW:s1 * HW], shared["skips"][1][s0 * HW0:s1 * HW0]]
    # ---- upsampling ------------------------------------------------------------------------------------------
    s_up1 = {} if sv is not None else None
    g2 = _up_forward(m.up1, x, imgs, h, w, skips[0], h, w, b, N, s_up1, remat)
    s_up2 = {} if sv is not None else None
    g4 = _up_forward(m.up2, g2, imgs, 2 * h, 2 * w, skips[1], h0, w0, b, N, s_up2, remat,
                     x_keep=s_up1["b"]["lazy"] if remat else None, out_for="cout1")   # (the head conv is g4's only consumer)
    g4d = g4 if isinstance(g4, _GNDeferred) else None     # never written: the head conv normalises up2's last `pre` itself
    C4 = g4d.dims[2] if g4d is not None else g4.shape[1]
    whf, whd = ops.pack_conv_w(m.head.weight)
    if logits_out is not None:
        direct = out_size == (4 * h, 4 * w)
        lg = ops.conv_cout1_fwd(g4d.pre if g4d is not None else g4, C4, imgs, 4 * h, 4 * w, C4, whf, 3, 3, 1, 1,
                                bias=m.head.bias, out=logits_out.view(-1, 1) if direct else None,
                                gn_in=g4d.table if g4d is not None else None)  # [(b n), 4h, 4w, 1]
        if not direct:
            ops.bilinear_planes_fwd(lg.view(b, N, 4 * h, 4 * w), 4 * h, 4 * w, m.align_corners, out_size[0], out_size[1],
                                    out=logits_out)
    if sv is not None:
        sv.update(dims=(b, N, h, w, HW, imgs, Ch, Ct, Ce, Cv, C0, HW0), embn=embn,
                  inv_e=inv_e, textn=textn, sim=sim, w1d=w1d, x1=x1, aspp=aspp_sv, gap=s_gap, pooled=pooled, proj=s_proj,
                  tp=tp, tr=tr_sv, up1=s_up1, up2=s_up2, g4=g4d if g4d is not None else (s_up2["b"]["lazy"] if remat else g4), whd=whd,
                  out_size=out_size)


def _semtr_forward(lyr, x, tp, imgs, b, N, h, w, Ch, Ct, sv):
    """vlg_head.py:39-67.  x [(b n) h w, Ch] is updated in place (x += upsample(transformer(pool(x) ++ text)))."""
    P = lyr.pool_size
    hp, wp = h // P, w // P
    t = lyr.transformer
    p = t.plist()
    E = Ch + Ct
    tok = ops.avgpool_cat_fwd(x, imgs, h, w, Ch, P, tp, N)  # rows [(b n), hp, wp], E channels
    y1, st1 = ops.layernorm_fwd(tok, p["ln1w"], p["ln1b"], t.eps)
    qkv = ops.linear(y1, p["win"], p["bin"])
    G = hp * wp
    E3 = qkv.shape[1]
    mfma_seq = N >= ops.SEQATTN_MFMA_MIN and (E3 // 3) == 64 * t.num_heads and qkv.is_cuda
    if mfma_seq:
        # long class sequences (COCO / ADE): '(b n) (h w) c -> (b h w) n c' as one row permutation, then the fused MFMA
        # attention of the ViT blocks on b*G sequences of N tokens (flash-style: log-sum-exp rows instead of the
        # [N, N] probabilities; the wave-per-query kernel ran these at ~9 TF)
        qkv = ops.permute_rows(qkv, b, N, G, E3)                  # kept in THIS layout for backward
        o_t, probs = ops.attention_fwd(qkv, b * G, N, t.num_heads)   # (`probs` slot: the LSE rows)
        o = ops.permute_rows(o_t, b, G, N, E3 // 3)
        del o_t
    else:
        o, probs = ops.seqattn_fwd(qkv, b * G, G, N, t.num_heads, N * G, 1, G)
    t2 = ops.linear(o, p["wout"], p["bout"], resid=tok)
    y2, st2 = ops.layernorm_fwd(t2, p["ln2w"], p["ln2b"], t.eps)
    h_pre = ops.empty(y2.shape[0], p["w1"].shape[0], device=x.device) if sv is not None else None
    hh = ops.linear(y2, p["w1"], p["b1"], act=ops.ACT_GELU, preact=h_pre)
    t3 = ops.linear(hh, p["w2"], p["b2"], resid=t2)
    # keep the image half (first Ch channels), bilinear up (align_corners=True), residual add
    ops.bilinear_nhwc_fwd(t3, E, imgs, hp, wp, Ch, True, 1, h, w, x, Ch, accumulate=True)
    if sv is not None:
        sv.update(tok=tok, y1=y1, st1=st1, qkv=qkv, o=o, probs=probs, t2=t2, y2=y2, st2=st2, h_pre=h_pre, hh=hh,
                  dims=(hp, wp, G, E), mfma_seq=mfma_seq)
    return x


def _semtr_backward(lyr, dx, dtp_acc, imgs, b, N, h, w, Ch, Ct, sv, gc):
    """dx [(b n) h w, Ch] grad wrt the block output; returns grad wrt its input, accumulates dtext into dtp_acc."""
    hp, wp, G, E = sv["dims"]
    t = lyr.transformer
    p = t.plist()
    a, f = t.attn.attn, t.ffn.layers
    rows = imgs * G
    dev = dx.device
    dt3 = ops.zeros(rows, E, device=dev)
    ops.bilinear_nhwc_bwd(dx, Ch, imgs, hp, wp, Ch, True, 1, h, w, dt3, E)
    # t3 = t2 + W2 hh + b2
    def wg_f2():
        with ops.wgrad_side(dt3, sv["hh"]):
            gc.put(f[1].weight, lambda d, acc: ops.matmul_tn(dt3, sv["hh"], out=d, accumulate=acc))
            gc.put(f[1].bias, lambda d, acc: ops.colsum(dt3, out=d, accumulate=acc))
    if not _WGRAD_AFTER_DGRAD:
        wg_f2()
    dh = ops.matmul_nn(dt3, p["w2"])
    if _WGRAD_AFTER_DGRAD:              # (weight gradients behind their input gradient: see _conv_gn_bwd)
        wg_f2()
    dhp = ops.eltwise(1, dh, sv["h_pre"], out=dh)
    def wg_f1():
        with ops.wgrad_side(dhp, sv["y2"]):
            gc.put(f[0][0].weight, lambda d, acc: ops.matmul_tn(dhp, sv["y2"], out=d, accumulate=acc))
            gc.put(f[0][0].bias, lambda d, acc: ops.colsum(dhp, out=d, accumulate=acc))
    if not _WGRAD_AFTER_DGRAD:
        wg_f1()
    dy2 = ops.matmul_nn(dhp, p["w1"])
    if _WGRAD_AFTER_DGRAD:
        wg_f1()
    dt2, dg2, db2 = ops.layernorm_bwd(dy2, sv["t2"], sv["st2"], p["ln2w"], dx_add=dt3, want_wgrad=True)
    gc.put_tensor(t.ln2.weight, dg2)
    gc.put_tensor(t.ln2.bias, db2)
    # t2 = tok + Wout o + bout
    def wg_o():
        with ops.wgrad_side(dt2, sv["o"]):
            gc.put(a.out_proj.weight, lambda d, acc: ops.matmul_tn(dt2, sv["o"], out=d, accumulate=acc))
            gc.put(a.out_proj.bias, lambda d, acc: ops.colsum(dt2, out=d, accumulate=acc))
    if not _WGRAD_AFTER_DGRAD:
        wg_o()
    do = ops.matmul_nn(dt2, p["wout"])
    if _WGRAD_AFTER_DGRAD:
        wg_o()
    if sv["mfma_seq"]:
        Ee = do.shape[1]
        dqkv_t = ops.attention_bwd(ops.permute_rows(do, b, N, G, Ee), sv["qkv"], ops.permute_rows(sv["o"], b, N, G, Ee),
                                   sv["probs"], b * G, N, t.num_heads)
        dqkv = ops.permute_rows(dqkv_t, b, G, N, 3 * Ee)
        del dqkv_t
    else:
        dqkv = ops.seqattn_bwd(do, sv["qkv"], sv["probs"], b * G, G, N, t.num_heads, N * G, 1, G)
    def wg_in():
        with ops.wgrad_side(dqkv, sv["y1"]):
            gc.put(a.in_proj_weight, lambda d, acc: ops.matmul_tn(dqkv, sv["y1"], out=d, accumulate=acc))
            gc.put(a.in_proj_bias, lambda d, acc: ops.colsum(dqkv, out=d, accumulate=acc))
    if not _WGRAD_AFTER_DGRAD:
        wg_in()
    dy1 = ops.matmul_nn(dqkv, p["win"])
    if _WGRAD_AFTER_DGRAD:
        wg_in()
    dtok, dg1, db1 = ops.layernorm_bwd(dy1, sv["tok"], sv["st1"], p["ln1w"], dx_add=dt2, want_wgrad=True)
    gc.put_tensor(t.ln1.weight, dg1)
    gc.put_tensor(t.ln1.bias, db1)
    dx, dtext = ops.avgpool_cat_bwd(dtok, imgs, h, w, Ch, lyr.pool_size, Ct, N, add_to=dx)   # dx += pooled gradient
    ops.add(dtp_acc, dtext, out=dtp_acc)
    return dx


def _up_forward(up, x, imgs, h, w, skip, sh, sw, b, N, sv, remat=False, x_keep=None, out_for=None):
    """vlg_head.py:129-137.  x [(b n) h w, Cin]; skip [b sh sw, Cs] -> [(b n) 2h 2w, Cout].  `remat` (memory plan): the
    ConvTranspose output and the two GroupNorm outputs are not kept; `x_keep`: lazy handle of x when x itself is one."""
    Cin = up.up.weight.shape[0]
    Cu = up.up.weight.shape[1]
    Cs = skip.shape[1]
    dev = x.device
    wp_ = ops.cached_pack(up.up.weight, "convT_fwd",
                          lambda w_: ops.permute4(w_.contiguous(), (2, 2, Cu, Cin), (2, 1, 4, 4 * Cu)).view(4 * Cu, Cin))  # n = (a, b, co)
    u = ops.empty(imgs * 4 * h * w, Cu, device=dev)
    ops.convT2x_fwd(x, Cin, imgs, h, w, Cin, wp_, Cu, up.up.bias, u, Cu)
    sup = ops.empty(b * 4 * h * w, Cs, device=dev)
    ops.bilinear_nhwc_fwd(skip, Cs, b, sh, sw, Cs, True, 1, 2 * h, 2 * w, sup, Cs)
    remat = remat and sv is not None
    xk = x if x_keep is None else x_keep
    sa = {} if sv is not None else None
    Cm = up.conv[0].weight.shape[0]
    g1 = _conv_gn_fwd(u, Cu, imgs, 2 * h, 2 * w, Cu, up.conv[0], up.conv[1], 3, 1, sa, src2=sup, ld2=Cs, C2=Cs, rep=N,
                      remat=remat, x_keep=_LazyConvT(xk, Cin, imgs, h, w, wp_, Cu, up.up.bias) if remat else None,
                      defer_for=up.conv[3].weight.shape[0])     # (conv b is g1's only consumer)
    del u
    sb = {} if sv is not None else None
    g2 = _conv_gn_fwd(g1, Cm, imgs, 2 * h, 2 * w, Cm, up.conv[3], up.conv[4], 3, 1, sb, remat=remat,
                      x_keep=sa["lazy"] if remat else None, defer_for=out_for)
    if sv is not None:
        sv.update(x=xk, wp=wp_, a=sa, b=sb, dims=(Cin, Cu, Cs, sh, sw))
    return g2


def _up_backward(up, dg2, imgs, h, w, b, N, sv, gc):
    """Returns (dx [(b n) h w, Cin], dskip [b h w, Cs])."""
    Cin, Cu, Cs, sh, sw = sv["dims"]
    dev = dg2.device
    # (conv b's input gradient is GroupNorm a's dy: its epilogue leaves that GroupNorm's backward sums, round 6)
    dg1, sums_a = _conv_gn_bwd(dg2, dg2.shape[1], up.conv[3], up.conv[4], sv["b"], gc, below=(sv["a"], up.conv[1]))
    xin = _mat(sv["x"])                 # the block's input (re-materialised once when it is a lazy GroupNorm output) ...
    xa = sv["a"]["x"]                   # ... feeds the re-created ConvTranspose output and the ConvTranspose weight gradient
    dcat = _conv_gn_bwd(dg1, dg1.shape[1], up.conv[0], up.conv[1], sv["a"], gc,
                        x=xa.get(xin) if isinstance(xa, _LazyConvT) else None, chan_sums=sums_a)  # [pix, Cu + Cs]
    ld = Cu + Cs
    # skip half: sum over the N repeats, then bilinear backward
    dskip = ops.empty(b * sh * sw, Cs, device=dev)
    ops.bilinear_nhwc_bwd(dcat[:, Cu:], ld, b, sh, sw, Cs, True, N, 2 * h, 2 * w, dskip, Cs)
    # ConvTranspose half
    def wgrad():
        with ops.wgrad_side(dcat, xin):
            gc.put(up.up.bias, lambda d, acc: ops.colsum(dcat, out=d, accumulate=acc, C_=Cu, ld=ld))
            dwb = ops.convT2x_wgrad(xin, Cin, dcat, ld, imgs, h, w, Cin, Cu)  # [Cin, (a,b,co)]
            gc.put_tensor(up.up.weight, ops.permute4(dwb, (Cin, Cu, 2, 2), (4 * Cu, 1, 2 * Cu, Cu)))   # [Cin, a, b, co] -> [Cin, co, a, b]
    if not _WGRAD_AFTER_DGRAD:
        wgrad()
    wb = ops.cached_pack(up.up.weight, "convT_bwd", lambda w_: ops.permute4(w_.contiguous(), (Cin, 2, 2, Cu), (4 * Cu, 2, 1, 4)).view(Cin, 4 * Cu))
    dx = ops.convT2x_dgrad(dcat, ld, imgs, h, w, Cu, wb, Cin)
    if _WGRAD_AFTER_DGRAD:
        wgrad()                         # (behind the input gradient: see _conv_gn_bwd)
    return dx, dskip


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, hw, fp_masks, fp_cfg, out_size, text, v0, v4, emb, *params):
        chunks = []
        with ops.prof_scope("head"):
            out, shared = _head_forward(m, hw, fp_masks, fp_cfg, out_size, text, [v0, v4, emb], chunks)
        ctx.m, ctx.chunks, ctx.shared, ctx.params = m, chunks, shared, params
        gradsync.expect(params)
        ctx.feat_req = (v0.requires_grad, v4.requires_grad, emb.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dlogits):
        with ops.prof_scope("head"):
            return _HeadFn._backward(ctx, dlogits)

    @staticmethod
    def _backward(ctx, dlogits):
        m, sh = ctx.m, ctx.shared
        b0, b, N = sh["b0"], sh["b"], sh["N"]
        (h, w), (h0, w0) = sh["hw"]
        HW, HW0 = h * w, h0 * w0
        v0f, v4f, embf = sh["feats"]
        C0, Cv, Ce = v0f.shape[1], v4f.shape[1], embf.shape[1]
        dev = dlogits.device
        gc = _GradCollector()
        dlogits = dlogits.contiguous()
        # Samples whose dlogits are identically zero (pred_w is detached, pred_x_fp unused: semivl.py:247,251) contribute
        # exactly nothing to any gradient: every op of the head is per-sample.  The step announced the live sample
        # ranges before the forward (`_bwd_ranges`); dead chunks were decoded without saving anything and are skipped here.
        all_live = all(c[2] for c in ctx.chunks)
        mk_ = ops.empty if (all_live and len(ctx.chunks) == 1) else ops.zeros
        sk4, sk0 = sh["skip_sv"]
        dv0, dv4, demb = ops.zeros(b * HW0, C0, device=dev), ops.zeros(b * HW, Cv, device=dev), mk_(b * HW, Ce, device=dev)
        dsk0, dsk4 = mk_(b * HW0, sk0["Cs"], device=dev), mk_(b * HW, sk4["Cs"], device=dev)
        lo, hi = b, 0          # sample span of the live chunks
        for i, (s0, s1, live, sv) in enumerate(ctx.chunks):
            if not live:
                continue
            lo, hi = min(lo, s0), max(hi, s1)
            if sv is None:     # activations were not kept (memory plan): recompute this chunk's forward, no logits
                sv = {}
                _head_core_forward(m, sh, s0, s1, sv, None)
            a0, a4, ae = _head_backward_core(m, sv, dlogits[s0:s1], gc)
            if _MEM_DEBUG:
                print(f"[head bwd] chunk {s0}:{s1} allocated {torch.cuda.memory_allocated(dev) / 2**30:.1f} GB, peak "
                      f"{torch.cuda.max_memory_allocated(dev) / 2**30:.1f} GB", flush=True)
            ctx.chunks[i] = None
            del sv
            for full, part, hw_ in ((dsk0, a0, HW0), (dsk4, a4, HW), (demb, ae, HW)):
                ops.eltwise(4, part.view(-1), None, out=full[s0 * hw_:s1 * hw_].view(-1))
        # ---- skip projections -> grads for v4 / v0, once over the live sample span (dead samples: exactly zero)
        if hi > lo:
            nb = hi - lo
            for proj, ss, dsk, dvf, hw_ in ((m.skip_proj[0], sk4, dsk4, dv4, HW), (m.skip_proj[1], sk0, dsk0, dv0, HW0)):
                Cs = ss["Cs"]
                Cf, fh, fw = ss["geo"]
                dsl, ysl, xsl = dsk[lo * hw_:hi * hw_], ss["y"][lo * hw_:hi * hw_], ss["x"][lo * hw_:hi * hw_]
                dpre = ops.eltwise(2, dsl, ysl, out=dsl)  # relu backward (post-activation mask)
                with ops.wgrad_side(dpre, xsl):
                    gc.put(proj[0].bias, lambda d, acc, dpre=dpre: ops.colsum(dpre, out=d, accumulate=acc))
                    dwf = ops.conv_wgrad(dpre, Cs, xsl, Cf, nb, fh, fw, Cf, Cs, 3, 3, 1, 1)
                    gc.put_tensor(proj[0].weight, ops.unpack_conv_wgrad(dwf, Cs, Cf, 3, 3))
                ops.conv_dgrad(dpre, Cs, nb, fh, fw, Cs, ss["wd"], Cf, 3, 3, 1, 1, out=dvf[lo * hw_:hi * hw_], ldo=Cf)
        # ---- undo the feature-perturbation doubling
        fp_masks, fp_rate, fp_range = sh["fp"]
        r0, r1 = fp_range if fp_range is not None else (0, b0)

        def undbl(dfull, mk, Cc, hw_):
            if fp_masks is None:
                return dfull.view(b0, hw_, Cc)
            sc = 1.0 / (1.0 - fp_rate)
            second = ops.chanmask(dfull[b0 * hw_:], mk.contiguous(), sc, hw_)
            first = dfull[:b0 * hw_]
            tgt = first[r0 * hw_:r1 * hw_]
            ops.add(tgt, second, out=tgt)
            return first.view(b0, hw_, Cc)
        mk = fp_masks if fp_masks is not None else (None, None, None)
        dv0 = undbl(dv0, mk[0], C0, HW0)
        dv4 = undbl(dv4, mk[1], Cv, HW)
        demb = undbl(demb, mk[2], Ce, HW)
        ctx.chunks = ctx.shared = None
        ops.wgrad_join(produced=gc.out.values())   # (the weight-gradient stream's work of this graph is ordered before what follows)
        gradsync.ready(ctx.params)     # the decoder's gradients of this graph are in the arena
        req = ctx.feat_req
        return (None, None, None, None, None, None, dv0 if req[0] else None, dv4 if req[1] else None,
                demb if req[2] else None) + tuple(gc.out.get(id(p)) for p in ctx.params)


def _head_backward_core(m, sv, dlogits, gc):
    """Backward of the head for the (sub-)batch described by `sv`; returns grads wrt the chunk's two projected skip features
    ([b h0 w0, Cs0], [b h w, Cs4]) and wrt its (doubled) emb tokens."""
    if True:
        b, N, h, w, HW, imgs, Ch, Ct, Ce, Cv, C0, HW0 = sv["dims"]
        dev = dlogits.device
        if sv["out_size"] != (4 * h, 4 * w):
            dlg = ops.bilinear_planes_bwd(dlogits, 4 * h, 4 * w, m.align_corners, sv["out_size"][0], sv["out_size"][1])
        else:
            dlg = dlogits
        dlg = dlg.view(imgs * 16 * HW, 1)
        # ---- head conv
        g4d = sv["g4"] if isinstance(sv["g4"], _GNDeferred) else None
        g4 = g4d.pre if g4d is not None else _mat(sv["g4"])
        C4 = g4.shape[1]
        g4t = g4d.table if g4d is not None else None
        with ops.wgrad_side(dlg, g4, g4t):
            gc.put(m.head.bias, lambda d, acc: ops.colsum(dlg, out=d, accumulate=acc))
            dwh = ops.conv_cout1_wgrad(dlg, g4, C4, imgs, 4 * h, 4 * w, C4, 1, 1, gn_in=g4t)
            gc.put_tensor(m.head.weight, ops.unpack_conv_wgrad(dwh, 1, C4, 3, 3))
        dg4 = ops.conv_dgrad(dlg, 1, imgs, 4 * h, 4 * w, 1, sv["whd"], C4, 3, 3, 1, 1)
        # ---- up2, up1
        dg2, dskip0 = _up_backward(m.up2, dg4, imgs, 2 * h, 2 * w, b, N, sv["up2"], gc)
        sv["up2"] = sv["g4"] = None
        dx, dskip4 = _up_backward(m.up1, dg2, imgs, h, w, b, N, sv["up1"], gc)
        sv["up1"] = None
        # (the skip projections' backward runs once for all chunks: _HeadFn._backward)
        # ---- semantic transformers (reverse)
        dtp = ops.zeros(N, Ct, device=dev)
        for lyr, s_ in zip(reversed(list(m.layers)), reversed(sv["tr"])):
            dx = _semtr_backward(lyr, dx, dtp, imgs, b, N, h, w, Ch, Ct, s_, gc)
        sv["tr"] = None
        dtp_pre = ops.eltwise(2, dtp, sv["tp"], out=dtp)
        gc.put(m.text_proj[0].weight, lambda d, acc: ops.matmul_tn(dtp_pre, sv["textn"], out=d, accumulate=acc))
        gc.put(m.text_proj[0].bias, lambda d, acc: ops.colsum(dtp_pre, out=d, accumulate=acc))
        # ---- ASPP: x2 = x1 + project(cat)
        dcat = _conv_gn_bwd(dx, Ch, m.aspp.project[0], m.aspp.project[1], sv["proj"], gc)  # [pix, 5Ch]
        dx1 = dx  # residual branch (dx is not used afterwards; accumulate into it)
        for j, d in enumerate(m.aspp.rates):
            seq = m.aspp.aspp_convs[j]
            _conv_gn_bwd(dcat[:, j * Ch:], 5 * Ch, seq[0], seq[1], sv["aspp"][j], gc, dx_acc=dx1)   # dx1 += branch dgrad
        gap = m.aspp.aspp_convs[4].gap
        dgy = ops.empty(imgs, Ch, device=dev)
        ops.bilinear_nhwc_bwd(dcat[:, 4 * Ch:], 5 * Ch, imgs, 1, 1, Ch, True, 1, h, w, dgy, Ch)
        dpooled = _conv_gn_bwd(dgy, Ch, gap[1], gap[2], sv["gap"], gc)  # [imgs, Ch]
        # avgpool over the whole map: every pixel gets dpooled / HW
        ops.avgpool_cat_bwd(dpooled, imgs, h, w, Ch, (h, w), 0, 1, add_to=dx1)   # dx1 += dpooled / HW
        sv["aspp"] = sv["proj"] = None
        # ---- conv1
        k1 = m.conv1_ksize
        with ops.wgrad_side(dx1, sv["sim"]):
            gc.put(m.conv1.bias, lambda d, acc: ops.colsum(dx1, out=d, accumulate=acc))
            dw1 = ops.conv_wgrad(dx1, Ch, sv["sim"], 1, imgs, h, w, 1, Ch, k1, k1, 1, (k1 - 1) // 2)
            gc.put_tensor(m.conv1.weight, ops.unpack_conv_wgrad(dw1, Ch, 1, k1, k1))
        wtap = ops.cached_pack(m.conv1.weight, "tap", lambda w_: w_.view(Ch, k1 * k1).t().contiguous())  # [tap, co]
        dsim = ops.conv_cin1_dgrad(dx1, Ch, imgs, h, w, Ch, wtap, k1, k1, 1, (k1 - 1) // 2)  # [(b n) hw, 1]
        # ---- cosine sim: demb_n[b,p,c] = sum_n dsim[b,n,p] textn[n,c]
        dembn = ops.empty(b * HW, Ce, device=dev)
        ops.gemm(ops.A_MC, ops.B_NC, HW, Ce, N, ops.Op(dsim, HW, 0, N * HW, 0), ops.Op(sv["textn"], Ce), dembn,
                 ldc_m=Ce, batch=b, c_bso=HW * Ce)
        demb = ops.l2norm_bwd(dembn, sv["embn"], sv["inv_e"])
        return dskip0, dskip4, demb
