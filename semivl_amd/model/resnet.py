"""Convolutional side encoder of the Cityscapes recipe: mmseg `ResNetV1c` restricted to what
`configs/_base_/models/vlm-vlg-aspp-s2p4-skr04-ftap-mcvitb.py:50-60` instantiates -- depth 101, `num_stages=1`
(deep stem 3x(conv3x3+BN+ReLU), MaxPool 3/2/1, layer1 = 3 Bottlenecks, stride 4, 256 channels), `norm_cfg` SyncBN,
trainable (lr_mult 0.1, `experiments.py:251`).  mmseg itself is not vendored in the reference, so the block structure
follows mmseg 0.24's published `ResNet`/`Bottleneck` (style 'pytorch', `deep_stem`, no `avg_down`,
`zero_init_residual`); parameter/buffer names are mmseg's (`stem.0.weight`, `stem.1.running_mean`,
`layer1.0.downsample.1.weight`, ...), which is also the key schema the CLIP->mmseg converter emits for ResNets.

Everything runs channels-last on the library: 3x3 convs as implicit GEMM (`svl_gemm_f32`, im2col on the fly, stride 2
for the first one), 1x1 convs as plain GEMMs, BatchNorm through `svl_bn_*` (batch statistics reduced in double; with
torch.distributed initialised the [2, C] sum vectors are all-reduced = SyncBN), max pooling through `svl_maxpool3x3s2_*`.
One `autograd.Function` spans the encoder; weight gradients go to the parameters' `main_grad` sinks when present.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import gradsync, ops
from .vlg_head import _GradCollector


class Bottleneck(nn.Module):
    """Parameter container with mmseg's names (conv1/bn1 1x1, conv2/bn2 3x3, conv3/bn3 1x1 x4, downsample.{0,1})."""
    expansion = 4

    def __init__(self, inplanes, planes, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample


def _sync_sums(sums):
    """SyncBN: the ranks exchange per-channel sums; returns the number of ranks that contributed.
    One small ([2, C] doubles) SUM all-reduce per BatchNorm layer per direction, stream-ordered under RCCL (no host
    sync).  The 13 exchanges cannot be merged into one: layer k+1's input is layer k's output normalised with the GLOBAL
    statistics, so each exchange depends on the previous one (same in backward) -- torch.nn.SyncBatchNorm has the same
    structure."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=_syncbn_group())
        return dist.get_world_size()
    return 1


_SYNCBN_PG = []


def _syncbn_group():
    """The SyncBN exchanges get a communicator of their own (created at the first exchange, the same program point on
    every rank): on the default one they would queue behind the 25 MB gradient buckets that the reducer launches from
    inside backward on its communication stream, and the side encoder's backward -- 13 dependent exchanges -- would wait
    for each of them.  Every rank issues the collectives of EACH communicator in the same program order."""
    default = dist.distributed_c10d._get_default_group()
    if not _SYNCBN_PG or _SYNCBN_PG[0] is not default:      # (a re-initialised default group gets a new companion)
        _SYNCBN_PG[:] = [default, dist.new_group(backend=dist.get_backend())]
    return _SYNCBN_PG[1]


# ------------------------------------------------------------------------------------------------ units
def _conv(x, imgs, H, W, conv, stride=1, sv=None):
    """x [imgs*H*W, Ci] -> ([imgs*Ho*Wo, Co], Ho, Wo)."""
    Co, Ci, k, _ = conv.weight.shape
    if k == 1:
        z = ops.linear(x, conv.weight.view(Co, Ci))
        Ho, Wo = H, W
    else:
        wf, wd = ops.pack_conv_w(conv.weight)
        Ho, Wo = ops.conv_out_size(H, W, k, k, 1, 1, stride)
        z = ops.conv_fwd(x, Ci, imgs, H, W, Ci, wf, Co, k, k, 1, 1, stride=stride)
        if sv is not None:
            sv["wd"] = wd
    if sv is not None:
        sv.update(x=x, dims=(imgs, H, W, Ci, Co, k, stride))
    return z, Ho, Wo


def _conv_bwd(dz, conv, sv, gc, need_dx=True):
    imgs, H, W, Ci, Co, k, stride = sv["dims"]
    if k == 1:
        gc.put_tensor(conv.weight, ops.matmul_tn(dz, sv["x"]).view(Co, Ci, 1, 1))
        return ops.matmul_nn(dz, conv.weight.view(Co, Ci)) if need_dx else None
    dwf = ops.conv_wgrad(dz, Co, sv["x"], Ci, imgs, H, W, Ci, Co, k, k, 1, 1, stride=stride)
    gc.put_tensor(conv.weight, ops.unpack_conv_wgrad(dwf, Co, Ci, k, k))
    if not need_dx:
        return None
    assert stride == 1, "input gradient of a strided conv is not on the path (only the image-facing conv is strided)"
    return ops.conv_dgrad(dz, Co, imgs, H, W, Co, sv["wd"], Ci, k, k, 1, 1)


def _bn(z, bn, training, relu, resid=None, sv=None):
    C = z.shape[1]
    if training:
        sums = ops.bn_stats(z, C)
        world = _sync_sums(sums)
        count = z.shape[0] * world
        mean, invstd = ops.bn_finalize(sums, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        bn.num_batches_tracked += 1
    else:
        mean, invstd, count = bn.running_mean, ops.bn_eval_invstd(bn.running_var, bn.eps), z.shape[0]
    y = ops.bn_apply(z, C, mean, invstd, bn.weight, bn.bias, relu=relu, resid=resid)
    if sv is not None:
        # BatchNorm + ReLU without a residual input: backward re-derives the ReLU mask from z (the forward's own expression,
        # bit-identical sign) and does not read y back; with a residual the post-activation y is the mask
        sv.update(z=z, y=y if (relu and resid is not None) else None, remask=relu and resid is None, mean=mean,
                  invstd=invstd, count=count)
    return y


def _bn_bwd(dy, bn, sv, gc, want_dres=False):
    z, y = sv["z"], sv["y"]
    C = z.shape[1]
    rm = sv.get("remask", False)
    sums = ops.bn_bwd_reduce(dy, z, y, C, sv["mean"], sv["invstd"], remask=(bn.weight, bn.bias) if rm else None)
    gc.put_tensor(bn.bias, sums[0].float())      # this rank's sums: the data-parallel mean is the reducer's job
    gc.put_tensor(bn.weight, sums[1].float())
    _sync_sums(sums)
    return ops.bn_bwd_apply(dy, z, y, C, sv["mean"], sv["invstd"], bn.weight, sums, sv["count"], want_dres=want_dres,
                            remask_beta=bn.bias if rm else None)


def _encoder_forward(m, img, sv):
    """img NCHW fp32 -> ([B*Hc*Wc, 256], Hc, Wc)."""
    B, Cin, H, W = img.shape
    tr = m.training and not m.norm_eval
    x = img.permute(0, 2, 3, 1).contiguous().view(B * H * W, Cin)
    S = (lambda: {}) if sv is not None else (lambda: None)
    units = []
    stride = 2
    for i in (0, 3, 6):
        sc, sb = S(), S()
        z, H, W = _conv(x, B, H, W, m.stem[i], stride=stride, sv=sc)
        x = _bn(z, m.stem[i + 1], tr, True, sv=sb)
        units.append((sc, sb))
        stride = 1
    xp, idx, Hp, Wp = ops.maxpool3x3s2_fwd(x, B, H, W, x.shape[1])
    pool = dict(idx=idx, dims=(B, H, W, x.shape[1]))
    x, H, W = xp, Hp, Wp
    blocks = []
    for blk in m.layer1:
        s = {k: S() for k in ("c1", "b1", "c2", "b2", "c3", "b3", "cd", "bd")}
        o, _, _ = _conv(x, B, H, W, blk.conv1, sv=s["c1"])
        o = _bn(o, blk.bn1, tr, True, sv=s["b1"])
        o, _, _ = _conv(o, B, H, W, blk.conv2, sv=s["c2"])
        o = _bn(o, blk.bn2, tr, True, sv=s["b2"])
        o, _, _ = _conv(o, B, H, W, blk.conv3, sv=s["c3"])
        if blk.downsample is not None:
            idn, _, _ = _conv(x, B, H, W, blk.downsample[0], sv=s["cd"])
            idn = _bn(idn, blk.downsample[1], tr, False, sv=s["bd"])
        else:
            idn = x
        x = _bn(o, blk.bn3, tr, True, resid=idn, sv=s["b3"])
        blocks.append(s)
    if sv is not None:
        sv.update(units=units, pool=pool, blocks=blocks)
    return x, H, W


def _encoder_backward(m, dfeat, sv, gc):
    dx = dfeat
    for blk, s in zip(reversed(list(m.layer1)), reversed(sv["blocks"])):
        do, dres = _bn_bwd(dx, blk.bn3, s["b3"], gc, want_dres=True)
        do = _conv_bwd(do, blk.conv3, s["c3"], gc)
        do = _bn_bwd(do, blk.bn2, s["b2"], gc)
        do = _conv_bwd(do, blk.conv2, s["c2"], gc)
        do = _bn_bwd(do, blk.bn1, s["b1"], gc)
        do = _conv_bwd(do, blk.conv1, s["c1"], gc)
        if blk.downsample is not None:
            dd = _bn_bwd(dres, blk.downsample[1], s["bd"], gc)
            dres = _conv_bwd(dd, blk.downsample[0], s["cd"], gc)
        dx = ops.add(do, dres)
    B, H, W, C = sv["pool"]["dims"]
    dx = ops.maxpool3x3s2_bwd(dx, sv["pool"]["idx"], B, H, W, C)
    for j, i in reversed(list(enumerate((0, 3, 6)))):
        sc, sb = sv["units"][j]
        dz = _bn_bwd(dx, m.stem[i + 1], sb, gc)
        dx = _conv_bwd(dz, m.stem[i], sc, gc, need_dx=(j > 0))


class _ResNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, img, *params):
        sv = {}
        x, H, W = _encoder_forward(m, img, sv)
        ctx.m, ctx.sv, ctx.params = m, sv, params
        gradsync.expect(params)
        return x.view(img.shape[0], H * W, x.shape[1])

    @staticmethod
    def backward(ctx, dfeat):
        m, sv = ctx.m, ctx.sv
        gc = _GradCollector()
        _encoder_backward(m, dfeat.contiguous().view(-1, dfeat.shape[-1]), sv, gc)
        ctx.sv = None
        gradsync.ready(ctx.params)
        return (None, None) + tuple(gc.out.get(id(p)) for p in ctx.params)


class ResNetV1c(nn.Module):
    def __init__(self, depth=101, in_channels=3, stem_channels=64, base_channels=64, num_stages=1, strides=(1,),
                 dilations=(1,), out_indices=(0,), style="pytorch", deep_stem=True, avg_down=False,
                 norm_cfg=dict(type="SyncBN", requires_grad=True), norm_eval=False, contract_dilation=True,
                 zero_init_residual=True, pretrained=None, init_cfg=None, type=None, allow_random_init=False, **kw):
        super().__init__()
        ok = (depth == 101 and num_stages == 1 and tuple(strides) == (1,) and tuple(dilations) == (1,) and
              tuple(out_indices) == (0,) and style == "pytorch" and deep_stem and not avg_down and in_channels == 3 and
              norm_cfg.get("type") in ("SyncBN", "BN") and norm_cfg.get("requires_grad", True) and not kw)
        if not ok:
            raise NotImplementedError("ResNetV1c (HIP): only the SemiVL conv_encoder configuration (depth 101, stem + "
                                      "layer1, stride 4) is implemented")
        self.norm_eval, self.pretrained, self.allow_random_init = norm_eval, pretrained, allow_random_init
        sc = stem_channels
        self.stem = nn.Sequential(
            nn.Conv2d(in_channels, sc // 2, 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(sc // 2), nn.ReLU(True),
            nn.Conv2d(sc // 2, sc // 2, 3, padding=1, bias=False), nn.BatchNorm2d(sc // 2), nn.ReLU(True),
            nn.Conv2d(sc // 2, sc, 3, padding=1, bias=False), nn.BatchNorm2d(sc), nn.ReLU(True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        planes = base_channels
        down = nn.Sequential(nn.Conv2d(sc, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4))
        self.layer1 = nn.Sequential(Bottleneck(sc, planes, down), Bottleneck(planes * 4, planes, None),
                                    Bottleneck(planes * 4, planes, None))
        self.out_channels = planes * 4
        self.zero_init_residual = zero_init_residual
        self.init_weights()

    def init_weights(self):
        """mmseg ResNet.init_weights: pretrained file when present, else kaiming (fan_out) convs, unit norms and
        zero-initialised last norm of every residual branch."""
        if isinstance(self.pretrained, str) and not os.path.exists(self.pretrained) and not self.allow_random_init:
            raise FileNotFoundError(f"pretrained weights '{self.pretrained}' not found (cwd {os.getcwd()}); pass "
                                    f"allow_random_init=True (cfg['allow_random_init']) for synthetic-weight runs")
        if isinstance(self.pretrained, str) and os.path.exists(self.pretrained):
            import warnings
            ck = torch.load(self.pretrained, map_location="cpu")
            sd = ck.get("state_dict", ck)
            res = self.load_state_dict({k.replace("backbone.", ""): v for k, v in sd.items()}, strict=False)
            if res.missing_keys:   # (the ImageNet file also holds layer2-4 / fc: unexpected keys are normal here)
                warnings.warn(f"{self.pretrained}: missing keys {res.missing_keys}")
            return
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.constant_(mod.weight, 1.0)
                nn.init.constant_(mod.bias, 0.0)
        if self.zero_init_residual:
            for blk in self.layer1:
                nn.init.constant_(blk.bn3.weight, 0.0)

    def forward_tokens(self, img):
        """[B, Hc*Wc, 256] channels-last tokens (on the autograd graph when training) and (Hc, Wc)."""
        Hc, Wc = ops.conv_out_size(*ops.conv_out_size(img.shape[2], img.shape[3], 3, 3, 1, 1, 2), 3, 3, 1, 1, 2)
        params = [p for p in self.parameters() if p.requires_grad]
        if torch.is_grad_enabled() and params:
            return _ResNetFn.apply(self, img, *params), (Hc, Wc)
        x, H, W = _encoder_forward(self, img, None)
        return x.view(img.shape[0], H * W, x.shape[1]), (H, W)

    def forward(self, img):
        """Reference signature: tuple of NCHW feature maps (one, stride 4)."""
        t, (H, W) = self.forward_tokens(img)
        return (t.view(img.shape[0], H, W, -1).permute(0, 3, 1, 2),)
