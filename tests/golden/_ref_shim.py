"""Import shim used ONLY by tests/golden/gen_golden.py in the build container (where /root/reference exists).

The reference imports un-vendored packages (mmcv-full 1.4.4, mmsegmentation 0.24, timm, clip, torchvision,
tensorboard — SURVEY §8(c)).  This module registers stand-ins in sys.modules BEFORE the reference is imported:
faithful restatements for the few symbols that execute on the hot path, permissive dummies for everything that is
only touched at import time.  Nothing here is reference source; nothing here travels into the product.
"""
import importlib.machinery
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Registry:
    def __init__(self, name):
        self.name, self.mods = name, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.mods[name or cls.__name__] = cls
            return cls
        return deco(module) if module is not None else deco

    def build(self, cfg, **kw):
        cfg = dict(cfg)
        cls = self.mods[cfg.pop("type")]
        cfg.update({k: v for k, v in kw.items() if v is not None})
        return cls(**cfg)

    def get(self, k):
        return self.mods.get(k)


BACKBONES, HEADS, SEGMENTORS, LOSSES, NECKS = (_Registry(n) for n in ("backbone", "head", "segmentor", "loss", "neck"))


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        for m in self.children():
            if hasattr(m, "init_weights"):
                m.init_weights()


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    assert t == "LN", t
    return "ln" + str(postfix), nn.LayerNorm(num_features, **cfg)


class MultiheadAttention(BaseModule):
    """mmcv.cnn.bricks.transformer.MultiheadAttention: wraps nn.MultiheadAttention (seq-first) behind batch_first."""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=None, init_cfg=None,
                 batch_first=False, **kwargs):
        super().__init__(init_cfg)
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kwargs)
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, **kw):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        out = self.attn(query=query, key=key, value=value)[0]
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.dropout_layer(self.proj_drop(out))


class FFN(BaseModule):
    def __init__(self, embed_dims=256, feedforward_channels=1024, num_fcs=2, act_cfg=dict(type="ReLU", inplace=True),
                 ffn_drop=0., dropout_layer=None, add_identity=True, init_cfg=None, **kw):
        super().__init__(init_cfg)
        assert num_fcs == 2 and act_cfg["type"] == "GELU"
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))
        self.dropout_layer = nn.Identity()
        self.add_identity = add_identity

    def forward(self, x, identity=None):
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        return (x if identity is None else identity) + self.dropout_layer(out)


class PatchEmbed(BaseModule):
    """mmseg.models.utils.PatchEmbed, conv_type='Conv2d', padding='corner'."""

    def __init__(self, in_channels=3, embed_dims=768, conv_type="Conv2d", kernel_size=16, stride=16, padding="corner",
                 dilation=1, bias=True, norm_cfg=None, input_size=None, init_cfg=None):
        super().__init__(init_cfg)
        assert padding == "corner" and norm_cfg is None
        self.k = kernel_size
        self.projection = nn.Conv2d(in_channels, embed_dims, kernel_size, stride=stride, bias=bias)

    def forward(self, x):
        H, W = x.shape[-2:]
        ph, pw = (-H) % self.k, (-W) % self.k
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph))
        x = self.projection(x)
        hw = (x.shape[2], x.shape[3])
        return x.flatten(2).transpose(1, 2), hw


def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True):
    return F.interpolate(input, size, scale_factor, mode, align_corners)


class EncoderDecoder(BaseModule):
    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        self.backbone = BACKBONES.build(backbone)
        self.decode_head = HEADS.build(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes


class _Builder(types.ModuleType):
    BACKBONES, HEADS, SEGMENTORS, LOSSES, NECKS = BACKBONES, HEADS, SEGMENTORS, LOSSES, NECKS

    @staticmethod
    def build_backbone(cfg):
        return BACKBONES.build(cfg)

    @staticmethod
    def build_head(cfg):
        return HEADS.build(cfg)

    @staticmethod
    def build_segmentor(cfg, train_cfg=None, test_cfg=None):
        return SEGMENTORS.build(cfg)

    @staticmethod
    def build_loss(cfg):
        return None


class _Dummy(types.ModuleType):
    """Any attribute resolves: CamelCase -> empty nn.Module subclass, lowercase -> identity decorator / no-op."""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name[:1].isupper():
            cls = type(name, (nn.Module,), {"__init__": lambda self, *a, **k: nn.Module.__init__(self)})
            reg = _Registry(name)
            cls.register_module = reg.register_module
            setattr(self, name, cls)
            return cls

        def fn(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        return fn


class _Finder:
    """Resolves ANY not-yet-registered submodule of the stand-in roots to a permissive dummy."""
    ROOTS = ("mmcv", "mmseg", "timm", "clip", "torchvision", "cv2", "matplotlib", "tensorboard")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.ROOTS or fullname.startswith("torch.utils.tensorboard"):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Dummy(spec.name)
        return m

    def exec_module(self, module):
        pass


class ResNetV1c(nn.Module):
    """Stand-in for mmseg's un-vendored ResNetV1c: the oracle's restatement of the (depth 101, num_stages=1) instance the
    skr04 config builds, accepting mmseg's kwargs.  Pins the reference's VLM / VLGHead WIRING around it (conv_encoder
    call, skip_from_conv_feat, conv-feature perturbation, renorm) -- not the ResNet itself."""

    def __new__(cls, depth=101, num_stages=1, **kw):
        assert depth == 101 and num_stages == 1
        from oracle.semivl_oracle import ResNetV1cStage1
        return ResNetV1cStage1()


BACKBONES.register_module("ResNetV1c")(ResNetV1c)


def install():
    sys.meta_path.insert(0, _Finder())
    real = {
        "mmcv.cnn": dict(build_norm_layer=build_norm_layer),
        "mmcv.cnn.bricks.transformer": dict(FFN=FFN, MultiheadAttention=MultiheadAttention),
        "mmcv.runner": dict(BaseModule=BaseModule, ModuleList=ModuleList, _load_checkpoint=None),
        "mmseg.models.utils": dict(PatchEmbed=PatchEmbed),
        "mmseg.ops": dict(resize=resize),
        "mmseg.models.segmentors.encoder_decoder": dict(EncoderDecoder=EncoderDecoder),
    }
    names = ["mmcv", "mmcv.cnn", "mmcv.cnn.bricks", "mmcv.cnn.bricks.transformer", "mmcv.cnn.bricks.drop",
             "mmcv.cnn.utils", "mmcv.cnn.utils.weight_init", "mmcv.runner", "mmcv.utils", "mmcv.ops",
             "mmseg", "mmseg.ops", "mmseg.utils", "mmseg.core", "mmseg.models", "mmseg.models.utils",
             "mmseg.models.builder", "mmseg.models.losses", "mmseg.models.segmentors",
             "mmseg.models.segmentors.encoder_decoder", "mmseg.models.decode_heads",
             "mmseg.models.decode_heads.decode_head", "mmseg.models.backbones", "mmseg.datasets",
             "mmseg.datasets.pipelines", "mmseg.datasets.pipelines.transforms",
             "timm", "timm.models", "timm.models.layers", "timm.models.vision_transformer", "clip", "clip.model",
             "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
             "torch.utils.tensorboard", "cv2", "matplotlib", "matplotlib.pyplot", "matplotlib.patches"]
    for n in names:
        if n == "mmseg.models.builder":
            m = _Builder(n)
        else:
            m = _Dummy(n)
        m.__spec__ = importlib.machinery.ModuleSpec(n, None, is_package=True)
        for k, v in real.get(n, {}).items():
            setattr(m, k, v)
        sys.modules[n] = m
    for n in names:  # make `import mmseg; mmseg.ops.resize(...)` style attribute access resolve to the sub-modules
        if "." in n:
            parent, child = n.rsplit(".", 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], child, sys.modules[n])
    # the reference's top-level `datasets/` directory is shadowed by the HuggingFace `datasets` wheel of this image
    dc = types.ModuleType("datasets.classes")
    dc.CLASSES = {}
    sys.modules["datasets.classes"] = dc
    dp = types.ModuleType("datasets.palettes")
    dp.get_palette = lambda *a, **k: None
    sys.modules["datasets.palettes"] = dp
    # mmseg.models re-exports the builder module and registries
    sys.modules["mmseg.models"].builder = sys.modules["mmseg.models.builder"]
    wi = sys.modules["mmcv.cnn.utils.weight_init"]
    wi.trunc_normal_ = nn.init.trunc_normal_
    wi.constant_init = lambda m, val, bias=0: None
    wi.kaiming_init = lambda m, **k: None
