"""CLIP ViT-B/16 dense encoder in MaskCLIP form on the HIP kernel library.

Mirror of `MaskClipVisionTransformer` / `TransformerEncoderLayer`
(/root/reference third_party/maskclip/models/backbones/maskclip_vit.py:29-144,147-603) for the configurations on
the SemiVL hot path (pre_norm, final_norm, return_clip_embed, return_qkv=True, no LoRA / prompt tokens), with the
same constructor kwargs and the same `state_dict` key schema.

The whole forward AND backward of the encoder is hand-scheduled over libsemivl_hip.so (one autograd.Function for
the region): tokens stay in [B*T, C] row-major, attention/FFN/LN backward are explicit kernel sequences, weight
gradients are written straight into the parameters' `main_grad` arena views when present.
Work that cannot change results is skipped (SURVEY App. C "algorithmic"): the v-path re-uses the block's own
in-proj, out_proj is applied to v only, and the last block's x-path is skipped unless the global embedding is
requested.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import gradsync, ops


# ------------------------------------------------------------------------------------------------ parameter holders
class _MHAParams(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dims, dims))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * dims))
        self.out_proj = nn.Linear(dims, dims)
        # nn.MultiheadAttention._reset_parameters (the decoder's transformer layers keep this init; the ViT's
        # init_weights overrides it)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)


class _AttnParams(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.attn = _MHAParams(dims)


class _FFNParams(nn.Module):
    def __init__(self, dims, hidden):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dims, hidden), nn.Identity(), nn.Identity()),
                                    nn.Linear(hidden, dims), nn.Identity())


class TransformerEncoderLayer(nn.Module):
    """Parameter container with the reference's names: ln1, attn.attn.{in_proj_*,out_proj}, ln2, ffn.layers.{0.0,1}."""

    def __init__(self, embed_dims, num_heads, feedforward_channels, eps=1e-5):
        super().__init__()
        self.embed_dims, self.num_heads, self.eps = embed_dims, num_heads, eps
        self.ln1 = nn.LayerNorm(embed_dims, eps=eps)
        self.attn = _AttnParams(embed_dims)
        self.ln2 = nn.LayerNorm(embed_dims, eps=eps)
        self.ffn = _FFNParams(embed_dims, feedforward_channels)

    def plist(self):
        a, f = self.attn.attn, self.ffn.layers
        return dict(ln1w=self.ln1.weight, ln1b=self.ln1.bias, win=a.in_proj_weight, bin=a.in_proj_bias,
                    wout=a.out_proj.weight, bout=a.out_proj.bias, ln2w=self.ln2.weight, ln2b=self.ln2.bias,
                    w1=f[0][0].weight, b1=f[0][0].bias, w2=f[1].weight, b2=f[1].bias)


class _PatchEmbedParams(nn.Module):
    def __init__(self, cin, dims, patch, bias):
        super().__init__()
        self.projection = nn.Conv2d(cin, dims, patch, stride=patch, bias=bias)


# ------------------------------------------------------------------------------------------------ grad sinks
def sink_grad(param, grad_fn, shape=None):
    """Accumulate a gradient for `param`: into its `main_grad` arena view if it has one (returns None so autograd does
    not touch it), else return the tensor to autograd.  `grad_fn(out, accumulate)` writes / accumulates the grad."""
    mg = getattr(param, "main_grad", None)
    if mg is not None:
        grad_fn(mg, True)
        return None
    out = ops.empty(*param.shape, device=param.device)
    grad_fn(out, False)
    return out


# ------------------------------------------------------------------------------------------------ block forward / backward
def block_forward(x, p, heads, eps, Bn, T, want_v, skip_x, save):
    """x [Bn*T, E] -> (x_out or None, v or None).  `save` is a dict to stash what backward needs (or None).
    Under the packed-planes GEMM path (ops.planes_eligible) every A operand is produced directly in that format: both
    LayerNorms, the FFN-1 epilogue and the fused attention kernel emit planes (ln2's output and gelu(h) exist ONLY as
    planes, the attention output too in gradient-free passes); the v slice of qkv takes the generic split pass."""
    E = x.shape[1]
    D = E // heads
    pp = ops.planes_eligible(x.shape[0], E, E)
    if pp:   # y1 in fp32 is only read by the in_proj weight gradient
        y1, st1, y1a = ops.layernorm_fwd(x, p["ln1w"], p["ln1b"], eps, planes=True, want_y=save is not None)
    else:
        y1, st1 = ops.layernorm_fwd(x, p["ln1w"], p["ln1b"], eps)
        y1a = y1
    if skip_x:
        vproj = ops.linear(y1a, p["win"][2 * E:], p["bin"][2 * E:])
        qkv = None
    else:
        qkv = ops.linear(y1a, p["win"], p["bin"])
        vproj = qkv[:, 2 * E:]
    v = None
    if save is not None:
        save.update(x=x, y1=y1, st1=st1, qkv=qkv, vproj=vproj, want_v=want_v, skip_x=skip_x, eps=eps)

    def ffn(pre, want_pre):
        """pre + FFN(LN2(pre)); returns (out, ln2 stats, saved pre-activation or None)."""
        if pp:
            _, st2_, y2a = ops.layernorm_fwd(pre, p["ln2w"], p["ln2b"], eps, planes=True, want_y=False)
        else:
            y2a, st2_ = ops.layernorm_fwd(pre, p["ln2w"], p["ln2b"], eps)
        h_pre_ = ops.empty(pre.shape[0], p["w1"].shape[0], device=x.device) if want_pre else None
        h_ = ops.linear(y2a, p["w1"], p["b1"], act=ops.ACT_GELU, preact=h_pre_, planes_only=True)   # only feeds FFN-2
        return ops.linear(h_, p["w2"], p["b2"], resid=pre), st2_, h_pre_

    if want_v:
        vo = ops.linear(ops.split_planes(vproj) if pp else vproj, p["wout"], p["bout"], resid=x)  # out_proj(v) + x   (maskclip_vit.py:115-117)
        v, st2v, hv_pre = ffn(vo, save is not None)
        if save is not None:
            save.update(vo=vo, st2v=st2v, hv_pre=hv_pre)
    xo = None
    if not skip_x:
        o_p = None
        if D == 64 and pp and ops.attention_planes_ok():
            # the kernel's epilogue emits the output as planes (the out-projection's A operand); its fp32 copy is only
            # kept for backward (dsum, out_proj weight gradient)
            o, P, o_p = ops.attention_fwd(qkv, Bn, T, heads, want_lse=save is not None, planes=True, want_out=save is not None)
        elif D == 64:
            o, P = ops.attention_fwd(qkv, Bn, T, heads, want_lse=save is not None)  # P := log-sum-exp rows
        else:  # generic head dim: batched-GEMM attention with materialised probabilities
            o, P = ops.vit_attention_fwd(qkv, Bn, T, heads, D)
        x2 = ops.linear(o_p if o_p is not None else (ops.split_planes(o) if pp else o), p["wout"], p["bout"], resid=x)
        xo, st2, h_pre = ffn(x2, save is not None)
        if save is not None:
            save.update(o=o, P=P, x2=x2, st2=st2, h_pre=h_pre)
    return xo, v


def block_backward(dxo, dv, p, s, heads, Bn, T, train_ffn_ln=False, dxo_p=None):
    """Returns (dx_in, dx_in planes or None, grads dict).  Attention projections are the trainable part of the backbone
    (vlm.py:66-67); `train_ffn_ln` additionally produces FFN / LN weight grads (decoder's SemanticTransformer).
    `dxo_p`: dxo as packed planes when the block above already produced them (its LN1 backward)."""
    x, y1 = s["x"], s["y1"]
    E = x.shape[1]
    D = E // heads
    pp = ops.planes_eligible(x.shape[0], E, E)
    g = {}
    dx_res = None
    dqkv = dqkv_p = None
    wout_parts = []  # (dy, input) pairs contributing to out_proj wgrad

    def ffn_ln_bwd(dout, dout_p, pre_ln_in, st2, h_pre, tag):
        """-> (d pre_ln_in, the same as planes or None)"""
        # (dout W2) * GELU'(h_pre), one pass; with frozen FFN weights its only consumer is the next GEMM (bf16 planes)
        a_ = dout_p if dout_p is not None else dout
        dhp = ops.matmul_nn(a_, p["w2"], dact=ops.ACT_MUL_DGELU, z=h_pre, planes_only=not train_ffn_ln)
        dy2 = ops.matmul_nn(dhp, p["w1"])
        if train_ffn_ln:
            # recompute h = gelu(h_pre) and y2 = LN(pre_ln_in) for the weight grads (cheap vs. saving them)
            y2, _ = ops.layernorm_fwd(pre_ln_in, p["ln2w"], p["ln2b"], s["eps"])
            hh = ops.gelu(h_pre)
            g.setdefault("w2", []).append((dout, hh))
            g.setdefault("b2", []).append(dout)
            g.setdefault("w1", []).append((dhp, y2))
            g.setdefault("b1", []).append(dhp)
            r_ = ops.layernorm_bwd(dy2, pre_ln_in, st2, p["ln2w"], dx_add=dout, want_wgrad=True, planes=pp)
            g.setdefault("ln2w", []).append(r_[1])
            g.setdefault("ln2b", []).append(r_[2])
            return r_[0], (r_[3] if pp else None)
        if pp:
            return ops.layernorm_bwd(dy2, pre_ln_in, st2, p["ln2w"], dx_add=dout, planes=True)
        return ops.layernorm_bwd(dy2, pre_ln_in, st2, p["ln2w"], dx_add=dout), None

    if not s["skip_x"] and dxo is not None:
        dx2, dx2p = ffn_ln_bwd(dxo, dxo_p, s["x2"], s["st2"], s["h_pre"], "x")
        do = ops.matmul_nn(dx2p if dx2p is not None else dx2, p["wout"])
        wout_parts.append((dx2, s["o"]))
        if D == 64 and pp and ops.attention_planes_ok() and not (s["want_v"] and dv is not None):
            # dqkv also as planes (in_proj's input-gradient GEMM), from the two kernels' epilogues
            dqkv, dqkv_p = ops.attention_bwd(do, s["qkv"], s["o"], s["P"], Bn, T, heads, planes=True)
        elif D == 64:
            dqkv = ops.attention_bwd(do, s["qkv"], s["o"], s["P"], Bn, T, heads)
        else:
            dqkv = ops.vit_attention_bwd(do, s["qkv"], s["P"], Bn, T, heads, D)
        dx_res = dx2
    dvproj = None
    if s["want_v"] and dv is not None:
        dvo, dvop = ffn_ln_bwd(dv, None, s["vo"], s["st2v"], s["hv_pre"], "v")
        dvproj = ops.matmul_nn(dvop if dvop is not None else dvo, p["wout"])
        wout_parts.append((dvo, s["vproj"]))
        dx_res = dvo if dx_res is None else ops.add(dx_res, dvo)
    if dqkv is not None and dvproj is not None:
        rows = dqkv.shape[0]
        ops.copy2d(dvproj, 0, rows, 0, E, dqkv, 2 * E, rows, 0, 3 * E, rows, E, accumulate=True)
        dvproj = None
    g["wout_parts"] = wout_parts
    if dqkv is not None:
        g["in_full"] = (dqkv, y1)
        dy1 = ops.matmul_nn(dqkv_p if dqkv_p is not None else dqkv, p["win"])
    elif dvproj is not None:
        g["in_v"] = (dvproj, y1)
        dy1 = ops.matmul_nn(dvproj, p["win"][2 * E:])
    else:
        return dx_res, None, g
    if train_ffn_ln:
        dx_in, dg_, db_ = ops.layernorm_bwd(dy1, x, s["st1"], p["ln1w"], dx_add=dx_res, want_wgrad=True)
        g["ln1w"], g["ln1b"] = dg_, db_
        return dx_in, None, g
    if pp:   # the block below consumes dx_in as the A operand of its d FFN-2 GEMM
        dx_in, dx_in_p = ops.layernorm_bwd(dy1, x, s["st1"], p["ln1w"], dx_add=dx_res, planes=True)
        return dx_in, dx_in_p, g
    return ops.layernorm_bwd(dy1, x, s["st1"], p["ln1w"], dx_add=dx_res), None, g


def attn_wgrads(p_mod, g, E):
    """Write the attention-projection weight grads of one block from the pieces collected by block_backward.
    Returns a dict name -> tensor|None (None when written into main_grad)."""
    a = p_mod.attn.attn
    out = {}

    def wout_fn(dst, acc):
        first = not acc
        for dy, xin in g["wout_parts"]:
            ops.matmul_tn(dy, xin, out=dst, accumulate=not first)
            first = False

    def bout_fn(dst, acc):
        first = not acc
        for dy, _ in g["wout_parts"]:
            ops.colsum(dy, out=dst, accumulate=not first)
            first = False

    if g["wout_parts"]:
        out["wout"] = sink_grad(a.out_proj.weight, wout_fn)
        out["bout"] = sink_grad(a.out_proj.bias, bout_fn)
    else:
        out["wout"] = out["bout"] = None
    if "in_full" in g:
        dqkv, y1 = g["in_full"]
        out["win"] = sink_grad(a.in_proj_weight, lambda dst, acc: ops.matmul_tn(dqkv, y1, out=dst, accumulate=acc))
        out["bin"] = sink_grad(a.in_proj_bias, lambda dst, acc: ops.colsum(dqkv, out=dst, accumulate=acc))
    elif "in_v" in g:
        dvp, y1 = g["in_v"]

        def win_fn(dst, acc):
            if not acc:
                ops.fill(dst, 0.0)
            ops.matmul_tn(dvp, y1, out=dst[2 * E:], accumulate=acc)

        def bin_fn(dst, acc):
            if not acc:
                ops.fill(dst, 0.0)
            ops.colsum(dvp, out=dst[2 * E:], accumulate=acc)

        out["win"] = sink_grad(a.in_proj_weight, win_fn)
        out["bin"] = sink_grad(a.in_proj_bias, bin_fn)
    else:
        out["win"] = out["bin"] = None
    return out


# ------------------------------------------------------------------------------------------------ encoder
class MaskClipVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, patch_bias=True, in_channels=3, embed_dims=768, num_layers=12,
                 num_heads=12, mlp_ratio=4, out_indices=-1, qkv_bias=True, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., with_cls_token=True, output_cls_token=False, norm_cfg=dict(type='LN'),
                 act_cfg=dict(type='GELU'), patch_norm=False, pre_norm=False, final_norm=False, return_qkv=False,
                 return_clip_embed=False, skip_last_attn=False, interpolate_mode='bicubic', num_fcs=2,
                 norm_eval=False, with_cp=False, pretrained=None, num_prompt_tokens=None, lora_layers=[], lora_r=4,
                 lora_scaling=1, lora_dropout=0, lora_targets='qkvo', init_cfg=None, type=None,
                 allow_random_init=False):
        super().__init__()
        if isinstance(img_size, int):
            img_size = (img_size, img_size)
        unsupported = dict(patch_bias=patch_bias, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                           drop_path_rate=drop_path_rate, patch_norm=patch_norm, skip_last_attn=skip_last_attn,
                           output_cls_token=output_cls_token, num_prompt_tokens=num_prompt_tokens,
                           lora_layers=list(lora_layers))
        bad = {k: v for k, v in unsupported.items() if v not in (False, 0, 0.0, None, [])}
        if bad or not (pre_norm and final_norm and return_clip_embed and with_cls_token and qkv_bias):
            raise NotImplementedError(f"MaskClipVisionTransformer (HIP): only the SemiVL hot-path configuration is "
                                      f"implemented (off-path options: {bad})")
        assert act_cfg.get("type") == "GELU" and norm_cfg.get("type") == "LN" and interpolate_mode == "bicubic"
        self.img_size, self.patch_size = tuple(img_size), patch_size
        self.embed_dims, self.num_layers, self.num_heads = embed_dims, num_layers, num_heads
        self.eps = norm_cfg.get("eps", 1e-5)
        self.pretrained = pretrained
        self.allow_random_init = allow_random_init   # benchmarks / tests without the CLIP file (build_model cfg key)
        self.patch_embed = _PatchEmbedParams(in_channels, embed_dims, patch_size, False)
        npatch = (img_size[0] // patch_size) * (img_size[1] // patch_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dims))
        self.pos_embed = nn.Parameter(torch.zeros(1, npatch + 1, embed_dims))
        if out_indices is None:
            self.out_indices = [num_layers]
        elif isinstance(out_indices, int):
            self.out_indices = [num_layers - 1 if out_indices == -1 else out_indices]
        else:
            self.out_indices = list(out_indices)
        self.layers = nn.ModuleList([TransformerEncoderLayer(embed_dims, num_heads, mlp_ratio * embed_dims, self.eps)
                                     for _ in range(num_layers)])
        self.ln0 = nn.LayerNorm(embed_dims, eps=self.eps)
        self.ln1 = nn.LayerNorm(embed_dims, eps=self.eps)
        self.proj = nn.Conv2d(embed_dims, 512, 1, bias=False)
        # which blocks run the v-path (maskclip_vit.py:341-355)
        self.return_qkv = [False] * num_layers
        rq = return_qkv if isinstance(return_qkv, (list, tuple)) else [return_qkv] * len(self.out_indices)
        for j, o in enumerate(self.out_indices):
            if o < num_layers:
                self.return_qkv[o] = bool(rq[j])
        self.return_qkv[num_layers - 1] = True
        for o in self.out_indices:
            if o < num_layers and not self.return_qkv[o]:
                raise NotImplementedError("out_indices without return_qkv is off the SemiVL path")
        self.init_weights()

    # -- init -----------------------------------------------------------------------------------------------
    def init_weights(self):
        """maskclip_vit.py:378-429.  The pretrained CLIP file (pretrained/clip2mmseg_ViT16_clip_backbone.pth) is
        loaded (bicubic pos-embed resize included).  A configured-but-missing file raises, like mmcv's load_checkpoint:
        a silently random backbone -- and a random frozen clip_encoder feeding garbage MaskCLIP guidance -- is never
        what a training run wants.  `pretrained=None` (or allow_random_init=True) gives the reference's own random
        init (maskclip_vit.py:416-429)."""
        import os
        import warnings
        if isinstance(self.pretrained, str) and not os.path.exists(self.pretrained) and not self.allow_random_init:
            raise FileNotFoundError(
                f"pretrained weights '{self.pretrained}' not found (cwd {os.getcwd()}); convert them with "
                f"`python -m semivl_amd.tools.convert_clip_weights`, or pass allow_random_init=True "
                f"(cfg['allow_random_init']) for synthetic-weight benchmarking")
        if isinstance(self.pretrained, str) and os.path.exists(self.pretrained):
            ck = torch.load(self.pretrained, map_location="cpu")
            sd = ck.get("state_dict", ck)
            sd = {k.replace("backbone.", ""): v for k, v in sd.items()}
            if "pos_embed" in sd and sd["pos_embed"].shape != self.pos_embed.shape:
                n = int(math.sqrt(sd["pos_embed"].shape[1] - 1))
                h, w = self.img_size
                sd["pos_embed"] = self.resize_pos_embed(sd["pos_embed"], (h // self.patch_size, w // self.patch_size),
                                                        (n, n))
            if "proj.weight" in sd and sd["proj.weight"].dim() == 2:
                sd["proj.weight"] = sd["proj.weight"][:, :, None, None]
            res = self.load_state_dict(sd, strict=False)
            if res.missing_keys or res.unexpected_keys:
                warnings.warn(f"{self.pretrained}: missing keys {res.missing_keys}, unexpected keys {res.unexpected_keys}")
            return
        with torch.no_grad():
            nn.init.trunc_normal_(self.pos_embed, std=.02)
            nn.init.trunc_normal_(self.cls_token, std=.02)
            for n, m in self.named_modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=.02)
                    if m.bias is not None:
                        if "ffn" in n:
                            nn.init.normal_(m.bias, mean=0., std=1e-6)
                        else:
                            nn.init.constant_(m.bias, 0)
                elif isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
                elif isinstance(m, nn.LayerNorm):
                    nn.init.constant_(m.weight, 1.0)
                    nn.init.constant_(m.bias, 0.)
                elif isinstance(m, _MHAParams):
                    nn.init.xavier_uniform_(m.in_proj_weight)
                    nn.init.constant_(m.in_proj_bias, 0.)

    @staticmethod
    def resize_pos_embed(pos_embed, hw, pos_hw):
        """maskclip_vit.py:460-490 (init-time / off-size inputs; host-side bicubic, not on the step's hot path)."""
        ph, pw = pos_hw
        cls_w = pos_embed[:, 0:1]
        w = pos_embed[:, -ph * pw:].reshape(1, ph, pw, pos_embed.shape[2]).permute(0, 3, 1, 2)
        w = F.interpolate(w, size=hw, mode="bicubic", align_corners=False)
        return torch.cat((cls_w, w.flatten(2).transpose(1, 2)), dim=1)

    # -- forward --------------------------------------------------------------------------------------------
    def _trainable(self):
        return [p for p in self.parameters() if p.requires_grad]

    def _pos_resize_matrix(self, hw):
        """R [hw0*hw1, ph*pw] with resize_pos_embed(pos)[patches] == R @ pos[patches] (cached per target grid/device)."""
        Pz = self.patch_size
        ph, pw = self.img_size[0] // Pz, self.img_size[1] // Pz
        key = (hw, (ph, pw), str(self.pos_embed.device))
        cache = self.__dict__.setdefault("_pos_R", {})
        if key not in cache:
            with torch.no_grad():
                eye = torch.eye(ph * pw, device=self.pos_embed.device).view(ph * pw, 1, ph, pw)
                cache[key] = ops.StreamCached(
                    F.interpolate(eye, size=hw, mode="bicubic", align_corners=False).view(ph * pw, -1).t().contiguous())
        return cache[key].get()

    def forward_tokens(self, img, need_global=False):
        """Returns (feat_tokens list of [B, P, C] tensors, global or None) on the autograd graph."""
        tr = self._trainable()
        Pz = self.patch_size
        hp, wp = (img.shape[2] + Pz - 1) // Pz, (img.shape[3] + Pz - 1) // Pz
        pos_in = None
        if hp * wp + 1 != self.pos_embed.shape[1]:
            # token count differs from the trained grid (e.g. 801 -> 816 -> 51x51 vs 50x50): per-forward bicubic resize
            # (maskclip_vit.py:447-459).  Bicubic interpolation is a fixed linear map of the patch positions: it is
            # tabulated once per (grid, grid') pair as a matrix R by pushing the identity through F.interpolate, and every
            # forward is then one GEMM R @ pos (backward R^T @ d) on the library -- ATen's bicubic kernel takes 6 ms on
            # this 7.7 MB tensor, 4 x per step.
            pos_in = _PosResizeFn.apply(self.pos_embed, self._pos_resize_matrix((hp, wp)))
            tr = [p for p in tr if p is not self.pos_embed]
        if torch.is_grad_enabled() and (tr or (pos_in is not None and pos_in.requires_grad)):
            outs = _EncoderFn.apply(self, img, need_global, pos_in, *tr)
        else:
            with ops.prof_scope("vit"):
                outs = _encoder_forward(self, img, need_global, None, pos_in)
        n = len(outs) - 1
        return list(outs[:n]), outs[n]

    def forward(self, inputs):
        """Reference-compatible output: [tuple(NCHW feats), global_embedding] (maskclip_vit.py:577-596).
        The NCHW tensors are zero-copy channels-last views of the token tensors."""
        feats, g = self.forward_tokens(inputs, need_global=True)
        B = inputs.shape[0]
        hp = (inputs.shape[2] + self.patch_size - 1) // self.patch_size
        wp = (inputs.shape[3] + self.patch_size - 1) // self.patch_size
        return [tuple(f.view(B, hp, wp, f.shape[-1]).permute(0, 3, 1, 2) for f in feats), g]


def _encoder_forward(m, img, need_global, saved, pos_in=None):
    assert img.is_cuda and img.dtype == torch.float32
    img = img.contiguous()
    B, Cin, H, W = img.shape
    Pz, E, L = m.patch_size, m.embed_dims, m.num_layers
    hp, wp = (H + Pz - 1) // Pz, (W + Pz - 1) // Pz  # 'corner' padding: bottom/right zero-fill up to a patch multiple
    NP = hp * wp
    T = NP + 1
    dev = img.device
    pos = m.pos_embed[0] if pos_in is None else pos_in.detach()
    assert pos.shape[0] == T, (pos.shape, T)
    x = ops.empty(B * T, E, device=dev)
    wpe = m.patch_embed.projection.weight.view(E, Cin * Pz * Pz)
    g = ops.conv_geom(H, W, Cin, Pz, Pz, patch=Pz)
    ops.gemm(ops.A_PATCH, ops.B_KC, B * NP, E, Cin * Pz * Pz, ops.Op(img, 0), ops.Op(wpe, Cin * Pz * Pz), x, ldc_m=E,
             out_mode=ops.OUT_PATCH, ct=(NP, 0, 0), conv=g, resid=pos, ldr_m=E)
    cls_row = ops.add(m.cls_token.view(E), pos[0])
    ops.copy2d(cls_row, 0, 1, 0, 0, x, 0, 1, T * E, 0, B, E)  # x[b*T] = cls + pos[0]
    x0, st0 = ops.layernorm_fwd(x, m.ln0.weight, m.ln0.bias, m.eps)
    if saved is not None:
        saved["x_pre"], saved["st0"] = x, st0
        saved["layers"] = []
        saved["dims"] = (B, T, NP, hp, wp)
    x = x0
    feats = []
    v_last = None
    for i, layer in enumerate(m.layers):
        last = i == L - 1
        want_v = m.return_qkv[i]
        skip_x = last and not need_global
        sv = {} if saved is not None else None
        xo, v = block_forward(x, layer.plist(), m.num_heads, m.eps, B, T, want_v, skip_x, sv)
        if saved is not None:
            saved["layers"].append(sv)
        if i in m.out_indices:
            f = ops.empty(B * NP, E, device=dev)
            ops.copy2d(v, E, NP, T * E, E, f, 0, NP, NP * E, E, B * NP, E)  # v[:, 1:]
            feats.append(f.view(B, NP, E))
        if last:
            v_last = v
        x = xo
    # tail: ln1 on v, proj 1x1, channel L2-normalise (maskclip_vit.py:537-555)
    vn, stv = ops.layernorm_fwd(v_last, m.ln1.weight, m.ln1.bias, m.eps)
    vtok = ops.empty(B * NP, E, device=dev)
    ops.copy2d(vn, E, NP, T * E, E, vtok, 0, NP, NP * E, E, B * NP, E)
    wproj = m.proj.weight.view(m.proj.weight.shape[0], E)
    pe = ops.linear(vtok, wproj)
    emb, inv = ops.l2norm_fwd(pe, 0.0)
    if saved is not None:
        saved.update(v_last=v_last, stv=stv, emb=emb, inv=inv)
    if L in m.out_indices:
        feats.append(emb.view(B, NP, -1))
    glob = None
    if need_global:
        xn, _ = ops.layernorm_fwd(x, m.ln1.weight, m.ln1.bias, m.eps)
        c = ops.empty(B, E, device=dev)
        ops.copy2d(xn, 0, 1, T * E, 0, c, 0, 1, E, 0, B, E)
        gp = ops.linear(c, wproj)
        glob, _ = ops.l2norm_fwd(gp, 0.0)
    return tuple(feats) + (glob,)


class _PosResizeFn(torch.autograd.Function):
    """pos_embed [1, 1 + P, C] -> [1 + P', C] with the patch rows mapped through the tabulated bicubic matrix R [P', P]."""

    @staticmethod
    def forward(ctx, pos, R):
        ctx.R = R
        C = pos.shape[2]
        out = ops.empty(R.shape[0] + 1, C, device=pos.device)
        ops.eltwise(4, pos[0, 0].contiguous(), None, out=out[0])
        ops.matmul_nn(R, pos[0, 1:].contiguous(), out=out[1:])
        return out

    @staticmethod
    def backward(ctx, dout):
        R = ctx.R
        dout = dout.contiguous()
        C = dout.shape[1]
        dpos = ops.empty(1, R.shape[1] + 1, C, device=dout.device)
        ops.eltwise(4, dout[0].contiguous(), None, out=dpos[0, 0])
        ops.matmul_tn(R, dout[1:], out=dpos[0, 1:])
        return dpos, None


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m, img, need_global, pos_in, *params):
        saved = {}
        with ops.prof_scope("vit"):
            outs = _encoder_forward(m, img, need_global, saved, pos_in)
        ctx.m, ctx.saved, ctx.need_global = m, saved, need_global
        ctx.pos_is_input = pos_in is not None
        ctx.n_feats = len(outs) - 1
        ctx.params = params
        gradsync.expect(params)
        if outs[-1] is not None:
            ctx.mark_non_differentiable(outs[-1])  # the global embedding is a side output (unused by VLGHead)
        return outs

    @staticmethod
    def backward(ctx, *douts):
        with ops.prof_scope("vit"):
            return _EncoderFn._backward(ctx, *douts)

    @staticmethod
    def _backward(ctx, *douts):
        m, s = ctx.m, ctx.saved
        B, T, NP, hp, wp = s["dims"]
        E, L = m.embed_dims, m.num_layers
        dev = s["x_pre"].device
        dfeats = list(douts[:ctx.n_feats])
        # map feature outputs back to (layer index) order
        feat_layers = [o for o in m.out_indices if o < L]
        has_emb = L in m.out_indices
        demb = dfeats[len(feat_layers)] if has_emb else None
        # ---- tail
        dv_last = None
        if demb is not None:
            demb = demb.contiguous().view(B * NP, -1)
            dpe = ops.l2norm_bwd(demb, s["emb"], s["inv"])
            wproj = m.proj.weight.view(m.proj.weight.shape[0], E)
            dvtok = ops.matmul_nn(dpe, wproj)
            dvn = ops.zeros(B * T, E, device=dev)
            ops.copy2d(dvtok, 0, NP, NP * E, E, dvn, E, NP, T * E, E, B * NP, E)
            dv_last = ops.layernorm_bwd(dvn, s["v_last"], s["stv"], m.ln1.weight)
        dvs = {L - 1: dv_last}
        for j, li in enumerate(feat_layers):
            d = dfeats[j]
            if d is None:
                continue
            d = d.contiguous().view(B * NP, E)
            full = ops.zeros(B * T, E, device=dev)
            ops.copy2d(d, 0, NP, NP * E, E, full, E, NP, T * E, E, B * NP, E)
            dvs[li] = full if dvs.get(li) is None else ops.add(dvs[li], full)
        # ---- blocks, last to first
        dx = dxp = pend = None
        grads = {}
        for i in range(L - 1, -1, -1):
            layer = m.layers[i]
            sv = s["layers"][i]
            dx, dxp, g = block_backward(dx, dvs.get(i), layer.plist(), sv, m.num_heads, B, T, dxo_p=dxp)
            # this block's weight gradients (two split-K GEMMs, their slab reductions, two column sums) are off the
            # dependency chain: they go to the weight-gradient stream and run next to the chain's LayerNorm-backward /
            # pack passes of the blocks below (ops.wgrad_side)
            used = [t_ for pair in g["wout_parts"] for t_ in pair] + [t_ for k_ in ("in_full", "in_v") for t_ in g.get(k_, ())]
            with ops.wgrad_side(*used):
                wg = attn_wgrads(layer, g, E)
            a = layer.attn.attn
            grads[id(a.in_proj_weight)], grads[id(a.in_proj_bias)] = wg["win"], wg["bin"]
            grads[id(a.out_proj.weight)], grads[id(a.out_proj.bias)] = wg["wout"], wg["bout"]
            s["layers"][i] = None  # free this block's activations
            # this block's attention gradients are complete for this graph ONE BLOCK LATER (when its side-stream work has
            # had a block's time to finish): then their all-reduce bucket may go (train.py)
            if pend is not None:
                ops.wgrad_join(pend[0])
                gradsync.ready(pend[1])
            pend = (ops.wgrad_event(), [q for q in (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias)
                                        if any(q is r_ for r_ in ctx.params)])
        ops.wgrad_join(produced=grads.values())     # every weight gradient of this graph is complete from here on
        if pend is not None:
            gradsync.ready(pend[1])
        rest = [q for q in ctx.params if q is m.pos_embed or not any(q is r_ for l_ in m.layers for r_ in l_.parameters())]
        if dx is None:
            gradsync.ready(rest)
            ctx.saved = None
            return (None, None, None, None) + tuple(None for _ in ctx.params)
        # ---- ln0 + pos_embed
        dxpre = ops.layernorm_bwd(dx, s["x_pre"], s["st0"], m.ln0.weight)
        def pos_fn(dst, acc):
            # sum over the batch: rows b*T + t -> t
            ops.copy2d(dxpre, 0, B * T, 0, E, dst.view(-1), 0, T, 0, E, T, E, accumulate=acc)
            for b in range(1, B):
                ops.copy2d(dxpre, b * T * E, T, 0, E, dst.view(-1), 0, T, 0, E, T, E, accumulate=True)
        dpos_in = None
        if ctx.pos_is_input:  # resized pos_embed: hand the gradient back to torch autograd (bicubic backward)
            if m.pos_embed.requires_grad:
                dpos_in = ops.empty(T, E, device=dev)
                pos_fn(dpos_in, False)
        elif m.pos_embed.requires_grad:
            grads[id(m.pos_embed)] = sink_grad(m.pos_embed, pos_fn)
        gradsync.ready(rest)
        ctx.saved = None
        return (None, None, None, dpos_in) + tuple(grads.get(id(p)) for p in ctx.params)
