import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import torch, numpy as np
import torch.nn.functional as F
from golden_util import *
from semivl_amd.model import vlg_head as VH
name = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
z, c = load_fixture(name)
dev = torch.device('cuda:0')
hip = build_hip(c); sd = fixture_state(z, c, hip); hip.load_state_dict(sd, strict=True); hip.to(dev)
orc = build_oracle(c); orc.load_state_dict(sd, strict=True)
batch = fixture_batch(z, c)
img = batch['img_x']
def cmp(nm, a, b):
    a = a.detach().cpu().float(); b = b.detach().cpu().float()
    print(f'{nm:28s} shape {tuple(a.shape)} max|d| {(a-b).abs().max().item():.3e}  ref max {b.abs().max().item():.3e}')
with torch.no_grad():
    rf, rg = orc.backbone(img)
    feats, g = hip.backbone.forward_tokens(img.to(dev), need_global=True)
    B = img.shape[0]; hp = c['S']//16
    for i,(a,b) in enumerate(zip(feats, rf)):
        cmp(f'backbone feat{i}', a.view(B,hp,hp,-1).permute(0,3,1,2), b)
    cmp('global', g, rg)
    # decoder with oracle feats (isolates the head)
    toks = [f.permute(0,2,3,1).reshape(B, hp*hp, -1).contiguous().to(dev) for f in rf]
    text = hip.text_feat(dev)
    chunks = []
    out, _ = VH._head_forward(hip.decode_head, ((hp,hp),(hp,hp)), None, (0.5, None), (c['S'],c['S']), text,
                              [toks[0], toks[1], toks[2]], chunks)
    assert len(chunks) == 1, 'debug_stages expects a single chunk'
    sv = chunks[0][3]
    H = orc.decode_head
    N = 21
    imgf = F.normalize(rf[-1], dim=1); tx = F.normalize(orc.text_feat.float(), dim=-1)
    sim = torch.einsum('bchw,nc->bnhw', imgf, tx)
    cmp('sim', sv['sim'].view(B,N,hp,hp), sim)
    x = sim.reshape(B*N,1,hp,hp)
    x1 = H.conv1(x)
    nhwc = lambda t, n, h: t.view(n,h,h,-1).permute(0,3,1,2)
    cmp('conv1', nhwc(sv['x1'], B*N, hp), x1)
    brs = [cv(x1) for cv in H.aspp.aspp_convs]
    cat = torch.cat(brs,1)
    cmp('aspp cat', nhwc(sv['cat'], B*N, hp), cat)
    y = H.aspp.project(cat)
    cmp('aspp project', nhwc(sv['proj']['y'], B*N, hp), y)
    x2 = x1 + y
    tpo = H.text_proj(tx)
    cmp('text_proj', sv['tp'], tpo)
    xx = x2.view(B,N,-1,hp,hp).permute(0,2,1,3,4)
    tb = tpo[None].repeat(B,1,1)
    for li,l in enumerate(H.layers):
        xx = l(xx, tb)
    xo = xx.permute(0,2,1,3,4).reshape(B*N,-1,hp,hp)
    cmp('after semtr (up1 input)', nhwc(sv['up1']['x'], B*N, hp), xo)
    skips = [p(f) for p,f in zip(H.skip_proj, [rf[1], rf[0]])]
    cmp('skip0', nhwc(sv['skip'][0]['y'], B, hp), skips[0])
    cmp('skip1', nhwc(sv['skip'][1]['y'], B, hp), skips[1])
    u1 = H.up1(xo, skips[0])
    cmp('up1 out (up2 input)', nhwc(sv['up2']['x'], B*N, 2*hp), u1)
    u2 = H.up2(u1, skips[1])
    cmp('up2 out', nhwc(sv['g4'], B*N, 4*hp), u2)
    lg = H.head(u2).view(B,N,4*hp,4*hp)
    fin = F.interpolate(lg, size=(c['S'],c['S']), mode='bilinear', align_corners=False)
    cmp('final logits', out, fin)
    full = hip(img.to(dev))
    cmp('hip(img) vs oracle(img)', full, orc(img))
    # ---- inside up2
    U = H.up2
    uu = U.up(u1)
    a = sv['up2']['a']; bq = sv['up2']['b']
    cmp('up2.convT', nhwc(a['x'], B*N, 4*hp), uu)
    sk = F.interpolate(skips[1], size=uu.shape[-2:], mode='bilinear', align_corners=True)
    cmp('up2.skip_up', nhwc(a['src2'], B, 4*hp), sk)
    catu = torch.cat([uu, sk.repeat_interleave(N, 0)], 1)
    pre = U.conv[0](catu)
    cmp('up2.conv0 pre-GN', nhwc(a['pre'], B*N, 4*hp), pre)
    y0 = U.conv[2](U.conv[1](pre))
    cmp('up2.gn0+relu', nhwc(a['y'], B*N, 4*hp), y0)
    pre1 = U.conv[3](y0)
    cmp('up2.conv1 pre-GN', nhwc(bq['pre'], B*N, 4*hp), pre1)
    cmp('up2.gn1+relu', nhwc(bq['y'], B*N, 4*hp), U.conv[5](U.conv[4](pre1)))
