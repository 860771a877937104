"""CPU: the oracle restatement (oracle/semivl_oracle.py) against the golden vectors captured from the REFERENCE's own
modules (tests/golden/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from golden_util import build_oracle, fixture_batch, fixture_fp_masks, fixture_state, load_fixture


@pytest.mark.parametrize("name", ["tiny", "vlgdim", "offsize", "skr", "conf"])
def test_oracle_reproduces_reference_step(name):
    from oracle import semivl_oracle as O
    z, c = load_fixture(name)
    torch.set_num_threads(8)
    orc = build_oracle(c)
    orc.load_state_dict(fixture_state(z, c, orc), strict=True)
    batch = fixture_batch(z, c)
    masks = fixture_fp_masks(z, c)
    iters, total = [int(v) for v in z["iters"]]
    loss, aux = O.semivl_step(orc, batch, iters, total, conf_thresh=c["conf_thresh"],
                              conf_mode=c.get("conf_mode", "pixelwise"), fp_masks=masks)
    loss.backward()
    assert abs(loss.item() - float(z["loss"])) < 1e-6
    for k in ("loss_x", "loss_s1", "loss_s2", "loss_fp", "loss_mc_s1", "loss_mc_s2", "loss_mc_fp"):
        assert abs(aux[k].item() - float(z[k])) < 1e-6, k
    for k in ("mask_w", "mask_w_other", "mclip", "mclip_other"):
        assert np.array_equal(aux[k].numpy().astype(np.uint8), z[k]), k  # label indexing is bit-exact
    assert np.abs(aux["conf_w"].numpy() - z["conf_w"]).max() < 1e-6
    assert np.abs(aux["pred_x"].detach()[:, :, ::4, ::4].numpy() - z["pred_x_s4"]).max() < 1e-4
    grads = {k: p.grad for k, p in orc.named_parameters() if p.grad is not None}
    assert sorted(grads) == [str(s) for s in z["grad_names"]]
    for k, g in grads.items():
        ref = z["gnorm/" + k]
        assert abs(g.norm().item() - ref[0]) <= 1e-4 * max(ref[0], 1e-6) + 1e-9, k
    # optimizer: mmcv-style groups + one AdamW step (exp-40 hyper-parameters)
    ck = dict(backbone=dict(lr_mult=0.01), text_encoder=dict(lr_mult=0.0), conv_encoder=dict(lr_mult=1.0),
              norm=dict(decay_mult=0.0), ln=dict(decay_mult=0.0), head=dict(lr_mult=10.0))
    groups = [g for g in O.param_groups(orc, 1e-4, 0.01, ck) if g["params"][0].grad is not None]
    assert [g["name"] for g in groups] == [str(s) for s in z["opt_group_names"]]
    assert np.allclose([g["lr"] for g in groups], z["opt_group_lr"]) and np.allclose(
        [g["weight_decay"] for g in groups], z["opt_group_wd"])
    names = [g.pop("name") for g in groups]
    lrs = {k: g["lr"] for k, g in zip(names, groups)}
    torch.optim.AdamW(groups, lr=1e-4, weight_decay=0.01).step()
    sd = orc.state_dict()
    for k in names:
        ref = z["after/" + k]
        # Adam's first step moves every element by lr * sign(g): entries whose gradient is pure rounding noise (the key
        # bias of an attention layer has an exactly-zero gradient in exact arithmetic) may go either way
        noise = int((grads[k].abs() < 1e-6 * grads[k].abs().max()).sum())
        assert abs(sd[k].double().sum().item() - ref[0]) < 1e-5 * max(1.0, abs(ref[0])) + 2.0 * lrs[k] * noise, k


def test_oracle_eval_and_maskclip():
    z, c = load_fixture("tiny")
    orc = build_oracle(c)
    orc.load_state_dict(fixture_state(z, c, orc), strict=True)
    batch = fixture_batch(z, c)
    orc.eval()
    with torch.no_grad():
        out = orc(batch["img_x"])
        mc = orc.forward_maskclip(batch["img_x"], 0.9)
    assert np.abs(out[:, :, ::4, ::4].numpy() - z["logits_eval_s4"]).max() < 1e-4
    assert np.array_equal(mc.numpy().astype(np.uint8), z["mclip_x"])
