"""Which stage of the forward depends on the batch size?  (tests/test_fullsize_gpu.py::test_forward_is_batch_invariant...)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import seeded_state
from semivl_amd import ops
from semivl_amd.model.builder import build_model
from semivl_amd.synthetic import exp40_cfg, synthetic_batch
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "16"))
cfg = exp40_cfg(B, 512, 21, "pascal")
hip = build_model(cfg)
hip.load_state_dict(seeded_state([(k, tuple(v.shape)) for k, v in hip.state_dict().items()], 5171), strict=True)
hip.to(dev).eval()
img = synthetic_batch(B, 512, 21, seed=77, device=dev)["img_w"]
for mode in (0, 6):
    ops.set_gemm_emulation(mode)
    with torch.no_grad():
        for nb in (2, 4, 8):
            fa, _ = hip.backbone.forward_tokens(img, need_global=False)
            fb, _ = hip.backbone.forward_tokens(img[:nb].contiguous(), need_global=False)
            d = [float((a[:nb] - b).abs().max()) for a, b in zip(fa, fb)]
            full, part = hip(img), hip(img[:nb].contiguous())
            print(f"mode {mode} B={B} vs {nb}: backbone feature max diffs {d}; logits max diff {float((full[:nb] - part).abs().max()):.3e}")
ops.set_gemm_emulation(0)
