"""Golden vectors for the CLIP weight converter (build container only: needs /root/reference).

Runs the reference's own `third_party/maskclip/convert_clip_weights.py` (as __main__, both with and without
--backbone) on a synthetic OpenAI-CLIP-shaped state dict: `clip.load` is replaced by a stub that returns the synthetic
model and `torch.save` by a recorder.  Stores the inputs and the reference's outputs in tests/golden/clip_convert.npz.
"""
import os
import runpy
import sys
import types

import numpy as np
import torch

REF = "/root/reference/third_party/maskclip/convert_clip_weights.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_clip_state_dict(width=32, layers=2, patch=4, grid=3, embed=512, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).half()
    sd = {"visual.class_embedding": r(width), "visual.positional_embedding": r(1 + grid * grid, width),
          "visual.proj": r(width, embed), "visual.conv1.weight": r(width, 3, patch, patch),
          "visual.ln_pre.weight": r(width), "visual.ln_pre.bias": r(width),
          "visual.ln_post.weight": r(width), "visual.ln_post.bias": r(width)}
    for i in range(layers):
        p = f"visual.transformer.resblocks.{i}."
        sd.update({p + "attn.in_proj_weight": r(3 * width, width), p + "attn.in_proj_bias": r(3 * width),
                   p + "attn.out_proj.weight": r(width, width), p + "attn.out_proj.bias": r(width),
                   p + "ln_1.weight": r(width), p + "ln_1.bias": r(width), p + "ln_2.weight": r(width),
                   p + "ln_2.bias": r(width), p + "mlp.c_fc.weight": r(4 * width, width),
                   p + "mlp.c_fc.bias": r(4 * width), p + "mlp.c_proj.weight": r(width, 4 * width),
                   p + "mlp.c_proj.bias": r(width)})
    # text tower / scalars: must be ignored by the converter
    sd.update({"transformer.resblocks.0.attn.in_proj_weight": r(3 * width, width), "token_embedding.weight": r(7, width),
               "positional_embedding": r(5, width), "logit_scale": torch.tensor(4.6), "text_projection": r(width, embed)})
    return sd


def run_reference(sd, backbone):
    class _M:
        def state_dict(self):
            return {k: v.clone() for k, v in sd.items()}

    clip = types.ModuleType("clip")
    clip.load = lambda name, device="cpu": (_M(), None)
    saved = {}
    real_save, real_clip = torch.save, sys.modules.get("clip")
    sys.modules["clip"] = clip
    torch.save = lambda obj, path: saved.update(obj=obj, path=path)
    argv = sys.argv
    sys.argv = ["convert_clip_weights.py", "--model", "ViT16"] + (["--backbone"] if backbone else [])
    try:
        runpy.run_path(REF, run_name="__main__")
    finally:
        torch.save, sys.argv = real_save, argv
        if real_clip is None:
            del sys.modules["clip"]
        else:
            sys.modules["clip"] = real_clip
    return saved["obj"], saved["path"]


def main():
    sd = synthetic_clip_state_dict()
    out = {f"in::{k}": v.float().numpy() for k, v in sd.items()}
    bb, bb_path = run_reference(sd, True)
    for k, v in bb["state_dict"].items():
        out[f"backbone::{k}"] = v.numpy()
    al, al_path = run_reference(sd, False)
    for k, v in al["clip"].items():
        out[f"clip::{k}"] = v.numpy()
    out["proj::weight"] = al["proj"]["weight"].numpy()
    out["meta::paths"] = np.array([bb_path, al_path])
    out["meta::top_keys"] = np.array(sorted(bb.keys()) + ["|"] + sorted(al.keys()))
    np.savez_compressed(os.path.join(HERE, "clip_convert.npz"), **out)
    print("wrote clip_convert.npz:", len(bb["state_dict"]), "backbone keys,", len(al["clip"]), "clip keys;", bb_path, al_path)


if __name__ == "__main__":
    main()
