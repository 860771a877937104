"""`python -m semivl_amd.tools.convert_clip_weights --src ViT-B-16.pt [--backbone]`

Same outputs as the reference's `third_party/maskclip/convert_clip_weights.py` for the ViT models, but reading a local
CLIP file instead of downloading one (`clip.load`): pretrained/clip2mmseg_<model>_clip_backbone.pth (--backbone) or
pretrained/clip2mmseg_<model>_clip_weights.pth."""
import argparse
import os

import torch

from ..checkpoint import convert_clip_visual, load_clip_archive


def main():
    ap = argparse.ArgumentParser(description="Extract and save the CLIP visual weights")
    ap.add_argument("--src", required=True, help="OpenAI CLIP archive (e.g. ViT-B-16.pt) or a state-dict file")
    ap.add_argument("--model", default="ViT16", choices=["ViT32", "ViT16", "ViT14"])
    ap.add_argument("--backbone", action="store_true",
                    help="prefix keys with 'backbone.' so the file loads directly as a backbone checkpoint")
    ap.add_argument("--out-dir", default="pretrained")
    a = ap.parse_args()
    res = convert_clip_visual(load_clip_archive(a.src), backbone=a.backbone)
    os.makedirs(a.out_dir, exist_ok=True)
    name = f"clip2mmseg_{a.model}_clip_backbone.pth" if a.backbone else f"clip2mmseg_{a.model}_clip_weights.pth"
    torch.save(res, os.path.join(a.out_dir, name))
    print(os.path.join(a.out_dir, name))


if __name__ == "__main__":
    main()
