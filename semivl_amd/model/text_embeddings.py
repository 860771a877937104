"""Runtime part of the reference's model/text_embeddings.py:188-215 — per-class max over concept logits.

The concept lists themselves are data of the reference (45 background concepts etc.); only the NUMBER of concepts
per class matters at run time because `flatten_class_concepts` lays concepts out contiguously per class
(text_embeddings.py:195-206).  Keyed, like the reference, by the path string of the .npy
(text_embeddings.py:208-215), suffix-matched so package-relative paths resolve too.
"""
import torch

from .. import ops

_CONCEPT_COUNTS = {
    # VOC12_wbg_classes_w_concepts4 (98 concepts -> 21 classes)
    "voc12_wbg_concept4_single.npy": [45, 3, 3, 1, 4, 4, 2, 4, 2, 3, 1, 2, 2, 4, 4, 4, 3, 1, 1, 2, 3],
    # Cityscapes_classes_w_concepts3 (54 concepts -> 19 classes)
    "cityscapes_concept3_single.npy": [3, 1, 7, 1, 2, 3, 1, 3, 3, 4, 1, 7, 3, 4, 4, 1, 2, 3, 1],
}


def get_class_to_concept_idxs(save_path):
    """dict class index -> list of concept channel indices (same return type as the reference)."""
    for key, counts in _CONCEPT_COUNTS.items():
        if str(save_path).endswith(key):
            out, o = {}, 0
            for i, c in enumerate(counts):
                out[i] = list(range(o, o + c))
                o += c
            return out
    raise ValueError(save_path)


_OFFS_CACHE = {}


def concept_offsets(class_to_concept_idxs, device):
    """int32 prefix offsets [N+1] for svl_concept_max_f32 (concepts of a class are contiguous channels).  Cached per
    (concept counts, device): building the tensor is a synchronous host-to-device copy, once per step otherwise."""
    key = (tuple(len(class_to_concept_idxs[i]) for i in range(len(class_to_concept_idxs))), str(device))
    if key in _OFFS_CACHE:
        return _OFFS_CACHE[key]
    offs = [0]
    for i in range(len(class_to_concept_idxs)):
        idx = class_to_concept_idxs[i]
        assert idx == list(range(offs[-1], offs[-1] + len(idx))), "concept channels must be contiguous per class"
        offs.append(offs[-1] + len(idx))
    _OFFS_CACHE[key] = torch.tensor(offs, dtype=torch.int32, device=device)
    return _OFFS_CACHE[key]


def aggregate_concept_predictions(pred, class_to_concept_idxs):
    """[B, n_concepts, H, W] -> [B, n_classes, H, W]: per-class max over its concept channels (HIP kernel)."""
    offs = concept_offsets(class_to_concept_idxs, pred.device)
    return ops.concept_max(pred.contiguous(), offs, len(class_to_concept_idxs))
