"""ctypes view of oracle/_ref/libsemivl_cpu.so -- the CPU reference backend of the C-ABI (test infrastructure).  Same
signature table as the product binding (semivl_amd.lib.SIGNATURES), restricted to what oracle/cabi_cpu.c implements."""
import ctypes as C
import os
import subprocess

import numpy as np

import semivl_amd.lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libsemivl_cpu.so")
IMPLEMENTED = ["svl_version", "svl_last_error", "svl_fill_f32", "svl_softmax_max_f32", "svl_cutmix_f32", "svl_cutmix_i64",
               "svl_count_valid_i64", "svl_ce_num_blocks", "svl_ce_fused_f32", "svl_ce_finalize", "svl_ce_up_num_blocks",
               "svl_ce_up_fused_f32", "svl_softmax_max_up_f32", "svl_semivl_gscale",
               "svl_semivl_loss", "svl_conf_avg_ws_doubles", "svl_conf_avg_factor", "svl_conf_ratio_f32", "svl_maskclip_labels",
               "svl_concept_max_f32", "svl_iou_hist_i64", "svl_adamw_step"]
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(PATH):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)
        lib = C.CDLL(PATH)
        for n in IMPLEMENTED:
            fn = getattr(lib, n)
            fn.restype, fn.argtypes = L.SIGNATURES[n]
        _lib = lib
    return _lib


def ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def check(rc, what=""):
    if rc != 0:
        buf = C.create_string_buffer(512)
        load().svl_last_error(buf, 512)
        raise RuntimeError(f"{what}: status {rc}: {buf.value.decode()}")


def ce_fused(logits, target, use_ignore_t, conf=None, ign=None, conf_thresh=0.0, mc=None, gscale=None, all_pixels=False,
             img_weight=None):
    """numpy in / out through svl_ce_fused_f32 + svl_ce_finalize of the CPU backend: (sums double[4], dlogits or None)."""
    lib = load()
    B, N = logits.shape[:2]
    HW = int(np.prod(logits.shape[2:]))
    nblk = lib.svl_ce_num_blocks(B, N, HW)
    partials = np.zeros((nblk, 4), np.float32)
    dl = np.zeros_like(logits) if gscale is not None else None
    d = L.CeDesc(ptr(logits), B, N, HW, ptr(target), int(use_ignore_t), ptr(conf), ptr(ign), float(conf_thresh),
                 int(all_pixels), ptr(mc), ptr(partials), ptr(dl), ptr(gscale), ptr(img_weight))
    check(lib.svl_ce_fused_f32(C.byref(d), None), "svl_ce_fused_f32")
    sums = np.zeros(4, np.float64)
    check(lib.svl_ce_finalize(ptr(partials), nblk, ptr(sums), None), "svl_ce_finalize")
    return sums, dl


def ce_up_fused(logits, H, W, align, target, use_ignore_t, conf=None, ign=None, conf_thresh=0.0, mc=None, gscale=None,
                all_pixels=False, img_weight=None):
    """numpy in / out through svl_ce_up_fused_f32 + svl_ce_finalize of the CPU backend: logits [B, N, h, w], maps [B, H, W];
    returns (sums double[4], dlogits [B, N, h, w] or None)."""
    lib = load()
    B, N, h, w = logits.shape
    nblk = lib.svl_ce_up_num_blocks(B, N, h, w, H, W, int(align))
    assert nblk > 0
    partials = np.zeros((nblk, 4), np.float32)
    dl = np.zeros_like(logits) if gscale is not None else None
    d = L.CeUpDesc(ptr(logits), B, N, h, w, H, W, int(align), ptr(target), int(use_ignore_t), ptr(conf), ptr(ign),
                   float(conf_thresh), int(all_pixels), ptr(mc), ptr(partials), ptr(dl), ptr(gscale), ptr(img_weight))
    check(lib.svl_ce_up_fused_f32(C.byref(d), None), "svl_ce_up_fused_f32")
    sums = np.zeros(4, np.float64)
    check(lib.svl_ce_finalize(ptr(partials), nblk, ptr(sums), None), "svl_ce_finalize")
    return sums, dl
