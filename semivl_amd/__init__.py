"""semivl_amd — MI355X-native (gfx950) SemiVL training hot path behind the reference's Python surface.

Public surface mirrors google-research/semivl: `build_model(cfg)` (model/builder.py:104-159), the `VLM` module
(model/vlm.py), loss helpers (utils/train_utils.py:19-49) and the two-branch step (semivl.py:223-345).
All tensor math is issued through libsemivl_hip.so (include/semivl_hip.h); there is no CPU or eager fallback.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def multi_rank_defaults(force=False):
    """Process-environment defaults of a rank of a multi-GPU job; called once at import, callable explicitly by a launcher
    (`force=True`: also for WORLD_SIZE = 1, what `bench.py --as-multi` measures).

    * GPU_MAX_HW_QUEUES=8 for WORLD_SIZE > 1: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES (default 4)
      hardware queues in creation order and serialises streams that share one; a rank owns main, second, helper and
      communication streams, so with 4 the bucketed gradient all-reduce can queue BEHIND the backward kernels it is meant to
      overlap.
    * HSA_ENABLE_IPC_MODE_LEGACY=0: MACHINE-SPECIFIC -- this pool's host driver only supports dmabuf IPC (RCCL and
      cross-process tensor sharing fail with `hipIpcGetMemHandle: invalid argument` without it); harmless elsewhere.

    Both are read by the runtime when it initialises (first GPU call): an explicit setting always wins (`setdefault`), and
    a warning says so when the runtime is already up and the values set here can no longer take effect.  Returns the
    variables this call set."""
    changed = {}
    if force or int(_os.environ.get("WORLD_SIZE", "1")) > 1:
        if "GPU_MAX_HW_QUEUES" not in _os.environ:
            _os.environ["GPU_MAX_HW_QUEUES"] = changed["GPU_MAX_HW_QUEUES"] = "8"
    if "HSA_ENABLE_IPC_MODE_LEGACY" not in _os.environ:
        _os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = changed["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch = _sys.modules.get("torch")
    if changed and torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        import warnings
        warnings.warn(f"semivl_amd: {sorted(changed)} set after the HIP runtime initialised -- they take effect only in "
                      f"processes started from here; export them before the first GPU call (or import semivl_amd first)")
    return changed


multi_rank_defaults()
