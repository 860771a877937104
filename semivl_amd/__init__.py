"""semivl_amd — MI355X-native (gfx950) SemiVL training hot path behind the reference's Python surface.

Public surface mirrors google-research/semivl: `build_model(cfg)` (model/builder.py:104-159), the `VLM` module
(model/vlm.py), loss helpers (utils/train_utils.py:19-49) and the two-branch step (semivl.py:223-345).
All tensor math is issued through libsemivl_hip.so (include/semivl_hip.h); there is no CPU or eager fallback.
"""
__version__ = "0.1.0"

import os as _os

# Ranks of a multi-GPU job: one hardware queue per stream.  The HIP runtime maps a process's streams onto
# GPU_MAX_HW_QUEUES (default 4) queues in creation order and serialises streams that share one; a rank owns main, second,
# helper, communication streams and the communicator's own, so with 4 the bucketed gradient all-reduce can queue BEHIND
# the backward kernels it is meant to overlap.  Read by the runtime when it initialises (first GPU call),
# so it is set at import; an explicit setting wins.
if int(_os.environ.get("WORLD_SIZE", "1")) > 1:
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL, tensor sharing)
