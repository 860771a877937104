"""semivl_amd — MI355X-native (gfx950) SemiVL training hot path behind the reference's Python surface.

Public surface mirrors google-research/semivl: `build_model(cfg)` (model/builder.py:104-159), the `VLM` module
(model/vlm.py), loss helpers (utils/train_utils.py:19-49) and the two-branch step (semivl.py:223-345).
All tensor math is issued through libsemivl_hip.so (include/semivl_hip.h); there is no CPU or eager fallback.
"""
__version__ = "0.1.0"
