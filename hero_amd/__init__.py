"""hero_amd — MI355X-native (gfx950) implementation of HERO's hierarchical-encoder hot path.

Python host code keeps the reference's module API (hero_amd.model mirrors `model/`), the arithmetic
runs in hand-written HIP kernels behind a C ABI (include/hero_hip.h, hero_amd/libhero_hip.so).
"""
from .functional import (advance_seed, compute_dtype, manual_seed,  # noqa: F401
                         notify_weights_updated, set_compute_dtype)

__version__ = "0.1.0"
