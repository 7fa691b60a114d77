"""revo_amd -- MI355X-native (gfx950, HIP) implementation of REVO's per-frame hot path.

Host-side mirror of the reference interface (ImgPyramidRGBD / TrackerNew /
Optimizer) over the C ABI of include/revo_hip.h.  The HIP library is loaded on
first use and its absence is a hard error (no CPU fallback exists).
"""
from .settings import (ImgPyramidSettings, OptimizerSettings, TrackerSettings,  # noqa: F401
                       ResidualInfo, PairResult)
