"""matdeeplearn_amd — MI355X-native message-passing engine behind MatDeepLearn's model/operator API.

Only the hot path named by BASELINE.json:north_star lives here: hand-written gfx950
HIP kernels in csrc/ behind the C ABI of include/mdl_hip.h, the ctypes binding, the PyG-shaped
operator modules, the model registry, and the data/training harness that feeds them.
"""
from . import _lib, ops, nn, models  # noqa: F401

__version__ = "0.1.0"
