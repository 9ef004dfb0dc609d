"""MI355X-native ResNet forward path behind the reference's ops/nn/tensor API.

The arithmetic lives in ``csrc/`` (hand-written HIP for gfx950) behind the
C-ABI declared in ``include/rn_hip.h``; this package is the Python host-side
mirror of the reference's operator interface.  Importing the package is cheap;
the shared library is loaded on first use and its absence is a hard error --
there is no CPU fallback.
"""
from . import weights, preprocess  # noqa: F401  (pure-numpy helpers)

__all__ = ["weights", "preprocess"]
