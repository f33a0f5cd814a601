"""qsp_slam_amd -- MI355X-native joint object optimisation for QSP-SLAM (hot path only).

Importing the package does not touch the GPU; the first call that needs libqsp_hip.so loads it and raises if it is
missing -- there is no CPU fallback in the product path (the CPU restatement under oracle/ is test infrastructure)."""
from . import _lib  # noqa: F401
from .decoder import DeepSdfDecoder  # noqa: F401

__all__ = ["DeepSdfDecoder", "_lib"]
