"""Host-side mirror of the reference's `reconstruct` package for the hot path only (optimizer + the config helpers it
needs).  Same names, arguments and error behaviour as reconstruct/optimizer.py and reconstruct/utils.py of the reference;
all numerics run in libqsp_hip.so."""
from .optimizer import MeshExtractor, Optimizer  # noqa: F401
from .utils import ForceKeyErrorDict, get_configs, get_decoder  # noqa: F401
