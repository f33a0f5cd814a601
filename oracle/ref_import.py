"""TEST INFRASTRUCTURE ONLY -- imports the read-only reference Python path (hot path A).

Used by the golden-vector generators under oracle/ (gen_golden_sdf.py, fit_decoder.py) in the
build container, where /root/reference exists.  Nothing in the product (qsp_slam_amd/) or in the
GPU-box tests imports this module: the reference cannot travel, only the committed fixtures do.

Three harness-side shims, none of which touches a reference file:
  1. `.cuda()` is hard-coded in reconstruct/loss.py, loss_utils.py, optimizer.py -> neutralised
     (there is no GPU in the build container); torch.cuda.synchronize/empty_cache -> no-ops.
  2. reconstruct/utils.py imports addict, plyfile, skimage.measure at top level (not installed):
     stub modules are injected into sys.modules.  Only marching cubes becomes unavailable.
  3. No DeepSDF weights ship with the reference: fit_decoder.py builds the published 8x512
     architecture with the reference's own Decoder class and fits it to an analytic SDF family.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("QSP_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "reconstruct", "optimizer.py"))


def _install_shims():
    import torch

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None

    if "addict" not in sys.modules:
        addict = types.ModuleType("addict")

        class Dict(dict):
            """attribute-dict with a __missing__ hook, the only addict behaviour the path uses"""

            def __init__(self, *args, **kwargs):
                super().__init__()
                for k, v in dict(*args, **kwargs).items():
                    self[k] = self._wrap(v)

            @classmethod
            def _wrap(cls, v):
                if isinstance(v, dict) and not isinstance(v, cls):
                    return cls(**v)
                return v

            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    return self.__missing__(k)

            def __setattr__(self, k, v):
                self[k] = v

            def __missing__(self, k):
                v = type(self)()
                self[k] = v
                return v

        addict.Dict = Dict
        sys.modules["addict"] = addict
    for name in ("plyfile", "skimage", "skimage.measure"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]


def import_reference():
    """returns (optimizer_module, loss_module, loss_utils_module, decoder_module, utils_module)"""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib

    opt = importlib.import_module("reconstruct.optimizer")
    loss = importlib.import_module("reconstruct.loss")
    lu = importlib.import_module("reconstruct.loss_utils")
    dec = importlib.import_module("deep_sdf.deep_sdf_decoder")
    utils = importlib.import_module("reconstruct.utils")
    return opt, loss, lu, dec, utils
