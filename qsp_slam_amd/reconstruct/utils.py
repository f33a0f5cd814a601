"""Config / decoder helpers with the reference's names (reconstruct/utils.py:82-95, deep_sdf/workspace.py:202-224)."""
import json
import os

from ..decoder import DeepSdfDecoder


class ForceKeyErrorDict(dict):
    """Attribute-style dict whose missing keys raise KeyError (reference: `class ForceKeyErrorDict(addict.Dict)` with
    `__missing__` raising, reconstruct/utils.py:82-84).  Nested dicts are wrapped on construction, as addict does."""

    def __init__(self, *args, **kwargs):
        dict.__init__(self)
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, ForceKeyErrorDict):
            v = ForceKeyErrorDict(**v)
        dict.__setitem__(self, k, v)

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return self[k]

    def __setattr__(self, k, v):
        self[k] = v

    def __missing__(self, key):
        raise KeyError(key)


def get_configs(cfg_file):
    """reconstruct/utils.py:87-90"""
    with open(cfg_file) as f:
        return ForceKeyErrorDict(**json.load(f))


def get_decoder(configs, device=0):
    """reconstruct/utils.py:93-95 -> deep_sdf/workspace.py:202-224 (config_decoder): reads specs.json and
    ModelParameters/latest.pth of configs.DeepSDF_DIR and uploads the folded weights.

    One key the reference's JSON does not have is honoured if present: `"decoder_precision": "f32" | "bf16x3" | "fp16x2"`
    (DeepSdfDecoder.set_precision; absent = "f32", the exact-f32 matrix pipe).  An unknown name is an error, not a fallback."""
    dec = DeepSdfDecoder.from_experiment_dir(configs.DeepSDF_DIR, device=device)
    if "decoder_precision" in configs:
        dec.set_precision(configs["decoder_precision"])
    return dec
