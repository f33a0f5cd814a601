"""CPU: the C-ABI library loads and exports every symbol include/qsp_hip.h declares (no compute calls without a GPU),
and the product path refuses to run without a device instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "qsp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qsp_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    from qsp_slam_amd import _lib
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "libqsp_hip.so does not export %s" % s
    assert sorted(_lib.SYMBOLS) == syms, "qsp_slam_amd/_lib.py SYMBOLS out of sync with include/qsp_hip.h"
    assert L.qsp_version() == 1


def test_no_cpu_fallback_without_device(golden_dir):
    """On a box without a GPU the decoder upload must fail loudly (QSP_ERR_NO_DEVICE), never compute on the CPU."""
    from qsp_slam_amd import DeepSdfDecoder, _lib
    if _lib.lib().qsp_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(_lib.QspError) as e:
        DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    assert e.value.code == _lib.QSP_ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    import qsp_slam_amd
    pkg = os.path.dirname(qsp_slam_amd.__file__)
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "sdf_oracle" not in src and "ba_oracle" not in src, os.path.join(dp, f)


def test_force_key_error_dict_contract():
    """reconstruct/utils.py:82-84: attribute access, nested wrapping, KeyError on a missing key"""
    from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict
    d = ForceKeyErrorDict(a=1, b=dict(c=2))
    assert d.a == 1 and d.b.c == 2 and d["b"]["c"] == 2
    with pytest.raises(KeyError):
        d.missing
    with pytest.raises(KeyError):
        d["missing"]


def test_voxel_grid_matches_reference_quirk(golden_dir):
    from qsp_slam_amd.reconstruct.optimizer import create_voxel_grid
    z = np.load(os.path.join(golden_dir, "sdf_voxel_grid.npz"))
    assert np.abs(create_voxel_grid(int(z["dim"])) - z["grid"]).max() < 1e-6
