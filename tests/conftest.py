import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    from tests import margins
    margins.dump()


@pytest.fixture(scope="session", autouse=True)
def _built_in_tree():
    """The HIP library and the pybind11 module are build products (git-ignored): a clean checkout builds them once here
    (hipcc cross-compiles without a GPU), exactly as __graft_entry__.build() does."""
    import glob
    import subprocess
    pkg = os.path.join(ROOT, "qsp_slam_amd")
    if not os.path.isfile(os.path.join(pkg, "libqsp_hip.so")) or not glob.glob(os.path.join(pkg, "reconstruct_hip*.so")):
        subprocess.check_call(["bash", os.path.join(pkg, "csrc", "build.sh")])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle_decoder():
    from oracle import sdf_oracle
    return sdf_oracle.load_decoder_npz(os.path.join(GOLDEN, "decoder_8x512.npz"))


@pytest.fixture(scope="session")
def sdf_isa(tmp_path_factory):
    """the gfx950 assembly of csrc/sdf_refine.hip (device side only), compiled once per session for the static ISA tests"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mfma_hazards as chk
    path = str(tmp_path_factory.mktemp("isa") / "sdf_refine.s")
    chk.compile_isa(path)
    return path
