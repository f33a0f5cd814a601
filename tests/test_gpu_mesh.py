"""GPU parity tests of the device mesh extraction through the C-ABI: qsp_mesh_from_volume /
qsp_mesh_extract against oracle/mc_oracle.py.  Index work (faces, vertex order, counts) is bit-exact; vertex coordinates
are bit-exact too (same float32 operations, no contraction); the decoded volume is compared with the numpy decoder to
the decoder tolerance of tests/test_gpu_sdf.py (2e-5 abs on tanh outputs)."""
import os
import time

import numpy as np
import pytest

from oracle import mc_oracle as mo
from oracle import sdf_oracle as so
from tests.test_oracle_mesh import canonical_mesh, noise_volume, sphere_volume

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_decoder(golden_dir):
    from qsp_slam_amd import DeepSdfDecoder
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    yield d
    d.close()


def extractor(dec, dim):
    from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
    return MeshExtractor(dec, code_len=64, voxels_dim=dim)


@pytest.mark.parametrize("dim,kind", [(16, "sphere"), (32, "sphere"), (64, "sphere"), (12, "noise"), (33, "noise"),
                                      (20, "smooth")])
def test_marching_cubes_bit_exact(gpu_decoder, dim, kind):
    if kind == "sphere":
        vol = sphere_volume(dim)
    elif kind == "noise":
        vol = noise_volume(dim, dim)
    else:
        g = np.linspace(-1, 1, dim, dtype=np.float32)
        x, y, z = np.meshgrid(g, g, g, indexing="ij")
        vol = (np.sin(3 * x) * np.cos(2 * y) + 0.5 * np.sin(4 * z + x) - 0.1).astype(np.float32)   # open at the borders
    me = extractor(gpu_decoder, dim)
    v, f = me.mesh_from_volume(vol)
    ov, of = mo.marching_cubes(vol)
    assert v.dtype == np.float32 and f.dtype == np.int32
    assert v.shape == ov.shape and f.shape == of.shape
    assert np.array_equal(f, of)
    assert np.array_equal(v.view(np.uint32), ov.view(np.uint32))


def test_empty_volume_and_reuse(gpu_decoder):
    me = extractor(gpu_decoder, 8)
    v, f = me.mesh_from_volume(np.ones((8, 8, 8), np.float32))
    assert v.shape == (0, 3) and f.shape == (0, 3)
    vol = sphere_volume(8, r=0.6)
    v, f = me.mesh_from_volume(vol)                    # the same extractor grows its buffers
    ov, of = mo.marching_cubes(vol)
    assert np.array_equal(f, of) and np.array_equal(v, ov)
    v2, f2 = me.mesh_from_volume(np.full((8, 8, 8), -1.0, np.float32))
    assert len(v2) == 0 and len(f2) == 0


def test_bad_arguments(gpu_decoder):
    from qsp_slam_amd import _lib
    from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
    with pytest.raises(_lib.QspError):
        MeshExtractor(gpu_decoder, 64, 129)
    me = extractor(gpu_decoder, 8)
    with pytest.raises(ValueError):
        me.mesh_from_volume(np.zeros((4, 4, 4), np.float32))


@pytest.mark.parametrize("dim", [32, 64])
def test_extract_mesh_from_code(gpu_decoder, golden_dir, dim):
    """MeshExtractor.extract_mesh_from_code (reconstruct/optimizer.py:284-304): decoded volume vs the numpy decoder,
    mesh vs the oracle's marching cubes on the very volume the GPU decoded."""
    dec = so.load_decoder_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    rng = np.random.default_rng(dim)
    code = (0.05 * rng.standard_normal(64)).astype(np.float32)
    me = extractor(gpu_decoder, dim)
    out = me.extract_mesh_from_code(code, return_volume=True)
    vol = out["sdf_volume"]
    assert vol.shape == (dim, dim, dim)
    ref = so.decode_sdf(dec, code, so.create_voxel_grid(dim)).reshape(dim, dim, dim)
    assert np.abs(vol - ref).max() < 2e-5
    ov, of = mo.marching_cubes(vol)
    assert np.array_equal(out.faces, of)
    assert np.array_equal(out.vertices, ov)
    assert out.vertices.dtype == np.float32 and out.faces.dtype == np.int32
    assert len(of) > 100 and mo.signed_volume(ov, of) > 0
    # the plain reference entry point (no volume) returns the same mesh
    out2 = me.extract_mesh_from_code(code)
    assert np.array_equal(out2.faces, out.faces) and np.array_equal(out2.vertices, out.vertices)
    with pytest.raises(KeyError):
        out2["sdf_volume"]


def test_full_size_properties_128(gpu_decoder):
    """create_voxel_grid's default 128^3 grid (reconstruct/utils.py:98): size-independent properties only."""
    me = extractor(gpu_decoder, 128)
    t = time.time()
    out = me.extract_mesh_from_code(np.zeros(64, np.float32), return_volume=True)
    dt = time.time() - t
    vol = out["sdf_volume"]
    ins = vol < 0
    n_cross = sum(int((np.take(ins, range(127), ax) != np.take(ins, range(1, 128), ax)).sum()) for ax in range(3))
    assert len(out.vertices) == n_cross
    dup, missing = mo.directed_edge_defects(out.faces)
    assert dup == 0
    if not (ins[0].any() or ins[-1].any() or ins[:, 0].any() or ins[:, -1].any() or ins[:, :, 0].any() or ins[:, :, -1].any()):
        assert missing == 0                              # closed when the shape does not touch the grid border
    assert out.faces.min() >= 0 and out.faces.max() < len(out.vertices)
    assert dt < 5.0


@pytest.mark.parametrize("dim,kind", [(24, "sphere"), (17, "noise")])
def test_mesh_equals_oracle_after_canonical_sorting(gpu_decoder, dim, kind):
    """the order-independent comparison (vertex set + oriented triangles), which is the one that stays meaningful against
    another marching-cubes implementation's output order"""
    vol = sphere_volume(dim) if kind == "sphere" else noise_volume(dim, dim)
    v, f = extractor(gpu_decoder, dim).mesh_from_volume(vol)
    ov, of = mo.marching_cubes(vol)
    cv, cf = canonical_mesh(v, f)
    cov, cof = canonical_mesh(ov, of)
    assert np.array_equal(cv.view(np.uint32), cov.view(np.uint32)) and np.array_equal(cf, cof)
