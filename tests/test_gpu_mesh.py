"""GPU parity tests of the device mesh extraction through the C-ABI: qsp_mesh_from_volume /
qsp_mesh_extract.  The default method is Lewiner's marching cubes -- what the reference calls (reconstruct/utils.py:131) --:
compared with scikit-image 0.18.3's own output (tests/golden/mc_lewiner_*.npz, oracle/gen_golden_mc.py) and with the CPU
restatement oracle/mc_lewiner_oracle.py, vertices (float64) and faces bit for bit, IN ORDER.  method="table" (rounds 2-3)
against oracle/mc_oracle.py as before.  The decoded volume is compared with the numpy decoder to the decoder tolerance of
tests/test_gpu_sdf.py (2e-5 abs on tanh outputs)."""
import os
import time

import numpy as np
import pytest

from oracle import mc_lewiner_oracle as ml
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


def extractor(dec, dim, method="table"):
    from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
    return MeshExtractor(dec, code_len=64, voxels_dim=dim, method=method)


# ---- Lewiner's marching cubes (the default): against scikit-image's output and the CPU restatement -------------------------
def test_lewiner_equals_scikit_image_on_the_golden_volumes(gpu_decoder, golden_dir):
    """vertices float64 and faces int32, the same values in the same order as skimage.measure.marching_cubes_lewiner +
    convert_sdf_voxels_to_mesh on: the decoder's 32^3 grid (three codes), a sphere, a smooth open field, two white-noise volumes
    (every ambiguous configuration), a volume with grid values exactly on the level"""
    g = np.load(os.path.join(golden_dir, "mc_lewiner_volumes.npz"))
    names = [k[4:] for k in g.files if k.startswith("vol_")]
    assert len(names) == 8
    for k in names:
        vol = g["vol_" + k]
        me = extractor(gpu_decoder, vol.shape[0], "lewiner")
        v, f = me.mesh_from_volume(vol)
        assert v.dtype == np.float64 and f.dtype == np.int32, k
        assert v.shape == g[k + "_verts"].shape and f.shape == g[k + "_faces"].shape, k
        assert np.array_equal(f, g[k + "_faces"]), k
        assert np.array_equal(v.view(np.uint64), g[k + "_verts"].view(np.uint64)), k
        ov, of = ml.convert_sdf_voxels_to_mesh(vol)
        assert np.array_equal(f, of) and np.array_equal(v, ov), k


def test_lewiner_equals_scikit_image_cell_by_cell(gpu_decoder, golden_dir):
    """~4000 single cells (random corner values, a seventh nearly degenerate, some with a corner exactly 0; 1000 with the corner
    signs of the ambiguous cases): the sub-cases of the face and interior tests of the 33 cases, decided and triangulated as
    scikit-image does"""
    g = np.load(os.path.join(golden_dir, "mc_lewiner_cells.npz"))
    me = extractor(gpu_decoder, 2, "lewiner")
    n_faces = set()
    for i in range(len(g["vals"])):
        nf, nv = int(g["cells_nf"][i]), int(g["cells_nv"][i])
        if nf == 0:
            with pytest.raises((ValueError, RuntimeError)):
                me.mesh_from_volume(g["vals"][i])
            continue
        v, f = me.mesh_from_volume(g["vals"][i])
        assert len(f) == nf and len(v) == nv, i
        assert np.array_equal(f, g["cells_faces"][i, :nf]), i
        assert np.array_equal(v, g["cells_verts"][i, :nv]), i
        n_faces.add(nf)
    assert n_faces == {1, 2, 3, 4, 5, 6, 8, 9, 10, 12}       # (every triangle count the 33 cases produce)


@pytest.mark.parametrize("dim,kind", [(64, "sphere"), (33, "noise"), (128, "smooth")])
def test_lewiner_equals_the_restatement_at_larger_sizes(gpu_decoder, dim, kind):
    if kind == "sphere":
        vol = sphere_volume(dim)
    elif kind == "noise":
        vol = noise_volume(dim, dim)
    else:
        g = np.linspace(-1, 1, dim, dtype=np.float32)
        x, y, z = np.meshgrid(g, g, g, indexing="ij")
        vol = (np.sin(3 * x) * np.cos(2 * y) + 0.5 * np.sin(4 * z + x) - 0.1).astype(np.float32)
    v, f = extractor(gpu_decoder, dim, "lewiner").mesh_from_volume(vol)
    ov, of = ml.convert_sdf_voxels_to_mesh(vol)
    assert np.array_equal(f, of) and np.array_equal(v, ov)
    dup, missing = mo.directed_edge_defects(f)
    assert dup == 0 and (missing == 0 or kind != "sphere")


def test_lewiner_empty_surfaces_raise_like_scikit_image(gpu_decoder):
    me = extractor(gpu_decoder, 8, "lewiner")
    with pytest.raises(ValueError):          # level outside the volume's range
        me.mesh_from_volume(np.ones((8, 8, 8), np.float32))
    with pytest.raises(ValueError):
        me.mesh_from_volume(np.full((8, 8, 8), -1.0, np.float32))
    vol = np.ones((8, 8, 8), np.float32)
    vol[3, 3, 3] = 0.0                       # 0 is within the range, but no corner is > 0 on one side and <= 0 ... one cell is crossed
    v, f = me.mesh_from_volume(vol)
    ov, of = ml.convert_sdf_voxels_to_mesh(vol)
    assert np.array_equal(f, of) and np.array_equal(v, ov) and len(f) == 8
    v, f = me.mesh_from_volume(sphere_volume(8, r=0.6))      # the same extractor grows its buffers
    ov, of = ml.convert_sdf_voxels_to_mesh(sphere_volume(8, r=0.6))
    assert np.array_equal(f, of) and np.array_equal(v, ov)


def test_lewiner_mesh_from_code_is_the_default(gpu_decoder, golden_dir):
    """MeshExtractor.extract_mesh_from_code as the reference calls it (reconstruct/optimizer.py:284-304): the mesh of the very
    volume the GPU decoded, as scikit-image would triangulate it"""
    from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
    me = MeshExtractor(gpu_decoder, code_len=64, voxels_dim=32)
    code = (0.05 * np.random.default_rng(5).standard_normal(64)).astype(np.float32)
    out = me.extract_mesh_from_code(code, return_volume=True)
    ov, of = ml.convert_sdf_voxels_to_mesh(out["sdf_volume"])
    assert out.vertices.dtype == np.float64 and np.array_equal(out.vertices, ov) and np.array_equal(out.faces, of)
    assert len(of) > 100 and mo.signed_volume(ov, of) > 0


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
