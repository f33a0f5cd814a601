"""CPU tests of the mesh-extraction oracle (oracle/mc_oracle.py) and of the library's generated case table.
The reference step: convert_sdf_voxels_to_mesh, reconstruct/utils.py:120-141 (skimage absent -> properties, see the
oracle's header)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mc_oracle as mo  # noqa: E402


def sphere_volume(d, r=0.5, c=(0.05, -0.1, 0.02)):
    g = np.linspace(-1, 1, d, dtype=np.float32)
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    return (np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) - r).astype(np.float32)


def noise_volume(d, seed):
    v = np.random.default_rng(seed).standard_normal((d, d, d)).astype(np.float32)
    v[0], v[-1], v[:, 0], v[:, -1], v[:, :, 0], v[:, :, -1] = 1, 1, 1, 1, 1, 1      # keep the surface closed
    return v


def test_table_all_cases():
    ntri, tri = mo.tables()
    assert ntri[0] == 0 and ntri[255] == 0 and ntri.max() <= 5
    for case in range(256):
        crossed = set()
        for e in range(12):
            c0, c1 = mo._edge_ends(e)
            if ((case >> c0) & 1) != ((case >> c1) & 1):
                crossed.add(e)
        used = set(int(x) for x in tri[case, :3 * ntri[case]])
        assert used == crossed, case                         # every crossing is a mesh vertex and nothing else is
        loops = mo.case_polygons(case)
        assert sum(len(l) for l in loops) == len(crossed)    # loops partition the crossings
        assert sum(len(l) - 2 for l in loops) == ntri[case]
        # the complement crosses the same edges (its loops may differ: ambiguous faces cut off INSIDE corners)
        assert set(e for l in mo.case_polygons(255 - case) for e in l) == crossed


def test_library_table_matches_oracle():
    from qsp_slam_amd import _lib
    L = _lib.lib()
    nt = np.zeros(256, np.int8)
    tri = np.zeros((256, 24), np.int8)
    assert L.qsp_mc_tables(nt.ctypes.data_as(C.POINTER(C.c_int8)), tri.ctypes.data_as(C.POINTER(C.c_int8))) == 0
    ont, otri = mo.tables()
    assert np.array_equal(nt, ont) and np.array_equal(tri, otri)


@pytest.mark.parametrize("d", [16, 32])
def test_sphere_mesh(d):
    vol = sphere_volume(d)
    v, f = mo.marching_cubes(vol)
    inside = vol < 0
    n_cross = sum(int((np.take(inside, range(d - 1), ax) != np.take(inside, range(1, d), ax)).sum()) for ax in range(3))
    assert len(v) == n_cross and v.dtype == np.float32 and f.dtype == np.int32
    assert mo.directed_edge_defects(f) == (0, 0)             # closed, consistently oriented
    E = 3 * len(f) // 2
    assert len(v) - E + len(f) == 2                           # a sphere
    vol_mesh = mo.signed_volume(v, f)
    assert vol_mesh > 0                                       # outward normals
    assert abs(vol_mesh - 4 / 3 * np.pi * 0.5 ** 3) < 0.03 * (32 / d) ** 2
    r = np.linalg.norm(v - np.array([0.05, -0.1, 0.02], np.float32), axis=1)
    assert np.abs(r - 0.5).max() < (2.0 / (d - 1)) ** 2       # linear interpolation of a smooth field


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_noise_volume_is_closed_and_oriented(seed):
    # white noise exercises every ambiguous configuration; the surface must stay edge-manifold and oriented
    vol = noise_volume(12, seed)
    v, f = mo.marching_cubes(vol)
    assert len(f) > 1000
    assert mo.directed_edge_defects(f) == (0, 0)
    cases = set()
    ins = vol < 0
    d = vol.shape[0]
    cs = np.zeros((d - 1,) * 3, int)
    for c in range(8):
        o = [(c >> a) & 1 for a in range(3)]
        cs |= ins[o[0]:d - 1 + o[0], o[1]:d - 1 + o[1], o[2]:d - 1 + o[2]].astype(int) << c
    cases.update(cs.reshape(-1).tolist())
    assert len(cases) > 200                                   # nearly all 256 configurations occur
    # enclosed volume = volume of the inside region up to discretisation: positive and bounded by the box
    assert 0 < mo.signed_volume(v, f) < 8.0


def test_vertices_lie_on_their_grid_edges():
    d = 10
    vol = noise_volume(d, 5)
    v, f = mo.marching_cubes(vol)
    g = (v + 1.0) / np.float32(2.0 / (d - 1))                  # back to index coordinates
    frac = np.abs(g - np.round(g))
    assert ((frac > 1e-4).sum(1) <= 1).all()                  # at most one non-integer coordinate
    assert g.min() >= 0 and g.max() <= d - 1


def test_empty_and_full_volumes():
    for val in (1.0, -1.0):
        v, f = mo.marching_cubes(np.full((8, 8, 8), val, np.float32))
        assert v.shape == (0, 3) and f.shape == (0, 3)


def test_zero_is_outside():
    vol = np.ones((4, 4, 4), np.float32)
    vol[1, 1, 1] = -1.0
    vol[2, 1, 1] = 0.0                                        # exact zero: outside, crossing at the zero itself
    v, f = mo.marching_cubes(vol)
    assert len(v) == 6 and len(f) == 8                        # an octahedron around the single inside sample
    assert mo.directed_edge_defects(f) == (0, 0)
    d = 4
    hit = np.isclose(v, np.array([2, 1, 1], np.float32) * np.float32(2.0 / (d - 1)) - 1).all(1)
    assert hit.sum() == 1


def canonical_mesh(verts, faces):
    """order-independent form of a triangle mesh: vertices sorted lexicographically (bit patterns), faces re-indexed, each
    face rotated so that its smallest index comes first (orientation kept), faces sorted.  Two meshes with the same vertex
    set and the same oriented triangles are equal in this form whatever order a marching-cubes implementation emits them
    in -- the comparison a maintainer would run against skimage's Lewiner output (reconstruct/utils.py:131), whose ORDER
    differs from this library's by construction (known deviations)."""
    verts = np.asarray(verts, np.float32)
    faces = np.asarray(faces, np.int64)
    order = np.lexsort((verts[:, 2], verts[:, 1], verts[:, 0]))
    rank = np.empty(len(verts), np.int64)
    rank[order] = np.arange(len(verts))
    f = rank[faces] if len(faces) else faces.reshape(0, 3)
    k = np.argmin(f, axis=1) if len(f) else np.zeros(0, np.int64)
    f = np.stack([np.take_along_axis(f, ((k + i) % 3)[:, None], 1)[:, 0] for i in range(3)], axis=1) if len(f) else f
    f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))] if len(f) else f
    return verts[order], f


def test_canonical_form_is_order_independent():
    vol = sphere_volume(14)
    v, f = mo.marching_cubes(vol)
    rng = np.random.default_rng(3)
    pv = rng.permutation(len(v))                       # shuffle vertices, faces and the starting corner of every face
    inv = np.empty(len(v), np.int64)
    inv[pv] = np.arange(len(v))
    f2 = inv[f][rng.permutation(len(f))]
    rot = rng.integers(0, 3, len(f2))
    f2 = np.stack([np.take_along_axis(f2, ((rot + i) % 3)[:, None], 1)[:, 0] for i in range(3)], axis=1)
    cv, cf = canonical_mesh(v, f)
    cv2, cf2 = canonical_mesh(v[pv], f2)
    assert np.array_equal(cv.view(np.uint32), cv2.view(np.uint32)) and np.array_equal(cf, cf2)
    # a flipped triangle is a different mesh
    f3 = f.copy()
    f3[0] = f3[0][::-1]
    assert not np.array_equal(canonical_mesh(v, f3)[1], cf)


def test_decoder_volumes_have_no_ambiguous_marching_cubes_faces(oracle_decoder):
    """VERDICT r3 item 8.  The reference triangulates with skimage's marching_cubes_lewiner (reconstruct/utils.py:131); what sets
    Lewiner's 33-case algorithm apart from a 256-case table are its face and interior tests, which only ever decide cells with an
    AMBIGUOUS face (two diagonally opposite corners inside, the other two outside).  The volumes this path actually meshes are the
    decoder's: smooth signed-distance fields sampled at 2/(n-1).  Counted here on the fitted decoder at three code scales: not one
    ambiguous face among the surface cells of the 32^3 grid (the 64^3 grid likewise, tools run) -- on such volumes every
    marching-cubes variant produces the same vertex set and the same surface topology; what is left unpinned against skimage is
    the choice of diagonals inside a cell's polygons and the order of vertices / faces."""
    from oracle import sdf_oracle as so
    rng = np.random.default_rng(0)
    n = 32
    g = so.create_voxel_grid(n).reshape(-1, 3).astype(np.float32)
    faces = [(0, 1, 3, 2), (4, 5, 7, 6), (0, 1, 5, 4), (2, 3, 7, 6), (0, 2, 6, 4), (1, 3, 7, 5)]
    for scale in (0.0, 0.1, 0.3):
        code = (scale * rng.normal(size=64)).astype(np.float32)
        ins = so.decode_sdf(oracle_decoder, code, g).reshape(n, n, n) < 0
        c = np.stack([ins[dx:n - 1 + dx, dy:n - 1 + dy, dz:n - 1 + dz] for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], 0)
        cnt = c.sum(0)
        surface = (cnt > 0) & (cnt < 8)
        assert surface.sum() > 40
        amb = np.zeros_like(surface)
        for f in faces:
            a, b, cc, d = [c[i] for i in f]
            amb |= (a == cc) & (b == d) & (a != b)
        assert int((amb & surface).sum()) == 0


# ---- Lewiner's marching cubes: the CPU restatement pinned by scikit-image's own output ---------------------------------------
def test_lewiner_restatement_equals_scikit_image_on_the_golden_volumes(golden_dir):
    """oracle/mc_lewiner_oracle.py against tests/golden/mc_lewiner_volumes.npz (scikit-image 0.18.3 run by oracle/gen_golden_mc.py,
    the dependency reconstruct/utils.py:131 calls): vertices float64 and faces int32 bit for bit, in scikit-image's order"""
    from oracle import mc_lewiner_oracle as ml
    g = np.load(os.path.join(golden_dir, "mc_lewiner_volumes.npz"))
    assert str(g["skimage_version"]) == "0.18.3"
    names = [k[4:] for k in g.files if k.startswith("vol_")]
    assert len(names) == 8
    for k in names:
        v, f = ml.convert_sdf_voxels_to_mesh(g["vol_" + k])
        assert v.dtype == np.float64 and f.dtype == np.int32
        assert np.array_equal(f, g[k + "_faces"]), k
        assert np.array_equal(v.view(np.uint64), g[k + "_verts"].view(np.uint64)), k


def test_lewiner_restatement_equals_scikit_image_cell_by_cell(golden_dir):
    """... and on single cells: every case and sub-case of the 33 (which tiling table a cell took is recorded and must cover every
    table), corner values exactly 0, nearly degenerate saddles"""
    from oracle import mc_lewiner_oracle as ml
    g = np.load(os.path.join(golden_dir, "mc_lewiner_cells.npz"))
    T = ml.tables()
    used = set()
    orig = ml.cell_triangles

    def spy(Tt, index, v):
        out = orig(Tt, index, v)
        used.add((Tt["CASES"][index][0], len(out) // 3))
        return out
    ml.cell_triangles = spy
    try:
        for i in range(len(g["vals"])):
            nf, nv = int(g["cells_nf"][i]), int(g["cells_nv"][i])
            if nf == 0:
                continue
            v, f = ml.convert_sdf_voxels_to_mesh(g["vals"][i])
            assert len(f) == nf and len(v) == nv, i
            assert np.array_equal(f, g["cells_faces"][i, :nf]) and np.array_equal(v, g["cells_verts"][i, :nv]), i
    finally:
        ml.cell_triangles = orig
    # (case, triangles) pairs the 33 cases can produce: all of them occur in the fixture
    # (6.1.2 -- nine triangles -- needs a false face test AND a false interior test; it did not occur in 200 000 random case-6 cells)
    want = {(1, 1), (2, 2), (3, 2), (3, 4), (4, 2), (4, 6), (5, 3), (6, 3), (6, 5), (7, 3), (7, 5), (7, 9), (8, 2), (9, 4),
            (10, 4), (10, 8), (11, 4), (12, 4), (12, 8), (13, 4), (13, 6), (13, 10), (13, 12), (14, 4)}
    assert want <= used, sorted(want - used)
    assert T["CASES"][0][0] == 0 and T["CASES"][255][0] == 0


def test_lewiner_and_the_table_method_share_their_vertices_on_smooth_volumes():
    """what rounds 2-3 shipped (oracle/mc_oracle.py: face-consistent table) against Lewiner's: the same vertices (to float32
    rounding of two different interpolation formulas) and the same number of faces on a smooth closed surface; about half of the
    triangles are the same, the others pick the other diagonal of a polygon -- the reason the table method is no longer the default"""
    from oracle import mc_lewiner_oracle as ml
    from scipy.spatial import cKDTree
    vol = sphere_volume(24)
    v, f = mo.marching_cubes(vol)
    lv, lf = ml.convert_sdf_voxels_to_mesh(vol)
    assert v.shape == lv.shape and f.shape == lf.shape
    dist, idx = cKDTree(lv).query(v.astype(np.float64))
    assert dist.max() < 5e-7 and len(set(idx.tolist())) == len(idx)

    def canon(F):
        return {tuple(np.roll(t, -int(np.argmin(t)))) for t in F.tolist()}
    same = len(canon(idx[f]) & canon(lf))
    assert 0.3 * len(f) < same < 0.8 * len(f)


def test_every_lewiner_tiling_uses_exactly_the_sign_changing_edges():
    """The premise of the device kernel's vertex numbering (csrc/mesh_lewiner.hpp): a vertex is created by the FIRST cell of the
    sweep that uses it, and the kernel takes that to be the first of the cells around the edge -- true iff every tiling a cell can
    take refers to the vertex of each of its sign-changing edges (and to no other edge).  Checked for every corner pattern against
    every tiling of its case: 728 (pattern, tiling) pairs."""
    from oracle import mc_lewiner_oracle as ml
    T = ml.tables()
    e1, e2 = [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3], [1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7]
    cand = {1: [("TILING1", 1, ())], 2: [("TILING2", 2, ())], 3: [("TILING3_1", 2, ()), ("TILING3_2", 4, ())],
            4: [("TILING4_1", 2, ()), ("TILING4_2", 6, ())], 5: [("TILING5", 3, ())],
            6: [("TILING6_1_1", 3, ()), ("TILING6_1_2", 9, ()), ("TILING6_2", 5, ())],
            7: [("TILING7_1", 3, ())] + [("TILING7_2", 5, (k,)) for k in range(3)] + [("TILING7_3", 9, (k,)) for k in range(3)]
            + [("TILING7_4_1", 5, ()), ("TILING7_4_2", 9, ())],
            8: [("TILING8", 2, ())], 9: [("TILING9", 4, ())],
            10: [("TILING10_1_1", 4, ()), ("TILING10_1_1_", 4, ()), ("TILING10_1_2", 8, ()), ("TILING10_2", 8, ()), ("TILING10_2_", 8, ())],
            11: [("TILING11", 4, ())],
            12: [("TILING12_1_1", 4, ()), ("TILING12_1_1_", 4, ()), ("TILING12_1_2", 8, ()), ("TILING12_2", 8, ()), ("TILING12_2_", 8, ())],
            13: [("TILING13_1", 4, ()), ("TILING13_1_", 4, ())] + [("TILING13_2", 6, (k,)) for k in range(6)]
            + [("TILING13_2_", 6, (k,)) for k in range(6)] + [("TILING13_3", 10, (k,)) for k in range(12)]
            + [("TILING13_3_", 10, (k,)) for k in range(12)] + [("TILING13_4", 12, (k,)) for k in range(4)]
            + [("TILING13_5_1", 6, (k,)) for k in range(4)] + [("TILING13_5_2", 10, (k,)) for k in range(4)],
            14: [("TILING14", 4, ())]}
    n = 0
    for index in range(1, 255):
        case, config = T["CASES"][index]
        crossing = {e for e in range(12) if ((index >> e1[e]) & 1) != ((index >> e2[e]) & 1)}
        for name, nt, sub in cand[case]:
            row = T[name][config]
            for s_ in sub:
                row = row[s_]
            assert {e for e in row[:3 * nt] if e != 12} == crossing, (index, name, sub)
            n += 1
    assert n == 728
