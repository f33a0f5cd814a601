"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain Python over the active cells) of the marching cubes the reference's mesh
extraction calls: skimage.measure.marching_cubes_lewiner(volume, level=0.0, spacing=[2/(n-1)]*3) followed by `+ (-1,-1,-1)`
(reference reconstruct/utils.py:120-141).  Never imported by the product path.

The dependency is absent from /root/reference (scikit-image; environment.yml does not pin it; the function exists up to 0.18).
The image holds scikit-image 0.18.3 under /opt/conda (python3.9, no torch): oracle/gen_golden_mc.py runs THAT implementation and
commits its meshes as tests/golden/mc_lewiner_*.npz; this file is pinned against them vertex for vertex and face for face, IN
ORDER (tests/test_oracle_mesh.py).

The algorithm (Lewiner, Lopes, Vieira, Tavares: "Efficient implementation of Marching Cubes' cases with topological guarantees",
JGT 2003, as implemented by scikit-image's _marching_cubes_lewiner_cy):
  * the volume's axes are (z, y, x); cells are visited z-outermost, x-innermost; a corner is "positive" when value > level;
  * cube corners 0..3 = (x,y), (x+1,y), (x+1,y+1), (x,y+1) at z, 4..7 the same at z+1; edges 0..3 the bottom ring, 4..7 the top
    ring, 8..11 the verticals, 12 the extra vertex inside the cell;
  * CASES[index] = (case 1..14, configuration); the ambiguous cases are resolved by the face test (sign of the bilinear saddle)
    and the interior test (does the positive region connect through the cell) -- `test_face`, `test_internal`;
  * a vertex is created when a triangle first refers to it, so vertices are numbered by first use; its position is the
    inverse-|value|-weighted mean of the edge's two corners (linear interpolation up to the 2.2e-16 in the weights), in
    double, stored as float32; the extra vertex is the same mean over all eight corners;
  * at the end: vertices (x,y,z) -> (z,y,x) = the volume's axis order, triangles reversed (gradient_direction='descent'),
    vertices * spacing (float64).
The look-up tables are scikit-image's own (oracle/mc_lewiner_tables.npz, written by tools/gen_lewiner_tables.py)."""
import os

import numpy as np

FLT_EPSILON = float(np.spacing(1.0))      # (scikit-image's constant of that name is the DOUBLE epsilon, 2.2e-16)
_T = None


def tables():
    global _T
    if _T is None:
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mc_lewiner_tables.npz"))
        _T = {k: z[k].astype(np.int64).tolist() for k in z.files if k != "skimage_version"}      # (nested lists: fast scalar indexing)
    return _T


def test_face(face, v):
    """v: the eight corner values minus the level.  True when the face's saddle joins the positive corners (sign per `face`)."""
    f = abs(face)
    A, B, C, D = {1: (v[0], v[4], v[5], v[1]), 2: (v[1], v[5], v[6], v[2]), 3: (v[2], v[6], v[7], v[3]),
                  4: (v[3], v[7], v[4], v[0]), 5: (v[0], v[3], v[2], v[1]), 6: (v[4], v[7], v[6], v[5])}[f]
    acbd = A * C - B * D
    if -FLT_EPSILON < acbd < FLT_EPSILON:
        return face >= 0
    return face * A * acbd >= 0


def test_internal(T, case, config, subconfig, s, v):
    At = Bt = Ct = Dt = 0.0
    if case in (4, 10):
        a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1])
        b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1])
        with np.errstate(all="ignore"):
            t = float(np.float64(-b) / np.float64(2.0 * a))      # (a == 0: +-inf or NaN, as in C)
        if t < 0 or t > 1:
            return s > 0
        At = v[0] + (v[4] - v[0]) * t
        Bt = v[3] + (v[7] - v[3]) * t
        Ct = v[2] + (v[6] - v[2]) * t
        Dt = v[1] + (v[5] - v[1]) * t
    else:
        if case == 6:
            edge = T["TEST6"][config][2]
        elif case == 7:
            edge = T["TEST7"][config][4]
        elif case == 12:
            edge = T["TEST12"][config][3]
        else:
            edge = T["TILING13_5_1"][config][subconfig][0]
        # (reference corner a -> b of the edge, then the three edges "parallel" to it, in the order Lewiner lists them)
        E = {0: (0, 1, 3, 2, 7, 6, 4, 5), 1: (1, 2, 0, 3, 4, 7, 5, 6), 2: (2, 3, 1, 0, 5, 4, 6, 7), 3: (3, 0, 2, 1, 6, 5, 7, 4),
             4: (4, 5, 7, 6, 3, 2, 0, 1), 5: (5, 6, 4, 7, 0, 3, 1, 2), 6: (6, 7, 5, 4, 1, 0, 2, 3), 7: (7, 4, 6, 5, 2, 1, 3, 0),
             8: (0, 4, 3, 7, 2, 6, 1, 5), 9: (1, 5, 0, 4, 3, 7, 2, 6), 10: (2, 6, 1, 5, 0, 4, 3, 7), 11: (3, 7, 2, 6, 1, 5, 0, 4)}[int(edge)]
        t = v[E[0]] / (v[E[0]] - v[E[1]])
        At = 0.0
        Bt = v[E[2]] + (v[E[3]] - v[E[2]]) * t
        Ct = v[E[4]] + (v[E[5]] - v[E[4]]) * t
        Dt = v[E[6]] + (v[E[7]] - v[E[6]]) * t
    test = (1 if At >= 0 else 0) + (2 if Bt >= 0 else 0) + (4 if Ct >= 0 else 0) + (8 if Dt >= 0 else 0)
    if test in (0, 1, 2, 3, 4, 6, 8, 9, 12):
        return s > 0
    if test in (7, 11, 13, 14, 15):
        return s < 0
    # test 5 / 10: two opposite "pillars" positive -- joined when the saddle of that cross-section says so.  (Lewiner's C code leaves
    # its switch here and returns s < 0 when the condition fails; scikit-image's if / elif chain ends without a value there, i.e.
    # returns false whatever s is -- measured on single-cell volumes, oracle/gen_golden_mc.py -- and that is what the reference runs.)
    if test == 5:
        return s > 0 if At * Ct - Bt * Dt < FLT_EPSILON else False
    return s > 0 if At * Ct - Bt * Dt >= FLT_EPSILON else False


def cell_triangles(T, index, v):
    """edge ids (0..12), three per triangle, of one cell: Lewiner's case switch"""
    case, config = int(T["CASES"][index][0]), int(T["CASES"][index][1])

    def tl(name, n, *sub):
        a = T[name][config]
        for s_ in sub:
            a = a[s_]
        return [int(e) for e in a[:3 * n]]
    if case == 0:
        return []
    if case == 1:
        return tl("TILING1", 1)
    if case == 2:
        return tl("TILING2", 2)
    if case == 3:
        return tl("TILING3_2", 4) if test_face(int(T["TEST3"][config]), v) else tl("TILING3_1", 2)
    if case == 4:
        return tl("TILING4_1", 2) if test_internal(T, case, config, 0, int(T["TEST4"][config]), v) else tl("TILING4_2", 6)
    if case == 5:
        return tl("TILING5", 3)
    if case == 6:
        if test_face(int(T["TEST6"][config][0]), v):
            return tl("TILING6_2", 5)
        if test_internal(T, case, config, 0, int(T["TEST6"][config][1]), v):
            return tl("TILING6_1_1", 3)
        return tl("TILING6_1_2", 9)
    if case == 7:
        sub = sum(b for k, b in ((0, 1), (1, 2), (2, 4)) if test_face(int(T["TEST7"][config][k]), v))
        if sub == 0:
            return tl("TILING7_1", 3)
        if sub == 1:
            return tl("TILING7_2", 5, 0)
        if sub == 2:
            return tl("TILING7_2", 5, 1)
        if sub == 3:
            return tl("TILING7_3", 9, 0)
        if sub == 4:
            return tl("TILING7_2", 5, 2)
        if sub == 5:
            return tl("TILING7_3", 9, 1)
        if sub == 6:
            return tl("TILING7_3", 9, 2)
        return tl("TILING7_4_2", 9) if test_internal(T, case, config, 0, int(T["TEST7"][config][3]), v) else tl("TILING7_4_1", 5)
    if case == 8:
        return tl("TILING8", 2)
    if case == 9:
        return tl("TILING9", 4)
    if case in (10, 12):
        P = "TILING10" if case == 10 else "TILING12"
        tst = T["TEST10" if case == 10 else "TEST12"][config]
        if test_face(int(tst[0]), v):
            return tl(P + "_1_1_", 4) if test_face(int(tst[1]), v) else tl(P + "_2", 8)
        if test_face(int(tst[1]), v):
            return tl(P + "_2_", 8)
        return tl(P + "_1_1", 4) if test_internal(T, case, config, 0, int(tst[2]), v) else tl(P + "_1_2", 8)
    if case == 11:
        return tl("TILING11", 4)
    if case == 13:
        sub = sum(1 << k for k in range(6) if test_face(int(T["TEST13"][config][k]), v))
        sc = int(T["SUBCONFIG13"][sub])
        if sc == 0:
            return tl("TILING13_1", 4)
        if sc <= 6:
            return tl("TILING13_2", 6, sc - 1)
        if sc <= 18:
            return tl("TILING13_3", 10, sc - 7)
        if sc <= 22:
            return tl("TILING13_4", 12, sc - 19)
        if sc <= 26:
            k = sc - 23
            if test_internal(T, case, config, k, int(T["TEST13"][config][6]), v):
                return tl("TILING13_5_1", 6, k)
            return tl("TILING13_5_2", 10, k)
        if sc <= 38:
            return tl("TILING13_3_", 10, sc - 27)
        if sc <= 44:
            return tl("TILING13_2_", 6, sc - 39)
        if sc == 45:
            return tl("TILING13_1_", 4)
        raise AssertionError("impossible case 13 sub-configuration")
    if case == 14:
        return tl("TILING14", 4)
    raise AssertionError(case)


# corner c of a cell -> (dx, dy, dz)
CORNER = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]


def marching_cubes_lewiner(volume, level=0.0):
    """-> verts (V,3) float32 in INDEX coordinates of the volume's axes (axis0, axis1, axis2), faces (F,3) int32 -- what
    skimage.measure.marching_cubes_lewiner(volume, level) returns for spacing (1,1,1), in its order."""
    T = tables()
    im = np.ascontiguousarray(volume, np.float32)
    nz, ny, nx = im.shape
    pos = im > np.float32(level)
    # cells with a sign change
    cidx = np.zeros((nz - 1, ny - 1, nx - 1), np.int64)
    for c, (dx, dy, dz) in enumerate(CORNER):
        cidx |= pos[dz:nz - 1 + dz, dy:ny - 1 + dy, dx:nx - 1 + dx].astype(np.int64) << c
    active = np.argwhere((cidx != 0) & (cidx != 255))          # C order: z, y, x ascending -- the traversal order
    verts, faces, slot = [], [], {}
    ex, ey, ez = T["EDGEX"], T["EDGEY"], T["EDGEZ"]
    for z, y, x in active:
        v = [float(im[z + dz, y + dy, x + dx]) - float(level) for dx, dy, dz in CORNER]
        for e in cell_triangles(T, int(cidx[z, y, x]), v):
            if e == 12:
                key = (3, x, y, z)
            else:
                d1 = (int(ex[e][0]), int(ey[e][0]), int(ez[e][0]))
                d2 = (int(ex[e][1]), int(ey[e][1]), int(ez[e][1]))
                lo = (min(d1[0], d2[0]), min(d1[1], d2[1]), min(d1[2], d2[2]))
                axis = 0 if d1[0] != d2[0] else (1 if d1[1] != d2[1] else 2)
                key = (axis, x + lo[0], y + lo[1], z + lo[2])
            i = slot.get(key)
            if i is None:
                if e == 12:
                    fx = fy = fz = ff = 0.0
                    for c, (dx, dy, dz) in enumerate(CORNER):
                        w = 1.0 / (FLT_EPSILON + abs(v[c]))
                        fx += dx * w
                        fy += dy * w
                        fz += dz * w
                        ff += w
                else:
                    c1 = CORNER.index(d1)
                    c2 = CORNER.index(d2)
                    w1 = 1.0 / (FLT_EPSILON + abs(v[c1]))
                    w2 = 1.0 / (FLT_EPSILON + abs(v[c2]))
                    fx, fy, fz, ff = d1[0] * w1 + d2[0] * w2, d1[1] * w1 + d2[1] * w2, d1[2] * w1 + d2[2] * w2, w1 + w2
                i = len(verts)
                slot[key] = i
                verts.append((np.float32(x + fx / ff), np.float32(y + fy / ff), np.float32(z + fz / ff)))
            faces.append(i)
    V = np.array(verts, np.float32).reshape(-1, 3)[:, ::-1].copy()      # (x,y,z) -> (z,y,x) = the volume's axis order
    F = np.array(faces, np.int32).reshape(-1, 3)[:, ::-1].copy()        # gradient_direction='descent'
    return V, F


def convert_sdf_voxels_to_mesh(volume):
    """reference reconstruct/utils.py:120-141 on a cubic (n,n,n) volume: verts float64, faces int32"""
    n = volume.shape[0]
    V, F = marching_cubes_lewiner(volume, 0.0)
    verts = V * np.r_[[2.0 / (n - 1)] * 3]          # float32 * float64 -> float64, like skimage's `vertices * np.r_[spacing]`
    return verts + np.array([-1.0, -1.0, -1.0]), F
