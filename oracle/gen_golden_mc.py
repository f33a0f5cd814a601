"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/mc_lewiner_volumes.npz and mc_lewiner_cells.npz: what
skimage.measure.marching_cubes_lewiner (the call of reference reconstruct/utils.py:131) returns, with the reference's own
post-processing (utils.py:134-141), on
  * volumes: the fitted decoder's SDF grid (32^3, three codes), a 24^3 sphere, a smooth field open at the borders (20^3), two
    12^3 white-noise volumes (every ambiguous configuration of the 33 cases), a 9^3 volume with grid values exactly 0;
  * 4000 single cells (2x2x2 volumes): 3000 with random corner values, a seventh of them nearly degenerate, and 1000 with the
    corner signs of the ambiguous cases (13, 7, 10, 12, 6, 4) and random magnitudes -- the sub-cases of the face and interior tests
    several times over.
scikit-image is not installed for the interpreter the tests run on; the image holds 0.18.3 (the last release with
marching_cubes_lewiner) under /opt/conda for python3.9.  This script runs on the tests' interpreter (it needs oracle/sdf_oracle.py
for the decoder volumes) and calls THAT interpreter for the marching cubes:
    python oracle/gen_golden_mc.py
The fixtures are data: input volumes and the dependency's outputs."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SKIMAGE_PYTHON = os.environ.get("QSP_SKIMAGE_PYTHON", "/opt/conda/bin/python3.9")

HELPER = r'''
import sys, warnings
import numpy as np
warnings.filterwarnings("ignore")
import skimage
from skimage import measure
src, dst = sys.argv[1], sys.argv[2]
z = np.load(src)
out = {"skimage_version": skimage.__version__}
for k in z.files:
    if k == "cells":
        vals = z[k]
        n = len(vals)
        nf = np.zeros(n, np.int32); nv = np.zeros(n, np.int32)
        faces = np.full((n, 12, 3), -1, np.int8); verts = np.zeros((n, 13, 3), np.float64)
        for i in range(n):
            try:        # the reference's call on one cell: spacing 2 / (2 - 1), origin -1
                v, f, _, _ = measure.marching_cubes_lewiner(vals[i], level=0.0, spacing=[2.0] * 3)
            except (ValueError, RuntimeError):
                continue
            v = v + np.array([-1., -1., -1.])
            nf[i], nv[i] = len(f), len(v)
            faces[i, :len(f)] = f; verts[i, :len(v)] = v
        out.update(cells_nf=nf, cells_nv=nv, cells_faces=faces, cells_verts=verts)
        continue
    vol = z[k]
    d = vol.shape[0]
    verts, faces, _, _ = measure.marching_cubes_lewiner(vol, level=0.0, spacing=[2.0 / (d - 1)] * 3)
    verts[:, 0] = -1.0 + verts[:, 0]; verts[:, 1] = -1.0 + verts[:, 1]; verts[:, 2] = -1.0 + verts[:, 2]
    out[k + "_verts"] = verts
    out[k + "_faces"] = faces
np.savez(dst, **out)
'''


def volumes():
    from oracle import sdf_oracle as so
    vols = {}
    dec = so.load_decoder_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
    rng = np.random.default_rng(0)
    n = 32
    g = so.create_voxel_grid(n).reshape(-1, 3).astype(np.float32)
    for i, scale in enumerate((0.0, 0.1, 0.3)):
        code = (scale * rng.normal(size=64)).astype(np.float32)
        vols["decoder32_%d" % i] = so.decode_sdf(dec, code, g).reshape(n, n, n).astype(np.float32)
    x = np.linspace(-1, 1, 24, dtype=np.float32)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    vols["sphere24"] = (np.sqrt((X - 0.05) ** 2 + (Y + 0.1) ** 2 + (Z - 0.02) ** 2) - 0.5).astype(np.float32)
    x = np.linspace(-1, 1, 20, dtype=np.float32)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    vols["smooth20"] = (np.sin(3 * X) * np.cos(2 * Y) + 0.5 * np.sin(4 * Z + X) - 0.1).astype(np.float32)
    vols["noise12_0"] = np.random.default_rng(0).normal(size=(12, 12, 12)).astype(np.float32)
    vols["noise12_1"] = np.random.default_rng(1).normal(size=(12, 12, 12)).astype(np.float32)
    z9 = np.random.default_rng(2).normal(size=(9, 9, 9)).astype(np.float32)
    z9[np.random.default_rng(3).uniform(size=z9.shape) < 0.15] = 0.0          # grid values exactly on the level
    vols["zeros9"] = z9
    return vols


def cells(n=3000, n_targeted=1000):
    rng = np.random.default_rng(7)
    vals = rng.normal(size=(n, 2, 2, 2)).astype(np.float32)
    k = len(vals[::7])
    vals[::7] *= rng.uniform(0.001, 1, size=(k, 2, 2, 2)).astype(np.float32)
    vals[5::50][:, 0, 0, 0] = 0.0
    # + cells whose corner SIGNS are those of the most ambiguous cases (13: two of the 256 patterns; 7, 10, 12, 6, 4), random
    # magnitudes: the rare sub-cases (13.2 - 13.5, 7.4, 10.1.2 ...) several times each
    from oracle import mc_lewiner_oracle as ml
    cases = np.array(ml.tables()["CASES"])[:, 0]
    pool = [int(i) for c, w in ((13, 6), (7, 2), (10, 2), (12, 2), (6, 2), (4, 1)) for i in np.where(cases == c)[0] for _ in range(w)]
    tv = np.zeros((n_targeted, 2, 2, 2), np.float32)
    for j in range(n_targeted):
        index = pool[rng.integers(len(pool))]
        mag = np.abs(rng.normal(size=8)) * (rng.uniform(0.01, 1, size=8) if j % 3 == 0 else 1.0)
        for c, (dx, dy, dz) in enumerate(ml.CORNER):
            tv[j, dz, dy, dx] = mag[c] if (index >> c) & 1 else -mag[c]
    return np.concatenate([vals, tv])


def main():
    vols = volumes()
    with tempfile.TemporaryDirectory() as tmp:
        src, dst, helper = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz"), os.path.join(tmp, "helper.py")
        np.savez(src, cells=cells(), **vols)
        open(helper, "w").write(HELPER)
        subprocess.run([SKIMAGE_PYTHON, helper, src, dst], check=True)
        out = dict(np.load(dst))
    ver = str(out.pop("skimage_version"))
    cell_out = {k: out.pop(k) for k in list(out) if k.startswith("cells_")}
    g = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(g, "mc_lewiner_volumes.npz"), skimage_version=ver,
                        **{"vol_" + k: v for k, v in vols.items()}, **out)
    np.savez_compressed(os.path.join(g, "mc_lewiner_cells.npz"), skimage_version=ver, vals=cells(), **cell_out)
    for k in vols:
        print(k, vols[k].shape, out[k + "_verts"].shape, out[k + "_faces"].shape)
    print("cells", len(cell_out["cells_nf"]), "with a surface", int((cell_out["cells_nf"] > 0).sum()), "skimage", ver)


if __name__ == "__main__":
    main()
