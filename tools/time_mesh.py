"""Times MeshExtractor.extract_mesh_from_code on the GPU (decode + marching cubes + copy-out) and the numpy oracle beside it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import DeepSdfDecoder
from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
npz = os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz")
dec = DeepSdfDecoder.from_npz(npz)
code = np.zeros(64, np.float32)
for dim in (32, 64, 128):
    me = MeshExtractor(dec, 64, dim)
    me.extract_mesh_from_code(code)
    t = time.time(); n = 5
    for _ in range(n):
        out = me.extract_mesh_from_code(code)
    dt = (time.time() - t) / n
    flop = 2 * 1835520 * dim ** 3
    print("dim %3d: %.2f ms per mesh (%d verts, %d faces), decode alone >= %.2f ms at 157 TF; %.1f TFLOP/s end to end" % (
        dim, 1e3 * dt, len(out.vertices), len(out.faces), 1e3 * flop / 157.3e12, flop / dt / 1e12))
if "--cpu" in sys.argv:
    from oracle import sdf_oracle as so, mc_oracle as mo
    d = so.load_decoder_npz(npz)
    for dim in (32, 64):
        t = time.time()
        vol = so.decode_sdf(d, code, so.create_voxel_grid(dim)).reshape(dim, dim, dim)
        t1 = time.time()
        v, f = mo.marching_cubes(vol)
        print("oracle dim %d: decode %.2f s, marching cubes %.3f s" % (dim, t1 - t, time.time() - t1))
