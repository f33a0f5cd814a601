"""Mesh extraction alone (MeshExtractor.extract_mesh_from_code on the fitted decoder): ms per call at 32^3 / 64^3 / 128^3 for
Lewiner's marching cubes (the default) and the table method of rounds 2-3.  Under rocprofv3 --kernel-trace --stats the kernel
table shows k_decode*, k_lew_count / k_lew_verts / k_lew_faces and the two scan kernels.   python tools/mesh_only.py [precision]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import DeepSdfDecoder
from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
import builtins
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
dec.set_precision(prec)
code = np.zeros(64, np.float32)
_print = builtins.print
for method in ("lewiner", "table"):
    for dim in (32, 64, 128):
        me = MeshExtractor(dec, 64, dim, method=method)
        builtins.print = lambda *a, **k: None          # (the mirror prints the reference's "Extract mesh takes ..." line)
        for _ in range(2):
            out = me.extract_mesh_from_code(code)
        t0 = time.perf_counter()
        for _ in range(5):
            out = me.extract_mesh_from_code(code)
        dt = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            me.mesh_from_volume(np.asarray(me.extract_sdf_grid(code)))
        builtins.print = _print
        print("%s %-8s %3d^3: %7.2f ms per extract_mesh_from_code (%d vertices, %d faces)" % (prec, method, dim, 1e3 * dt, len(out.vertices), len(out.faces)))
