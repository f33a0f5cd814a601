"""profiling target: 3 x (128^3 grid decode) in each forward precision (rocprofv3 ... -- python3 tools/bf3_run.py)"""
import os, sys, io, contextlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import DeepSdfDecoder
from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests/golden/decoder_8x512.npz"))
code = np.zeros(64, np.float32)
for mode in (False, True):
    dec.set_forward_precision(mode)
    me = MeshExtractor(dec, 64, 128)
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(3):
            me.extract_mesh_from_code(code)
