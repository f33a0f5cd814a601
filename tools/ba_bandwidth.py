"""Measures the BA linearisation kernels on a large graph: algorithmic bytes / HIP-event time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_pt = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
t = time.time(); sc = synth.make_ba_scene_large(7, n_kf, n_pt); print("scene %.1fs, edges %d" % (time.time() - t, len(sc["mono_pt"])))
p = BaProblem(sc); p.profile(True)
for _ in range(2):
    p.set_state(sc["kf_pose"], sc["pt_xyz"], sc["obj_pose"])
    tr = p.optimize(2, 0, 0, 0)
    st = p.profile(True)
    us = 1e3 * st.ms_linearize / max(st.n_linearize, 1)
    print("linearize %.1f us  bytes %d  -> %.1f GB/s (%.1f%% of 8 TB/s)  total %.1f ms  chi2 %s" % (us, st.bytes_linearize, st.bytes_linearize / us / 1e3, 100 * st.bytes_linearize / us / 1e3 / 8000, st.ms_total, tr["chi2"]))
