"""Runs only the linearisation-bandwidth graph of bench.py (64 key-frames x 250 000 landmarks x 8 observations) for
rocprofv3 --kernel-trace --stats: python tools/ba_large_only.py [reps]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
big = synth.make_ba_scene_large(7, 64, 250000)
bb = BaProblem(big)
bb.profile(True)
for _ in range(reps):
    bb.set_state(big["kf_pose"], big["pt_xyz"], big["obj_pose"])
    bb.optimize(2, 0, 0, 0)
    st = bb.profile(True)
    print("linearise: %.1f us per build (%d builds)" % (1e3 * st.ms_linearize / max(st.n_linearize, 1), st.n_linearize))
bb.close()
