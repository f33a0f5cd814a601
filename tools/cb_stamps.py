"""Phase stamps of k_chol_back (library built with -DQSP_CB_STAMPS; QSP_HIP_LIB points at it): per step the cycles of
W^T y | barrier | column batches | barrier.  python tools/cb_stamps.py [c5]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from qsp_slam_amd import synth, _lib
from qsp_slam_amd.ba import BaProblem
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
w = bench.WORKLOADS[name]
scene = synth.make_ba_scene(2000, w["n_kf"], w["n_map"], w["n_obj"], stereo_frac=0.2)
ba = BaProblem(scene)
for _ in range(2):
    ba.set_state(scene["kf_pose"], scene["pt_xyz"], scene["obj_pose"])
    ba.local_joint_ba()
out = (C.c_ulonglong * (64 * 5))()
L = _lib.lib()
assert L.qsp_debug_cb_stamps(out) == 0
ts = np.array(out[:], np.int64).reshape(64, 5)
used = [j for j in range(64) if ts[j, 0]]
print("step: Wty  barrier  batches  barrier | total   (cycles of the stamp counter)")
for j in sorted(used, reverse=True):
    d = np.diff(ts[j])
    nxt = ts[j - 1, 0] - ts[j, 4] if j - 1 in used else 0
    print("%3d: %6d %6d %6d %6d | %6d  (+%d to the next step's first stamp)" % (j, d[0], d[1], d[2], d[3], ts[j, 4] - ts[j, 0], nxt))
tot = ts[min(used), 4] - ts[max(used), 0]
print("all steps: %d cycles" % tot)
