"""Phase stamps of k_chol_chain (library built with -DQSP_CB_STAMPS; QSP_HIP_LIB points at it): per block step the cycles of
wait + fetch (only when the tiles were not fetched ahead) | staging | W A | P^T P + store + rhs | factorisation | publish.  python tools/chain_stamps.py [c5]"""
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
print("chain form in use:", ba.cholesky_chain)
for _ in range(2):
    ba.set_state(scene["kf_pose"], scene["pt_xyz"], scene["obj_pose"])
    ba.local_joint_ba()
out = (C.c_ulonglong * (64 * 8))()
L = _lib.lib()
assert L.qsp_debug_chain_stamps(out) == 0
ts = np.array(out[:], np.int64).reshape(64, 8)
used = [j for j in range(64) if ts[j, 0]]
print("step:  fetch  stage   W A    P^TP  factor publish | total (cycles of the stamp counter)")
for j in used:
    d = np.diff(ts[j, :7])
    if j == 0:
        d[3] = ts[j, 4] - ts[j, 0]
        d[:3] = 0
    print("%3d: %6d %6d %6d %6d %6d %6d | %6d" % (j, *d, ts[j, 6] - ts[j, 0]))
print("all steps: %d cycles" % (ts[max(used), 6] - ts[min(used), 0]))
