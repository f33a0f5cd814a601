"""Runs only the joint BA of a bench workload (for rocprofv3 --kernel-trace --stats): python tools/ba_only.py c4 [reps]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = bench.WORKLOADS[name]
scene = synth.make_ba_scene(2000, w["n_kf"], w["n_map"], w["n_obj"], stereo_frac=0.2)
ba = BaProblem(scene)
prof = os.environ.get("QSP_BA_PROFILE") == "1"      # event pairs around every linearisation: ~6 us of idle device each
if prof:
    ba.profile(True)
if os.environ.get("QSP_BA_ELIM") == "0":
    ba.set_object_elimination(False)   # objects inside the dense system (round-1 behaviour)
if os.environ.get("QSP_BA_DET") == "0":
    ba.set_deterministic(False)        # the atomic kernels
for r in range(reps):
    ba.set_state(scene["kf_pose"], scene["pt_xyz"], scene["obj_pose"])
    t = time.time(); t1, t2 = ba.local_joint_ba(); dt = time.time() - t
    if prof:
        st = ba.profile(True)
    print("%s rep %d: %.2f ms wall, %d+%d LM iterations, trials %s %s, chi2 %.6g -> %.6g" % (
        name, r, 1e3 * dt, len(t1["chi2"]), len(t2["chi2"]), list(t1["trials"]), list(t2["trials"]), t1["chi2"][0], t2["chi2"][-1]))
