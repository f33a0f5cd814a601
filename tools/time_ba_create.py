"""Host cost of qsp_ba_create + qsp_ba_destroy (the drop-in Optimizer creates a problem per LocalJointBundleAdjustment call).
   python tools/time_ba_create.py [c4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
w = bench.WORKLOADS[name]
scene = synth.make_ba_scene(2000, w["n_kf"], w["n_map"], w["n_obj"], stereo_frac=0.2)
for m in ("", "steps"):              # first launches of every kernel of both forms (code-object load, hardware-queue set-up)
    if m:
        os.environ["QSP_BA_CHOL"] = m
    else:
        os.environ.pop("QSP_BA_CHOL", None)
    for _ in range(2):
        b = BaProblem(scene)
        b.local_joint_ba()
        b.close()
os.environ.pop("QSP_BA_CHOL", None)
for mode in ("chain (default)", "QSP_BA_CHOL=steps"):
    if mode != "chain (default)":
        os.environ["QSP_BA_CHOL"] = "steps"
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        b = BaProblem(scene)
        b.close()
    t1 = time.perf_counter()
    b = BaProblem(scene)
    t2 = time.perf_counter()
    b.local_joint_ba()
    t3 = time.perf_counter()
    b.close()
    print("%-20s create + destroy %.3f ms   (one local joint BA on a fresh problem: %.3f ms)" % (mode, 1e3 * (t1 - t0) / n, 1e3 * (t3 - t2)))
