"""A local joint BA on a FRESH problem, several times in a row (the drop-in Optimizer's pattern): python tools/time_ba_fresh.py [c4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
w = bench.WORKLOADS[name]
scene = synth.make_ba_scene(2000, w["n_kf"], w["n_map"], w["n_obj"], stereo_frac=0.2)
for _ in range(3):
    b = BaProblem(scene); b.local_joint_ba(); b.close()
for idle in (0, 20, 0):
    for _ in range(idle):
        b = BaProblem(scene); b.close()
    out = []
    for _ in range(4):
        t0 = time.perf_counter(); b = BaProblem(scene); t1 = time.perf_counter(); b.local_joint_ba(); t2 = time.perf_counter(); b.close(); t3 = time.perf_counter()
        out.append("create %.2f  BA %.2f  destroy %.2f" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    print("after %2d idle create/destroy pairs:" % idle, " | ".join(out))
