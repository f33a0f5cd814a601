"""A global-BA-sized probe (Optimizer::GlobalBundleAdjustemnt runs through the same kernels with n_obj = 0): n_kf key-frames,
n_pt landmarks with `obs` random observers each.  python tools/ba_global_probe.py [n_kf n_pt obs iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_pt = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
obs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
sc = synth.make_ba_scene_large(11, n_kf, n_pt, obs_per_pt=obs)
ba = BaProblem(sc)
for det in (True, False):
    ba.set_deterministic(det)
    for r in range(2):
        ba.set_state(sc["kf_pose"], sc["pt_xyz"], sc["obj_pose"])
        t = time.time(); tr = ba.optimize(iters, 0, 0, 0); dt = time.time() - t
        print("%d KF / %d landmarks / %d edges, deterministic=%s: %.2f ms for %d LM iterations (%d trials), chi2 %.6g -> %.6g" % (
            n_kf, n_pt, len(sc["mono_pt"]), det, 1e3 * dt, len(tr["chi2"]), int(sum(tr["trials"])), tr["chi2"][0], tr["chi2"][-1]))
ba.close()
