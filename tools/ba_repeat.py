"""Repeats the two-stage BA of a small scene and reports how many distinct (trials, accepted, levels) outcomes occur."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_ba import SCENES
for name in ("tiny", "mono", "two_fixed"):
    sc = synth.make_ba_scene(**dict(SCENES[name], outlier_frac=0.08))
    out = collections.Counter()
    chis = []
    for rep in range(30):
        p = BaProblem(sc)
        if os.environ.get("QSP_BA_DET") == "0":
            p.set_deterministic(False)
        t1, t2 = p.local_joint_ba()
        key = (tuple(t1["trials"]), tuple(t1["accepted"]), tuple(t2["trials"]), tuple(t2["accepted"]))
        out[key] += 1
        chis.append(t2["chi2"][-1]); last = t2["chi2"]
        p.close()
    print(os.environ.get("QSP_HIP_LIB", "default")[-16:], name, "distinct outcomes:", len(out), "chi2 spread %.3e" % (max(chis) - min(chis)), list(out.values()), "rel %.2e" % ((max(chis) - min(chis)) / max(chis)))
