"""Randomised parity sweep of path B on the GPU box: Optimizer::LocalJointBundleAdjustment's two-stage schedule on random small
scenes (sizes, stereo share, outlier share, number of fixed key frames, objects or none) against the C oracle: hessianIndex tables
exact, LM trial / accept sequence equal while chi2 still moves, chi2 trace and final estimates to the stated tolerances.
python tools/ba_parity_sweep.py [n_scenes]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo
from qsp_slam_amd import synth
from qsp_slam_amd.ba import BaProblem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(7)
worst = dict(chi2=0.0, pose=0.0)
t0 = time.time()
for c in range(n):
    n_kf = int(rng.integers(4, 16)); n_pt = int(rng.integers(60, 600)); n_obj = int(rng.integers(0, 5))
    kw = dict(seed=int(rng.integers(1, 10 ** 6)), n_kf=n_kf, n_pt=n_pt, n_obj=n_obj, stereo_frac=float(rng.choice([0.0, 0.2, 0.6])),
              outlier_frac=float(rng.choice([0.0, 0.05, 0.1])), n_fixed=int(rng.integers(1, 3)))
    sc = synth.make_ba_scene(**kw)
    ref = bo.BaProblem(sc); r1, r2 = ref.local_joint_ba()
    gpu = BaProblem(sc); g1, g2 = gpu.local_joint_ba()
    for g, r in ((g1, r1), (g2, r2)):
        m = min(len(g["chi2"]), len(r["chi2"]))
        prev = None
        for i in range(m):
            rel = abs(g["chi2"][i] - r["chi2"][i]) / max(r["chi2"][i], 1e-30)
            worst["chi2"] = max(worst["chi2"], rel)
            assert rel < 1e-6, (c, kw, i, rel)
            moving = prev is None or (prev - r["chi2"][i]) > 1e-6 * prev
            if moving:
                assert g["trials"][i] == r["trials"][i] and g["accepted"][i] == r["accepted"][i], (c, kw, i)
            prev = r["chi2"][i]
    kf, pt, ob = gpu.state(); rkf, rpt, rob = ref.state()
    e = max(np.abs(kf - rkf).max() / max(np.abs(rkf).max(), 1e-30), np.abs(pt - rpt).max() / max(np.abs(rpt).max(), 1e-30))
    if len(rob):
        e = max(e, np.abs(ob - rob).max() / np.abs(rob).max())
    worst["pose"] = max(worst["pose"], e)
    assert e < 1e-4, (c, kw, e)                              # north_star's bar; typical values are printed below
    kh, oh, ph = gpu.index()
    assert np.array_equal(kh, r2["kf_hidx"]) and np.array_equal(oh, r2["obj_hidx"]) and np.array_equal(ph, r2["pt_hidx"]), (c, kw)
    gpu.close()
print("%d scenes in %.0f s: index tables exact, LM paths equal while chi2 moves; worst relative difference chi2 %.1e, estimates %.1e" % (
    n, time.time() - t0, worst["chi2"], worst["pose"]))
