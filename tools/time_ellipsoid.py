"""Times qsp_ellipsoid_fit_planes (batched OptimizeEllipsoidUsingPlanes) end to end, and the numpy restatement on a few."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import ellipsoid_oracle as EO
from qsp_slam_amd.ellipsoid import optimize_ellipsoids_using_planes
rng = np.random.default_rng(0)
def scene(n, k):
    ells, planes = [], []
    for i in range(n):
        q = rng.normal(size=4)
        gt = np.concatenate([rng.normal(size=3) + [0, 0, 3], q / np.linalg.norm(q), rng.uniform(0.2, 1.2, size=3)])
        pl = EO.tangent_planes(gt, rng.normal(size=(k, 3)))
        pl[:, 3] += rng.normal(scale=0.01, size=k)
        s = gt.copy(); s[:3] += rng.normal(scale=0.05, size=3); s[7:] *= np.exp(rng.normal(scale=0.1, size=3))
        ells.append(s); planes.append(pl)
    return np.array(ells), planes
ells, planes = scene(2000, 12)
optimize_ellipsoids_using_planes(ells[:4], planes[:4])
for n in (1, 100, 2000):
    t = time.time(); reps = 5
    for _ in range(reps):
        out, chi2, iters = optimize_ellipsoids_using_planes(ells[:n], planes[:n])
    dt = (time.time() - t) / reps
    print("%5d ellipsoids x 12 planes: %.3f ms per call (%.2f us per ellipsoid), mean iterations %.1f" % (n, 1e3 * dt, 1e6 * dt / n, iters.mean()))
t = time.time()
for i in range(5):
    EO.fit(ells[i], planes[i])
print("numpy restatement: %.1f ms per ellipsoid" % (1e3 * (time.time() - t) / 5))
