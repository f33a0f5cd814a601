"""GPU parity tests of the single-launch pose-only optimisation (qsp_pose_optimize) through the C-ABI
against the C restatement of Optimizer::PoseOptimization (oracle/ba_oracle.c, itself pinned by the dense formulation in
tests/test_oracle_pose.py).  Bars: outlier flags and inlier count bit-exact; iteration counts and LM trials bit-exact while an
iteration still makes progress (the stop rules fire on the last bit afterwards); chi2 1e-9, lambda 1e-6, pose 1e-9 (FP64 both sides, different summation order only)."""
import time

import numpy as np
import pytest

from oracle import ba_oracle as bo
from qsp_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def po():
    from qsp_slam_amd.ba import PoseOptimizer
    p = PoseOptimizer(max_points=4096)
    yield p
    p.close()


def compare(r, g):
    """iteration and trial counts are compared while an iteration still makes progress: once chi2 has converged to
    rounding level, g2o's stop rules (rho == 0, ten rejected trials, three stalled iterations) fire on the last bit"""
    assert np.array_equal(g["outlier"], r["outlier"]) and g["n_inliers"] == r["n_inliers"]
    for rnd in range(4):
        kr, kg = int(r["iters"][rnd]), int(g["iters"][rnd])
        if kr == 0:                                  # round not run (fewer than 10 edges: one round only)
            assert kg == 0
            continue
        chi = r["trace"][rnd, :kr, 0]
        live = np.ones(kr, bool)
        live[1:] = (chi[:-1] - chi[1:]) > 1e-9 * chi[1:]
        fd = int(np.argmin(live)) if not live.all() else kr
        assert kg >= fd and (kg == kr or min(kg, kr) >= fd)
        assert np.array_equal(g["trace"][rnd, :fd, 2], r["trace"][rnd, :fd, 2])
        k = min(kr, kg)
        assert np.allclose(g["trace"][rnd, :k, 0], r["trace"][rnd, :k, 0], rtol=1e-9)
        assert np.allclose(g["trace"][rnd, :fd, 1], r["trace"][rnd, :fd, 1], rtol=1e-6)
        assert abs(g["trace"][rnd, kg - 1, 0] - r["trace"][rnd, kr - 1, 0]) <= 1e-9 * r["trace"][rnd, kr - 1, 0]
    assert np.abs(g["pose"] - r["pose"]).max() < 1e-9


@pytest.mark.parametrize("seed,n,stereo_frac,outl", [(1, 400, 0.3, 0.1), (2, 1500, 0.0, 0.2), (3, 257, 1.0, 0.05),
                                                     (4, 9, 0.5, 0.0), (5, 40, 0.5, 0.5), (6, 3000, 0.4, 0.1)])
def test_pose_optimisation_matches_oracle(po, seed, n, stereo_frac, outl):
    pp = synth.make_pose_problem(seed, n=n, stereo_frac=stereo_frac, outlier_frac=outl)
    r = bo.pose_optimization(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    g = po.optimize(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    compare(r, g)


def test_fewer_than_three_correspondences(po):
    pp = synth.make_pose_problem(7, n=2)
    g = po.optimize(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    assert g["n_inliers"] == 0 and np.array_equal(g["pose"], pp["pose"]) and not g["iters"].any()
    g0 = po.optimize(pp["K"], pp["pose"], np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), np.zeros(0, np.uint8))
    assert g0["n_inliers"] == 0 and np.array_equal(g0["pose"], pp["pose"])


def test_capacity_and_argument_errors(po):
    from qsp_slam_amd import _lib
    pp = synth.make_pose_problem(8, n=5000)
    with pytest.raises(_lib.QspError):
        po.optimize(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])      # created for 4096


def test_determinism_and_latency(po):
    pp = synth.make_pose_problem(9, n=1000, stereo_frac=0.3, outlier_frac=0.1)
    a = po.optimize(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    t = time.perf_counter()
    for _ in range(20):
        b = po.optimize(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    dt = (time.perf_counter() - t) / 20
    assert np.array_equal(a["pose"], b["pose"]) and np.array_equal(a["trace"][~np.isnan(a["trace"])], b["trace"][~np.isnan(b["trace"])])
    t = time.perf_counter()
    bo.pose_optimization(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    dt_cpu = time.perf_counter() - t
    print("pose optimisation, 1000 correspondences: GPU %.3f ms per call (incl. H2D/D2H), C oracle %.3f ms" % (1e3 * dt, 1e3 * dt_cpu))
    assert dt < 5e-3
