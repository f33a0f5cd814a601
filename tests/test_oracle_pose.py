"""CPU tests pinning the pose-only restatement (oracle/ba_oracle.c: ba_oracle_pose_optimization, reference
Optimizer::PoseOptimization, src/Optimizer.cc:244-456) against the independent dense formulation in
oracle/ba_dense_check.py (numeric Jacobians, 4x4 matrices, scipy solve), plus behavioural checks."""
import numpy as np
import pytest

from oracle import ba_dense_check as dc
from oracle import ba_oracle as bo
from qsp_slam_amd import synth


@pytest.mark.parametrize("seed,n,stereo_frac", [(1, 120, 0.3), (2, 80, 0.0), (3, 60, 1.0), (4, 9, 0.5)])
def test_pose_only_lm_matches_the_independent_formulation(seed, n, stereo_frac):
    pp = synth.make_pose_problem(seed, n=n, stereo_frac=stereo_frac)
    r = bo.pose_optimization(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    d = dc.pose_optimization_dense(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    assert list(r["iters"]) == list(d["iters"])
    assert np.array_equal(r["outlier"], d["outlier"]) and r["n_inliers"] == d["n_inliers"]
    for rnd in range(4):
        k = r["iters"][rnd]
        # LM trials per iteration -- compared while the iteration still makes progress: once chi2 has converged to
        # rounding level the sign of rho (and with it the number of rejected trials) is decided by the last bit
        chi = r["trace"][rnd, :k, 0]
        live = np.ones(k, bool)
        live[1:] = (chi[:-1] - chi[1:]) > 1e-9 * chi[1:]
        first_dead = int(np.argmin(live)) if not live.all() else k
        assert np.array_equal(r["trace"][rnd, :first_dead, 2], d["trace"][rnd, :first_dead, 2])
        assert np.allclose(r["trace"][rnd, :k, 0], d["trace"][rnd, :k, 0], rtol=1e-6)          # chi2
        assert np.allclose(r["trace"][rnd, :first_dead, 1], d["trace"][rnd, :first_dead, 1], rtol=1e-4)   # lambda
    assert np.abs(dc.T_from_pose7(r["pose"]) - d["T"]).max() < 1e-7
    if n < 10:
        assert r["iters"][1] == 0                                                              # edges().size() < 10: one round


def test_pose_only_recovers_the_pose_and_the_gross_outliers():
    pp = synth.make_pose_problem(11, n=600, stereo_frac=0.4, outlier_frac=0.15)
    r = bo.pose_optimization(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    assert (r["outlier"].astype(bool) & pp["gross"]).sum() == pp["gross"].sum()                # every gross outlier is found
    assert r["outlier"].sum() <= pp["gross"].sum() + 0.08 * len(pp["gross"])                   # + the chi2 tail of the inliers
    assert np.abs(r["pose"] - pp["gt_pose"]).max() < 5e-3 < np.abs(pp["pose"] - pp["gt_pose"]).max()
    assert r["n_inliers"] == len(pp["gross"]) - r["outlier"].sum()


def test_pose_only_fewer_than_three_correspondences_returns_zero():
    pp = synth.make_pose_problem(5, n=2)
    r = bo.pose_optimization(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    assert r["n_inliers"] == 0 and np.array_equal(r["pose"], pp["pose"]) and not r["iters"].any()
