"""CPU: the ellipsoid-plane restatement (oracle/ellipsoid_oracle.py, reference src/pca/EllipsoidExtractorEdges.cpp:35-175 and
src/pca/EllipsoidExtractorLocalOptimization.cpp:16-85) against closed-form geometry and a recovery problem."""
import numpy as np

from oracle import ellipsoid_oracle as EO


def _rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def test_sphere_and_axis_aligned_distances():
    R = np.eye(3)
    t = np.array([0.3, -0.2, 1.0])
    # sphere of radius 0.5: distance = | |signed centre distance| - r |
    for pl in ([0, 0, 1, -3.0], [1, 0, 0, 0.0], [0.6, 0.0, 0.8, -0.1], [0, 2, 0, 5.0]):
        pl = np.array(pl, float)
        c = abs(pl[:3] @ t + pl[3]) / np.linalg.norm(pl[:3])
        assert abs(EO.plane_error(t, R, np.array([0.5, 0.5, 0.5]), pl) - abs(c - 0.5)) < 1e-12
    # axis-aligned ellipsoid, plane z = 3: nearest tangent point is the top, distance 3 - (tz + c)
    assert abs(EO.plane_error(t, R, np.array([0.4, 0.7, 0.9]), np.array([0, 0, 1, -3.0])) - (3 - 1.9)) < 1e-12
    # a plane through the ellipsoid: distance to the nearer tangent plane, still non-negative
    assert abs(EO.plane_error(t, R, np.array([0.4, 0.7, 0.9]), np.array([0, 0, 1, -1.5])) - 0.4) < 1e-12


def test_tangent_planes_have_zero_error_and_direction_rule():
    rng = np.random.default_rng(1)
    for _ in range(20):
        ell = np.concatenate([rng.normal(size=3), _rand_quat(rng), rng.uniform(0.2, 1.5, size=3)])
        R = EO.quat_to_R(ell[3:7])
        planes = EO.tangent_planes(ell, rng.normal(size=(8, 3)))
        for pl in planes:
            assert EO.plane_error(ell[:3], R, ell[7:], pl) < 1e-9
            # outward normal: the centre is on the negative side -> the direction rule returns the FARTHEST tangent distance
            far = EO.plane_error(ell[:3], R, ell[7:], pl, direction=True)
            h = np.sqrt(np.sum((ell[7:] * (R.T @ pl[:3])) ** 2))
            assert abs(far - 2 * h) < 1e-9
            assert EO.plane_error(ell[:3], R, ell[7:], -pl, direction=True) < 1e-9   # inward normal: nearest


def test_numeric_jacobian_matches_analytic_support_function():
    rng = np.random.default_rng(2)
    ell = np.concatenate([rng.normal(size=3), _rand_quat(rng), rng.uniform(0.3, 1.0, size=3)])
    R = EO.quat_to_R(ell[3:7])
    est = np.concatenate([ell[:3], ell[7:]])
    n = rng.normal(size=3)
    n /= np.linalg.norm(n)
    pl = np.array([n[0], n[1], n[2], -(n @ ell[:3]) - 3.0])          # plane 3 away from the centre, outside
    J = EO.numeric_jacobian(est, R, pl)
    m = R.T @ n
    h = np.sqrt(np.sum((ell[7:] * m) ** 2))
    # e = |n.t + d| - h with n.t + d = -3  ->  de/dt = -n, de/ds_i = -s_i m_i^2 / h
    assert np.abs(J[:3] + n).max() < 1e-6 and np.abs(J[3:] + ell[7:] * m * m / h).max() < 1e-6


def test_fit_recovers_ellipsoid_from_tangent_planes():
    rng = np.random.default_rng(3)
    gt = np.concatenate([[0.2, -0.1, 2.0], _rand_quat(rng), [0.5, 0.3, 0.8]])
    planes = EO.tangent_planes(gt, rng.normal(size=(14, 3)))
    start = gt.copy()
    start[:3] += [0.05, -0.04, 0.06]
    start[7:] *= [1.15, 0.9, 1.1]
    r = EO.fit(start, planes)
    assert r["iters"] >= 3 and r["chi2"] < 1e-10
    assert np.abs(r["ell"] - gt).max() < 1e-5 and np.array_equal(r["ell"][3:7], gt[3:7])
    assert np.all(np.diff(r["trace"][:, 0]) <= 1e-15)                 # chi2 never increases over accepted iterations
    # no planes: unchanged, zero iterations
    r0 = EO.fit(start, np.zeros((0, 4)))
    assert r0["iters"] == 0 and np.array_equal(r0["ell"], start)


# ---- priorInfer::infer's problem (src/core/PriorInfer.cpp:331-427): 7 unknowns incl. yaw, plane-with-normal / plane / prior edges
def _prior_scene(rng, yaw_err=0.1):
    yaw = rng.uniform(-1.0, 1.0)
    q = np.array([0, 0, np.sin(yaw / 2), np.cos(yaw / 2)])
    gt = np.concatenate([rng.normal(size=3) * 0.3 + [0, 0, 1.0], q, np.sort(rng.uniform(0.3, 1.0, size=3))])
    planes = -EO.tangent_planes(gt, rng.normal(size=(10, 3)))                       # inward normals (the direction rule)
    axes = [[0, 0, -1], [np.cos(yaw), np.sin(yaw), 0], [-np.sin(yaw), np.cos(yaw), 0]]
    planes_n = -EO.tangent_planes(gt, axes)                                         # supporting plane + two axis-aligned faces
    init = gt.copy()
    init[:3] += rng.normal(scale=0.04, size=3)
    init[7:] += rng.normal(scale=0.05, size=3)
    y0 = yaw + yaw_err
    init[3:7] = [0, 0, np.sin(y0 / 2), np.cos(y0 / 2)]
    return gt, init, planes_n, planes


def test_prior_problem_pieces_in_closed_form():
    q = np.array([0, 0, np.sin(0.15), np.cos(0.15)])                                # yaw 0.3
    assert abs(EO.min_angle([np.cos(0.3), np.sin(0.3), 0.0], q)) < 1e-12            # along the x axis of the ellipsoid
    assert abs(EO.min_angle([-np.sin(0.3), np.cos(0.3), 0.0], q)) < 1e-12           # along its y axis
    assert abs(EO.min_angle([np.cos(0.5), np.sin(0.5), 0.0], q) - 0.2) < 1e-12      # 0.2 rad off the x axis
    assert EO.min_angle([0.1, 0.0, 1.0], q) == 0.0                                  # within 30 deg of z: no angle constraint
    assert np.allclose(EO.pri_of([0.6, -0.3, 0.9]), [2.0, 3.0])
    t, q2, s = EO.yaw_update([1.0, 2.0, 3.0], q, [0.5, 0.6, 0.7], [0.1, 0, 0, 0.01, 0.02, 0.03, 0.2])
    assert np.allclose(t, [1.0 + 0.1 * np.cos(0.3), 2.0 + 0.1 * np.sin(0.3), 3.0])  # translation in the ellipsoid's own frame
    assert np.allclose(q2, [0, 0, np.sin(0.25), np.cos(0.25)]) and np.allclose(s, [0.51, 0.62, 0.73])


def test_prior_fit_recovers_a_known_ellipsoid_with_its_yaw():
    rng = np.random.default_rng(11)
    for _ in range(3):
        gt, init, planes_n, planes = _prior_scene(rng)
        r = EO.prior_fit(init, planes_n, planes, EO.pri_of(gt[7:]), weight=1.0, angle_sigma_deg=10.0)
        assert r["chi2"] < 1e-12 and np.abs(r["ell"] - gt).max() < 1e-6
        assert np.all(np.diff(r["trace"][:, 0]) <= 1e-15)                           # chi2 never goes up


def test_prior_edge_pulls_the_axes_towards_the_prior_ratio():
    rng = np.random.default_rng(12)
    gt, init, planes_n, planes = _prior_scene(rng)
    free = EO.prior_fit(init, planes_n[:1], planes[:2], EO.pri_of(gt[7:]) * 1.5, weight=0.0)     # under-constrained, prior off
    held = EO.prior_fit(init, planes_n[:1], planes[:2], EO.pri_of(gt[7:]) * 1.5, weight=10.0)
    target = EO.pri_of(gt[7:]) * 1.5
    assert np.abs(EO.pri_of(held["ell"][7:]) - target).max() < 0.2 * np.abs(EO.pri_of(free["ell"][7:]) - target).max()
