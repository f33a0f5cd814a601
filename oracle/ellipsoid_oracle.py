"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy float64) of the single-ellipsoid plane fit, the checker for
qsp_ellipsoid_fit_planes.  Never imported by the product path.

What it restates: EllipsoidExtractor::OptimizeEllipsoidUsingPlanes, reference
src/pca/EllipsoidExtractorLocalOptimization.cpp:16-85 --
    plane_error()   EdgeEllipsoidPlane::computeError -> distanceFromPlaneToEllipsoid -> GetNearestAndFarthestPointOnEllipsoidToPlane,
                    src/pca/EllipsoidExtractorEdges.cpp:35-175, following the reference's chain LITERALLY with 4x4 matrices:
                    Tew = pose^-1, plane.transform(Tew) = (Tew^T)^-1 param (src/core/Plane.cpp:151-156), dual quadric
                    (src/core/Ellipsoid.cpp:395-405) -> inverse -> normalise -> a^2 b^2 c^2 -> the two tangent points.
                    (The kernel uses the closed forms of these matrices; the two must agree to rounding.)
                    direction=True: EdgeSE3EllipsoidPlane with an identity camera, :151-226.
    numeric_jacobian()  Thirdparty/g2o/g2o/core/base_unary_edge.hpp:82-123 (central differences, delta = 1e-9)
    fit()           VertexEllipsoidXYZABC::oplusImpl = exp_update_XYZABC (src/core/Ellipsoid.cpp:62-76: translation and
                    half-axes add, rotation kept), BlockSolverX + LinearSolverDense (LDLT), OptimizationAlgorithmLevenberg
                    (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-189), optimize(10).

PARITY UNPINNED: the reference holds no test or fixture for this function, it has no live caller in this revision (call
sites commented out, src/pca/EllipsoidExtractorMultiPlanes.cpp:692-693), and g2o / Eigen cannot be built in the image.  Pins
used instead: closed-form geometry (sphere and axis-aligned cases, tangent planes give zero error) and recovery of a known
ellipsoid from its tangent planes (tests/test_oracle_ellipsoid.py).  The difference quotient with delta = 1e-9 turns 1e-16
rounding differences into ~1e-7 in the Jacobian, so LM iterates of two correct implementations agree to ~1e-6, not to the bit.
"""
import numpy as np


def quat_to_R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _hom(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def _dist(param, p, keep=False):
    v = (param[0] * p[0] + param[1] * p[1] + param[2] * p[2] + param[3]) / np.sqrt(param[0] ** 2 + param[1] ** 2 + param[2] ** 2)
    return v if keep else abs(v)


def plane_error(t, R, s, plane, direction=False):
    plane = np.asarray(plane, np.float64)
    Twe = _hom(R, t)
    Tew = np.linalg.inv(Twe)
    pose_e = Tew @ Twe                                        # e.transform_from(Tew)
    param_e = np.linalg.inv(Tew.T) @ plane                    # plane::transform
    Q_star = pose_e @ np.diag([s[0] ** 2, s[1] ** 2, s[2] ** 2, -1.0]) @ pose_e.T
    Q = np.linalg.inv(Q_star)
    Q = Q / (-Q[3, 3])
    a2, b2, c2 = 1 / Q[0, 0], 1 / Q[1, 1], 1 / Q[2, 2]
    A, B, C, D = param_e
    alpha = np.sqrt(4 / (A * A * a2 + B * B * b2 + C * C * c2))
    ext = alpha * np.array([A * a2 / 2, B * b2 / 2, C * c2 / 2])
    d1, d2 = _dist(param_e, ext), _dist(param_e, -ext)
    nearest, farthest = (d2, d1) if abs(d1) > abs(d2) else (d1, d2)
    if not direction:
        return nearest
    dis = nearest if _dist(plane, t, keep=True) > 0 else farthest
    return 0.0 if np.isnan(dis) else dis


def numeric_jacobian(est, R, plane, direction=False):
    delta = 1e-9
    scalar = 1.0 / (2 * delta)
    J = np.zeros(6)
    for d in range(6):
        q = est.copy()
        q[d] = est[d] + delta
        e1 = plane_error(q[:3], R, q[3:], plane, direction)
        q[d] = est[d] + -delta
        e2 = plane_error(q[:3], R, q[3:], plane, direction)
        J[d] = scalar * (e1 - e2)
    return J


def fit(ell, planes, n_iter=10, direction=False):
    """ell (10,) t q(xyzw) half-axes; planes (P,4).  -> dict(ell (10,), chi2, iters, trace (iters,3))"""
    ell = np.asarray(ell, np.float64)
    planes = np.asarray(planes, np.float64).reshape(-1, 4)
    R = quat_to_R(ell[3:7])
    est = np.concatenate([ell[:3], ell[7:10]])
    trace = []
    if len(planes) == 0:
        return dict(ell=ell.copy(), chi2=0.0, iters=0, trace=np.zeros((0, 3)))

    def chi2_at(v):
        return sum(plane_error(v[:3], R, v[3:], p, direction) ** 2 for p in planes)

    lam, ni, nbad, cur = 0.0, 2.0, 0, 0.0
    for it in range(n_iter):
        H, b, cur = np.zeros((6, 6)), np.zeros(6), 0.0
        for p in planes:
            e = plane_error(est[:3], R, est[3:], p, direction)
            J = numeric_jacobian(est, R, p, direction)
            H += np.outer(J, J)
            b += J * -e
            cur += e * e
        ini = cur
        if it == 0:
            lam, ni, nbad = 1e-5 * np.max(np.abs(np.diag(H))), 2.0, 0
        qmax, rho = 0, 0.0
        while True:
            bk = est.copy()
            x = np.zeros(6)
            ok = True
            A = H + lam * np.eye(6)
            if np.any(A != 0):
                try:
                    np.linalg.cholesky(A)
                    x = np.linalg.solve(A, b)
                except np.linalg.LinAlgError:
                    ok = False
            if ok:
                est = est + x
            temp = chi2_at(est) if ok else np.finfo(np.float64).max
            rho = (cur - temp) / (1e-3 + float(np.sum(x * (lam * x + b))))
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                cur = temp
            else:
                lam *= ni
                ni *= 2
                est = bk
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        trace.append((cur, lam, qmax))
        if qmax == 10 or rho == 0:
            break
        nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
        if nbad >= 3:
            break
    out = ell.copy()
    out[:3], out[7:10] = est[:3], est[3:]
    return dict(ell=out, chi2=cur, iters=len(trace), trace=np.array(trace).reshape(-1, 3))


def tangent_planes(ell, normals):
    """planes (A B C D, unit normal) tangent to the ellipsoid, normal pointing away from the centre: test scenes"""
    ell = np.asarray(ell, np.float64)
    R, t, s = quat_to_R(ell[3:7]), ell[:3], ell[7:10]
    out = []
    for n in np.asarray(normals, np.float64).reshape(-1, 3):
        n = n / np.linalg.norm(n)
        h = np.sqrt(np.sum((s * (R.T @ n)) ** 2))             # support function
        out.append([n[0], n[1], n[2], -(n @ t + h)])
    return np.array(out)


# ---------------------------------------------------------------------------------------------------------------------------
# priorInfer::infer's single-ellipsoid problem (reference src/core/PriorInfer.cpp:331-427) -- SURVEY section 8f row 4, the other
# live-code-shaped quadric piece: one VertexEllipsoidXYZABCYaw (7 unknowns: translation in the ellipsoid's own frame, half-axes,
# yaw; oplus = exp_update_XYZABCYaw, src/core/Ellipsoid.cpp:78-106: pose * SE3(fromXYZPRY(t, 0, 0, yaw)), scale + ds), a FIXED
# identity VertexSE3Expmap, and
#   EdgeSE3EllipsoidPlaneWithNormal (2-D: nearest tangent distance, calculateMinAngle of the plane normal against the ellipsoid's
#       axes; information diag(1, 1 / sigma_angle^2) * w^2; RobustKernelHuber, delta 1)   src/pca/EllipsoidExtractorEdges.cpp:297-375
#   EdgeSE3EllipsoidPlane with setNormalDirection(true) (1-D; information w^2; Huber)      :197-226
#   EdgePri (2-D: Pri(ellipsoid) - prior, Pri = (mid / min, max / min) of |half-axes|; information weight^2, no kernel)
#       src/core/PriorInfer.cpp:60-78,437-450
# with g2o's numeric Jacobians (delta 1e-9), BlockSolverX + dense solver, Levenberg-Marquardt, optimize(10).
# Like OptimizeEllipsoidUsingPlanes above the function has no live caller in this revision (`priorInfer` is never instantiated):
# PARITY UNPINNED, pins = closed-form geometry and recovery of a known ellipsoid (tests/test_oracle_ellipsoid.py).
# ---------------------------------------------------------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def yaw_update(t, q, s, u):
    """exp_update_XYZABCYaw: pose * SE3Quat(zyx_euler_to_quat(0, 0, yaw), trans) (se3quat.h:110-116,208-225), scale + ds"""
    u = np.asarray(u, np.float64)
    dq = np.array([0.0, 0.0, np.sin(u[6] * 0.5), np.cos(u[6] * 0.5)])
    R = quat_to_R(q)
    t2 = np.asarray(t, np.float64) + R @ u[:3]
    q2 = quat_mul(q, dq)
    if q2[3] < 0:
        q2 = -q2
    q2 = q2 / np.linalg.norm(q2)
    return t2, q2, np.asarray(s, np.float64) + u[3:6]


def min_angle(normal, q):
    """EdgeSE3EllipsoidPlaneWithNormal::calculateMinAngle, EllipsoidExtractorEdges.cpp:297-358"""
    R = quat_to_R(q)
    Nc = np.linalg.inv(R) @ np.asarray(normal, np.float64)
    cz = Nc[2] / np.linalg.norm(Nc)
    az = np.arccos(cz)
    if min(abs(az), abs(az - np.pi)) < np.pi / 180.0 * 30:
        return 0.0
    nxy = np.array([Nc[0], Nc[1], 0.0])
    ang = np.arccos(nxy[0] / np.linalg.norm(nxy))
    return min(min(ang, abs(ang - np.pi / 2)), abs(ang - np.pi))


def pri_of(s):
    a = np.sort(np.abs(np.asarray(s, np.float64)))
    return np.array([a[1] / a[0], a[2] / a[0]])


def _huber(e2, delta=1.0):
    if e2 <= delta * delta:
        return e2, 1.0
    r = np.sqrt(e2)
    return 2 * r * delta - delta * delta, delta / r


def prior_fit(ell, planes_normal, planes, pri, weight, angle_sigma_deg=10.0, ground_plane_weight=None, n_iter=10):
    """-> dict(ell (10,), chi2, iters, trace (iters,3)).  ground_plane_weight: weight of the FIRST plane of each list
    (bUseGroundPlaneWeight), None = 1."""
    ell = np.asarray(ell, np.float64)
    PN = np.asarray(planes_normal, np.float64).reshape(-1, 4)
    PL = np.asarray(planes, np.float64).reshape(-1, 4)
    pri = np.asarray(pri, np.float64)
    sig = angle_sigma_deg / 180.0 * np.pi
    state = (ell[:3].copy(), ell[3:7].copy(), ell[7:10].copy())

    def w_of(i):
        return ground_plane_weight if (ground_plane_weight is not None and i == 0) else 1.0

    edges = []                                  # (kind, data, omega diagonal, robust)
    for i, p in enumerate(PN):
        w = w_of(i)
        edges.append((0, p, np.array([w * w, (w / sig) ** 2]), True))
    for i, p in enumerate(PL):
        w = w_of(i)
        edges.append((1, p, np.array([w * w]), True))
    edges.append((2, pri, np.array([weight * weight, weight * weight]), False))

    def err(kind, data, st):
        t, q, s = st
        if kind == 0:
            return np.array([plane_error(t, quat_to_R(q), s, data, False), min_angle(data[:3], q)])
        if kind == 1:
            return np.array([plane_error(t, quat_to_R(q), s, data, True)])
        return pri_of(s) - data

    def chi2_at(st):
        tot = 0.0
        for kind, data, om, rob in edges:
            e = err(kind, data, st)
            c = float(np.sum(om * e * e))
            tot += _huber(c)[0] if rob else c
        return tot

    trace = []
    lam, ni, nbad, cur = 0.0, 2.0, 0, 0.0
    delta = 1e-9
    for it in range(n_iter):
        H, b, cur = np.zeros((7, 7)), np.zeros(7), 0.0
        for kind, data, om, rob in edges:
            e = err(kind, data, state)
            J = np.zeros((e.shape[0], 7))
            for d in range(7):
                u = np.zeros(7)
                u[d] = delta
                e1 = err(kind, data, yaw_update(*state, u))
                u[d] = -delta
                e2 = err(kind, data, yaw_update(*state, u))
                J[:, d] = (1.0 / (2 * delta)) * (e1 - e2)
            c = float(np.sum(om * e * e))
            r0, r1 = _huber(c) if rob else (c, 1.0)
            W = np.diag(r1 * om)
            H += J.T @ W @ J
            b -= J.T @ (W @ e)
            cur += r0
        ini = cur
        if it == 0:
            lam, ni, nbad = 1e-5 * np.max(np.abs(np.diag(H))), 2.0, 0
        qmax, rho = 0, 0.0
        while True:
            bk = state
            x = np.zeros(7)
            ok = True
            A = H + lam * np.eye(7)
            if np.any(A != 0):
                try:
                    np.linalg.cholesky(A)
                    x = np.linalg.solve(A, b)
                except np.linalg.LinAlgError:
                    ok = False
            if ok:
                state = yaw_update(*state, x)
            temp = chi2_at(state) if ok else np.finfo(np.float64).max
            rho = (cur - temp) / (1e-3 + float(np.sum(x * (lam * x + b))))
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                cur = temp
            else:
                lam *= ni
                ni *= 2
                state = bk
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        trace.append((cur, lam, qmax))
        if qmax == 10 or rho == 0:
            break
        nbad = nbad + 1 if (ini - cur) * 1e3 < ini else 0
        if nbad >= 3:
            break
    out = np.concatenate([state[0], state[1], state[2]])
    return dict(ell=out, chi2=cur, iters=len(trace), trace=np.array(trace).reshape(-1, 3))
