"""TEST INFRASTRUCTURE ONLY -- an INDEPENDENT formulation of the joint bundle adjustment, used to pin oracle/ba_oracle.c.

Nothing here shares code or structure with ba_oracle.c:
  * poses are 4x4 matrices; the exponential is a Taylor series of the 4x4 twist matrix, the logarithm is scipy's logm
    (ba_oracle.c: quaternions + the closed forms of g2o's SE3Quat);
  * Jacobians are central differences of the residual functions (ba_oracle.c: g2o's analytic Jacobians);
  * the normal equations are the FULL dense system over poses, objects and points, solved with scipy's Cholesky
    (ba_oracle.c: per-landmark Schur complement + reduced solve + back-substitution);
  * what IS shared, because it is the specification: residual definitions (reference
    Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:79-140 incl. the float 1/z of the stereo projection,
    include/ObjectPoseGraph.h:69-73), Huber re-weighting (g2o/core/robust_kernel_impl.cpp:78-91, base_edge.h:96-102)
    and the Levenberg-Marquardt schedule (g2o/core/optimization_algorithm_levenberg.cpp:61-189).
"""
import numpy as np
import scipy.linalg as sla


def twist(u):
    w, v = u[:3], u[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    return M


def exp_series(u, terms=40):
    M = twist(u)
    T = np.eye(4)
    P = np.eye(4)
    for k in range(1, terms):
        P = P @ M / k
        T = T + P
    return T


def log_vec(T):
    L = np.real(sla.logm(T))
    return np.array([L[2, 1], L[0, 2], L[1, 0], L[0, 3], L[1, 3], L[2, 3]])


def T_from_pose7(p):
    x, y, z, w = p[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = p[:3]
    return T


def res_mono(T, X, K, obs):
    p = T[:3, :3] @ X + T[:3, 3]
    return np.array([obs[0] - (K[0] * p[0] / p[2] + K[2]), obs[1] - (K[1] * p[1] / p[2] + K[3])])


def res_stereo(T, X, K, obs, exact=False):
    """exact=True drops the float32 rounding of 1/z (used only when differentiating numerically: the quantised function
    has no meaningful finite difference)"""
    p = T[:3, :3] @ X + T[:3, 3]
    if exact:
        u = p[0] / p[2] * K[0] + K[2]
        v = p[1] / p[2] * K[1] + K[3]
        return np.array([obs[0] - u, obs[1] - v, obs[2] - (u - K[4] / p[2])])
    invz = np.float32(1.0) / np.float32(p[2])
    u = p[0] * float(invz) * K[0] + K[2]
    v = p[1] * float(invz) * K[1] + K[3]
    ur = u - float(np.float32(K[4])) * float(invz)
    return np.array([obs[0] - u, obs[1] - v, obs[2] - ur])


def res_obj(Tcw, Tow, Z):
    return log_vec(np.linalg.inv(Z) @ Tcw @ np.linalg.inv(Tow))


def obj_edge_jacobians(e, Z):
    """The reference's own linearisation of the object edge (include/ObjectPoseGraph.h:75-88), which is a first-order
    approximation of the true derivative and therefore part of the specification:
    J = 1/2 [[w^, 0], [t^, w^]] + I ; d e / d xi_cam = J Adj(Z^-1) ; d e / d xi_obj = -J, Adj(T) = [[R, 0], [t^ R, R]]"""
    def hat(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])
    J = np.zeros((6, 6))
    J[:3, :3] = hat(e[:3]); J[3:, :3] = hat(e[3:]); J[3:, 3:] = hat(e[:3])
    J = 0.5 * J + np.eye(6)
    Zi = np.linalg.inv(Z)
    A = np.zeros((6, 6))
    A[:3, :3] = Zi[:3, :3]; A[3:, 3:] = Zi[:3, :3]; A[3:, :3] = hat(Zi[:3, 3]) @ Zi[:3, :3]
    return J @ A, -J


def huber(chi2, delta):
    if delta <= 0 or chi2 <= delta * delta:
        return chi2, 1.0
    s = np.sqrt(chi2)
    return 2 * s * delta - delta * delta, delta / s


class DenseBA(object):
    def __init__(self, scene, levels=None):
        s = scene
        self.kf = [T_from_pose7(p) for p in s["kf_pose"]]
        self.ob = [T_from_pose7(p) for p in s["obj_pose"]]
        self.pt = [np.array(x, float) for x in s["pt_xyz"]]
        self.s = s
        nm, ns, no = len(s["mono_pt"]), len(s["st_pt"]), len(s["oe_kf"])
        self.Z = [T_from_pose7(z) for z in s["oe_meas"]]
        self.edges = [("m", k) for k in range(nm)] + [("s", k) for k in range(ns)] + [("o", k) for k in range(no)]
        self.level = {e: 0 for e in self.edges}
        if levels is not None:
            for e, l in zip(self.edges, levels):
                self.level[e] = int(l)
        self.chi2_e = {e: 0.0 for e in self.edges}

    # ---- unknown ordering: free key-frames and objects by vertex id, then points by vertex id ---------------------
    def index(self):
        s = self.s
        act_kf, act_ob, act_pt = set(), set(), set()
        for e in self.edges:
            if self.level[e]:
                continue
            t, k = e
            if t == "m":
                act_kf.add(int(s["mono_kf"][k])); act_pt.add(int(s["mono_pt"][k]))
            elif t == "s":
                act_kf.add(int(s["st_kf"][k])); act_pt.add(int(s["st_pt"][k]))
            else:
                act_kf.add(int(s["oe_kf"][k])); act_ob.add(int(s["oe_obj"][k]))
        poses = [(int(s["kf_id"][i]), ("k", i)) for i in sorted(act_kf) if not s["kf_fixed"][i]]
        poses += [(int(s["obj_id"][i]), ("o", i)) for i in sorted(act_ob)]
        poses.sort()
        pts = sorted((int(s["pt_id"][i]), ("p", i)) for i in act_pt)
        order = [v for _, v in poses] + [v for _, v in pts]
        off, o = {}, 0
        for v in order:
            off[v] = o
            o += 3 if v[0] == "p" else 6
        return order, off, o

    def edge_vertices(self, e):
        s = self.s
        t, k = e
        if t == "m":
            return ("k", int(s["mono_kf"][k])), ("p", int(s["mono_pt"][k]))
        if t == "s":
            return ("k", int(s["st_kf"][k])), ("p", int(s["st_pt"][k]))
        return ("k", int(s["oe_kf"][k])), ("o", int(s["oe_obj"][k]))

    def get(self, v):
        return {"k": self.kf, "o": self.ob, "p": self.pt}[v[0]][v[1]]

    def residual(self, e, va=None, vb=None, exact=False):
        s = self.s
        t, k = e
        a, b = self.edge_vertices(e)
        A = self.get(a) if va is None else va
        B = self.get(b) if vb is None else vb
        if t == "m":
            return res_mono(A, B, s["kf_K"][a[1]], s["mono_obs"][k]), s["mono_info"][k]
        if t == "s":
            return res_stereo(A, B, s["kf_K"][a[1]], s["st_obs"][k], exact), s["st_info"][k]
        return res_obj(A, B, self.Z[k]), s["oe_info"]

    def delta_of(self, e, deltas):
        return deltas[{"m": 0, "s": 1, "o": 2}[e[0]]]

    def chi2(self, deltas):
        tot = 0.0
        for e in self.edges:
            if self.level[e]:
                continue
            r, info = self.residual(e)
            c = info * float(r @ r)
            self.chi2_e[e] = c
            tot += huber(c, self.delta_of(e, deltas))[0]
        return tot

    def perturbed(self, v, d):
        x = self.get(v)
        if v[0] == "p":
            return x + d
        return exp_series(d) @ x

    def build(self, deltas, h=1e-6):
        order, off, n = self.index()
        H = np.zeros((n, n))
        b = np.zeros(n)
        for e in self.edges:
            if self.level[e]:
                continue
            r, info = self.residual(e)
            c = info * float(r @ r)
            w = huber(c, self.delta_of(e, deltas))[1]
            va, vb = self.edge_vertices(e)
            Js = {}
            if e[0] == "o":
                Ji, Jj = obj_edge_jacobians(r, self.Z[e[1]])
                if va in off:
                    Js[va] = Ji
                if vb in off:
                    Js[vb] = Jj
            else:
                for which, v in ((0, va), (1, vb)):
                    if v not in off:
                        continue
                    dim = 3 if v[0] == "p" else 6
                    J = np.zeros((len(r), dim))
                    for i in range(dim):
                        d = np.zeros(dim)
                        d[i] = h
                        xp, xm = self.perturbed(v, d), self.perturbed(v, -d)
                        rp = self.residual(e, *((xp, None) if which == 0 else (None, xp)), exact=True)[0]
                        rm = self.residual(e, *((xm, None) if which == 0 else (None, xm)), exact=True)[0]
                        J[:, i] = (rp - rm) / (2 * h)
                    Js[v] = J
            for v1, J1 in Js.items():
                b[off[v1]: off[v1] + J1.shape[1]] += -w * info * (J1.T @ r)
                for v2, J2 in Js.items():
                    H[off[v1]: off[v1] + J1.shape[1], off[v2]: off[v2] + J2.shape[1]] += w * info * (J1.T @ J2)
        return order, off, H, b

    def snapshot(self):
        return [x.copy() for x in self.kf], [x.copy() for x in self.ob], [x.copy() for x in self.pt]

    def restore(self, snap):
        self.kf, self.ob, self.pt = [x.copy() for x in snap[0]], [x.copy() for x in snap[1]], [x.copy() for x in snap[2]]

    def optimize(self, n_iter, deltas):
        """SparseOptimizer::optimize(n_iter) with Levenberg-Marquardt; returns the per-iteration trace"""
        tr = dict(chi2=[], lam=[], trials=[], accepted=[])
        lam, ni, n_bad = 0.0, 2.0, 0
        for it in range(n_iter):
            cur = self.chi2(deltas)
            ini = cur
            order, off, H, b = self.build(deltas)
            if it == 0:
                lam = 1e-5 * np.abs(np.diag(H)).max()
                ni, n_bad = 2.0, 0
            q, rho, acc = 0, 0.0, 0
            while True:
                snap = self.snapshot()
                try:
                    x = sla.cho_solve(sla.cho_factor(H + lam * np.eye(len(b))), b)
                    ok = True
                except sla.LinAlgError:
                    x = np.zeros_like(b)
                    ok = False
                for v in order:
                    d = x[off[v]: off[v] + (3 if v[0] == "p" else 6)]
                    {"k": self.kf, "o": self.ob, "p": self.pt}[v[0]][v[1]] = self.perturbed(v, d)
                tmp = self.chi2(deltas) if ok else np.finfo(float).max
                scale = float(x @ (lam * x + b)) + 1e-3
                rho = (cur - tmp) / scale
                if rho > 0 and np.isfinite(tmp):
                    lam *= max(1.0 / 3.0, min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0))
                    ni = 2.0
                    cur = tmp
                    acc = 1
                else:
                    lam *= ni
                    ni *= 2
                    self.restore(snap)
                    acc = 0
                q += 1
                if not (rho < 0 and q < 10):
                    break
            tr["chi2"].append(cur); tr["lam"].append(lam); tr["trials"].append(q); tr["accepted"].append(acc)
            if q == 10 or rho == 0:
                break
            n_bad = n_bad + 1 if (ini - cur) * 1e3 < ini else 0
            if n_bad >= 3:
                break
        return {k: np.array(v) for k, v in tr.items()}


def pose_optimization_dense(K, pose7_in, X, obs, info, stereo):
    """Independent formulation of Optimizer::PoseOptimization (reference src/Optimizer.cc:244-456) for pinning
    ba_oracle_pose_optimization: 4x4 poses, Taylor-series exponential, central-difference Jacobians of the residuals with
    respect to the left perturbation exp(d) T, scipy solve of the 6x6 system.  Shared with the oracle only as
    specification: residuals (types_six_dof_expmap.cpp:290-306, float 1/z and DOUBLE bf in the stereo only-pose edge),
    Huber re-weighting, the LM schedule and the four-round inlier/outlier logic.
    Returns dict(T (4,4), outlier (n,), n_inliers, iters (4,), trace (4,10,3))."""
    n = len(info)
    T0 = T_from_pose7(np.asarray(pose7_in, float))
    dM, dS = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815)))

    def res(T, k, exact=False):
        if not stereo[k]:
            return res_mono(T, X[k], K, obs[k])
        p = T[:3, :3] @ X[k] + T[:3, 3]
        invz = 1.0 / p[2] if exact else float(np.float32(1.0) / np.float32(p[2]))
        u = p[0] * invz * K[0] + K[2]
        v = p[1] * invz * K[1] + K[3]
        return np.array([obs[k][0] - u, obs[k][1] - v, obs[k][2] - (u - K[4] * invz)])

    def errors(T, level, robust):
        chi, c2 = 0.0, np.zeros(n)
        for k in range(n):
            if level[k]:
                continue
            e = res(T, k)
            c2[k] = info[k] * float(e @ e)
            chi += huber(c2[k], (dS if stereo[k] else dM) if robust else 0.0)[0]
        return chi, c2

    outlier = np.zeros(n, np.uint8)
    level = np.zeros(n, np.uint8)
    trace = np.full((4, 10, 3), np.nan)
    iters = np.zeros(4, np.int32)
    if n < 3:
        return dict(T=T0, outlier=outlier, n_inliers=0, iters=iters, trace=trace)
    chi2_last = np.zeros(n)
    robust = True
    T = T0.copy()
    n_bad_edges = 0
    h = 1e-6
    for rnd in range(4):
        T = T0.copy()
        lam, ni, n_bad = 0.0, 2.0, 0
        for it in range(10):
            cur, c2 = errors(T, level, robust)
            chi2_last[level == 0] = c2[level == 0]
            ini = cur
            H, b = np.zeros((6, 6)), np.zeros(6)
            for k in range(n):
                if level[k]:
                    continue
                e = res(T, k)
                J = np.zeros((len(e), 6))
                for a in range(6):
                    d = np.zeros(6)
                    d[a] = h
                    J[:, a] = (res(exp_series(d) @ T, k, exact=True) - res(exp_series(-d) @ T, k, exact=True)) / (2 * h)
                _, r1 = huber(info[k] * float(e @ e), (dS if stereo[k] else dM) if robust else 0.0)
                H += r1 * info[k] * (J.T @ J)
                b += J.T @ (-info[k] * e) * r1
            if it == 0:
                lam, ni, n_bad = 1e-5 * np.abs(np.diag(H)).max(), 2.0, 0
            rho, qmax = 0.0, 0
            while True:
                Tb = T.copy()
                try:
                    x = sla.cho_solve(sla.cho_factor(H + lam * np.eye(6)), b)
                    T = exp_series(x) @ T
                    tmp, c2 = errors(T, level, robust)
                    chi2_last[level == 0] = c2[level == 0]
                except sla.LinAlgError:
                    x = np.zeros(6)
                    tmp = np.inf
                rho = (cur - tmp) / (float(x @ (lam * x + b)) + 1e-3)
                if rho > 0 and np.isfinite(tmp):
                    alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                    lam *= max(1.0 / 3.0, alpha)
                    ni = 2.0
                    cur = tmp
                else:
                    lam *= ni
                    ni *= 2
                    T = Tb
                qmax += 1
                if not (rho < 0 and qmax < 10):
                    break
            trace[rnd, it] = (cur, lam, qmax)
            iters[rnd] += 1
            if qmax == 10 or rho == 0:
                break
            n_bad = n_bad + 1 if (ini - cur) * 1e3 < ini else 0
            if n_bad >= 3:
                break
        n_bad_edges = 0
        for k in range(n):
            if outlier[k]:
                e = res(T, k)
                chi2_last[k] = info[k] * float(e @ e)
            c = np.float32(chi2_last[k])
            if c > np.float32(7.815 if stereo[k] else 5.991):
                outlier[k], level[k] = 1, 1
                n_bad_edges += 1
            else:
                outlier[k], level[k] = 0, 0
        if rnd == 2:
            robust = False
        if n < 10:
            break
    return dict(T=T, outlier=outlier, n_inliers=n - n_bad_edges, iters=iters, trace=trace)
