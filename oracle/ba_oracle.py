"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/ba_oracle.c (the CPU restatement of hot path B) plus the
two-stage schedule of Optimizer::LocalJointBundleAdjustment (src/Optimizer_util.cc:598-661) and the single-stage one of
JointBundleAdjustment (:44-307).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
up = C.POINTER(C.c_ubyte)
lp = C.POINTER(C.c_longlong)


class Problem(C.Structure):
    _fields_ = [("n_kf", C.c_int), ("n_pt", C.c_int), ("n_obj", C.c_int), ("n_mono", C.c_int), ("n_stereo", C.c_int),
                ("n_oe", C.c_int),
                ("kf_pose", dp), ("kf_fixed", up), ("kf_id", lp), ("kf_K", dp), ("pt_xyz", dp), ("pt_id", lp),
                ("obj_pose", dp), ("obj_id", lp), ("mono_pt", ip), ("mono_kf", ip), ("mono_obs", dp), ("mono_info", dp),
                ("st_pt", ip), ("st_kf", ip), ("st_obs", dp), ("st_info", dp), ("oe_kf", ip), ("oe_obj", ip),
                ("oe_meas", dp), ("oe_info", C.c_double),
                ("mono_level", up), ("st_level", up), ("oe_level", up), ("mono_chi2", dp), ("st_chi2", dp),
                ("oe_chi2", dp)]


class Opts(C.Structure):
    _fields_ = [("delta_mono", C.c_double), ("delta_stereo", C.c_double), ("delta_obj", C.c_double), ("stop_flag", up)]


class Trace(C.Structure):
    _fields_ = [("cap", C.c_int), ("n", C.c_int), ("chi2", dp), ("lam", dp), ("trials", ip), ("accepted", ip),
                ("kf_hidx", ip), ("obj_hidx", ip), ("pt_hidx", ip), ("result", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "_build", "libba_oracle.so")
        if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "ba_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _LIB = C.CDLL(so)
        _LIB.ba_oracle_optimize.argtypes = [C.POINTER(Problem), C.c_int, C.POINTER(Opts), C.POINTER(Trace)]
        _LIB.ba_oracle_optimize.restype = C.c_int
        _LIB.ba_oracle_depth_positive.argtypes = [C.POINTER(Problem), up, up]
        _LIB.ba_se3_exp.argtypes = [dp, dp]
        _LIB.ba_se3_log.argtypes = [dp, dp]
        _LIB.ba_mono_edge.argtypes = [dp, dp, dp, dp, dp, dp, dp]
        _LIB.ba_mono_edge.restype = C.c_double
        _LIB.ba_stereo_edge.argtypes = [dp, dp, dp, dp, dp, dp, dp]
        _LIB.ba_stereo_edge.restype = C.c_double
        _LIB.ba_obj_edge.argtypes = [dp, dp, dp, dp, dp, dp]
        _LIB.ba_oracle_pose_optimization.argtypes = [C.c_int, dp, dp, dp, dp, dp, up, dp, up, dp, ip]
        _LIB.ba_oracle_pose_optimization.restype = C.c_int
        _LIB.ba_oracle_set_sparse_solver.argtypes = [C.c_int]
    return _LIB


def set_sparse_solver(on):
    """the reduced camera system by a block-sparse Cholesky in minimum-degree order (what g2o's LinearSolverEigen does with AMD +
    sparse LDL^T, linear_solver_eigen.h:94-124,147-201) instead of the dense solve; the TIMED baseline of bench.py uses it, the
    parity tests keep the dense one.  Process-wide switch of the oracle library."""
    lib().ba_oracle_set_sparse_solver(1 if on else 0)


def _p(a, t):
    return a.ctypes.data_as(t)


class BaProblem(object):
    """Owns copies of a scene's arrays (qsp_slam_amd.synth.make_ba_scene layout) and the C struct over them."""

    def __init__(self, scene):
        s = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in scene.items()}
        self.s = s
        nm, ns, no = len(s["mono_pt"]), len(s["st_pt"]), len(s["oe_kf"])
        s["mono_level"] = np.zeros(max(nm, 1), np.uint8)
        s["st_level"] = np.zeros(max(ns, 1), np.uint8)
        s["oe_level"] = np.zeros(max(no, 1), np.uint8)
        s["mono_chi2"] = np.zeros(max(nm, 1))
        s["st_chi2"] = np.zeros(max(ns, 1))
        s["oe_chi2"] = np.zeros(max(no, 1))
        for k in ("mono_pt", "mono_kf", "st_pt", "st_kf", "oe_kf", "oe_obj"):
            s[k] = np.ascontiguousarray(s[k], np.int32) if len(s[k]) else np.zeros(1, np.int32)
        for k in ("mono_obs", "mono_info", "st_obs", "st_info", "oe_meas"):
            s[k] = np.ascontiguousarray(s[k], np.float64) if s[k].size else np.zeros(8, np.float64)
        self.c = Problem(len(s["kf_pose"]), len(s["pt_xyz"]), len(s["obj_pose"]), nm, ns, no,
                         _p(s["kf_pose"], dp), _p(s["kf_fixed"], up), _p(s["kf_id"], lp), _p(s["kf_K"], dp),
                         _p(s["pt_xyz"], dp), _p(s["pt_id"], lp), _p(s["obj_pose"], dp), _p(s["obj_id"], lp),
                         _p(s["mono_pt"], ip), _p(s["mono_kf"], ip), _p(s["mono_obs"], dp), _p(s["mono_info"], dp),
                         _p(s["st_pt"], ip), _p(s["st_kf"], ip), _p(s["st_obs"], dp), _p(s["st_info"], dp),
                         _p(s["oe_kf"], ip), _p(s["oe_obj"], ip), _p(s["oe_meas"], dp), float(s["oe_info"]),
                         _p(s["mono_level"], up), _p(s["st_level"], up), _p(s["oe_level"], up),
                         _p(s["mono_chi2"], dp), _p(s["st_chi2"], dp), _p(s["oe_chi2"], dp))
        self.nm, self.ns, self.no = nm, ns, no

    def optimize(self, n_iter, delta_mono=0.0, delta_stereo=0.0, delta_obj=0.0, stop=None):
        """SparseOptimizer::optimize(n_iter) on the active (level 0) edges -> trace dict"""
        cap = max(n_iter, 1)
        chi2, lam = np.zeros(cap), np.zeros(cap)
        trials, acc = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        kh = np.zeros(max(self.c.n_kf, 1), np.int32)
        oh = np.zeros(max(self.c.n_obj, 1), np.int32)
        ph = np.zeros(max(self.c.n_pt, 1), np.int32)
        tr = Trace(cap, 0, _p(chi2, dp), _p(lam, dp), _p(trials, ip), _p(acc, ip), _p(kh, ip), _p(oh, ip), _p(ph, ip), 0)
        flag = np.zeros(1, np.uint8) if stop is None else stop
        o = Opts(delta_mono, delta_stereo, delta_obj, _p(flag, up))
        done = lib().ba_oracle_optimize(C.byref(self.c), n_iter, C.byref(o), C.byref(tr))
        n = tr.n
        return dict(iterations=done, chi2=chi2[:n].copy(), lam=lam[:n].copy(), trials=trials[:n].copy(),
                    accepted=acc[:n].copy(), kf_hidx=kh[: self.c.n_kf].copy(), obj_hidx=oh[: self.c.n_obj].copy(),
                    pt_hidx=ph[: self.c.n_pt].copy(), result=tr.result)

    def depth_positive(self):
        m, s = np.zeros(max(self.nm, 1), np.uint8), np.zeros(max(self.ns, 1), np.uint8)
        lib().ba_oracle_depth_positive(C.byref(self.c), _p(m, up), _p(s, up))
        return m[: self.nm].astype(bool), s[: self.ns].astype(bool)

    def local_joint_ba(self, stop=None):
        """Optimizer::LocalJointBundleAdjustment schedule, src/Optimizer_util.cc:598-661: optimize(5) with Huber
        sqrt(5.991) / sqrt(7.815) / sqrt(1e3); mark chi2 > 5.991 / 7.815 / 1e3 or non-positive depth as level 1; drop
        the robust kernels; optimize(10)."""
        s = self.s
        t1 = self.optimize(5, np.float32(np.sqrt(5.991)), np.float32(np.sqrt(7.815)), np.float32(np.sqrt(1e3)), stop)
        mp, sp = self.depth_positive()
        s["mono_level"][: self.nm] = ((s["mono_chi2"][: self.nm] > 5.991) | ~mp).astype(np.uint8)
        s["st_level"][: self.ns] = ((s["st_chi2"][: self.ns] > 7.815) | ~sp).astype(np.uint8)
        s["oe_level"][: self.no] = (s["oe_chi2"][: self.no] > 1e3).astype(np.uint8)
        t2 = self.optimize(10, 0.0, 0.0, 0.0, stop)
        return t1, t2

    def state(self):
        return self.s["kf_pose"].copy(), self.s["pt_xyz"].copy(), self.s["obj_pose"].copy()


def pose_optimization(K, pose, X, obs, info, stereo):
    """Optimizer::PoseOptimization (src/Optimizer.cc:244-456) on flattened inputs: K (5,) fx fy cx cy bf, pose (7,) T_cw,
    X (n,3) world points, obs (n,3) u v u_right (u_right ignored for mono rows), info (n,) invSigma2, stereo (n,) 0/1.
    Returns dict(pose (7,), outlier (n,) uint8, n_inliers, iters (4,), trace (4,10,3): chi2, lambda, trials)."""
    n = len(info)
    K = np.ascontiguousarray(K, np.float64)
    pose = np.ascontiguousarray(pose, np.float64)
    X = np.ascontiguousarray(X, np.float64).reshape(-1, 3)
    obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 3)
    info = np.ascontiguousarray(info, np.float64)
    stereo = np.ascontiguousarray(stereo, np.uint8)
    out = np.zeros(7)
    outlier = np.zeros(max(n, 1), np.uint8)
    trace = np.full((4, 10, 3), np.nan)
    iters = np.zeros(4, np.int32)
    r = lib().ba_oracle_pose_optimization(n, _p(K, dp), _p(pose, dp), _p(X, dp), _p(obs, dp), _p(info, dp), _p(stereo, up),
                                          _p(out, dp), _p(outlier, up), _p(trace, dp), _p(iters, ip))
    return dict(pose=out, outlier=outlier[:n], n_inliers=int(r), iters=iters, trace=trace)
