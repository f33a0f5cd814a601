"""Host-side handle of the joint bundle adjustment (path B) over the C-ABI of include/qsp_hip.h.

`scene` is the flattened g2o graph (see qsp_ba_scene in the header and qsp_slam_amd.synth.make_ba_scene): what
Optimizer::LocalJointBundleAdjustment / JointBundleAdjustment (src/Optimizer_util.cc) assemble from KeyFrame / MapPoint /
MapObject getters.  The C++ shim that does that flattening inside ORB-SLAM2 is include/qsp_optimizer_shim.h."""
import ctypes as C

import numpy as np

from . import _lib


def _arr(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a if a.size else np.zeros(8, dt)


class Trace(object):
    def __init__(self, cap):
        cap = max(int(cap), 1)
        self.chi2, self.lam = np.zeros(cap), np.zeros(cap)
        self.trials, self.accepted = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        self.c = _lib.BaTrace(cap, 0, _lib.dptr(self.chi2), _lib.dptr(self.lam), _lib.i32ptr(self.trials),
                              _lib.i32ptr(self.accepted), 0, 0, 0, 0)

    def dict(self):
        n = self.c.n
        return dict(iterations=self.c.iterations, chi2=self.chi2[:n].copy(), lam=self.lam[:n].copy(),
                    trials=self.trials[:n].copy(), accepted=self.accepted[:n].copy(), result=self.c.result,
                    n_pose_blocks=self.c.n_pose_blocks, n_landmarks=self.c.n_landmarks)


class BaProblem(object):
    def __init__(self, scene, device=0):
        L = _lib.lib()
        s = scene
        self.n_kf, self.n_pt, self.n_obj = len(s["kf_pose"]), len(s["pt_xyz"]), len(s["obj_pose"])
        self.nm, self.ns, self.no = len(s["mono_pt"]), len(s["st_pt"]), len(s["oe_kf"])
        k = self._keep = dict(
            kf_pose=_arr(s["kf_pose"], np.float64), kf_fixed=_arr(s["kf_fixed"], np.uint8),
            kf_id=_arr(s["kf_id"], np.int64), kf_K=_arr(s["kf_K"], np.float64), pt_xyz=_arr(s["pt_xyz"], np.float64),
            pt_id=_arr(s["pt_id"], np.int64), obj_pose=_arr(s["obj_pose"], np.float64),
            obj_id=_arr(s["obj_id"], np.int64), mono_pt=_arr(s["mono_pt"], np.int32), mono_kf=_arr(s["mono_kf"], np.int32),
            mono_obs=_arr(s["mono_obs"], np.float64), mono_info=_arr(s["mono_info"], np.float64),
            st_pt=_arr(s["st_pt"], np.int32), st_kf=_arr(s["st_kf"], np.int32), st_obs=_arr(s["st_obs"], np.float64),
            st_info=_arr(s["st_info"], np.float64), oe_kf=_arr(s["oe_kf"], np.int32), oe_obj=_arr(s["oe_obj"], np.int32),
            oe_meas=_arr(s["oe_meas"], np.float64))
        sc = _lib.BaScene(self.n_kf, self.n_pt, self.n_obj, self.nm, self.ns, self.no,
                          _lib.dptr(k["kf_pose"]), _lib.u8ptr(k["kf_fixed"]), _lib.i64ptr(k["kf_id"]),
                          _lib.dptr(k["kf_K"]), _lib.dptr(k["pt_xyz"]), _lib.i64ptr(k["pt_id"]),
                          _lib.dptr(k["obj_pose"]), _lib.i64ptr(k["obj_id"]),
                          _lib.i32ptr(k["mono_pt"]), _lib.i32ptr(k["mono_kf"]), _lib.dptr(k["mono_obs"]),
                          _lib.dptr(k["mono_info"]), _lib.i32ptr(k["st_pt"]), _lib.i32ptr(k["st_kf"]),
                          _lib.dptr(k["st_obs"]), _lib.dptr(k["st_info"]), _lib.i32ptr(k["oe_kf"]),
                          _lib.i32ptr(k["oe_obj"]), _lib.dptr(k["oe_meas"]), float(s["oe_info"]))
        h = C.c_void_p()
        _lib.check(L.qsp_ba_create(C.byref(sc), int(device), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().qsp_ba_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_shard(self, rank, world, allreduce):
        """landmark sharding over `world` ranks; `allreduce(dev_ptr: int, count: int, hip_stream: int)` sums `count` float64
        at the device address in place over all ranks (qsp_slam_amd.parallel.TorchAllreduce wraps torch.distributed)."""
        def _cb(ctx, buf, n, stream):
            try:
                allreduce(int(buf), int(n), int(stream or 0))
                return 0
            except Exception as e:          # never let an exception cross the C boundary
                print("qsp all-reduce callback failed:", repr(e))
                return 1
        self._cb = _lib.ALLREDUCE_FN(_cb) if world > 1 else _lib.ALLREDUCE_FN(0)
        _lib.check(_lib.lib().qsp_ba_set_shard(self.handle, int(rank), int(world), self._cb, None))

    def set_shard_rccl(self, comm):
        """landmark sharding with the collectives issued by the library: ncclAllReduce on the problem's own stream
        (qsp_ba_set_shard_rccl).  `comm` is a qsp_slam_amd.parallel.RcclComm (one rank per GPU)."""
        self._comm = comm                       # keep the communicator alive as long as the problem uses it
        _lib.check(_lib.lib().qsp_ba_set_shard_rccl(self.handle, int(comm.rank), int(comm.world), comm.nccl()))

    def set_object_elimination(self, on=True):
        """objects eliminated in closed form in front of the dense solve (default) or kept inside it (qsp_ba_set_option)"""
        _lib.check(_lib.lib().qsp_ba_set_option(self.handle, 1, 1 if on else 0))

    def set_cholesky_chain(self, on=True):
        """the dense factorisation as one launch (a resident chain workgroup + tile workgroups taking tickets; the default) or as
        one launch per block step; same bits either way (QSP_BA_OPT_CHOLESKY_CHAIN)"""
        _lib.check(_lib.lib().qsp_ba_set_option(self.handle, 2, 1 if on else 0))

    @property
    def cholesky_chain(self):
        prof = _lib.BaProfile()
        L = _lib.lib()
        # (qsp_ba_profile reports and sets the profiling switch: read it, then put the switch back)
        _lib.check(L.qsp_ba_profile(self.handle, 1 if getattr(self, "_profiling", False) else 0, C.byref(prof)))
        return bool(prof.cholesky_chain)

    def set_deterministic(self, on=True):
        """no atomics in the Schur complement: repeated runs give the same bits (qsp_ba_set_deterministic)"""
        _lib.check(_lib.lib().qsp_ba_set_deterministic(self.handle, 1 if on else 0))

    def set_levels(self, mono=None, stereo=None, obj=None):
        def p(a):
            return _lib.c_uint8_p() if a is None else _lib.u8ptr(_arr(a, np.uint8))
        keep = [None if a is None else _arr(a, np.uint8) for a in (mono, stereo, obj)]
        _lib.check(_lib.lib().qsp_ba_set_levels(self.handle, *[(_lib.c_uint8_p() if a is None else _lib.u8ptr(a))
                                                               for a in keep]))

    def optimize(self, n_iter, delta_mono=0.0, delta_stereo=0.0, delta_obj=0.0, stop=None):
        tr = Trace(n_iter)
        flag = _lib.c_uint8_p() if stop is None else _lib.u8ptr(stop)
        _lib.check(_lib.lib().qsp_ba_optimize(self.handle, int(n_iter), float(delta_mono), float(delta_stereo),
                                              float(delta_obj), flag, C.byref(tr.c)))
        d = tr.dict()
        kh, oh, ph = self.index()
        d.update(kf_hidx=kh, obj_hidx=oh, pt_hidx=ph)
        return d

    def local_joint_ba(self, stop=None):
        t1, t2 = Trace(5), Trace(10)
        flag = _lib.c_uint8_p() if stop is None else _lib.u8ptr(stop)
        _lib.check(_lib.lib().qsp_ba_local_joint(self.handle, flag, C.byref(t1.c), C.byref(t2.c)))
        return t1.dict(), t2.dict()

    def index(self):
        kh = np.zeros(max(self.n_kf, 1), np.int32)
        oh = np.zeros(max(self.n_obj, 1), np.int32)
        ph = np.zeros(max(self.n_pt, 1), np.int32)
        _lib.check(_lib.lib().qsp_ba_get_index(self.handle, _lib.i32ptr(kh), _lib.i32ptr(oh), _lib.i32ptr(ph)))
        return kh[: self.n_kf], oh[: self.n_obj], ph[: self.n_pt]

    def state(self):
        kf = np.zeros((max(self.n_kf, 1), 7))
        pt = np.zeros((max(self.n_pt, 1), 3))
        ob = np.zeros((max(self.n_obj, 1), 7))
        _lib.check(_lib.lib().qsp_ba_get_state(self.handle, _lib.dptr(kf), _lib.dptr(pt), _lib.dptr(ob)))
        return kf[: self.n_kf], pt[: self.n_pt], ob[: self.n_obj]

    def set_state(self, kf=None, pt=None, ob=None):
        keep = [None if a is None else _arr(a, np.float64) for a in (kf, pt, ob)]
        _lib.check(_lib.lib().qsp_ba_set_state(self.handle, *[(_lib.c_double_p() if a is None else _lib.dptr(a))
                                                              for a in keep]))

    def edges(self):
        cm, cs, co = np.zeros(max(self.nm, 1)), np.zeros(max(self.ns, 1)), np.zeros(max(self.no, 1))
        pm, ps = np.zeros(max(self.nm, 1), np.uint8), np.zeros(max(self.ns, 1), np.uint8)
        _lib.check(_lib.lib().qsp_ba_get_edges(self.handle, _lib.dptr(cm), _lib.dptr(cs), _lib.dptr(co), _lib.u8ptr(pm),
                                               _lib.u8ptr(ps)))
        return dict(mono_chi2=cm[: self.nm], st_chi2=cs[: self.ns], oe_chi2=co[: self.no],
                    mono_pos=pm[: self.nm].astype(bool), st_pos=ps[: self.ns].astype(bool))

    def profile(self, enable=True):
        self._profiling = bool(enable)
        p = _lib.BaProfile()
        _lib.check(_lib.lib().qsp_ba_profile(self.handle, 1 if enable else 0, C.byref(p)))
        return p


def release_caches():
    """give back the streams, device chunks and pinned buffers qsp_ba_destroy keeps for the next problem (qsp_ba_release_caches)"""
    _lib.lib().qsp_ba_release_caches()


class PoseOptimizer(object):
    """Optimizer::PoseOptimization (reference src/Optimizer.cc:244-456) on flattened inputs, one kernel launch per call
    (qsp_pose_optimize, include/qsp_hip.h)."""

    def __init__(self, max_points=4096, device=0):
        self.handle = C.c_void_p()
        _lib.check(_lib.lib().qsp_pose_optimizer_create(int(device), int(max_points), C.byref(self.handle)))

    def close(self):
        if self.handle is not None and self.handle.value:
            _lib.lib().qsp_pose_optimizer_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def optimize(self, K, pose, X, obs, info, stereo):
        """K (5,) fx fy cx cy bf; pose (7,) T_cw; X (n,3); obs (n,3) u v u_right; info (n,); stereo (n,) 0/1.
        Returns dict(pose (7,), outlier (n,) uint8, n_inliers, iters (4,), trace (4,10,3) chi2 / lambda / trials)."""
        n = int(np.size(info))                 # (_arr pads empty arrays so that their pointers stay valid)
        info = _arr(info, np.float64)
        K, pose = _arr(K, np.float64), _arr(pose, np.float64)
        X, obs = _arr(np.reshape(X, (-1, 3)), np.float64), _arr(np.reshape(obs, (-1, 3)), np.float64)
        stereo = _arr(stereo, np.uint8)
        out = np.zeros(7)
        outlier = np.zeros(max(n, 1), np.uint8)
        ninl = C.c_int32()
        tr = _lib.PoseTrace()
        _lib.check(_lib.lib().qsp_pose_optimize(self.handle, n, _lib.dptr(K), _lib.dptr(pose), _lib.dptr(X), _lib.dptr(obs),
                                                _lib.dptr(info), _lib.u8ptr(stereo), _lib.dptr(out), _lib.u8ptr(outlier),
                                                C.byref(ninl), C.byref(tr)))
        trace = np.array(tr.trace[:], np.float64).reshape(4, 10, 3)
        return dict(pose=out, outlier=outlier[:n], n_inliers=int(ninl.value), iters=np.array(tr.iters[:], np.int32), trace=trace)
