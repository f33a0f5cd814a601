"""Drop-in for reconstruct/optimizer.py of the reference: same class names, constructor and method signatures, return
types and failure behaviour; the numerics are batched HIP launches behind the C-ABI (include/qsp_hip.h).

Added on top of the reference interface (not replacing it): `reconstruct_objects_batched`, which runs many objects x
yaw-flip hypotheses in one resident batch -- what src/LocalMapping_util.cc:705-760 does as 4 serial Python calls per
object."""
import ctypes as C
import math
import time

import numpy as np

from .. import _lib
from .utils import ForceKeyErrorDict


def _joint_cfg(o):
    return _lib.JointCfg(o.k1, o.k2, o.k3, o.k4, o.b1, o.b2, o.lr, o.s_damp, o.cut_off, int(o.num_iterations_joint_optim),
                         int(o.num_depth_samples), int(o.code_len))


def _flip_rotation(T, k, flip_angle):
    """T with its rotation block right-multiplied by Eigen::AngleAxisf(double(k) * flip_sample_angle, e_y).matrix()
    (src/LocalMapping_util.cc:722-726): the angle is narrowed to float before cos / sin, the (1,1) entry is (1-c)+c, and
    the 3x3 product sums over k in increasing order in float."""
    Tk = np.array(T, dtype=np.float32).reshape(4, 4).copy()
    if k == 0:
        return Tk
    f = np.float32
    a = float(f(float(k) * flip_angle))
    c, s = f(math.cos(a)), f(math.sin(a))
    Ry = np.array([[c, 0, s], [0, f(f(1) - c) + c, 0], [f(0) - s, 0, c]], dtype=np.float32)
    R = Tk[:3, :3].copy()
    for j in range(3):
        Tk[:3, j] = (R[:, 0] * Ry[0, j] + R[:, 1] * Ry[1, j]) + R[:, 2] * Ry[2, j]
    return Tk


def _reconstruct_objects(decoder, cfg, pts, rays, depth, hyp_obj, t_cam_obj, code):
    """qsp_reconstruct_objects: one call = fill + set_state + run + get on the batch that stays resident with the decoder (no device
    allocation per call once its capacities have settled) -- the reference's call pattern, src/LocalMapping_util.cc:705-760"""
    n_hyp = len(hyp_obj)
    L = decoder.code_len
    pts = [_lib.f32c(p).reshape(-1, 3) for p in pts]
    rays = [_lib.f32c(r).reshape(-1, 3) for r in rays]
    depth = [_lib.f32c(d).reshape(-1) for d in depth]
    n_pts = np.array([p.shape[0] for p in pts], np.int32)
    n_rays = np.array([r.shape[0] for r in rays], np.int32)
    n_fg = np.array([d.shape[0] for d in depth], np.int32)
    hyp = np.ascontiguousarray(hyp_obj, dtype=np.int32)
    pp, rp, dp = _lib.ptr_array(pts), _lib.ptr_array(rays), _lib.ptr_array(depth)
    T0 = _lib.f32c(t_cam_obj).reshape(n_hyp, 16)
    c0 = None if code is None else _lib.f32c(code).reshape(n_hyp, L)
    T = np.empty((n_hyp, 4, 4), np.float32)
    c = np.empty((n_hyp, L), np.float32)
    loss = np.empty(n_hyp, np.float32)
    good = np.empty(n_hyp, np.uint8)
    _lib.check(_lib.lib().qsp_reconstruct_objects(
        decoder.handle, C.byref(cfg), len(pts), C.cast(pp, C.POINTER(_lib.c_float_p)), _lib.i32ptr(n_pts),
        C.cast(rp, C.POINTER(_lib.c_float_p)), _lib.i32ptr(n_rays), C.cast(dp, C.POINTER(_lib.c_float_p)), _lib.i32ptr(n_fg),
        n_hyp, _lib.i32ptr(hyp), _lib.fptr(T0), _lib.fptr(c0) if c0 is not None else _lib.c_float_p(), _lib.fptr(T), _lib.fptr(c),
        _lib.fptr(loss), _lib.u8ptr(good)))
    return T, c, loss, good.astype(bool)


class RefineBatch(object):
    """Thin owner of a qsp_refine_batch* (resident device batch)."""

    def __init__(self, decoder, cfg, pts, rays, depth, hyp_obj):
        L = _lib.lib()
        self.n_obj = len(pts)
        self.n_hyp = len(hyp_obj)
        self.code_len = decoder.code_len
        self._pts = [_lib.f32c(p).reshape(-1, 3) for p in pts]
        self._rays = [_lib.f32c(r).reshape(-1, 3) for r in rays]
        self._depth = [_lib.f32c(d).reshape(-1) for d in depth]
        n_pts = np.array([p.shape[0] for p in self._pts], np.int32)
        n_rays = np.array([r.shape[0] for r in self._rays], np.int32)
        n_fg = np.array([d.shape[0] for d in self._depth], np.int32)
        hyp = np.ascontiguousarray(hyp_obj, dtype=np.int32)
        pp, rp, dp = _lib.ptr_array(self._pts), _lib.ptr_array(self._rays), _lib.ptr_array(self._depth)
        h = C.c_void_p()
        _lib.check(L.qsp_refine_batch_create(decoder.handle, C.byref(cfg), self.n_obj,
                                             C.cast(pp, C.POINTER(_lib.c_float_p)), _lib.i32ptr(n_pts),
                                             C.cast(rp, C.POINTER(_lib.c_float_p)), _lib.i32ptr(n_rays),
                                             C.cast(dp, C.POINTER(_lib.c_float_p)), _lib.i32ptr(n_fg),
                                             self.n_hyp, _lib.i32ptr(hyp), C.byref(h)))
        self.handle = h
        self.decoder = decoder

    def set_state(self, t_cam_obj, code=None):
        T = _lib.f32c(t_cam_obj).reshape(self.n_hyp, 16)
        c = None if code is None else _lib.f32c(code).reshape(self.n_hyp, self.code_len)
        _lib.check(_lib.lib().qsp_refine_batch_set_state(self.handle, _lib.fptr(T), _lib.fptr(c) if c is not None
                                                         else _lib.c_float_p()))

    def run(self, n_iter=0):
        _lib.check(_lib.lib().qsp_refine_batch_run(self.handle, int(n_iter)))

    def get(self):
        T = np.empty((self.n_hyp, 4, 4), np.float32)
        code = np.empty((self.n_hyp, self.code_len), np.float32)
        loss = np.empty(self.n_hyp, np.float32)
        good = np.empty(self.n_hyp, np.uint8)
        _lib.check(_lib.lib().qsp_refine_batch_get(self.handle, _lib.fptr(T), _lib.fptr(code), _lib.fptr(loss),
                                                   _lib.u8ptr(good)))
        return T, code, loss, good.astype(bool)

    def trace(self):
        n = self.n_hyp
        H = np.empty((n, 71, 71), np.float32)
        b = np.empty((n, 71), np.float32)
        dx = np.empty((n, 71), np.float32)
        nv = np.empty(n, np.int32)
        nr = np.empty(n, np.int32)
        lt = np.empty((n, 2), np.float32)
        _lib.check(_lib.lib().qsp_refine_batch_trace(self.handle, _lib.fptr(H), _lib.fptr(b), _lib.fptr(dx),
                                                     _lib.i32ptr(nv), _lib.i32ptr(nr), _lib.fptr(lt)))
        return dict(H=H, b=b, dx=dx, n_valid=nv, K=nr, loss_sdf=lt[:, 0], loss_render=lt[:, 1])

    def trace_rot(self):
        """(n_hyp, 4): the rotation prior's J_rot (entries 3..5 of J_sim3) and res_rot of the last iteration (loss.py:155-178)"""
        r = np.empty((self.n_hyp, 4), np.float32)
        _lib.check(_lib.lib().qsp_refine_batch_trace_rot(self.handle, _lib.fptr(r)))
        return r

    def enable_rows(self, enable=True):
        _lib.check(_lib.lib().qsp_refine_batch_rows(self.handle, 1 if enable else 0, 0, _lib.c_float_p(),
                                                    _lib.c_float_p()))

    def rows(self, hyp, n_pts, n_render):
        """augmented Jacobian rows [J_pose(7) | J_code(64) | robust residual] of the last iteration (parity tests)"""
        a = np.empty((max(n_pts, 1), 72), np.float32)
        r = np.empty((max(n_render, 1), 72), np.float32)
        _lib.check(_lib.lib().qsp_refine_batch_rows(self.handle, 1, int(hyp), _lib.fptr(a), _lib.fptr(r)))
        return a[:n_pts], r[:n_render]

    def profile(self, enable=True):
        p = _lib.RefineProfile()
        _lib.check(_lib.lib().qsp_refine_batch_profile(self.handle, 1 if enable else 0, C.byref(p)))
        return p

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().qsp_refine_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Optimizer(object):
    """reconstruct/optimizer.py:26-281.  `decoder` is a qsp_slam_amd.decoder.DeepSdfDecoder."""

    def __init__(self, decoder, configs, debug=False):
        self.decoder = decoder
        optim_cfg = configs.optimizer
        self.k1 = optim_cfg.joint_optim.k1
        self.k2 = optim_cfg.joint_optim.k2
        self.k3 = optim_cfg.joint_optim.k3
        self.k4 = optim_cfg.joint_optim.k4
        self.b1 = optim_cfg.joint_optim.b1
        self.b2 = optim_cfg.joint_optim.b2
        self.lr = optim_cfg.joint_optim.learning_rate
        self.s_damp = optim_cfg.joint_optim.scale_damping
        self.num_iterations_joint_optim = optim_cfg.joint_optim.num_iterations
        self.code_len = optim_cfg.code_len
        self.num_depth_samples = optim_cfg.num_depth_samples
        self.cut_off = optim_cfg.cut_off_threshold
        self.debug = debug
        if configs.data_type == "KITTI":
            self.num_iterations_pose_only = optim_cfg.pose_only_optim.num_iterations

    # ---- reference entry point: one object, one hypothesis -----------------------------------------------------------
    def reconstruct_object(self, t_cam_obj, pts, rays, depth, code=None):
        """Same contract as the reference: returns an object with attrs t_cam_obj (4,4) f32 | None, code (L,) f32 | None,
        is_good, loss.  No exception for numeric failure."""
        r = self.reconstruct_objects_batched([dict(t_cam_obj=t_cam_obj, pts=pts, rays=rays, depth=depth, code=code)],
                                             flip_sample_num=1, select=False)
        return r[0][0]

    # ---- batched form: objects x yaw flips in one launch sequence ------------------------------------------------------
    def reconstruct_objects_batched(self, objects, flip_sample_num=1, select=True):
        """objects: list of dicts(t_cam_obj, pts, rays, depth, code=None).  For every object `flip_sample_num`
        hypotheses are refined: hypothesis k starts from t_cam_obj with its ROTATION BLOCK right-multiplied by
        R_y(k * 2pi / flip_sample_num) (src/LocalMapping_util.cc:713-726).
        select=False -> list (per object) of lists (per flip) of result objects;
        select=True  -> list (per object) of the result the reference's selection rule keeps
                        (LocalMapping_util.cc:748-752: replace if the kept one is not good, or if the new one is good
                        and has a smaller loss)."""
        n_obj = len(objects)
        hyp_obj, T0, codes = [], [], []
        any_code = any(o.get("code") is not None for o in objects)
        for i, o in enumerate(objects):
            T = np.asarray(o["t_cam_obj"], dtype=np.float32).reshape(4, 4)
            for k in range(flip_sample_num):
                Tk = _flip_rotation(T, k, 2.0 * math.pi / flip_sample_num)
                hyp_obj.append(i)
                T0.append(Tk)
                c0 = o.get("code")
                codes.append(np.zeros(self.code_len, np.float32) if c0 is None
                             else np.asarray(c0, np.float32)[: self.code_len])
        T, code, loss, good = _reconstruct_objects(self.decoder, _joint_cfg(self), [o["pts"] for o in objects],
                                                   [o["rays"] for o in objects], [o["depth"] for o in objects], hyp_obj,
                                                   np.stack(T0), np.stack(codes) if any_code else None)
        out = []
        for i in range(n_obj):
            res = []
            for k in range(flip_sample_num):
                h = i * flip_sample_num + k
                if good[h]:
                    res.append(ForceKeyErrorDict(t_cam_obj=T[h].copy(), code=code[h].copy(), is_good=True,
                                                 loss=float(loss[h])))
                else:
                    res.append(ForceKeyErrorDict(t_cam_obj=None, code=None, is_good=False, loss=float(loss[h])))
            if select:
                best = res[0]
                for r in res[1:]:
                    if (not best.is_good) or (r.is_good and r.loss < best.loss):
                        best = r
                out.append(best)
            else:
                out.append(res)
        return out

    # ---- the caller's loop around reconstruct_object, on the device ---------------------------------
    def refine_detections(self, detections, flip_sample_num=4, taps=False):
        """LocalMapping::ProcessDetectedObjects' marshalling + flip loop + keep rule (src/LocalMapping_util.cc:585-760) for a
        list of detections in ONE call (qsp_refine_detections, include/qsp_hip.h).  Each detection is a dict of WORLD-frame
        inputs, as the caller holds them:
            T_cw (4,4) key-frame pose, K (4,) fx fy cx cy, T_wo (4,4) Sim3Two of the map object, code (64,) | None,
            pts_world (M,3) map points on the object, fg_px (F,2) key-point pixels, fg_world (F,3) their map points,
            bg_rays (B,3), found_good_orientation (bool, default False -> flip_sample_num hypotheses, else one).
        Returns per detection the object the reference keeps in pyMapObjectLeastLoss (t_cam_obj None when not good), with
        the extra keys kept_flip and losses; taps=True adds the assembled pts / rays / depth / initial poses."""
        n = len(detections)
        f = _lib.f32c

        def cat(key, width):
            arrs = [f(d[key]).reshape(-1, width) for d in detections]
            off = np.zeros(n + 1, np.int32)
            off[1:] = np.cumsum([a.shape[0] for a in arrs])
            flat = np.concatenate(arrs, axis=0) if off[-1] else np.zeros((1, width), np.float32)
            return off, np.ascontiguousarray(flat)

        pts_off, pts_world = cat("pts_world", 3)
        fg_off, fg_px = cat("fg_px", 2)
        fg_off2, fg_world = cat("fg_world", 3)
        if not np.array_equal(fg_off, fg_off2):
            raise ValueError("fg_px and fg_world must have one row per feature point")
        bg_off, bg_rays = cat("bg_rays", 3)
        T_cw = f(np.stack([np.asarray(d["T_cw"], np.float32).reshape(4, 4) for d in detections]))
        T_wo = f(np.stack([np.asarray(d["T_wo"], np.float32).reshape(4, 4) for d in detections]))
        K = f(np.stack([np.asarray(d["K"], np.float32).reshape(4) for d in detections]))
        any_code = any(d.get("code") is not None for d in detections)
        code = f(np.stack([np.zeros(self.code_len, np.float32) if d.get("code") is None
                           else np.asarray(d["code"], np.float32)[: self.code_len] for d in detections]))
        n_flip = np.array([1 if d.get("found_good_orientation") else int(flip_sample_num) for d in detections], np.int32)
        n_hyp = int(n_flip.sum())
        inp = _lib.Detections(n, _lib.fptr(T_cw), _lib.fptr(K), _lib.fptr(T_wo),
                              _lib.fptr(code) if any_code else _lib.c_float_p(), _lib.i32ptr(n_flip),
                              2.0 * math.pi / float(flip_sample_num), _lib.i32ptr(pts_off), _lib.fptr(pts_world),
                              _lib.i32ptr(fg_off), _lib.fptr(fg_px), _lib.fptr(fg_world), _lib.i32ptr(bg_off),
                              _lib.fptr(bg_rays))
        T = np.empty((n, 4, 4), np.float32)
        c_out = np.empty((n, self.code_len), np.float32)
        loss = np.empty(n, np.float32)
        good = np.empty(n, np.uint8)
        kept = np.empty(n, np.int32)
        losses = np.empty(n_hyp, np.float32)
        res = _lib.DetectionResults(_lib.fptr(T), _lib.fptr(c_out), _lib.fptr(loss), _lib.u8ptr(good), _lib.i32ptr(kept),
                                    _lib.fptr(losses))
        if taps:
            t_pts = np.empty((max(int(pts_off[-1]), 1), 3), np.float32)
            t_rays = np.empty((max(int(fg_off[-1] + bg_off[-1]), 1), 3), np.float32)
            t_depth = np.empty(max(int(fg_off[-1]), 1), np.float32)
            t_init = np.empty((n_hyp, 4, 4), np.float32)
            res.pts_cam, res.rays, res.depth_obs, res.t_cam_obj_init = (_lib.fptr(t_pts), _lib.fptr(t_rays),
                                                                        _lib.fptr(t_depth), _lib.fptr(t_init))
        _lib.check(_lib.lib().qsp_refine_detections(self.decoder.handle, C.byref(_joint_cfg(self)), C.byref(inp),
                                                    C.byref(res)))
        out = []
        hyp_off = np.concatenate([[0], np.cumsum(n_flip)])
        ray_off = fg_off + bg_off
        for i in range(n):
            r = ForceKeyErrorDict(t_cam_obj=T[i].copy() if good[i] else None, code=c_out[i].copy() if good[i] else None,
                                  is_good=bool(good[i]), loss=float(loss[i]), kept_flip=int(kept[i]),
                                  losses=losses[hyp_off[i]:hyp_off[i + 1]].copy())
            if taps:
                r["pts"] = t_pts[pts_off[i]:pts_off[i + 1]].copy()
                r["rays"] = t_rays[ray_off[i]:ray_off[i + 1]].copy()
                r["depth"] = t_depth[fg_off[i]:fg_off[i + 1]].copy()
                r["t_cam_obj_init"] = t_init[hyp_off[i]:hyp_off[i + 1]].copy()
            out.append(r)
        return out

    def estimate_pose_cam_obj(self, t_co_se3, scale, pts, code):
        """reconstruct/optimizer.py:47-93 -> (4,4) float32 SE3 (the reference returns a torch tensor that C++ casts to
        Eigen::Matrix4f, src/LocalMapping_util.cc:139-140; a numpy array casts the same way)."""
        T = _lib.f32c(t_co_se3).reshape(1, 16)
        sc = np.array([scale], np.float32)
        p = _lib.f32c(pts).reshape(-1, 3)
        n = np.array([p.shape[0]], np.int32)
        c = _lib.f32c(np.asarray(code)[: self.code_len]).reshape(1, -1)
        out = np.empty((1, 4, 4), np.float32)
        pp = _lib.ptr_array([p])
        _lib.check(_lib.lib().qsp_estimate_pose(self.decoder.handle, 1, _lib.fptr(T), _lib.fptr(sc),
                                                C.cast(pp, C.POINTER(_lib.c_float_p)), _lib.i32ptr(n), _lib.fptr(c),
                                                int(getattr(self, "num_iterations_pose_only", 5)), _lib.fptr(out)))
        return out[0]


def create_voxel_grid(vol_dim=128):
    """reconstruct/utils.py:98-117, including its true-division quirk: `overall_index.long() / vol_dim` is a float
    division on torch >= 1.6, so the y and x coordinates keep their fractional part."""
    i = np.arange(vol_dim ** 3, dtype=np.int64)
    size = np.float32(2.0 / (vol_dim - 1))
    v = np.zeros((vol_dim ** 3, 3), np.float32)
    v[:, 2] = (i % vol_dim).astype(np.float32)
    v[:, 1] = np.mod((i / vol_dim).astype(np.float32), np.float32(vol_dim))
    v[:, 0] = np.mod((i / vol_dim).astype(np.float32) / np.float32(vol_dim), np.float32(vol_dim))
    return v * size - np.float32(1)


class MeshExtractor(object):
    """reconstruct/optimizer.py:284-304.  The SDF volume over create_voxel_grid(voxels_dim) is decoded and triangulated on
    the GPU (qsp_mesh_extract: MLP tile kernel + Lewiner's marching cubes, include/qsp_hip.h); only vertices and faces come
    back: vertices (V,3) float64 and faces (F,3) int32, the values and the order skimage.measure.marching_cubes_lewiner +
    convert_sdf_voxels_to_mesh (reconstruct/utils.py:120-141) give for the same volume.  Like there, a volume without a zero
    crossing raises (ValueError when 0 is outside its range, RuntimeError when no cell is crossed).
    method="table": the triangulation of rounds 2-3 (float32 vertices ordered by grid point; no exception for an empty mesh)."""

    def __init__(self, decoder, code_len=64, voxels_dim=64, method="lewiner"):
        if method not in ("lewiner", "table"):
            raise ValueError("method: 'lewiner' or 'table'")
        self.decoder = decoder
        self.code_len = code_len
        self.voxels_dim = voxels_dim
        self.method = method
        self.voxel_points = create_voxel_grid(vol_dim=self.voxels_dim)
        self.handle = C.c_void_p()
        pts = _lib.f32c(self.voxel_points)
        _lib.check(_lib.lib().qsp_mesh_extractor_create(decoder.handle, voxels_dim, _lib.fptr(pts), C.byref(self.handle)))
        _lib.check(_lib.lib().qsp_mesh_extractor_set_method(self.handle, 0 if method == "lewiner" else 1))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            _lib.lib().qsp_mesh_extractor_destroy(h)
            self.handle = C.c_void_p()

    def _fetch(self, nv, nf, volume=False):
        lewiner = self.method == "lewiner"
        verts = np.empty((nv.value, 3), np.float32)
        faces = np.empty((nf.value, 3), np.int32)
        want_vol = volume or (lewiner and nv.value == 0)
        vol = np.empty((self.voxels_dim,) * 3, np.float32) if want_vol else None
        _lib.check(_lib.lib().qsp_mesh_fetch(self.handle, _lib.fptr(verts), _lib.i32ptr(faces),
                                             _lib.fptr(vol) if want_vol else None))
        if lewiner:
            if nv.value == 0:      # skimage/measure/_marching_cubes_lewiner.py: the two ways an empty surface is reported
                if 0.0 < float(vol.min()) or 0.0 > float(vol.max()):
                    raise ValueError("Surface level must be within volume data range.")
                raise RuntimeError("No surface found at the given iso value.")
            verts = np.empty((nv.value, 3), np.float64)
            _lib.check(_lib.lib().qsp_mesh_fetch_f64(self.handle, verts.ctypes.data_as(C.POINTER(C.c_double))))
        return verts, faces, vol if volume else None

    def extract_sdf_grid(self, code):
        """(dim,dim,dim) SDF volume the reference hands to convert_sdf_voxels_to_mesh (optimizer.py:296-297)."""
        sdf = self.decoder.decode_sdf(np.asarray(code, np.float32)[: self.code_len], self.voxel_points)
        return sdf.reshape(self.voxels_dim, self.voxels_dim, self.voxels_dim)

    def mesh_from_volume(self, sdf_volume):
        """convert_sdf_voxels_to_mesh (reconstruct/utils.py:120-141) on a given (dim,dim,dim) volume."""
        vol = _lib.f32c(np.asarray(sdf_volume, np.float32).reshape(-1))
        if vol.size != self.voxels_dim ** 3:
            raise ValueError("volume must be (%d,)*3" % self.voxels_dim)
        nv, nf = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().qsp_mesh_from_volume(self.handle, _lib.fptr(vol), C.byref(nv), C.byref(nf)))
        verts, faces, _ = self._fetch(nv, nf)
        return verts, faces

    def extract_mesh_from_code(self, code, return_volume=False):
        start = time.time()
        code = _lib.f32c(np.asarray(code, np.float32)[: self.code_len])
        if code.size < 64:
            code = np.concatenate([code, np.zeros(64 - code.size, np.float32)])
        nv, nf = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().qsp_mesh_extract(self.handle, _lib.fptr(code), C.byref(nv), C.byref(nf)))
        verts, faces, vol = self._fetch(nv, nf, return_volume)
        print("Extract mesh takes %f seconds" % (time.time() - start))
        out = ForceKeyErrorDict(vertices=verts, faces=faces)
        if return_volume:
            out["sdf_volume"] = vol
        return out
