"""ctypes binding of libqsp_hip.so -- the C-ABI declared in include/qsp_hip.h.

There is NO CPU fallback: if the library is missing or no MI355X is visible the product path raises.  The oracle under
oracle/ is test infrastructure and is never imported from here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QSP_HIP_LIB", os.path.join(_HERE, "libqsp_hip.so"))   # override: experiments only

QSP_OK, QSP_ERR_INVALID, QSP_ERR_UNSUPPORTED, QSP_ERR_DEVICE, QSP_ERR_NO_DEVICE = 0, 1, 2, 3, 4

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class QspError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "qsp_hip error %d: %s" % (code, msg))
        self.code = code


class DecoderDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("code_len", C.c_int32), ("latent_in_layer", C.c_int32),
                ("in_dim", c_int32_p), ("out_dim", c_int32_p),
                ("weight", C.POINTER(c_float_p)), ("weight_g", C.POINTER(c_float_p)), ("bias", C.POINTER(c_float_p))]


class JointCfg(C.Structure):
    _fields_ = [("k1", C.c_float), ("k2", C.c_float), ("k3", C.c_float), ("k4", C.c_float),
                ("b1", C.c_float), ("b2", C.c_float), ("lr", C.c_float), ("s_damp", C.c_float),
                ("cut_off", C.c_float), ("n_iter", C.c_int32), ("n_depth", C.c_int32), ("code_len", C.c_int32)]


class RefineProfile(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_mlp_jtj", C.c_float), ("ms_mlp_fwd", C.c_float),
                ("ms_other", C.c_float), ("n_launch_jtj", C.c_int32), ("n_launch_fwd", C.c_int32),
                ("pts_jtj", C.c_int64), ("pts_fwd", C.c_int64), ("tiles_jtj", C.c_int64), ("tiles_fwd", C.c_int64),
                ("pts_band", C.c_int64), ("range_fallbacks", C.c_int32), ("screen_fallbacks", C.c_int32),
                ("screen_max_diff", C.c_float), ("screen_audit_failures", C.c_int32), ("pts_audit", C.c_int64)]


class BaScene(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_pt", C.c_int32), ("n_obj", C.c_int32), ("n_mono", C.c_int32),
                ("n_stereo", C.c_int32), ("n_objedge", C.c_int32),
                ("kf_pose", c_double_p), ("kf_fixed", c_uint8_p), ("kf_id", c_int64_p), ("kf_K", c_double_p),
                ("pt_xyz", c_double_p), ("pt_id", c_int64_p), ("obj_pose", c_double_p), ("obj_id", c_int64_p),
                ("mono_pt", c_int32_p), ("mono_kf", c_int32_p), ("mono_obs", c_double_p), ("mono_info", c_double_p),
                ("stereo_pt", c_int32_p), ("stereo_kf", c_int32_p), ("stereo_obs", c_double_p),
                ("stereo_info", c_double_p), ("objedge_kf", c_int32_p), ("objedge_obj", c_int32_p),
                ("objedge_meas", c_double_p), ("objedge_info", C.c_double)]


class BaTrace(C.Structure):
    _fields_ = [("cap", C.c_int32), ("n", C.c_int32), ("chi2", c_double_p), ("lam", c_double_p),
                ("trials", c_int32_p), ("accepted", c_int32_p), ("result", C.c_int32), ("iterations", C.c_int32),
                ("n_pose_blocks", C.c_int32), ("n_landmarks", C.c_int32)]


class BaProfile(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_linearize", C.c_float), ("ms_schur", C.c_float),
                ("ms_solve", C.c_float), ("ms_update", C.c_float), ("n_linearize", C.c_int32),
                ("n_trials", C.c_int32), ("bytes_linearize", C.c_int64), ("cholesky_chain", C.c_int32), ("chain_timeouts", C.c_int32),
                ("boundary_device", C.c_int32), ("boundary_host", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)   # qsp_allreduce_fn

_lib = None

# every symbol include/qsp_hip.h declares; tests/test_abi.py checks the list against the header
class PoseTrace(C.Structure):
    _fields_ = [("iters", C.c_int32 * 4), ("trace", C.c_double * 120)]


class Detections(C.Structure):
    _fields_ = [("n_det", C.c_int32), ("T_cw", c_float_p), ("K", c_float_p), ("T_wo", c_float_p), ("code", c_float_p),
                ("n_flip", c_int32_p), ("flip_angle", C.c_double), ("pts_off", c_int32_p), ("pts_world", c_float_p),
                ("fg_off", c_int32_p), ("fg_px", c_float_p), ("fg_world", c_float_p), ("bg_off", c_int32_p),
                ("bg_rays", c_float_p)]


class DetectionResults(C.Structure):
    _fields_ = [("t_cam_obj", c_float_p), ("code", c_float_p), ("loss", c_float_p), ("is_good", c_uint8_p),
                ("kept_flip", c_int32_p), ("losses", c_float_p), ("pts_cam", c_float_p), ("rays", c_float_p),
                ("depth_obs", c_float_p), ("t_cam_obj_init", c_float_p)]


SYMBOLS = [
    "qsp_last_error", "qsp_version", "qsp_device_count",
    "qsp_decoder_create", "qsp_decoder_destroy", "qsp_decoder_set_option", "qsp_decoder_get_counter", "qsp_decode_sdf",
    "qsp_decode_sdf_screen", "qsp_sdf_value_grad",
    "qsp_refine_batch_create", "qsp_refine_batch_destroy", "qsp_refine_batch_set_state", "qsp_refine_batch_run",
    "qsp_refine_batch_get", "qsp_refine_batch_trace", "qsp_refine_batch_trace_rot", "qsp_refine_batch_profile", "qsp_refine_batch_rows",
    "qsp_reconstruct_objects", "qsp_estimate_pose", "qsp_refine_detections",
    "qsp_mesh_extractor_create", "qsp_mesh_extractor_destroy", "qsp_mesh_extract", "qsp_mesh_from_volume", "qsp_mesh_fetch", "qsp_mesh_fetch_f64", "qsp_mesh_extractor_set_method",
    "qsp_mc_tables",
    "qsp_pose_optimizer_create", "qsp_pose_optimizer_destroy", "qsp_pose_optimize", "qsp_ellipsoid_fit_planes", "qsp_ellipsoid_fit_prior",
    "qsp_ba_create", "qsp_ba_destroy", "qsp_ba_set_levels", "qsp_ba_optimize", "qsp_ba_local_joint",
    "qsp_ba_set_state", "qsp_ba_get_state", "qsp_ba_get_edges", "qsp_ba_get_index", "qsp_ba_profile", "qsp_ba_set_shard", "qsp_ba_set_deterministic",
    "qsp_ba_set_shard_rccl", "qsp_ba_set_option", "qsp_ba_release_caches", "qsp_comm_unique_id", "qsp_comm_create", "qsp_comm_adopt", "qsp_comm_destroy", "qsp_comm_nccl", "qsp_comm_stub_counts",
    "qsp_comm_rank", "qsp_comm_world", "qsp_comm_allreduce_f64", "qsp_comm_allgather_f32",
]


def lib():
    """Loads the shared library once.  Raises if it was not built (python __graft_entry__.py / csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise QspError(QSP_ERR_NO_DEVICE, "libqsp_hip.so not built (%s); run qsp_slam_amd/csrc/build.sh -- there is "
                       "no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.qsp_last_error.restype = C.c_char_p
    L.qsp_version.restype = C.c_int
    L.qsp_device_count.restype = C.c_int
    vp = C.c_void_p
    L.qsp_decoder_create.argtypes = [C.POINTER(DecoderDesc), C.c_int, C.POINTER(vp)]
    L.qsp_decoder_destroy.argtypes = [vp]
    L.qsp_decoder_set_option.argtypes = [vp, C.c_int32, C.c_int32]
    L.qsp_decoder_get_counter.argtypes = [vp, C.c_int32]
    L.qsp_decoder_get_counter.restype = C.c_int64
    L.qsp_decoder_destroy.restype = None
    L.qsp_decode_sdf.argtypes = [vp, c_float_p, c_float_p, C.c_int64, c_float_p]
    L.qsp_decode_sdf_screen.argtypes = [vp, c_float_p, c_float_p, C.c_int64, c_float_p]
    L.qsp_sdf_value_grad.argtypes = [vp, c_float_p, c_float_p, C.c_int64, c_float_p, c_float_p]
    pp_f = C.POINTER(c_float_p)
    L.qsp_refine_batch_create.argtypes = [vp, C.POINTER(JointCfg), C.c_int32, pp_f, c_int32_p, pp_f, c_int32_p, pp_f,
                                          c_int32_p, C.c_int32, c_int32_p, C.POINTER(vp)]
    L.qsp_refine_batch_destroy.argtypes = [vp]
    L.qsp_refine_batch_destroy.restype = None
    L.qsp_refine_batch_set_state.argtypes = [vp, c_float_p, c_float_p]
    L.qsp_refine_batch_run.argtypes = [vp, C.c_int32]
    L.qsp_refine_batch_get.argtypes = [vp, c_float_p, c_float_p, c_float_p, c_uint8_p]
    L.qsp_refine_batch_trace.argtypes = [vp, c_float_p, c_float_p, c_float_p, c_int32_p, c_int32_p, c_float_p]
    L.qsp_refine_batch_trace_rot.argtypes = [vp, c_float_p]
    L.qsp_refine_batch_profile.argtypes = [vp, C.c_int, C.POINTER(RefineProfile)]
    L.qsp_refine_batch_rows.argtypes = [vp, C.c_int, C.c_int32, c_float_p, c_float_p]
    L.qsp_reconstruct_objects.argtypes = [vp, C.POINTER(JointCfg), C.c_int32, pp_f, c_int32_p, pp_f, c_int32_p, pp_f,
                                          c_int32_p, C.c_int32, c_int32_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                          c_float_p, c_uint8_p]
    L.qsp_estimate_pose.argtypes = [vp, C.c_int32, c_float_p, c_float_p, pp_f, c_int32_p, c_float_p, C.c_int32,
                                    c_float_p]
    L.qsp_refine_detections.argtypes = [vp, C.POINTER(JointCfg), C.POINTER(Detections), C.POINTER(DetectionResults)]
    L.qsp_mesh_extractor_create.argtypes = [vp, C.c_int32, c_float_p, C.POINTER(vp)]
    L.qsp_mesh_extractor_destroy.argtypes = [vp]
    L.qsp_mesh_extractor_destroy.restype = None
    L.qsp_mesh_extract.argtypes = [vp, c_float_p, c_int64_p, c_int64_p]
    L.qsp_mesh_from_volume.argtypes = [vp, c_float_p, c_int64_p, c_int64_p]
    L.qsp_mesh_fetch.argtypes = [vp, c_float_p, c_int32_p, c_float_p]
    L.qsp_mesh_fetch_f64.argtypes = [vp, C.POINTER(C.c_double)]
    L.qsp_mesh_extractor_set_method.argtypes = [vp, C.c_int32]
    L.qsp_mc_tables.argtypes = [C.POINTER(C.c_int8), C.POINTER(C.c_int8)]
    L.qsp_pose_optimizer_create.argtypes = [C.c_int, C.c_int32, C.POINTER(vp)]
    L.qsp_pose_optimizer_destroy.argtypes = [vp]
    L.qsp_pose_optimizer_destroy.restype = None
    L.qsp_pose_optimize.argtypes = [vp, C.c_int32, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_uint8_p,
                                    c_double_p, c_uint8_p, c_int32_p, C.POINTER(PoseTrace)]
    L.qsp_ellipsoid_fit_planes.argtypes = [C.c_int, C.c_int32, c_double_p, c_int32_p, c_double_p, C.c_int32, C.c_int32,
                                           c_double_p, c_double_p, c_int32_p, c_double_p]
    L.qsp_ellipsoid_fit_prior.argtypes = [C.c_int, C.c_int32, c_double_p, c_int32_p, c_double_p, c_int32_p, c_double_p, c_double_p,
                                          c_double_p, c_double_p, C.c_double, C.c_int32, c_double_p, c_double_p, c_int32_p, c_double_p]
    L.qsp_ba_create.argtypes = [C.POINTER(BaScene), C.c_int, C.POINTER(vp)]
    L.qsp_ba_destroy.argtypes = [vp]
    L.qsp_ba_destroy.restype = None
    L.qsp_ba_set_levels.argtypes = [vp, c_uint8_p, c_uint8_p, c_uint8_p]
    L.qsp_ba_optimize.argtypes = [vp, C.c_int32, C.c_double, C.c_double, C.c_double, c_uint8_p, C.POINTER(BaTrace)]
    L.qsp_ba_local_joint.argtypes = [vp, c_uint8_p, C.POINTER(BaTrace), C.POINTER(BaTrace)]
    L.qsp_ba_set_state.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    L.qsp_ba_get_state.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    L.qsp_ba_get_edges.argtypes = [vp, c_double_p, c_double_p, c_double_p, c_uint8_p, c_uint8_p]
    L.qsp_ba_get_index.argtypes = [vp, c_int32_p, c_int32_p, c_int32_p]
    L.qsp_ba_profile.argtypes = [vp, C.c_int, C.POINTER(BaProfile)]
    L.qsp_ba_set_shard.argtypes = [vp, C.c_int32, C.c_int32, ALLREDUCE_FN, C.c_void_p]
    L.qsp_ba_set_deterministic.argtypes = [vp, C.c_int]
    L.qsp_ba_set_shard_rccl.argtypes = [vp, C.c_int32, C.c_int32, vp]
    L.qsp_ba_set_option.argtypes = [vp, C.c_int32, C.c_int32]
    L.qsp_comm_unique_id.argtypes = [c_uint8_p]
    L.qsp_comm_create.argtypes = [c_uint8_p, C.c_int32, C.c_int32, C.c_int, C.POINTER(vp)]
    L.qsp_comm_adopt.argtypes = [vp, C.c_int32, C.c_int32, C.c_int, C.POINTER(vp)]
    L.qsp_comm_destroy.argtypes = [vp]
    L.qsp_comm_destroy.restype = None
    L.qsp_comm_stub_counts.argtypes = [vp, c_int64_p]
    L.qsp_comm_nccl.argtypes = [vp]
    L.qsp_comm_nccl.restype = vp
    L.qsp_comm_rank.argtypes = [vp]
    L.qsp_comm_rank.restype = C.c_int32
    L.qsp_comm_world.argtypes = [vp]
    L.qsp_comm_world.restype = C.c_int32
    L.qsp_comm_allreduce_f64.argtypes = [vp, vp, C.c_int64, vp]
    L.qsp_comm_allgather_f32.argtypes = [vp, vp, vp, C.c_int64, vp]
    _lib = L
    return L


def check(rc):
    if rc != QSP_OK:
        raise QspError(rc, lib().qsp_last_error().decode("utf-8", "replace"))


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def i32ptr(a):
    return a.ctypes.data_as(c_int32_p)


def i64ptr(a):
    return a.ctypes.data_as(c_int64_p)


def u8ptr(a):
    return a.ctypes.data_as(c_uint8_p)


def f32c(a):
    """C-contiguous float32 copy/view.  pybind11 hands Eigen::MatrixXf to Python as Fortran-ordered (M,3) arrays
    (src/LocalMapping_util.cc:705-706), so non-contiguous inputs are the normal case."""
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr_array(arrs):
    """C array of float* over a list of contiguous float32 arrays (kept alive by the caller)."""
    t = c_float_p * len(arrs)
    return t(*[fptr(a) for a in arrs])
