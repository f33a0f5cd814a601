"""The pybind11 host layer (qsp_slam_amd/csrc/reconstruct_hip.cpp): the reference's B1 entry points as C++ over the C-ABI.
CPU part: the module builds, imports without a GPU, mirrors the reference's constructor contract (reconstruct/optimizer.py:27-44:
attribute access on the configs object, KeyError for a missing key, `code_len` readable from C++).  GPU part: same bits as the
ctypes twin (qsp_slam_amd/reconstruct/optimizer.py) for reconstruct_object, estimate_pose_cam_obj and the mesh."""
import os

import numpy as np
import pytest

import bench


class _FakeDecoder(object):
    handle = None


def test_module_imports_without_gpu_and_reads_configs():
    from qsp_slam_amd import reconstruct_hip as rh
    from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict
    assert rh.version() == 1
    opt = rh.Optimizer(_FakeDecoder(), bench.joint_cfg(7))
    assert opt.code_len == 64
    bad = ForceKeyErrorDict(data_type="Redwood", optimizer=dict(code_len=64, num_depth_samples=50, cut_off_threshold=0.01,
                                                                joint_optim=dict(k1=1.0)))
    with pytest.raises(KeyError):
        rh.Optimizer(_FakeDecoder(), bad)
    kitti = bench.joint_cfg(5)
    kitti["data_type"] = "KITTI"
    with pytest.raises(KeyError):                     # pose_only_optim is required for KITTI (optimizer.py:43-44)
        rh.Optimizer(_FakeDecoder(), kitti)
    # a closed / missing decoder handle is an error at call time, not a crash
    with pytest.raises(RuntimeError):
        opt.reconstruct_object(np.eye(4, dtype=np.float32), np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32),
                               np.zeros(2, np.float32))


@pytest.mark.gpu
def test_pybind_layer_gives_the_same_bits_as_the_ctypes_twin(golden_dir):
    from qsp_slam_amd import DeepSdfDecoder, reconstruct_hip as rh, synth
    from qsp_slam_amd.reconstruct import optimizer as tw
    dec = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    cfg = bench.joint_cfg(3)
    a, b = rh.Optimizer(dec, cfg), tw.Optimizer(dec, cfg)
    o = synth.make_object_views(21, 1, 700, n_fg=96, n_bg=40)[0]
    # Fortran-ordered inputs, as pybind11 hands Eigen matrices over
    pts, rays = np.asfortranarray(o["pts"]), np.asfortranarray(o["rays"])
    for code in (None, (0.05 * np.random.default_rng(0).normal(size=64)).astype(np.float32)):
        ra = a.reconstruct_object(o["t_cam_obj"], pts, rays, o["depth"], code)
        rb = b.reconstruct_object(o["t_cam_obj"], pts, rays, o["depth"], code)
        assert ra.is_good and rb.is_good and ra.loss == rb.loss
        assert np.array_equal(ra.t_cam_obj, rb.t_cam_obj) and np.array_equal(ra.code, rb.code)
        assert ra.t_cam_obj.dtype == np.float32 and ra.t_cam_obj.shape == (4, 4) and ra.code.shape == (64,)
        with pytest.raises(KeyError):
            ra["no_such_key"]
    # failure contract: too few ray samples -> is_good False, t_cam_obj None, no exception
    far = o["rays"] + np.float32(50.0)
    rf = a.reconstruct_object(o["t_cam_obj"], pts, far, o["depth"])
    assert not rf.is_good and rf.t_cam_obj is None and rf.code is None
    T = o["gt_t_cam_obj"].copy()
    s = float(np.cbrt(np.linalg.det(T[:3, :3].astype(np.float64))))
    T[:3, :3] /= np.float32(s)
    pa = a.estimate_pose_cam_obj(T, s, pts, np.zeros(64, np.float32))
    pb = b.estimate_pose_cam_obj(T, s, pts, np.zeros(64, np.float32))
    assert np.array_equal(np.asarray(pa), np.asarray(pb)) and np.asarray(pa).shape == (4, 4)
    ma, mb = rh.MeshExtractor(dec, 64, 32), tw.MeshExtractor(dec, 64, 32)
    xa, xb = ma.extract_mesh_from_code(np.zeros(64, np.float32)), mb.extract_mesh_from_code(np.zeros(64, np.float32))
    assert np.array_equal(xa.vertices, xb.vertices) and np.array_equal(xa.faces, xb.faces) and xa.faces.dtype == np.int32
    # lifetime: extractors destroyed AFTER their decoder only free their own memory (include/qsp_hip.h) and leave no HIP
    # error behind for the next call's launch check
    import gc
    from qsp_slam_amd.ba import BaProblem
    dec.close()
    del ma, mb, xa, xb, a, b
    gc.collect()
    prob = BaProblem(synth.make_ba_scene(5, 6, 200, 2, stereo_frac=0.2))
    t1, t2 = prob.local_joint_ba()
    assert t1["iterations"] >= 1
    prob.close()
