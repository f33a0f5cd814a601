"""GPU parity of qsp_refine_detections (the caller-side marshalling of path A on the device,
reference src/LocalMapping_util.cc:585-760) through the C-ABI.

  * the assembled inputs (camera-frame surface points, rays, observed depths, flip initial poses) equal the restatement in
    oracle/detections_oracle.py BIT FOR BIT.  Tolerance against a build of the reference itself: its OpenCV / Eigen may
    contract a*b+c into an FMA (-march=native), which moves a ray or pose entry by at most 1 ulp (6e-8 relative); the refined
    poses are insensitive to that at the 1e-4 level north_star asks for;
  * the refinement that follows is the SAME arithmetic as the host-fed batch: every hypothesis equals, to the bit, the result of
    Optimizer.reconstruct_object on the oracle-assembled inputs (that entry point is pinned against the reference's golden
    vectors in tests/test_gpu_sdf.py);
  * the kept result is the one the reference's keep rule selects (oracle keep_rule), including not-good hypotheses, detections
    that already have a good orientation (one hypothesis), ragged and empty inputs."""
import math
import os

import numpy as np
import pytest

import bench
from oracle import detections_oracle as DO
from qsp_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_decoder(golden_dir):
    from qsp_slam_amd import DeepSdfDecoder
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    yield d
    d.close()


def _optimizer(gpu_decoder, n_iter=3):
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    return Optimizer(gpu_decoder, bench.joint_cfg(n_iter))


def _check_against_oracle(opt, dets, flips, res):
    ang = 2.0 * math.pi / flips
    for d, r in zip(dets, res):
        nf = 1 if d.get("found_good_orientation") else flips
        pts, rays, depth = DO.assemble(d)
        T0 = DO.init_poses(d, nf, ang)
        assert np.array_equal(r.pts, pts) and np.array_equal(r.rays, rays) and np.array_equal(r.depth, depth)
        assert np.array_equal(r.t_cam_obj_init, T0)
        singles = [opt.reconstruct_object(T0[k], pts, rays, depth, d.get("code")) for k in range(nf)]
        good = [s.is_good for s in singles]
        loss = [np.float32(s.loss) for s in singles]
        assert np.array_equal(np.asarray(r.losses, np.float32), np.asarray(loss, np.float32), equal_nan=True)
        k = DO.keep_rule(good, loss)
        assert r.kept_flip == k and r.is_good == good[k]
        if good[k]:
            assert np.array_equal(r.t_cam_obj, singles[k].t_cam_obj) and np.array_equal(r.code, singles[k].code)
            assert np.float32(r.loss) == loss[k]
        else:
            assert r.t_cam_obj is None and r.code is None


def test_detections_match_oracle_assembly_and_serial_calls(gpu_decoder):
    opt = _optimizer(gpu_decoder)
    dets = synth.make_detections(11, 5, 700, n_fg=96, n_bg=40, n_kf=2)
    dets[3]["found_good_orientation"] = True
    dets[4]["code"] = (0.05 * np.random.default_rng(0).normal(size=64)).astype(np.float32)
    res = opt.refine_detections(dets, flip_sample_num=4, taps=True)
    _check_against_oracle(opt, dets, 4, res)
    # the un-flipped start is the perturbed ground truth: it is kept unless a flip is better by the rule, and it moved closer
    assert all(r.is_good for r in res)


def test_detections_ragged_empty_and_failing(gpu_decoder):
    """ragged counts, a detection whose rays all miss the object (fewer than 10 samples in the unit ball -> every
    hypothesis not good -> the LAST one is kept by the rule and t_cam_obj is None), six flips"""
    opt = _optimizer(gpu_decoder, n_iter=2)
    dets = synth.make_detections(12, 3, 300, n_fg=48, n_bg=16)
    dets[0]["pts_world"] = dets[0]["pts_world"][:37]
    dets[1]["bg_rays"] = dets[1]["bg_rays"][:0]
    far = dets[2]
    far["fg_px"] = far["fg_px"] + np.float32(4000.0)            # rays that never enter the unit ball
    far["bg_rays"] = far["bg_rays"] + np.float32(9.0)
    res = opt.refine_detections(dets, flip_sample_num=6, taps=True)
    _check_against_oracle(opt, dets, 6, res)
    assert not res[2].is_good and res[2].kept_flip == 5 and res[2].t_cam_obj is None
    assert res[0].is_good and res[1].is_good


def test_detections_argument_errors(gpu_decoder):
    from qsp_slam_amd import _lib
    opt = _optimizer(gpu_decoder)
    d = synth.make_detections(13, 1, 50, n_fg=16, n_bg=4)[0]
    bad = dict(d)
    bad["fg_world"] = d["fg_world"][:-1]
    with pytest.raises(ValueError):
        opt.refine_detections([bad])
    with pytest.raises(_lib.QspError):
        opt.refine_detections([d], flip_sample_num=65)


def test_detections_c4_sized_call_equals_host_fed_batch(gpu_decoder):
    """64 detections x 4 flips x 8 k points in one call (the C4 batch) == RefineBatch fed with the oracle-assembled arrays"""
    from qsp_slam_amd.reconstruct.optimizer import RefineBatch, _joint_cfg
    w = bench.WORKLOADS["c4"]
    opt = _optimizer(gpu_decoder, n_iter=2)
    dets = synth.make_detections(14, 64, w["n_pts"], n_fg=w["n_fg"], n_bg=w["n_bg"], n_kf=5)
    res = opt.refine_detections(dets, flip_sample_num=4)
    asm = [DO.assemble(d) for d in dets]
    T0 = np.concatenate([DO.init_poses(d, 4, 2.0 * math.pi / 4) for d in dets])
    hyp = np.repeat(np.arange(64), 4)
    batch = RefineBatch(gpu_decoder, _joint_cfg(opt), [a[0] for a in asm], [a[1] for a in asm], [a[2] for a in asm], hyp)
    batch.set_state(T0, None)
    batch.run(0)
    T, code, loss, good = batch.get()
    batch.close()
    for i, r in enumerate(res):
        k = DO.keep_rule(list(good[4 * i:4 * i + 4]), list(loss[4 * i:4 * i + 4]))
        assert r.kept_flip == k and np.array_equal(r.losses, loss[4 * i:4 * i + 4])
        assert np.array_equal(r.t_cam_obj, T[4 * i + k]) and np.array_equal(r.code, code[4 * i + k])
