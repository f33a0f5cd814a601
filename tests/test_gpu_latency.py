"""The reference's call pattern is ONE object per call (src/LocalMapping_util.cc:705-760 -> Optimizer.reconstruct_object).  The
library's one-shot entry point (qsp_reconstruct_objects) keeps one batch resident with the decoder, sized by the high-water mark
of the calls so far, and refills it instead of allocating ~25 device buffers per call.  What a call returns must not depend on
what the resident batch held before or on its capacities: every call here is compared, bit for bit, with a fresh batch created
for exactly that call."""
import os

import numpy as np
import pytest

from oracle import sdf_oracle as so
from tests.test_gpu_sdf import make_cfg

pytestmark = pytest.mark.gpu


def fresh(dec, opt, o, T0, code=None):
    from qsp_slam_amd.reconstruct.optimizer import RefineBatch, _joint_cfg
    b = RefineBatch(dec, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0] * len(T0))
    b.set_state(T0, code)
    b.run(0)
    out = b.get()
    b.close()
    return out


@pytest.mark.parametrize("prec", ["f32", "fp16x2", "fp16x2_t32"])
def test_resident_batch_is_reused_and_results_do_not_depend_on_it(golden_dir, prec):
    import bench
    from qsp_slam_amd import DeepSdfDecoder, synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    dec = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    dec.set_precision(prec.split("_")[0])
    if prec.endswith("t32"):
        dec.set_tile_points(32)
    opt = Optimizer(dec, make_cfg(so.JointConfig(n_iter=2)))
    # sizes going up, down, to zero rays in the ball, and up beyond the capacity again
    shapes = [(300, 64, 32), (1200, 200, 100), (150, 20, 4), (1100, 256, 200), (2500, 256, 200), (40, 16, 0)]
    for i, (m, n_fg, n_bg) in enumerate(shapes):
        o = synth.make_object_views(700 + i, 1, m, n_fg=n_fg, n_bg=n_bg)[0]
        r = opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
        T, code, loss, good = fresh(dec, opt, o, o["t_cam_obj"][None])
        assert bool(r.is_good) == bool(good[0]) and np.float32(r.loss) == loss[0], (prec, i)
        if r.is_good:
            assert np.array_equal(r.t_cam_obj, T[0]) and np.array_equal(r.code, code[0]), (prec, i)
        # the four yaw flips of the same object as one call, through the same resident batch
        sel = opt.reconstruct_objects_batched([dict(t_cam_obj=o["t_cam_obj"], pts=o["pts"], rays=o["rays"], depth=o["depth"])],
                                              flip_sample_num=4, select=False)[0]
        T0, _ = bench.flip_states([o], 4)
        T4, code4, loss4, good4 = fresh(dec, opt, o, T0)
        for k in range(4):
            assert bool(sel[k].is_good) == bool(good4[k])
            if sel[k].is_good:
                assert np.array_equal(sel[k].t_cam_obj, T4[k]) and np.array_equal(sel[k].code, code4[k]), (prec, i, k)
    reused, created = dec.arena_stats
    assert created <= 4 and reused >= 8, (reused, created)       # a high-water mark: it settles
    dec.close()


def test_an_initial_code_and_a_failure_through_the_resident_batch(golden_dir):
    """the code argument and the reference's failure exit (fewer than 10 ray samples in the unit ball -> is_good False, no
    exception) survive the reuse; a good call after a failed one is unaffected by it"""
    from qsp_slam_amd import DeepSdfDecoder, synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    dec = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    opt = Optimizer(dec, make_cfg(so.JointConfig(n_iter=2)))
    o = synth.make_object_views(11, 1, 400, n_fg=100, n_bg=50)[0]
    code0 = (0.05 * np.random.default_rng(3).normal(size=64)).astype(np.float32)
    good1 = opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"], code=code0)
    bad = dict(o)
    bad["rays"] = o["rays"].copy()
    bad["rays"][:, 0] += 3.0                      # every ray misses the unit ball
    rb = opt.reconstruct_object(bad["t_cam_obj"], bad["pts"], bad["rays"], bad["depth"])
    assert not rb.is_good and rb.t_cam_obj is None
    good2 = opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"], code=code0)
    assert good1.is_good and good2.is_good
    assert np.array_equal(good1.t_cam_obj, good2.t_cam_obj) and np.array_equal(good1.code, good2.code) and good1.loss == good2.loss
    T, code, loss, good = fresh(dec, opt, o, o["t_cam_obj"][None], code0[None])
    assert np.array_equal(good1.t_cam_obj, T[0]) and np.array_equal(good1.code, code[0])
    dec.close()


@pytest.mark.timeout(300)
def test_two_host_threads_sharing_one_decoder_get_the_serial_results(golden_dir):
    """qsp_reconstruct_objects refills a batch that is resident with the DECODER and rewrites decoder fields for the
    duration of a call.  Entry points that launch on a decoder serialise on its lock (include/qsp_hip.h, "Threads"): two threads
    hammering the same decoder with objects of different sizes -- so that the resident batch is refilled, and regrown, under
    them -- and with decode calls in between get, call for call, the bits a single thread gets."""
    import threading
    from qsp_slam_amd import DeepSdfDecoder, synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    dec = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    dec.set_precision("fp16x2")
    opt = Optimizer(dec, make_cfg(so.JointConfig(n_iter=2)))
    objs = [synth.make_object_views(900 + i, 1, m, n_fg=nf, n_bg=nb)[0]
            for i, (m, nf, nb) in enumerate([(300, 64, 32), (1500, 200, 100), (700, 128, 64), (2200, 256, 200)])]
    x = np.random.default_rng(1).uniform(-0.8, 0.8, size=(500, 3)).astype(np.float32)
    want = [opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"]) for o in objs]
    want_sdf = dec.decode_sdf(np.zeros(64, np.float32), x)
    errors = []

    def work(order):
        try:
            for rep in range(6):
                for i in order:
                    o = objs[i]
                    r = opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
                    assert r.is_good == want[i].is_good and np.array_equal(r.t_cam_obj, want[i].t_cam_obj)
                    assert np.array_equal(r.code, want[i].code) and r.loss == want[i].loss
                    assert np.array_equal(dec.decode_sdf(np.zeros(64, np.float32), x), want_sdf)
        except Exception as e:      # noqa: BLE001 (reported below)
            errors.append(e)

    th = [threading.Thread(target=work, args=(order,)) for order in ([0, 1, 2, 3], [3, 2, 1, 0])]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dec.close()
    assert not errors, errors


def test_state_lives_on_the_device_between_calls(golden_dir):
    """The batch's inputs are written to a pinned mirror and uploaded by the next consumer (csrc/sdf_refine.hip:batch_upload).  What
    must hold whatever the order of calls: a state that was set and never run reads back as set; a run continues from the state the
    previous run left on the DEVICE (the mirror still holds the older one and must not be uploaded again); setting a state between
    two runs replaces it."""
    from qsp_slam_amd import DeepSdfDecoder, synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    dec = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    opt = Optimizer(dec, make_cfg(so.JointConfig(n_iter=5)))
    o = synth.make_object_views(4242, 1, 500, n_fg=128, n_bg=64)[0]
    T0 = o["t_cam_obj"][None].astype(np.float32)
    code0 = (0.02 * np.random.default_rng(1).standard_normal((1, 64))).astype(np.float32)

    def batch():
        return RefineBatch(dec, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0])
    b = batch()
    b.set_state(T0, code0)
    T, code, _, good = b.get()                                  # (no run in between: the upload happens for the read-back)
    assert np.array_equal(code[0], code0[0]) and bool(good[0]) and np.abs(T[0] - T0[0]).max() < 1e-5
    b.run(5)
    five = b.get()
    b.close()
    b = batch()
    b.set_state(T0, code0)
    b.run(2)
    two = b.get()
    b.run(3)                                                    # continues from the device's state, not from the mirror's
    cont = b.get()
    assert not np.array_equal(two[1], five[1])
    for x, y in zip(cont, five):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
    b.set_state(T0, code0)                                      # ... and a new state replaces it
    b.run(5)
    again = b.get()
    b.close()
    for x, y in zip(again, five):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
    dec.close()
