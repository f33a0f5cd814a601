"""The split-bf16 decoder pipe (QSP_DEC_OPT_FORWARD_PRECISION / QSP_DEC_OPT_JACOBIAN_PRECISION = 1, csrc/sdf_mlp.hpp: three bf16
terms per f32 operand, six products per multiply-add on the bf16 matrix pipe, f32 accumulation) is held to the SAME gates as
the exact-f32 tile before bench.py may quote it:
  * decoder value / input gradient against the reference-generated vectors at the f32 tile's tolerances;
  * every Gauss-Newton iteration of every golden case, teacher-forced from the reference's own state: K (a discrete count of
    threshold decisions on decoder outputs) exact, H, b, next state within north_star's 1e-4, dx within the f32 tile's bar;
  * K and n_valid exact against the numpy oracle on a randomised sweep of sizes / seeds / codes (discrete decisions);
  * bit-reproducible, and batch independent (a hypothesis in a batch == the same hypothesis alone)."""
import os

import numpy as np
import pytest

from oracle import sdf_oracle as so
from tests.margins import within
from tests.test_gpu_sdf import make_cfg
from tests.test_oracle_sdf import JOINT_CASES, cfg_from, relerr, rows_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bf3_decoder(golden_dir):
    from qsp_slam_amd import DeepSdfDecoder
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    d.set_precision("bf16x3")
    yield d
    d.close()


def test_decoder_value_and_grad_vs_reference_vectors(bf3_decoder, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_decoder_vectors.npz"))
    assert within("bf16x3/decoder/sdf_abs", np.abs(bf3_decoder.decode_sdf(z["code"], z["x"]) - z["sdf"]).max(), 2e-6)
    y, g = bf3_decoder.sdf_value_grad(z["code"], z["x"])
    assert within("bf16x3/decoder/y_abs", np.abs(y - z["y"]).max(), 2e-6)
    assert rows_close(g, z["grad"], tol=1e-5, max_bad=0.01)


@pytest.mark.parametrize("name", JOINT_CASES)
def test_every_iteration_teacher_forced_vs_reference(bf3_decoder, golden_dir, name):
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    kitti = cfg_from(z).k4 != 0
    opt = Optimizer(bf3_decoder, make_cfg(z))
    batch = RefineBatch(bf3_decoder, _joint_cfg(opt), [z["pts"]], [z["rays"]], [z["depth"]], [0])
    n_it = z["it_H"].shape[0]
    for i in range(n_it):
        T_co = np.linalg.inv(z["it_T_oc"][i].astype(np.float64)).astype(np.float32)
        batch.set_state(T_co[None], z["it_code"][i][None])
        batch.run(1)
        tr = batch.trace()
        T, code, loss, good = batch.get()
        assert good[0] and int(tr["K"][0]) == int(z["it_K"][i])
        tag = "bf16x3/" + name + "/teacher_forced/"
        assert within(tag + "H", relerr(tr["H"][0], z["it_H"][i]), 1e-4)
        assert within(tag + "b", relerr(tr["b"][0], z["it_b"][i]), 1.5e-2 if kitti else 1e-4)     # (k4 = 1e7: test_gpu_sdf.py)
        assert within(tag + "dx", relerr(tr["dx"][0], z["it_dx"][i]), 6e-3 if kitti else 2.5e-3)
        if i + 1 < n_it:
            assert within(tag + "T_oc_next", relerr(np.linalg.inv(T[0].astype(np.float64)), z["it_T_oc"][i + 1]),
                          2.5e-4 if kitti else 1e-4)
            assert within(tag + "code_next_abs", np.abs(code[0] - z["it_code"][i + 1]).max(), 1e-4)
    batch.close()


def test_discrete_decisions_match_the_oracle_on_a_random_sweep(bf3_decoder, oracle_decoder):
    """n_valid (samples in the unit ball) and K (kept render rows: |sdf| < cut-off, de/do > 1e-2) are counts of threshold
    decisions on decoder outputs: exact in 40 random cases, and H, b within 1e-4 except where single ReLU knife-edge rows
    explain it (tools/parity_sweep.py has the row-wise accounting)"""
    from qsp_slam_amd import synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    cfg = so.JointConfig()
    opt = Optimizer(bf3_decoder, make_cfg(cfg))
    rng = np.random.default_rng(321)
    over = 0
    for c in range(40):
        m, n_fg, n_bg = int(rng.integers(1, 900)), int(rng.integers(12, 160)), int(rng.integers(0, 60))
        o = synth.make_object_views(int(rng.integers(1, 10 ** 6)), 1, m, n_fg=n_fg, n_bg=n_bg, code_scale=float(rng.choice([0.0, 0.05])))[0]
        code = (0.05 * rng.normal(size=64)).astype(np.float32) if c % 3 == 0 else np.zeros(64, np.float32)
        batch = RefineBatch(bf3_decoder, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0])
        batch.set_state(o["t_cam_obj"][None], code[None])
        batch.run(1)
        tr = batch.trace()
        good = bool(batch.get()[3][0])
        batch.close()
        T_oc = np.linalg.inv(o["t_cam_obj"].astype(np.float64)).astype(np.float32)
        it = so.gn_iteration(oracle_decoder, cfg, T_oc, code, o["pts"], o["rays"], np.concatenate([o["depth"], np.zeros(n_bg, np.float32)]), n_fg)
        if it["fail"] is not None:
            assert not good
            continue
        assert good and int(tr["n_valid"][0]) == it["n_valid"] and int(tr["K"][0]) == it["K"], c
        eH, eb = relerr(tr["H"][0], it["H"]), relerr(tr["b"][0], it["b"])
        over += not (eH < 1e-4 and eb < 1e-4)
    assert over <= 2           # the f32 tile: 2 of 120 (each explained by one knife-edge row)


def test_bit_reproducible_and_batch_independent(bf3_decoder):
    from qsp_slam_amd import synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    objs = synth.make_object_views(606, 6, 700, n_fg=120, n_bg=60)
    opt = Optimizer(bf3_decoder, make_cfg(so.JointConfig(n_iter=3)))
    import bench
    T0, hyp = bench.flip_states(objs, 4)
    batch = RefineBatch(bf3_decoder, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp)
    outs = []
    for _ in range(2):
        batch.set_state(T0, None)
        batch.run(0)
        outs.append(batch.get())
    batch.close()
    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
    T, code, loss, good = outs[0]
    for h in (0, 7, 13, 23):
        o = objs[hyp[h]]
        single = RefineBatch(bf3_decoder, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0])
        single.set_state(T0[h:h + 1], None)
        single.run(0)
        Ts, cs, ls, gs = single.get()
        single.close()
        assert bool(gs[0]) == bool(good[h]) and np.array_equal(Ts[0], T[h]) and np.array_equal(cs[0], code[h]) and ls[0] == loss[h]


def test_mesh_grid_decode_agrees_with_the_f32_pipe(bf3_decoder, golden_dir):
    """the 64^3 voxel decode of MeshExtractor on either pipe: values within 5e-7, so the zero crossings (vertices) move by less
    than 1e-5 of a voxel; the sign pattern (topology) may differ only in voxels whose |sdf| is below that"""
    from qsp_slam_amd import DeepSdfDecoder
    from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
    f32 = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    code = np.zeros(64, np.float32)
    a = MeshExtractor(f32, 64, 32).extract_sdf_grid(code)
    b = MeshExtractor(bf3_decoder, 64, 32).extract_sdf_grid(code)
    assert within("bf16x3/grid/sdf_abs_vs_f32", np.abs(a - b).max(), 5e-7)
    flips = (a > 0) != (b > 0)
    assert np.abs(a[flips]).max(initial=0.0) < 5e-7
    f32.close()
