"""CPU: the numpy restatement of hot path A (oracle/sdf_oracle.py) against vectors produced by RUNNING the reference
(oracle/gen_golden_sdf.py, committed under tests/golden/).  Tolerances: the restatement and the reference both compute in
float32 but sum in different orders; the north_star bar is 1e-4 relative on residuals and pose updates."""
import ast
import os

import numpy as np
import pytest

from oracle import sdf_oracle as so

JOINT_CASES = ["sdf_joint_redwood_m600", "sdf_joint_redwood_m2000", "sdf_joint_kitti_m250", "sdf_joint_code_m500"]


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def rows_close(a, b, tol=1e-5, max_bad=0.003):
    """Row-wise agreement of Jacobians.  d sdf / d input of a ReLU network is discontinuous where a pre-activation is
    exactly at zero; two f32 evaluations that sum in different orders put a handful of knife-edge units on different
    sides (measured: 1 row in 2000), which changes that ROW by up to ~1e-2 while every other row agrees to ~2e-7 and
    the normal matrix to ~3e-5.  So: all but `max_bad` of the rows within `tol` of the largest entry."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.shape != b.shape:
        return False
    d = np.abs(a - b).reshape(a.shape[0], -1).max(1) / max(np.abs(b).max(), 1e-30)
    return (d > tol).mean() <= max_bad


def cfg_from(z):
    j = ast.literal_eval(str(z["joint"]))
    return so.JointConfig(k1=j["k1"], k2=j["k2"], k3=j["k3"], k4=j["k4"], b1=j["b1"], b2=j["b2"], lr=j["learning_rate"],
                          s_damp=j["scale_damping"], n_iter=j["num_iterations"])


def test_decoder_value_and_grad(oracle_decoder, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_decoder_vectors.npz"))
    sdf = so.decode_sdf(oracle_decoder, z["code"], z["x"])
    assert np.abs(sdf - z["sdf"]).max() < 2e-6
    inp = np.concatenate([np.broadcast_to(z["code"], (z["x"].shape[0], 64)), z["x"]], -1)
    y, g = so.decoder_value_and_input_grad(oracle_decoder, inp)
    assert np.abs(y - z["y"]).max() < 2e-6
    assert rows_close(g, z["grad"], tol=1e-5, max_bad=0.01)


def test_lie_exponentials(golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_lie_vectors.npz"))
    for i, x in enumerate(z["x"]):
        assert np.abs(so.exp_sim3(x) - z["exp_sim3"][i]).max() < 2e-6
        assert np.abs(so.exp_se3(x[:6]) - z["exp_se3"][i]).max() < 2e-6


def test_voxel_grid_true_division_quirk(golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_voxel_grid.npz"))
    g = so.create_voxel_grid(int(z["dim"]))
    assert np.abs(g - z["grid"]).max() < 1e-6


@pytest.mark.parametrize("name", JOINT_CASES)
def test_first_iteration_terms(oracle_decoder, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    T_oc = np.linalg.inv(z["t_cam_obj"]).astype(np.float32)
    code = np.zeros(64, np.float32)
    Jp, Jc, res, _ = so.sdf_term(oracle_decoder, z["pts"], T_oc, code)
    assert relerr(res, z["it0_res_sdf"]) < 1e-4
    assert rows_close(Jp, z["it0_Jp_sdf"])
    assert rows_close(Jc, z["it0_Jc_sdf"])
    J = np.concatenate([Jp, Jc], 1).astype(np.float64)
    Jr = np.concatenate([z["it0_Jp_sdf"], z["it0_Jc_sdf"]], 1).astype(np.float64)
    assert relerr(J.T @ J, Jr.T @ Jr) < 1e-4
    T_co = np.linalg.inv(T_oc).astype(np.float32)
    scale = np.float32(np.linalg.det(T_co[:3, :3])) ** np.float32(1 / 3)
    depths = np.linspace(T_co[2, 3] - scale, T_co[2, 3] + scale, 50, dtype=np.float32)
    n_fg = z["depth"].shape[0]
    dobs = np.concatenate([z["depth"], np.full(z["rays"].shape[0] - n_fg, np.float32(1.1) * depths[-1], np.float32)])
    rt = so.render_term(oracle_decoder, z["rays"], dobs, T_oc, depths, code, th=0.01)
    assert rt["res"].shape == z["it0_res_render"].shape      # same K: identical row selection
    assert relerr(rt["res"], z["it0_res_render"]) < 1e-4
    assert rows_close(rt["J_pose"], z["it0_Jp_render"], tol=2e-5, max_bad=0.02)
    assert rows_close(rt["J_code"], z["it0_Jc_render"], tol=2e-5, max_bad=0.02)


@pytest.mark.parametrize("name", JOINT_CASES)
def test_every_iteration_teacher_forced(oracle_decoder, golden_dir, name):
    """The parity contract of path A.  Each Gauss-Newton iteration is restarted from the state the REFERENCE had at that
    iteration (it_T_oc / it_code, tapped while the reference ran): same render-row count K, normal matrix, right-hand
    side, update and next state within 1e-4 (relative to the largest entry)."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = cfg_from(z)
    n_fg = z["depth"].shape[0]
    dobs = np.concatenate([z["depth"], np.zeros(z["rays"].shape[0] - n_fg, np.float32)])
    n_it = z["it_H"].shape[0]
    assert n_it == cfg.n_iter
    # With the KITTI weights (k4 = 1e7) the rotation prior multiplies res_rot = 1 - cos(tilt), an f32 cancellation whose
    # own rounding noise is ~1e-2 relative (loss.py:155-178 computes it through two f32 matrix inverses); the prior's
    # share of H and b can only be reproduced to that level, by the reference itself on another BLAS as well.
    tol = 1e-4 if cfg.k4 == 0 else 5e-3
    for i in range(n_it):
        it = so.gn_iteration(oracle_decoder, cfg, z["it_T_oc"][i], z["it_code"][i], z["pts"], z["rays"], dobs, n_fg)
        assert it["fail"] is None
        assert it["K"] == int(z["it_K"][i])
        assert relerr(it["H"], z["it_H"][i]) < tol
        assert relerr(it["b"], z["it_b"][i]) < tol
        assert relerr(it["dx"], z["it_dx"][i]) < 20 * tol     # cond(H) ~ 1e3 amplifies the differences in H
        if i + 1 < n_it:
            assert relerr(it["T_oc_new"], z["it_T_oc"][i + 1]) < tol
            assert np.abs(it["code_new"] - z["it_code"][i + 1]).max() < tol


@pytest.mark.parametrize("name", JOINT_CASES)
def test_teacher_forced_error_against_the_references_own_rounding_noise(oracle_decoder, golden_dir, name):
    """VERDICT r3 item 1.  The quantities the test above holds ABOVE 1e-4 (`dx`, the KITTI `b`, the next state) are compared with
    the reference's float64 evaluation of the same iteration, next to how far the reference's own seven float32 evaluations
    land from it (tests/noise.py, tests/golden/sdf_noise_*.npz <- oracle/gen_noise_sdf.py): within 1e-4, or within twice the
    reference's own scatter, for every quantity and iteration."""
    from tests import noise
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    nz = noise.load(golden_dir, name)
    cfg = cfg_from(z)
    n_fg = z["depth"].shape[0]
    dobs = np.concatenate([z["depth"], np.zeros(z["rays"].shape[0] - n_fg, np.float32)])
    worst = {}
    for i in range(z["it_H"].shape[0]):
        it = so.gn_iteration(oracle_decoder, cfg, z["it_T_oc"][i], z["it_code"][i], z["pts"], z["rays"], dobs, n_fg)
        assert it["K"] == int(nz["K64"][i]) == int(z["it_K"][i]) and all(int(k) == it["K"] for k in nz["K32"][i])
        mine = dict(H=it["H"], b=it["b"], dx=it["dx"], T_next=it["T_oc_new"], code_next=it["code_new"])
        for q in noise.QUANTITIES:
            r, err, nse = noise.ratio(z, nz, q, i, mine[q], cfg.k4, (it["J_rot"], it["res_rot"]))
            worst[q] = max(worst.get(q, 0.0), r)
            assert r <= 2.0, (name, i, q, err, nse)
        if cfg.k4 != 0.0:
            e, bar = noise.res_rot_error(nz, i, it["res_rot"])
            assert e <= bar, (name, i, e, bar)
    print(name, {k: round(v, 2) for k, v in worst.items()})


@pytest.mark.parametrize("name", JOINT_CASES)
def test_reconstruct_object_free_running(oracle_decoder, golden_dir, name):
    """Free-running end-to-end result.  The iteration map of reconstruct_object amplifies a perturbation of its state by
    about 5-8x per iteration (measured between the reference under torch-CPU and this restatement: 1e-7 -> 6e-6 -> 5e-5
    -> 2e-4 -> 8e-4 on sdf_joint_redwood_m2000), so two correct float32 implementations that only differ in summation
    order end 5 iterations ~1e-3 apart and 10 iterations further.  The bound here is therefore loose by construction;
    the tight bound is the teacher-forced test above."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    r = so.reconstruct_object(oracle_decoder, cfg_from(z), z["t_cam_obj"], z["pts"], z["rays"], z["depth"])
    assert r["is_good"] == bool(z["is_good"])
    assert relerr(r["t_cam_obj"], z["out_t_cam_obj"]) < 2e-2
    assert np.abs(r["code"] - z["out_code"]).max() < 2e-2
    assert abs(r["loss"] - float(z["loss"])) < 5e-2 * abs(float(z["loss"]))


def test_reconstruct_object_failure_exit(oracle_decoder, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_joint_fail_norays.npz"))
    r = so.reconstruct_object(oracle_decoder, cfg_from(z), z["t_cam_obj"], z["pts"], z["rays"], z["depth"])
    assert r["is_good"] is False and not bool(z["is_good"])
    assert r["t_cam_obj"] is None and r["code"] is None
    assert r["loss"] == float(z["loss"]) == 0.0


def test_pose_only(oracle_decoder, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_pose_only_m250.npz"))
    out = so.estimate_pose_cam_obj(oracle_decoder, so.JointConfig(), z["t_co_se3"], float(z["scale"]), z["pts"], z["code"])
    assert relerr(out, z["out"]) < 1e-4


# ---- a second member of the decoder family: 4 hidden layers x 256, code 32, latent_in [2] (the `code_len == 32` branch of
# ---- src/LocalMapping_util.cc:789-800); fixtures produced by running the reference (oracle/gen_golden_sdf.py small)
@pytest.fixture(scope="module")
def small_decoder(golden_dir):
    return so.load_decoder_npz(os.path.join(golden_dir, "decoder_4x256_c32.npz"))


def test_small_decoder_value_and_grad(small_decoder, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_small_decoder_vectors.npz"))
    assert small_decoder.code_len == 32 and len(small_decoder.layers) == 5 and tuple(small_decoder.latent_in) == (2,)
    assert np.abs(so.decode_sdf(small_decoder, z["code"], z["x"]) - z["sdf"]).max() < 2e-6
    inp = np.concatenate([np.broadcast_to(z["code"], (z["x"].shape[0], 32)), z["x"]], -1)
    y, g = so.decoder_value_and_input_grad(small_decoder, inp)
    assert g.shape == (300, 35) and np.abs(y - z["y"]).max() < 2e-6
    assert rows_close(g, z["grad"], tol=1e-5, max_bad=0.01)


def test_small_decoder_teacher_forced_iterations(small_decoder, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_small_joint_m400.npz"))
    c = cfg_from(z)
    cfg = so.JointConfig(k1=c.k1, k2=c.k2, k3=c.k3, k4=c.k4, b1=c.b1, b2=c.b2, lr=c.lr, s_damp=c.s_damp, n_iter=c.n_iter, code_len=32)
    dobs = np.concatenate([z["depth"], np.zeros(z["rays"].shape[0] - z["depth"].shape[0], np.float32)])
    assert z["it_H"].shape[1:] == (39, 39)
    for i in range(z["it_H"].shape[0]):
        it = so.gn_iteration(small_decoder, cfg, z["it_T_oc"][i], z["it_code"][i], z["pts"], z["rays"], dobs, z["depth"].shape[0])
        assert it["fail"] is None and it["K"] == int(z["it_K"][i])
        assert relerr(it["H"], z["it_H"][i]) < 1e-4 and relerr(it["b"], z["it_b"][i]) < 1e-4
