"""The split-precision decoder pipes (QSP_DEC_OPT_FORWARD_PRECISION / QSP_DEC_OPT_JACOBIAN_PRECISION, csrc/sdf_mlp.hpp) --
"bf16x3": three bf16 terms per f32 operand, six products per multiply-add on the bf16 matrix pipe; "fp16x2": two fp16 terms
(the second pre-scaled by 2^11), three products on the fp16 matrix pipe, four waves per workgroup; f32 accumulation in both --
are held to the SAME gates as the exact-f32 tile before bench.py may quote them:
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


@pytest.fixture(scope="module", params=["bf16x3", "fp16x2", "fp16x2_t32"])
def bf3_decoder(golden_dir, request):
    """"fp16x2_t32": the split-fp16 pipe with 32-point tiles (QSP_DEC_OPT_TILE_POINTS, the one-object latency option)"""
    from qsp_slam_amd import DeepSdfDecoder
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    d.set_precision(request.param.split("_")[0])
    if request.param.endswith("_t32"):
        d.set_tile_points(32)
        d.precision = request.param          # (tag of the recorded margins)
    yield d
    d.close()


def test_decoder_value_and_grad_vs_reference_vectors(bf3_decoder, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_decoder_vectors.npz"))
    assert within(bf3_decoder.precision + "/decoder/sdf_abs", np.abs(bf3_decoder.decode_sdf(z["code"], z["x"]) - z["sdf"]).max(), 2e-6)
    y, g = bf3_decoder.sdf_value_grad(z["code"], z["x"])
    assert within(bf3_decoder.precision + "/decoder/y_abs", np.abs(y - z["y"]).max(), 2e-6)
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
        tag = bf3_decoder.precision + "/" + name + "/teacher_forced/"
        assert within(tag + "H", relerr(tr["H"][0], z["it_H"][i]), 1e-4)
        assert within(tag + "b", relerr(tr["b"][0], z["it_b"][i]), 1.5e-2 if kitti else 1e-4)     # (k4 = 1e7: test_gpu_sdf.py)
        assert within(tag + "dx", relerr(tr["dx"][0], z["it_dx"][i]), 6e-3 if kitti else 2.5e-3)
        if i + 1 < n_it:
            assert within(tag + "T_oc_next", relerr(np.linalg.inv(T[0].astype(np.float64)), z["it_T_oc"][i + 1]),
                          2.5e-4 if kitti else 1e-4)
            assert within(tag + "code_next_abs", np.abs(code[0] - z["it_code"][i + 1]).max(), 1e-4)
    batch.close()


@pytest.mark.parametrize("name", JOINT_CASES)
def test_every_iteration_vs_the_references_float64_and_its_own_float32_scatter(bf3_decoder, golden_dir, name):
    """the same yardstick as the f32 tile's (tests/test_gpu_sdf.py, tests/noise.py): within 1e-4 of the reference's float64
    evaluation, or within twice the reference's own float32 scatter around it"""
    from tests import noise
    noise.check_gpu_iterations(bf3_decoder.precision, bf3_decoder, golden_dir, name, make_cfg, cfg_from, within)


@pytest.mark.parametrize("name", JOINT_CASES)
def test_free_running_result_vs_the_references_float64_and_its_own_float32_scatter(bf3_decoder, golden_dir, name):
    """the whole refinement (5 / 10 iterations, a map that amplifies rounding 5-8 x per iteration) ends within twice the distance
    at which the reference's own float32 evaluations end from its float64 one"""
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    from tests import noise
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    r = Optimizer(bf3_decoder, make_cfg(z)).reconstruct_object(z["t_cam_obj"], z["pts"], z["rays"], z["depth"])
    assert r.is_good
    noise.check_free_running(bf3_decoder.precision, name, golden_dir, r, within)


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
    assert within(bf3_decoder.precision + "/grid/sdf_abs_vs_f32", np.abs(a - b).max(), 5e-7)
    flips = (a > 0) != (b > 0)
    assert np.abs(a[flips]).max(initial=0.0) < 5e-7
    f32.close()


# ---- split fp16 only: range guard, the decoder family, and the generic decode entry points on odd sizes ---------------------
@pytest.fixture(scope="module")
def h2_decoder(golden_dir):
    from qsp_slam_amd import DeepSdfDecoder
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    d.set_precision("fp16x2")
    yield d
    d.close()


def _scaled_decoder(golden_dir, layer, factor, rows=None):
    """the golden decoder with the weight-norm gain of one layer (or of some of its rows) multiplied: same function shape,
    larger activations / weights.  (A WHOLE layer's gain is undone by the library's exact power-of-two gain equalisation,
    csrc/sdf_refine.hip:equalize_gains; single rows are not.)"""
    from qsp_slam_amd import DeepSdfDecoder
    z = dict(np.load(os.path.join(golden_dir, "decoder_8x512.npz")))
    g = z["lin%d.weight_g" % layer].copy()
    if rows is None:
        g *= np.float32(factor)
    else:
        g[rows] *= np.float32(factor)
    z["lin%d.weight_g" % layer] = g
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".npz", delete=False) as f:
        np.savez(f.name, **z)
        path = f.name
    try:
        return DeepSdfDecoder.from_npz(path)
    finally:
        os.unlink(path)


def test_fp16_refuses_weights_outside_its_range(golden_dir, monkeypatch):
    from qsp_slam_amd._lib import QspError
    monkeypatch.delenv("QSP_PRECISION", raising=False)  # a session-wide default would refuse at construction
    d = _scaled_decoder(golden_dir, 2, 3e6, rows=[0, 5])   # |w| of two units of layer 2 up to ~1e5 > 65504
    with pytest.raises(QspError):
        d.set_precision("fp16x2")
    d.set_precision("bf16x3")                         # the other pipes take it
    d.close()


def test_fp16_range_excursion_falls_back_to_f32_or_fails_loudly(golden_dir, monkeypatch):
    """weights inside fp16's range but an activation beyond 65504: the kernels raise the decoder's range flag.  Default
    (QSP_DEC_OPT_RANGE_FALLBACK = 1): the library repeats THAT call on its exact-f32 pipe and returns normally -- the reference
    never fails a call for a numeric condition (reconstruct/optimizer.py:161-194) -- and counts it.  With the option off the call
    fails with QSP_ERR_UNSUPPORTED instead of returning values computed from infinite planes (round 2's behaviour)."""
    from qsp_slam_amd import _lib, synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    monkeypatch.delenv("QSP_PRECISION", raising=False)
    d = _scaled_decoder(golden_dir, 1, 6e5, rows=list(range(64)))    # 64 units of layer 1: weights up to 5.9e4, activations to 3e5
    x = np.random.default_rng(0).uniform(-1, 1, size=(300, 3)).astype(np.float32)
    code = np.zeros(64, np.float32)
    ref = d.decode_sdf(code, x)
    ref_y, ref_g = d.sdf_value_grad(code, x)
    assert np.isfinite(ref).all()
    d.set_precision("fp16x2")
    assert d.range_fallbacks == 0
    assert np.array_equal(d.decode_sdf(code, x), ref) and d.range_fallbacks == 1          # the f32 pipe's own bits
    y, g = d.sdf_value_grad(code, x)
    assert np.array_equal(y, ref_y) and np.array_equal(g, ref_g) and d.range_fallbacks == 2
    # the refinement entry point: same result as a decoder that was on the f32 pipe all along, is_good, no exception
    o = synth.make_object_views(9, 1, 300, n_fg=64, n_bg=32)[0]
    opt = Optimizer(d, make_cfg(so.JointConfig(n_iter=2)))
    r = opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
    n_fb = d.range_fallbacks
    assert n_fb >= 3
    d.set_precision("f32")
    r32 = Optimizer(d, make_cfg(so.JointConfig(n_iter=2))).reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
    assert r.is_good == r32.is_good and np.array_equal(r.t_cam_obj, r32.t_cam_obj) and np.array_equal(r.code, r32.code)
    assert d.range_fallbacks == n_fb                  # nothing to fall back from on the f32 pipe
    # the old contract on request
    d.set_precision("fp16x2")
    d.set_range_fallback(False)
    with pytest.raises(_lib.QspError) as e:
        d.decode_sdf(code, x)
    assert e.value.code == _lib.QSP_ERR_UNSUPPORTED
    with pytest.raises(_lib.QspError):
        d.sdf_value_grad(code, x)
    d.set_range_fallback(True)
    d.set_precision("bf16x3")
    assert np.abs(d.decode_sdf(code, x) - ref).max() < 1e-5
    d.close()


def test_fp16_low_end_of_the_range_skewed_layer_gains(golden_dir, monkeypatch):
    """the golden decoder with one layer's gain x 1e-4 and the next layer's x 1e4 is the same function, but
    as given it puts a whole layer's activations at ~1e-4 x O(1), where x_hi is an fp16 subnormal.  The library's exact
    power-of-two gain equalisation (equalize_gains) brings the layer back before the planes are packed: the skewed decoder meets
    the same 2e-6 / 1e-5 decoder gates on the split-fp16 pipe, gradients of magnitude 1e-6..1e-8 included; and a skew by exact
    powers of two gives the unskewed decoder's bits on the f32 and split-bf16 pipes (on the split-fp16 pipe to 2e-7: the few
    activations below fp16's smallest normal number round differently when the layer is rescaled by a power of two)."""
    from qsp_slam_amd import DeepSdfDecoder
    monkeypatch.delenv("QSP_PRECISION", raising=False)
    gold = os.path.join(golden_dir, "decoder_8x512.npz")
    z = np.load(os.path.join(golden_dir, "sdf_decoder_vectors.npz"))
    od = so.load_decoder_npz(gold)

    def skewed(down, up, layer):
        layers = []
        for l, (W, b) in enumerate(od.layers):
            W, b = W.copy(), b.copy()
            if l == layer:
                W *= np.float32(down)
                b *= np.float32(down)
            if l == layer + 1:
                W[:, :od.layers[layer][0].shape[0]] *= np.float32(up)      # (the columns fed by layer `layer`, not a skip's)
            layers.append((W, None, b))
        return DeepSdfDecoder(layers, latent_in=od.latent_in, code_len=od.code_len)

    base = skewed(1.0, 1.0, 0)           # (the same folded weights as the skewed ones: numpy's fold, not the library's)
    for layer in (1, 3, 5):
        d = skewed(1e-4, 1e4, layer)
        p2 = skewed(2.0 ** -13, 2.0 ** 13, layer)
        for prec in ("fp16x2", "bf16x3", "f32"):
            d.set_precision(prec)
            p2.set_precision(prec)
            base.set_precision(prec)
            tag = "%s/skew_layer%d/" % (prec, layer)
            assert within(tag + "sdf_abs", np.abs(d.decode_sdf(z["code"], z["x"]) - z["sdf"]).max(), 2e-6)
            y, g = d.sdf_value_grad(z["code"], z["x"])
            assert within(tag + "y_abs", np.abs(y - z["y"]).max(), 2e-6)
            assert rows_close(g, z["grad"], tol=1e-5, max_bad=0.01)
            yb, gb = base.sdf_value_grad(z["code"], z["x"])
            y2, g2 = p2.sdf_value_grad(z["code"], z["x"])
            if prec == "fp16x2":      # (activations below 6.1e-5 are fp16 subnormals: their rounding is not scale invariant)
                assert within(tag + "pow2_skew_y_abs", np.abs(y2 - yb).max(), 2e-7)
                assert rows_close(g2, gb, tol=2e-6, max_bad=0.01)
            else:
                assert np.array_equal(y2, yb) and np.array_equal(g2, gb), (prec, layer)
        assert d.range_fallbacks == 0
        d.close()
        p2.close()
    base.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000])
def test_fp16_decode_entry_points_on_ragged_sizes(h2_decoder, oracle_decoder, n):
    rng = np.random.default_rng(n)
    x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    code = (0.1 * rng.normal(size=64)).astype(np.float32)
    inp = np.concatenate([np.broadcast_to(code, (n, 64)), x], -1).astype(np.float32)
    yo, go = so.decoder_value_and_input_grad(oracle_decoder, inp)
    y = h2_decoder.decode_sdf(code, x)
    y2, g = h2_decoder.sdf_value_grad(code, x)
    assert within("fp16x2/decode/y_abs_vs_oracle", np.abs(y - np.asarray(yo).reshape(-1)).max(), 2e-6)
    assert np.array_equal(y, y2)
    assert rows_close(g, go, tol=1e-5, max_bad=0.01)


def test_fp16_on_the_small_decoder_of_the_family(golden_dir):
    """4 x 256, code 32 (embedded exactly into the 8 x 512 tile): values, gradients and one joint iteration against the
    reference-generated vectors"""
    from qsp_slam_amd import DeepSdfDecoder
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_4x256_c32.npz"))
    d.set_precision("fp16x2")
    z = np.load(os.path.join(golden_dir, "sdf_small_decoder_vectors.npz"))
    assert within("fp16x2/small/sdf_abs", np.abs(d.decode_sdf(z["code"], z["x"]) - z["sdf"]).max(), 2e-6)
    y, g = d.sdf_value_grad(z["code"], z["x"])
    assert within("fp16x2/small/y_abs", np.abs(y - z["y"]).max(), 2e-6)
    assert rows_close(g, z["grad"], tol=1e-5, max_bad=0.01)
    d.close()


def test_32_point_tiles_exist_on_the_split_fp16_pipe_only(golden_dir):
    from qsp_slam_amd import DeepSdfDecoder, synth
    from qsp_slam_amd._lib import QspError
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    d.set_tile_points(32)
    o = synth.make_object_views(5, 1, 200, n_fg=40, n_bg=10)[0]
    opt = Optimizer(d, make_cfg(so.JointConfig()))
    for prec in ("f32", "bf16x3"):
        d.set_precision(prec)
        with pytest.raises(QspError):
            RefineBatch(d, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0])
    d.set_precision("fp16x2")
    RefineBatch(d, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0]).close()
    with pytest.raises(QspError):
        d.set_tile_points(48)
    d.close()
