"""The NARROW form of the split-fp16 tile (csrc/sdf_mlp.hpp: mlp_tile_h2<.., NARROW>): a decoder much smaller than the 8 x 512 shape it
is embedded in -- the 4 x 256 / code 32 member of the family deep_sdf_decoder.py:29-63 builds, the `code_len == 32` branch of
src/LocalMapping_util.cc:789-800 -- skips identity slots, all-zero k-slabs and all-zero column blocks.  Everything skipped is a
product with an exact zero: the reference-generated fixtures of that decoder hold as they do for the embedded form, the two forms
agree to float32 rounding, and the narrow form is several times faster."""
import os
import time

import numpy as np
import pytest

from oracle import sdf_oracle as so
from tests.margins import within
from tests.test_gpu_sdf import make_cfg
from tests.test_oracle_sdf import cfg_from, relerr, rows_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(golden_dir):
    from qsp_slam_amd import DeepSdfDecoder
    d = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_4x256_c32.npz"))
    d.set_precision("fp16x2")
    yield d
    d.close()


def test_which_decoders_run_the_narrow_form(small, golden_dir):
    from qsp_slam_amd import DeepSdfDecoder, _lib
    assert small.narrow_tile
    big = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    assert not big.narrow_tile
    with pytest.raises(_lib.QspError) as e:
        big.set_narrow_tile(True)                       # fills the shape: refused
    assert e.value.code == _lib.QSP_ERR_UNSUPPORTED
    big.close()


def test_small_decoder_vectors_and_joint_case_vs_reference_on_the_narrow_tile(small, golden_dir):
    z = np.load(os.path.join(golden_dir, "sdf_small_decoder_vectors.npz"))
    assert within("fp16x2_narrow/small/sdf_abs", np.abs(small.decode_sdf(z["code"], z["x"]) - z["sdf"]).max(), 2e-6)
    y, g = small.sdf_value_grad(z["code"], z["x"])
    assert within("fp16x2_narrow/small/y_abs", np.abs(y - z["y"]).max(), 2e-6)
    assert rows_close(g, z["grad"], tol=1e-5, max_bad=0.01)
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    j = np.load(os.path.join(golden_dir, "sdf_small_joint_m400.npz"))
    opt = Optimizer(small, make_cfg(j, code_len=32))
    batch = RefineBatch(small, _joint_cfg(opt), [j["pts"]], [j["rays"]], [j["depth"]], [0])
    n_it = j["it_H"].shape[0]
    for i in range(n_it):
        T_co = np.linalg.inv(j["it_T_oc"][i].astype(np.float64)).astype(np.float32)
        batch.set_state(T_co[None], j["it_code"][i][None])
        batch.run(1)
        tr = batch.trace()
        T, code, loss, good = batch.get()
        n = 7 + 32
        assert good[0] and int(tr["K"][0]) == int(j["it_K"][i])
        tag = "fp16x2_narrow/small_joint/teacher_forced/"
        assert within(tag + "H", relerr(tr["H"][0][:n, :n], j["it_H"][i]), 1e-4)
        assert within(tag + "b", relerr(tr["b"][0][:n], j["it_b"][i]), 1e-4)
        assert within(tag + "dx", relerr(tr["dx"][0][:n], j["it_dx"][i]), 2.5e-3)
    batch.close()


def test_narrow_and_embedded_forms_agree_and_the_narrow_one_is_faster(small):
    """the same batch on both forms of the tile: K / n_valid equal, H, b, final states to float32 rounding (an identity layer of the
    embedded form rounds x_hi + 2^-11 x_lo' to float32 where the skip keeps the pair); timing recorded, >= 2.5 x asserted"""
    import bench
    from qsp_slam_amd import synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    objs = synth.make_object_views(808, 16, 2000, n_fg=200, n_bg=100)
    T0, hyp = bench.flip_states(objs, 4)
    cfg = so.JointConfig(n_iter=3)
    out, ms = {}, {}
    for form in (True, False):
        small.set_narrow_tile(form)
        opt = Optimizer(small, make_cfg(cfg, code_len=32))
        b = RefineBatch(small, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp)
        b.set_state(T0, None)
        b.run(1)                       # one iteration: discrete counts must agree exactly
        first = b.trace()
        for rep in range(2):           # three free-running iterations: timing, and the final state to a loose bar (chaotic map)
            b.set_state(T0, None)
            t0 = time.perf_counter()
            b.run(0)
            ms[form] = 1e3 * (time.perf_counter() - t0)
        out[form] = (first, b.get())
        b.close()
    small.set_narrow_tile(True)
    (ta, sa), (tb, sb) = out[True], out[False]
    assert np.array_equal(ta["K"], tb["K"]) and np.array_equal(ta["n_valid"], tb["n_valid"]) and sa[3].all() and sb[3].all()
    assert within("fp16x2_narrow/vs_embedded/H", relerr(ta["H"], tb["H"]), 2e-5)
    assert within("fp16x2_narrow/vs_embedded/b", relerr(ta["b"], tb["b"]), 2e-5)
    assert within("fp16x2_narrow/vs_embedded/T_final_3_iterations", relerr(sa[0], sb[0]), 2e-2)
    assert within("fp16x2_narrow/narrow_over_embedded_time", ms[True] / ms[False], 0.5)    # measured 0.29 (3.5 x); the bar leaves room for a noisy box


def test_random_narrow_members_of_the_family_vs_the_oracle():
    """narrow shapes specs.json may ask for, random weights, against the numpy decoder on the split-fp16 pipe: with and without
    a latent_in layer, unequal widths, codes of 8 / 32 / 64"""
    from qsp_slam_amd import DeepSdfDecoder
    rng = np.random.default_rng(15)

    def family(L, dims, latent_in):
        full = [L + 3] + list(dims) + [1]
        layers = []
        for l in range(len(full) - 1):
            out = full[l + 1] - (full[0] if (l + 1) in latent_in else 0)
            w = (rng.normal(size=(out, full[l])) / np.sqrt(full[l])).astype(np.float32)
            layers.append((w, None, (0.1 * rng.normal(size=out)).astype(np.float32)))
        return layers
    cases = [(32, [256] * 4, (2,)), (64, [128, 192, 96], ()), (8, [64, 64], (1,)), (16, [200] * 7, (3,)), (32, [96, 320, 160, 40], (3,)),
             (64, [256] * 8, (4,))]
    for L, dims, lin in cases:
        layers = family(L, dims, lin)
        dec = DeepSdfDecoder(layers, latent_in=lin, code_len=L)
        dec.set_precision("fp16x2")
        assert dec.narrow_tile, (L, dims, lin)
        ref = so.DecoderWeights([(w, b) for w, _, b in layers], lin, L)
        x = rng.uniform(-1, 1, size=(300, 3)).astype(np.float32)
        code = (0.3 * rng.normal(size=L)).astype(np.float32)
        assert np.abs(dec.decode_sdf(code, x) - so.decode_sdf(ref, code, x)).max() < 5e-6, (L, dims, lin)
        inp = np.concatenate([np.broadcast_to(code, (300, L)), x], -1)
        yr, gr = so.decoder_value_and_input_grad(ref, inp)
        y, g = dec.sdf_value_grad(code, x)
        assert g.shape == (300, L + 3) and np.abs(y - yr).max() < 5e-6
        d = np.abs(g - gr).max(1) / np.abs(gr).max()
        assert (d > 1e-5).mean() <= 0.02, (L, dims, lin, float(d.max()))
        dec.close()
