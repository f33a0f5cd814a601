"""Config / decoder loading entry points with the reference's names: reconstruct/utils.py:82-95 (ForceKeyErrorDict,
get_configs, get_decoder) and deep_sdf/workspace.py:202-224 (config_decoder: specs.json + ModelParameters/latest.pth saved
from a DataParallel module, "module." prefixes)."""
import ast
import json
import os

import numpy as np
import pytest


def test_get_configs_force_key_error_semantics(tmp_path):
    from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict, get_configs
    cfg = {"data_type": "KITTI", "DeepSDF_DIR": "weights/x", "optimizer": {"code_len": 64, "joint_optim": {"k1": 1.0}}}
    p = tmp_path / "config.json"
    p.write_text(json.dumps(cfg))
    c = get_configs(str(p))
    assert isinstance(c, ForceKeyErrorDict) and isinstance(c.optimizer, ForceKeyErrorDict)
    assert c.data_type == "KITTI" and c.optimizer.joint_optim.k1 == 1.0 and c["optimizer"]["code_len"] == 64
    with pytest.raises(KeyError):
        c.missing_key                                 # reconstruct/utils.py:82-84: __missing__ raises
    with pytest.raises(KeyError):
        c.optimizer.joint_optim.k9
    c.new_value = {"a": 1}                            # nested dicts are wrapped on assignment, as addict does
    assert c.new_value.a == 1


def write_experiment(dirname, golden_dir, prefix="module."):
    """an experiment directory in the reference's layout, from the committed decoder fixture"""
    import torch
    z = np.load(os.path.join(golden_dir, "decoder_8x512.npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    os.makedirs(os.path.join(dirname, "ModelParameters"), exist_ok=True)
    specs = {"NetworkArch": "deep_sdf_decoder", "CodeLength": int(meta["latent_size"]),
             "NetworkSpecs": {"dims": [512] * 8, "dropout": list(range(8)), "dropout_prob": 0.2, "norm_layers": list(range(8)),
                              "latent_in": list(meta["latent_in"]), "xyz_in_all": False, "use_tanh": False,
                              "latent_dropout": False, "weight_norm": True}}
    with open(os.path.join(dirname, "specs.json"), "w") as f:
        json.dump(specs, f)
    state = {prefix + k: torch.from_numpy(np.array(z[k])) for k in z.files if k != "meta"}
    torch.save({"epoch": 2000, "model_state_dict": state}, os.path.join(dirname, "ModelParameters", "latest.pth"))


def test_missing_specs_file_raises(tmp_path):
    from qsp_slam_amd import DeepSdfDecoder
    with pytest.raises(Exception, match="specs.json"):
        DeepSdfDecoder.from_experiment_dir(str(tmp_path))


@pytest.mark.gpu
def test_get_decoder_from_experiment_directory(tmp_path, golden_dir):
    """get_decoder(configs) -> config_decoder: the decoder loaded from specs.json + latest.pth (DataParallel prefixes)
    evaluates exactly like the one built from the fixture directly"""
    from qsp_slam_amd import DeepSdfDecoder
    from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict, get_decoder
    write_experiment(str(tmp_path / "exp"), golden_dir)
    dec = get_decoder(ForceKeyErrorDict(DeepSDF_DIR=str(tmp_path / "exp")))
    ref = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(300, 3)).astype(np.float32)
    code = (0.1 * rng.standard_normal(64)).astype(np.float32)
    assert np.array_equal(dec.decode_sdf(code, x), ref.decode_sdf(code, x))
    ya, ga = dec.sdf_value_grad(code, x)
    yb, gb = ref.sdf_value_grad(code, x)
    assert np.array_equal(ya, yb) and np.array_equal(ga, gb)
    assert dec.code_len == 64
    dec.close()
    ref.close()


@pytest.mark.gpu
def test_get_decoder_honours_the_decoder_precision_key(tmp_path, golden_dir):
    """`"decoder_precision"` in the JSON config selects the decoder's arithmetic pipe; absent = exact f32; unknown = error"""
    from qsp_slam_amd import DeepSdfDecoder
    from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict, get_decoder
    write_experiment(str(tmp_path / "exp"), golden_dir)
    ref = DeepSdfDecoder.from_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    ref.set_precision("fp16x2")
    x = np.random.default_rng(1).uniform(-1, 1, size=(200, 3)).astype(np.float32)
    code = np.zeros(64, np.float32)
    dec = get_decoder(ForceKeyErrorDict(DeepSDF_DIR=str(tmp_path / "exp"), decoder_precision="fp16x2"))
    assert dec.precision == "fp16x2" and np.array_equal(dec.decode_sdf(code, x), ref.decode_sdf(code, x))
    dec.close()
    ref.close()
    with pytest.raises(ValueError):
        get_decoder(ForceKeyErrorDict(DeepSDF_DIR=str(tmp_path / "exp"), decoder_precision="fp8"))


@pytest.mark.gpu
def test_unsupported_decoder_family_is_refused(golden_dir):
    """what cannot be mapped exactly onto the 8 x 512 tile (csrc/sdf_refine.hip:embed_family) is QSP_ERR_UNSUPPORTED, not a
    silent fallback: a latent_in layer with more than 4 hidden layers in front of or from it on, hidden widths above 512,
    two latent_in layers, input widths that do not chain"""
    from qsp_slam_amd import DeepSdfDecoder, _lib
    z = np.load(os.path.join(golden_dir, "decoder_8x512.npz"), allow_pickle=False)
    state = {k: z[k] for k in z.files if k != "meta"}
    for latent_in in ((3,), (5,), (4, 6)):            # dims no longer chain / two skips
        with pytest.raises(_lib.QspError) as e:
            DeepSdfDecoder.from_state_dict(state, latent_in=latent_in, code_len=64)
        assert e.value.code == _lib.QSP_ERR_UNSUPPORTED
    rng = np.random.default_rng(0)

    def mlp(dims_in_out):
        return [(rng.normal(size=(o, i)).astype(np.float32) * 0.05, None, np.zeros(o, np.float32)) for i, o in dims_in_out]
    wide = mlp([(67, 640), (640, 640 - 67), (640, 640), (640, 1)])           # width 640 > 512
    deep = mlp([(67, 64)] + [(64, 64)] * 9 + [(64, 1)])                       # 10 hidden layers
    # no latent_in and every candidate for the slot in front of the (absent) skip wider than 445: the slot is 445 wide with or
    # without a skip (these used to be accepted and evaluated wrongly / written past the packed rows)
    full4 = mlp([(67, 512), (512, 512), (512, 512), (512, 512), (512, 1)])
    full8 = mlp([(67, 512)] + [(512, 512)] * 7 + [(512, 1)])
    for layers, lin in ((wide, (2,)), (deep, ()), (full4, ()), (full8, ())):
        with pytest.raises(_lib.QspError) as e:
            DeepSdfDecoder(layers, latent_in=lin, code_len=64)
        assert e.value.code == _lib.QSP_ERR_UNSUPPORTED


@pytest.mark.gpu
def test_decoder_family_members_match_the_oracle():
    """shapes specs.json may ask for (deep_sdf/deep_sdf_decoder.py:29-63), random weights, against the numpy decoder: no
    latent_in at all, latent_in at the first possible layer, 8 hidden layers of unequal widths, code lengths 8 / 32 / 64"""
    from oracle import sdf_oracle as so
    from qsp_slam_amd import DeepSdfDecoder
    rng = np.random.default_rng(5)

    def family(L, dims, latent_in):
        full = [L + 3] + list(dims) + [1]
        layers = []
        for l in range(len(full) - 1):
            out = full[l + 1] - (full[0] if (l + 1) in latent_in else 0)
            w = (rng.normal(size=(out, full[l])) / np.sqrt(full[l])).astype(np.float32)
            layers.append((w, None, (0.1 * rng.normal(size=out)).astype(np.float32)))
        return layers
    cases = [(32, [256] * 4, (2,)), (64, [128, 192, 96], ()), (8, [64, 64], (1,)),
             (64, [512, 300, 400, 512, 256, 512, 100, 512], (4,)), (16, [200] * 7, (3,)), (64, [512] * 5, (1,)),
             # no latent_in, wide layers: the split goes behind the first layer that fits the 445-wide slot
             (64, [512, 400, 512, 512], ()), (64, [512, 512, 512, 445, 512, 512, 512, 512], ()), (32, [500, 500, 300, 500, 500], ())]
    for L, dims, lin in cases:
        layers = family(L, dims, lin)
        dec = DeepSdfDecoder(layers, latent_in=lin, code_len=L)
        ref = so.DecoderWeights([(w, b) for w, _, b in layers], lin, L)
        x = rng.uniform(-1, 1, size=(200, 3)).astype(np.float32)
        code = (0.3 * rng.normal(size=L)).astype(np.float32)
        assert np.abs(dec.decode_sdf(code, x) - so.decode_sdf(ref, code, x)).max() < 5e-6, (L, dims, lin)
        inp = np.concatenate([np.broadcast_to(code, (200, L)), x], -1)
        yr, gr = so.decoder_value_and_input_grad(ref, inp)
        y, g = dec.sdf_value_grad(code, x)
        assert g.shape == (200, L + 3) and np.abs(y - yr).max() < 5e-6
        d = np.abs(g - gr).max(1) / np.abs(gr).max()
        assert (d > 1e-5).mean() <= 0.02, (L, dims, lin, float(d.max()))
        dec.close()
