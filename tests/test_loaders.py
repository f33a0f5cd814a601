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
def test_unsupported_decoder_family_is_refused(golden_dir):
    """anything but 9 layers / code 64 / latent_in [4] / 8x512 is QSP_ERR_UNSUPPORTED, not a silent fallback"""
    from qsp_slam_amd import DeepSdfDecoder, _lib
    z = np.load(os.path.join(golden_dir, "decoder_8x512.npz"), allow_pickle=False)
    state = {k: z[k] for k in z.files if k != "meta"}
    with pytest.raises(_lib.QspError):
        DeepSdfDecoder.from_state_dict(state, latent_in=(3,), code_len=64)
    small = {k: v for k, v in state.items() if not k.startswith("lin8")}
    with pytest.raises(_lib.QspError):
        DeepSdfDecoder.from_state_dict(small, latent_in=(4,), code_len=64)
