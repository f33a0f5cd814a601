"""The split-fp16 Jacobian kernel exists with four waves of 512 registers per workgroup and with eight of 256 (csrc/sdf_mlp.hpp:
mlp_tile_h2<.., NW>), and so does the screening pass of the render forward (mlp_tile_h1<.., NW>): the same arithmetic in the same
order, so the two must give the same bits.  Each variant is selected per process (QSP_JTJ_WAVES, QSP_JTJ_WAVES_T32,
QSP_SCREEN_WAVES), so the comparison runs the same seeded batch in two child processes."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import bench
from oracle import sdf_oracle as so            # (test process: the checker's config type only)
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
from tests.test_gpu_sdf import make_cfg
dec = DeepSdfDecoder.from_npz(os.path.join(sys.argv[1], "tests", "golden", "decoder_8x512.npz"))
dec.set_precision("fp16x2")
dec.set_render_screening(0.01)          # (the screening pass has the two forms as well: QSP_SCREEN_WAVES)
dec.set_screening_min_samples(0)
out = {}
for tile in (64, 32):
    dec.set_tile_points(tile)
    objs = synth.make_object_views(4242, 5, 900, n_fg=140, n_bg=70)
    T0, hyp = bench.flip_states(objs, 4)
    opt = Optimizer(dec, make_cfg(so.JointConfig(n_iter=3)))
    b = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp)
    b.set_state(T0, None)
    b.run(0)
    tr = b.trace()
    T, code, loss, good = b.get()
    for k, v in dict(H=tr["H"], b=tr["b"], K=tr["K"], T=T, code=code, loss=loss, good=good).items():
        out["t%d_%s" % (tile, k)] = np.asarray(v)
    b.close()
np.savez(sys.argv[2], **out)
'''


def test_four_and_eight_wave_kernels_give_the_same_bits(tmp_path):
    res = {}
    for waves in ("4", "8"):
        path = str(tmp_path / ("w%s.npz" % waves))
        env = dict(os.environ, QSP_JTJ_WAVES=waves, QSP_JTJ_WAVES_T32=waves, QSP_SCREEN_WAVES=waves)
        subprocess.run([sys.executable, "-c", CHILD, ROOT, path], check=True, env=env, timeout=600)
        res[waves] = np.load(path)
    assert sorted(res["4"].files) == sorted(res["8"].files)
    for k in res["4"].files:
        assert np.array_equal(res["4"][k], res["8"][k], equal_nan=True), k
    assert res["4"]["t64_good"].all()
