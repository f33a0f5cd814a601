"""debug: narrow vs embedded form of the small decoder on a multi-hypothesis batch, one iteration, against the numpy oracle"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from oracle import sdf_oracle as so
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
from tests.test_gpu_sdf import make_cfg
gold = os.path.join(ROOT, "tests", "golden", "decoder_4x256_c32.npz")
dec = DeepSdfDecoder.from_npz(gold); dec.set_precision("fp16x2")
odec = so.load_decoder_npz(gold)
objs = synth.make_object_views(808, 4, 600, n_fg=100, n_bg=50)
T0, hyp = bench.flip_states(objs, 4)
cfg = so.JointConfig(n_iter=1)
for form in (True, False):
    dec.set_narrow_tile(form)
    opt = Optimizer(dec, make_cfg(cfg, code_len=32))
    b = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp)
    for n_it in (1, 2, 3):
        b.set_state(T0, None); b.run(n_it); tr = b.trace()
        print("narrow" if form else "embedded", "iters", n_it, "K", list(tr["K"]), "nv", list(tr["n_valid"][:6]))
    # single-hypothesis runs of hyp 9
    h = 9
    o = objs[hyp[h]]
    s1 = RefineBatch(dec, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0])
    s1.set_state(T0[h:h+1], None); s1.run(2); t1 = s1.trace(); s1.close()
    b.set_state(T0, None); b.run(2); tb = b.trace()
    print("   hyp 9 alone vs in batch: K", t1["K"][0], tb["K"][h], "H equal", np.array_equal(t1["H"][0], tb["H"][h]))
    b.close()
# oracle, one iteration, hyp 9
o = objs[hyp[9]]
T_oc = np.linalg.inv(T0[9].astype(np.float64)).astype(np.float32)
it = so.gn_iteration(odec, cfg, T_oc, np.zeros(32, np.float32), o["pts"], o["rays"], np.concatenate([o["depth"], np.zeros(50, np.float32)]), 100)
print("oracle hyp 9 it 1: K", it["K"], "n_valid", it["n_valid"])
