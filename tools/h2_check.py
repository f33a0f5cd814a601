"""Split-fp16 tile (precision "fp16x2") against the f32 tile, the split-bf16 tile and float64: forward values, gradients
(sdf_value_grad), one Gauss-Newton iteration (H, b, K) and a whole refinement, then timing of the two MLP kernels on a C4-like
batch.   python tools/h2_check.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import sdf_oracle as so
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict
gold = os.path.join(ROOT, "tests/golden/decoder_8x512.npz")
dec = DeepSdfDecoder.from_npz(gold)
odec = so.load_decoder_npz(gold)
rng = np.random.default_rng(0)
n = 20000
x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
code = (0.2 * rng.normal(size=64)).astype(np.float32)
res = {}
for p in ("f32", "bf16x3", "fp16x2"):
    dec.set_precision(p)
    y = dec.decode_sdf(code, x)
    y2, g = dec.sdf_value_grad(code, x)
    res[p] = (y, y2, g)
y0, _, g0 = res["f32"]
for p in ("bf16x3", "fp16x2"):
    y, y2, g = res[p]
    print("%s: max|y - y_f32| %.3e  (value_grad's y: %.3e)   max|g - g_f32| / max|g| %.3e" % (
        p, np.abs(y - y0).max(), np.abs(y2 - y0).max(), np.abs(g - g0).max() / np.abs(g0).max()))
# float64 gradient by the oracle on a subset
sub = slice(0, 256)
inp = np.concatenate([np.broadcast_to(code, (256, 64)), x[sub]], -1).astype(np.float32)
yo, go = so.decoder_value_and_input_grad(odec, inp)
yo = np.asarray(yo).reshape(-1)
if True:
    for p in ("f32", "bf16x3", "fp16x2"):
        print("%s vs oracle: y %.3e  g rel %.3e" % (p, np.abs(res[p][0][sub] - yo).max(), np.abs(res[p][2][sub] - go).max() / np.abs(go).max()))

cfg = so.JointConfig()
conf = ForceKeyErrorDict(data_type="Redwood", optimizer=dict(
    code_len=64, num_depth_samples=50, cut_off_threshold=0.01,
    joint_optim=dict(k1=cfg.k1, k2=cfg.k2, k3=cfg.k3, k4=cfg.k4, b1=cfg.b1, b2=cfg.b2, learning_rate=cfg.lr,
                     scale_damping=cfg.s_damp, num_iterations=cfg.n_iter)))
objs = synth.make_object_views(3, 6, 500, n_fg=96, n_bg=48)
opt = Optimizer(dec, conf)
tr = {}
for p in ("f32", "bf16x3", "fp16x2"):
    dec.set_precision(p)
    batch = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs],
                        list(range(len(objs))))
    batch.set_state(np.stack([o["t_cam_obj"] for o in objs]), None)
    batch.run(1)
    tr[p] = batch.trace()
    batch.close()
for p in ("bf16x3", "fp16x2"):
    H0, b0 = tr["f32"]["H"], tr["f32"]["b"]
    print("%s one iteration: K equal %s   relerr(H) %.3e  relerr(b) %.3e" % (
        p, np.array_equal(tr[p]["K"], tr["f32"]["K"]), np.abs(tr[p]["H"] - H0).max() / np.abs(H0).max(),
        np.abs(tr[p]["b"] - b0).max() / np.abs(b0).max()))
# timing: 64 objects x 4 hypotheses, 8k points (C4-like) through the batch profile
objs = synth.make_object_views(5, 64, 8192, n_fg=1024, n_bg=512)
hyp_obj = [i for i in range(64) for _ in range(4)]
T0 = np.stack([objs[i]["t_cam_obj"] for i in hyp_obj])
for p in ("f32", "bf16x3", "fp16x2"):
    dec.set_precision(p)
    batch = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp_obj)
    batch.set_state(T0, None)
    batch.run(2)
    batch.set_state(T0, None)
    batch.profile(True)
    t = time.perf_counter(); batch.run(10); dt = time.perf_counter() - t
    pr = batch.profile(True)
    print("%s: 10 iterations %.1f ms   k_mlp_jtj %.2f ms/launch   k_mlp_fwd %.2f ms/launch" % (
        p, 1e3 * dt, pr.ms_mlp_jtj / max(pr.n_launch_jtj, 1), pr.ms_mlp_fwd / max(pr.n_launch_fwd, 1)))
    batch.close()
