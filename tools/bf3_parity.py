"""Parity of the split-bf16 tiles (forward-only and forward+backward) on every golden case, teacher-forced, against the
reference-generated fixtures -- the same quantities as tools/parity_report.py -- and the step time of a bench workload with
either precision.   python tools/bf3_parity.py [c2|c4|c5]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
from tests.test_gpu_sdf import make_cfg
from tests.test_oracle_sdf import JOINT_CASES, cfg_from, relerr
GOLD = os.path.join(ROOT, "tests", "golden")
dec = DeepSdfDecoder.from_npz(os.path.join(GOLD, "decoder_8x512.npz"))
for fwd, jac in ((0, 0), (1, 0), (0, 1), (1, 1)):
    dec.set_forward_precision(bool(fwd)); dec.set_jacobian_precision(bool(jac))
    worst = dict(H=0.0, b=0.0, dx=0.0, T=0.0, code=0.0); k_ok = True
    for name in JOINT_CASES:
        z = np.load(os.path.join(GOLD, name + ".npz"))
        opt = Optimizer(dec, make_cfg(z))
        batch = RefineBatch(dec, _joint_cfg(opt), [z["pts"]], [z["rays"]], [z["depth"]], [0])
        n_it = z["it_H"].shape[0]
        for i in range(n_it):
            T_co = np.linalg.inv(z["it_T_oc"][i].astype(np.float64)).astype(np.float32)
            batch.set_state(T_co[None], z["it_code"][i][None]); batch.run(1)
            tr = batch.trace(); T, code, loss, good = batch.get()
            k_ok &= int(tr["K"][0]) == int(z["it_K"][i])
            if cfg_from(z).k4 == 0:
                worst["H"] = max(worst["H"], relerr(tr["H"][0], z["it_H"][i])); worst["b"] = max(worst["b"], relerr(tr["b"][0], z["it_b"][i]))
                worst["dx"] = max(worst["dx"], relerr(tr["dx"][0], z["it_dx"][i]))
                if i + 1 < n_it:
                    worst["T"] = max(worst["T"], relerr(np.linalg.inv(T[0].astype(np.float64)), z["it_T_oc"][i + 1]))
                    worst["code"] = max(worst["code"], float(np.abs(code[0] - z["it_code"][i + 1]).max()))
        batch.close()
    print("forward %s  jacobian %s : K exact on all fixtures %s; worst (k4 = 0 cases) %s" % (
        "bf16x3" if fwd else "f32   ", "bf16x3" if jac else "f32   ", k_ok, {k: "%.1e" % v for k, v in worst.items()}))
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
objs = synth.make_object_views(1000, w["n_obj"], w["n_pts"], n_fg=w["n_fg"], n_bg=w["n_bg"])
opt = Optimizer(dec, bench.joint_cfg(w["n_iter"]))
T0, hyp = bench.flip_states(objs, 4)
batch = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp)
batch.profile(True)
res = {}
for fwd, jac in ((0, 0), (1, 1)):
    dec.set_forward_precision(bool(fwd)); dec.set_jacobian_precision(bool(jac))
    for _ in range(2):
        batch.set_state(T0, None); t = time.perf_counter(); batch.run(0); dt = time.perf_counter() - t
    p = batch.profile(True)
    T, code, loss, good = batch.get()
    res[(fwd, jac)] = (T, code, loss, good)
    print("%s: step %.1f ms; k_mlp_jtj %.1f ms (%.0f TFLOP/s eff.), k_mlp_fwd %.1f ms (%.0f TFLOP/s eff.), good %d" % (
        "split-bf16" if fwd else "f32 MFMA ", 1e3 * dt, p.ms_mlp_jtj, (bench.FLOP_FWDBWD + 2 * 72 * 72) * p.pts_jtj / p.ms_mlp_jtj / 1e9,
        p.ms_mlp_fwd, bench.FLOP_FWD * p.pts_fwd / p.ms_mlp_fwd / 1e9, int(good.sum())))
a, b = res[(0, 0)], res[(1, 1)]
g = a[3] & b[3]
print("free-running result, split-bf16 vs f32: good flags equal %s; max rel diff T %.2e, code %.2e, loss %.2e" % (
    bool((a[3] == b[3]).all()), np.abs(a[0][g] - b[0][g]).max() / np.abs(a[0][g]).max(), np.abs(a[1][g] - b[1][g]).max(),
    np.abs(a[2][g] / b[2][g] - 1).max()))
