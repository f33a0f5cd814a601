"""Runs only the refinement batch of a bench workload for a few iterations (short enough for rocprofv3 --pmc):
python tools/refine_only.py c5 [n_obj] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
w = bench.WORKLOADS[name]
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else w["n_obj"]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
# (QSP_PRECISION=fp16x2 QSP_SCREENING=0.01 in the environment select the bench default: DeepSdfDecoder reads them itself)
objs = synth.make_object_views(1000, n_obj, w["n_pts"], n_fg=w["n_fg"], n_bg=w["n_bg"])
opt = Optimizer(dec, bench.joint_cfg(iters))
T0, hyp = bench.flip_states(objs, 4)
b = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp)
b.profile(True)
for rep in range(2):
    b.set_state(T0, None)
    t = time.time(); b.run(0); dt = time.time() - t
    p = b.profile(True)
    print("%s %d hyps x %d it: %.1f ms; jtj %.2f ms/launch (%.0f pts, %.0f tiles), fwd %.2f ms/launch (%.0f pts)" % (
        name, len(hyp), iters, 1e3 * dt, p.ms_mlp_jtj / max(p.n_launch_jtj, 1), p.pts_jtj / max(p.n_launch_jtj, 1),
        p.tiles_jtj / max(p.n_launch_jtj, 1), p.ms_mlp_fwd / max(p.n_launch_fwd, 1), p.pts_fwd / max(p.n_launch_fwd, 1)))
