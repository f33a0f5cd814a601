"""Where the time of ONE online reconstruct_object call goes (create / set_state / run / get / destroy on the host; GPU spans from
the batch profile): 2000 surface points, 456 rays, 5 iterations."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
opt = Optimizer(dec, bench.joint_cfg(5))
o = synth.make_object_views(3, 1, 2000, n_fg=256, n_bg=200)[0]
for rep in range(3):
    t0 = time.time()
    b = RefineBatch(dec, _joint_cfg(opt), [o["pts"]], [o["rays"]], [o["depth"]], [0]); t1 = time.time()
    b.profile(True)
    b.set_state(o["t_cam_obj"][None], None); t2 = time.time()
    b.run(0); t3 = time.time()
    r = b.get(); t4 = time.time()
    p = b.profile(True)
    tr = b.trace()
    b.close(); t5 = time.time()
    print("create %.2f set %.2f run %.2f get %.2f close %.2f ms | gpu total %.2f jtj %.2f (%d) fwd %.2f (%d) other %.2f | n_valid %d K %d" % (
        1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4), p.ms_total, p.ms_mlp_jtj, p.n_launch_jtj, p.ms_mlp_fwd, p.n_launch_fwd, p.ms_other, tr["n_valid"][0], tr["K"][0]))
