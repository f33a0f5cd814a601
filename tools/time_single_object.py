"""Latency of ONE online call, as the reference makes them (src/LocalMapping_util.cc:705-760): reconstruct_object for one
hypothesis, the four flips as one batched call, and the world-frame detection call; 2 k and 8 k surface points."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
opt = Optimizer(dec, bench.joint_cfg(5))
def timeit(f, n=10):
    f(); t = time.time()
    for _ in range(n): f()
    return 1e3 * (time.time() - t) / n
for m in (2000, 8000):
    o = synth.make_object_views(3, 1, m, n_fg=256, n_bg=200)[0]
    d = synth.make_detections(3, 1, m, n_fg=256, n_bg=200)[0]
    a = timeit(lambda: opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"]))
    b = timeit(lambda: opt.reconstruct_objects_batched([dict(t_cam_obj=o["t_cam_obj"], pts=o["pts"], rays=o["rays"], depth=o["depth"])], 4, True))
    c = timeit(lambda: opt.refine_detections([d], 4))
    print("%d surface points, 456 rays, 5 iterations: reconstruct_object %.2f ms; 4 flips batched %.2f ms; refine_detections (4 flips) %.2f ms" % (m, a, b, c))
