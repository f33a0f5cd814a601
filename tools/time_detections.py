"""Times the C4-sized detection call (64 detections x 4 flips x 8 k points) through qsp_refine_detections -- world-frame
inputs up, marshalling + Gauss-Newton + keep rule on the device, kept results down -- beside (a) the same refinement fed with
host-assembled arrays through reconstruct_objects_batched and (b) the numpy restatement of the marshalling alone."""
import math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer
from oracle import detections_oracle as DO

w = bench.WORKLOADS["c4"]
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
opt = Optimizer(dec, bench.joint_cfg(w["n_iter"]))
dets = synth.make_detections(14, w["n_obj"], w["n_pts"], n_fg=w["n_fg"], n_bg=w["n_bg"], n_kf=5)
opt.refine_detections(dets[:2], 4)
n = 3
t = time.time()
for _ in range(n):
    res = opt.refine_detections(dets, 4)
t_dev = (time.time() - t) / n
t = time.time()
asm = [DO.assemble(d) for d in dets]
T0 = [DO.init_poses(d, 1, 2 * math.pi / 4)[0] for d in dets]
t_host_asm = time.time() - t
objs = [dict(t_cam_obj=T0[i], pts=a[0], rays=a[1], depth=a[2]) for i, a in enumerate(asm)]
opt.reconstruct_objects_batched(objs[:2], 4, True)
t = time.time()
for _ in range(n):
    best = opt.reconstruct_objects_batched(objs, 4, True)
t_fed = (time.time() - t) / n
iters = w["n_iter"]
print("C4 detections (%d x 4 flips x %d pts, %d iterations):" % (len(dets), w["n_pts"], iters))
print("  qsp_refine_detections, world-frame inputs -> kept results : %.1f ms per call" % (1e3 * t_dev))
print("  reconstruct_objects_batched on host-assembled arrays      : %.1f ms per call (+ %.1f ms numpy marshalling)" % (
    1e3 * t_fed, 1e3 * t_host_asm))
same = sum(int(r.kept_flip == 0 and np.array_equal(r.t_cam_obj, b.t_cam_obj)) for r, b in zip(res, best) if r.is_good)
print("  un-flipped hypothesis kept and identical to the host-fed result in %d of %d detections (flipped starts differ in the "
      "last bit: R_cw (R_wo R_y) vs (R_cw R_wo) R_y)" % (same, len(dets)))
