"""The reference's call pattern on the resident batch: N calls of Optimizer.reconstruct_object (one object, 2000 surface points,
456 rays, 5 iterations) and of the four-flip batched call; prints ms per call.  Under rocprofv3 --kernel-trace --stats the
kernel table shows the per-launch times of one tile-deep kernels.   python tools/lat_calls.py [precision] [tile_points]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
dec.set_precision(prec)
if prec == "fp16x2":
    dec.set_render_screening(0.01)
    if len(sys.argv) > 2:
        dec.set_tile_points(int(sys.argv[2]))
opt = Optimizer(dec, bench.joint_cfg(5))
o = synth.make_object_views(3003, 1, 2000, n_fg=256, n_bg=200)[0]
obj = dict(t_cam_obj=o["t_cam_obj"], pts=o["pts"], rays=o["rays"], depth=o["depth"])
for _ in range(3):
    opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
n = 20
t0 = time.perf_counter()
for _ in range(n):
    opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
one = 1e3 * (time.perf_counter() - t0) / n
opt.reconstruct_objects_batched([obj], flip_sample_num=4)
t0 = time.perf_counter()
for _ in range(n):
    opt.reconstruct_objects_batched([obj], flip_sample_num=4)
four = 1e3 * (time.perf_counter() - t0) / n
print("%s tile %s: reconstruct_object %.3f ms, four flips in one call %.3f ms, arena (reused, created) %r" % (
    prec, sys.argv[2] if len(sys.argv) > 2 else "64", one, four, dec.arena_stats))
