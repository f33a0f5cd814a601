"""profiling target: 20 calls of Optimizer.reconstruct_object (2000 surface points, 256+200 rays, 5 iterations) in the precision
given by argv[1] (f32 | bf16x3 | fp16x2) and the tile size argv[2] (64 | 32):
   rocprofv3 --kernel-trace --stats ... -- python3 tools/lat_profile.py fp16x2 32"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.reconstruct.optimizer import Optimizer
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests/golden/decoder_8x512.npz"))
dec.set_precision(sys.argv[1] if len(sys.argv) > 1 else "fp16x2")
if len(sys.argv) > 2:
    dec.set_tile_points(int(sys.argv[2]))          # 32: the latency option (fp16x2 only)
opt = Optimizer(dec, bench.joint_cfg(5))
o = synth.make_object_views(3003, 1, 2000, n_fg=256, n_bg=200)[0]
for _ in range(3):
    opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
t = time.perf_counter()
for _ in range(20):
    opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
print("%.2f ms per call" % (1e3 * (time.perf_counter() - t) / 20))
