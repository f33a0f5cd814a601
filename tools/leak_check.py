"""Calls every stateless entry point in a loop and prints the free device memory before / after (leak check)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from qsp_slam_amd import DeepSdfDecoder, synth
from qsp_slam_amd.ba import BaProblem, PoseOptimizer
from qsp_slam_amd.ellipsoid import optimize_ellipsoids_using_planes
from qsp_slam_amd.reconstruct.optimizer import Optimizer, MeshExtractor
from oracle import ellipsoid_oracle as EO
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2 ** 20
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
opt = Optimizer(dec, bench.joint_cfg(2))
dets = synth.make_detections(1, 3, 500, n_fg=64, n_bg=32)
objs = synth.make_object_views(1, 2, 500, n_fg=64, n_bg=32)
scene = synth.make_ba_scene(5, 6, 200, 2, stereo_frac=0.2)
pp = synth.make_pose_problem(1, n=200)
rng = np.random.default_rng(0)
ell = np.array([[0, 0, 3, 0, 0, 0, 1, 0.5, 0.4, 0.6]] * 8, float)
planes = [EO.tangent_planes(e, rng.normal(size=(9, 3))) for e in ell]
me = MeshExtractor(dec, 64, 32)
po = PoseOptimizer(512)
def once():
    opt.refine_detections(dets, 4)
    opt.reconstruct_objects_batched([dict(t_cam_obj=o["t_cam_obj"], pts=o["pts"], rays=o["rays"], depth=o["depth"]) for o in objs], 4, True)
    opt.estimate_pose_cam_obj(objs[0]["t_cam_obj"], 1.0, objs[0]["pts"], np.zeros(64, np.float32))
    dec.decode_sdf(np.zeros(64, np.float32), objs[0]["pts"])
    me.extract_mesh_from_code(np.zeros(64, np.float32))
    b = BaProblem(scene); b.local_joint_ba(); b.close()
    po.optimize(pp["K"], pp["pose"], pp["X"], pp["obs"], pp["info"], pp["stereo"])
    optimize_ellipsoids_using_planes(ell, planes)
for _ in range(3): once()
f0 = free_mb()
for _ in range(60): once()
f1 = free_mb()
print("free device memory before %.1f MiB, after 60 rounds of every entry point %.1f MiB (delta %.1f MiB)" % (f0, f1, f0 - f1))
