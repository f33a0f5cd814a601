"""Where k_sample and k_scan spend a one-object call's launches: ticks of the constant 100 MHz counter between the marks thread 0 of
workgroup 0 stamps (PHASE_MARK in csrc/sdf_kernels.hpp).  Needs build/exp/libqsp_phase.so:
   (here)     bash tools/phase_clock.sh        (GPU box)  python tools/phase_clock.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["QSP_HIP_LIB"] = os.path.join(ROOT, "build", "exp", "libqsp_phase.so")
import bench
from qsp_slam_amd import DeepSdfDecoder, synth, _lib
from qsp_slam_amd.reconstruct.optimizer import Optimizer
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
dec.set_precision("fp16x2"); dec.set_render_screening(0.01)
opt = Optimizer(dec, bench.joint_cfg(5))
o = synth.make_object_views(3003, 1, 2000, n_fg=256, n_bg=200)[0]
L = _lib.lib()
t = (C.c_ulonglong * 48)()
for _ in range(3):
    opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
L.qsp_debug_phase_ticks(t)
for _ in range(20):
    opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
L.qsp_debug_phase_ticks(t)
names = {0: ("k_sample", ["code", "bias vectors", "pose inverse", "ray masks", "offsets", "list", "state", "plan tail"]),
         2: ("k_solve", ["state, active points", "slot sums, H and b", "rotation prior, taps", "elimination", "update"]),
         1: ("k_scan", ["table clear", "table fill", "walk", "offsets", "write", "state", "plan tail"])}
for k, (name, marks) in names.items():
    n = max(t[16 * k], 1)
    parts = ["%s %.1f" % (m, t[16 * k + 1 + i] / n / 100.0) for i, m in enumerate(marks)]
    print("%s: %d launches, us between marks: %s; sum %.1f" % (name, n, ", ".join(parts), sum(t[16 * k + 1 + i] for i in range(len(marks))) / n / 100.0))
print("k_solve elimination: %.0f shader-clock cycles per launch (%.0f per column), shader clock %.2f GHz" % (
    t[32 + 8] / max(t[32], 1), t[32 + 8] / max(t[32], 1) / 71, t[32 + 8] / max(t[32 + 4], 1) * 0.1))
