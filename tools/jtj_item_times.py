"""Kernel-level phase stamps of k_mlp_jtj's first work item on workgroup 0 (needs build/exp/libqsp_v16.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["QSP_HIP_LIB"] = os.path.join(ROOT, "build", "exp", "libqsp_v16.so")
import numpy as np
import bench
from qsp_slam_amd import DeepSdfDecoder, synth, _lib
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
w = bench.WORKLOADS["c4"]
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
objs = synth.make_object_views(1000, 64, w["n_pts"], n_fg=w["n_fg"], n_bg=w["n_bg"])
opt = Optimizer(dec, bench.joint_cfg(1))
T0, hyp = bench.flip_states(objs, 4)
b = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs], [o["depth"] for o in objs], hyp)
b.set_state(T0, None); b.run(0)
ts = (C.c_ulonglong * 96)(); rt = (C.c_ulonglong * 96)(); cnt = C.c_int()
_lib.lib().qsp_debug_timestamps(ts, C.byref(cnt), rt)
t = np.array(ts[80:86], dtype=np.int64); r = np.array(rt[80:86], dtype=np.int64)
names = ["queue pop (atomic + barrier)", "item fetch + staging + xin", "mlp_tile", "Jt build + JtJ MFMA", "partial write"]
for i, nm in enumerate(names):
    print("%-32s %8d cycles  %7.2f us" % (nm, t[i + 1] - t[i], (r[i + 1] - r[i]) / 100.0))
print("item total %d cycles, %.1f us" % (t[5] - t[0], (r[5] - r[0]) / 100.0))
