"""Prints the per-phase shader-clock deltas of one MLP tile (needs build/exp/libqsp_v16.so: tools/exp_variants.sh 16)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["QSP_HIP_LIB"] = os.path.join(ROOT, "build", "exp", "libqsp_v16.so")
from qsp_slam_amd import DeepSdfDecoder, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 20
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests/golden/decoder_8x512.npz"))
x = np.random.default_rng(0).uniform(-1, 1, size=(n, 3)).astype(np.float32); code = np.zeros(64, np.float32)
L = _lib.lib()
for name, fn in (("fwd+bwd", lambda: dec.sdf_value_grad(code, x)), ("fwd", lambda: dec.decode_sdf(code, x))):
    fn(); fn()
    ts = (C.c_ulonglong * 96)(); rt = (C.c_ulonglong * 96)(); cnt = C.c_int()
    L.qsp_debug_timestamps(ts, C.byref(cnt), rt)
    t = np.array(ts[:cnt.value], dtype=np.int64)
    r = np.array(rt[:cnt.value], dtype=np.int64)
    d = np.diff(t)
    print(name, "stamps", cnt.value, "total", t[-1] - t[0], "cycles in %.1f us -> shader clock %.3f GHz" % (
        (r[-1] - r[0]) / 100.0, (t[-1] - t[0]) / max(r[-1] - r[0], 1) * 0.1))
    print(" ".join("%d" % v for v in d))
