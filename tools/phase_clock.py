"""Shader clock and tile cycles of the stamped variant builds (16 = as shipped, 17 = no weight loads, 18 = no LDS operand reads, 19 = neither)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    os.environ["QSP_HIP_LIB"] = os.path.join(ROOT, "build", "exp", "libqsp_v%s.so" % sys.argv[1])
    import numpy as np
    from qsp_slam_amd import DeepSdfDecoder, _lib
    n = 2 ** 21
    dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests/golden/decoder_8x512.npz"))
    x = np.random.default_rng(0).uniform(-1, 1, size=(n, 3)).astype(np.float32); code = np.zeros(64, np.float32)
    L = _lib.lib()
    for name, fn in (("fwd+bwd", lambda: dec.sdf_value_grad(code, x)), ("fwd", lambda: dec.decode_sdf(code, x)), ("fwd+bwd", lambda: dec.sdf_value_grad(code, x))):
        fn(); fn(); fn()
        ts = (C.c_ulonglong * 96)(); rt = (C.c_ulonglong * 96)(); cnt = C.c_int()
        L.qsp_debug_timestamps(ts, C.byref(cnt), rt)
        t = np.array(ts[:cnt.value], dtype=np.int64); r = np.array(rt[:cnt.value], dtype=np.int64)
        print("variant %s %-8s tile %8d cycles, %.1f us, clock %.3f GHz" % (sys.argv[1], name, t[-1] - t[0], (r[-1] - r[0]) / 100.0, (t[-1] - t[0]) / max(r[-1] - r[0], 1) * 0.1))
else:
    for v in ("16", "17", "18", "19", "16"):
        subprocess.run([sys.executable, os.path.abspath(__file__), v], check=True)
