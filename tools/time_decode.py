"""times qsp_decode_sdf / qsp_sdf_value_grad on n points with the library given by QSP_HIP_LIB"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import DeepSdfDecoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2 ** 20
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests/golden/decoder_8x512.npz"))
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32); code = np.zeros(64, np.float32)
for name, fn, flop in (("fwd", lambda: dec.decode_sdf(code, x), 3.671e6), ("fwdbwd", lambda: dec.sdf_value_grad(code, x), 7.342e6)):
    fn()
    ts = []
    for _ in range(3):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    t = min(ts)
    print("%s %s n=%d  %.2f ms  %.1f TFLOP/s (incl. H2D/D2H)" % (os.environ.get("QSP_HIP_LIB", "default")[-12:], name, n, 1e3 * t, flop * n / t / 1e12))
