"""Split-bf16 forward tile (QSP_DEC_OPT_FORWARD_PRECISION = 1) against the f32 tile and a float64 evaluation of the same
network: accuracy on random points and the golden vectors, and the kernel rate through the mesh extractor's resident grid
(no host transfers in the timed region).   python tools/bf3_check.py [grid_dim]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import sdf_oracle as so
from qsp_slam_amd import DeepSdfDecoder
from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
gold = os.path.join(ROOT, "tests/golden/decoder_8x512.npz")
dec = DeepSdfDecoder.from_npz(gold)
odec = so.load_decoder_npz(gold)
rng = np.random.default_rng(0)
n = 20000
x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
code = (0.2 * rng.normal(size=64)).astype(np.float32)


def f64_forward(dec_w, code, x):
    inp = np.concatenate([np.broadcast_to(code.astype(np.float64), (x.shape[0], 64)), x.astype(np.float64)], -1)
    h = inp
    nl = len(dec_w.layers)
    for l, (W, b) in enumerate(dec_w.layers):
        if l in dec_w.latent_in:
            h = np.concatenate([h, inp], -1)
        a = h @ W.astype(np.float64).T + b.astype(np.float64)
        h = np.maximum(a, 0) if l < nl - 1 else a
    return np.tanh(h[:, 0])


ref = f64_forward(odec, code, x)
y32 = dec.decode_sdf(code, x)
dec.set_forward_precision(True)
y3 = dec.decode_sdf(code, x)
print("max |f32 - f64|   = %.3e   rel to max|y| %.3e" % (np.abs(y32 - ref).max(), np.abs(y32 - ref).max() / np.abs(ref).max()))
print("max |bf16x3 - f64| = %.3e   rel to max|y| %.3e" % (np.abs(y3 - ref).max(), np.abs(y3 - ref).max() / np.abs(ref).max()))
print("max |bf16x3 - f32| = %.3e ; rms %.3e" % (np.abs(y3 - y32).max(), np.sqrt(np.mean((y3 - y32) ** 2))))
z = np.load(os.path.join(ROOT, "tests/golden/sdf_decoder_vectors.npz"))
print("golden vectors: max |bf16x3 - reference| = %.3e" % np.abs(dec.decode_sdf(z["code"], z["x"]) - z["sdf"]).max())
dec.set_forward_precision(2)
yh = dec.decode_sdf(code, x)
print("max |fp16x2 - f64| = %.3e   rel to max|y| %.3e" % (np.abs(yh - ref).max(), np.abs(yh - ref).max() / np.abs(ref).max()))
print("max |fp16x2 - f32| = %.3e ; rms %.3e" % (np.abs(yh - y32).max(), np.sqrt(np.mean((yh - y32) ** 2))))
print("golden vectors: max |fp16x2 - reference| = %.3e" % np.abs(dec.decode_sdf(z["code"], z["x"]) - z["sdf"]).max())
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for mode in (0, 1, 2):
    dec.set_forward_precision(mode)
    me = MeshExtractor(dec, 64, dim)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        me.extract_mesh_from_code(code)
        ts = []
        for _ in range(5):
            t = time.perf_counter(); me.extract_mesh_from_code(code); ts.append(time.perf_counter() - t)
    t = min(ts)
    print("%s: %d^3 grid decode + marching cubes %.2f ms  -> >= %.1f TFLOP/s of decoder work (3.671 MFLOP/voxel)" % (
        ("f32 MFMA  ", "split-bf16", "split-fp16")[mode], dim, 1e3 * t, 3.671e6 * dim ** 3 / t / 1e12))
