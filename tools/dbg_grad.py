import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import sdf_oracle as so
from qsp_slam_amd import DeepSdfDecoder
g = os.path.join(ROOT, "tests/golden/decoder_8x512.npz")
dec = DeepSdfDecoder.from_npz(g); od = so.load_decoder_npz(g)
rng = np.random.default_rng(0)
n = 70
x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
for code in (np.zeros(64, np.float32), rng.normal(scale=0.2, size=64).astype(np.float32)):
    inp = np.concatenate([np.broadcast_to(code, (n, 64)), x], -1)
    yr, gr = so.decoder_value_and_input_grad(od, inp)
    y, gg = dec.sdf_value_grad(code, x)
    d = np.abs(gg - gr).max(1) / np.abs(gr).max()
    print("code0" if not code.any() else "codeR", "bad rows:", np.where(d > 1e-5)[0][:20], "n_bad", (d > 1e-5).sum())
    dc = np.abs(gg - gr).max(0) / np.abs(gr).max()
    print("  bad cols:", np.where(dc > 1e-5)[0])
    print("  row0 err", d[0], "row1", d[1], gg[1, :4], gr[1, :4], gg[1, -3:], gr[1, -3:])
np.save(os.path.join(ROOT, "gpurun_out", "dbg_gg.npy"), gg); np.save(os.path.join(ROOT, "gpurun_out", "dbg_x.npy"), x); np.save(os.path.join(ROOT, "gpurun_out", "dbg_code.npy"), code)
