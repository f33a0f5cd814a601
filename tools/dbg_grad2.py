import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import sdf_oracle as so
from qsp_slam_amd import DeepSdfDecoder
g = os.path.join(ROOT, "tests/golden/decoder_8x512.npz")
dec = DeepSdfDecoder.from_npz(g); od = so.load_decoder_npz(g)
rng = np.random.default_rng(0)
n = 64
x = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
code = rng.normal(scale=0.2, size=64).astype(np.float32)
inp = np.concatenate([np.broadcast_to(code, (n, 64)), x], -1)
y, masks = so.decoder_forward(od, inp, keep=True)
gg = ((1 - y * y)[:, None] * od.layers[-1][0][0][None, :]).astype(np.float32)
for l in range(7, -1, -1):
    gg = (gg * masks[l]) @ od.layers[l][0]
    if l == 4:
        skip = gg[:, 445:].copy(); gg = gg[:, :445]
g0 = gg
_, out = dec.sdf_value_grad(code, x)
for name, ref in (("skip", skip), ("g0", g0), ("full", skip + g0)):
    d = np.abs(out - ref).max(1) / np.abs(ref).max()
    print(name, "bad rows", np.where(d > 1e-5)[0][:40], (d > 1e-5).sum())
