"""CPU study (numpy, uses the oracle decoder -- test infrastructure, not product): accuracy of evaluating the 8x512 decoder with
split-bf16 operands and f32 accumulation, the arithmetic a bf16-MFMA version of the MLP tile would do.
  bf2: x = hi + lo (two bf16 terms), products hi*hi + hi*lo + lo*hi  (3 MFMAs, 16x rate -> 5.3x the f32 MFMA)
  bf3: three terms, the six products with i + j < 3                  (6 MFMAs -> 2.7x the f32 MFMA)
Printed: error of the SDF value against float64 for 4000 points, next to plain float32.  DESIGN.md section 4 quotes the result."""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sdf_oracle as so
d = so.load_decoder_npz(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'decoder_8x512.npz'))
L = d.layers
print(type(L[0]), (L[0][0].shape, L[0][1].shape) if isinstance(L[0], (tuple, list)) else L[0])
Ws = [np.asarray(l[0], np.float64) for l in L]; bs = [np.asarray(l[1], np.float64) for l in L]
rng = np.random.default_rng(0)
N = 4000
x = rng.uniform(-0.6, 0.6, size=(N, 3)); code = np.zeros(64)
inp = np.concatenate([np.tile(code, (N, 1)), x], axis=1)

def bf16(a):
    a = np.asarray(a, np.float32)
    u = a.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)

def split(a, n):
    parts = []; rem = np.asarray(a, np.float32).copy()
    for _ in range(n):
        p = bf16(rem); parts.append(p); rem = (rem - p).astype(np.float32)
    return parts

def mm(A, B, mode):
    if mode == 'f64': return A.astype(np.float64) @ B.astype(np.float64)
    if mode == 'f32': return (A.astype(np.float32) @ B.astype(np.float32))
    n = int(mode[-1])
    Ap, Bp = split(A, n), split(B, n)
    acc = np.zeros((A.shape[0], B.shape[1]), np.float32)
    for i in range(n):
        for j in range(n):
            if i + j < n:                      # n=2: 3 products, n=3: 6 products
                acc += (Ap[i].astype(np.float32) @ Bp[j].astype(np.float32))
    return acc

def forward(mode):
    h = inp.copy()
    dt = np.float64 if mode == 'f64' else np.float32
    h = h.astype(dt)
    for l in range(9):
        if l == 4: h = np.concatenate([h, inp.astype(dt)], axis=1)
        z = mm(h, Ws[l].T.astype(dt), mode) + bs[l].astype(dt)
        h = np.maximum(z, 0) if l < 8 else np.tanh(z)
    return h[:, 0].astype(np.float64)
ref = forward('f64')
for m in ('f32', 'bf2', 'bf3'):
    y = forward(m)
    e = np.abs(y - ref)
    print(m, 'max abs err', e.max(), 'rel to max|y|', e.max() / np.abs(ref).max(), 'median', np.median(e))
