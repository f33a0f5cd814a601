"""CPU study (numpy, uses the oracle decoder -- test infrastructure, not product): accuracy of evaluating the 8x512 decoder
(forward value AND input gradient) with operands as TWO fp16 terms, x = hi + lo' * 2^-11 (lo' = (x - hi) * 2^11, so that both
terms sit in fp16's normal range whatever x is), f32 accumulation, against float64:
  h2p3: products hi*hi | hi*lo' + lo'*hi  (3 fp16 MFMAs, the cross terms in a second accumulator scaled by 2^-11 at the end)
  h2p4: + lo'*lo' in a third accumulator (2^-22)
next to plain float32 and the three-term bf16 split (6 products) that is shipped.  Also counts how many of the render term's
threshold decisions (|sdf| < 0.01) differ from the float64 evaluation."""
import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sdf_oracle as so
d = so.load_decoder_npz(os.path.join(ROOT, 'tests', 'golden', 'decoder_8x512.npz'))
Ws = [np.asarray(l[0], np.float64) for l in d.layers]; bs = [np.asarray(l[1], np.float64) for l in d.layers]
rng = np.random.default_rng(0)
N = 20000
x = rng.uniform(-0.6, 0.6, size=(N, 3)); code = 0.1 * rng.normal(size=64)
inp = np.concatenate([np.tile(code, (N, 1)), x], axis=1)


def bf16(a):
    a = np.asarray(a, np.float32); u = a.view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) & 0xFFFF0000).view(np.float32)


def split_bf(a, n):
    parts = []; rem = np.asarray(a, np.float32).copy()
    for _ in range(n):
        p = bf16(rem); parts.append(p); rem = (rem - p).astype(np.float32)
    return parts


def split_h2(a):
    a = np.asarray(a, np.float32)
    hi = a.astype(np.float16).astype(np.float32)
    lo = ((a - hi) * np.float32(2048.0)).astype(np.float16).astype(np.float32)
    return hi, lo


def mm(A, B, mode):
    if mode == 'f64': return A.astype(np.float64) @ B.astype(np.float64)
    if mode == 'f32': return A.astype(np.float32) @ B.astype(np.float32)
    if mode == 'bf3':
        Ap, Bp = split_bf(A, 3), split_bf(B, 3)
        acc = np.zeros((A.shape[0], B.shape[1]), np.float32)
        for i in range(3):
            for j in range(3):
                if i + j < 3: acc += Ap[i] @ Bp[j]
        return acc
    ah, al = split_h2(A); bh, bl = split_h2(B)
    acc = ah @ bh
    acc2 = ah @ bl + al @ bh
    out = acc + acc2 * np.float32(2.0 ** -11)
    if mode == 'h2p4': out = out + (al @ bl) * np.float32(2.0 ** -22)
    return out.astype(np.float32)


def forward_backward(mode):
    dt = np.float64 if mode == 'f64' else np.float32
    h = inp.astype(dt); x0 = h
    pre = []
    for l in range(9):
        if l == 4: h = np.concatenate([h, x0], axis=1)
        z = mm(h, Ws[l].T.astype(dt), mode) + bs[l].astype(dt)
        pre.append(z)
        h = np.maximum(z, 0) if l < 8 else np.tanh(z)
    y = h[:, 0]
    g = ((1 - y * y)[:, None] * Ws[8].astype(dt)).astype(dt)        # (N, 512)
    gin = np.zeros_like(x0)
    for l in range(7, -1, -1):
        g = g * (pre[l] > 0)
        g = mm(g, Ws[l].astype(dt), mode)
        if l == 4:
            gin += g[:, -67:]; g = g[:, :-67]
    gin += g
    return y.astype(np.float64), gin.astype(np.float64)


yr, gr = forward_backward('f64')
for m in ('f32', 'bf3', 'h2p3', 'h2p4'):
    y, g = forward_backward(m)
    ey = np.abs(y - yr).max() / np.abs(yr).max()
    same_mask = np.abs(g - gr).max(1) / np.abs(gr).max() < 1e-3                     # rows not on a ReLU knife edge
    eg = (np.abs(g - gr).max(1) / np.abs(gr).max())[same_mask].max()
    flips = int(((np.abs(y) < 0.01) != (np.abs(yr) < 0.01)).sum())
    print("%-5s sdf: max rel err %.2e  median abs %.2e | grad rows (non-knife-edge): max rel err %.2e | threshold decisions "
          "differing from f64: %d of %d" % (m, ey, np.median(np.abs(y - yr)), eg, flips, N))
