"""What the screening margin has to cover: |s1 - s3| between the one-product screening tile (k_mlp_fwd_h1 / qsp_decode_sdf_screen)
and the split-fp16 tile (qsp_decode_sdf on "fp16x2"), measured on the GPU over
  (a) random points of the cube at three code scales,
  (b) the ray samples the reference itself evaluated: every Gauss-Newton iteration of every golden joint case (tests/golden), the
      samples of all rays x 50 depths inside the unit ball under the reference's own pose of that iteration, with its code,
  (c) the ray samples of 120 random objects / poses / codes (the sizes of tools/parity_sweep.py).
Writes a table (max, 99.99 % quantile, share of samples with |s3| < cut_off + margin) to stdout; committed as
profiles/r03_screen_margin.txt.    python tools/screen_margin.py [margin]"""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from qsp_slam_amd import DeepSdfDecoder, synth

margin = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
TH, D = 0.01, 50
dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
dec.set_precision("fp16x2")
rows = []


def ray_samples(T_oc, rays):
    """object-frame positions of the depth samples inside the unit ball (reconstruct/loss.py:60-74), float32 like k_sample"""
    T_co = np.linalg.inv(T_oc.astype(np.float64)).astype(np.float32)
    scale = np.float32(np.linalg.det(T_co[:3, :3].astype(np.float64)) ** (1.0 / 3.0))
    d = np.linspace(T_co[2, 3] - scale, T_co[2, 3] + scale, D, dtype=np.float32)
    p = (rays[:, None, :] * d[None, :, None]).reshape(-1, 3).astype(np.float32)
    x = p @ T_oc[:3, :3].T + T_oc[:3, 3]
    return x[np.linalg.norm(x, axis=1) < 1.0].astype(np.float32)


def measure(name, code, x):
    s3 = dec.decode_sdf(code, x)
    s1 = dec.decode_sdf_screen(code, x)
    e = np.abs(s1 - s3)
    rows.append((name, len(x), float(e.max()), float(np.quantile(e, 0.9999)), float((np.abs(s3) < TH + margin).mean()),
                 float(e[np.abs(s3) < 5 * TH].max()) if (np.abs(s3) < 5 * TH).any() else 0.0))


rng = np.random.default_rng(2026)
for sc in (0.0, 0.05, 0.25):
    for rep in range(4):
        measure("cube, code scale %.2f" % sc, (sc * rng.normal(size=64)).astype(np.float32), rng.uniform(-1, 1, size=(65536, 3)).astype(np.float32))
for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "sdf_joint_*.npz"))):
    z = np.load(f)
    if "it_T_oc" not in z.files:
        continue
    for i in range(z["it_T_oc"].shape[0]):
        x = ray_samples(z["it_T_oc"][i], z["rays"])
        if len(x):
            measure("%s it %d" % (os.path.basename(f)[4:-4], i), z["it_code"][i], x)
for c in range(120):
    m, n_fg, n_bg = int(rng.integers(1, 900)), int(rng.integers(12, 160)), int(rng.integers(0, 60))
    o = synth.make_object_views(int(rng.integers(1, 10 ** 6)), 1, m, n_fg=n_fg, n_bg=n_bg, code_scale=float(rng.choice([0.0, 0.05])))[0]
    code = (0.05 * rng.normal(size=64)).astype(np.float32) if c % 3 == 0 else np.zeros(64, np.float32)
    x = ray_samples(np.linalg.inv(o["t_cam_obj"].astype(np.float64)).astype(np.float32), o["rays"])
    if len(x):
        measure("random object %3d" % c, code, x)

worst = max(r[2] for r in rows)
print("# |s1 - s3|: screening tile (one fp16 product) against the split-fp16 tile, golden decoder, MI355X")
print("# margin %.4f = %.1f x the largest difference below; cut_off %.3f" % (margin, margin / worst, TH))
print("%-36s %9s %12s %12s %14s %16s" % ("set", "samples", "max", "q99.99", "band share", "max, |s3|<5th"))
agg = {}
for name, n, mx, q, share, mxn in rows:
    key = name if not name.startswith("random object") else "random objects (120)"
    key = key if not name.startswith("cube") else name
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0, 0.0])
    a[3] = (a[3] * a[0] + share * n) / (a[0] + n)
    a[0] += n
    a[1] = max(a[1], mx)
    a[2] = max(a[2], q)
    a[4] = max(a[4], mxn)
for k, a in agg.items():
    print("%-36s %9d %12.3e %12.3e %13.2f%% %16.3e" % (k, a[0], a[1], a[2], 100 * a[3], a[4]))
print("%-36s %9d %12.3e" % ("ALL", sum(r[1] for r in rows), worst))
