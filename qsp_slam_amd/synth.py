"""Synthetic-scene generator for the joint object-optimisation hot path.

Data generation only: seeded numpy, no reference code, no oracle.  Used by bench.py, the tests and the
oracle's fixture generators so that every consumer sees byte-identical inputs for a given seed.

Two families of inputs:
  * object refinement (path A): per object a Sim(3) pose T_co, M surface points in the camera frame,
    n_fg foreground rays with observed depth, n_bg background rays (reference producers:
    src/LocalMapping_util.cc:585-672, reconstruct/mono_sequence.py:142-144)
  * bundle adjustment (path B): key-frames on an arc, map points with mono/stereo observations,
    objects with per-key-frame SE3 detections (reference consumers: src/Optimizer_util.cc:309-771)

The analytic shape family below is what the test decoder (tests/golden/decoder_8x512.npz) was fitted to:
an axis-aligned ellipsoid whose semi-axes depend on the first three latent dimensions, united with a
small sphere on +x that breaks the yaw symmetry (so the four yaw-flip hypotheses of
src/LocalMapping_util.cc:713-760 have different losses).
"""
import numpy as np

BUMP_RADIUS = 0.18


def shape_axes(code):
    """semi-axes a(z) = 0.45 * exp(0.5 * z[0:3]) * (1.0, 0.8, 0.6); code (..., >=3)"""
    code = np.asarray(code, dtype=np.float64)
    return 0.45 * np.exp(0.5 * code[..., 0:3]) * np.array([1.0, 0.8, 0.6])


def analytic_sdf(x, code):
    """approximate signed distance of the shape family; x (..., 3), code broadcastable (..., >=3)"""
    x = np.asarray(x, dtype=np.float64)
    a = shape_axes(code)
    k0 = np.linalg.norm(x / a, axis=-1)
    k1 = np.linalg.norm(x / (a * a), axis=-1)
    d_ell = k0 * (k0 - 1.0) / np.maximum(k1, 1e-9)
    c = np.stack([0.9 * a[..., 0], 0.35 * a[..., 1], np.zeros_like(a[..., 0])], axis=-1)
    d_sph = np.linalg.norm(x - c, axis=-1) - BUMP_RADIUS
    return np.minimum(d_ell, d_sph)


def sample_surface(rng, n, code, noise=0.0):
    """n points close to the zero level set of analytic_sdf (object frame), by rejection + projection"""
    a = shape_axes(code)
    out = np.zeros((0, 3))
    while out.shape[0] < n:
        m = 2 * (n - out.shape[0]) + 16
        u = rng.normal(size=(m, 3))
        u /= np.linalg.norm(u, axis=-1, keepdims=True)
        pick_bump = rng.random(m) < 0.12
        c = np.array([0.9 * a[0], 0.35 * a[1], 0.0])
        p = np.where(pick_bump[:, None], c + BUMP_RADIUS * u, a * u)
        # two Newton-like projection steps along the numeric gradient
        for _ in range(3):
            d = analytic_sdf(p, code)
            g = np.stack([(analytic_sdf(p + e, code) - analytic_sdf(p - e, code)) / 2e-4
                          for e in 1e-4 * np.eye(3)], axis=-1)
            p = p - d[:, None] * g / np.maximum((g * g).sum(-1, keepdims=True), 1e-9)
        keep = np.abs(analytic_sdf(p, code)) < 2e-3
        out = np.concatenate([out, p[keep]], axis=0)
    out = out[:n]
    if noise > 0:
        out = out + rng.normal(scale=noise, size=out.shape)
    return out


# ----------------------------------------------------------------------------------------------------------
# small Lie-group helpers (float64) used only to build scenes
# ----------------------------------------------------------------------------------------------------------

def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = hat(w / th)
    return np.eye(3) + np.sin(th) * k + (1 - np.cos(th)) * (k @ k)


def se3(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


# ----------------------------------------------------------------------------------------------------------
# path A: object refinement inputs
# ----------------------------------------------------------------------------------------------------------

def make_object_views(seed, n_obj, n_pts, n_fg=256, n_bg=200, code_scale=0.0, perturb=True):
    """Per object: the inputs of Optimizer.reconstruct_object (reference reconstruct/optimizer.py:96-103).

    returns a list of dicts with float32 arrays
       t_cam_obj (4,4) initial Sim3 object->camera, pts (M,3) camera frame, rays (n_fg+n_bg,3),
       depth (n_fg,), and ground truth gt_t_cam_obj, gt_code (64,)
    """
    rng = np.random.default_rng(seed)
    objs = []
    for _ in range(n_obj):
        code = np.zeros(64)
        if code_scale > 0:
            code[:3] = rng.normal(scale=code_scale, size=3)
        scale = rng.uniform(0.6, 1.2)
        yaw = rng.uniform(0, 2 * np.pi)
        # object y axis points to camera -y ("up"), reference reconstruct/loss.py:167-171
        R_co = np.diag([1.0, -1.0, -1.0]) @ rot_y(yaw)
        t_co = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.3, 0.3), rng.uniform(2.5, 4.5)])
        T_gt = se3(scale * R_co, t_co)
        # surface points, camera frame
        p_obj = sample_surface(rng, n_pts, code, noise=0.002)
        pts = (T_gt[:3, :3] @ p_obj.T).T + t_co
        # foreground rays: through surface points that face the camera
        cand = sample_surface(rng, 6 * n_fg, code)
        cand_cam = (T_gt[:3, :3] @ cand.T).T + t_co
        # visibility: keep the nearest candidate per coarse angular bin
        dirs = cand_cam / cand_cam[:, 2:3]
        bins = np.floor(dirs[:, :2] / (0.02 * scale / t_co[2])).astype(np.int64)
        key = bins[:, 0] * 100003 + bins[:, 1]
        order = np.lexsort((cand_cam[:, 2], key))
        first = np.ones(len(order), bool)
        first[1:] = key[order][1:] != key[order][:-1]
        vis = order[first]
        rng.shuffle(vis)
        vis = vis[:n_fg]
        fg_rays = dirs[vis]
        depth = cand_cam[vis, 2]
        # background rays: in the 2-D box around the object but off the silhouette
        ext = 1.25 * scale * 0.45 / t_co[2]
        centre = t_co[:2] / t_co[2]
        bg = []
        while len(bg) < n_bg:
            uv = centre + rng.uniform(-2.2 * ext, 2.2 * ext, size=2)
            ray = np.array([uv[0], uv[1], 1.0])
            # march the ray in the object frame, reject if it hits the shape
            ds = np.linspace(t_co[2] - 1.2 * scale, t_co[2] + 1.2 * scale, 96)
            P = ray[None, :] * ds[:, None]
            Po = (np.linalg.inv(T_gt) @ np.c_[P, np.ones(len(P))].T).T[:, :3]
            if analytic_sdf(Po, code).min() > 0.02:
                bg.append(ray)
        rays = np.concatenate([fg_rays, np.array(bg).reshape(-1, 3)], axis=0)
        if perturb:
            d = np.concatenate([rng.normal(scale=0.03, size=3), rng.normal(scale=np.deg2rad(3.0), size=3)])
            T0 = se3(rodrigues(d[3:]), d[:3]) @ T_gt
            T0[:3, :3] *= np.exp(rng.normal(scale=0.03))
        else:
            T0 = T_gt.copy()
        objs.append(dict(t_cam_obj=T0.astype(np.float32), pts=pts.astype(np.float32),
                         rays=rays.astype(np.float32), depth=depth.astype(np.float32),
                         gt_t_cam_obj=T_gt.astype(np.float32), gt_code=code.astype(np.float32)))
    return objs
