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


def make_detections(seed, n_det, n_pts, n_fg=256, n_bg=200, n_kf=3, code_scale=0.0):
    """World-frame inputs of LocalMapping::ProcessDetectedObjects' reconstruction block (src/LocalMapping_util.cc:585-706)
    for `n_det` detections spread over `n_kf` key frames: the object views of make_object_views moved into a world frame.

    returns a list of dicts: T_cw (4,4), K (4,) fx fy cx cy, T_wo (4,4) Sim3Two, code (64,), pts_world (M,3),
    fg_px (F,2), fg_world (F,3), bg_rays (B,3) -- float32 -- plus gt_t_cam_obj / gt_code."""
    rng = np.random.default_rng(seed + 7919)
    views = make_object_views(seed, n_det, n_pts, n_fg=n_fg, n_bg=n_bg, code_scale=code_scale)
    K = np.array([535.4, 539.2, 320.1, 247.6])         # configs/tum_fr1_desk.yaml
    kfs = []
    for _ in range(n_kf):
        w = rng.normal(scale=0.4, size=3)
        kfs.append(se3(rodrigues(w), rng.uniform(-2.0, 2.0, size=3)))
    dets = []
    for i, v in enumerate(views):
        T_cw = kfs[i % n_kf]
        T_wc = np.linalg.inv(T_cw)

        def to_world(x):
            return (T_wc[:3, :3] @ np.asarray(x, np.float64).T).T + T_wc[:3, 3]

        n_f = len(v["depth"])
        fg_cam = v["rays"][:n_f].astype(np.float64) * v["depth"][:, None].astype(np.float64)
        px = np.stack([K[0] * v["rays"][:n_f, 0] + K[2], K[1] * v["rays"][:n_f, 1] + K[3]], axis=1)
        dets.append(dict(T_cw=T_cw.astype(np.float32), K=K.astype(np.float32),
                         T_wo=(T_wc @ v["t_cam_obj"].astype(np.float64)).astype(np.float32),
                         code=np.zeros(64, np.float32), pts_world=to_world(v["pts"]).astype(np.float32),
                         fg_px=px.astype(np.float32), fg_world=to_world(fg_cam).astype(np.float32),
                         bg_rays=v["rays"][n_f:].copy(), gt_t_cam_obj=v["gt_t_cam_obj"], gt_code=v["gt_code"]))
    return dets


# ----------------------------------------------------------------------------------------------------------
# path B: joint bundle-adjustment scenes (flattened graph of src/Optimizer_util.cc:309-771)
# ----------------------------------------------------------------------------------------------------------

def quat_from_R(R):
    """x y z w, w >= 0"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[3] = (R[k, j] - R[j, k]) / s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def pose7(T):
    """4x4 -> (tx ty tz qx qy qz qw), the flattened SE3Quat layout of include/qsp_hip.h"""
    return np.concatenate([T[:3, 3], quat_from_R(T[:3, :3])])


def pose7_to_T(p):
    x, y, z, w = p[3:7]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return se3(R, p[:3])


def make_ba_scene(seed, n_kf, n_pt, n_obj, stereo_frac=0.0, outlier_frac=0.05, n_fixed=1, obs_per_obj=10,
                  pixel_noise=1.0):
    """Seeded scene in the flattened layout of the C-ABI: key-frames on an arc looking inwards,
    points in the common frustum, each observed by the key-frames that see it, objects observed by ~obs_per_obj
    key-frames through noisy SE3 detections.  Estimates are perturbed around the ground truth; `outlier_frac` of the
    observations get a gross 50 px error.  All float64; measurements pass through float32 as in src/Converter.cc.

    returns dict of arrays (see qsp_ba_scene in include/qsp_hip.h for the field meaning)"""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 535.4, 539.2, 320.1, 247.6          # configs/tum_fr1_desk.yaml
    bf = 40.0
    W, H = 640, 480
    # key-frames on an arc of radius 4 m around the origin, 0.3 m apart, looking at the centre
    kf_T = []
    for i in range(n_kf):
        a = (i - n_kf / 2) * 0.3 / 4.0
        c = np.array([4.0 * np.sin(a), 0.1 * np.sin(0.7 * i), -4.0 * np.cos(a)])      # camera centre, world
        zc = -c / np.linalg.norm(c)
        xc = np.cross(np.array([0.0, 1.0, 0.0]), zc)
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R_wc = np.stack([xc, yc, zc], axis=1)
        kf_T.append(se3(R_wc.T, -R_wc.T @ c))             # T_cw
    kf_T = np.array(kf_T)
    pts = rng.uniform([-2.5, -1.2, -2.0], [2.5, 1.2, 2.5], size=(n_pt, 3))
    mono_pt, mono_kf, mono_obs, mono_info = [], [], [], []
    st_pt, st_kf, st_obs, st_info = [], [], [], []
    max_obs = 8
    for j in range(n_pt):
        seen = []
        for i in rng.permutation(n_kf):
            pc = kf_T[i, :3, :3] @ pts[j] + kf_T[i, :3, 3]
            if pc[2] < 0.5:
                continue
            u, v = fx * pc[0] / pc[2] + cx, fy * pc[1] / pc[2] + cy
            if 0 <= u < W and 0 <= v < H:
                seen.append((i, u, v, pc[2]))
            if len(seen) >= max_obs:
                break
        for (i, u, v, z) in sorted(seen):
            octave = int(rng.integers(0, 8))
            inv_sigma2 = 1.0 / (1.2 ** (2 * octave))
            du, dv = rng.normal(scale=pixel_noise, size=2)
            if rng.random() < outlier_frac:
                du += 50.0 * rng.choice([-1, 1])
            if rng.random() < stereo_frac:
                st_pt.append(j); st_kf.append(i)
                st_obs.append([np.float32(u + du), np.float32(v + dv), np.float32(u + du - bf / z)])
                st_info.append(inv_sigma2)
            else:
                mono_pt.append(j); mono_kf.append(i)
                mono_obs.append([np.float32(u + du), np.float32(v + dv)])
                mono_info.append(inv_sigma2)
    # objects: SE3 T_ow, observed by up to obs_per_obj key-frames: Z = T_co = T_cw * T_wo  (+ noise)
    obj_T, oe_kf, oe_obj, oe_meas = [], [], [], []
    for o in range(n_obj):
        c = rng.uniform([-2.0, -0.5, -1.5], [2.0, 0.5, 2.0])
        T_wo = se3(rot_y(rng.uniform(0, 2 * np.pi)), c)
        T_ow = np.linalg.inv(T_wo)
        obj_T.append(T_ow)
        for i in sorted(rng.permutation(n_kf)[: min(obs_per_obj, n_kf)]):
            d = np.concatenate([rng.normal(scale=np.deg2rad(1.0), size=3), rng.normal(scale=0.02, size=3)])
            Z = se3(rodrigues(d[:3]), d[3:]) @ kf_T[i] @ T_wo
            oe_kf.append(i); oe_obj.append(o)
            oe_meas.append(pose7(Z.astype(np.float32).astype(np.float64)))
    # perturbed estimates (float32 round trip: Converter::toSE3Quat reads cv::Mat float32 poses, src/Converter.cc:37-54)
    kf_pose = []
    for i in range(n_kf):
        T = kf_T[i]
        if i >= n_fixed:
            d = np.concatenate([rng.normal(scale=np.deg2rad(1.0), size=3), rng.normal(scale=0.05, size=3)])
            T = se3(rodrigues(d[:3]), d[3:]) @ T
        kf_pose.append(pose7(T.astype(np.float32).astype(np.float64)))
    obj_pose = []
    for T in obj_T:
        d = np.concatenate([rng.normal(scale=np.deg2rad(2.0), size=3), rng.normal(scale=0.05, size=3)])
        obj_pose.append(pose7((se3(rodrigues(d[:3]), d[3:]) @ T).astype(np.float32).astype(np.float64)))
    pt_xyz = (pts + rng.normal(scale=0.02, size=pts.shape)).astype(np.float32).astype(np.float64)
    kf_id = np.arange(n_kf, dtype=np.int64) * 2            # mnId with gaps, as culled key-frames leave
    max_kf = int(kf_id.max())
    pt_mn = rng.permutation(3 * n_pt)[:n_pt].astype(np.int64)
    pt_id = pt_mn + max_kf + 1
    obj_id = np.arange(n_obj, dtype=np.int64)[::-1].copy() + max_kf + int(pt_mn.max() if n_pt else 0) + 2
    fixed = np.zeros(n_kf, np.uint8)
    fixed[:n_fixed] = 1
    K = np.tile(np.array([fx, fy, cx, cy, bf]), (n_kf, 1))

    def arr(x, dt, shape):
        return np.ascontiguousarray(np.array(x, dtype=dt).reshape(shape))
    return dict(
        kf_pose=arr(kf_pose, np.float64, (n_kf, 7)), kf_fixed=fixed, kf_id=kf_id, kf_K=arr(K, np.float64, (n_kf, 5)),
        pt_xyz=arr(pt_xyz, np.float64, (n_pt, 3)), pt_id=pt_id,
        obj_pose=arr(obj_pose, np.float64, (n_obj, 7)), obj_id=obj_id,
        mono_pt=arr(mono_pt, np.int32, (-1,)), mono_kf=arr(mono_kf, np.int32, (-1,)),
        mono_obs=arr(mono_obs, np.float64, (-1, 2)), mono_info=arr(mono_info, np.float64, (-1,)),
        st_pt=arr(st_pt, np.int32, (-1,)), st_kf=arr(st_kf, np.int32, (-1,)),
        st_obs=arr(st_obs, np.float64, (-1, 3)), st_info=arr(st_info, np.float64, (-1,)),
        oe_kf=arr(oe_kf, np.int32, (-1,)), oe_obj=arr(oe_obj, np.int32, (-1,)),
        oe_meas=arr(oe_meas, np.float64, (-1, 7)), oe_info=1e3,
        gt_kf=np.array([pose7(T) for T in kf_T]), gt_pt=pts, gt_obj=np.array([pose7(T) for T in obj_T]))


def make_ba_scene_large(seed, n_kf, n_pt, obs_per_pt=8, n_obj=0):
    """Vectorised generator for BANDWIDTH measurements of the linearisation kernels (million-edge graphs): every point
    lies in the region all key-frames see; each is observed by `obs_per_pt` distinct random key-frames (mono)."""
    rng = np.random.default_rng(seed)
    small = make_ba_scene(seed, n_kf, 1, n_obj, obs_per_obj=min(10, n_kf))
    fx, fy, cx, cy = small["kf_K"][0, :4]
    T = np.array([pose7_to_T(p) for p in small["gt_kf"]])
    pts = rng.uniform([-0.8, -0.5, -0.5], [0.8, 0.5, 0.8], size=(n_pt, 3))
    k = min(obs_per_pt, n_kf)
    kf = np.argsort(rng.random((n_pt, n_kf)), axis=1)[:, :k].astype(np.int32)      # distinct key-frames per point
    kf.sort(axis=1)
    pt_idx = np.repeat(np.arange(n_pt, dtype=np.int32), k)
    kf_idx = kf.reshape(-1)
    pc = np.einsum("eij,ej->ei", T[kf_idx, :3, :3], pts[pt_idx]) + T[kf_idx, :3, 3]
    uv = np.stack([fx * pc[:, 0] / pc[:, 2] + cx, fy * pc[:, 1] / pc[:, 2] + cy], 1) + rng.normal(size=(len(pt_idx), 2))
    out = dict(small)
    out.update(pt_xyz=(pts + rng.normal(scale=0.02, size=pts.shape)), pt_id=np.arange(n_pt, dtype=np.int64) + int(small["kf_id"].max()) + 1,
               mono_pt=pt_idx, mono_kf=kf_idx, mono_obs=uv.astype(np.float32).astype(np.float64),
               mono_info=np.ones(len(pt_idx)), st_pt=np.zeros(0, np.int32), st_kf=np.zeros(0, np.int32),
               st_obs=np.zeros((0, 3)), st_info=np.zeros(0), gt_pt=pts)
    if n_obj:
        out["obj_id"] = np.arange(n_obj, dtype=np.int64) + int(out["pt_id"].max()) + 2
    return out


def make_pose_problem(seed, n=400, stereo_frac=0.3, outlier_frac=0.1, pose_noise=(0.03, 0.02)):
    """One frame for Optimizer::PoseOptimization (reference src/Optimizer.cc:244-456): n map points in front of a camera,
    pixel noise 1 px, `outlier_frac` gross outliers (30-80 px), initial pose = truth * exp(noise).
    Returns dict(K (5,), pose (7,) initial T_cw, X (n,3), obs (n,3) u v u_right, info (n,), stereo (n,) uint8, gt_pose)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy, bf = 535.4, 539.2, 320.1, 247.6, 40.0
    T_gt = se3(rot_y(rng.uniform(-0.2, 0.2)) @ np.diag([1.0, 1.0, 1.0]), rng.uniform(-0.3, 0.3, 3))
    Xc = np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(1.5, 8.0, n)], 1)
    Xw = (np.linalg.inv(T_gt) @ np.concatenate([Xc, np.ones((n, 1))], 1).T).T[:, :3]
    u = fx * Xc[:, 0] / Xc[:, 2] + cx
    v = fy * Xc[:, 1] / Xc[:, 2] + cy
    ur = u - bf / Xc[:, 2]
    obs = np.stack([u, v, ur], 1) + rng.normal(size=(n, 3))
    bad = rng.random(n) < outlier_frac
    obs[bad, :2] += rng.uniform(30, 80, size=(bad.sum(), 2)) * rng.choice([-1, 1], size=(bad.sum(), 2))
    stereo = (rng.random(n) < stereo_frac).astype(np.uint8)
    obs[stereo == 0, 2] = -1.0
    octave = rng.integers(0, 8, n)
    info = 1.2 ** (-2.0 * octave)
    w = rng.normal(scale=pose_noise[1], size=3)
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    dR = np.eye(3) + np.sin(th) / max(th, 1e-12) * Kx + (1 - np.cos(th)) / max(th * th, 1e-12) * Kx @ Kx
    T0 = se3(dR, rng.normal(scale=pose_noise[0], size=3)) @ T_gt
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)      # the map holds float32 pixels / poses
    return dict(K=np.array([fx, fy, cx, cy, bf]), pose=pose7(f32(T0)), X=f32(Xw), obs=f32(obs), info=f32(info),
                stereo=stereo, gt_pose=pose7(T_gt), gross=bad)
