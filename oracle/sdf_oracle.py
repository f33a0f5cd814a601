"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, float32) restatement of hot path A, the DeepSDF object refinement.

This file is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product path (qsp_slam_amd/) never does; it fails loudly when the HIP library is missing.

Parity pin: oracle/gen_golden_sdf.py runs the reference's own Python path (imported from /root/reference in the
build container) and this restatement on identical inputs; tests/test_oracle_sdf.py checks this file against
those committed outputs (tests/golden/sdf_*.npz).  The reference has no tests or golden vectors of its own
(SURVEY.md section 4).

Every function cites the reference lines it follows (paths relative to the reference root).
"""
import ast
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------------------
# decoder  (deep_sdf/deep_sdf_decoder.py:9-110, deep_sdf/workspace.py:202-224)
# ----------------------------------------------------------------------------------------------------------

class DecoderWeights(object):
    """Folded DeepSDF decoder: list of (W (out,in) f32, b (out,) f32) and the latent-skip layer index."""

    def __init__(self, layers, latent_in, code_len, use_tanh=False):
        self.layers = [(np.ascontiguousarray(W, dtype=F32), np.ascontiguousarray(b, dtype=F32)) for W, b in layers]
        self.latent_in = tuple(latent_in)
        self.code_len = int(code_len)
        self.use_tanh = bool(use_tanh)      # NetworkSpecs.use_tanh: tanh on the output layer, then the final tanh (:92-94,107-108)

    @property
    def mac_per_point(self):
        return int(sum(W.shape[0] * W.shape[1] for W, _ in self.layers))


def fold_state_dict(state, latent_in=(4,), code_len=64):
    """weight_norm (old-style, dim=0): W = g * v / ||v||_row  (deep_sdf_decoder.py:49-54; torch weight_norm).
    `state` maps 'lin{l}.weight_v/.weight_g/.bias' (or plain '.weight') -> arrays; 'module.' prefixes (DataParallel
    checkpoints, workspace.py:215-220) are stripped."""
    st = {}
    for k, v in state.items():
        if k.startswith("module."):
            k = k[len("module."):]
        st[k] = np.asarray(v)
    layers = []
    l = 0
    while ("lin%d.bias" % l) in st:
        if ("lin%d.weight_v" % l) in st:
            v = st["lin%d.weight_v" % l].astype(F32)
            g = st["lin%d.weight_g" % l].astype(F32).reshape(-1, 1)
            nrm = np.sqrt((v.astype(F32) ** 2).sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
            W = (v * (g / nrm)).astype(F32)
        else:
            W = st["lin%d.weight" % l].astype(F32)
        layers.append((W, st["lin%d.bias" % l].astype(F32)))
        l += 1
    return DecoderWeights(layers, latent_in, code_len)


def load_decoder_npz(path, use_tanh=None):
    z = np.load(path, allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))  # written by oracle/fit_decoder.py, a literal dict
    state = {k: z[k] for k in z.files if k != "meta"}
    dec = fold_state_dict(state, latent_in=meta["latent_in"], code_len=meta["latent_size"])
    dec.use_tanh = bool(meta.get("use_tanh", False)) if use_tanh is None else bool(use_tanh)
    return dec


def decoder_forward(dec, inp, keep=False):
    """eval-mode forward (deep_sdf_decoder.py:75-110): ReLU after all but the last layer, input re-concatenated at
    the latent_in layer, dropout = identity, final tanh.  inp (N, code_len+3) f32 -> y (N,) [, cache]"""
    x = inp.astype(F32, copy=False)
    h = x
    pre = []
    n_layers = len(dec.layers)
    for l, (W, b) in enumerate(dec.layers):
        if l in dec.latent_in:
            h = np.concatenate([h, x], axis=-1)
        a = h @ W.T + b
        if l < n_layers - 1:
            if keep:
                pre.append(a > 0)
            h = np.maximum(a, F32(0))
        else:
            h = a
    y = np.tanh(h[:, 0]).astype(F32)
    if getattr(dec, "use_tanh", False):
        pre.append(y) if keep else None     # (the inner tanh's output: the backward pass needs it)
        y = np.tanh(y).astype(F32)
    return (y, pre) if keep else y


def decoder_value_and_input_grad(dec, inp):
    """y and dy/d(inp) per row: the result of get_batch_sdf_jacobian (reconstruct/loss_utils.py:82-103) without the
    unused weight gradients.  Returns y (N,), grad (N, code_len+3)."""
    y, masks = decoder_forward(dec, inp, keep=True)
    n_layers = len(dec.layers)
    W_last = dec.layers[-1][0]
    dy = (F32(1) - y * y).astype(F32)                                 # d tanh (the outer one with use_tanh)
    if getattr(dec, "use_tanh", False):
        y1 = masks.pop()                                               # tanh(t); autograd multiplies the outer factor first
        dy = (dy * (F32(1) - y1 * y1)).astype(F32)
    g = (dy[:, None] * W_last[0][None, :]).astype(F32)
    g_in = np.zeros_like(inp, dtype=F32)
    for l in range(n_layers - 2, -1, -1):
        W = dec.layers[l][0]
        g = (g * masks[l]) @ W
        if l in dec.latent_in:
            k = W.shape[1] - inp.shape[1]
            g_in += g[:, k:]
            g = g[:, :k]
    g_in += g
    return y, g_in.astype(F32)


def decode_sdf(dec, code, x):
    """no-grad forward with the latent broadcast to every point (loss_utils.py:51-79); x (N,3) -> (N,)"""
    x = np.asarray(x, dtype=F32)
    inp = np.concatenate([np.broadcast_to(np.asarray(code, F32)[: dec.code_len], (x.shape[0], dec.code_len)), x], -1)
    return decoder_forward(dec, inp)


# ----------------------------------------------------------------------------------------------------------
# Lie-group pieces (reconstruct/loss_utils.py:107-233)
# ----------------------------------------------------------------------------------------------------------

def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=F32)


def exp_se3(x):
    """loss_utils.py:129-163 (translation first, then rotation; theta<=1e-8 -> identity blocks)"""
    x = np.asarray(x, F32)
    v, w = x[:3], x[3:6]
    W = _hat(w)
    W2 = W @ W
    th = F32(np.sqrt((w * w).sum(dtype=F32)))
    I = np.eye(3, dtype=F32)
    if th <= 1e-8:
        R, J = I, I
    else:
        s, c = F32(np.sin(th)), F32(np.cos(th))
        R = I + W * (s / th) + W2 * ((F32(1) - c) / (th * th))
        J = I + ((F32(1) - c) / (th * th)) * W + ((th - s) / (th * th * th)) * W2
    T = np.eye(4, dtype=F32)
    T[:3, :3] = R
    T[:3, 3] = J @ v
    return T.astype(F32)


def exp_sim3(x):
    """loss_utils.py:188-233, including `c = 0 if s <= eps` on the theta>eps branch (:223)"""
    x = np.asarray(x, F32)
    v, w, s = x[:3], x[3:6], F32(x[6])
    W = _hat(w)
    W2 = W @ W
    th = F32(np.sqrt((w * w).sum(dtype=F32)))
    es = F32(np.exp(s))
    I = np.eye(3, dtype=F32)
    if th <= 1e-8:
        R = I
        J = I if s == 0 else ((es - F32(1)) / s) * I
    else:
        sn, cs = F32(np.sin(th)), F32(np.cos(th))
        R = I + W * (sn / th) + W2 * ((F32(1) - cs) / (th * th))
        a, b = es * sn, es * cs
        c = F32(0) if s <= 1e-8 else (es - F32(1)) / s
        den = s * s + th * th
        k1 = (a * s + (F32(1) - b) * th) / den
        k2 = c - ((b - F32(1)) * s + a * th) / den
        J = c * I + k1 * W / th + k2 * W2 / (th * th)
    T = np.eye(4, dtype=F32)
    T[:3, :3] = es * R
    T[:3, 3] = J @ v
    return T.astype(F32)


def points_to_pose_jacobian(pts, sim3=True):
    """d x_o / d xi for a left perturbation: [I | -[x]x | x]  (loss_utils.py:107-126,166-185); (N,3,6|7)"""
    n = pts.shape[0]
    J = np.zeros((n, 3, 7 if sim3 else 6), dtype=F32)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1
    J[:, 0, 4], J[:, 0, 5] = z, -y
    J[:, 1, 3], J[:, 1, 5] = -z, x
    J[:, 2, 3], J[:, 2, 4] = y, -x
    if sim3:
        J[:, :, 6] = pts
    return J


# ----------------------------------------------------------------------------------------------------------
# robust weights (loss_utils.py:236-265)
# ----------------------------------------------------------------------------------------------------------

def robust_residual(res, b):
    """returns (w*res, mean((w*res)^2), w) with w = sqrt(rho(|r|))/|r| (|r|==0 -> divide by 1)"""
    res = np.asarray(res, F32).reshape(-1)
    a = np.abs(res)
    b = F32(b)
    rho = np.where(a <= b, a * a, F32(2) * b * a - b * b).astype(F32)
    den = np.where(a == 0, F32(1), a)
    w = (np.sqrt(rho) / den).astype(F32)
    rr = (w * res).astype(F32)
    loss = F32(np.mean(rr * rr, dtype=F32)) if rr.size else F32(np.nan)
    return rr, loss, w


# ----------------------------------------------------------------------------------------------------------
# loss terms (reconstruct/loss.py)
# ----------------------------------------------------------------------------------------------------------

def transform_points(T_oc, pts):
    """x_o = R x_c + t, written as the reference's broadcast-multiply-sum (loss.py:31-32)"""
    T_oc = np.asarray(T_oc, F32)
    return ((pts[..., None, :] * T_oc[:3, :3]).sum(-1, dtype=F32) + T_oc[:3, 3]).astype(F32)


def sdf_term(dec, pts_cam, T_oc, code):
    """compute_sdf_loss, loss.py:22-43 -> J_pose (N,7), J_code (N,L), res (N,)"""
    x_o = transform_points(T_oc, np.asarray(pts_cam, F32))
    inp = np.concatenate([np.broadcast_to(np.asarray(code, F32), (x_o.shape[0], dec.code_len)), x_o], -1)
    y, g = decoder_value_and_input_grad(dec, inp)
    gx = g[:, -3:]
    Jp = np.einsum("ni,nij->nj", gx, points_to_pose_jacobian(x_o)).astype(F32)
    return Jp, g[:, :-3], y, x_o


def render_term(dec, rays, depth_obs, T_oc, depths, code, th=0.01):
    """compute_render_loss, loss.py:46-152.  Returns None when fewer than 10 samples fall in the unit ball, else
    dict(J_pose (K,7), J_code (K,L), res (K,), n_valid, ray (K,), k (K,), pts (K,3), de_ds (K,))."""
    rays = np.asarray(rays, F32)
    depths = np.asarray(depths, F32)
    T_oc = np.asarray(T_oc, F32)
    R, D = rays.shape[0], depths.shape[0]
    p_cam = rays[:, None, :] * depths[:, None]                      # (R,D,3)  loss.py:60
    p_obj = transform_points(T_oc, p_cam)                           # loss.py:62-63
    vr, vk = np.where(np.sqrt((p_obj * p_obj).sum(-1, dtype=F32)) < 1.0)   # row-major (ray, depth) order, :68
    q = p_obj[vr, vk]
    if q.shape[0] < 10:
        return None                                                 # :73-74
    s = decode_sdf(dec, code, q)                                    # :78
    occ = np.zeros((R, D), dtype=F32)
    th = F32(th)
    occ[vr, vk] = F32(0.5) - np.clip(s, -th, th) / (F32(2) * th)    # loss_utils.py:40-48
    wg = (s > -th) & (s < th)                                       # :89
    gr, gk = vr[wg], vk[wg]
    rows = occ[gr, :]                                               # (m,D) one row per point with gradient
    m = rows.shape[0]
    d_min, d_max = depths[0], depths[-1]
    acc = np.cumprod(F32(1) - rows, axis=-1, dtype=F32)             # :99
    acc_aug = np.concatenate([np.ones((m, 1), F32), acc], -1)
    o = np.concatenate([rows, np.ones((m, 1), F32)], -1)
    d = np.concatenate([depths, np.array([F32(1.1) * d_max], F32)])
    term = o * acc_aug
    d_u = (d * term).sum(-1, dtype=F32)                             # :113
    o_k = occ[gr, gk]
    acc_m = np.where(np.arange(D)[None, :] < gk[:, None], F32(0), acc)   # :120-121
    de_do = acc_m.sum(-1, dtype=F32) / (F32(1) - o_k)               # :122
    nz = de_do > 1e-2                                               # :125
    de_do, d_u = de_do[nz], d_u[nz]
    delta_d = (d_max - d_min) / F32(D - 1)
    de_ds = (de_do * delta_d * (F32(-1.0) / (F32(2) * th))).astype(F32)  # :128-130
    gr, gk = gr[nz], gk[nz]
    res = (np.asarray(depth_obs, F32)[gr] - d_u).astype(F32)
    res = np.clip(res, F32(-0.30), F32(0.30))                       # :136-141
    pts = p_obj[gr, gk]
    inp = np.concatenate([np.broadcast_to(np.asarray(code, F32), (pts.shape[0], dec.code_len)), pts], -1)
    _, g = decoder_value_and_input_grad(dec, inp) if pts.shape[0] else (None, np.zeros((0, inp.shape[1]), F32))
    g = (de_ds[:, None] * g).astype(F32)                            # :145
    Jp = np.einsum("ni,nij->nj", g[:, -3:], points_to_pose_jacobian(pts)).astype(F32)
    return dict(J_pose=Jp, J_code=g[:, :-3], res=res, n_valid=int(q.shape[0]), ray=gr, k=gk, pts=pts, de_ds=de_ds,
                sdf_valid=s, valid_ray=vr, valid_k=vk)


def rotation_term(T_oc):
    """compute_rotation_loss_sim3, loss.py:155-178 -> (J (7,), res)"""
    T_co = np.linalg.inv(np.asarray(T_oc, F32)).astype(F32)
    r_co = T_co[:3, :3].copy()
    scale = F32(np.linalg.det(r_co)) ** F32(1.0 / 3.0)
    r_co = (r_co / scale).astype(F32)
    r_oc = np.linalg.inv(r_co).astype(F32)
    ey = np.array([0, 1, 0], F32)
    ng = np.array([0, -1, 0], F32)
    res = F32(1) - F32((r_co @ ey) @ ng)
    J = np.zeros(7, F32)
    if res < 1e-7:
        return J, F32(0)
    J[3:6] = np.cross(r_oc @ ng, ey)
    return J, res


# ----------------------------------------------------------------------------------------------------------
# entry points (reconstruct/optimizer.py)
# ----------------------------------------------------------------------------------------------------------

class JointConfig(object):
    """the `optimizer` block of the detector config JSON (configs/config_*.json:21-41)"""

    def __init__(self, k1=10.0, k2=100.0, k3=2.5, k4=0.0, b1=0.2, b2=0.02, lr=1.0, s_damp=100.0, n_iter=5,
                 n_depth=50, cut_off=0.01, code_len=64, n_iter_pose=5):
        self.k1, self.k2, self.k3, self.k4 = k1, k2, k3, k4
        self.b1, self.b2, self.lr, self.s_damp = b1, b2, lr, s_damp
        self.n_iter, self.n_depth, self.cut_off, self.code_len = n_iter, n_depth, cut_off, code_len
        self.n_iter_pose = n_iter_pose


def normal_equations(cfg, Jp_s, Jc_s, rr_s, Jp_r, Jc_r, rr_r, code, J_rot, res_rot):
    """optimizer.py:217-252: unweighted J in H, Huber-weighted r in b, /N per term, code prior k3, rotation prior k4,
    unit damping on the pose block and s_damp on scale.  All float32."""
    L = cfg.code_len
    Js = np.concatenate([Jp_s, Jc_s], -1).astype(F32)
    Jr = np.concatenate([Jp_r, Jc_r], -1).astype(F32)
    H = F32(cfg.k1) * (Jr.T @ Jr) / F32(Jr.shape[0]) + F32(cfg.k2) * (Js.T @ Js) / F32(Js.shape[0])
    b = -F32(cfg.k1) * (Jr.T @ rr_r) / F32(Jr.shape[0]) - F32(cfg.k2) * (Js.T @ rr_s) / F32(Js.shape[0])
    H = H.astype(F32)
    b = b.astype(F32)
    H[7:7 + L, 7:7 + L] += F32(cfg.k3) * np.eye(L, dtype=F32)
    b[7:7 + L] -= F32(cfg.k3) * np.asarray(code, F32)
    H[:7, :7] += F32(cfg.k4) * np.outer(J_rot, J_rot).astype(F32)
    b[:7] -= F32(cfg.k4) * (-(J_rot * res_rot)).astype(F32)
    H[:7, :7] += np.eye(7, dtype=F32)
    H[6, 6] += F32(cfg.s_damp)
    return H, b


def gn_iteration(dec, cfg, T_oc, z, pts, rays, depth_obs, n_fg):
    """One Gauss-Newton iteration of reconstruct_object from the state (T_oc, z): optimizer.py:141-263.
    Returns None-valued 'fail' key on the reference's early exits, else all intermediate quantities and the new state.
    depth_obs is (n_rays,) with the foreground depths in front; the background entries are overwritten here."""
    L = cfg.code_len
    T_oc = np.asarray(T_oc, F32)
    z = np.asarray(z, F32)
    T_co = np.linalg.inv(T_oc).astype(F32)
    scale = F32(np.linalg.det(T_co[:3, :3])) ** F32(1.0 / 3.0)
    d_min, d_max = T_co[2, 3] - scale, T_co[2, 3] + scale
    depths = np.linspace(d_min, d_max, cfg.n_depth, dtype=F32)           # optimizer.py:148-151
    depth_obs = np.array(depth_obs, F32)
    depth_obs[n_fg:] = F32(1.1) * d_max                                  # :153
    Jp_s, Jc_s, res_s, _ = sdf_term(dec, pts, T_oc, z)
    rr_s, loss_s, _ = robust_residual(res_s, cfg.b2)
    if np.isnan(loss_s):
        return dict(fail="sdf_nan")                                      # :168-169
    rt = render_term(dec, rays, depth_obs, T_oc, depths, z, th=cfg.cut_off)
    if rt is None:
        return dict(fail="render_none")                                  # :171-172
    rr_r, loss_r, _ = robust_residual(rt["res"], cfg.b1)
    if np.isnan(loss_r):
        return dict(fail="render_nan", n_valid=rt["n_valid"], K=0)       # :193-194
    J_rot, res_rot = rotation_term(T_oc)
    loss = float(F32(cfg.k1) * loss_r + F32(cfg.k2) * loss_s)            # :203
    H, b = normal_equations(cfg, Jp_s, Jc_s, rr_s, rt["J_pose"], rt["J_code"], rr_r, z, J_rot, res_rot)
    dx = (np.linalg.inv(H).astype(F32) @ b).astype(F32)                  # :254
    T_new = (exp_sim3(F32(cfg.lr) * dx[:7]) @ T_oc).astype(F32)          # :261-262
    z_new = (z + F32(cfg.lr) * dx[7:7 + L]).astype(F32)                  # :263
    return dict(fail=None, T_oc=T_oc.copy(), code=z.copy(), res_sdf=res_s, Jp_sdf=Jp_s, Jc_sdf=Jc_s,
                res_render=rt["res"], Jp_render=rt["J_pose"], Jc_render=rt["J_code"], n_valid=rt["n_valid"],
                K=int(rt["res"].shape[0]), H=H, b=b, dx=dx, loss=loss, loss_sdf=float(loss_s),
                loss_render=float(loss_r), T_oc_new=T_new, code_new=z_new, J_rot=J_rot, res_rot=res_rot)


def reconstruct_object(dec, cfg, t_cam_obj, pts, rays, depth, code=None, trace=None):
    """Optimizer.reconstruct_object, optimizer.py:96-281.
    Returns dict(t_cam_obj (4,4) f32 | None, code (L,) f32 | None, is_good, loss)."""
    L = cfg.code_len
    z = np.zeros(L, F32) if code is None else np.asarray(code, F32)[:L].copy()
    T_oc = np.linalg.inv(np.asarray(t_cam_obj, F32)).astype(F32)
    rays = np.asarray(rays, F32)
    n_fg = int(np.asarray(depth).shape[0])
    depth_obs = np.concatenate([np.asarray(depth, F32), np.zeros(rays.shape[0] - n_fg, F32)])
    pts = np.asarray(pts, F32)
    loss = 0.0
    for e in range(cfg.n_iter):
        it = gn_iteration(dec, cfg, T_oc, z, pts, rays, depth_obs, n_fg)
        if it["fail"] is not None:
            return dict(t_cam_obj=None, code=None, is_good=False, loss=loss)
        if trace is not None:
            trace.append(it)
        loss = it["loss"]
        T_oc, z = it["T_oc_new"], it["code_new"]
    return dict(t_cam_obj=np.linalg.inv(T_oc).astype(F32), code=z, is_good=True, loss=loss)


def estimate_pose_cam_obj(dec, cfg, t_co_se3, scale, pts, code, trace=None):
    """Optimizer.estimate_pose_cam_obj, optimizer.py:47-93: SDF-only GN on the 6 pose dims, 1e-2 damping, raw
    residual in b, inlier filter |res|<=0.05 after iteration index 4.  Returns (4,4) f32 SE3."""
    T_co = np.asarray(t_co_se3, F32).copy()
    T_co[:3, :3] *= F32(scale)
    T_oc = np.linalg.inv(T_co).astype(F32)
    z = np.asarray(code, F32)[: cfg.code_len]
    pts = np.asarray(pts, F32)
    for e in range(cfg.n_iter_pose):
        Jp, _, res, _ = sdf_term(dec, pts, T_oc, z)
        J = Jp[:, :6]
        n = F32(J.shape[0])
        H = (J.T @ J).astype(F32) / n + F32(1e-2) * np.eye(6, dtype=F32)
        b = -(J.T @ res).astype(F32) / n
        dx = (np.linalg.inv(H).astype(F32) @ b).astype(F32)
        if trace is not None:
            trace.append(dict(T_oc=T_oc.copy(), H=H, b=b, dx=dx, n=int(J.shape[0])))
        T_oc = (exp_se3(dx) @ T_oc).astype(F32)
        if e == 4:
            pts = pts[np.abs(res) <= 0.05]
    T_co = np.linalg.inv(T_oc).astype(F32)
    T_co[:3, :3] /= F32(scale)
    return T_co


def create_voxel_grid(vol_dim):
    """reconstruct/utils.py:98-117.  NOTE: the reference divides a LongTensor with `/`, which is true division on
    every torch >= 1.6, so the y and x coordinates are *not* floored: values[:,1] = (i/vol_dim) % vol_dim keeps its
    fractional part.  Restated as written."""
    i = np.arange(vol_dim ** 3, dtype=np.int64)
    size = F32(2.0 / (vol_dim - 1))
    v = np.zeros((vol_dim ** 3, 3), dtype=F32)
    v[:, 2] = (i % vol_dim).astype(F32)
    v[:, 1] = np.mod((i / vol_dim).astype(F32), F32(vol_dim))
    v[:, 0] = np.mod(((i / vol_dim).astype(F32) / F32(vol_dim)), F32(vol_dim))
    v[:, 0] = v[:, 0] * size - F32(1)
    v[:, 1] = v[:, 1] * size - F32(1)
    v[:, 2] = v[:, 2] * size - F32(1)
    return v
