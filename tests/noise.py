"""The reference's OWN float32 rounding noise as the yardstick of path A's tolerances.

tests/golden/sdf_noise_<case>.npz (oracle/gen_noise_sdf.py, produced by RUNNING the reference) holds, for every Gauss-Newton
iteration of every joint golden case restarted from the committed fixture's state, the reference's float64 evaluation (`*64`) and
six further float32 evaluations of it (`*32[iteration][sample]`: one thread, four row permutations, correctly rounded 4x4
inverses); the committed fixture's own `it_*` values are the seventh.  For a quantity q

    noise(q, i) = max over the seven float32 samples s of  relerr(ref32_s(q, i), ref64(q, i))

is how far the reference lands from its own exact value by rounding alone.  An implementation is held to

    relerr(mine(q, i), ref64(q, i))  <=  max(1e-4, 2 x noise(q, i))

i.e. north_star's 1e-4, or -- where the reference itself cannot meet 1e-4 -- twice the reference's own scatter.  `ratio()` returns
err / max(noise, 5e-5), which must stay <= 2; the measured ratios go to the margins file."""
import os

import numpy as np

NORTH_STAR = 1e-4
QUANTITIES = ("H", "b", "dx", "T_next", "code_next")


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name.replace("sdf_joint_", "sdf_noise_") + ".npz"))


def fixture_sample(z, nz, q, i):
    """the committed fixture's own float32 value of q at iteration i (the seventh sample); the next state of the last
    iteration is not in the fixture (the thread-count sample stands in: same arithmetic, same thread-independent order)"""
    if q in ("H", "b", "dx"):
        return z["it_" + q][i]
    n_it = z["it_H"].shape[0]
    if i + 1 < n_it:
        return z["it_T_oc"][i + 1] if q == "T_next" else z["it_code"][i + 1]
    return nz[q + "32"][i][0]


def b_decoder_part(b, k4, Jrot, res_rot):
    """the right-hand side without the rotation prior's share: optimizer.py:233-237 adds k4 * J_rot * res_rot to b[:7]"""
    b = np.asarray(b, np.float64).copy()
    b[:7] -= k4 * np.asarray(Jrot, np.float64) * float(res_rot)
    return b


def noise(z, nz, q, i, k4=0.0):
    ref = nz[q + "64"][i]
    samples = [nz[q + "32"][i][s] for s in range(nz[q + "32"].shape[1])] + [fixture_sample(z, nz, q, i)]
    if q == "b" and k4 != 0.0:
        ref = b_decoder_part(ref, k4, nz["Jrot64"][i], nz["res_rot64"][i])
        rots = [(nz["Jrot32"][i][s], nz["res_rot32"][i][s]) for s in range(nz["Jrot32"].shape[1])]
        rots.append((z["it_Jrot"][i], z["it_res_rot"][i]))
        samples = [b_decoder_part(s_, k4, j_, r_) for s_, (j_, r_) in zip(samples, rots)]
    return max(relerr(s_, ref) for s_ in samples)


def ratio(z, nz, q, i, mine, k4=0.0, mine_rot=None):
    """err(mine, ref64) / max(noise, 5e-5); <= 2 means: within 1e-4, or within twice the reference's own float32 scatter.
    With k4 != 0, `b` is compared without each side's own rotation-prior share (`mine_rot` = (J_rot, res_rot) of `mine`): that
    share is k4 = 1e7 times a float32 cancellation and is checked on its own, absolutely (`res_rot_error`)."""
    ref = nz[q + "64"][i]
    if q == "b" and k4 != 0.0:
        ref = b_decoder_part(ref, k4, nz["Jrot64"][i], nz["res_rot64"][i])
        mine = b_decoder_part(mine, k4, mine_rot[0], mine_rot[1])
    err = relerr(mine, ref)
    nse = noise(z, nz, q, i, k4)
    return err / max(nse, NORTH_STAR / 2), err, nse


def res_rot_error(nz, i, mine_res_rot):
    """|res_rot - ref64's| and the bar: `1 - cos(tilt)` evaluated in float32 from entries near 1 carries an ABSOLUTE rounding
    error of a few ulp(1) = 6e-8 each (the 4x4 inverse, the determinant's cube root, the division, the subtraction): 4 ulp, or
    twice what the reference's own samples show, whichever is larger"""
    ref = float(nz["res_rot64"][i])
    sampled = max(abs(float(r) - ref) for r in nz["res_rot32"][i])
    return abs(float(mine_res_rot) - ref), max(4 * 5.97e-8, 2 * sampled)


def case_noise(z, nz, q, k4=0.0):
    """the largest scatter the reference shows for q at ANY iteration of the case (same observations, same conditioning): nine
    samples per iteration estimate a scale, they do not bound a tail, and a rare event -- a ReLU pre-activation within rounding
    of zero, a clamp boundary -- lands in one iteration or the next by chance"""
    return max(noise(z, nz, q, i, k4) for i in range(nz["H64"].shape[0]))


def knife_edge_replacement(cfg, z, i, mine, rows_sdf, rows_render, oracle_it):
    """ReLU knife-edge accounting.  d sdf / d input of a ReLU network jumps where a pre-activation crosses zero; two evaluations
    that round differently put roughly one row in 2000 on different sides (tests/test_oracle_sdf.py:rows_close), which moves that
    ROW by 1e-3 .. 1e-2 of the largest entry while every other row agrees to ~3e-7.  Rows of `mine` that differ from the numpy
    oracle's by more than 1e-4 are taken from the oracle instead: H, b are corrected by the rank-one terms, dx and the next state
    recomputed in float64.  Returns (corrected quantities, number of rows replaced)."""
    from oracle import sdf_oracle as so
    m, K = rows_sdf.shape[0], rows_render.shape[0]
    H = np.asarray(mine["H"], np.float64).copy()
    b = np.asarray(mine["b"], np.float64).copy()
    n_rep = 0
    for rows, Jp, Jc, res, hub, coef in ((rows_sdf, oracle_it["Jp_sdf"], oracle_it["Jc_sdf"], oracle_it["res_sdf"], cfg.b2, cfg.k2 / m),
                                         (rows_render, oracle_it["Jp_render"], oracle_it["Jc_render"], oracle_it["res_render"], cfg.b1,
                                          cfg.k1 / max(K, 1))):
        n = rows.shape[0]
        o = np.concatenate([np.asarray(Jp).reshape(n, -1), np.asarray(Jc).reshape(n, -1)], 1).astype(np.float64)
        ro = so.robust_residual(np.asarray(res).reshape(-1), hub)[0].reshape(-1).astype(np.float64)
        g = rows[:, :71].astype(np.float64)
        rg = rows[:, 71].astype(np.float64)
        d = np.abs(g - o).max(1) / np.abs(o).max()
        for r in np.nonzero(d > 1e-4)[0]:
            H += coef * (np.outer(o[r], o[r]) - np.outer(g[r], g[r]))
            b -= coef * (o[r] * ro[r] - g[r] * rg[r])
            n_rep += 1
    dx = np.linalg.solve(H, b)
    T_next = so.exp_sim3((cfg.lr * dx[:7]).astype(np.float32)).astype(np.float64) @ z["it_T_oc"][i].astype(np.float64)
    code_next = z["it_code"][i].astype(np.float64) + cfg.lr * dx[7:]
    return dict(H=H, b=b, dx=dx, T_next=T_next, code_next=code_next), n_rep


def check_gpu_iterations(tag, decoder, golden_dir, name, make_cfg, cfg_from, within):
    """The teacher-forced loop of the GPU tests against the reference's float64 evaluation and the reference's own float32
    scatter; one place for the f32 tile, the split pipes and the screened forward.  For every iteration and quantity
        err(mine, ref64) <= max(1e-4, 2 x case_noise)
    holds, where an iteration that misses it must be explained by at most ceil(0.3 %) ReLU knife-edge rows (those rows taken from
    the numpy oracle, `knife_edge_replacement`).  The per-iteration ratio err / max(noise(q, i), 5e-5) goes to the margins file
    for the record (it is <= 2 in 22 of the 25 iterations per pipe)."""
    from oracle import sdf_oracle as so
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    nz = load(golden_dir, name)
    cfg = cfg_from(z)
    odec = so.load_decoder_npz(os.path.join(golden_dir, "decoder_8x512.npz"))
    opt = Optimizer(decoder, make_cfg(z))
    batch = RefineBatch(decoder, _joint_cfg(opt), [z["pts"]], [z["rays"]], [z["depth"]], [0])
    batch.enable_rows(True)
    cnoise = {q: case_noise(z, nz, q, cfg.k4) for q in QUANTITIES}
    n_fg = z["depth"].shape[0]
    dobs = np.concatenate([z["depth"], np.zeros(z["rays"].shape[0] - n_fg, np.float32)])
    strict_ok = strict_all = replaced = 0
    try:
        for i in range(z["it_H"].shape[0]):
            T_co = np.linalg.inv(z["it_T_oc"][i].astype(np.float64)).astype(np.float32)
            batch.set_state(T_co[None], z["it_code"][i][None])
            batch.run(1)
            tr = batch.trace()
            rot = batch.trace_rot()[0].astype(np.float64)
            T, code, loss, good = batch.get()
            K = int(tr["K"][0])
            assert good[0] and K == int(nz["K64"][i]) == int(z["it_K"][i])
            mine = dict(H=tr["H"][0], b=tr["b"][0], dx=tr["dx"][0], T_next=np.linalg.inv(T[0].astype(np.float64)), code_next=code[0])
            J_rot = np.zeros(7)
            J_rot[3:6] = rot[:3]
            errs = {}
            for q in QUANTITIES:
                r, errs[q], nse = ratio(z, nz, q, i, mine[q], cfg.k4, (J_rot, rot[3]))
                within("%s/%s/per_iteration/err_vs_ref64_over_max(reference_noise,5e-5)/%s" % (tag, name, q), r, 2.0)
                strict_all += 1
                strict_ok += r <= 2.0
            if any(errs[q] > max(NORTH_STAR, 2 * cnoise[q]) for q in QUANTITIES):
                it = so.gn_iteration(odec, cfg, z["it_T_oc"][i], z["it_code"][i], z["pts"], z["rays"], dobs, n_fg)
                assert it["fail"] is None and it["K"] == K
                rs, rr = batch.rows(0, z["pts"].shape[0], K)
                mine, n_rep = knife_edge_replacement(cfg, z, i, mine, rs, rr, it)
                assert 1 <= n_rep <= int(np.ceil(0.003 * (z["pts"].shape[0] + K))), (name, i, n_rep)
                replaced += n_rep
                errs = {q: ratio(z, nz, q, i, mine[q], cfg.k4, (J_rot, rot[3]))[1] for q in QUANTITIES}
            for q in QUANTITIES:
                assert within("%s/%s/err_vs_ref64_over_max(1e-4,2x_case_noise)/%s" % (tag, name, q),
                              errs[q] / max(NORTH_STAR, 2 * cnoise[q]), 1.0), (name, i, q, errs[q], cnoise[q])
            if cfg.k4 != 0.0:
                e, bar = res_rot_error(nz, i, rot[3])
                assert within("%s/%s/vs_ref64/res_rot_abs_over_bar" % (tag, name), e / bar, 1.0), (name, i, e, bar)
    finally:
        batch.close()
    within("%s/%s/knife_edge_rows_replaced" % (tag, name), replaced, 2)
    within("%s/%s/per_iteration/share_of_checks_over_2x_noise" % (tag, name), 1.0 - strict_ok / max(strict_all, 1), 0.1)
    return replaced


def free_running_bars(z, nz):
    """(ref64 finals, bars): twice the largest distance of the reference's own float32 free-running results (nine variants + the
    committed fixture) from its float64 result, never below 1e-4 -- t_cam_obj relative to its largest entry, code absolute, loss
    relative"""
    T64, c64, l64 = nz["free_T64"], nz["free_code64"], float(nz["free_loss64"])
    Ts = list(nz["free_T32"]) + [z["out_t_cam_obj"]]
    cs = list(nz["free_code32"]) + [z["out_code"]]
    ls = [float(v) for v in nz["free_loss32"]] + [float(z["loss"])]
    bars = dict(T=max(NORTH_STAR, 2 * max(relerr(t, T64) for t in Ts)),
                code=max(NORTH_STAR, 2 * max(float(np.abs(np.asarray(c, np.float64) - c64).max()) for c in cs)),
                loss=max(NORTH_STAR, 2 * max(abs(v / l64 - 1) for v in ls)))
    return (T64, c64, l64), bars


def check_free_running(tag, name, golden_dir, r, within):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    (T64, c64, l64), bars = free_running_bars(z, load(golden_dir, name))
    assert within("%s/%s/free_running_vs_ref64_over_max(1e-4,2x_reference_scatter)/t_cam_obj" % (tag, name),
                  relerr(r.t_cam_obj, T64) / bars["T"], 1.0)
    assert within("%s/%s/free_running_vs_ref64_over_max(1e-4,2x_reference_scatter)/code_abs" % (tag, name),
                  float(np.abs(r.code - c64).max()) / bars["code"], 1.0)
    assert within("%s/%s/free_running_vs_ref64_over_max(1e-4,2x_reference_scatter)/loss_rel" % (tag, name),
                  abs(r.loss / l64 - 1) / bars["loss"], 1.0)
