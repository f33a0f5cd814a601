"""Measured parity margins of the HIP paths, written as JSON (run on the GPU box; the copy judged is profiles/r02_parity.json).

  python tools/parity_report.py [out.json]

Path A, per golden case (fixtures produced by RUNNING the reference, oracle/gen_golden_sdf.py) and per Gauss-Newton
iteration, teacher-forced from the reference's own state: K and n_valid equal?, relative error (max |a-b| / max |b|) of H, b,
dx, next T_oc, next code.  Beside dx: how far the REFERENCE's own dx (float32 torch.inverse on the CPU,
reconstruct/optimizer.py:254) is from the float64 solution of the reference's own H, b -- the noise floor any float32
comparison of dx has.  Row-wise: fraction of Jacobian rows off by more than 1e-5 / 1e-3 (ReLU knife edges).  Free-running:
error of the final t_cam_obj / code / loss after all iterations.  Pose-only: error of the final SE3.
Path B: a set of seeded scenes through the two-stage local joint BA against the C oracle: index tables equal, LM paths equal,
relative error of the chi2 trace and of the final key-frame / point / object estimates.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo                                     # noqa: E402
from oracle import sdf_oracle as so                                    # noqa: E402
from qsp_slam_amd import DeepSdfDecoder, synth                         # noqa: E402
from qsp_slam_amd.ba import BaProblem                                  # noqa: E402
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg   # noqa: E402
from tests.test_gpu_sdf import make_cfg                                # noqa: E402
from tests.test_oracle_sdf import JOINT_CASES, cfg_from, relerr        # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def row_fractions(got, ref):
    ref = np.asarray(ref, np.float64).reshape(got.shape[0], -1)
    d = np.abs(np.asarray(got, np.float64) - ref).max(1) / max(np.abs(ref).max(), 1e-30)
    return dict(rows=int(got.shape[0]), frac_gt_1e5=float((d > 1e-5).mean()), frac_gt_1e3=float((d > 1e-3).mean()),
                median=float(np.median(d)), worst_good_row=float(d[d <= 1e-3].max()) if (d <= 1e-3).any() else None)


def path_a(dec, odec):
    out = {}
    for name in JOINT_CASES:
        z = np.load(os.path.join(GOLD, name + ".npz"))
        cfg = cfg_from(z)
        opt = Optimizer(dec, make_cfg(z))
        batch = RefineBatch(dec, _joint_cfg(opt), [z["pts"]], [z["rays"]], [z["depth"]], [0])
        batch.enable_rows(True)
        its = []
        n_it = z["it_H"].shape[0]
        for i in range(n_it):
            T_co = np.linalg.inv(z["it_T_oc"][i].astype(np.float64)).astype(np.float32)
            batch.set_state(T_co[None], z["it_code"][i][None])
            batch.run(1)
            tr = batch.trace()
            T, code, loss, good = batch.get()
            H64, b64 = z["it_H"][i].astype(np.float64), z["it_b"][i].astype(np.float64)
            dx64 = np.linalg.solve(H64, b64)
            rec = dict(K_equal=bool(int(tr["K"][0]) == int(z["it_K"][i])), K=int(z["it_K"][i]),
                       H=relerr(tr["H"][0], z["it_H"][i]), b=relerr(tr["b"][0], z["it_b"][i]),
                       dx=relerr(tr["dx"][0], z["it_dx"][i]),
                       dx_vs_f64_solution_of_reference_system=relerr(tr["dx"][0], dx64),
                       reference_dx_vs_f64_solution_of_its_own_system=relerr(z["it_dx"][i], dx64),
                       cond_H=float(np.linalg.cond(H64)))
            if i + 1 < n_it:
                rec["T_oc_next"] = relerr(np.linalg.inv(T[0].astype(np.float64)), z["it_T_oc"][i + 1])
                rec["code_next_abs"] = float(np.abs(code[0] - z["it_code"][i + 1]).max())
            # the oracle's rows from the same state, for the row-wise view
            dobs = np.concatenate([z["depth"], np.zeros(z["rays"].shape[0] - z["depth"].shape[0], np.float32)])
            it = so.gn_iteration(odec, cfg, z["it_T_oc"][i], z["it_code"][i], z["pts"], z["rays"], dobs, z["depth"].shape[0])
            if it["fail"] is None and it["K"] == int(tr["K"][0]):
                m = z["pts"].shape[0]
                rs, rr = batch.rows(0, m, it["K"])
                ref_s = np.concatenate([it["Jp_sdf"].reshape(m, -1), it["Jc_sdf"].reshape(m, -1)], axis=1)
                ref_r = np.concatenate([it["Jp_render"].reshape(it["K"], -1), it["Jc_render"].reshape(it["K"], -1)], axis=1)
                rec["rows_sdf_vs_oracle"] = row_fractions(rs[:, :71], ref_s)
                rec["rows_render_vs_oracle"] = row_fractions(rr[:, :71], ref_r)
            its.append(rec)
        batch.close()
        r = opt.reconstruct_object(z["t_cam_obj"], z["pts"], z["rays"], z["depth"])
        free = dict(is_good_equal=bool(r.is_good == bool(z["is_good"])))
        if r.is_good:
            free.update(t_cam_obj=relerr(r.t_cam_obj, z["out_t_cam_obj"]), code_abs=float(np.abs(r.code - z["out_code"]).max()),
                        loss_rel=float(abs(r.loss - float(z["loss"])) / abs(float(z["loss"]))))
        out[name] = dict(k4=float(cfg.k4), iterations=its, free_running=free,
                         worst=dict(H=max(x["H"] for x in its), b=max(x["b"] for x in its), dx=max(x["dx"] for x in its),
                                    T_oc_next=max(x.get("T_oc_next", 0.0) for x in its),
                                    code_next_abs=max(x.get("code_next_abs", 0.0) for x in its)))
    z = np.load(os.path.join(GOLD, "sdf_pose_only_m250.npz"))
    opt = Optimizer(dec, make_cfg(so.JointConfig()))
    got = opt.estimate_pose_cam_obj(z["t_co_se3"], float(z["scale"]), z["pts"], z["code"])
    out["sdf_pose_only_m250"] = dict(t_co=relerr(got, z["out"]))
    z = np.load(os.path.join(GOLD, "sdf_decoder_vectors.npz"))
    y, g = dec.sdf_value_grad(z["code"], z["x"])
    out["sdf_decoder_vectors"] = dict(sdf_abs=float(np.abs(dec.decode_sdf(z["code"], z["x"]) - z["sdf"]).max()),
                                      y_abs=float(np.abs(y - z["y"]).max()), grad_rows=row_fractions(g, z["grad"]))
    return out


BA_SCENES = [dict(seed=11, n_kf=6, n_pt=150, n_obj=2), dict(seed=12, n_kf=10, n_pt=400, n_obj=3, stereo_frac=0.3, outlier_frac=0.06),
             dict(seed=13, n_kf=20, n_pt=2000, n_obj=8, stereo_frac=0.2), dict(seed=14, n_kf=8, n_pt=300, n_obj=0, stereo_frac=0.5),
             dict(seed=15, n_kf=12, n_pt=500, n_obj=4, n_fixed=2), dict(seed=16, n_kf=50, n_pt=5000, n_obj=64, stereo_frac=0.2)]


def path_b():
    out = []
    for kw in BA_SCENES:
        sc = synth.make_ba_scene(**kw)
        ref = bo.BaProblem(sc)
        r1, r2 = ref.local_joint_ba()
        gpu = BaProblem(sc)
        g1, g2 = gpu.local_joint_ba()
        rec = dict(scene=kw, lm_path_equal=bool(list(g1["trials"]) == list(r1["trials"]) and list(g2["trials"]) == list(r2["trials"])
                                                and list(g1["accepted"]) == list(r1["accepted"]) and list(g2["accepted"]) == list(r2["accepted"])))
        m1, m2 = min(len(g1["chi2"]), len(r1["chi2"])), min(len(g2["chi2"]), len(r2["chi2"]))
        rec["chi2_rel"] = float(max(np.abs(g1["chi2"][:m1] / r1["chi2"][:m1] - 1).max(), np.abs(g2["chi2"][:m2] / r2["chi2"][:m2] - 1).max()))
        rec["lambda_rel"] = float(max(np.abs(g1["lam"][:m1] / r1["lam"][:m1] - 1).max(), np.abs(g2["lam"][:m2] / r2["lam"][:m2] - 1).max()))
        kf, pt, ob = gpu.state()
        rkf, rpt, rob = ref.state()
        rec["kf_pose_rel"] = relerr(kf, rkf)
        rec["points_rel"] = relerr(pt, rpt)
        rec["obj_pose_rel"] = relerr(ob, rob) if len(rob) else None
        kh, oh, ph = gpu.index()
        rec["index_tables_equal"] = bool(np.array_equal(kh, r2["kf_hidx"]) and np.array_equal(oh, r2["obj_hidx"]) and np.array_equal(ph, r2["pt_hidx"]))
        gpu.close()
        out.append(rec)
    return out


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_parity.json")
    dec = DeepSdfDecoder.from_npz(os.path.join(GOLD, "decoder_8x512.npz"))
    odec = so.load_decoder_npz(os.path.join(GOLD, "decoder_8x512.npz"))
    rep = dict(_comment="measured on MI355X by tools/parity_report.py; relative error = max|a-b| / max|b|; path A against the "
                        "reference-generated fixtures in tests/golden, path B against oracle/ba_oracle.c",
               path_a=path_a(dec, odec), path_b=path_b())
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    for name, c in rep["path_a"].items():
        if "worst" in c:
            print(name, {k: "%.2e" % v for k, v in c["worst"].items()}, "free", {k: (("%.2e" % v) if isinstance(v, float) else v)
                                                                                 for k, v in c["free_running"].items()})
            print("   dx noise floor (reference dx vs f64 solve of its own system):",
                  ["%.1e" % x["reference_dx_vs_f64_solution_of_its_own_system"] for x in c["iterations"]],
                  "ours vs same:", ["%.1e" % x["dx_vs_f64_solution_of_reference_system"] for x in c["iterations"]])
    for r in rep["path_b"]:
        print(r["scene"], "chi2 %.1e kf %.1e pt %.1e lm_equal %s" % (r["chi2_rel"], r["kf_pose_rel"], r["points_rel"], r["lm_path_equal"]))


if __name__ == "__main__":
    main()
