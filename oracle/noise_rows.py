"""TEST INFRASTRUCTURE ONLY (build container; needs /root/reference) -- row-wise view of a GPU dump (tools/noise_dump.py) against
the reference's float64 evaluation of the same teacher-forced iteration: which Jacobian rows differ, by how much, and how much of
the error in H, b, dx they explain.

    python oracle/noise_rows.py gpurun_out/r4_noise_dump.npz [case] [iteration]
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.gen_golden_sdf import GOLD, ref_decoder  # noqa: E402
from oracle.ref_import import import_reference  # noqa: E402


def ref_rows(mods, dec, z, i, dtype):
    """[J_pose | J_code | robust residual] rows of the surface and the render term from the reference's own loss functions"""
    opt_mod, loss_mod, lu, dec_mod, utils_mod = mods
    joint = ast.literal_eval(str(z["joint"]))
    npdt = np.float64 if dtype == torch.float64 else np.float32
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        T_oc = torch.from_numpy(z["it_T_oc"][i].astype(npdt))
        code = torch.from_numpy(z["it_code"][i].astype(npdt))
        jp, jc, res = loss_mod.compute_sdf_loss(dec, torch.from_numpy(z["pts"].astype(npdt)), T_oc, code)
        rob, _, _ = lu.get_robust_res(res, joint["b2"])
        rs = torch.cat([jp.squeeze(1), jc.squeeze(1), rob.reshape(-1, 1)], 1).numpy()
        T_co = torch.inverse(T_oc)
        scale = torch.det(T_co[:3, :3]) ** (1 / 3)
        dmin, dmax = T_co[2, 3] - scale, T_co[2, 3] + scale
        depths = torch.linspace(dmin, dmax, 50)
        n_fg = z["depth"].shape[0]
        dobs = torch.from_numpy(np.concatenate([z["depth"], np.zeros(z["rays"].shape[0] - n_fg, np.float32)]).astype(np.float32))
        dobs[n_fg:] = 1.1 * dmax
        rr = loss_mod.compute_render_loss(dec, torch.from_numpy(z["rays"].astype(npdt)), dobs, T_oc, depths, code, th=0.01)
        rob_r, _, _ = lu.get_robust_res(rr[2], joint["b1"])
        rrow = torch.cat([rr[0].squeeze(1), rr[1].squeeze(1), rob_r.reshape(-1, 1)], 1).numpy()
    finally:
        torch.set_default_dtype(old)
    return rs, rrow


def main():
    d = np.load(sys.argv[1])
    only = sys.argv[2] if len(sys.argv) > 2 else None
    only_i = int(sys.argv[3]) if len(sys.argv) > 3 else None
    mods = import_reference()
    dec32 = ref_decoder(mods[3], os.path.join(GOLD, "decoder_8x512.npz"))
    dec64 = ref_decoder(mods[3], os.path.join(GOLD, "decoder_8x512.npz")).double()
    keys = sorted({k.rsplit("/", 1)[0] for k in d.files})
    for key in keys:
        prec, name, i = key.split("/")
        i = int(i)
        if (only and only not in name) or (only_i is not None and i != only_i):
            continue
        z = np.load(os.path.join(GOLD, name + ".npz"))
        r64s, r64r = ref_rows(mods, dec64, z, i, torch.float64)
        r32s, r32r = ref_rows(mods, dec32, z, i, torch.float32)
        for term, mine, r64, r32 in (("sdf", d[key + "/rows_sdf"], r64s, r32s), ("render", d[key + "/rows_render"], r64r, r32r)):
            sc = np.abs(r64[:, :71]).max()
            dm = np.abs(mine[:, :71] - r64[:, :71]).max(1) / sc
            dr = np.abs(r32[:, :71] - r64[:, :71]).max(1) / sc
            rm = np.abs(mine[:, 71] - r64[:, 71]).max() / np.abs(r64[:, 71]).max()
            rr_ = np.abs(r32[:, 71] - r64[:, 71]).max() / np.abs(r64[:, 71]).max()
            print("%-8s %-26s it %d %-6s rows %5d  J row err vs ref64: GPU median %.1e p99 %.1e max %.1e (#>1e-5: %d)   ref32 median "
                  "%.1e p99 %.1e max %.1e (#>1e-5: %d)   residual: GPU %.1e ref32 %.1e"
                  % (prec, name, i, term, mine.shape[0], np.median(dm), np.quantile(dm, .99), dm.max(), int((dm > 1e-5).sum()),
                     np.median(dr), np.quantile(dr, .99), dr.max(), int((dr > 1e-5).sum()), rm, rr_))


if __name__ == "__main__":
    main()
