"""GPU box: dump what one teacher-forced Gauss-Newton iteration of every joint golden case produces on each decoder pipe -- H, b,
dx, next state, rotation terms and the augmented Jacobian rows -- into gpurun_out/<out>.npz, for the row-wise analysis against
the reference's float64 evaluation that oracle/noise_rows.py does in the build container (where the reference is).

    python tools/noise_dump.py gpurun_out/r4_noise_dump.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qsp_slam_amd import DeepSdfDecoder                                                    # noqa: E402
from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg         # noqa: E402
from tests.test_gpu_sdf import make_cfg                                                    # noqa: E402
from tests.test_oracle_sdf import JOINT_CASES                                              # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
out = {}
for prec in ("f32", "fp16x2"):
    dec = DeepSdfDecoder.from_npz(os.path.join(GOLD, "decoder_8x512.npz"))
    dec.set_precision(prec)
    for name in JOINT_CASES:
        z = np.load(os.path.join(GOLD, name + ".npz"))
        opt = Optimizer(dec, make_cfg(z))
        batch = RefineBatch(dec, _joint_cfg(opt), [z["pts"]], [z["rays"]], [z["depth"]], [0])
        batch.enable_rows(True)
        for i in range(z["it_H"].shape[0]):
            T_co = np.linalg.inv(z["it_T_oc"][i].astype(np.float64)).astype(np.float32)
            batch.set_state(T_co[None], z["it_code"][i][None])
            batch.run(1)
            tr = batch.trace()
            T, code, loss, good = batch.get()
            rs, rr = batch.rows(0, z["pts"].shape[0], int(tr["K"][0]))
            key = "%s/%s/%d/" % (prec, name, i)
            out[key + "H"], out[key + "b"], out[key + "dx"] = tr["H"][0], tr["b"][0], tr["dx"][0]
            out[key + "T_co_next"], out[key + "code_next"] = T[0], code[0]
            out[key + "rot"] = batch.trace_rot()[0]
            out[key + "rows_sdf"], out[key + "rows_render"] = rs.copy(), rr.copy()
        batch.close()
    dec.close()
np.savez_compressed(sys.argv[1], **out)
print("wrote", sys.argv[1], len(out), "arrays")
