"""TEST INFRASTRUCTURE ONLY -- produces tests/golden/decoder_8x512.npz.

The reference ships no DeepSDF weights, so the published DSP-SLAM decoder layout
    Decoder(64, [512]*8, dropout=range(8), dropout_prob=0.2, norm_layers=range(8), latent_in=[4],
            weight_norm=True, xyz_in_all=False, use_tanh=False, latent_dropout=False)
is instantiated with the reference's own class (deep_sdf/deep_sdf_decoder.py:9-110) and fitted, with a fixed
seed, to the analytic shape family of qsp_slam_amd/synth.py.  The file stores the *raw* parameters
(weight_v, weight_g, bias per layer, exactly the state_dict a DeepSDF checkpoint holds,
deep_sdf/workspace.py:202-224) so that weight-norm folding is exercised by the loader under test.

Run in the build container only:  python oracle/fit_decoder.py [--steps 3000]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402
from qsp_slam_amd import synth  # noqa: E402


def make_batch(rng, n_codes, n_per, code_len=64):
    codes = np.zeros((n_codes, code_len))
    codes[:, :3] = rng.normal(scale=0.25, size=(n_codes, 3))
    codes[:, 3:] = rng.normal(scale=0.05, size=(n_codes, code_len - 3))
    xs, zs = [], []
    for c in codes:
        n_u = n_per // 2
        xu = rng.uniform(-1, 1, size=(n_u, 3))
        a = synth.shape_axes(c)
        u = rng.normal(size=(n_per - n_u, 3))
        u /= np.linalg.norm(u, axis=-1, keepdims=True)
        bump = rng.random(n_per - n_u) < 0.15
        cen = np.array([0.9 * a[0], 0.35 * a[1], 0.0])
        xs_ = np.where(bump[:, None], cen + synth.BUMP_RADIUS * u, a * u)
        xs_ = xs_ + rng.normal(scale=0.03, size=xs_.shape)
        x = np.concatenate([xu, xs_], 0)
        xs.append(x)
        zs.append(np.broadcast_to(c, (n_per, code_len)))
    x = np.concatenate(xs, 0)
    z = np.concatenate(zs, 0)
    y = synth.analytic_sdf(x, z)
    inp = np.concatenate([z, x], -1).astype(np.float32)
    return torch.from_numpy(inp), torch.from_numpy(y.astype(np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--arch", default="8x512", choices=["8x512", "4x256_c32"],
                    help="8x512: the DSP-SLAM layout (code 64, latent_in [4]); 4x256_c32: a second member of the family the "
                         "reference's Decoder class builds (deep_sdf_decoder.py:29-63) -- 4 hidden layers of 256, code 32 "
                         "(the code_len == 32 branch of src/LocalMapping_util.cc:789-800), latent_in [2]")
    args = ap.parse_args()
    L, dims, lin = (64, [512] * 8, [4]) if args.arch == "8x512" else (32, [256] * 4, [2])
    if args.out is None:
        args.out = os.path.join(os.path.dirname(HERE), "tests", "golden", "decoder_%s.npz" % args.arch)

    _, _, _, dec_mod, _ = import_reference()
    torch.manual_seed(20261003)
    torch.set_num_threads(6)
    rng = np.random.default_rng(20261003)
    dec = dec_mod.Decoder(L, dims, dropout=list(range(len(dims))), dropout_prob=0.2, norm_layers=list(range(len(dims))),
                          latent_in=lin, weight_norm=True, xyz_in_all=False, use_tanh=False, latent_dropout=False)
    dec.eval()  # dropout off: plain regression
    opt = torch.optim.Adam(dec.parameters(), lr=5e-4)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=max(args.steps // 4, 1), gamma=0.5)
    clamp = 0.1
    t0 = time.time()
    for step in range(args.steps):
        inp, y = make_batch(rng, 32, 256, L)
        pred = dec(inp).squeeze(-1)
        loss = (torch.clamp(pred, -clamp, clamp) - torch.clamp(y, -clamp, clamp)).abs().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if step % 100 == 0 or step == args.steps - 1:
            print("step %5d  loss %.5f  %.0fs" % (step, loss.item(), time.time() - t0), flush=True)
    sd = {k: v.detach().numpy() for k, v in dec.state_dict().items()}
    meta = dict(latent_size=L, dims=dims, latent_in=lin, weight_norm=True, norm_layers=list(range(len(dims))),
                xyz_in_all=False, use_tanh=False)
    np.savez(args.out, meta=np.array(repr(meta)), **sd)
    print("wrote", args.out, {k: v.shape for k, v in sd.items()})


if __name__ == "__main__":
    main()
