#!/usr/bin/env python3
"""bench.py -- joint object-optimisation hot path on MI355X, one process per GPU.

A "step" is one pass of the hot path over one synthetic scene that is already resident in HBM:
  A. DeepSDF refinement of every object: flip_sample_num (4) yaw hypotheses x n_iter (5) Gauss-Newton iterations
     (reference: 4 serial calls of Optimizer.reconstruct_object per object, src/LocalMapping_util.cc:705-760)
  B. local joint bundle adjustment of the scene: optimize(5) + outlier pass + optimize(10)
     (reference: Optimizer::LocalJointBundleAdjustment, src/Optimizer_util.cc:309-771)      [when built]

metric  "joint-opt iters/sec (BA+SDF)": one joint-opt iteration = one Gauss-Newton iteration of one object hypothesis
        (71 unknowns, SDF + render terms) or one Levenberg-Marquardt iteration of the scene's BA; value = all such
        iterations of all ranks / wall time of the timed steps (max over ranks).  `ms_per_object_refine` is reported beside.
scaling weak: every rank owns its own scene of the same size (objects are independent units; no data-path collective
        in A; B's shared camera block is reduced with one RCCL all-reduce per LM trial when ranks share a scene).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|c5] [--no-cpu-baseline]
  (N > 1: launched by torch.distributed.run, one rank per GPU)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_kf, n_obj, n_pts, n_fg, n_bg, n_iter, cfg-name)
    "c2": dict(n_kf=20, n_obj=8, n_pts=2000, n_fg=256, n_bg=200, n_iter=5, n_map=2000,
               desc="C2: synthetic 20 KF / 8 objects / 2k SDF samples, 4 yaw flips x 5 GN iterations"),
    "c4": dict(n_kf=50, n_obj=64, n_pts=8000, n_fg=256, n_bg=200, n_iter=5, n_map=5000,
               desc="C4: synthetic 50 KF / 64 objects / 8k SDF samples, 4 yaw flips x 5 GN iterations"),
    "c5": dict(n_kf=200, n_obj=256, n_pts=250, n_fg=250, n_bg=200, n_iter=10, n_map=20000,
               desc="C5 stand-in: synthetic 200 KF / 256 objects / 250 LiDAR points, 4 yaw flips x 10 GN iterations"),
}
DM, DS, DO = float(np.float32(np.sqrt(5.991))), float(np.float32(np.sqrt(7.815))), float(np.float32(np.sqrt(1e3)))
FLOP_FWD = 2.0 * 1835520          # per point, decoder forward            
FLOP_FWDBWD = 2.0 * FLOP_FWD      # forward + backward-data
PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"


def joint_cfg(n_iter):
    from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict
    # configs/config_redwood_chair_01053.json (the Redwood/RGB-D weights of the reference)
    return ForceKeyErrorDict(data_type="Redwood", optimizer=dict(
        code_len=64, num_depth_samples=50, cut_off_threshold=0.01,
        joint_optim=dict(k1=10.0, k2=100.0, k3=2.5, k4=0.0, b1=0.2, b2=0.02, learning_rate=1.0, scale_damping=100.0,
                         num_iterations=n_iter)))


def flip_states(objs, flips):
    from qsp_slam_amd.reconstruct.optimizer import _flip_rotation
    T0, hyp = [], []
    for i, o in enumerate(objs):
        T = o["t_cam_obj"]
        for k in range(flips):
            T0.append(_flip_rotation(T, k, 2.0 * np.pi / flips))     # src/LocalMapping_util.cc:722-726
            hyp.append(i)
    return np.stack(T0), hyp


def cpu_baseline(w, objs, scene, n_hyp, budget_s=20.0):
    """The oracle (kind "port": numpy restatement of path A, C restatement of path B) timed on this box's host cores on
    a bounded sample of the same workload: whole Gauss-Newton iterations of single hypotheses until ~budget_s of CPU time
    is spent, plus ONE full local joint BA of the scene (single thread, as the reference's g2o runs).  `value` is the
    step rate extrapolated from that sample to the full step (all hypotheses + the BA)."""
    from oracle import ba_oracle as bo
    from oracle import sdf_oracle as so
    dec = so.load_decoder_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
    cfg = so.JointConfig(n_iter=w["n_iter"])
    cores = len(os.sched_getaffinity(0))
    t0 = time.time()
    iters = 0
    hyps = 0
    for o in objs:
        T_oc = np.linalg.inv(o["t_cam_obj"].astype(np.float64)).astype(np.float32)
        z = np.zeros(64, np.float32)
        dobs = np.concatenate([o["depth"], np.zeros(o["rays"].shape[0] - o["depth"].shape[0], np.float32)])
        for _ in range(cfg.n_iter):
            it = so.gn_iteration(dec, cfg, T_oc, z, o["pts"], o["rays"], dobs, o["depth"].shape[0])
            if it["fail"] is not None:
                break
            T_oc, z = it["T_oc_new"], it["code_new"]
            iters += 1
        hyps += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    t1 = time.time()
    prob = bo.BaProblem(scene)
    n_pose_blocks = int((~scene["kf_fixed"].astype(bool)).sum()) + len(scene["obj_pose"])
    if n_pose_blocks <= 150:
        b1, b2 = prob.local_joint_ba()
        dt_ba = time.time() - t1
        ba_iters = int(b1["iterations"] + b2["iterations"])
        ba_note = "one full local joint BA"
    else:
        # the restatement solves the reduced system densely (O(dim^3) per trial): bound the sample to one LM iteration
        # and scale to the 5+10 schedule's usual 11 iterations
        tr = prob.optimize(1, DM, DS, DO)
        ba_iters = 11
        dt_ba = (time.time() - t1) * ba_iters / max(int(tr["iterations"]), 1)
        ba_note = "ONE LM iteration of the joint BA scaled to 11"
    step_iters = n_hyp * cfg.n_iter + ba_iters
    step_s = (dt / max(iters, 1)) * n_hyp * cfg.n_iter + dt_ba
    return dict(value=step_iters / step_s, unit="iters/s", cores=cores, kind="port",
                sample="A: %d hypotheses x %d GN iterations (numpy+BLAS oracle, %d threads) in %.1f s, extrapolated to %d "
                       "hypotheses; B: %s (C oracle, 1 thread, %d LM iterations) in %.2f s"
                       % (hyps, cfg.n_iter, cores, dt, n_hyp, ba_note, ba_iters, dt_ba),
                sdf_iters_per_s=iters / dt, ba_ms=1e3 * dt_ba)


def pmc_traffic(workload, kernel):
    """Memory-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (PMC counters cannot be read from inside the process); None when the workload was not profiled."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")) as f:
            return json.load(f)[workload][kernel]["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--flips", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # QSP_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a one-GPU box -- all ranks share device 0 and the two
    # scalar reductions go over gloo.  Never set by the driver; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("QSP_BENCH_REHEARSAL") == "1"
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = local_rank if (world > 1 and not rehearsal) else 0
    red_dev = "cpu" if rehearsal else "cuda:%d" % dev

    from qsp_slam_amd import DeepSdfDecoder, synth
    from qsp_slam_amd.ba import BaProblem
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg

    w = WORKLOADS[args.workload]
    dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"), device=dev)
    objs = synth.make_object_views(1000 + rank, w["n_obj"], w["n_pts"], n_fg=w["n_fg"], n_bg=w["n_bg"])
    opt = Optimizer(dec, joint_cfg(w["n_iter"]))
    T0, hyp = flip_states(objs, args.flips)
    batch = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs],
                        [o["depth"] for o in objs], hyp)          # inputs resident in HBM from here on
    batch.profile(True)
    scene = synth.make_ba_scene(2000 + rank, w["n_kf"], w["n_map"], w["n_obj"], stereo_frac=0.2)
    ba = BaProblem(scene, device=dev)                              # flattened graph resident in HBM
    ba.profile(True)
    kf0, pt0, ob0 = scene["kf_pose"], scene["pt_xyz"], scene["obj_pose"]
    ba_stat = dict(ms=0.0, ms_lin=0.0, n_lin=0, bytes_lin=0, iters=0, trials=0)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def step(record=False):
        batch.set_state(T0, None)      # 256 x (16+64) floats H2D: part of the step, as the caller hands poses over
        batch.run(0)
        ba.set_state(kf0, pt0, ob0)    # the caller's current estimates (float64 poses / points)
        t_a = time.perf_counter()
        t1, t2 = ba.local_joint_ba()
        if record:
            ba_stat["ms"] += 1e3 * (time.perf_counter() - t_a)
            ba_stat["iters"] += int(t1["iterations"] + t2["iterations"])
            ba_stat["trials"] += int(t1["trials"].sum() + t2["trials"].sum())
            bp = ba.profile(True)      # stage-2 call's record
            ba_stat["ms_lin"] += bp.ms_linearize
            ba_stat["n_lin"] += bp.n_linearize
            ba_stat["bytes_lin"] = bp.bytes_linearize

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    prof = dict(ms_total=0.0, ms_mlp_jtj=0.0, ms_mlp_fwd=0.0, ms_other=0.0, n_jtj=0, n_fwd=0, pts_jtj=0, pts_fwd=0,
                tiles_jtj=0, tiles_fwd=0)
    for _ in range(args.steps):
        step(record=True)
        p = batch.profile(True)
        prof["ms_total"] += p.ms_total
        prof["ms_mlp_jtj"] += p.ms_mlp_jtj
        prof["ms_mlp_fwd"] += p.ms_mlp_fwd
        prof["ms_other"] += p.ms_other
        prof["n_jtj"] += p.n_launch_jtj
        prof["n_fwd"] += p.n_launch_fwd
        prof["pts_jtj"] += p.pts_jtj
        prof["pts_fwd"] += p.pts_fwd
        prof["tiles_jtj"] += p.tiles_jtj
        prof["tiles_fwd"] += p.tiles_fwd
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    _, _, loss, good = batch.get()

    n_hyp = len(hyp)
    iters_total = n_hyp * w["n_iter"] * args.steps + ba_stat["iters"]
    if dist is not None:
        t = torch.tensor([float(iters_total)], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        iters_total_all = float(t.item())
    else:
        iters_total_all = float(iters_total)
    value = iters_total_all / dt
    if rank == 0:
        flop_jtj = FLOP_FWDBWD * prof["pts_jtj"] + 2.0 * 72 * 72 * prof["pts_jtj"]
        avg_ms = prof["ms_mlp_jtj"] / max(prof["n_jtj"], 1)
        achieved = flop_jtj / max(prof["n_jtj"], 1) / (avg_ms * 1e-3) / 1e12
        fwd_tf = FLOP_FWD * prof["pts_fwd"] / max(prof["ms_mlp_fwd"] * 1e-3, 1e-9) / 1e12
        out = {
            "metric": "joint-opt iters/sec (BA+SDF)", "value": value, "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded scene, decoder fitted to an analytic shape family)",
            "config": {"workload": w["desc"], "objects_per_gpu": w["n_obj"], "hypotheses_per_gpu": n_hyp,
                       "surface_points": w["n_pts"], "rays": w["n_fg"] + w["n_bg"], "depth_samples": 50,
                       "gn_iterations": w["n_iter"],
                       "ba": "local joint BA 5+10 LM iterations: %d KF / %d map points / %d objects, %d mono + %d stereo "
                             "+ %d camera-object edges" % (w["n_kf"], w["n_map"], w["n_obj"], len(scene["mono_pt"]),
                                                            len(scene["st_pt"]), len(scene["oe_kf"]))},
            "ms_per_object_refine": 1e3 * dt / args.steps / w["n_obj"],
            "good_hypotheses": int(good.sum()),
            "roofline": {"bound": "mfma", "kernel": "k_mlp_jtj (decoder fwd+bwd+JtJ, f32 MFMA)",
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(args.workload, "k_mlp_jtj"),
                         "traffic_unit": "bytes/launch, rocprofv3 PMC passes of this command (profiles/r01_traffic.json)",
                         "avg_launch_ms": avg_ms, "launches": prof["n_jtj"],
                         "points_per_launch": prof["pts_jtj"] / max(prof["n_jtj"], 1),
                         "tile_padding_overhead": 64.0 * prof["tiles_jtj"] / max(prof["pts_jtj"], 1)},
            "kernels": {"k_mlp_fwd_TFLOPs": fwd_tf, "k_mlp_fwd_frac": fwd_tf / PEAK_F32_MFMA_TFLOPS,
                        "ms_mlp_jtj": prof["ms_mlp_jtj"] / args.steps, "ms_mlp_fwd": prof["ms_mlp_fwd"] / args.steps,
                        "ms_other": prof["ms_other"] / args.steps, "ms_gpu_total": prof["ms_total"] / args.steps,
                        "ms_ba": ba_stat["ms"] / args.steps, "ba_lm_iterations": ba_stat["iters"] / args.steps,
                        "ba_lm_trials": ba_stat["trials"] / args.steps,
                        "ba_linearize_us": 1e3 * ba_stat["ms_lin"] / max(ba_stat["n_lin"], 1),
                        "ba_linearize_bytes": ba_stat["bytes_lin"],
                        "ba_linearize_GBps": ba_stat["bytes_lin"] / max(1e-3 * ba_stat["ms_lin"] / max(ba_stat["n_lin"], 1), 1e-12) / 1e9,
                        "ba_linearize_frac_of_8TBps": ba_stat["bytes_lin"] / max(1e-3 * ba_stat["ms_lin"] / max(ba_stat["n_lin"], 1), 1e-12) / 8e12},
        }
        if world == 1:
            # the BA linearisation kernels are launch-bound at the BASELINE sizes (7.5 MB per build); their bandwidth is
            # measured on a graph large enough to stream: 64 key-frames x 250 000 landmarks x 8 observations
            big = synth.make_ba_scene_large(7, 64, 250000)
            bb = BaProblem(big, device=dev)
            bb.profile(True)
            for _ in range(2):
                bb.set_state(big["kf_pose"], big["pt_xyz"], big["obj_pose"])
                bb.optimize(2, 0, 0, 0)
                st = bb.profile(True)
            us = 1e3 * st.ms_linearize / max(st.n_linearize, 1)
            out["kernels"]["ba_linearize_large"] = {
                "graph": "64 KF / 250000 landmarks / %d mono edges" % len(big["mono_pt"]), "us": us,
                "algorithmic_bytes": int(st.bytes_linearize), "GBps": st.bytes_linearize / us / 1e3,
                "frac_of_8TBps": st.bytes_linearize / us / 1e3 / 8000.0}
            bb.close()
            # the boundary as the reference calls it: host buffers in, host results out (upload + allocation inside the
            # step); reported beside `value`, never as `value`
            th = time.perf_counter()
            b2 = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs],
                             [o["depth"] for o in objs], hyp)
            b2.set_state(T0, None)
            b2.run(0)
            b2.get()
            ba2 = BaProblem(scene, device=dev)
            ba2.local_joint_ba()
            ba2.state()
            th = time.perf_counter() - th
            out["host_buffers"] = {"ms_per_step": 1e3 * th, "value": iters_total / args.steps / th, "unit": "iters/s",
                                   "note": "one step with observations, graph and results crossing PCIe and device "
                                           "buffers allocated inside the step"}
            b2.close()
            ba2.close()
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(w, objs, scene, n_hyp)
        print(json.dumps(out))
    batch.close()
    ba.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
