#!/usr/bin/env python3
"""bench.py -- joint object-optimisation hot path on MI355X, one process per GPU.

A "step" is one pass of the hot path over one synthetic scene that is already resident in HBM:
  A. DeepSDF refinement of every object: flip_sample_num (4) yaw hypotheses x n_iter (5) Gauss-Newton iterations
     (reference: 4 serial calls of Optimizer.reconstruct_object per object, src/LocalMapping_util.cc:705-760)
  B. local joint bundle adjustment of the scene: optimize(5) + outlier pass + optimize(10)
     (reference: Optimizer::LocalJointBundleAdjustment, src/Optimizer_util.cc:309-771)

metric  "joint-opt iters/sec (BA+SDF)": one joint-opt iteration = one Gauss-Newton iteration of one object hypothesis
        (71 unknowns, SDF + render terms) or one Levenberg-Marquardt iteration of the scene's BA; value = all such
        iterations of the job / wall time of the timed steps (max over ranks).  `ms_per_object_refine` is reported beside.

scaling strong (default; BASELINE.json configs[3] and north_star): ONE fixed scene for every N.  Object o with its four yaw
        hypotheses is refined on rank o % N (no collective inside the Gauss-Newton iterations, one all_gather of 82 floats
        per object at the end of A); the scene's joint BA is ONE solve whose landmarks are sharded pt_id % N, with the shared
        camera/object block summed by RCCL (ncclAllReduce on the library's own stream, dimp^2 + dimp doubles per LM trial).
        weak (--scaling weak): every rank owns its own scene of the same size (independent key-frame windows; replicas).

precision fp16x2 (default): the decoder's multiply-adds with every f32 operand as two fp16 terms (x = hi + 2^-11 lo'), three fp16
        products per multiply-add on the fp16 matrix pipe, f32 accumulation -- float32-equivalent (1.8e-7 against float64 on the
        SDF value; the f32 pipe: 2.0e-7), parity-gated like the f32 tile (tests/test_gpu_split_precision.py: K / n_valid exact on
        every fixture, teacher-forced H, b, dx, next state inside the same bars; out-of-range decoders are refused or fail loudly).
        bf16x3: three bf16 terms, six products.  f32: the exact-f32 matrix pipe.  The other two pipes' step times are reported
        beside `value` in every default run (`f32_mfma`, `bf16x3_mfma`).

screening (default with fp16x2, --screening-margin 0.01): the ray-sample forward pass of the refinement in two passes -- every
        sample on a one-product fp16 tile (sign and rough magnitude), the samples with |s1| < cut_off + margin again on the
        split-fp16 tile.  The render term clamps outside the band, so K, n_valid, H, b and every iterate are bit-identical to the
        one-pass result (tests/test_gpu_screening.py); `fp16x2_unscreened` in the line is the same workload with one pass.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4|c5] [--scaling strong|weak] [--precision fp16x2|bf16x3|f32]
                  [--screening-margin M] [--no-sublines] [--no-cpu-baseline]
  N > 1: one rank per GPU under torch.distributed.run (the driver's launch line); `python bench.py --gpus N` without that
  environment starts it as a child process.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# dmabuf IPC is the only mode the host driver supports: RCCL across processes fails with the legacy mode (set before any
# process touches the GPU; the launcher normally exports it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
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
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table, "Peak BF16/FP16 MFMA", dense
PRODUCTS = {"f32": 1, "bf16x3": 6, "fp16x2": 3}     # matrix-pipe products per algorithmic multiply-add
# what the fp16 pipe SUSTAINS with realistic operands and nothing else running (bare v_mfma_f32_32x32x16_f16 loop on every SIMD,
# tools/micro/mfma_rate_f16.hip -> profiles/r02_mfma_rate_f16.txt): switching power draws the clock from 2.4 to 1.6 GHz
SUSTAINED_FP16_MFMA_TFLOPS = 1668.0


def peak_for(precision):
    """peak for ALGORITHMIC flops of the decoder kernels on a pipe: the dense 16-bit MFMA peak / products per multiply-add"""
    return PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_BF16_MFMA_TFLOPS / PRODUCTS[precision]
PEAK_HBM_GBPS = 8000.0


def joint_cfg(n_iter, kitti=False):
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
    if not T0:
        return np.zeros((0, 4, 4), np.float32), hyp
    return np.stack(T0), hyp


def select_flips(T, code, loss, good, n_obj, flips):
    """the keep rule of src/LocalMapping_util.cc:748-752 over each object's hypotheses -> (n_obj, 82) result rows"""
    out = np.zeros((n_obj, 82), np.float32)
    for i in range(n_obj):
        best = i * flips
        for k in range(1, flips):
            h = i * flips + k
            if (not good[best]) or (good[h] and loss[h] < loss[best]):
                best = h
        if good[best]:
            out[i, :16] = T[best].reshape(-1)
            out[i, 16:80] = code[best][:64]
        out[i, 80] = loss[best]
        out[i, 81] = 1.0 if good[best] else 0.0
    return out


def cpu_baseline(w, objs, scene, n_hyp, budget_s=12.0):
    """The oracle (kind "port": numpy restatement of path A, C restatement of path B) timed on this box's host cores on a
    bounded sample of the same workload.  A: whole Gauss-Newton iterations of single hypotheses, once with the
    BLAS pool limited to ONE thread (the faithful baseline: the reference's Python path runs one object at a time) and once on all
    cores, ~budget_s of wall time each.  B: the FULL 5 + 10 schedule of the local joint BA, single thread like g2o, with the reduced
    camera system solved by a block-sparse minimum-degree Cholesky (the stand-in for g2o's AMD-ordered sparse LDL^T,
    linear_solver_eigen.h:94-124,147-201; oracle/ba_oracle.c) -- not the dense solve the parity tests use.  `value` is the step
    rate extrapolated from the all-core sample of A plus B; `single_core` the same with the one-thread sample of A."""
    from threadpoolctl import threadpool_limits
    from oracle import ba_oracle as bo
    from oracle import sdf_oracle as so
    dec = so.load_decoder_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"))
    cfg = so.JointConfig(n_iter=w["n_iter"])
    cores = len(os.sched_getaffinity(0))

    def sample_a(budget):
        t0 = time.time()
        iters = hyps = 0
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
                if time.time() - t0 > budget:
                    break
            hyps += 1
            if time.time() - t0 > budget:
                break
        return iters, hyps, time.time() - t0

    with threadpool_limits(limits=1):
        it1, hy1, dt1 = sample_a(budget_s)
    itn, hyn, dtn = sample_a(budget_s)
    t1 = time.time()
    bo.set_sparse_solver(True)
    try:
        prob = bo.BaProblem(scene)
        b1, b2 = prob.local_joint_ba()
    finally:
        bo.set_sparse_solver(False)
    dt_ba = time.time() - t1
    ba_iters = int(b1["iterations"] + b2["iterations"])
    step_iters = n_hyp * cfg.n_iter + ba_iters

    def rate(iters, dt):
        return step_iters / ((dt / max(iters, 1)) * n_hyp * cfg.n_iter + dt_ba)

    return dict(value=rate(itn, dtn), unit="iters/s", cores=cores, kind="port",
                sample="A: %d GN iterations of %d hypotheses (numpy+BLAS oracle, %d threads) in %.1f s, extrapolated to %d hypotheses x %d "
                       "iterations; B: the full local joint BA, %d LM iterations (C oracle, 1 thread as g2o, block-sparse minimum-degree "
                       "Cholesky of the reduced system) in %.2f s" % (itn, hyn, cores, dtn, n_hyp, cfg.n_iter, ba_iters, dt_ba),
                sdf_iters_per_s=itn / dtn, ba_ms=1e3 * dt_ba,
                single_core=dict(value=rate(it1, dt1), unit="iters/s", cores=1,
                                 sample="A: %d GN iterations in %.1f s with the BLAS pool limited to 1 thread; B as above"
                                        % (it1, dt1), sdf_iters_per_s=it1 / dt1))


def ba_latency_model(n_kf_free, ms_ba, lm_trials):
    """The BA's roof at the BASELINE sizes is not HBM (one linearisation of C5 moves 24 MB: 3 us at 8 TB/s) but the chain of
    DEPENDENT steps of one Levenberg-Marquardt trial.  A stated critical-path model, every term measured on MI355X:
      launches   9 dependent launches per trial (stage 1, pair Schur, factorisation, back-substitution, update, errors, publish, and
                 the two linearisation launches of the next system) x 1.5 us, the middle of the 1.1-1.9 us boundary cost of a
                 kernel that depends on its predecessor (guide; tools/micro/graph_launch.hip);
      chain      nb block steps of the blocked Cholesky, each at its serial minimum: the 64x64 diagonal factorisation with the
                 explicit inverse, 14.1 us (the micro-benchmark's best form, profiles/r03_chol_factor_micro.txt: a dependent
                 v_fma_f64 issues every 8.4 cycles, v_rsq_f64 every 20), plus the two 64^3 products that sit between two
                 factorisations on the same data, 2 x 5.3 k cycles at 2.4 GHz = 4.4 us (FP64 MFMA form);
      back       the triangular factor streamed once through ONE compute unit at 148 GB/s (tools/micro/cu_stream.hip, 16-byte
                 loads) + 2 us of dependent round trip per block row.
    frac = model / achieved time per trial (1.0 = at the floor of this algorithm's dependent chain)."""
    dimp = ((6 * n_kf_free + 63) // 64) * 64
    nb = dimp // 64
    t_launch = 9 * 1.5
    t_chain = nb * (14.1 + 4.4)
    t_back = (dimp * dimp * 4) / 148e3 + 2.0 * nb
    model = t_launch + t_chain + t_back
    achieved = 1e3 * ms_ba / max(lm_trials, 1)
    return dict(us_per_trial_model=model, us_per_trial_achieved=achieved, frac=model / max(achieved, 1e-9), block_rows=nb,
                dense_unknowns=dimp, terms_us=dict(launch_boundaries=t_launch, factor_chain=t_chain, back_substitution=t_back))


def pmc_traffic(workload, kernel):
    """Memory-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (PMC counters cannot be read from inside the process); None when the workload was not profiled."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f)[workload][kernel]["bytes_per_launch"], name
        except (OSError, KeyError, ValueError):
            continue
    return None, None


class Ctx(object):
    """process-wide state of one bench process: rank / world, device, torch.distributed group, the library's RCCL comm"""

    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        import torch
        self.torch = torch
        self.dist = None
        # QSP_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a one-GPU box -- all ranks share device 0, the control
        # reductions and the BA's all-reduce hook go over gloo (RCCL refuses two ranks on one device).  Never set by the
        # driver; the numbers of such a run mean nothing.
        self.rehearsal = os.environ.get("QSP_BENCH_REHEARSAL") == "1"
        if self.world > 1:
            import torch.distributed as dist
            self.dist = dist
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        self.dev = self.local_rank if (self.world > 1 and not self.rehearsal) else 0
        self.red_dev = "cpu" if (self.rehearsal or self.world == 1) else "cuda:%d" % self.dev
        self.comm = None
        self.comm_kind = "none"
        if self.world > 1 and args.scaling == "strong":
            from qsp_slam_amd import parallel
            if self.rehearsal:
                self.comm_kind = "gloo hook (rehearsal)"
            else:
                try:
                    self.comm = parallel.RcclComm(self.rank, self.world, self.dev)     # collective over all ranks
                except Exception as e:                                                  # e.g. librccl not resolvable
                    print("bench.py rank %d: library RCCL communicator failed (%r)" % (self.rank, e), file=sys.stderr)
                    self.comm = None
                ok = torch.tensor([1.0 if self.comm is not None else 0.0], device="cuda:%d" % self.dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)                               # the same decision on every rank
                if float(ok.item()) > 0.5:
                    self.comm_kind = "rccl on the library stream (qsp_ba_set_shard_rccl)"
                else:
                    if self.comm is not None:
                        self.comm.close()
                    self.comm = None
                    if not args.allow_hook_fallback:
                        # a broken primary path must not masquerade as a slow one in a scaling run: the hook costs two host
                        # synchronisations per collective and is a different code path (qsp_ba_set_shard, not ..._rccl)
                        print("bench.py rank %d: the library's RCCL communicator could not be created on every rank; refusing to "
                              "fall back to the torch.distributed hook (pass --allow-hook-fallback to measure that path)"
                              % self.rank, file=sys.stderr)
                        dist.destroy_process_group()
                        sys.exit(3)
                    self.comm_kind = "torch.distributed nccl hook (fallback: qsp_ba_set_shard + TorchAllreduce)"

    def sync_all(self):
        self.torch.cuda.synchronize(self.dev)
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def allreduce(self, x, op):
        if self.dist is None:
            return float(x)
        t = self.torch.tensor([float(x)], device=self.red_dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return float(t.item())


def run_workload(ctx, name, steps, warmup, detailed, precision=None, screening=None, audit=None, staging=None):
    """`steps` timed passes of workload `name` (after `warmup` untimed ones); every rank returns the same dict of whole-job
    numbers (rank 0's kernel timings)."""
    from qsp_slam_amd import DeepSdfDecoder, parallel, synth
    from qsp_slam_amd.ba import BaProblem
    from qsp_slam_amd.reconstruct.optimizer import Optimizer, RefineBatch, _joint_cfg
    args, rank, world, dev = ctx.args, ctx.rank, ctx.world, ctx.dev
    strong = args.scaling == "strong"
    w = WORKLOADS[name]
    flips = args.flips
    dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"), device=dev)
    dec.set_precision(precision or args.precision)
    margin = args.screening_margin if screening is None else screening
    if dec.precision == "fp16x2" and margin > 0:
        dec.set_screen_audit(args.screen_audit if audit is None else audit)
        dec.set_depth_staging(not args.no_depth_staging if staging is None else staging)
        dec.set_render_screening(margin)     # two-pass ray-sample forward, bit-identical to the unscreened pipe (tests/test_gpu_screening.py)
    opt = Optimizer(dec, joint_cfg(w["n_iter"]))
    seed_off = 0 if strong else rank
    objs_all = synth.make_object_views(1000 + seed_off, w["n_obj"], w["n_pts"], n_fg=w["n_fg"], n_bg=w["n_bg"])
    mine = parallel.shard_objects(w["n_obj"], rank, world) if strong else list(range(w["n_obj"]))
    objs = [objs_all[i] for i in mine]
    T0, hyp = flip_states(objs, flips)
    batch = None
    if objs:
        batch = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs],
                            [o["depth"] for o in objs], hyp)          # inputs resident in HBM from here on
        batch.profile(True)
    scene = synth.make_ba_scene(2000 + seed_off, w["n_kf"], w["n_map"], w["n_obj"], stereo_frac=0.2)
    ba = BaProblem(scene, device=dev)                              # flattened graph resident in HBM
    if strong and world > 1:
        if ctx.comm is not None:
            ba.set_shard_rccl(ctx.comm)
        elif ctx.rehearsal:
            ba.set_shard(rank, world, parallel.GlooAllreduce())
        else:
            ba.set_shard(rank, world, parallel.TorchAllreduce(dev))
    ba.profile(True)
    kf0, pt0, ob0 = scene["kf_pose"], scene["pt_xyz"], scene["obj_pose"]
    ba_stat = dict(ms=0.0, ms_lin=0.0, n_lin=0, bytes_lin=0, iters=0, trials=0, ms_gather=0.0)
    results = {}

    def step(record=False, lin_stats=False):
        if batch is not None:
            batch.set_state(T0, None)  # (16+64) floats per hypothesis H2D: part of the step, as the caller hands poses over
            batch.run(0)
        if strong and world > 1:       # every rank ends with the kept result of every object (82 floats each)
            t_g = time.perf_counter()
            if batch is not None:
                T, code, loss, good = batch.get()
                table = select_flips(T, code, loss, good, len(objs), flips)
            else:
                table = np.zeros((0, 82), np.float32)
            results["table"] = parallel.gather_object_results(table, w["n_obj"], rank, world,
                                                              device=None if ctx.red_dev == "cpu" else ctx.red_dev)
            if record:
                ba_stat["ms_gather"] += 1e3 * (time.perf_counter() - t_g)
        ba.set_state(kf0, pt0, ob0)    # the caller's current estimates (float64 poses / points)
        t_a = time.perf_counter()
        t1, t2 = ba.local_joint_ba()
        if record:
            ba_stat["ms"] += 1e3 * (time.perf_counter() - t_a)
            ba_stat["iters"] += int(t1["iterations"] + t2["iterations"])
            ba_stat["trials"] += int(t1["trials"].sum() + t2["trials"].sum())
        if lin_stats:
            bp = ba.profile(True)      # stage-2 call's record
            ba_stat["ms_lin"] += bp.ms_linearize
            ba_stat["n_lin"] += bp.n_linearize
            ba_stat["bytes_lin"] = bp.bytes_linearize

    # The BA's linearisation time comes from an event pair around every system it builds; an event record is a barrier packet
    # (~6 us of idle device in front of the next kernel, 12 per BA), so the pairs are taken in the warm-up steps and switched
    # off for the timed region (with --warmup 0 they stay on: there is no other step to take them from).
    lin_in_timed = warmup == 0
    for _ in range(warmup):
        step(lin_stats=True)
    if not lin_in_timed:
        ba.profile(False)
    ctx.sync_all()
    t0 = time.perf_counter()
    prof = dict(ms_total=0.0, ms_mlp_jtj=0.0, ms_mlp_fwd=0.0, ms_other=0.0, n_jtj=0, n_fwd=0, pts_jtj=0, pts_fwd=0,
                tiles_jtj=0, tiles_fwd=0, pts_band=0, pts_audit=0, audit_failures=0, fallbacks=0)
    for _ in range(steps):
        step(record=True, lin_stats=lin_in_timed)
        if batch is not None:
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
            prof["pts_band"] += p.pts_band
            prof["pts_audit"] += p.pts_audit
            prof["audit_failures"] += p.screen_audit_failures
            prof["fallbacks"] += p.range_fallbacks
            prof["screen_fallbacks"] = prof.get("screen_fallbacks", 0) + p.screen_fallbacks
            prof["screen_max_diff"] = max(prof.get("screen_max_diff", 0.0), p.screen_max_diff)
    ctx.sync_all()
    dt = ctx.allreduce(time.perf_counter() - t0, "MAX")
    n_good = 0
    if batch is not None:
        _, _, _, good = batch.get()
        n_good = int(good.sum())
    n_good = int(ctx.allreduce(n_good, "SUM"))
    n_hyp_job = (w["n_obj"] * flips) if strong else int(ctx.allreduce(len(hyp), "SUM"))
    sdf_iters = n_hyp_job * w["n_iter"] * steps
    # strong: ONE shared BA (every rank reports the same trace); weak: one BA per rank
    ba_iters_job = ba_stat["iters"] if strong else ctx.allreduce(ba_stat["iters"], "SUM")
    iters_total = sdf_iters + ba_iters_job
    flop_jtj = (FLOP_FWDBWD + 2.0 * 72 * 72) * prof["pts_jtj"]
    avg_ms = prof["ms_mlp_jtj"] / max(prof["n_jtj"], 1)
    achieved = flop_jtj / max(prof["n_jtj"], 1) / max(avg_ms * 1e-3, 1e-12) / 1e12
    fwd_tf = FLOP_FWD * prof["pts_fwd"] / max(prof["ms_mlp_fwd"] * 1e-3, 1e-9) / 1e12
    lin_us = 1e3 * ba_stat["ms_lin"] / max(ba_stat["n_lin"], 1)
    # bytes one linearisation actually moves (upper bound with every edge active): the 56-byte edges are read by the landmark
    # pass and by the key-frame pass, points 24 B in x 2 + Hll / bl 96 B out, pose blocks 56 B + K 40 B in x 2 and 336 B out,
    # camera-object edges 56 B in, 84 + 36 doubles out
    n_e = len(scene["mono_pt"]) + len(scene["st_pt"])
    moved_lin = (2 * 56 * n_e + (2 * 24 + 96) * w["n_map"] + (2 * 96 + 336) * (w["n_kf"] + w["n_obj"]) + (56 + 960) * len(scene["oe_kf"]))
    res = dict(
        workload=name, desc=w["desc"], value=iters_total / dt, ms_per_step=1e3 * dt / steps, steps=steps,
        iters_per_step=iters_total / steps, n_hyp_job=n_hyp_job, hyp_this_rank=len(hyp), good_hypotheses=n_good,
        ms_per_object_refine=1e3 * dt / steps / w["n_obj"] / (1 if strong else world),
        ms_ba=ba_stat["ms"] / steps, ba_lm_iterations=ba_stat["iters"] / steps, ba_lm_trials=ba_stat["trials"] / steps,
        ba_iters_per_s=ba_stat["iters"] / max(1e-3 * ba_stat["ms"], 1e-12),
        ms_gather=ba_stat["ms_gather"] / steps,
        fallbacks=dict(range_reruns_on_f32=int(prof["fallbacks"]) + (dec.range_fallbacks if batch is None else 0),
                       decoder_counter=dec.range_fallbacks,
                       screened_runs_repeated_in_one_pass=int(prof.get("screen_fallbacks", 0)),
                       note="timed steps that were repeated on the exact-f32 pipe because a value left fp16's range "
                            "(QSP_DEC_OPT_RANGE_FALLBACK); any non-zero count invalidates the line's dtype"),
        screening=(dict(margin=margin, max_abs_s1_minus_s3_on_band=float(prof.get("screen_max_diff", 0.0)),
                        band_share=prof["pts_band"] / max(prof["pts_fwd"], 1),
                        depth_staging=dict(on=bool(dec.precision == "fp16x2" and not (args.no_depth_staging if staging is None else not staging)),
                                           note="two depth stages: samples behind a ray's first opaque sample (exact zero transmittance, "
                                                "loss.py:101) are not evaluated; samples_per_launch counts the evaluated ones; "
                                                "`fp16x2_screened_one_depth_stage` is the same line with every valid sample evaluated"),
                        audit=dict(one_in=args.screen_audit, out_of_band_samples_re_evaluated_per_launch=prof["pts_audit"] / max(prof["n_fwd"], 1),
                                   found_inside_the_cut_off=int(prof["audit_failures"]),
                                   note="the second pass also re-evaluates one in N of the samples the screening pass put OUTSIDE the "
                                        "band (they are part of band_share); one found inside the cut-off repeats the run in one pass; "
                                        "`fp16x2_screened_no_audit` is the same line with the audit off (its cost)"),
                        samples_per_launch=prof["pts_fwd"] / max(prof["n_fwd"], 1),
                        band_samples_per_launch=prof["pts_band"] / max(prof["n_fwd"], 1),
                        note="ray-sample forward in two passes: all samples on the one-product fp16 tile, the band "
                             "|s1| < cut_off + margin again on the split-fp16 tile; K, n_valid, H, b bit-identical to the "
                             "unscreened path (tests/test_gpu_screening.py)")
                   if (dec.precision == "fp16x2" and margin > 0) else None),
        ba_desc="local joint BA 5+10 LM iterations: %d KF / %d map points / %d objects, %d mono + %d stereo + %d "
                "camera-object edges; dense factorisation: %s"
                % (w["n_kf"], w["n_map"], w["n_obj"], len(scene["mono_pt"]), len(scene["st_pt"]), len(scene["oe_kf"]),
                   "one launch: resident chain workgroup + tile workgroups by ticket" if ba.cholesky_chain else "one launch per block step"),
        jtj=dict(achieved=achieved, avg_ms=avg_ms, launches=prof["n_jtj"],
                 points_per_launch=prof["pts_jtj"] / max(prof["n_jtj"], 1),
                 tile_padding_overhead=64.0 * prof["tiles_jtj"] / max(prof["pts_jtj"], 1)),
        kernels={"k_mlp_fwd_TFLOPs": fwd_tf, "k_mlp_fwd_frac_of_its_peak": fwd_tf / peak_for(precision or args.precision),
                 "k_mlp_fwd_vs_f32_mfma_peak": fwd_tf / PEAK_F32_MFMA_TFLOPS,
                 "ms_mlp_jtj": prof["ms_mlp_jtj"] / steps, "ms_mlp_fwd": prof["ms_mlp_fwd"] / steps,
                 "ms_other": prof["ms_other"] / steps, "ms_gpu_total": prof["ms_total"] / steps,
                 "ms_ba": ba_stat["ms"] / steps, "ba_lm_iterations": ba_stat["iters"] / steps,
                 "ba_lm_trials": ba_stat["trials"] / steps, "ba_linearize_us": lin_us,
                 "ba_linearize_bytes": ba_stat["bytes_lin"],
                 "ba_linearize_GBps": ba_stat["bytes_lin"] / max(lin_us, 1e-9) / 1e3,
                 "ba_linearize_frac_of_8TBps": ba_stat["bytes_lin"] / max(lin_us, 1e-9) / 1e3 / PEAK_HBM_GBPS,
                 "ba_linearize_moved_bytes": moved_lin, "ba_linearize_moved_GBps": moved_lin / max(lin_us, 1e-9) / 1e3,
                 "ba_linearize_events_in": "timed steps" if lin_in_timed else "warm-up steps (no event records inside the timed BA)",
                 "ba_linearize_note": "launch-bound at the BASELINE sizes: 2 launches for a few MB; `ba_linearize_large` "
                                      "(N = 1 only) is the same code on a graph large enough to stream"})
    if detailed:
        res["_objs"], res["_scene"], res["_hyp"], res["_T0"] = objs, scene, hyp, T0
        res["_dec"], res["_opt"] = dec, opt
    if batch is not None:
        batch.close()
    ba.close()
    return res


def latency_block(dec):
    """The reference's real call pattern (BASELINE config 3 stand-in: one object per call, src/LocalMapping_util.cc:705):
    Optimizer.reconstruct_object on 2 k surface points / 456 rays / 5 iterations, and the four yaw flips as one call."""
    from qsp_slam_amd import synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    opt = Optimizer(dec, joint_cfg(5))
    o = synth.make_object_views(3003, 1, 2000, n_fg=256, n_bg=200)[0]
    obj = dict(t_cam_obj=o["t_cam_obj"], pts=o["pts"], rays=o["rays"], depth=o["depth"])
    n = 20

    def timed():
        for _ in range(3):
            opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
        t0 = time.perf_counter()
        for _ in range(n):
            opt.reconstruct_object(o["t_cam_obj"], o["pts"], o["rays"], o["depth"])
        one = 1e3 * (time.perf_counter() - t0) / n
        opt.reconstruct_objects_batched([obj], flip_sample_num=4)
        t0 = time.perf_counter()
        for _ in range(n):
            opt.reconstruct_objects_batched([obj], flip_sample_num=4)
        return one, 1e3 * (time.perf_counter() - t0) / n

    one, four = timed()
    out = {"workload": "C3 stand-in: one object per call, 2000 surface points, 256+200 rays, 5 GN iterations",
           "ms_reconstruct_object": one, "ms_four_flips_one_call": four, "calls_timed": n}
    if dec.precision == "fp16x2":       # the explicit latency option: 32-point tiles in the Jacobian kernel (QSP_DEC_OPT_TILE_POINTS)
        dec.set_tile_points(32)
        one32, four32 = timed()
        dec.set_tile_points(64)
        out["tile_points_32"] = {"ms_reconstruct_object": one32, "ms_four_flips_one_call": four32}
    # path B the way the drop-in Optimizer calls it (src/Optimizer_util.cc:309-771 builds its graph per call): a qsp_ba_problem
    # made from host arrays, one two-stage local joint BA, the state read back, the problem destroyed -- per call, C4's graph
    from qsp_slam_amd.ba import BaProblem
    w = WORKLOADS["c4"]
    scene = synth.make_ba_scene(2000, w["n_kf"], w["n_map"], w["n_obj"], stereo_frac=0.2)

    def ba_call():
        b = BaProblem(scene, device=dec.device)
        b.local_joint_ba()
        b.state()
        b.close()

    for _ in range(2):
        ba_call()
    t0 = time.perf_counter()
    for _ in range(10):
        ba_call()
    out["ms_local_joint_ba_problem_per_call"] = 1e3 * (time.perf_counter() - t0) / 10
    out["ba_workload"] = "C4's graph (%s), created from host arrays and destroyed in every call" % w["desc"].split(":")[0]
    # the call that follows every successful refinement (src/LocalMapping_util.cc:832): MeshExtractor.extract_mesh_from_code --
    # decode over the voxel grid + Lewiner's marching cubes on the device (32^3: the reference's KITTI setting, 64^3: the others)
    import builtins
    from qsp_slam_amd.reconstruct.optimizer import MeshExtractor
    code = np.zeros(64, np.float32)
    _print = builtins.print
    builtins.print = lambda *a, **k: None       # (the mirror prints the reference's "Extract mesh takes ..." line per call)
    try:
        for dim in (32, 64):
            me = MeshExtractor(dec, 64, dim)
            for _ in range(2):
                me.extract_mesh_from_code(code)
            t0 = time.perf_counter()
            for _ in range(10):
                m = me.extract_mesh_from_code(code)
            out["ms_extract_mesh_%d" % dim] = 1e3 * (time.perf_counter() - t0) / 10
            out["mesh_%d" % dim] = "%d vertices, %d faces" % (len(m.vertices), len(m.faces))
    finally:
        builtins.print = _print
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--precision", default="fp16x2", choices=["fp16x2", "bf16x3", "f32"])
    ap.add_argument("--flips", type=int, default=4)
    ap.add_argument("--screening-margin", type=float, default=0.01,
                    help="fp16x2 only: band margin of the two-pass ray-sample forward (QSP_DEC_OPT_RENDER_SCREENING); 0 = one pass")
    ap.add_argument("--no-depth-staging", action="store_true",
                    help="screened forward: evaluate every valid ray sample in one depth stage (QSP_DEC_OPT_DEPTH_STAGING = 0; same bits)")
    ap.add_argument("--screen-audit", type=int, default=100,
                    help="screened forward: one in N out-of-band samples re-evaluated as well (QSP_DEC_OPT_SCREEN_AUDIT); 0 = off")
    ap.add_argument("--allow-hook-fallback", action="store_true",
                    help="N > 1, strong scaling: if the library's own RCCL communicator cannot be created, measure the "
                         "torch.distributed all-reduce hook instead of exiting with an error")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sublines", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the large-graph linearisation, host-buffer and latency measurements (for profiler runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing has touched the GPU yet: start the ranks as a CHILD (never exec) and hand its exit code on
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    ctx = Ctx(args)
    if ctx.world != max(args.gpus, 1) and ctx.rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, ctx.world), file=sys.stderr)
    world, rank = ctx.world, ctx.rank
    main_res = run_workload(ctx, args.workload, args.steps, args.warmup, detailed=True)
    other = {}
    if not args.no_sublines:      # the other matrix pipes on the same workload, beside `value`
        for pr in ("f32", "bf16x3"):
            if pr != args.precision:
                other[pr] = run_workload(ctx, args.workload, 3, 1, detailed=False, precision=pr)
        if args.precision == "fp16x2" and args.screening_margin > 0:     # the same pipe with the one-pass forward
            other["fp16x2_unscreened"] = run_workload(ctx, args.workload, 3, 1, detailed=False, precision="fp16x2", screening=0.0)
            if args.screen_audit > 0:        # ... and screened without the out-of-band audit: what the audit costs
                other["fp16x2_screened_no_audit"] = run_workload(ctx, args.workload, 3, 1, detailed=False, precision="fp16x2", audit=0)
            if not args.no_depth_staging:    # ... and screened in ONE depth stage: what the samples behind opaque ones cost
                other["fp16x2_screened_one_depth_stage"] = run_workload(ctx, args.workload, 3, 1, detailed=False, precision="fp16x2", staging=False)
    subs = {}
    if not args.no_sublines:
        for name in ("c2", "c5"):
            if name != args.workload:
                r = run_workload(ctx, name, 2, 1, detailed=False)
                subs[name] = {k: v for k, v in r.items() if not k.startswith("_") and k != "kernels"}
                subs[name]["roofline_frac_k_mlp_jtj"] = r["jtj"]["achieved"] / peak_for(args.precision)
                subs[name]["ba_linearize_us"] = r["kernels"]["ba_linearize_us"]
                subs[name]["ba_linearize_algorithmic_GBps"] = r["kernels"]["ba_linearize_GBps"]
                subs[name]["ba_linearize_moved_GBps"] = r["kernels"]["ba_linearize_moved_GBps"]
                subs[name]["ba_latency_model"] = ba_latency_model(WORKLOADS[name]["n_kf"] - 1, r["ms_ba"], r["ba_lm_trials"])

    if rank == 0:
        w = WORKLOADS[args.workload]
        m = main_res
        pr = args.precision
        traffic, traffic_src = pmc_traffic(args.workload, "k_mlp_jtj" if pr == "f32" else "k_mlp_jtj_" + pr)
        split = pr != "f32"
        # roofline of the dominant kernel.  f32 pipe: algorithmic FLOP / time against the f32 MFMA peak.  Split pipes: every
        # algorithmic multiply-add is 3 (fp16x2) or 6 (bf16x3) 16-bit multiply-adds on the matrix pipe, so the peak for
        # ALGORITHMIC flops is the dense 16-bit peak / that; the pipe's own rate (products x achieved) against the 2.5 PF peak
        # is the same fraction.
        peak = peak_for(pr)
        dtype = {"f32": "f32 (decoder, exact f32 matrix pipe); f64 (bundle adjustment)",
                 "bf16x3": "f32 as 3 x bf16 per operand, 6 bf16 products per multiply-add, f32 accumulate (decoder); f64 (bundle "
                           "adjustment)",
                 "fp16x2": "f32 as 2 x fp16 per operand (hi + 2^-11 lo'), 3 fp16 products per multiply-add, f32 accumulate "
                           "(decoder); f64 (bundle adjustment)"}[pr]
        mfma = {"f32": "v_mfma_f32_32x32x2_f32", "bf16x3": "v_mfma_f32_32x32x16_bf16, 6 products per f32 multiply-add",
                "fp16x2": "v_mfma_f32_32x32x16_f16, 3 products per f32 multiply-add"}[pr]
        out = {
            "metric": "joint-opt iters/sec (BA+SDF)", "value": m["value"], "unit": "iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic (seeded scene, decoder fitted to an analytic shape family)",
            "config": {"workload": w["desc"], "objects": w["n_obj"], "hypotheses": m["n_hyp_job"],
                       "hypotheses_on_rank0": m["hyp_this_rank"],
                       "surface_points": w["n_pts"], "rays": w["n_fg"] + w["n_bg"], "depth_samples": 50,
                       "gn_iterations": w["n_iter"], "ba": m["ba_desc"],
                       "parallelism": ("1 GPU" if world == 1 else
                                       ("objects o %% %d; BA landmarks pt_id %% %d + all-reduce of the reduced camera/object "
                                        "system per LM trial [%s]" % (world, world, ctx.comm_kind)) if args.scaling == "strong"
                                       else "%d independent scenes (replicas)" % world)},
            "ms_per_object_refine": m["ms_per_object_refine"],
            "ms_ba": m["ms_ba"], "ba_iters_per_s": m["ba_iters_per_s"], "ms_result_gather": m["ms_gather"],
            "good_hypotheses": m["good_hypotheses"],
            "roofline": {"bound": "mfma",
                         "kernel": "k_mlp_jtj%s (decoder fwd+bwd+JtJ; %s)" % ("_h2" if pr == "fp16x2" else "", mfma),
                         "achieved": m["jtj"]["achieved"], "peak": peak, "unit": "TFLOP/s",
                         "frac": m["jtj"]["achieved"] / peak,
                         "peak_note": ("dense 16-bit MFMA peak 2500 TFLOP/s / %d products per algorithmic multiply-add"
                                       % PRODUCTS[pr] if split else "f32 MFMA peak"),
                         "matrix_pipe_TFLOPs": PRODUCTS[pr] * m["jtj"]["achieved"] if split else None,
                         "frac_of_sustained_pipe_rate": (PRODUCTS[pr] * m["jtj"]["achieved"] / SUSTAINED_FP16_MFMA_TFLOPS
                                                         if pr == "fp16x2" else None),
                         "sustained_note": ("bare fp16 MFMA loop with random operands on all SIMDs: %.0f TFLOP/s at 1.6 GHz "
                                            "(profiles/r02_mfma_rate_f16.txt); `peak` is the data-sheet figure at 2.4 GHz"
                                            % SUSTAINED_FP16_MFMA_TFLOPS) if pr == "fp16x2" else None,
                         "effective_vs_f32_mfma_peak": m["jtj"]["achieved"] / PEAK_F32_MFMA_TFLOPS,
                         "traffic": traffic,
                         "traffic_unit": "bytes/launch, rocprofv3 PMC passes of this command (profiles/%s)" % traffic_src,
                         "avg_launch_ms": m["jtj"]["avg_ms"], "launches": m["jtj"]["launches"],
                         "points_per_launch": m["jtj"]["points_per_launch"],
                         "tile_padding_overhead": m["jtj"]["tile_padding_overhead"]},
            "kernels": m["kernels"],
        }
        out["fallbacks"] = m["fallbacks"]
        out["screening"] = m["screening"]
        for opr, o in other.items():
            base = opr.split("_")[0]
            out[opr + ("_mfma" if opr == base else "")] = {
                "note": ("the same workload with --precision %s, 3 timed steps" % opr) if opr == base else
                        ("the same workload and pipe, screened, with --screen-audit 0, 3 timed steps" if opr.endswith("no_audit") else
                         "the same workload and pipe, screened, with --no-depth-staging (every valid sample evaluated), 3 timed steps" if opr.endswith("depth_stage") else
                         "the same workload and pipe with --screening-margin 0 (every ray sample on the split-fp16 tile), 3 timed steps"),
                "value": o["value"], "ms_per_step": o["ms_per_step"],
                "k_mlp_jtj_TFLOPs": o["jtj"]["achieved"],
                "k_mlp_jtj_frac_of_its_peak": o["jtj"]["achieved"] / peak_for(base),
                "k_mlp_fwd_TFLOPs": o["kernels"]["k_mlp_fwd_TFLOPs"], "ms_mlp_fwd": o["kernels"]["ms_mlp_fwd"],
                "fallbacks": o["fallbacks"]["range_reruns_on_f32"]}
        if subs:
            out["sublines"] = subs
        # The BA's own roofline line.  Its bandwidth kernel is the linearisation (J^T J build); at the BASELINE sizes that is a few
        # MB behind two launches, so the headline is the largest BASELINE graph, C5 (200 KF / 160 k edges), with the bytes the
        # kernels actually move; the same code on a graph large enough to stream is quoted beside it.
        c5 = subs.get("c5") if args.workload != "c5" else {"ba_linearize_us": m["kernels"]["ba_linearize_us"],
                                                            "ba_linearize_moved_GBps": m["kernels"]["ba_linearize_moved_GBps"],
                                                            "ba_linearize_algorithmic_GBps": m["kernels"]["ba_linearize_GBps"],
                                                            "ms_ba": m["ms_ba"]}
        lat = {name: sub["ba_latency_model"] for name, sub in subs.items() if "ba_latency_model" in sub}
        lat[args.workload] = ba_latency_model(w["n_kf"] - 1, m["ms_ba"], m["ba_lm_trials"])
        out["ba_latency_roofline"] = {
            "bound": "latency (dependent chain of one LM trial)", "unit": "us per LM trial", "per_workload": lat,
            "frac": lat[args.workload]["frac"],
            "note": "model = 9 dependent launches x 1.5 us + block rows x (14.1 us diagonal factorisation + 4.4 us of products) + the "
                    "factor through one compute unit at 148 GB/s; frac = model / achieved; see bench.py:ba_latency_model"}
        if c5:
            out["ba_roofline"] = {"bound": "hbm", "kernel": "k_lin_edges + k_lin_vertices (BA linearisation, J^T J build)",
                                  "workload": "C5: 200 KF / 20000 map points / 256 objects", "unit": "GB/s",
                                  "achieved": c5["ba_linearize_moved_GBps"], "achieved_basis": "bytes the two kernels read and write",
                                  "peak": PEAK_HBM_GBPS, "frac": c5["ba_linearize_moved_GBps"] / PEAK_HBM_GBPS,
                                  "algorithmic_GBps": c5["ba_linearize_algorithmic_GBps"], "us_per_linearisation": c5["ba_linearize_us"],
                                  "ms_local_joint_ba": c5["ms_ba"],
                                  "note": "latency-bound at every BASELINE size (two launches for <= 30 MB); see "
                                          "kernels.ba_linearize_large for the streaming regime of the same kernels"}
        if world == 1 and not args.no_extras:
            from qsp_slam_amd import synth
            from qsp_slam_amd.ba import BaProblem
            from qsp_slam_amd.reconstruct.optimizer import RefineBatch, _joint_cfg
            dec, opt, objs, scene, hyp, T0 = m["_dec"], m["_opt"], m["_objs"], m["_scene"], m["_hyp"], m["_T0"]
            # the BA linearisation kernels are launch-bound at the BASELINE sizes (7.5 MB per build); their bandwidth is
            # measured on a graph large enough to stream: 64 key-frames x 250 000 landmarks x 8 observations
            big = synth.make_ba_scene_large(7, 64, 250000)
            bb = BaProblem(big, device=ctx.dev)
            bb.profile(True)
            for _ in range(2):
                bb.set_state(big["kf_pose"], big["pt_xyz"], big["obj_pose"])
                bb.optimize(2, 0, 0, 0)
                st = bb.profile(True)
            us = 1e3 * st.ms_linearize / max(st.n_linearize, 1)
            n_edge = len(big["mono_pt"])
            moved = 2 * 56 * n_edge + 24 * 250000 + (72 + 24) * 250000 + 392 * 64   # two passes over the 56 B edges + points
            out["kernels"]["ba_linearize_large"] = {
                "graph": "64 KF / 250000 landmarks / %d mono edges" % n_edge, "us": us,
                "algorithmic_bytes": int(st.bytes_linearize), "algorithmic_GBps": st.bytes_linearize / us / 1e3,
                "algorithmic_frac_of_8TBps": st.bytes_linearize / us / 1e3 / PEAK_HBM_GBPS,
                "moved_bytes": int(moved), "moved_GBps": moved / us / 1e3,
                "moved_frac_of_8TBps": moved / us / 1e3 / PEAK_HBM_GBPS,
                "note": "algorithmic = SURVEY 8d's count (includes a 144 B Hpl write per edge the kernels no longer "
                        "perform); moved = bytes the kernels actually read and write"}
            bb.close()
            # the boundary as the reference calls it: host buffers in, host results out (upload + allocation inside the
            # step); reported beside `value`, never as `value`
            th = time.perf_counter()
            b2 = RefineBatch(dec, _joint_cfg(opt), [o["pts"] for o in objs], [o["rays"] for o in objs],
                             [o["depth"] for o in objs], hyp)
            b2.set_state(T0, None)
            b2.run(0)
            b2.get()
            ba2 = BaProblem(scene, device=ctx.dev)
            ba2.local_joint_ba()
            ba2.state()
            th = time.perf_counter() - th
            out["host_buffers"] = {"ms_per_step": 1e3 * th, "value": m["iters_per_step"] / th, "unit": "iters/s",
                                   "note": "one step with observations, graph and results crossing PCIe and device "
                                           "buffers allocated inside the step"}
            b2.close()
            ba2.close()
            out["latency"] = latency_block(dec)
        if world == 1 and not args.no_cpu_baseline:      # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(w, m["_objs"], m["_scene"], len(m["_hyp"]))
        print(json.dumps(out))
        sys.stdout.flush()
    if ctx.comm is not None:
        ctx.torch.cuda.synchronize(ctx.dev)
        ctx.comm.close()
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
