"""N > 1 on the GPU: every rank is a FRESH child process (subprocess, never a re-exec of this process), all on device 0 of
the one-GPU box.  Covers SURVEY.md section 8e: the landmark-sharded joint BA against the unsharded solve, the object-sharded
refinement against the single-process batch, and bench.py's strong-scaling control flow (QSP_BENCH_REHEARSAL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(mode, world, tmp_path, extra_env=None):
    port = _free_port()
    outs = [str(tmp_path / ("%s_%d.npz" % (mode, r))) for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker.py"), mode, str(r), str(world),
                               str(port), outs[r]], env=env, cwd=ROOT) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [np.load(o) for o in outs]


def close(a, b, rtol, atol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= atol + rtol * np.abs(b).max()


def test_two_process_landmark_sharded_ba_matches_unsharded(tmp_path):
    """two ranks, two processes, one scene: same LM path, same estimates (1e-7), every rank leaves with ALL landmarks and
    all per-edge chi2 values"""
    import mp_worker
    from qsp_slam_amd import synth
    from qsp_slam_amd.ba import BaProblem
    sc = synth.make_ba_scene(**mp_worker.BA_SCENE)
    ref = BaProblem(sc)
    r1, r2 = ref.local_joint_ba()
    rkf, rpt, rob = ref.state()
    re = ref.edges()
    ref.close()
    for got in _run_ranks("ba", 2, tmp_path):
        assert list(got["trials_1"]) == list(r1["trials"]) and list(got["trials_2"]) == list(r2["trials"])
        assert close(got["chi2_1"], r1["chi2"], 1e-8) and close(got["chi2_2"], r2["chi2"], 1e-8)
        assert close(got["lam_1"], r1["lam"], 1e-7) and close(got["lam_2"], r2["lam"], 1e-7)
        assert close(got["kf"], rkf, 1e-7, 1e-9) and close(got["pt"], rpt, 1e-7, 1e-9) and close(got["ob"], rob, 1e-7, 1e-9)
        assert close(got["mono_chi2"], re["mono_chi2"], 1e-6, 1e-9) and close(got["oe_chi2"], re["oe_chi2"], 1e-6, 1e-9)


def _stub_rccl():
    """tests/stub_rccl/librccl_stub.so, built here if the snapshot does not carry it"""
    d = os.path.join(ROOT, "tests", "stub_rccl")
    so = os.path.join(d, "librccl_stub.so")
    if not os.path.isfile(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(d, "stub_rccl.cpp"), "-lrt",
                               "-lpthread"])
    return so


def test_two_process_rccl_on_stream_path_with_the_stub_library(tmp_path):
    """the production collective path -- qsp_ba_set_shard_rccl, ncclAllReduce(SUM) per LM trial and
    ncclAllReduce(MAX) for lambda's start, issued on the BA's own stream (csrc/ba_solver.hip) -- had only ever run with one rank:
    RCCL refuses two ranks on one device.  Here two fresh processes on device 0 run it against the shared-memory stand-in
    librccl (QSP_RCCL_LIB): the result equals the gloo-hook path's bit for bit (both sum two ranks; a + b is commutative),
    follows the unsharded solve to 1e-7, the MAX branch ran, and the library's all-gather entry point works with two ranks."""
    import mp_worker
    from qsp_slam_amd import synth
    from qsp_slam_amd.ba import BaProblem
    sc = synth.make_ba_scene(**mp_worker.BA_SCENE)
    ref = BaProblem(sc)
    r1, r2 = ref.local_joint_ba()
    rkf, rpt, rob = ref.state()
    ref.close()
    hook = _run_ranks("ba", 2, tmp_path)
    rccl = _run_ranks("ba_rccl", 2, tmp_path, extra_env={"QSP_RCCL_LIB": _stub_rccl()})
    for got, want in zip(rccl, hook):
        for k in ("kf", "pt", "ob", "chi2_1", "chi2_2", "lam_1", "lam_2", "trials_1", "trials_2", "mono_chi2", "oe_chi2"):
            assert np.array_equal(got[k], want[k]), k
        assert list(got["trials_1"]) == list(r1["trials"]) and list(got["trials_2"]) == list(r2["trials"])
        assert close(got["kf"], rkf, 1e-7, 1e-9) and close(got["pt"], rpt, 1e-7, 1e-9) and close(got["ob"], rob, 1e-7, 1e-9)
        n_sum, n_max, n_gather = [int(v) for v in got["counts"]]
        assert n_max >= 2                      # lambda's start, once per optimize() call: the ncclMax branch
        assert n_sum >= 2 * int(np.sum(r1["trials"]) + np.sum(r2["trials"]))   # reduced system + scalars per LM trial, at least
        assert n_gather >= 1 and np.array_equal(got["gathered"], np.repeat([1.0, 2.0], 5).astype(np.float32))


def test_two_process_object_sharded_refinement_is_bit_identical(tmp_path):
    """objects o % 2 on two processes + one all_gather == the single-process batch, bit for bit (a hypothesis gives the same
    bits in any batch)"""
    import mp_worker
    from qsp_slam_amd import DeepSdfDecoder, parallel, synth
    from qsp_slam_amd.reconstruct.optimizer import Optimizer
    c = mp_worker.REFINE
    dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"), device=0)
    opt = Optimizer(dec, mp_worker.refine_config())
    objs = synth.make_object_views(c["seed"], c["n_obj"], c["n_pts"], n_fg=c["n_fg"], n_bg=c["n_bg"])
    objs = [dict(t_cam_obj=o["t_cam_obj"], pts=o["pts"], rays=o["rays"], depth=o["depth"]) for o in objs]
    want = parallel.pack_results(opt.reconstruct_objects_batched(objs, flip_sample_num=4, select=True))
    assert want[:, 81].sum() >= 3
    for got in _run_ranks("refine", 2, tmp_path):
        assert np.array_equal(got["table"], want)


def test_bench_strong_scaling_rehearsal_two_ranks():
    """bench.py's N = 2 strong-scaling control flow on one GPU (gloo reductions): one JSON line, the shared BA walks the
    same number of LM iterations as at N = 1, every hypothesis is accounted for"""
    def run(n):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        if n > 1:
            env["QSP_BENCH_REHEARSAL"] = "1"
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "c2", "--steps", "1",
               "--warmup", "1", "--no-sublines", "--no-cpu-baseline"]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])
    one, two = run(1), run(2)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and one["n_gpus"] == 1
    assert two["config"]["hypotheses"] == one["config"]["hypotheses"] == 32
    assert two["config"]["hypotheses_on_rank0"] == 16
    assert two["kernels"]["ba_lm_iterations"] == one["kernels"]["ba_lm_iterations"]
    assert two["good_hypotheses"] == one["good_hypotheses"]
    assert two["value"] > 0 and "roofline" in two
    # the blocks round 4 added to the driver line
    lat = one["ba_latency_roofline"]
    assert lat["per_workload"]["c2"]["block_rows"] == 2 and 0 < lat["frac"] <= 1.0
    assert one["roofline"]["bound"] == "mfma" and 0 < one["roofline"]["frac"] < 1
