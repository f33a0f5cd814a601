"""The ORB-SLAM2 `Optimizer` shim (include/qsp_optimizer_shim.h) compiled against stand-in map types (tests/shim_mock/).

CPU: linked with a recording stub of the C-ABI, it must flatten a mock map into exactly the graph the reference would
build (vertex ids, fixed flags, intrinsics, edge lists, float32->float64 pose conversion) and write results back through
the same setters, including outlier erasure.
GPU: linked with the real libqsp_hip.so, its result on the mock map equals qsp_slam_amd.ba.BaProblem on the same graph."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from qsp_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "shim_mock")


_BUILD_CACHE = {}


def _cache_dir():
    """the sanitizer builds take ~10 s each: build every binary once per test session"""
    import atexit
    import shutil
    if "dir" not in _BUILD_CACHE:
        _BUILD_CACHE["dir"] = tempfile.mkdtemp(prefix="qsp_shim_")
        atexit.register(shutil.rmtree, _BUILD_CACHE["dir"], True)
    return _BUILD_CACHE["dir"]


def build_driver(tmp, real):
    key = "real" if real else "stub"
    if key not in _BUILD_CACHE:
        _BUILD_CACHE[key] = _build_driver(os.path.join(_cache_dir()), real)
    return _BUILD_CACHE[key]


def _build_driver(tmp, real):
    out = os.path.join(tmp, "shim_driver_real" if real else "shim_driver_stub")
    inc = ["-I" + MOCK, "-I" + os.path.join(ROOT, "include")]
    # the stub build (CPU tests) runs the shim's flattening / write-back code under AddressSanitizer + UBSan: any out-of-bounds
    # index into the map vectors or the flat arrays fails the test.  (Sanitizers are for the CPU build only.)
    san = [] if real else ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + san + inc + ["-c", os.path.join(MOCK, "shim_driver.cpp"), "-o",
                                                                        os.path.join(tmp, "drv_%d.o" % real)])
    if real:
        lib = os.path.join(ROOT, "qsp_slam_amd")
        subprocess.check_call(["g++", "-o", out, os.path.join(tmp, "drv_%d.o" % real), "-L" + lib, "-lqsp_hip", "-Wl,-rpath," + lib])
    else:
        subprocess.check_call(["gcc", "-std=c11", "-O1"] + san + inc + ["-c", os.path.join(MOCK, "stub_qsp.c"), "-o",
                                                                          os.path.join(tmp, "stub_c.o")])
        subprocess.check_call(["g++"] + san + ["-o", out, os.path.join(tmp, "drv_%d.o" % real), os.path.join(tmp, "stub_c.o")])
    return out


def make_map(seed=5, n_kf=9, n_pt=120, n_obj=2):
    """a synthetic scene re-expressed as what the MAP holds: float32 4x4 poses, key-point pixels + octaves"""
    rng = np.random.default_rng(seed)
    sc = synth.make_ba_scene(seed, n_kf, n_pt, n_obj, stereo_frac=0.3, obs_per_obj=4)
    m = dict(sc=sc)
    m["kfT"] = np.array([synth.pose7_to_T(p) for p in sc["kf_pose"]], np.float32)
    m["objT"] = np.array([synth.pose7_to_T(p) for p in sc["obj_pose"]], np.float32)
    m["oeZ"] = np.array([synth.pose7_to_T(p) for p in sc["oe_meas"]], np.float32)
    m["kf_local"] = np.array([1 if i < 6 else 0 for i in range(n_kf)], np.int32)
    m["pt_mn"] = (sc["pt_id"] - int(sc["kf_id"].max()) - 1).astype(np.int64)
    m["obj_mn"] = np.arange(n_obj, dtype=np.int64) + 3
    m["mono_oct"] = np.round(-np.log(sc["mono_info"]) / (2 * np.log(1.2))).astype(np.int32)
    m["st_oct"] = np.round(-np.log(sc["st_info"]) / (2 * np.log(1.2))).astype(np.int32)
    return m


def write_scene(m, path):
    sc = m["sc"]
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", len(sc["kf_pose"]), len(sc["pt_xyz"]), len(sc["obj_pose"]), len(sc["mono_pt"]),
                            len(sc["st_pt"]), len(sc["oe_kf"])))
        for a, dt in ((m["kfT"], np.float32), (sc["kf_id"], np.int64), (sc["kf_K"], np.float32), (m["kf_local"], np.int32),
                      (sc["pt_xyz"], np.float32), (m["pt_mn"], np.int64), (m["objT"], np.float32), (m["obj_mn"], np.int64),
                      (sc["mono_pt"], np.int32), (sc["mono_kf"], np.int32), (sc["mono_obs"], np.float32),
                      (m["mono_oct"], np.int32), (sc["st_pt"], np.int32), (sc["st_kf"], np.int32),
                      (sc["st_obs"], np.float32), (m["st_oct"], np.int32), (sc["oe_kf"], np.int32),
                      (sc["oe_obj"], np.int32), (m["oeZ"], np.float32)):
            f.write(np.ascontiguousarray(a, dtype=dt).tobytes())


def read_dump(path):
    b = open(path, "rb").read()
    o = [0]

    def take(dt, n, shape=None):
        a = np.frombuffer(b, dtype=dt, count=n, offset=o[0]).copy()
        o[0] += a.nbytes
        return a.reshape(shape) if shape else a
    n_kf, n_pt, n_obj, nm, ns, no = take(np.int32, 6)
    d = dict(kf_pose=take(np.float64, 7 * n_kf, (n_kf, 7)), kf_fixed=take(np.uint8, n_kf), kf_id=take(np.int64, n_kf),
             kf_K=take(np.float64, 5 * n_kf, (n_kf, 5)), pt_xyz=take(np.float64, 3 * n_pt, (n_pt, 3)),
             pt_id=take(np.int64, n_pt), obj_pose=take(np.float64, 7 * n_obj, (n_obj, 7)), obj_id=take(np.int64, n_obj),
             mono_pt=take(np.int32, nm), mono_kf=take(np.int32, nm), mono_obs=take(np.float64, 2 * nm, (nm, 2)),
             mono_info=take(np.float64, nm), st_pt=take(np.int32, ns), st_kf=take(np.int32, ns),
             st_obs=take(np.float64, 3 * ns, (ns, 3)), st_info=take(np.float64, ns), oe_kf=take(np.int32, no),
             oe_obj=take(np.int32, no), oe_meas=take(np.float64, 7 * no, (no, 7)))
    d["oe_info"] = float(take(np.float64, 1)[0])
    return d


def read_out(path, n_kf, n_pt, n_obj):
    b = open(path, "rb").read()
    kf = np.frombuffer(b, np.float32, 16 * n_kf).reshape(n_kf, 4, 4)
    off = 64 * n_kf
    pt = np.frombuffer(b, np.float32, 3 * n_pt, off).reshape(n_pt, 3)
    off += 12 * n_pt
    ob = np.frombuffer(b, np.float32, 16 * n_obj, off).reshape(n_obj, 4, 4)
    off += 64 * n_obj
    nobs, nba = struct.unpack_from("<2i", b, off)
    return kf, pt, ob, nobs, nba


def test_shim_flattens_the_map_like_the_reference_and_writes_back():
    m = make_map()
    sc = m["sc"]
    with tempfile.TemporaryDirectory() as tmp:
        drv = build_driver(tmp, real=False)
        write_scene(m, os.path.join(tmp, "scene.bin"))
        env = dict(os.environ, QSP_STUB_DUMP=os.path.join(tmp, "dump.bin"))
        subprocess.check_call([drv, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.bin")], env=env)
        d = read_dump(os.path.join(tmp, "dump.bin"))
        kf_o, pt_o, ob_o, nobs, nba = read_out(os.path.join(tmp, "out.bin"), len(sc["kf_pose"]), len(sc["pt_xyz"]),
                                               len(sc["obj_pose"]))
    n_kf = len(sc["kf_pose"])
    local = [i for i in range(n_kf) if i == 0 or m["kf_local"][i]]
    # local map points = points matched in a local key-frame, in discovery order (src/Optimizer_util.cc:330-349)
    seen, lpts = set(), []
    edges = [(int(k), int(p)) for p, k in zip(sc["mono_pt"], sc["mono_kf"])] + \
            [(int(k), int(p)) for p, k in zip(sc["st_pt"], sc["st_kf"])]
    per_kf = {}
    for e_i, (k, p) in enumerate(edges):        # the driver appends key-points in edge order: mono first, then stereo
        per_kf.setdefault(k, []).append(p)
    for k in local:
        for p in per_kf.get(k, []):
            if p not in seen:
                seen.add(p)
                lpts.append(p)
    fixed = sorted({k for (k, p) in edges if p in seen and k not in local})     # std::map<KeyFrame*> = index order here
    order = local + [k for k in fixed]
    # fixed cameras are discovered per local point in pointer order; as a set they must match, local part exactly in order
    assert list(d["kf_id"][: len(local)]) == [int(sc["kf_id"][i]) for i in local]
    assert sorted(d["kf_id"][len(local):]) == sorted(int(sc["kf_id"][i]) for i in fixed)
    assert d["kf_fixed"][0] == 1 and list(d["kf_fixed"][1: len(local)]) == [0] * (len(local) - 1)
    assert all(d["kf_fixed"][len(local):] == 1)
    # Converter::toSE3Quat on float32 matrices
    for j, i in enumerate(local):
        assert np.abs(d["kf_pose"][j] - synth.pose7(m["kfT"][i].astype(np.float64))).max() < 1e-6
        assert np.allclose(d["kf_K"][j], sc["kf_K"][i].astype(np.float32))
    max_kf = int(max(sc["kf_id"][i] for i in order))
    assert list(d["pt_id"]) == [int(m["pt_mn"][p]) + max_kf + 1 for p in lpts]
    assert np.allclose(d["pt_xyz"], sc["pt_xyz"][lpts].astype(np.float32))
    max_mp = int(max(m["pt_mn"][p] for p in lpts))
    assert sorted(d["obj_id"]) == sorted(int(x) + max_kf + max_mp + 2 for x in m["obj_mn"])
    # every observation of a local point by a key-frame in the graph is an edge, mono/stereo split by mvuRight < 0
    kf_of = {int(i): n for n, i in enumerate(d["kf_id"])}
    # (a "stereo" observation whose right coordinate is negative IS a monocular one for ORB-SLAM2: mvuRight < 0)
    neg = sc["st_obs"][:, 2].astype(np.float32) < 0
    exp_m = sorted([(int(m["pt_mn"][p]) + max_kf + 1, int(sc["kf_id"][k])) for p, k in zip(sc["mono_pt"], sc["mono_kf"])
                    if p in seen and int(sc["kf_id"][k]) in kf_of] +
                   [(int(m["pt_mn"][p]) + max_kf + 1, int(sc["kf_id"][k]))
                    for p, k, ng in zip(sc["st_pt"], sc["st_kf"], neg) if ng and p in seen and int(sc["kf_id"][k]) in kf_of])
    got_m = sorted((int(d["pt_id"][p]), int(d["kf_id"][k])) for p, k in zip(d["mono_pt"], d["mono_kf"]))
    assert got_m == exp_m
    exp_s = sorted((int(m["pt_mn"][p]) + max_kf + 1, int(sc["kf_id"][k])) for p, k, ng in zip(sc["st_pt"], sc["st_kf"], neg)
                   if not ng and p in seen and int(sc["kf_id"][k]) in kf_of)
    got_s = sorted((int(d["pt_id"][p]), int(d["kf_id"][k])) for p, k in zip(d["st_pt"], d["st_kf"]))
    assert got_s == exp_s
    assert d["oe_info"] == 1e3 and len(d["oe_kf"]) == sum(1 for k in sc["oe_kf"] if int(sc["kf_id"][k]) in kf_of)
    assert np.all((d["mono_info"] > 0) & (d["mono_info"] <= 1.0))
    # write-back of the stub's visible fake update: +0.5 in x for free local key-frames, +0.25 in y for points
    assert nba == 1
    for i in local[1:]:
        assert abs(kf_o[i][0, 3] - (m["kfT"][i][0, 3] + 0.5)) < 1e-5
    assert abs(kf_o[0][0, 3] - m["kfT"][0][0, 3]) < 1e-6                          # mnId 0 stays fixed
    for i in fixed:
        assert np.array_equal(kf_o[i], m["kfT"][i])                               # fixed cameras are not written back
    for p in lpts:
        assert abs(pt_o[p][1] - (np.float32(sc["pt_xyz"][p][1]) + 0.25)) < 1e-5
    assert nobs == len(edges) - 1                                                  # the stub flagged one mono edge as outlier


@pytest.mark.gpu
def test_shim_end_to_end_equals_python_binding_on_the_same_graph():
    from qsp_slam_amd.ba import BaProblem
    m = make_map(seed=8, n_kf=8, n_pt=150, n_obj=2)
    sc = m["sc"]
    with tempfile.TemporaryDirectory() as tmp:
        write_scene(m, os.path.join(tmp, "scene.bin"))
        stub = build_driver(tmp, real=False)
        subprocess.check_call([stub, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "o0.bin")],
                              env=dict(os.environ, QSP_STUB_DUMP=os.path.join(tmp, "dump.bin")))
        d = read_dump(os.path.join(tmp, "dump.bin"))
        real = build_driver(tmp, real=True)
        subprocess.check_call([real, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "o1.bin")])
        kf_o, pt_o, ob_o, nobs, nba = read_out(os.path.join(tmp, "o1.bin"), len(sc["kf_pose"]), len(sc["pt_xyz"]),
                                               len(sc["obj_pose"]))
    prob = BaProblem(d)
    prob.local_joint_ba()
    kf, pt, ob = prob.state()
    id2idx = {int(i): n for n, i in enumerate(sc["kf_id"])}
    n_checked = 0
    for j, vid in enumerate(d["kf_id"]):
        if d["kf_fixed"][j] and int(vid) != 0:
            continue
        T = synth.pose7_to_T(kf[j]).astype(np.float32)
        assert np.abs(kf_o[id2idx[int(vid)]] - T).max() < 2e-6
        n_checked += 1
    assert n_checked >= 5 and nba == 1


def _with_unobserved_extras(m):
    """the same map + 3 map points nobody observes + 1 object nobody observes (dropped by the global BA, never written)"""
    sc = dict(m["sc"])
    m2 = dict(m, sc=sc)
    sc["pt_xyz"] = np.concatenate([sc["pt_xyz"], np.array([[9.0, 9.0, 9.0], [8.0, 8.0, 8.0], [7.0, 7.0, 7.0]])])
    m2["pt_mn"] = np.concatenate([m["pt_mn"], m["pt_mn"].max() + 1 + np.arange(3)]).astype(np.int64)
    sc["obj_pose"] = np.concatenate([sc["obj_pose"], sc["obj_pose"][:1]])
    m2["objT"] = np.concatenate([m["objT"], m["objT"][:1]])
    m2["obj_mn"] = np.concatenate([m["obj_mn"], [m["obj_mn"].max() + 5]]).astype(np.int64)
    return m2


def _run_stub(m, mode, loop_kf=0):
    sc = m["sc"]
    with tempfile.TemporaryDirectory() as tmp:
        drv = build_driver(tmp, real=False)
        write_scene(m, os.path.join(tmp, "scene.bin"))
        env = dict(os.environ, QSP_STUB_DUMP=os.path.join(tmp, "dump.bin"))
        subprocess.check_call([drv, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.bin"), mode, str(loop_kf)], env=env)
        d = read_dump(os.path.join(tmp, "dump.bin"))
        args = np.fromfile(os.path.join(tmp, "dump.bin.args"), np.float64)
        out = read_out(os.path.join(tmp, "out.bin"), len(sc["kf_pose"]), len(sc["pt_xyz"]), len(sc["obj_pose"]))
    return d, args, out


@pytest.mark.parametrize("loop_kf", [0, 7])
def test_global_joint_ba_flattening_and_write_back(loop_kf):
    """Optimizer::GlobalJointBundleAdjustemnt -> JointBundleAdjustment (src/Optimizer_util.cc:36-307): every key-frame
    (only mnId 0 fixed), every point WITH an edge, every static object WITH an observation; optimize(nIterations) with the
    robust deltas; write-back to the map (nLoopKF == 0) or to the *GBA members (loop closing)."""
    m = _with_unobserved_extras(make_map(seed=9, n_kf=7, n_pt=60, n_obj=2))
    sc = m["sc"]
    d, args, (kf_o, pt_o, ob_o, nobs, nba) = _run_stub(m, "global_joint", loop_kf)
    n_kf, n_pt_all, n_obj_all = len(sc["kf_pose"]), len(sc["pt_xyz"]), len(sc["obj_pose"])
    assert len(d["kf_pose"]) == n_kf and list(d["kf_fixed"]) == [1 if i == 0 else 0 for i in sc["kf_id"]]
    observed = sorted(set(sc["mono_pt"].tolist()) | set(sc["st_pt"].tolist()))
    unobserved = sorted(set(range(n_pt_all)) - set(observed))
    assert len(d["pt_xyz"]) == len(observed) and set(range(n_pt_all - 3, n_pt_all)) <= set(unobserved)
    assert len(d["obj_pose"]) == n_obj_all - 1 and len(d["oe_kf"]) == len(sc["oe_kf"])
    assert len(d["mono_pt"]) + len(d["st_pt"]) == len(sc["mono_pt"]) + len(sc["st_pt"])
    max_kf = int(sc["kf_id"].max())
    max_mp = int(m["pt_mn"].max())                      # every point offered counts, also the dropped ones (:97-98)
    assert np.array_equal(d["pt_id"], m["pt_mn"][observed] + max_kf + 1)
    assert np.array_equal(d["obj_id"], m["obj_mn"][: n_obj_all - 1] + max_kf + max_mp + 2)
    assert args[0] == 10 and np.allclose(args[1:], [np.float32(np.sqrt(5.99)), np.float32(np.sqrt(7.815)),
                                                    np.float32(np.sqrt(np.float32(0.1) * np.float32(1e3)))])
    # write-back: the stub shifts free key-frames by +0.5 in x, points by +0.25 in y, objects by +0.125 in z (of T_ow)
    for i in range(n_kf):
        shift = 0.0 if sc["kf_id"][i] == 0 else 0.5
        assert abs(kf_o[i][0, 3] - (m["kfT"][i][0, 3] + shift)) < 1e-5
    moved = np.abs(pt_o[:, 1] - sc["pt_xyz"][:, 1].astype(np.float32)) > 0.2
    assert moved[observed].all() and not moved[unobserved].any()
    Tow_in = m["objT"]
    assert np.abs(ob_o[-1] - Tow_in[-1]).max() < 1e-6           # the unobserved object is untouched
    assert all(abs(ob_o[i][2, 3] - (Tow_in[i][2, 3] + 0.125)) < 1e-4 for i in range(n_obj_all - 1))


def test_global_points_only_ba():
    """Optimizer::GlobalBundleAdjustemnt -> BundleAdjustment (src/Optimizer.cc:46-242): no objects in the graph, no robust
    kernel when bRobust is false, nIterations passed through."""
    m = _with_unobserved_extras(make_map(seed=10, n_kf=6, n_pt=40, n_obj=2))
    sc = m["sc"]
    d, args, (kf_o, pt_o, ob_o, nobs, nba) = _run_stub(m, "global_points")
    assert len(d["obj_pose"]) == 0 and len(d["oe_kf"]) == 0
    n_observed = len(set(sc["mono_pt"].tolist()) | set(sc["st_pt"].tolist()))
    assert len(d["kf_pose"]) == len(sc["kf_pose"]) and len(d["pt_xyz"]) == n_observed <= len(sc["pt_xyz"]) - 3
    assert args[0] == 20 and not args[1:].any()
    assert np.abs(ob_o - m["objT"]).max() < 1e-6                # objects are not part of this call


def _frame_of_kf0(m):
    """what the driver's "pose" mode builds: key-frame 0's observations in key-point order (mono edges first, then stereo)"""
    sc = m["sc"]
    rows = [(int(p), sc["mono_obs"][e][0], sc["mono_obs"][e][1], -1.0, int(m["mono_oct"][e]))
            for e, (p, k) in enumerate(zip(sc["mono_pt"], sc["mono_kf"])) if k == 0]
    rows += [(int(p), sc["st_obs"][e][0], sc["st_obs"][e][1], sc["st_obs"][e][2], int(m["st_oct"][e]))
             for e, (p, k) in enumerate(zip(sc["st_pt"], sc["st_kf"])) if k == 0]
    return rows


def test_pose_optimization_flattening_and_write_back():
    """Optimizer::PoseOptimization (src/Optimizer.cc:244-456) through the shim with the stub library: matched slots only,
    mono / stereo split by mvuRight < 0, float32 pixels and world points, K and the pose of the frame; write-back of the
    outlier flags to the matched slots, of the pose, and the inlier count as return value."""
    m = make_map(seed=12, n_kf=5, n_pt=200, n_obj=0)
    sc = m["sc"]
    with tempfile.TemporaryDirectory() as tmp:
        drv = build_driver(tmp, real=False)
        write_scene(m, os.path.join(tmp, "scene.bin"))
        env = dict(os.environ, QSP_STUB_DUMP=os.path.join(tmp, "dump.bin"))
        subprocess.check_call([drv, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.bin"), "pose"], env=env)
        raw = open(os.path.join(tmp, "dump.bin.pose"), "rb").read()
        out = open(os.path.join(tmp, "out.bin"), "rb").read()
    rows = _frame_of_kf0(m)
    matched = [i for i in range(len(rows)) if i % 5 != 4]
    n = struct.unpack_from("<i", raw, 0)[0]
    assert n == len(matched) >= 20
    off = 4
    K = np.frombuffer(raw, np.float64, 5, off); off += 40
    pose = np.frombuffer(raw, np.float64, 7, off); off += 56
    X = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
    obs = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
    info = np.frombuffer(raw, np.float64, n, off); off += 8 * n
    stereo = np.frombuffer(raw, np.uint8, n, off)
    assert np.allclose(K, sc["kf_K"][0].astype(np.float32))
    assert np.abs(pose - synth.pose7(m["kfT"][0].astype(np.float64))).max() < 1e-6
    for e, i in enumerate(matched):
        p, u, v, ur, octv = rows[i]
        st = not (np.float32(ur) < 0)
        assert stereo[e] == (1 if st else 0)
        assert obs[e][0] == np.float32(u) and obs[e][1] == np.float32(v) and obs[e][2] == (np.float32(ur) if st else -1.0)
        assert np.array_equal(X[e], sc["pt_xyz"][p].astype(np.float32).astype(np.float64))
        assert np.isclose(info[e], np.float32(1.0) / np.float32(1.2) ** np.float32(2.0 * octv), rtol=1e-6)
    T = np.frombuffer(out, np.float32, 16).reshape(4, 4)
    ninl, N = struct.unpack_from("<2i", out, 64)
    flags = np.frombuffer(out, np.uint8, N, 72)
    assert N == len(rows) and ninl == n - (n + 2) // 3
    assert abs(T[1, 3] - (m["kfT"][0][1, 3] + 0.75)) < 1e-5                 # the stub's fake update reached SetPose
    for e, i in enumerate(matched):
        assert flags[i] == (1 if e % 3 == 0 else 0)
    assert all(flags[i] == 1 for i in range(len(rows)) if i % 5 == 4)       # unmatched slots keep their stale flag


@pytest.mark.gpu
def test_pose_optimization_shim_equals_python_binding():
    from qsp_slam_amd.ba import PoseOptimizer
    m = make_map(seed=13, n_kf=5, n_pt=300, n_obj=0)
    sc = m["sc"]
    with tempfile.TemporaryDirectory() as tmp:
        write_scene(m, os.path.join(tmp, "scene.bin"))
        real = build_driver(tmp, real=True)
        subprocess.check_call([real, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.bin"), "pose"])
        out = open(os.path.join(tmp, "out.bin"), "rb").read()
    rows = _frame_of_kf0(m)
    matched = [i for i in range(len(rows)) if i % 5 != 4]
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    X = f32([sc["pt_xyz"][rows[i][0]] for i in matched])
    obs = f32([[rows[i][1], rows[i][2], rows[i][3] if not (np.float32(rows[i][3]) < 0) else -1.0] for i in matched])
    stereo = np.array([0 if np.float32(rows[i][3]) < 0 else 1 for i in matched], np.uint8)
    info = f32([np.float32(1.0) / np.float32(1.2) ** np.float32(2.0 * rows[i][4]) for i in matched])
    po = PoseOptimizer(4096)
    g = po.optimize(f32(sc["kf_K"][0]), synth.pose7(m["kfT"][0].astype(np.float64)), X, obs, info, stereo)
    po.close()
    T = np.frombuffer(out, np.float32, 16).reshape(4, 4)
    ninl, N = struct.unpack_from("<2i", out, 64)
    flags = np.frombuffer(out, np.uint8, N, 72)
    assert ninl == g["n_inliers"]
    assert np.array_equal(flags[matched], g["outlier"])
    assert np.abs(T - synth.pose7_to_T(g["pose"]).astype(np.float32)).max() < 2e-6


# ---------------------------------------------------------------------------------------------------------------------
# the drop-in `class Optimizer` (qsp_slam_amd/orbslam/Optimizer_hip.cc, boundary B2 of SURVEY.md section 8b)
# ---------------------------------------------------------------------------------------------------------------------

def build_dropin(tmp):
    if "dropin" not in _BUILD_CACHE:
        _BUILD_CACHE["dropin"] = _build_dropin(_cache_dir())
    return _BUILD_CACHE["dropin"]


def _build_dropin(tmp):
    """A caller that only knows `Optimizer.h` (tests/shim_mock/dropin/Optimizer.h mirrors the reference's declaration) +
    Optimizer_hip.cc + stand-ins of the reference's two g2o source files + the recording stub of the C-ABI."""
    drop = os.path.join(MOCK, "dropin")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g"]
    inc = ["-I" + drop, "-I" + MOCK, "-I" + os.path.join(ROOT, "include")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-DQSP_SHIM_MOCK_TYPES=1"] + san + inc +
                          ["-c", os.path.join(ROOT, "qsp_slam_amd", "orbslam", "Optimizer_hip.cc"), "-o", os.path.join(tmp, "hip.o")])
    # the caller sees the reference-shaped header only: no qsp include path, no shim macro
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + san + ["-I" + drop, "-I" + MOCK, "-c",
                                                                 os.path.join(drop, "dropin_caller.cpp"), "-o", os.path.join(tmp, "caller.o")])
    subprocess.check_call(["gcc", "-std=c11", "-O1"] + san + ["-I" + os.path.join(ROOT, "include"), "-c",
                                                               os.path.join(MOCK, "stub_qsp.c"), "-o", os.path.join(tmp, "stub_c.o")])
    out = os.path.join(tmp, "dropin_caller")
    subprocess.check_call(["g++"] + san + ["-o", out, os.path.join(tmp, "caller.o"), os.path.join(tmp, "hip.o"), os.path.join(tmp, "stub_c.o")])
    return out


def _run_dropin(fail=None, allow_g2o=False):
    m = make_map(seed=21, n_kf=7, n_pt=80, n_obj=2)
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_dropin(tmp)
        write_scene(m, os.path.join(tmp, "scene.bin"))
        env = dict(os.environ, QSP_G2O_LOG=os.path.join(tmp, "g2o.log"))
        env.pop("QSP_SHIM_ALLOW_G2O_FALLBACK", None)
        env.pop("QSP_SHIM_NO_FALLBACK", None)
        if fail:
            env["QSP_STUB_FAIL"] = fail
        if allow_g2o:
            env["QSP_SHIM_ALLOW_G2O_FALLBACK"] = "1"
        r = subprocess.run([exe, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.txt")], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        out = dict(l.split(" ", 1) for l in open(os.path.join(tmp, "out.txt")).read().splitlines())
        log = open(os.path.join(tmp, "g2o.log")).read().splitlines() if os.path.exists(os.path.join(tmp, "g2o.log")) else []
    return m, out, log, r.stderr


def test_dropin_optimizer_class_routes_hot_path_to_the_library_and_loop_closing_to_g2o():
    """every member of include/Optimizer.h:75-107 exists with the reference's signature; the bundle adjustments and
    PoseOptimization reach the C-ABI (the stub's visible fake updates arrive in the map), OptimizeSim3 /
    OptimizeEssentialGraph reach the reference's g2o code (here: its logging stand-in), nBAdone is a static int that counts
    joint local BAs only, SetGroundPlane / the constructor behave as src/Optimizer.cc:41-44, Optimizer_util.cc:773-776."""
    m, out, log, err = _run_dropin()
    assert out["nBAdone0"] == "0" and out["nBAdone1"] == "1" and out["nBAdone2"] == "1"
    tx = float(m["kfT"][1][0, 3])
    assert abs(float(out["kf1_tx"]) - (tx + 0.5)) < 1e-4            # local joint BA wrote back through SetPose
    assert abs(float(out["kf1_tx2"]) - (tx + 1.0)) < 1e-4           # ... and so did the points-only local BA
    assert out["gba_marks"] == "7"                                  # loop-closing mode parked the result in the *GBA members
    n_frame = int(out["pose_inliers"].split()[2])      # (the two local BAs erased the flagged match: n_frame - 2 .. n_frame matched)
    assert int(out["pose_inliers"].split()[0]) in [n - (n + 2) // 3 for n in range(n_frame - 2, n_frame + 1)]
    assert out["sim3"] == "17 42.0"
    assert out["ground0"] == "0" and out["ground1"] == "1 -1.0 1.5"
    assert [l.split()[0] for l in log] == ["g2o:OptimizeSim3", "g2o:OptimizeEssentialGraph"]
    assert log[0].split()[1:] == ["5", "10"] and log[1].split()[1] == "2"
    assert "[qsp_hip]" not in err
    assert out["fallbacks"] == "0" and out["failures"] == "0"       # the counters a deployment asserts on


@pytest.mark.parametrize("fail", ["create", "local", "optimize", "pose"])
def test_dropin_fails_loudly_and_leaves_the_map_untouched_by_default(fail):
    """a failing qsp_ba_* / qsp_pose_* call is logged and counted, the BA marks are rolled back, NOTHING is
    written to the map and NOTHING runs on the reference's g2o code -- the entry points return (include/Optimizer.h:78-107 is
    void / no throw; the embedding application reads qsp_optimizer_failure_count()).  Only the loop-closing pass-throughs
    (OptimizeSim3 / OptimizeEssentialGraph, CPU by design) reach g2o."""
    m, out, log, err = _run_dropin(fail)
    names = [l.split()[0] for l in log]
    assert sorted(set(names)) == ["g2o:OptimizeEssentialGraph", "g2o:OptimizeSim3"]          # no BA, no pose optimisation
    assert int(out["failures"]) >= 1 and out["fallbacks"] == "0"
    assert "the map is left untouched" in err and "falls back" not in err
    if fail in ("create", "local"):
        assert "[qsp_hip] qsp_ba_%s" % ("create" if fail == "create" else "local_joint") in err
        assert abs(float(out["kf1_tx"]) - float(m["kfT"][1][0, 3])) < 1e-6   # nobody wrote a pose
        assert out["nBAdone1"] == "0"                                        # a failed joint BA is not counted as done
    if fail in ("create", "optimize"):
        assert out["gba_marks"] == "0"                                       # the global BAs parked nothing in *GBA
    if fail == "pose":
        assert out["pose_inliers"].split()[0] == "0"                         # 0 inliers: Tracking treats the frame as lost


@pytest.mark.parametrize("fail", ["create", "local", "optimize", "pose"])
def test_dropin_falls_back_to_g2o_only_when_the_deployment_opted_in(fail):
    """QSP_SHIM_ALLOW_G2O_FALLBACK=1 restores the hand-over: the error is logged, the map is left as found (the BA marks are
    rolled back so that g2o's own walk finds the same local sets) and the call runs on the reference's g2o code."""
    m, out, log, err = _run_dropin(fail, allow_g2o=True)
    names = [l.split()[0] for l in log]
    if fail in ("create", "local"):
        assert "[qsp_hip] qsp_ba_%s" % ("create" if fail == "create" else "local_joint") in err
        assert names[:2] == ["g2o:LocalJointBundleAdjustment", "g2o:LocalBundleAdjustment"]
        assert log[0].split()[2] == "1" and log[1].split()[2] == "1"          # marks were clean when g2o started
        assert out["nBAdone1"] == "1"                                        # counted by the g2o implementation
        assert abs(float(out["kf1_tx"]) - float(m["kfT"][1][0, 3])) < 1e-6   # the GPU path wrote nothing
    if fail in ("create", "optimize"):
        assert "g2o:GlobalJointBundleAdjustemnt" in names and names.count("g2o:GlobalBundleAdjustemnt") == 2
        assert out["gba_marks"] == "0"
    if fail == "pose":
        assert "g2o:PoseOptimization" in names and out["pose_inliers"].split()[0] == "-7"
    else:
        assert "g2o:PoseOptimization" not in names
    assert "g2o:OptimizeSim3" in names
    assert int(out["fallbacks"]) >= 1 and "falls back to the reference's g2o path" in err     # counted and said, not silent
    assert int(out["failures"]) >= int(out["fallbacks"])


def test_dropin_fallback_can_be_made_fatal():
    """QSP_SHIM_NO_FALLBACK=1: a deployment that must not run on the CPU unnoticed turns the first failed library call into an
    abort (after the message) instead of a g2o run"""
    m = make_map(seed=21, n_kf=7, n_pt=80, n_obj=2)
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_dropin(tmp)
        write_scene(m, os.path.join(tmp, "scene.bin"))
        env = dict(os.environ, QSP_G2O_LOG=os.path.join(tmp, "g2o.log"), QSP_STUB_FAIL="local", QSP_SHIM_NO_FALLBACK="1")
        r = subprocess.run([exe, os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.txt")], env=env, capture_output=True, text=True)
        assert r.returncode != 0 and "[qsp_hip] qsp_ba_local_joint failed" in r.stderr
        assert not os.path.exists(os.path.join(tmp, "g2o.log")) or "g2o:LocalJointBundleAdjustment" not in open(os.path.join(tmp, "g2o.log")).read()
