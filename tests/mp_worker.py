"""One rank of a multi-process GPU test (started by tests/test_gpu_multiproc.py as a fresh child process, one per rank; all
ranks share device 0 of the one-GPU box, so the data-path reductions go over gloo -- RCCL refuses two ranks on one device).

  python tests/mp_worker.py <mode> <rank> <world> <port> <out.npz>
     ba      one landmark-sharded local joint BA of a fixed scene (qsp_ba_set_shard + the gloo hook)
     ba_rccl the same BA through RcclComm + qsp_ba_set_shard_rccl (ncclAllReduce on the BA's stream) on tests/stub_rccl's librccl
     refine  object-sharded DeepSDF refinement + all_gather of the kept results (parallel.refine_objects_sharded)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BA_SCENE = dict(seed=77, n_kf=12, n_pt=600, n_obj=4, stereo_frac=0.3, outlier_frac=0.06)
REFINE = dict(seed=91, n_obj=5, n_pts=300, n_fg=64, n_bg=32)


def refine_config():
    from qsp_slam_amd.reconstruct.utils import ForceKeyErrorDict
    return ForceKeyErrorDict(data_type="Redwood", optimizer=dict(
        code_len=64, num_depth_samples=50, cut_off_threshold=0.01,
        joint_optim=dict(k1=10.0, k2=100.0, k3=2.5, k4=0.0, b1=0.2, b2=0.02, learning_rate=1.0, scale_damping=100.0,
                         num_iterations=3)))


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qsp_slam_amd import parallel, synth
    if mode == "ba":
        from qsp_slam_amd.ba import BaProblem
        sc = synth.make_ba_scene(**BA_SCENE)
        p = BaProblem(sc)
        p.set_shard(rank, world, parallel.GlooAllreduce())
        t1, t2 = p.local_joint_ba()
        kf, pt, ob = p.state()
        e = p.edges()
        np.savez(out, kf=kf, pt=pt, ob=ob, chi2_1=t1["chi2"], chi2_2=t2["chi2"], lam_1=t1["lam"], lam_2=t2["lam"],
                 trials_1=t1["trials"], trials_2=t2["trials"], mono_chi2=e["mono_chi2"], oe_chi2=e["oe_chi2"])
        p.close()
    elif mode == "ba_rccl":
        # the PRODUCTION path -- RcclComm + qsp_ba_set_shard_rccl: ncclAllReduce (SUM and MAX) on the BA's own stream -- with the
        # shared-memory stand-in librccl of tests/stub_rccl (QSP_RCCL_LIB, set by the test): two ranks on one device
        import torch
        from qsp_slam_amd.ba import BaProblem
        assert os.environ.get("QSP_RCCL_LIB"), "the test must select the stand-in library"
        comm = parallel.RcclComm(rank, world, 0)              # unique id broadcast over the gloo group
        sc = synth.make_ba_scene(**BA_SCENE)
        p = BaProblem(sc)
        p.set_shard_rccl(comm)
        t1, t2 = p.local_joint_ba()
        kf, pt, ob = p.state()
        e = p.edges()
        # the library's all-gather entry point with two ranks as well
        send = torch.full((5,), float(rank + 1), dtype=torch.float32, device="cuda:0")
        recv = torch.zeros(5 * world, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        comm.allgather_f32(send.data_ptr(), recv.data_ptr(), 5)
        torch.cuda.synchronize()
        np.savez(out, kf=kf, pt=pt, ob=ob, chi2_1=t1["chi2"], chi2_2=t2["chi2"], lam_1=t1["lam"], lam_2=t2["lam"],
                 trials_1=t1["trials"], trials_2=t2["trials"], mono_chi2=e["mono_chi2"], oe_chi2=e["oe_chi2"],
                 counts=comm.stub_counts(), gathered=recv.cpu().numpy())
        p.close()
        dist.barrier()
        comm.close()
    elif mode == "refine":
        from qsp_slam_amd import DeepSdfDecoder
        from qsp_slam_amd.reconstruct.optimizer import Optimizer
        dec = DeepSdfDecoder.from_npz(os.path.join(ROOT, "tests", "golden", "decoder_8x512.npz"), device=0)
        opt = Optimizer(dec, refine_config())
        objs = synth.make_object_views(REFINE["seed"], REFINE["n_obj"], REFINE["n_pts"], n_fg=REFINE["n_fg"], n_bg=REFINE["n_bg"])
        objs = [dict(t_cam_obj=o["t_cam_obj"], pts=o["pts"], rays=o["rays"], depth=o["depth"]) for o in objs]
        res = parallel.refine_objects_sharded(opt, objs, 4, rank, world)
        np.savez(out, table=parallel.pack_results(res))
    else:
        raise SystemExit("unknown mode " + mode)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
