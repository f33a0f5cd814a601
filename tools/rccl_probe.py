"""Probe: does RCCL accept `world` ranks on ONE device?  (Used once to decide how the multi-rank GPU tests exchange data.)
  python tools/rccl_probe.py            parent: spawns the ranks with a 90 s limit
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(rank, world, path):
    import numpy as np
    import ctypes
    from qsp_slam_amd import parallel

    def exchange(raw):
        if rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.write(raw)
            os.rename(path + ".tmp", path)
            return raw
        for _ in range(600):
            if os.path.exists(path):
                return open(path, "rb").read()
            time.sleep(0.05)
        raise SystemExit("no id")
    c = parallel.RcclComm(rank, world, 0, exchange=exchange)
    hip = parallel._hip()
    buf = ctypes.c_void_p()
    hip.hipMalloc(ctypes.byref(buf), 64)
    host = np.full(8, float(rank + 1))
    hip.hipMemcpy(buf, host.ctypes.data_as(ctypes.c_void_p), 64, 1)
    c.allreduce_f64(buf.value, 8)
    hip.hipDeviceSynchronize()
    hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), buf, 64, 2)
    print("rank", rank, "sum", host[0], flush=True)
    c.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3])
    else:
        world = 2
        path = "/tmp/qsp_rccl_id_%d" % os.getpid()
        ps = [subprocess.Popen([sys.executable, __file__, str(r), str(world), path]) for r in range(world)]
        t0 = time.time()
        while time.time() - t0 < 90 and any(p.poll() is None for p in ps):
            time.sleep(0.5)
        for p in ps:
            if p.poll() is None:
                p.kill()
                print("rank killed after 90 s")
        print("exit codes", [p.returncode for p in ps])
