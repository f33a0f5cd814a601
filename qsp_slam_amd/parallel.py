"""Multi-GPU layer of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" on
CPU for tests).  PyTorch is plumbing here (process group + one all_gather), not the product.

Path A shards naturally: the work unit is an object with all of its yaw-flip hypotheses (kept together so that the
arg-min of src/LocalMapping_util.cc:748-752 is local).  There is NO collective inside the Gauss-Newton iterations; the
per-object results (4x4 pose, 64 code, loss, is_good = 82 floats) are all-gathered once at the end.
Path B: independent key-frame windows are replicas (no exchange)."""
import numpy as np

RESULT_WIDTH = 16 + 64 + 2


def shard_objects(n_obj, rank, world):
    """round-robin by object index: object o lives on rank o % world"""
    return list(range(rank, n_obj, world))


def owner_of(obj, world):
    return obj % world


def pack_results(results, with_flip=False):
    """list of reconstruct_object results (attr-dicts) -> (n, 82) float32 table (83 with the kept flip index of
    Optimizer.refine_detections)"""
    out = np.zeros((len(results), RESULT_WIDTH + (1 if with_flip else 0)), np.float32)
    for i, r in enumerate(results):
        if r.is_good:
            out[i, :16] = np.asarray(r.t_cam_obj, np.float32).reshape(-1)
            out[i, 16:80] = np.asarray(r.code, np.float32)[:64]
        out[i, 80] = r.loss
        out[i, 81] = 1.0 if r.is_good else 0.0
        if with_flip:
            out[i, 82] = float(r.kept_flip)
    return out


def unpack_results(table):
    from .reconstruct.utils import ForceKeyErrorDict
    res = []
    for row in table:
        if row[81] > 0.5:
            res.append(ForceKeyErrorDict(t_cam_obj=row[:16].reshape(4, 4).copy(), code=row[16:80].copy(), is_good=True,
                                         loss=float(row[80])))
        else:
            res.append(ForceKeyErrorDict(t_cam_obj=None, code=None, is_good=False, loss=float(row[80])))
        if len(row) > RESULT_WIDTH:
            res[-1]["kept_flip"] = int(row[82])
    return res


def gather_object_results(local_table, n_obj, rank, world, device=None):
    """all_gather of the per-object result rows; returns the full (n_obj, 82) table in object order on every rank.
    local_table: rows of shard_objects(n_obj, rank, world), in that order."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return np.asarray(local_table, np.float32)
    width = int(np.shape(local_table)[1])
    per = (n_obj + world - 1) // world                      # padded shard size
    buf = torch.zeros(per, width, dtype=torch.float32, device=device)
    if len(local_table):
        buf[: len(local_table)] = torch.from_numpy(np.ascontiguousarray(local_table, np.float32)).to(buf.device)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    full = np.zeros((n_obj, width), np.float32)
    for r in range(world):
        idx = shard_objects(n_obj, r, world)
        full[idx] = parts[r][: len(idx)].cpu().numpy()
    return full


def refine_objects_sharded(optimizer, objects, flip_sample_num, rank, world, device=None):
    """Every rank refines its shard of `objects` (list of dicts as Optimizer.reconstruct_objects_batched takes) on its own
    GPU and all ranks end with the complete, selected result list."""
    mine = shard_objects(len(objects), rank, world)
    local = optimizer.reconstruct_objects_batched([objects[i] for i in mine], flip_sample_num=flip_sample_num,
                                                  select=True) if mine else []
    table = gather_object_results(pack_results(local), len(objects), rank, world, device=device)
    return unpack_results(table)


def refine_detections_sharded(optimizer, detections, flip_sample_num, rank, world, device=None):
    """As refine_objects_sharded for the world-frame entry point (Optimizer.refine_detections): detection
    d is marshalled, refined over its yaw flips and selected on rank d % world; one all_gather of 83 floats per detection."""
    mine = shard_objects(len(detections), rank, world)
    local = optimizer.refine_detections([detections[i] for i in mine], flip_sample_num=flip_sample_num) if mine else []
    table = gather_object_results(pack_results(local, with_flip=True), len(detections), rank, world, device=device)
    return unpack_results(table)


class _DevArray(object):
    """exposes a raw device address as a CUDA-array-interface object so that torch can view it without a copy"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(ptr, False), version=2)


class RcclComm(object):
    """The library's own RCCL communicator (qsp_comm_*, include/qsp_hip.h): one rank per GPU, collectives issued on the
    library's HIP streams.  The 128-byte unique id is made on rank 0 and handed to the other ranks by `exchange`, a
    callable bytes -> bytes that every rank calls (default: torch.distributed.broadcast_object_list on the default group,
    any backend).  This is the production path of BaProblem.set_shard_rccl."""

    def __init__(self, rank, world, device, exchange=None):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        ident = (C.c_uint8 * 128)()
        raw, err = bytes(ident), None
        if self.rank == 0:
            try:
                _lib.check(L.qsp_comm_unique_id(ident))
                raw = bytes(ident)
            except Exception as e:      # every rank still takes part in the exchange below: a rank 0 that raised here would
                raw, err = b"", e       # leave the others blocked in it
        if self.world > 1:
            if exchange is None:
                import torch.distributed as dist
                box = [raw]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                raw = exchange(raw)
        if len(raw) != 128:
            raise err if err is not None else RuntimeError("RcclComm: rank 0 could not create the RCCL unique id")
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        self.handle = C.c_void_p()
        _lib.check(L.qsp_comm_create(ident, self.rank, self.world, self.device, C.byref(self.handle)))

    def nccl(self):
        from . import _lib
        return _lib.lib().qsp_comm_nccl(self.handle)

    def allreduce_f64(self, dev_ptr, n, stream=0):
        from . import _lib
        _lib.check(_lib.lib().qsp_comm_allreduce_f64(self.handle, int(dev_ptr), int(n), int(stream) or None))

    def allgather_f32(self, send_ptr, recv_ptr, n_per_rank, stream=0):
        from . import _lib
        _lib.check(_lib.lib().qsp_comm_allgather_f32(self.handle, int(send_ptr), int(recv_ptr), int(n_per_rank),
                                                     int(stream) or None))

    def stub_counts(self):
        """(tests) with the stand-in librccl of tests/stub_rccl: [sum all-reduces, max all-reduces, all-gathers] that ran with
        more than one rank on this communicator; raises with a real RCCL"""
        from . import _lib
        out = np.zeros(3, np.int64)
        _lib.check(_lib.lib().qsp_comm_stub_counts(self.handle, _lib.i64ptr(out)))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            from . import _lib
            _lib.lib().qsp_comm_destroy(self.handle)
            self.handle.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchAllreduce(object):
    """An all-reduce hook for qsp_ba_set_shard (the callback form: `hook(dev_ptr, count, hip_stream)`) over
    torch.distributed: backend "nccl" is RCCL on ROCm.  The library's stream and torch's stream are different streams, so
    the hook synchronises both around the collective -- two host synchronisations per collective.  Kept for applications
    that must route every collective through their own process group; the production path is RcclComm +
    BaProblem.set_shard_rccl, which has none."""

    def __init__(self, device, group=None):
        self.device = device
        self.group = group

    def __call__(self, ptr, n, stream):
        import torch
        import torch.distributed as dist
        from . import _lib
        hip = _hip()
        hip.hipStreamSynchronize(_lib.C.c_void_p(stream))
        t = torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", self.device))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        torch.cuda.synchronize(self.device)


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        import ctypes
        _HIP = ctypes.CDLL("libamdhip64.so")
    return _HIP


class GlooAllreduce(object):
    """Callback hook for qsp_ba_set_shard that stages through the host and sums with a CPU process group (gloo): the way
    to run SEVERAL ranks on ONE GPU (tests, rehearsals) -- RCCL refuses two ranks on the same device."""

    def __init__(self, group=None):
        self.group = group

    def __call__(self, ptr, n, stream):
        import ctypes
        import torch
        import torch.distributed as dist
        hip = _hip()
        host = np.empty(n, np.float64)
        hip.hipStreamSynchronize(ctypes.c_void_p(stream))
        hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(8 * n), 2)
        t = torch.from_numpy(host)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        hip.hipMemcpy(ctypes.c_void_p(ptr), host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(8 * n), 1)


class ThreadAllreduce(object):
    """Test double for ONE GPU: `world` Python threads, each driving its own BaProblem shard on the same device, meet at a
    barrier; the sum is formed on the host in rank order.  Exercises the sharded algorithm where only one GPU exists."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def hook(self, rank):
        import ctypes

        def fn(ptr, n, stream):
            hip = _hip()
            hip.hipStreamSynchronize(ctypes.c_void_p(stream))
            host = np.empty(n, np.float64)
            hip.hipMemcpy(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(8 * n), 2)
            self.slots[rank] = host
            self.barrier.wait()
            tot = np.zeros(n, np.float64)
            for r in range(self.world):
                tot += self.slots[r]
            self.barrier.wait()
            hip.hipMemcpy(ctypes.c_void_p(ptr), tot.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(8 * n), 1)
        return fn
