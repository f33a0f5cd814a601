// TEST INFRASTRUCTURE ONLY -- a stand-in librccl for ONE purpose: to execute the library's RCCL-on-stream code path
// (csrc/comm_rccl.cpp, qsp_ba_set_shard_rccl: ncclAllReduce SUM / MAX on the BA's own stream) with MORE THAN ONE RANK on a
// one-GPU box, where the real RCCL refuses two ranks on one device (tools/rccl_probe.py).  It implements exactly the entry points
// comm_rccl.cpp binds -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce (double / float; sum, max),
// ncclAllGather, ncclGetErrorString -- over POSIX shared memory between processes of one host:
//   * a collective honours its stream argument by hipStreamSynchronize(stream) + synchronous copies: everything enqueued on the
//     stream before the call has finished, everything enqueued after it sees the result;
//   * ranks are summed in rank order on every rank (identical bits on all ranks, deterministic).
// It is selected with QSP_RCCL_LIB=<path> (an explicit path wins over the copy of librccl already in the process) and is never
// loaded otherwise.  Nothing here is performance code.
//   hipcc -shared -fPIC -O1 -o tests/stub_rccl/librccl_stub.so tests/stub_rccl/stub_rccl.cpp -lrt -lpthread
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
constexpr size_t CHUNK = 8u << 20;          // bytes per rank and round
constexpr int MAX_WORLD = 8;

struct Shared {
    std::atomic<int> arrived;               // sense-reversing barrier
    std::atomic<int> sense;
    std::atomic<int> attached;
    std::atomic<long long> n_allreduce_sum, n_allreduce_max, n_allgather;   // calls that ran with world > 1 (read by the test)
    char pad[64];
    unsigned char slot[MAX_WORLD][CHUNK];
};

struct Comm {
    Shared* sh = nullptr;
    int rank = 0, world = 1, local_sense = 0;
    char name[64] = {0};
    std::vector<unsigned char> host;
};

bool barrier(Comm* c) {
    c->local_sense ^= 1;
    if (c->sh->arrived.fetch_add(1) + 1 == c->world) {
        c->sh->arrived.store(0);
        c->sh->sense.store(c->local_sense);
        return true;
    }
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned long spin = 0; c->sh->sense.load() != c->local_sense; ++spin) {
        if ((spin & 0xfff) == 0xfff) {
            usleep(50);
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (t1.tv_sec - t0.tv_sec > 120) return false;         // a peer died: fail instead of hanging the test
        }
    }
    return true;
}

size_t elem_size(ncclDataType_t t) { return t == ncclDouble ? 8 : (t == ncclFloat ? 4 : 0); }

template <typename T>
void reduce(Comm* c, T* out, size_t n, ncclRedOp_t op) {
    for (size_t i = 0; i < n; ++i) {
        T v = reinterpret_cast<const T*>(c->sh->slot[0])[i];
        for (int r = 1; r < c->world; ++r) {
            const T x = reinterpret_cast<const T*>(c->sh->slot[r])[i];
            v = (op == ncclMax) ? (x > v ? x : v) : v + x;
        }
        out[i] = v;
    }
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<int> counter{0};
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/qsp_stub_rccl_%d_%d", (int)getpid(), counter.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { delete c; return ncclSystemError; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared*)p;                       // (a fresh shm object is zero-filled: counters and barrier start at 0)
    c->sh->attached.fetch_add(1);
    c->host.resize(CHUNK);
    if (!barrier(c)) { delete c; return ncclSystemError; }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    if (c->sh) {
        if (c->sh->attached.fetch_sub(1) == 1) shm_unlink(c->name);
        munmap(c->sh, sizeof(Shared));
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = elem_size(dt);
    if (!c || !es || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->world > 1 && c->rank == 0) (op == ncclSum ? c->sh->n_allreduce_sum : c->sh->n_allreduce_max).fetch_add(1);
    const size_t per = CHUNK / es;
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (hipMemcpy(c->sh->slot[c->rank], (const char*)send + off * es, n * es, hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        if (dt == ncclDouble) reduce(c, (double*)c->host.data(), n, op);
        else reduce(c, (float*)c->host.data(), n, op);
        if (!barrier(c)) return ncclSystemError;              // everybody has read every slot
        if (hipMemcpy((char*)recv + off * es, c->host.data(), n * es, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    const size_t es = elem_size(dt);
    if (!c || !es) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->world > 1 && c->rank == 0) c->sh->n_allgather.fetch_add(1);
    const size_t per = CHUNK / es;
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        if (hipMemcpy(c->sh->slot[c->rank], (const char*)send + off * es, n * es, hipMemcpyDeviceToHost) != hipSuccess)
            return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
        for (int r = 0; r < c->world; ++r)
            if (hipMemcpy((char*)recv + ((size_t)r * count + off) * es, c->sh->slot[r], n * es, hipMemcpyHostToDevice) != hipSuccess)
                return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclSystemError;
    }
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "success" : "stub rccl error"; }

// test hook: collectives that ran with more than one rank on this communicator: [sum all-reduces, max all-reduces, all-gathers]
void qsp_stub_rccl_counts(ncclComm_t comm, long long* out3) {
    Comm* c = (Comm*)comm;
    out3[0] = c->sh->n_allreduce_sum.load();
    out3[1] = c->sh->n_allreduce_max.load();
    out3[2] = c->sh->n_allgather.load();
}
}
