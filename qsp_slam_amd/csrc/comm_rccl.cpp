// comm_rccl.cpp -- the library's own RCCL communicator (multi-GPU rows of SURVEY.md section 8e).
//
// One process per GPU.  The collectives of the hot path (the reduced camera/object system of the joint BA once per
// Levenberg-Marquardt trial, the per-object result rows of the DeepSDF refinement once per batch) are issued on the
// LIBRARY's stream, so a collective is ordered against the kernels that produce / consume its buffer without any host
// synchronisation.  The reference has no counterpart (single process).
//
// librccl is resolved at run time: the copy already mapped into the process (PyTorch ships its own librccl.so.1) is
// preferred so that one process never holds two RCCL instances; a C++ embedder without PyTorch gets /opt/rocm/lib's.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.hpp"
#include "comm_rccl.hpp"

namespace qsp {

static RcclApi g_api;
static std::once_flag g_once;
static std::string g_load_error;

static void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // QSP_RCCL_LIB: an explicit library wins over the copy already in the process (a site-specific RCCL build; the shared-memory
    // stand-in of tests/stub_rccl/ that lets a one-GPU box run this file's code with two ranks).  A path that does not load is
    // an error, not a reason to fall back silently.
    const char* forced = getenv("QSP_RCCL_LIB");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char* e = dlerror();
            g_load_error = std::string("QSP_RCCL_LIB=") + forced + " could not be loaded: " + (e ? e : "");
            return;
        }
    }
    for (const char* n : names) {
        if (h) break;
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // already in the process?
    }
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        const char* e = dlerror();
        g_load_error = std::string("librccl.so.1 not found: ") + (e ? e : "");
        return;
    }
#define SYM(field, name)                                                   \
    g_api.field = (decltype(g_api.field))dlsym(h, name);                   \
    if (!g_api.field) { g_load_error = "librccl: missing symbol " name; return; }
    SYM(get_unique_id, "ncclGetUniqueId")
    SYM(comm_init_rank, "ncclCommInitRank")
    SYM(comm_destroy, "ncclCommDestroy")
    SYM(all_reduce, "ncclAllReduce")
    SYM(all_gather, "ncclAllGather")
    SYM(get_error_string, "ncclGetErrorString")
#undef SYM
    g_api.stub_counts = (decltype(g_api.stub_counts))dlsym(h, "qsp_stub_rccl_counts");      // (only tests/stub_rccl has it)
    g_api.ok = true;
}

const RcclApi* rccl_api() {
    std::call_once(g_once, load_rccl);
    return g_api.ok ? &g_api : nullptr;
}

int rccl_fail(ncclResult_t r, const char* what) {
    const RcclApi* a = rccl_api();
    std::string m = std::string(what) + ": " + ((a && a->get_error_string) ? a->get_error_string(r) : "rccl error");
    return qsp_fail(QSP_ERR_DEVICE, m.c_str());
}

}  // namespace qsp

using namespace qsp;

struct qsp_comm {
    ncclComm_t nccl = nullptr;
    int rank = 0, world = 1, device = 0;
    bool owned = false;
};

static_assert(sizeof(ncclUniqueId) == QSP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int qsp_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return qsp_fail(QSP_ERR_INVALID, "qsp_comm_unique_id: null argument");
    const RcclApi* a = rccl_api();
    if (!a) return qsp_fail(QSP_ERR_DEVICE, g_load_error.c_str());
    ncclUniqueId id;
    ncclResult_t r = a->get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return QSP_OK;
}

extern "C" int qsp_comm_create(const uint8_t* id_in, int32_t rank, int32_t world, int device, qsp_comm** out) {
    if (!id_in || !out || world < 1 || rank < 0 || rank >= world) return qsp_fail(QSP_ERR_INVALID, "qsp_comm_create: bad argument");
    const RcclApi* a = rccl_api();
    if (!a) return qsp_fail(QSP_ERR_DEVICE, g_load_error.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return qsp_fail(QSP_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return qsp_fail(QSP_ERR_INVALID, "device index out of range");
    QSP_HIP(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = a->comm_init_rank(&c, world, id, rank);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    qsp_comm* q = new qsp_comm();
    q->nccl = c; q->rank = rank; q->world = world; q->device = device; q->owned = true;
    *out = q;
    return QSP_OK;
}

extern "C" int qsp_comm_adopt(void* nccl_comm, int32_t rank, int32_t world, int device, qsp_comm** out) {
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return qsp_fail(QSP_ERR_INVALID, "qsp_comm_adopt: bad argument");
    if (!rccl_api()) return qsp_fail(QSP_ERR_DEVICE, g_load_error.c_str());
    qsp_comm* q = new qsp_comm();
    q->nccl = (ncclComm_t)nccl_comm; q->rank = rank; q->world = world; q->device = device; q->owned = false;
    *out = q;
    return QSP_OK;
}

extern "C" void qsp_comm_destroy(qsp_comm* c) {
    if (!c) return;
    if (c->owned && c->nccl) {
        (void)hipSetDevice(c->device);
        const RcclApi* a = rccl_api();
        if (a) (void)a->comm_destroy(c->nccl);
    }
    delete c;
}

extern "C" void* qsp_comm_nccl(qsp_comm* c) { return c ? (void*)c->nccl : nullptr; }
// Test hook: with the stand-in library of tests/stub_rccl (QSP_RCCL_LIB) the collectives that ran with more than one rank on this
// communicator -- [sum all-reduces, max all-reduces, all-gathers]; QSP_ERR_UNSUPPORTED with a real RCCL.
extern "C" int qsp_comm_stub_counts(qsp_comm* c, int64_t* out3) {
    const RcclApi* a = rccl_api();
    if (!c || !out3 || !a) return qsp_fail(QSP_ERR_INVALID, "qsp_comm_stub_counts: bad argument");
    if (!a->stub_counts) return qsp_fail(QSP_ERR_UNSUPPORTED, "qsp_comm_stub_counts: the loaded librccl is not the test stand-in");
    long long v[3] = {0, 0, 0};
    a->stub_counts(c->nccl, v);
    for (int i = 0; i < 3; ++i) out3[i] = v[i];
    return QSP_OK;
}
extern "C" int32_t qsp_comm_rank(qsp_comm* c) { return c ? c->rank : -1; }
extern "C" int32_t qsp_comm_world(qsp_comm* c) { return c ? c->world : 0; }

extern "C" int qsp_comm_allreduce_f64(qsp_comm* c, double* device_buf, int64_t count, void* hip_stream) {
    if (!c || !device_buf || count < 0) return qsp_fail(QSP_ERR_INVALID, "qsp_comm_allreduce_f64: bad argument");
    if (count == 0) return QSP_OK;
    QSP_HIP(hipSetDevice(c->device));
    const RcclApi* a = rccl_api();
    ncclResult_t r = a->all_reduce(device_buf, device_buf, (size_t)count, ncclDouble, ncclSum, c->nccl, (hipStream_t)hip_stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
    return QSP_OK;
}

extern "C" int qsp_comm_allgather_f32(qsp_comm* c, const float* send, float* recv, int64_t count_per_rank, void* hip_stream) {
    if (!c || !send || !recv || count_per_rank < 0) return qsp_fail(QSP_ERR_INVALID, "qsp_comm_allgather_f32: bad argument");
    if (count_per_rank == 0) return QSP_OK;
    QSP_HIP(hipSetDevice(c->device));
    const RcclApi* a = rccl_api();
    ncclResult_t r = a->all_gather(send, recv, (size_t)count_per_rank, ncclFloat, c->nccl, (hipStream_t)hip_stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
    return QSP_OK;
}
