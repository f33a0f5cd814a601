// comm_rccl.hpp -- run-time binding of the handful of RCCL entry points the hot path uses (see comm_rccl.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace qsp {

struct RcclApi {
    bool ok = false;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*get_error_string)(ncclResult_t) = nullptr;
    void (*stub_counts)(ncclComm_t, long long*) = nullptr;     // tests/stub_rccl only
};

const RcclApi* rccl_api();              // nullptr when librccl cannot be resolved (qsp_last_error() then says why)
int rccl_fail(ncclResult_t r, const char* what);

}  // namespace qsp
