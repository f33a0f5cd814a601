// common.hpp -- error plumbing shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/qsp_hip.h"

namespace qsp {

std::string& last_error_ref();

inline int qsp_fail(int code, const char* msg) {
    last_error_ref() = msg ? msg : "";
    return code;
}

#define QSP_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t qsp_e_ = (call);                                                               \
        if (qsp_e_ != hipSuccess) {                                                               \
            return ::qsp::qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(qsp_e_));                    \
        }                                                                                         \
    } while (0)

}  // namespace qsp
