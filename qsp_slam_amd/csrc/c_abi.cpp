// c_abi.cpp -- library-wide C-ABI entry points (version, error text, device count).
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace qsp {
std::string& last_error_ref() {
    static thread_local std::string e;
    return e;
}
}  // namespace qsp

extern "C" const char* qsp_last_error(void) { return qsp::last_error_ref().c_str(); }
extern "C" int qsp_version(void) { return 1; }
extern "C" int qsp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
