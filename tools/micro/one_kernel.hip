// One kernel of csrc/sdf_kernels.hpp as its own translation unit: compiles in seconds, for ISA listings and resource metadata
// (tools/spill_map.py, tests/test_isa_budget.py).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only
//   -DQSP_ONE_KERNEL='k_mlp_jtj_h2<2>' tools/micro/one_kernel.hip -o /tmp/k.s
#include "../../qsp_slam_amd/csrc/sdf_kernels.hpp"
#ifndef QSP_ONE_KERNEL
#define QSP_ONE_KERNEL k_mlp_jtj_h2<2>
#endif
namespace {
__attribute__((used)) auto* const qsp_one_kernel = &qsp::QSP_ONE_KERNEL;
}
