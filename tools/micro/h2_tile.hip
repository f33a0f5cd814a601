// Micro-benchmark of the split-fp16 tile (mlp_tile_h2) outside the library: one workgroup per CU, `tiles` tiles each, weights
// all zero-ish (timing only).  Build variants with -DQSP_H2_EXP=<bits> (see sdf_mlp.hpp) and -DBWDV=true|false.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DQSP_H2_EXP=n] [-DBWDV=true] tools/micro/h2_tile.hip -o h2_tile && ./h2_tile
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../../qsp_slam_amd/csrc/sdf_mlp.hpp"
using namespace qsp;
#ifndef PFV
#define PFV 2
#endif
#ifndef BWDV
#define BWDV false
#endif
__global__ __launch_bounds__(H2_THREADS) void kt(const MlpParams* P, float* y, int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    float amax = 0.f;
    for (int i = threadIdx.x; i < HID; i += H2_THREADS) { s.c0[i] = 0.01f * (i & 7); s.c4[i] = 0.02f; }
    for (int t = 0; t < tiles; ++t) {
        __syncthreads();
        if (threadIdx.x < 64) {
            s.xin[4 * threadIdx.x] = 0.1f; s.xin[4 * threadIdx.x + 1] = 0.2f; s.xin[4 * threadIdx.x + 2] = 0.3f; s.xin[4 * threadIdx.x + 3] = 0.f;
        }
        __syncthreads();
        mlp_tile_h2<BWDV, PFV>(s, P, amax, t == 0);
        if (threadIdx.x < 64) y[blockIdx.x * 64 + threadIdx.x] = s.y[threadIdx.x] + (BWDV ? s.act[threadIdx.x * LDG] : 0.f);
    }
    if (!(amax <= H2_MAX)) *P->range_flag = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 64;
    MlpParams P = {};
    auto dev = [&](size_t bytes, float fill) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes + 16384) != hipSuccess) return nullptr;
        std::vector<float> h(bytes / 4 + 4096, fill);
        (void)hipMemcpy(p, h.data(), bytes + 16384, hipMemcpyHostToDevice);
        return p;
    };
    const bool shared = argc > 2 && atoi(argv[2]);      // every layer reads the same 1 MiB: the weight set fits any L2
    for (int l = 0; l < 8; ++l) {
        P.wfh[l] = (shared && l > 1) ? P.wfh[1] : (const float4*)dev((size_t)16 * 32 * 2 * 64 * 16, 0.f);
        P.wbh[l] = (shared && l) ? P.wfh[1] : (const float4*)dev((size_t)16 * 32 * 2 * 64 * 16, 0.f);
        P.bias[l] = (const float*)dev(512 * 4, 0.01f);
    }
    P.wbh4s = (const float4*)dev((size_t)3 * 32 * 2 * 64 * 16, 0.f);
    P.w8 = (const float*)dev(512 * 4, 0.01f);
    P.w0x = (const float4*)dev(128 * 3 * 16, 0.01f);
    P.range_flag = (int*)dev(64, 0.f);
    MlpParams* Pd = (MlpParams*)dev(sizeof(P), 0.f);
    CK(hipMemcpy(Pd, &P, sizeof(P), hipMemcpyHostToDevice));
    float* y = (float*)dev(256 * 64 * 4, 0.f);
    CK(hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpSmem)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kt, dim3(256), dim3(H2_THREADS), sizeof(MlpSmem), 0, Pd, y, tiles);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        const double flop = (BWDV ? 2.0 : 1.0) * 3.671e6 * 64.0 * tiles * 256;
        printf("exp %d pf %d shared %d bwd %d: %d tiles/CU  %.3f ms  %.1f us/tile  %.1f TFLOP/s effective\n", (int)QSP_H2_EXP, PFV, (int)shared, (int)BWDV, tiles, ms,
               1e3 * ms / tiles, flop / ms / 1e9);
    }
#ifdef QSP_H2_STAMPS
    unsigned long long ts[64];
    int nts = 0;
    CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(qsp_h2_ts), sizeof(ts)));
    CK(hipMemcpyFromSymbol(&nts, HIP_SYMBOL(qsp_h2_nts), sizeof(nts)));
    printf("stamps of workgroup 0's last tile, deltas in s_memtime ticks:");
    for (int i = 1; i < nts; ++i) printf(" %lld", (long long)(ts[i] - ts[i - 1]));
    printf("\n");
#endif
    return 0;
}
