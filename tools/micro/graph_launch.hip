// Micro-benchmark: N tiny dependent kernel launches on one stream vs one hipGraph replay of the same sequence.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(double* p, int i) { if (threadIdx.x == 0) p[i & 7] += 1.0; }
int main() {
    double* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int N : {8, 21, 37}) {
        for (int w = 0; w < 3; ++w) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d, i); hipStreamSynchronize(s); }
        auto t0 = std::chrono::steady_clock::now();
        const int reps = 200;
        for (int r = 0; r < reps; ++r) { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d, i); hipStreamSynchronize(s); }
        double us_launch = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        hipGraph_t g; hipGraphExec_t ge;
        auto c0 = std::chrono::steady_clock::now();
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d, i);
        hipStreamEndCapture(s, &g);
        hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        double us_build = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count();
        if (e != hipSuccess) { printf("instantiate failed %s\n", hipGetErrorString(e)); return 1; }
        for (int w = 0; w < 3; ++w) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
        double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("N=%2d kernels: %7.1f us as launches + sync, %7.1f us as one graph replay + sync (capture + instantiate %.0f us)\n", N, us_launch, us_graph, us_build);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
