// How fast can ONE workgroup (1024 threads, one compute unit) pull L2-resident data?  dwordx2 against dwordx4 loads, 16 per wave in
// flight: the bound of the BA's backward substitution (csrc/ba_solver.hip:k_chol_back), which walks the factor with one workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o cu_stream tools/micro/cu_stream.hip && ./cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>   // W = doubles per lane per load (1: dwordx2, 2: dwordx4)
__global__ __launch_bounds__(1024) void k_stream(const double* __restrict__ src, size_t n_doubles, double* out, int reps) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double acc = 0;
    for (int r = 0; r < reps; ++r) {
        // each wave walks its own 1/16 of the buffer in wave-loads of 64 * W doubles, 16 in flight
        const size_t per_wave = n_doubles / 16;
        const double* base = src + wave * per_wave;
        for (size_t off = 0; off + 16 * 64 * W <= per_wave; off += 16 * 64 * W) {
            double v[16 * W];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (W == 1) v[i] = base[off + i * 64 + lane];
                else {
                    const double2 d2 = *reinterpret_cast<const double2*>(base + off + i * 128 + 2 * lane);
                    v[2 * i] = d2.x; v[2 * i + 1] = d2.y;
                }
            }
#pragma unroll
            for (int i = 0; i < 16 * W; ++i) acc += v[i];
        }
    }
    out[t] = acc;
}
int main() {
    const size_t bytes = 2u << 20;      // 2 MB: stays in one XCD's L2
    double *src, *out;
    hipMalloc(&src, bytes); hipMalloc(&out, 1024 * 8); hipMemset(src, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int W = 1; W <= 2; ++W) {
        const int reps = 20;
        float ms = 0;
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(a);
            if (W == 1) hipLaunchKernelGGL(k_stream<1>, dim3(1), dim3(1024), 0, 0, src, bytes / 8, out, reps);
            else hipLaunchKernelGGL(k_stream<2>, dim3(1), dim3(1024), 0, 0, src, bytes / 8, out, reps);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        }
        printf("dwordx%d loads: %.1f GB/s through one compute unit (%.2f ms for %d x 2 MB)\n", 2 * W, reps * bytes / (ms * 1e-3) / 1e9, ms, reps);
    }
    return 0;
}
