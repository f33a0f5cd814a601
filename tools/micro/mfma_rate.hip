// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 with 1 or 2 waves per SIMD (shader clock cycles per MFMA per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o build/exp/mfma_rate tools/micro/mfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a = 0; a < NACC; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int threads, int blocks, const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    const int iters = 20000;
    hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 64 / 4 > 0 ? threads / 64 / 4 : 1;
    const double mfma_per_simd = 16.0 * iters * waves_per_simd;
    const double tf = 4096.0 * 16.0 * iters * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
    printf("%-34s threads %4d blocks %4d: %.2f cycles/MFMA/SIMD (block 0), %.3f ms, %.1f TFLOP/s, clock %.3f GHz\n", name, threads, blocks,
           (double)h[0] / mfma_per_simd, ms, tf, (double)h[0] / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}

// the same loop with 16 different random operand pairs per lane (realistic toggling), long enough to reach steady state
__global__ void k_rate_random(float* out, unsigned long long* cyc, unsigned long long* rt, int iters, const float* rnd) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) { x[i] = rnd[(threadIdx.x * 16 + i) % 4096]; y[i] = rnd[(threadIdx.x * 16 + i + 1777) % 4096]; }
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[4 * u + a], y[4 * a + u], acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

void run_random(int threads, int iters) {
    const int blocks = 256;
    float *out, *rnd; unsigned long long *cyc, *rt;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipMalloc(&rt, sizeof(unsigned long long) * blocks);
    hipMalloc(&rnd, sizeof(float) * 4096);
    float h[4096];
    unsigned s = 12345;
    for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate_random, dim3(blocks), dim3(threads), 0, 0, out, cyc, rt, iters, rnd);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc, hr;
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hr, rt, 8, hipMemcpyDeviceToHost);
        const double tf = 4096.0 * 16.0 * iters * (threads / 64) * blocks / (ms * 1e-3) / 1e12;
        printf("random operands, %d waves/SIMD, %.1f ms: %.2f cycles/MFMA/SIMD, %.1f TFLOP/s, shader clock %.3f GHz\n", threads / 256, ms,
               (double)hc / (16.0 * iters * (threads / 256)), tf, (double)hc / (double)hr * 0.1);
    }
}

int main() {
    run_random(512, 200000);
    run_random(256, 400000);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(256, 256, "4 acc, 1 wave/SIMD, 1 WG/CU");
        run<4>(512, 256, "4 acc, 2 waves/SIMD, 1 WG/CU");
        run<2>(256, 256, "2 acc, 1 wave/SIMD");
        run<1>(256, 256, "1 acc (dependent), 1 wave/SIMD");
        run<1>(512, 256, "1 acc (dependent), 2 waves/SIMD");
        run<4>(256, 1, "4 acc, 1 wave/SIMD, ONE CU only");
        run<4>(512, 1, "4 acc, 2 waves/SIMD, ONE CU only");
    }
    return 0;
}
