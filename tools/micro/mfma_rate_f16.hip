// Micro-benchmark: what the fp16 matrix pipe sustains when NOTHING else runs -- v_mfma_f32_32x32x16_f16 back to back on every
// SIMD of the chip for ~0.2 s, with realistic operands (random fp16 values of the decoder's magnitudes: switching activity sets
// the power and the power sets the clock).  Reports TFLOP/s of the pipe, cycles per MFMA and the shader clock (cycle counter
// against the constant 100 MHz real-time counter).  The dense fp16 peak of the data sheet (2.5 PFLOP/s) assumes 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate_f16 tools/micro/mfma_rate_f16.hip && ./mfma_rate_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>     // 0: zero operands, 1: random operands (|x| < 0.6), 2 / 3: random, with 1 in 4 / 1 in 2 issue slots left empty
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, unsigned long long* rt, int iters, const _Float16* rnd) {
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    f16x8 x[4], y[4];
    for (int u = 0; u < 4; ++u)
        for (int j = 0; j < 8; ++j) {
            x[u][j] = MODE ? rnd[(threadIdx.x * 32 + u * 8 + j) % 8192] : (_Float16)0.f;
            y[u][j] = MODE ? rnd[(threadIdx.x * 32 + u * 8 + j + 4099) % 8192] : (_Float16)0.f;
        }
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                if ((MODE == 2 && (a & 3) == 3) || (MODE == 3 && (a & 1))) { asm volatile("s_nop 7"); continue; }     // (= 32 cycles)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[u], y[(u + a) & 3], acc[a], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < 8; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int iters, const _Float16* rnd, float* out, unsigned long long* cyc, unsigned long long* rt) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, rt, iters, rnd);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc, hr;
        (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hr, rt, 8, hipMemcpyDeviceToHost);
        const double n_mfma = (MODE == 2 ? 24.0 : MODE == 3 ? 16.0 : 32.0) * iters;            // per wave = per SIMD
        const double tf = 2.0 * 32 * 32 * 16 * n_mfma * 4 * 256 / (ms * 1e-3) / 1e12;
        printf("%-44s %7.1f ms: %6.2f cycles/MFMA/SIMD, %7.1f TFLOP/s = %.3f of 2500, shader clock %.3f GHz\n", name, ms,
               (double)hc / n_mfma, tf, tf / 2500.0, (double)hc / (double)hr * 0.1);
    }
}

int main() {
    float* out; unsigned long long *cyc, *rt; _Float16* rnd;
    (void)hipMalloc(&out, sizeof(float) * 256 * 256);
    (void)hipMalloc(&cyc, 8 * 256); (void)hipMalloc(&rt, 8 * 256); (void)hipMalloc(&rnd, 2 * 8192);
    std::vector<_Float16> h(8192);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 8) - (1 << 23)) * (0.6f / (1 << 23))); }
    (void)hipMemcpy(rnd, h.data(), 2 * 8192, hipMemcpyHostToDevice);
    run<0>("zero operands, one wave per SIMD", 200000, rnd, out, cyc, rt);
    run<1>("random operands, one wave per SIMD", 200000, rnd, out, cyc, rt);
    run<2>("random operands, 3 of 4 issue slots used", 200000, rnd, out, cyc, rt);
    run<3>("random operands, 1 of 2 issue slots used", 200000, rnd, out, cyc, rt);
    return 0;
}
