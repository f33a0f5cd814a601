// Dependent-issue latency of the FP64 instructions on the serial chain of the BA's diagonal-block factorisation
// (csrc/ba_solver.hip:factor_tile64): one wave, N dependent instructions of a kind, shader-clock stamps around them.
//   hipcc --offload-arch=gfx950 -O3 -o f64_latency tools/micro/f64_latency.hip && ./f64_latency
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ unsigned long long g_out[16];
__device__ double g_sink[64];

#define REP16(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP256(...) REP16(REP16(__VA_ARGS__))

__global__ __launch_bounds__(64) void k_lat(double seed) {
    double a = seed + threadIdx.x * 1e-9, b = 1.0000001, c = 1e-12;
    unsigned long long t0, t1;
    // 1) dependent v_fma_f64
    t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) g_out[0] = t1 - t0;
    // 2) dependent v_mul_f64
    t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) g_out[1] = t1 - t0;
    // 3) dependent v_rsq_f64 (value stays near 1)
    double r = 1.0 + a * 1e-30;
    t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_rsq_f64 %0, %0" : "+v"(r));)
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) g_out[2] = t1 - t0;
    // 4) independent v_fma_f64 (4 chains): issue rate
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3;
    t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));)
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) g_out[3] = t1 - t0;
    // 5) v_readlane_b32 -> v_fma_f64 with the scalar pair as an operand -> (dependent) readlane again
    int lo = __double_as_longlong(a) & 0xffffffff, hi = __double_as_longlong(a) >> 32;
    double y = a;
    t0 = __builtin_readcyclecounter();
    REP256({
        int s0; int s1;
        asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %3, 3" : "=s"(s0), "=s"(s1) : "v"(lo), "v"(hi));
        double sv = __hiloint2double(s1, s0);
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(y) : "s"(sv), "v"(c));
        lo = __double_as_longlong(y) & 0xffffffff;
        hi = __double_as_longlong(y) >> 32;
    })
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) g_out[4] = t1 - t0;
    // 6) dependent f32 fma for comparison
    float f = (float)seed, fb = 1.0000001f, fc = 1e-12f;
    t0 = __builtin_readcyclecounter();
    REP256(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(fb), "v"(fc));)
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) g_out[5] = t1 - t0;
    // 7) LDS round trip: dependent ds_write_b64 / ds_read_b64 of one's own slot
    __shared__ double sh[64];
    double z = a;
    t0 = __builtin_readcyclecounter();
    REP16(REP16({
        sh[threadIdx.x] = z;
        __builtin_amdgcn_s_waitcnt(0);
        z = sh[threadIdx.x ^ 1] + 1.0;
    }))
    t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) g_out[6] = t1 - t0;
    g_sink[threadIdx.x] = a + r + x0 + x1 + x2 + x3 + y + f + z;
}

int main() {
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, 1.0);
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, 1.0);
    (void)hipDeviceSynchronize();
    unsigned long long o[16];
    (void)hipMemcpyFromSymbol(o, HIP_SYMBOL(g_out), sizeof(o));
    printf("cycles per instruction (shader clock, one wave, 256 in a row):\n");
    printf("  dependent v_fma_f64        %.1f\n", o[0] / 256.0);
    printf("  dependent v_mul_f64        %.1f\n", o[1] / 256.0);
    printf("  dependent v_rsq_f64        %.1f\n", o[2] / 256.0);
    printf("  independent v_fma_f64      %.1f (4 chains)\n", o[3] / 1024.0);
    printf("  readlane x2 + v_fma_f64    %.1f per round\n", o[4] / 256.0);
    printf("  dependent v_fma_f32        %.1f\n", o[5] / 256.0);
    printf("  LDS write + read + add     %.1f per round\n", o[6] / 256.0);
    return 0;
}
