// VARIANT of regres_layer.hip: the weight ring is refilled with LDS-direct loads (global_load_lds_dwordx4, 4 slots, one vmcnt wait
// + barrier at stage 13 of every chunk) instead of global -> registers -> ds_write.  Measured: 0.812 of the pipe against 0.855.
// Prototype / micro-benchmark for a different MLP tile structure (round-2 candidate):
// activations stay in REGISTERS across layers, weights go through an LDS ring shared by the 4 waves of a workgroup.
//
//   * workgroup = 256 threads = 4 waves = one wave per SIMD, 64 points per workgroup, 16 points per wave;
//   * v_mfma_f32_16x16x4_f32, D = [unit][point]: lane (n = lane % 16, g = lane / 16) holds units 16 Mt + 4 g + r (r = 0..3) of
//     point n in accumulator quad Mt -- which is exactly the B operand (k = g) of the NEXT layer's k-quad {16 T + 4 k + j}
//     with T = Mt, j = r.  The accumulators of layer l therefore ARE the operands of layer l + 1 (after bias + ReLU in place):
//     no activation ever goes through LDS, and no barrier is needed for activations;
//   * the A operand (weights) is the same for all 4 waves: streamed global -> LDS in 32 KiB chunks (one input tile T = 16
//     units x all 512 outputs), 3-slot ring, one barrier per chunk (4096 MFMA cycles); every lane fetches the A values of 4
//     M-tiles with one ds_read_b128.
// The kernel computes L uniform 512 x 512 layers y = relu(W h + b) per 64-point tile and is checked against a host reference;
// the number printed is the fraction of the f32 MFMA peak it sustains.
// build: hipcc --offload-arch=gfx950 -O3 -o build/exp/regres_layer tools/micro/regres_layer.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int HID = 512, NT = HID / 16, CHUNK_F4 = 4 * 8 * 64;   // f4 per chunk (32 KiB)
constexpr int SLOTS = 4;
#ifndef VAR
#define VAR 0      // timing experiments (results wrong): 1 = no barrier, 2 = no global->LDS refill, 4 = no LDS operand reads
#endif

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k_layers(const f4* __restrict__ Wp, const float* __restrict__ bias, const float* __restrict__ x, float* __restrict__ y, int n_layers,
         int n_tiles, unsigned long long* __restrict__ stamps) {
    extern __shared__ f4 ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int chunks_per_pass = n_layers * NT;
    typedef const __attribute__((address_space(1))) f4* gp;
    typedef const __attribute__((address_space(1))) void* gp1;
    typedef __attribute__((address_space(3))) void* lp;
    gp W = (gp)Wp;
    int my_tiles = 0;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) ++my_tiles;
    if (my_tiles == 0) return;
    // prologue: chunks 0 and 1 of the stream into slots 0 and 1
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 8; ++i) ring[c * CHUNK_F4 + tid + 256 * i] = W[(size_t)(c % chunks_per_pass) * CHUNK_F4 + tid + 256 * i];
    __syncthreads();
    int slot = 0;          // slot of the chunk being consumed
    int next_chunk = 3 % chunks_per_pass;   // chunk index (within a pass over the layers) to load next
    // software pipeline of the A operand: stage = 2 ds_read_b128 (8 M-tiles of one k-quad) feeding 8 MFMAs; reads run PFD
    // stages ahead of their use, across chunk and layer boundaries (the next slot is complete one barrier earlier)
    constexpr int PFD = 2, NBUF = 4;
    f4 a[NBUF][2];
#pragma unroll
    for (int s0 = 0; s0 < PFD; ++s0) {
        a[s0][0] = ring[lane + (2 * s0) * 64];
        a[s0][1] = ring[lane + (2 * s0 + 1) * 64];
    }
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int p = 64 * t + 16 * wave + n;
        // h holds PRE-activations of the previous layer; its bias and ReLU are applied when a value is used as B operand
        // (2 VALU operations per 8 MFMAs, in the shadow of the matrix pipe).  The tile input counts as layer -1 with zero bias.
        f4 h[NT], acc[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) h[T] = *reinterpret_cast<const f4*>(x + (size_t)p * HID + 16 * T + 4 * g);
        for (int l = 0; l < n_layers; ++l) {
            const float* bprev = bias + (size_t)l * HID + 4 * g;          // bias[] has a leading all-zero layer
            f4 bp = *reinterpret_cast<const f4*>(bprev);
#pragma unroll
            for (int T = 0; T < NT; ++T) {
                const f4 bp_next = *reinterpret_cast<const f4*>(bprev + 16 * ((T + 1) % NT));
                const int nslot = slot + 1 == SLOTS ? 0 : slot + 1;
                const int wslot = (slot + 3) % SLOTS;     // chunk + 3 goes to the slot consumed in the previous chunk
                const f4* A = ring + slot * CHUNK_F4 + lane;
                const f4* An = ring + nslot * CHUNK_F4 + lane;
#if (VAR & 32)
                gp Wn = W + tid;                                     // timing experiment: always the same 32 KiB (cache-resident)
#else
                gp Wn = W + (size_t)next_chunk * CHUNK_F4 + tid;
#endif
                next_chunk = next_chunk + 1 == chunks_per_pass ? 0 : next_chunk + 1;
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    const int S = T * 16 + st;               // stage index within the layer (512 per layer, 512 % NBUF == 0)
                    const int sp = st + PFD;
                    const f4* src = sp < 16 ? A + (2 * sp) * 64 : An + (2 * (sp - 16)) * 64;
#if !(VAR & 4)
                    a[(S + PFD) % NBUF][0] = src[0];
                    a[(S + PFD) % NBUF][1] = src[64];
#endif
#if !(VAR & 2)
                    if (st >= 8) {      // chunk + 3: global -> LDS directly (no registers, no ds_write)
                        __builtin_amdgcn_global_load_lds((gp1)(Wn + 256 * (st - 8)),
                                                         (lp)(ring + wslot * CHUNK_F4 + wave * 64 + 256 * (st - 8)), 16, 0, 0);
                    }
                    if (st == 13) {
                        // the loads this wave issued during the previous chunk have landed (all but this chunk's five), then
                        // everybody's have: the next slot is complete before the first prefetch read of it (stage 14)
                        __builtin_amdgcn_s_waitcnt(0xF76);
                        __builtin_amdgcn_s_barrier();
                    }
#endif
                    const float b = fmaxf(h[T][st >> 2] + bp[st >> 2], 0.f);
                    __builtin_amdgcn_sched_barrier(0);
                    const int q0 = 2 * (st & 3);
                    const f4 a0 = a[S % NBUF][0], a1 = a[S % NBUF][1];
                    const bool first = (T == 0 && st < 4);   // first k-quad of the layer: C = 0
                    acc[4 * q0 + 0] = mfma16(a0.x, b, first ? zero4 : acc[4 * q0 + 0]);
                    acc[4 * q0 + 1] = mfma16(a0.y, b, first ? zero4 : acc[4 * q0 + 1]);
                    acc[4 * q0 + 2] = mfma16(a0.z, b, first ? zero4 : acc[4 * q0 + 2]);
                    acc[4 * q0 + 3] = mfma16(a0.w, b, first ? zero4 : acc[4 * q0 + 3]);
                    acc[4 * q0 + 4] = mfma16(a1.x, b, first ? zero4 : acc[4 * q0 + 4]);
                    acc[4 * q0 + 5] = mfma16(a1.y, b, first ? zero4 : acc[4 * q0 + 5]);
                    acc[4 * q0 + 6] = mfma16(a1.z, b, first ? zero4 : acc[4 * q0 + 6]);
                    acc[4 * q0 + 7] = mfma16(a1.w, b, first ? zero4 : acc[4 * q0 + 7]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                slot = nslot;
                bp = bp_next;
#if 0
                // every LDS write of this chunk was issued at least one stage ago and LDS operations of a wave complete in
                // order: the reads waited for since then prove the writes are done -- a bare barrier is enough
#if !(VAR & 1)
                __builtin_amdgcn_s_barrier();
#endif
#endif
            }
#pragma unroll
            for (int T = 0; T < NT; ++T) h[T] = acc[T];
        }
        const float* blast = bias + (size_t)n_layers * HID + 4 * g;
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const f4 bl = *reinterpret_cast<const f4*>(blast + 16 * T);
            f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = fmaxf(h[T][r] + bl[r], 0.f);
            *reinterpret_cast<f4*>(y + (size_t)p * HID + 16 * T + 4 * g) = o;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        stamps[0] = __builtin_readcyclecounter() - c0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 8;
    const int tiles_per_cu = argc > 2 ? atoi(argv[2]) : 8;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const int n_tiles = ncu * tiles_per_cu, n_pts = 64 * n_tiles;
    std::vector<float> W((size_t)L * HID * HID), B((size_t)L * HID), X((size_t)n_pts * HID);
    srand(1);
    for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.12f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    for (auto& v : X) v = rand() / (float)RAND_MAX;
    // pack: (((l*32 + T)*4 + j)*8 + q)*64 + lane)*4 + e = W[l][16*(4q+e) + lane%16][16T + 4*(lane/16) + j]
    std::vector<float> P((size_t)L * NT * CHUNK_F4 * 4);
    for (int l = 0; l < L; ++l)
        for (int T = 0; T < NT; ++T)
            for (int j = 0; j < 4; ++j)
                for (int q = 0; q < 8; ++q)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e)
                            P[((((((size_t)l * NT + T) * 4 + j) * 8 + q) * 64 + lane) * 4) + e] =
                                W[((size_t)l * HID + 16 * (4 * q + e) + lane % 16) * HID + 16 * T + 4 * (lane / 16) + j];
    float *dP, *dB, *dX, *dY;
    std::vector<float> Bz((size_t)(L + 1) * HID, 0.f);
    for (size_t i = 0; i < B.size(); ++i) Bz[HID + i] = B[i];
    hipMalloc(&dP, P.size() * 4 + 65536); hipMalloc(&dB, Bz.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, X.size() * 4);
    hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bz.data(), Bz.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    unsigned long long* dS;
    hipMalloc(&dS, 16);
    const size_t lds = (size_t)SLOTS * CHUNK_F4 * sizeof(f4);
    hipFuncSetAttribute((const void*)k_layers, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_layers, dim3(ncu), dim3(256), lds, 0, (const f4*)dP, dB, dX, dY, L, n_tiles, dS);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_layers, dim3(ncu), dim3(256), lds, 0, (const f4*)dP, dB, dX, dY, L, n_tiles, dS);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * HID * HID * (double)n_pts * L;
    unsigned long long st[2];
    hipMemcpy(st, dS, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)st[0] / ((double)st[1] * 10.0);     // s_memrealtime counts at 100 MHz
    const double cyc_per_chunk = (double)st[0] / ((double)tiles_per_cu * L * NT);
    printf("%d CUs, %d layers, %d tiles/CU: %.3f ms, %.1f TFLOP/s = %.3f of 157.3; shader clock %.3f GHz, %.0f cycles per chunk (4096 = MFMA pipe time) -> %.3f of the pipe\n",
           ncu, L, tiles_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, ghz, cyc_per_chunk, 4096.0 / cyc_per_chunk);
    // check a few points against the host
    std::vector<float> Y((size_t)n_pts * HID);
    hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int pi = 0; pi < 6; ++pi) {
        const int p = (pi * 7919 + 13) % n_pts;
        std::vector<double> h(X.begin() + (size_t)p * HID, X.begin() + (size_t)(p + 1) * HID), o(HID);
        for (int l = 0; l < L; ++l) {
            for (int u = 0; u < HID; ++u) {
                double s = B[(size_t)l * HID + u];
                for (int k = 0; k < HID; ++k) s += (double)W[((size_t)l * HID + u) * HID + k] * h[k];
                o[u] = s > 0 ? s : 0;
            }
            h = o;
        }
        double num = 0, den = 0;
        for (int u = 0; u < HID; ++u) { num = fmax(num, fabs(h[u] - Y[(size_t)p * HID + u])); den = fmax(den, fabs(h[u])); }
        worst = fmax(worst, num / fmax(den, 1e-30));
    }
    printf("max relative error vs host (6 points): %.2e  %s\n", worst, worst < 1e-4 ? "OK" : "MISMATCH");
    return 0;
}
