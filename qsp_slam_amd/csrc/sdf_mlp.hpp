// sdf_mlp.hpp -- the DeepSDF 8x512 decoder as ONE on-chip pass per 64-point tile (gfx950 / CDNA4).
//
// Replaces, for one tile of 64 query points that share a latent code:
//   forward   deep_sdf/deep_sdf_decoder.py:75-110   (9 Linear layers, ReLU, latent skip at layer 4, tanh)
//   backward  reconstruct/loss_utils.py:82-103      (d sdf / d [code | xyz], backward-DATA only; the reference's
//                                                    autograd also forms weight gradients nobody reads)
//
// Design (not a translation of the reference's cuBLAS-per-layer sequence):
//   * a 512-thread workgroup (8 waves, 2 per SIMD) owns a tile of 64 points for the whole network: activations live
//     in LDS as f32 [64][516] (row padded by one 16-B access so that ds_read_b128 of 16 distinct rows is
//     conflict-free), accumulators live in registers, nothing but weights is read from memory between the input
//     and the 71-wide Jacobian row;
//   * wave w owns output columns [64w, 64w+64) of every layer: 2x2 tiles of v_mfma_f32_32x32x2_f32 (exact f32, the
//     only MFMA that meets the 1e-4 parity bar without error compensation).  Its B operand (weights) is private to
//     the wave, so it is streamed global->VGPR with a software prefetch ring and never staged in LDS; the host
//     pre-packs the weights so that every wave-load is one contiguous 1 KiB line group (dwordx4 per lane);
//   * the ReLU masks of all 8 hidden layers stay in 16 VGPRs per lane: forward and backward use the same
//     (wave -> columns, lane -> row/col) map, so the backward pass applies them without any memory traffic;
//   * layer 3 has 445 outputs and layer 4 consumes [h3 | code | xyz] = 512 inputs exactly: the owners of columns
//     445..511 write the network input there instead of a ReLU output, which makes every hidden layer a uniform
//     512x512 GEMM.  In the backward pass the same columns are the skip-connection gradient and go to a stash.
//
// K order inside a k-group of 8 is permuted (lane half h takes k = 8g + 4h + e) identically for A and B; this only
// changes the f32 summation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qsp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef QSP_EXP_VARIANT
#define QSP_EXP_VARIANT 0   // timing experiments only (tools/exp_variants.sh): bit 0 = no weight loads in the k-loop,
#endif                      // bit 1 = no LDS operand reads in the k-loop (bits 0-3 compute garbage); bit 4 (16) = correct
                            // results + shader-clock stamps at every phase boundary of a tile (tools/phase_times.py)
#if (QSP_EXP_VARIANT & 16)
// wave 0 of workgroup (0,0) stamps the shader clock (and the constant 100 MHz counter) at every phase boundary of a tile
__device__ unsigned long long qsp_dbg_ts[96];
__device__ unsigned long long qsp_dbg_rt[96];
__device__ int qsp_dbg_n;
#define QSP_TS()                                                                        \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && ts_n < 96) {          \
        qsp_dbg_rt[ts_n] = __builtin_amdgcn_s_memrealtime();                            \
        qsp_dbg_ts[ts_n++] = __builtin_readcyclecounter();                              \
        qsp_dbg_n = ts_n;                                                               \
    }
// kernel-level stamps of k_mlp_jtj's item loop go to slots 80..95 (first item of workgroup 0)
#define QSP_TSK(i)                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0 && tsk_first) {                             \
        qsp_dbg_rt[80 + (i)] = __builtin_amdgcn_s_memrealtime();                        \
        qsp_dbg_ts[80 + (i)] = __builtin_readcyclecounter();                            \
    }
#else
#define QSP_TS()
#define QSP_TSK(i)
#endif

constexpr int TILE_P = 64;      // points per tile
constexpr int HID = 512;        // hidden width
constexpr int CODE_LEN = 64;    // latent code length
constexpr int NIN = CODE_LEN + 3;
constexpr int SKIP_COL = HID - NIN;   // 445: first pass-through column of the latent_in layer's input
constexpr int K4 = SKIP_COL + 3;      // forward K of the latent_in layer: [h3 (445) | xyz (3)]; its 64 code columns are folded
constexpr int KG4 = K4 / 8;           // into the per-hypothesis bias c4 (mlp_prepare / k_c0), like layer 0's
constexpr int K0_PAD = 96;      // layer-0 K (67) padded to a multiple of 8*PF
constexpr int LDA = HID + 4;    // activation row stride (floats)
constexpr int LDST = 68;        // stash row stride
constexpr int LDJ = 96;         // augmented-Jacobian row stride (72 used)
constexpr int NJ = 72;          // 7 pose + 64 code + 1 residual column
constexpr int MLP_THREADS = 512;
constexpr int HT_TILES = 6;     // upper-triangular 32x32 tiles of the 96x96 padded J~^T J~

struct MlpParams {
    const float4* wf[8];    // forward-packed weights of layers 1..7 (wf[0] unused: layer 0 is evaluated directly)
    const float4* wb[8];    // backward-packed weights of layers 0..7
    const float* bias[8];   // [512] zero-padded
    const float* w8;        // [512] last layer row
    float b8;
    const float* w0c;       // [64][512] layer-0 weights of the latent code, [code entry][unit]: a wave's loads are one line (code_bias)
    const float* w4c;       // [64][512] layer-4 weights of the latent code (input columns 445..508), same layout
    const float4* w0x;      // [128 unit quads][3] layer-0 weights of x, y, z for four consecutive units
    const float4* wf3[8];   // split-bf16 forward weights of layers 1..7: [col block 16][slab K/16][plane hi|mid|lo][lane 64][8 bf16]
    const float4* wb3[8];   // split-bf16 backward weights of layers 0..7 (column blocks over the layer's inputs, slabs over outputs)
    const float4* wfh[8];   // split-fp16 forward weights: [col block 16][slab K/16][plane hi|lo'][lane 64][8 fp16]; layer 0: its xyz columns, one slab
    const float4* wbh[8];   // split-fp16 backward weights of layers 0..7 (column blocks over the layer's inputs, slabs over outputs)
    const float4* wbh4s;    // split-fp16 backward weights of layer 4's skip columns (inputs 445..511 = [code | xyz]), packed like wbh[0]
    int* range_flag;        // set by the split-fp16 kernels when a value they had to split was outside fp16's range
    int use_tanh;           // NetworkSpecs.use_tanh: tanh on the output layer in front of the final tanh (deep_sdf_decoder.py:92-94)
    // Narrow decoders (a member of the family embedded into the 8 x 512 shape, e.g. 4 x 256 / code 32): what the NARROW form of the
    // split-fp16 tile may skip, exactly -- identity slots, all-zero k-slabs, all-zero column blocks (pack_weights fills these from
    // the embedded matrices; `narrow` = the decoder is small enough for the skipping to pay).
    int narrow;
    uint8_t skip[8];        // slot l is an identity layer of the embedding
    uint8_t ks_in[8];       // slabs of 16 input columns that carry weights (slot 4: of its [h3] part), even, >= 4
    uint8_t ks_out[8];      // slabs of 16 outputs that exist (the contraction length of the slot's backward product), even, >= 4
    uint8_t ncb_in[8];      // column blocks of 32 inputs that exist (the width of the slot's backward product)
    uint8_t ncb_out[8];     // column blocks of 32 outputs that exist
};

// The network's output and the seed of the backward pass from the last layer's pre-activation t (deep_sdf_decoder.py:92-94,
// 107-108): y = tanh(t), or tanh(tanh(t)) with use_tanh; dy = d y / d t in autograd's order (the outer tanh's factor first).
__device__ __forceinline__ float mlp_output(float t, int use_tanh, float& dy) {
    float y = tanhf(t);
    dy = 1.f - y * y;
    if (use_tanh) {
        const float y2 = tanhf(y);
        dy = (1.f - y2 * y2) * dy;
        y = y2;
    }
    return y;
}

// LDS carve (bytes): act 132096 | stash 17408 | inp 64*4*4 | code 256 | y 256 | red 2048 | row scale/res 512
struct __attribute__((aligned(16))) MlpSmem {
    float act[TILE_P * LDA];
    float stash[TILE_P * LDST];
    float xin[TILE_P * 4];      // object-frame xyz per row (4th = 0)
    float code[CODE_LEN];
    float c0[HID];              // layer-0 pre-activation without the xyz part: b0 + W0[:, :64] code (mlp_prepare)
    float c4[HID];              // layer-4 bias with the code part of the skip connection: b4 + W4[:, 445:509] code
    float y[TILE_P];            // tanh output
    float red[8 * TILE_P];      // layer-8 partial sums
    float rscale[TILE_P];       // row scale of the Jacobian (1 for SDF rows, de/ds for render rows, 0 for padding)
    float rres[TILE_P];         // residual supplied by the caller (render rows)
    float w8[HID];              // last layer's weight row (read by the layer-8 dot product and the backward seed)
    float dy[TILE_P];           // d y / d (last layer's pre-activation): the backward seed's row factor (mlp_output)
};

// The same instruction with the accumulator tile in AccVGPRs.  A kernel that fits 256 registers gets ArchVGPR accumulators from
// the compiler (it has no reason to use the second file); the matrix pipe then reads and writes C/D through the same register
// file that the weight and activation loads are landing in.  With the accumulators in the Acc file `k_mlp_jtj` runs 2.8 %
// faster (same box: 0.879 -> 0.904 of peak) although the compiler then splits the 256-register budget 128 / 128; the forward
// kernel, which needs ~166 ArchVGPRs next to its accumulators, loses 0.7 % to the extra copies and keeps the builtin.
// Written as inline asm because only an operand constraint can name the register file; the compiler therefore does not see an
// MFMA here and inserts none of the software wait states of the MFMA hazards -- mfma_acc_settle() provides them where a result
// is read by a non-matrix instruction (dependent MFMAs on the same tile are interlocked by the hardware).
template <bool AG>
__device__ __forceinline__ f32x16 mfma32t(float a, float b, f32x16 c) {
    if (AG) {
        asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        return c;
    }
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// 24 wait states: more than the 18-19 a 16-pass MFMA result needs before a VALU / accvgpr read (CDNA3 ISA, MFMA dependency
// table).  The accumulators are in/out operands of the asm so that every later read of them is ordered behind the wait states
// (a bare `asm volatile("s_nop")` does not stop the compiler from hoisting a v_accvgpr_read above it -- it did).
template <bool AG>
__device__ __forceinline__ void mfma_acc_settle(f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3) {
    if (AG) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
}
template <bool AG>
__device__ __forceinline__ void mfma_acc_settle(f32x16& c0) {
    if (AG) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(c0));
}

// D-row of accumulator register i of a 32x32 tile for this lane (C/D map of v_mfma_f32_32x32x2_f32); the D-column is
// lane & 31.  In the MLP GEMMs D-rows are units and D-columns points; in the J~^T J~ tile both are Jacobian columns.
__device__ __forceinline__ int acc_row(int i, int lane) { return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); }

// acc[r][c] += act[32r.., 0..8*KG) * Wpacked, for this wave's two column blocks.
// w0/w1 already include the lane offset; consecutive k-groups are 64 float4 apart.
// Packed weights are read through explicit global-address-space pointers: the bases come out of a table in memory, so
// without the cast the compiler emits flat_load (which also ties up lgkmcnt and forces vmcnt(0) waits).
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gptr4;

__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// The WEIGHT fragment is the MFMA A operand and the ACTIVATION fragment the B operand: D[i = unit][j = point].  A lane
// then holds, for ONE point (column j = lane & 31), units (reg&3) + 8(reg>>2) + 4(lane>>5) of the column block, i.e. four
// consecutive units per register quad -> the write-out is a 16-byte ds_write_b128 per quad into the [point][unit] image
// (with D = [point][unit] it would be sixteen 4-byte stores per tile).
#define QSP_MFMA_STEP_2x2(a0, a1, b0, b1)                   \
    acc[0][0] = mfma32t<AG>(b0.x, a0.x, acc[0][0]);         \
    acc[0][1] = mfma32t<AG>(b1.x, a0.x, acc[0][1]);              \
    acc[1][0] = mfma32t<AG>(b0.x, a1.x, acc[1][0]);              \
    acc[1][1] = mfma32t<AG>(b1.x, a1.x, acc[1][1]);              \
    acc[0][0] = mfma32t<AG>(b0.y, a0.y, acc[0][0]);              \
    acc[0][1] = mfma32t<AG>(b1.y, a0.y, acc[0][1]);              \
    acc[1][0] = mfma32t<AG>(b0.y, a1.y, acc[1][0]);              \
    acc[1][1] = mfma32t<AG>(b1.y, a1.y, acc[1][1]);              \
    acc[0][0] = mfma32t<AG>(b0.z, a0.z, acc[0][0]);              \
    acc[0][1] = mfma32t<AG>(b1.z, a0.z, acc[0][1]);              \
    acc[1][0] = mfma32t<AG>(b0.z, a1.z, acc[1][0]);              \
    acc[1][1] = mfma32t<AG>(b1.z, a1.z, acc[1][1]);              \
    acc[0][0] = mfma32t<AG>(b0.w, a0.w, acc[0][0]);              \
    acc[0][1] = mfma32t<AG>(b1.w, a0.w, acc[0][1]);              \
    acc[1][0] = mfma32t<AG>(b0.w, a1.w, acc[1][0]);              \
    acc[1][1] = mfma32t<AG>(b1.w, a1.w, acc[1][1]);

// The weight ring: PF k-groups (2 x 1 KiB wave-loads each) in flight per wave, global -> VGPR.  It is owned by mlp_tile
// and runs ACROSS layers: the last PF steps of a GEMM already fetch the first PF k-groups of the NEXT GEMM, so a layer
// never starts by waiting out a full L2 round trip behind its barrier.
template <int PF>
struct WRing {
    f32x4 q0[PF], q1[PF];
};

template <int PF>
__device__ __forceinline__ void ring_prime(WRing<PF>& R, const float4* __restrict__ w0_, const float4* __restrict__ w1_, int lane) {
    gptr4 w0 = (gptr4)w0_;
    gptr4 w1 = (gptr4)w1_;
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        R.q0[d] = w0[d * 64 + lane];
        R.q1[d] = w1[d * 64 + lane];
    }
}

// acc[r][c] += act[32r.., 0..8*KG) * Wpacked for this wave's two column blocks.
//   * w0 / w1 (this GEMM) and n0 / n1 (the next GEMM's column blocks) are wave-uniform bases (SGPR); the per-lane part
//     of the address is the single VGPR `lane`.  On entry the ring holds k-groups 0..PF-1 of this GEMM; on exit it
//     holds k-groups 0..PF-1 of the next one.
//   * activations: one k-group ahead (LDS -> VGPR); that prefetch runs one k-group past the end, inside MlpSmem.
//   * sched_barrier pins "issue next loads, then 16 MFMAs": without it the scheduler sinks each load to just before
//     its use and the ring degenerates to a load-wait-use sequence.
//   * the forward layers' bias quads for the write-out are fetched here, right BEFORE the next GEMM's prefetch is
//     issued: vector-memory results return in order, so a bias load issued after that prefetch could only be waited
//     for with vmcnt(0), i.e. by draining the ring behind every layer's barrier (measured: ~6 us per layer).
struct BiasQuads {
    f32x4 v[2][4];   // [column block][register quad]
};

template <int KG, int PF, bool BIAS, bool AG>
__device__ __forceinline__ void gemm_2x2(const float* __restrict__ act, const float4* __restrict__ w0_,
                                         const float4* __restrict__ w1_, const float4* __restrict__ n0_,
                                         const float4* __restrict__ n1_, WRing<PF>& R, f32x16 (&acc)[2][2], int lane,
                                         const float* __restrict__ bias_wave, BiasQuads& bq) {
    static_assert(KG % PF == 0 && KG >= 2 * PF, "KG must be a multiple of the prefetch depth, at least twice it");
    gptr4 w0 = (gptr4)w0_;
    gptr4 w1 = (gptr4)w1_;
    gptr4 n0 = (gptr4)n0_;
    gptr4 n1 = (gptr4)n1_;
    const float* a_row0 = act + (lane & 31) * LDA + 4 * (lane >> 5);
    const float* a_row1 = a_row0 + 32 * LDA;
    f32x4 a0 = lds4(a_row0), a1 = lds4(a_row1);
#pragma nounroll
    for (int kg = 0; kg < KG - PF; kg += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const f32x4 b0 = R.q0[d], b1 = R.q1[d];
#if (QSP_EXP_VARIANT & 8)
            R.q0[d] = w0[((kg + d + PF) & 15) * 64 + lane];   // timing experiment: 16 KiB window per stream (L2-resident)
            R.q1[d] = w1[((kg + d + PF) & 15) * 64 + lane];
#elif (QSP_EXP_VARIANT & 4)
            R.q0[d] = w0[d * 64 + lane];           // timing experiment: same 1 KiB every time (L1-resident)
            R.q1[d] = w1[d * 64 + lane];
#elif !(QSP_EXP_VARIANT & 1)
            R.q0[d] = w0[(kg + d + PF) * 64 + lane];
            R.q1[d] = w1[(kg + d + PF) * 64 + lane];
#endif
#if !(QSP_EXP_VARIANT & 2)
            const f32x4 a0n = lds4(a_row0 + 8 * (kg + d + 1));
            const f32x4 a1n = lds4(a_row1 + 8 * (kg + d + 1));
#else
            const f32x4 a0n = a1, a1n = a0;
#endif
            __builtin_amdgcn_sched_barrier(0);
            QSP_MFMA_STEP_2x2(a0, a1, b0, b1)
            __builtin_amdgcn_sched_barrier(0);
            a0 = a0n;
            a1 = a1n;
        }
    }
    if (BIAS) {
        typedef const __attribute__((address_space(1))) f32x4* gq;
        gq bp = (gq)(bias_wave + 4 * (lane >> 5));      // units 64w + 32c + 8g + 4h .. +3
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) bq.v[c][g] = bp[(32 * c + 8 * g) / 4];
    }
    // last PF k-groups: the ring refills with the NEXT GEMM's first k-groups
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const f32x4 b0 = R.q0[d], b1 = R.q1[d];
#if !(QSP_EXP_VARIANT & 1)
        R.q0[d] = n0[d * 64 + lane];
        R.q1[d] = n1[d * 64 + lane];
#endif
#if !(QSP_EXP_VARIANT & 2)
        const f32x4 a0n = lds4(a_row0 + 8 * (KG - PF + d + 1));
        const f32x4 a1n = lds4(a_row1 + 8 * (KG - PF + d + 1));
#else
        const f32x4 a0n = a1, a1n = a0;
#endif
        __builtin_amdgcn_sched_barrier(0);
        QSP_MFMA_STEP_2x2(a0, a1, b0, b1)
        __builtin_amdgcn_sched_barrier(0);
        a0 = a0n;
        a1 = a1n;
    }
    mfma_acc_settle<AG>(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
}

// one 32x32 tile over K = 8*KG (the 67-column backward of layer 0; waves 0..5).  Uses ring half q0, primed by the
// preceding GEMM; nothing follows it inside a tile, so it does not refill.
template <int KG, int PF, bool AG>
__device__ __forceinline__ void gemm_1x1(const float* __restrict__ act_rows, const float4* __restrict__ w0_, WRing<PF>& R,
                                         f32x16& acc, int lane) {
    gptr4 w0 = (gptr4)w0_;
    const float* a_row0 = act_rows + (lane & 31) * LDA + 4 * (lane >> 5);
    f32x4 a0 = lds4(a_row0);
#pragma nounroll
    for (int kg = 0; kg < KG; kg += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const f32x4 b0 = R.q0[d];
            if (kg + PF < KG) R.q0[d] = w0[(kg + d + PF) * 64 + lane];
            const f32x4 a0n = lds4(a_row0 + 8 * (kg + d + 1));
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32t<AG>(b0.x, a0.x, acc);
            acc = mfma32t<AG>(b0.y, a0.y, acc);
            acc = mfma32t<AG>(b0.z, a0.z, acc);
            acc = mfma32t<AG>(b0.w, a0.w, acc);
            __builtin_amdgcn_sched_barrier(0);
            a0 = a0n;
        }
    }
    mfma_acc_settle<AG>(acc);
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;
}

// v if bit k of mask is set, else +0: sign-extended 1-bit field extract + AND (no lane-mask SGPRs, no branches)
__device__ __forceinline__ float mask_sel(float v, uint32_t mask, int k) {
    const int sel = ((int)(mask << (31 - k))) >> 31;
    return __int_as_float(__float_as_int(v) & sel);
}

// Forward write-out of hidden layer L: bias, ReLU, mask capture.  Lane = one point per row tile, four register quads of
// four consecutive units each -> four 16-byte stores per 32x32 tile.
template <int L>
__device__ __forceinline__ void fwd_writeout(MlpSmem& s, const BiasQuads& bq, const f32x16 (&acc)[2][2], int wave,
                                             int lane, uint32_t& m_lo, uint32_t& m_hi) {
    const int h = lane >> 5;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int u0 = 64 * wave + 32 * c + 8 * g + 4 * h;            // first of 4 consecutive units
            const f32x4 bv = bq.v[c][g];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int p = 32 * r + (lane & 31);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * g + q;
                    float x = acc[r][c][i] + bv[q];
                    const bool pos = x > 0.f;
                    const uint32_t bit = pos ? 1u : 0u;
                    if (r == 0) lo |= bit << (c * 16 + i);
                    else hi |= bit << (c * 16 + i);
                    v[q] = pos ? x : 0.f;
                }
                *reinterpret_cast<f32x4*>(s.act + p * LDA + u0) = v;
            }
        }
    // opaque to the optimiser: otherwise it keeps the 64 v_cmp lane masks of every layer alive in SGPRs and spills them
    asm volatile("" : "+v"(lo), "+v"(hi));
    m_lo = lo;
    m_hi = hi;
}

// Layer 4 consumes [h3(445) | code(64) | xyz(3)].  The code columns are the same for every point of a hypothesis and live in
// the bias c4; the forward GEMM runs over K4 = 448 columns [h3 | xyz]: put xyz into columns 445..447 of every row (runs after
// a barrier behind fwd_writeout<3>, whose 16-byte stores cover those columns with zeros).
__device__ __forceinline__ void pass_through(MlpSmem& s) {
    if (threadIdx.x < TILE_P * 3) {
        const int row = threadIdx.x / 3, ci = threadIdx.x - row * 3;
        s.act[row * LDA + SKIP_COL + ci] = s.xin[row * 4 + ci];
    }
}

// Backward write-out of the gradient w.r.t. the INPUT of layer L (= post-ReLU output of layer L-1): apply the ReLU mask
// of layer L-1.  At L == 4 columns 445..511 are the skip-connection gradient w.r.t. the network input (no ReLU in front
// of them): stored raw here, moved to the stash by stash_extract() after a barrier.
template <int L>
__device__ __forceinline__ void bwd_writeout(MlpSmem& s, const f32x16 (&acc)[2][2], int wave, int lane, uint32_t m_lo,
                                             uint32_t m_hi) {
    const int h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int u0 = 64 * wave + 32 * c + 8 * g + 4 * h;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int p = 32 * r + (lane & 31);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * g + q;
                    float x = mask_sel(acc[r][c][i], r == 0 ? m_lo : m_hi, c * 16 + i);
                    if (L == 4) x = (u0 + q >= SKIP_COL) ? acc[r][c][i] : x;
                    v[q] = x;
                }
                *reinterpret_cast<f32x4*>(s.act + p * LDA + u0) = v;
            }
        }
}

// Backward counterpart of pass_through(): move d y / d [code | xyz] of the skip connection out of columns 445..511 and
// zero them (layer 3's backward GEMM must not see them; its packed weights are zero there anyway).
__device__ __forceinline__ void stash_extract(MlpSmem& s) {
    for (int e = threadIdx.x; e < TILE_P * NIN; e += MLP_THREADS) {
        const int row = e / NIN, ci = e - row * NIN;
        s.stash[row * LDST + ci] = s.act[row * LDA + SKIP_COL + ci];
        s.act[row * LDA + SKIP_COL + ci] = 0.f;
    }
}

// The code part of layers 0 and 4 for unit u: a = b0[u] + sum_k W0[u][k] code[k], a4 likewise with layer 4's skip columns, summed in
// ascending k.  The weights are stored [k][unit], so the 64 units of a wave read one line per k (stored [unit][k] every lane read
// its own line: 128 strided loads per thread, ~20 us in front of every decode workgroup and in k_sample's one-workgroup prologue).
__device__ __forceinline__ void code_bias(const MlpParams* __restrict__ P, int u, const float* code, float& a, float& a4) {
    const float* w = P->w0c + u;
    const float* w4 = P->w4c + u;
    a = P->bias[0][u], a4 = P->bias[4][u];
#pragma unroll 8
    for (int k = 0; k < CODE_LEN; ++k) {
        a += w[(size_t)k * HID] * code[k];
        a4 += w4[(size_t)k * HID] * code[k];
    }
}

// Once per workgroup, after s.code is written (all threads): the code part of layer 0 is the same for every point,
// c0[u] = b0[u] + sum_k W0[u][k] code[k].  The caller's next barrier (top of its tile loop) publishes c0.
__device__ __forceinline__ void mlp_prepare(MlpSmem& s, const MlpParams* __restrict__ Pm) {
    __syncthreads();
    const int u = threadIdx.x;
    float a, a4;
    code_bias(Pm, u, s.code, a, a4);
    s.c0[u] = a;
    s.c4[u] = a4;
}

// ---- split-bf16 GEMM primitives (see the note in front of mlp_tile_bf3) ---------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct Bf3 {
    bf16x8 hi, mid, lo;
};
__device__ __forceinline__ bf16x8 as_bf16x8_raw(f32x4 q) {
    union { f32x4 f; bf16x8 b; } u;
    u.f = q;
    return u.b;
}

// 8 consecutive f32 activations -> their three bf16 planes (the lane's B-operand fragments of one 32x32x16 MFMA)
#ifndef QSP_BF3_EXP
#define QSP_BF3_EXP 0     // timing experiments only: bit 0 = no operand split (garbage planes), bit 1 = no weight loads in the loop
#endif
__device__ __forceinline__ Bf3 split3(f32x4 a, f32x4 b) {
    Bf3 o;
#if (QSP_BF3_EXP & 1)
    o.hi = as_bf16x8_raw(a); o.mid = as_bf16x8_raw(b); o.lo = as_bf16x8_raw(a);
    return o;
#endif
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 h = (__bf16)v[j];
        const float r = v[j] - (float)h;
        const __bf16 m = (__bf16)r;
        const float r2 = r - (float)m;
        o.hi[j] = h;
        o.mid[j] = m;
        o.lo[j] = (__bf16)r2;
    }
    return o;
}

__device__ __forceinline__ bf16x8 as_bf16x8(f32x4 q) {
    union { f32x4 f; bf16x8 b; } u;
    u.f = q;
    return u.b;
}

#ifndef QSP_BF3_NT
#define QSP_BF3_NT 0      // experiment: non-temporal weight loads (the weights of a slab are used once per wave)
#endif
#if QSP_BF3_NT
#define QSP_WLD(p_) __builtin_nontemporal_load(&(p_))
#else
#define QSP_WLD(p_) (p_)
#endif
template <int PF>
struct WRing3 {
    f32x4 q[PF][2][3];      // [slab in flight][column block][plane]: 16-byte fragments (8 bf16 each)
};

template <int PF>
__device__ __forceinline__ void ring3_prime(WRing3<PF>& R, const float4* __restrict__ w0_, const float4* __restrict__ w1_, int lane) {
    gptr4 w0 = (gptr4)w0_;
    gptr4 w1 = (gptr4)w1_;
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            R.q[d][0][p] = w0[(d * 3 + p) * 64 + lane];
            R.q[d][1][p] = w1[(d * 3 + p) * 64 + lane];
        }
}

// v_mfma_f32_32x32x16_bf16 with the accumulator tile in AccVGPRs (AG; see mfma32t) or left to the compiler
template <bool AG>
__device__ __forceinline__ f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) {
    if (AG) {
        union { bf16x8 h; f32x4 f; } ua, ub;
        ua.h = a;
        ub.h = b;
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(ua.f), "v"(ub.f));
        return c;
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#define QSP_MFMA_BF(acc_, a_, b_) acc_ = mfma_bf<AG>(a_, b_, acc_)

// acc[r][c] += act[32r.., 0..16*KS) * W for this wave's two column blocks, six bf16 products per term.
// Ring protocol as gemm_2x2: on entry the ring holds slabs 0..PF-1 of this GEMM, on exit slabs 0..PF-1 of the next one.
template <int KS, int PF, bool AG = false, bool PIPE = false>
__device__ __forceinline__ void gemm_2x2_bf3(const float* __restrict__ act, const float4* __restrict__ w0_,
                                             const float4* __restrict__ w1_, const float4* __restrict__ n0_,
                                             const float4* __restrict__ n1_, WRing3<PF>& R, f32x16 (&acc)[2][2], int lane) {
    static_assert(KS % PF == 0 && KS >= 2 * PF, "slab count must be a multiple of the prefetch depth, at least twice it");
    gptr4 w0 = (gptr4)w0_;
    gptr4 w1 = (gptr4)w1_;
    gptr4 n0 = (gptr4)n0_;
    gptr4 n1 = (gptr4)n1_;
    // lane (r = lane & 31, h = lane >> 5) supplies act[point r][k = 16 s + 8 h + j], j = 0..7
    const float* a_row0 = act + (lane & 31) * LDA + 8 * (lane >> 5);
    const float* a_row1 = a_row0 + 32 * LDA;
    // PIPE: software pipeline -- the bf16 planes of slab s+1 are produced (VALU) while the 24 MFMAs of slab s run, the raw
    // f32 of slab s+2 is in flight from LDS meanwhile (24 more registers: the forward-only tile affords them, 298 instead of
    // 306 ms for C4's k_mlp_fwd; the forward+backward tile does not)
    Bf3 b0, b1;
    f32x4 x00, x01, x10, x11;
    if (PIPE) {
        b0 = split3(lds4(a_row0), lds4(a_row0 + 4));
        b1 = split3(lds4(a_row1), lds4(a_row1 + 4));
        x00 = lds4(a_row0 + 16); x01 = lds4(a_row0 + 20); x10 = lds4(a_row1 + 16); x11 = lds4(a_row1 + 20);
    } else {
        x00 = lds4(a_row0); x01 = lds4(a_row0 + 4); x10 = lds4(a_row1); x11 = lds4(a_row1 + 4);
    }
#pragma nounroll
    for (int ks = 0; ks < KS; ks += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            bf16x8 wa[2][3];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int p = 0; p < 3; ++p) wa[c][p] = as_bf16x8(R.q[d][c][p]);
            // refill this ring slot: slab ks + d + PF of this GEMM, or slab d of the next one
#if (QSP_BF3_EXP & 2)
            if (false) {
#else
            if (ks + PF < KS) {
#endif
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    R.q[d][0][p] = QSP_WLD(w0[((ks + d + PF) * 3 + p) * 64 + lane]);
                    R.q[d][1][p] = QSP_WLD(w1[((ks + d + PF) * 3 + p) * 64 + lane]);
                }
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    R.q[d][0][p] = QSP_WLD(n0[(d * 3 + p) * 64 + lane]);
                    R.q[d][1][p] = QSP_WLD(n1[(d * 3 + p) * 64 + lane]);
                }
            }
            if (!PIPE) {
                b0 = split3(x00, x01);
                b1 = split3(x10, x11);
                // next slab's activations (one slab past the end stays inside MlpSmem)
                x00 = lds4(a_row0 + 16 * (ks + d + 1));
                x01 = lds4(a_row0 + 16 * (ks + d + 1) + 4);
                x10 = lds4(a_row1 + 16 * (ks + d + 1));
                x11 = lds4(a_row1 + 16 * (ks + d + 1) + 4);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                // small terms first
                QSP_MFMA_BF(acc[0][c], wa[c][2], b0.hi);
                QSP_MFMA_BF(acc[1][c], wa[c][2], b1.hi);
                QSP_MFMA_BF(acc[0][c], wa[c][0], b0.lo);
                QSP_MFMA_BF(acc[1][c], wa[c][0], b1.lo);
                QSP_MFMA_BF(acc[0][c], wa[c][1], b0.mid);
                QSP_MFMA_BF(acc[1][c], wa[c][1], b1.mid);
                QSP_MFMA_BF(acc[0][c], wa[c][1], b0.hi);
                QSP_MFMA_BF(acc[1][c], wa[c][1], b1.hi);
                QSP_MFMA_BF(acc[0][c], wa[c][0], b0.mid);
                QSP_MFMA_BF(acc[1][c], wa[c][0], b1.mid);
                QSP_MFMA_BF(acc[0][c], wa[c][0], b0.hi);
                QSP_MFMA_BF(acc[1][c], wa[c][0], b1.hi);
            }
            if (PIPE) {   // (the slabs past the end of this GEMM stay inside MlpSmem; their fragments are never used)
                const Bf3 n0b = split3(x00, x01), n1b = split3(x10, x11);
                x00 = lds4(a_row0 + 16 * (ks + d + 2));
                x01 = lds4(a_row0 + 16 * (ks + d + 2) + 4);
                x10 = lds4(a_row1 + 16 * (ks + d + 2));
                x11 = lds4(a_row1 + 16 * (ks + d + 2) + 4);
                b0 = n0b;
                b1 = n1b;
            }
        }
    }
    mfma_acc_settle<AG>(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
}

// one 32x32 tile over 16*KS contraction columns (the 67-column backward of layer 0), no ring: PF slabs in flight
template <int KS, int PF, bool AG>
__device__ __forceinline__ void gemm_1x1_bf3(const float* __restrict__ act_rows, const float4* __restrict__ w0_, f32x16& acc, int lane) {
    gptr4 w0 = (gptr4)w0_;
    const float* a_row0 = act_rows + (lane & 31) * LDA + 8 * (lane >> 5);
    f32x4 q[PF][3];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int p = 0; p < 3; ++p) q[d][p] = w0[(d * 3 + p) * 64 + lane];
    f32x4 x0 = lds4(a_row0), x1 = lds4(a_row0 + 4);
#pragma nounroll
    for (int ks = 0; ks < KS; ks += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const bf16x8 whi = as_bf16x8(q[d][0]), wmid = as_bf16x8(q[d][1]), wlo = as_bf16x8(q[d][2]);
            if (ks + PF < KS) {
#pragma unroll
                for (int p = 0; p < 3; ++p) q[d][p] = w0[((ks + d + PF) * 3 + p) * 64 + lane];
            }
            const Bf3 b = split3(x0, x1);
            x0 = lds4(a_row0 + 16 * (ks + d + 1));
            x1 = lds4(a_row0 + 16 * (ks + d + 1) + 4);
            QSP_MFMA_BF(acc, wlo, b.hi);
            QSP_MFMA_BF(acc, whi, b.lo);
            QSP_MFMA_BF(acc, wmid, b.mid);
            QSP_MFMA_BF(acc, wmid, b.hi);
            QSP_MFMA_BF(acc, whi, b.mid);
            QSP_MFMA_BF(acc, whi, b.hi);
        }
    }
    mfma_acc_settle<AG>(acc);
}

// Whole network for the tile whose inputs are staged in s.code / s.xin.
// On return (all threads, after a barrier):
//   s.y[row]                         = sdf value
//   BWD: s.act viewed as G[row*LDST_G + c], c < 67 = d sdf / d [code | xyz] (skip gradient already added)
constexpr int LDG = 72;
template <bool BWD, int PF, bool AGT = false, bool B3 = false>
__device__ __forceinline__ void mlp_tile(MlpSmem& s, const MlpParams* __restrict__ Pm) {
    // B3: every GEMM of the tile on the split-bf16 pipe (gemm_2x2_bf3 / gemm_1x1_bf3: three bf16 terms per operand, six
    // products, f32 accumulation) instead of the f32 MFMA; everything else -- layer 0, write-outs, masks, layer 8, the seed --
    // is the same code.
    // AGT: AccVGPR accumulators (see mfma32t).  Only `k_mlp_jtj` asks for them.  Measured on one box, C4: forward-only tile with
    // them 0.908-0.910 of peak against 0.918 with the builtin; with the bias quads parked in AccVGPRs as well 0.910 and
    // `k_mlp_jtj` 0.895 instead of 0.900; accumulators pinned by an empty asm around the builtin MFMA 0.897 / 0.909; the
    // forward+backward tile inside `k_decode<true>` (no Jacobian rows, no normal equations, 230 registers) 1.3 % slower with them.
    const int tid = threadIdx.x;
    int ts_n = 0;
    (void)ts_n;
    QSP_TS()
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // The parameter table is read through an index the compiler cannot see through, once per tile: otherwise every
    // layer's base pointers and per-lane addresses are hoisted out of the caller's tile loop and spill.
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const MlpParams& P = Pm[oz];
    uint32_t mlo[8], mhi[8];
    f32x16 acc[2][2];

    // the last layer's row is read from LDS later so that no vector-memory wait has to drain the weight ring
    s.w8[tid] = P.w8[tid];
    if constexpr (B3) {      // and so are the biases of layers 1..7 on the split-bf16 pipe (s.stash is free until the backward
#pragma unroll           //  pass reaches layer 4; the f32 pipe fetches them ahead of its ring's cross-layer prefetch instead)
        for (int l = 1; l < 8; ++l) s.stash[(l - 1) * HID + tid] = P.bias[l][tid];
    }

    const int cb0 = 2 * wave;   // this wave's first column block
    constexpr int KGH = HID / 8;
    constexpr int KSH = HID / 16, KS4 = K4 / 16, PF3 = 2;     // split-bf16: slabs of 16 k, two slabs in flight (one: k_mlp_jtj 386 instead of 357 ms at C4)
    // column-block bases of this wave in every packed matrix
#define QSP_WF(L) (B3 ? P.wf3[L] + (size_t)(cb0 * KSH * 3) * 64 : P.wf[L] + (cb0 * KGH) * 64)
#define QSP_WF1(L) (B3 ? P.wf3[L] + (size_t)((cb0 + 1) * KSH * 3) * 64 : P.wf[L] + ((cb0 + 1) * KGH) * 64)
#define QSP_WF4 (B3 ? P.wf3[4] + (size_t)(cb0 * KS4 * 3) * 64 : P.wf[4] + (cb0 * KG4) * 64)
#define QSP_WF41 (B3 ? P.wf3[4] + (size_t)((cb0 + 1) * KS4 * 3) * 64 : P.wf[4] + ((cb0 + 1) * KG4) * 64)
#define QSP_WB(L) (B3 ? P.wb3[L] + (size_t)(cb0 * KSH * 3) * 64 : P.wb[L] + (cb0 * KGH) * 64)
#define QSP_WB1(L) (B3 ? P.wb3[L] + (size_t)((cb0 + 1) * KSH * 3) * 64 : P.wb[L] + ((cb0 + 1) * KGH) * 64)
    // one GEMM of the tile on whichever pipe: KG k-groups of 8 (f32) = KG / 2 slabs of 16 (split bf16)
#define QSP_GEMM(KG, BIAS, W0, W1, N0, N1, BIASPTR, LIDX)                                                        \
    if constexpr (B3) {                                                                                          \
        gemm_2x2_bf3<(KG) / 2, PF3, AGT>(s.act, W0, W1, N0, N1, ring3, acc, lane);                               \
        if (BIAS) {   /* biases staged in LDS (s.stash, free until the backward pass reaches layer 4): no vector-memory wait */ \
            const float* bp_ = s.stash + ((LIDX) - 1) * HID + 64 * wave + 4 * (lane >> 5);                       \
            _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_)   \
                bq.v[c_][g_] = lds4(bp_ + 32 * c_ + 8 * g_);                                                     \
        }                                                                                                        \
    } else {                                                                                                     \
        gemm_2x2<KG, PF, BIAS, AGT>(s.act, W0, W1, N0, N1, ring, acc, lane, BIASPTR, bq);                        \
    }
    // layer-0 backward: six 32x32 output tiles (2 point blocks x 3 column blocks of the 96 padded inputs) on four SIMDs:
    // waves 0..3 (one per SIMD) take tiles 0..3 over the whole K; tiles 4 and 5 are split in K between the two waves of
    // SIMDs 0/2 (waves 4, 6) and 1/3 (waves 5, 7) -- 1.5 tiles of MFMA work on every SIMD instead of 2/2/1/1
    const int l0_tile = wave < 4 ? wave : 4 + (wave & 1);
    const int l0_half = wave < 4 ? 0 : (wave >> 1) - 2;          // 0 = first (or whole) K range, 1 = second half
    const int r0 = l0_tile / 3, c0 = l0_tile % 3;
    const float4* wb0 = B3 ? P.wb3[0] + (size_t)((c0 * KSH + l0_half * (KSH / 2)) * 3) * 64
                           : P.wb[0] + (c0 * KGH + l0_half * (KGH / 2)) * 64;

    // ---- layer 0: a0 = relu(c0 + W0[:, 64:67] xyz), written in the MFMA write-out's lane/register pattern ------------
    // The 64 code columns of layer 0 are the same for every point of the workgroup: folded into c0 by mlp_prepare().
    static_assert(PF == 4, "ring depth");
    WRing<PF> ring;
    WRing3<PF3> ring3;
    if constexpr (B3) ring3_prime(ring3, QSP_WF(1), QSP_WF1(1), lane);
    else ring_prime(ring, QSP_WF(1), QSP_WF1(1), lane);
    BiasQuads bq;
    {
        const int h = lane >> 5;
        const f32x4 x0 = lds4(s.xin + 4 * (lane & 31)), x1 = lds4(s.xin + 4 * (32 + (lane & 31)));
        uint32_t lo = 0, hi = 0;
        typedef const __attribute__((address_space(1))) f32x4* gq;
        gq wx = (gq)P.w0x;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int u0 = 64 * wave + 32 * c + 8 * g + 4 * h;
                const f32x4 cq = lds4(s.c0 + u0);
                const f32x4 w0 = wx[3 * (u0 >> 2)], w1 = wx[3 * (u0 >> 2) + 1], w2 = wx[3 * (u0 >> 2) + 2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const f32x4 xp = r == 0 ? x0 : x1;
                    const int p = 32 * r + (lane & 31);
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 4 * g + q;
                        const float x = cq[q] + w0[q] * xp.x + w1[q] * xp.y + w2[q] * xp.z;
                        const bool pos = x > 0.f;
                        const uint32_t bit = pos ? 1u : 0u;
                        if (r == 0) lo |= bit << (c * 16 + i);
                        else hi |= bit << (c * 16 + i);
                        v[q] = pos ? x : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(s.act + p * LDA + u0) = v;
                }
            }
        asm volatile("" : "+v"(lo), "+v"(hi));
        mlo[0] = lo;
        mhi[0] = hi;
    }
    QSP_TS()
    __syncthreads();
    QSP_TS()

    // ---- layers 1..7 (K = 512) ---------------------------------------------------------------------------------
#define QSP_FWD_LAYER(L)                                                                                      \
    zero_acc(acc);                                                                                            \
    QSP_GEMM(KGH, true, QSP_WF(L), QSP_WF1(L), QSP_WF((L) + 1), QSP_WF1((L) + 1), P.bias[L] + 64 * wave, L)   \
    QSP_TS()                                                                                                  \
    __syncthreads();                                                                                          \
    QSP_TS()                                                                                                  \
    fwd_writeout<L>(s, bq, acc, wave, lane, mlo[L], mhi[L]);                                                  \
    QSP_TS()                                                                                                  \
    __syncthreads();                                                                                          \
    QSP_TS()
    QSP_FWD_LAYER(1)
    QSP_FWD_LAYER(2)
    zero_acc(acc);
    QSP_GEMM(KGH, true, QSP_WF(3), QSP_WF1(3), QSP_WF4, QSP_WF41, P.bias[3] + 64 * wave, 3)
    QSP_TS()
    __syncthreads();
    QSP_TS()
    fwd_writeout<3>(s, bq, acc, wave, lane, mlo[3], mhi[3]);
    QSP_TS()
    __syncthreads();
    QSP_TS()
    pass_through(s);
    QSP_TS()
    __syncthreads();
    QSP_TS()
    // layer 4: K = 448, bias = c4 of this hypothesis (LDS)
    zero_acc(acc);
    QSP_GEMM(KG4, false, QSP_WF4, QSP_WF41, QSP_WF(5), QSP_WF1(5), (const float*)nullptr, 4)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq.v[c][g] = lds4(s.c4 + 64 * wave + 32 * c + 8 * g + 4 * (lane >> 5));
    QSP_TS()
    __syncthreads();
    QSP_TS()
    fwd_writeout<4>(s, bq, acc, wave, lane, mlo[4], mhi[4]);
    QSP_TS()
    __syncthreads();
    QSP_TS()
    QSP_FWD_LAYER(5)
    QSP_FWD_LAYER(6)
    zero_acc(acc);
    if (BWD) { QSP_GEMM(KGH, true, QSP_WF(7), QSP_WF1(7), QSP_WB(7), QSP_WB1(7), P.bias[7] + 64 * wave, 7) }
    else { QSP_GEMM(KGH, true, QSP_WF(7), QSP_WF1(7), QSP_WF(1), QSP_WF1(1), P.bias[7] + 64 * wave, 7) }
    QSP_TS()
    __syncthreads();
    QSP_TS()
    fwd_writeout<7>(s, bq, acc, wave, lane, mlo[7], mhi[7]);
    QSP_TS()
    __syncthreads();
    QSP_TS()
#undef QSP_FWD_LAYER

    // ---- layer 8: 512 -> 1, tanh ---------------------------------------------------------------------------------
    {
        // wave = k segment of 64, lane = row
        const float* a = s.act + lane * LDA + 64 * wave;
        const float* w = s.w8 + 64 * wave;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 av = *reinterpret_cast<const float4*>(a + 4 * q);
            const float4 wv = *reinterpret_cast<const float4*>(w + 4 * q);
            part += av.x * wv.x;
            part += av.y * wv.y;
            part += av.z * wv.z;
            part += av.w * wv.w;
        }
        s.red[wave * TILE_P + lane] = part;
    }
    QSP_TS()
    __syncthreads();
    QSP_TS()
    if (tid < TILE_P) {
        float t = P.b8;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += s.red[q * TILE_P + tid];
        float dy_;
        s.y[tid] = mlp_output(t, P.use_tanh, dy_);
        s.dy[tid] = dy_;
    }
    QSP_TS()
    __syncthreads();
    QSP_TS()
    if (!BWD) return;

    // ---- backward seed: d y / d a7 = (1 - y^2) * w8[unit] * [a7 > 0] ---------------------------------------------
    {
        const int h = lane >> 5;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int p = 32 * r + (lane & 31);
            const float dy = s.dy[p];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int u0 = 64 * wave + 32 * c + 8 * g + 4 * h;
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(s.w8 + u0);
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = mask_sel(dy * wv[q], r == 0 ? mlo[7] : mhi[7], c * 16 + 4 * g + q);
                    *reinterpret_cast<f32x4*>(s.act + p * LDA + u0) = v;
                }
        }
    }
    QSP_TS()
    __syncthreads();
    QSP_TS()

    // ---- backward through layers 7..1: g_in = g_a . W_L, masked by layer L-1 ------------------------------------
#define QSP_BWD_LAYER(L)                                                                                      \
    zero_acc(acc);                                                                                            \
    QSP_GEMM(KGH, false, QSP_WB(L), QSP_WB1(L), (L) > 1 ? QSP_WB((L) - 1) : wb0, (L) > 1 ? QSP_WB1((L) - 1) : wb0, \
             (const float*)nullptr, L)                                                                        \
    QSP_TS()                                                                                                  \
    __syncthreads();                                                                                          \
    QSP_TS()                                                                                                  \
    bwd_writeout<L>(s, acc, wave, lane, mlo[L - 1], mhi[L - 1]);                                              \
    QSP_TS()                                                                                                  \
    __syncthreads();                                                                                          \
    QSP_TS()
    QSP_BWD_LAYER(7)
    QSP_BWD_LAYER(6)
    QSP_BWD_LAYER(5)
    QSP_BWD_LAYER(4)
    stash_extract(s);
    QSP_TS()
    __syncthreads();
    QSP_TS()
    // layer 3 has 445 outputs: its backward contraction runs over K4 = 448 gradient columns (445..447 were zeroed by
    // stash_extract, the packed rows 445..511 are zero)
    zero_acc(acc);
    QSP_GEMM(KG4, false, QSP_WB(3), QSP_WB1(3), QSP_WB(2), QSP_WB1(2), (const float*)nullptr, 3)
    QSP_TS()
    __syncthreads();
    QSP_TS()
    bwd_writeout<3>(s, acc, wave, lane, mlo[2], mhi[2]);
    QSP_TS()
    __syncthreads();
    QSP_TS()
    QSP_BWD_LAYER(2)
    QSP_BWD_LAYER(1)
#undef QSP_BWD_LAYER

    // ---- backward through layer 0: 67 (padded 96) input columns ----------------------------------------------------
    f32x16 g0;
#pragma unroll
    for (int i = 0; i < 16; ++i) g0[i] = 0.f;
    if constexpr (B3) {     // (the ring's prefetch of wb0 by the last GEMM is not used on this pipe: gemm_1x1_bf3 loads its own)
        if (wave < 4) gemm_1x1_bf3<KSH, PF3, AGT>(s.act + 32 * r0 * LDA, wb0, g0, lane);
        else gemm_1x1_bf3<KSH / 2, PF3, AGT>(s.act + 32 * r0 * LDA + l0_half * (HID / 2), wb0, g0, lane);
    } else {
        if (wave < 4) gemm_1x1<KGH, PF, AGT>(s.act + 32 * r0 * LDA, wb0, ring, g0, lane);
        else gemm_1x1<KGH / 2, PF, AGT>(s.act + 32 * r0 * LDA + l0_half * (HID / 2), wb0, ring, g0, lane);
    }
#undef QSP_WF
#undef QSP_WF1
#undef QSP_WF4
#undef QSP_WF41
#undef QSP_WB
#undef QSP_WB1
#undef QSP_GEMM
    QSP_TS()
    __syncthreads();
    QSP_TS()
    float* l0_scr = s.act + TILE_P * LDG;             // behind the G image; free now that every GEMM read is done
    if (wave >= 6) {
#pragma unroll
        for (int i = 0; i < 16; ++i) l0_scr[((wave - 6) * 16 + i) * 64 + lane] = g0[i];
    }
    __syncthreads();
    if (wave < 6) {
        if (wave >= 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) g0[i] += l0_scr[((wave - 4) * 16 + i) * 64 + lane];
        }
        // D[i = input column within block c0][j = point]: four consecutive input columns per register quad
        const int p = 32 * r0 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int k0 = 32 * c0 + 8 * g + 4 * (lane >> 5);
            if (k0 < NIN) {     // 64..67 is the last useful quad (67 itself is padding inside both row strides)
                const f32x4 st = *reinterpret_cast<const f32x4*>(s.stash + p * LDST + k0);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = g0[4 * g + q] + st[q];
                *reinterpret_cast<f32x4*>(s.act + p * LDG + k0) = v;
            }
        }
    }
    QSP_TS()
    __syncthreads();
    QSP_TS()
}

// ===================================================================================================================
// Split-bf16 forward tile (qsp_decoder_set_option(QSP_DEC_OPT_FORWARD_PRECISION, 1)).
//
// The f32 MFMA runs at 1/16 of the bf16 rate.  An f32 value is the EXACT sum of three bf16 values' worth of mantissa
// (8 + 8 + 8 bits): x = x_hi + x_mid + x_lo with x_hi = bf16(x), x_mid = bf16(x - x_hi), x_lo = bf16(x - x_hi - x_mid).
// A product of two such numbers needs six of the nine cross terms to keep every contribution above 2^-24 of it:
//     w x  ~=  w_hi x_hi + (w_hi x_mid + w_mid x_hi) + (w_hi x_lo + w_mid x_mid + w_lo x_hi)
// all accumulated in f32 by v_mfma_f32_32x32x16_bf16: 6 MFMAs of 32 cycles do the work of 8 f32 MFMAs of 64 cycles, 2.67 x
// the f32 pipe.  Measured against float64 on the fitted decoder (tools/studies/bf16_split_accuracy.py): 2.2e-7 relative on the
// SDF value, f32 itself 1.8e-7.
//   * weights: split once on the host, streamed as three 1 KiB fragments per (column block, slab of 16 k) -- 1.5 x the bytes
//     per multiply-add of the f32 path at 2.67 x its rate;
//   * activations stay f32 in LDS (6 bytes per value would not fit a 64-point tile) and are split by the consuming wave
//     after the LDS read: ~5.5 VALU operations per value, which ride in the issue slots the MFMAs leave free (an MFMA holds
//     the SIMD's vector issue for 8 of its 32 cycles);
//   * same wave -> (64 units x 64 points) map, same C/D layout as the f32 MFMA, so the write-out code is shared.
// Forward only: the discrete decisions of the render term (|sdf| < cut-off) see values that differ from the f32 tile's in
// the last bits, like any two float32 implementations do; the fwd+bwd tile that feeds the normal equations stays on the f32 pipe.
// ===================================================================================================================
// Forward network on the split-bf16 pipe for the tile staged in s.code / s.xin / s.c0 / s.c4; on return s.y[row] = sdf value.
// The biases of layers 1..7 are staged in s.stash (unused by forward-only kernels) so that no vector-memory wait has to drain
// the weight ring between layers.
#ifndef QSP_BF3_PF
#define QSP_BF3_PF 2
#endif
template <int PF>
__device__ __forceinline__ void mlp_tile_bf3(MlpSmem& s, const MlpParams* __restrict__ Pm) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const MlpParams& P = Pm[oz];
    f32x16 acc[2][2];
    uint32_t mlo, mhi;
    s.w8[tid] = P.w8[tid];
    float* bias_sh = s.stash;                              // [7][512] biases of layers 1..7 (17408 B >= 14336 B)
#pragma unroll
    for (int l = 1; l < 8; ++l) bias_sh[(l - 1) * HID + tid] = P.bias[l][tid];
    const int cb0 = 2 * wave;
    constexpr int KSH = HID / 16, KS4 = K4 / 16;
#define QSP_W3(L, KS_) (P.wf3[L] + (size_t)(cb0 * (KS_) * 3) * 64)
#define QSP_W31(L, KS_) (P.wf3[L] + (size_t)((cb0 + 1) * (KS_) * 3) * 64)
    WRing3<PF> ring;
    ring3_prime(ring, QSP_W3(1, KSH), QSP_W31(1, KSH), lane);
    BiasQuads bq;
    // ---- layer 0 (exact f32, as in mlp_tile) ---------------------------------------------------------------------
    {
        const int h = lane >> 5;
        const f32x4 x0 = lds4(s.xin + 4 * (lane & 31)), x1 = lds4(s.xin + 4 * (32 + (lane & 31)));
        typedef const __attribute__((address_space(1))) f32x4* gq;
        gq wx = (gq)P.w0x;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int u0 = 64 * wave + 32 * c + 8 * g + 4 * h;
                const f32x4 cq = lds4(s.c0 + u0);
                const f32x4 w0 = wx[3 * (u0 >> 2)], w1 = wx[3 * (u0 >> 2) + 1], w2 = wx[3 * (u0 >> 2) + 2];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const f32x4 xp = r == 0 ? x0 : x1;
                    const int p = 32 * r + (lane & 31);
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x = cq[q] + w0[q] * xp.x + w1[q] * xp.y + w2[q] * xp.z;
                        v[q] = x > 0.f ? x : 0.f;
                    }
                    *reinterpret_cast<f32x4*>(s.act + p * LDA + u0) = v;
                }
            }
    }
    __syncthreads();
#define QSP_BIAS_FROM(ptr)                                                                          \
    _Pragma("unroll") for (int c = 0; c < 2; ++c) _Pragma("unroll") for (int g = 0; g < 4; ++g)     \
        bq.v[c][g] = lds4((ptr) + 64 * wave + 32 * c + 8 * g + 4 * (lane >> 5));
#define QSP_FWD3(L, KS_, NL, NKS, BIASPTR)                                                                             \
    zero_acc(acc);                                                                                                     \
    gemm_2x2_bf3<KS_, PF, false, true>(s.act, QSP_W3(L, KS_), QSP_W31(L, KS_), QSP_W3(NL, NKS), QSP_W31(NL, NKS), ring, acc, lane); \
    QSP_BIAS_FROM(BIASPTR)                                                                                             \
    __syncthreads();                                                                                                   \
    fwd_writeout<L>(s, bq, acc, wave, lane, mlo, mhi);                                                                 \
    __syncthreads();
    QSP_FWD3(1, KSH, 2, KSH, bias_sh + 0 * HID)
    QSP_FWD3(2, KSH, 3, KSH, bias_sh + 1 * HID)
    QSP_FWD3(3, KSH, 4, KS4, bias_sh + 2 * HID)
    pass_through(s);
    __syncthreads();
    QSP_FWD3(4, KS4, 5, KSH, s.c4)
    QSP_FWD3(5, KSH, 6, KSH, bias_sh + 4 * HID)
    QSP_FWD3(6, KSH, 7, KSH, bias_sh + 5 * HID)
    QSP_FWD3(7, KSH, 1, KSH, bias_sh + 6 * HID)
#undef QSP_FWD3
#undef QSP_BIAS_FROM
#undef QSP_W3
#undef QSP_W31
    (void)mlo; (void)mhi;
    // ---- layer 8: 512 -> 1, tanh (f32) ---------------------------------------------------------------------------
    {
        const float* a = s.act + lane * LDA + 64 * wave;
        const float* w = s.w8 + 64 * wave;
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 av = *reinterpret_cast<const float4*>(a + 4 * q);
            const float4 wv = *reinterpret_cast<const float4*>(w + 4 * q);
            part += av.x * wv.x;
            part += av.y * wv.y;
            part += av.z * wv.z;
            part += av.w * wv.w;
        }
        s.red[wave * TILE_P + lane] = part;
    }
    __syncthreads();
    if (tid < TILE_P) {
        float t = P.b8;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += s.red[q * TILE_P + tid];
        float dy_;
        s.y[tid] = mlp_output(t, P.use_tanh, dy_);
        s.dy[tid] = dy_;
    }
    __syncthreads();
}


// ===================================================================================================================
// Split-fp16 tile (QSP_DEC_OPT_*_PRECISION = 2), four waves of 128 units x 64 points, forward and forward+backward.
//
// fp16 carries 11 significand bits, bf16 8: TWO fp16 terms hold 22 bits of an f32 value where the bf16 split needs three terms.
// With the second term pre-scaled so that it never leaves fp16's normal range,
//     x = x_hi + 2^-11 x_lo',   x_hi = fp16(x),   x_lo' = fp16((x - x_hi) 2^11)
// a product keeps everything above 2^-22 of it with THREE fp16 MFMAs instead of six bf16 ones:
//     w x  ~=  w_hi x_hi  +  2^-11 (w_hi x_lo' + w_lo' x_hi)
// (main and cross terms in separate f32 accumulators, combined once per layer in the write-out).  Measured against float64 on
// the fitted decoder (tools/studies/fp16_split_accuracy.py, 20 000 points): 2.5e-7 relative on the SDF value -- the f32 pipe
// 2.1e-7, the three-term bf16 split 2.5e-7.  Per multiply-add: half the matrix-pipe work of the bf16 split and 4 instead of 6
// bytes of weights.
//   * two fp16 planes are 4 bytes per value -- what the f32 activation image takes.  So the image in LDS IS the pair of planes:
//     the write-out of a layer splits every value once ([row][k-group of 8][hi x 8 | lo' x 8], the f32 image's row stride) and
//     the GEMM loop reads its B operands with two ds_read_b128 and no arithmetic at all.  (The bf16 split needs 6 bytes per
//     value and re-splits every activation in every consuming wave.)
//   * the two accumulator sets of a 2x2 tile would take 128 of an 8-wave kernel's 256 registers; this tile runs FOUR waves (one
//     per SIMD, 512 registers each): wave w owns units [128 w, 128 w + 128) of every layer as 4 x 2 MFMA tiles, 2 x 128
//     accumulator registers = the whole AccVGPR file;
//   * the bias is the accumulators' initial value; the skip connection's gradient goes from layer 4's backward write-out
//     straight to the stash in f32;
//   * |value| above fp16's range (65 504) cannot be split: the write-outs track the largest magnitude they split, the kernels
//     raise the decoder's range flag and the host fails the call (a DeepSDF decoder's activations are O(10); the weights are
//     checked when the planes are packed).
// ===================================================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#ifndef QSP_H2_EXP
#define QSP_H2_EXP 0      // timing experiments only (tools/micro/h2_tile.hip): bit 0 = no pinned issue order in the GEMM loop,
#endif                    // bit 1 = write-outs store without splitting, bit 2 = no weight loads in the loop, bit 3 = no MFMAs
#ifdef QSP_H2_STAMPS   // tools/micro/h2_tile.hip: shader-clock stamps of wave 0 at the phase boundaries of one tile
__device__ unsigned long long qsp_h2_ts[64];
__device__ int qsp_h2_nts;
#define QSP_HTS() { if (threadIdx.x == 0 && blockIdx.x == 0 && hts_n < 64) qsp_h2_ts[hts_n++] = __builtin_readcyclecounter(); }
#else
#define QSP_HTS()
#endif
constexpr int H2_THREADS = 256;
constexpr int LDH = 2 * LDA;            // row stride of the split image in halfs (= the f32 image's 2064 bytes)
constexpr float H2_MAX = 65504.f;

// position of (row, unit u) in the split image, in halfs; its lo' term is 8 halfs further
__device__ __forceinline__ int h2_at(int row, int u) { return row * LDH + (u >> 3) * 16 + (u & 7); }

// four consecutive units (u0 % 4 == 0) of one row: split and store (two 8-byte stores)
__device__ __forceinline__ void h2_store4(_Float16* img, int row, int u0, f32x4 v, float& amax) {
    amax = fmaxf(fmaxf(amax, fabsf(v[0])), fmaxf(fabsf(v[1]), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    asm volatile("" : "+v"(amax));      // (otherwise the maximum is deferred and every split value stays live until then)
    f16x4 hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const _Float16 h = (_Float16)v[q];
        hi[q] = h;
        lo[q] = (QSP_H2_EXP & 2) ? h : (_Float16)((v[q] - (float)h) * 2048.f);
    }
    // (two 8-byte stores: within a lane half rows r and r + 16 share banks, 24 % of the LDS-active cycles are such conflicts --
    // but the LDS pipe is only 14 % busy; exchanging halves with v_permlane32_swap for ONE conflict-free 16-byte store per lane
    // was measured 1 % slower: the write-out is bound by its VALU work, profiles/r02_c4_pmcL_summary.txt)
    _Float16* d = img + h2_at(row, u0);
    *reinterpret_cast<f16x4*>(d) = hi;
    *reinterpret_cast<f16x4*>(d + 8) = lo;
}

#define QSP_MFMA_H(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, acc_, 0, 0, 0)

template <int PF, int NCB>
struct WRingH {
    f32x4 q[PF][NCB][2];      // [slab in flight][column block][plane]
};

__device__ __forceinline__ f16x8 as_f16x8(f32x4 q) {
    union { f32x4 f; f16x8 h; } u;
    u.f = q;
    return u.h;
}

// Packed planes are addressed as (uniform base in SGPRs) + (this lane's 32-bit byte offset): one VGPR of address for the whole
// ring instead of a 64-bit pointer per fragment in flight.
typedef const __attribute__((address_space(1))) char* gbytes;
__device__ __forceinline__ f32x4 ldw(gbytes base, uint32_t voff) { return *reinterpret_cast<gptr4>(base + voff); }

template <int PF, int NCB>
__device__ __forceinline__ void ringh_prime(WRingH<PF, NCB>& R, const float4* __restrict__ w_, int cs, int lane) {
    gbytes w = (gbytes)w_;
    const uint32_t voff = 16u * lane;
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p) R.q[d][c][p] = ldw(w + ((size_t)c * cs + (d * 2 + p) * 64) * 16, voff);
}

// acc / acc2 [r][c] += image rows [32 r, 32 r + 32) x slabs [0, KS) * W for NCB column blocks; w = base of the first column
// block, consecutive column blocks `cs` float4 apart, nw / ncs the same for the next GEMM (its first PF slabs are fetched by
// this one's last iterations).  img = the first row of row block 0.
template <int KS, int PF, int NCB, int NR, bool HAND = true, bool PRIMED = false, bool RT = false>
__device__ __forceinline__ void gemm_h2(const _Float16* __restrict__ img, const float4* __restrict__ w_, int cs,
                                        const float4* __restrict__ nw_, int ncs, WRingH<PF, NCB>& R, f32x16 (&acc)[NR][NCB],
                                        f32x16 (&acc2)[NR][NCB], int lane, int ks_run = KS) {
    // RT: only the first ks_run slabs (a multiple of PF, at least 2 PF) carry weights -- narrow decoders
    const int ks_end = RT ? ks_run : KS;
    static_assert(KS % PF == 0 && KS >= 2 * PF, "slab count must be a multiple of the prefetch depth, at least twice it");
    static_assert(PF % 2 == 0, "the operand sets alternate between consecutive slabs");
    // HAND: this GEMM's first PF slabs were fetched by the previous one and it fetches the next one's (the ring lives across the
    // write-out in between).  Without: fetch them here (one exposed fetch per GEMM, 64 registers less held across the
    // write-outs) -- or, PRIMED, the caller has issued that fetch already (at the end of the preceding write-out: the ring is
    // then live across a barrier and the accumulator initialisation only); none for the next either way.
    if (!HAND && !PRIMED) ringh_prime(R, w_, cs, lane);
    gbytes w = (gbytes)w_;
    gbytes nw = HAND ? (gbytes)nw_ : w;         // (without: the last round's fetches land in registers nobody reads)
    if (!HAND) ncs = cs;
    uint32_t voff = 16u * lane;
    int b_idx = (lane & 31) * LDH + (lane >> 5) * 16;
    // Every vector-memory operation issued so far retires here -- the ring's first fragments (needed in a moment anyway) and,
    // above all, any register the allocator spilled around the write-out and reloads for the loop: a reload still pending at
    // the loop header makes the wait-count pass guard that register with vmcnt(0) in EVERY iteration, which serialises the
    // whole prefetch (measured: 38 instead of 29 kcycles per GEMM).  The empty asm pins the loop-carried address registers'
    // reloads in front of the wait.
    asm volatile("" : "+v"(voff), "+v"(b_idx));
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const _Float16* b_row = img + b_idx;
    // B operands of two consecutive slabs: set d & 1 is consumed by slab d of a round while the other is being read
    f16x8 bh[2][NR], bl[2][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        bh[0][r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH);
        bl[0][r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH + 8);
    }
    // The order below is the issue order (pinned with scheduling barriers: one wave per SIMD, nobody else fills the gaps):
    // a fragment register is refilled right after the last MFMA that reads it -- no second register set, no copies -- and the
    // refills and the LDS reads of the next slab sit between MFMAs, which keep the matrix pipe busy for 8 passes each.
#if QSP_H2_EXP & 1
#define QSP_PIN
#else
#define QSP_PIN __builtin_amdgcn_sched_barrier(0);
#endif
#if QSP_H2_EXP & 4
#define QSP_LDW(dst_, ...)
#else
#define QSP_LDW(dst_, ...) dst_ = ldw(__VA_ARGS__)
#endif
#if QSP_H2_EXP & 8
#undef QSP_MFMA_H
#define QSP_MFMA_H(acc_, a_, b_) asm volatile("" : "+a"(acc_) : "v"(a_), "v"(b_))
#endif
#pragma nounroll
    for (int ks = 0; ks < ks_end; ks += PF) {
        // fragments to fetch during this round: slabs ks + PF.. of this matrix, or the first PF slabs of the next one
        const bool more = ks + PF < ks_end;
        gbytes fb = more ? w + (size_t)(ks + PF) * 2 * 64 * 16 : nw;
        const size_t fcs = (size_t)(more ? cs : ncs) * 16;
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const _Float16* nb = b_row + 32 * (ks + d + 1);   // (the last slab reads one slab past the row: image or tail padding)
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const f16x8 wh = as_f16x8(R.q[d][c][0]), wl = as_f16x8(R.q[d][c][1]);
#pragma unroll
                for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc2[r][c], wl, bh[d & 1][r]);
                QSP_PIN
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) bh[(d & 1) ^ 1][r] = *reinterpret_cast<const f16x8*>(nb + r * 32 * LDH);
                } else {
                    QSP_LDW(R.q[d][c - 1][0], fb + (c - 1) * fcs + (d * 2 + 0) * 64 * 16, voff);
                }
                QSP_PIN
#pragma unroll
                for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc2[r][c], wh, bl[d & 1][r]);
                QSP_PIN
                if (c == 0) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) bl[(d & 1) ^ 1][r] = *reinterpret_cast<const f16x8*>(nb + r * 32 * LDH + 8);
                } else {
                    QSP_LDW(R.q[d][c - 1][1], fb + (c - 1) * fcs + (d * 2 + 1) * 64 * 16, voff);
                }
                QSP_PIN
#pragma unroll
                for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc[r][c], wh, bh[d & 1][r]);
                QSP_PIN
            }
            QSP_LDW(R.q[d][NCB - 1][0], fb + (NCB - 1) * fcs + (d * 2 + 0) * 64 * 16, voff);
            QSP_LDW(R.q[d][NCB - 1][1], fb + (NCB - 1) * fcs + (d * 2 + 1) * 64 * 16, voff);
            QSP_PIN
        }
    }
#undef QSP_PIN
#undef QSP_LDW
}

// ReLU masks of the split-fp16 tile: one 32-bit word per (row block, column-block pair), filled by 32 pushes in the write-out's
// (column block, register) order -- compare into VCC, then word = 2 word + carry: two instructions per value and no shift-count
// constants (the compiler's form keeps 32 of them in registers across the whole tile).  The k-th push ends up at bit 31 - k.
__device__ __forceinline__ void h2_mask_push(uint32_t& m, float x) {
    asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(x) : "vcc");
}
__device__ __forceinline__ float h2_mask_sel(float v, uint32_t m, int k) {      // v if push k was positive, else +0
    return __int_as_float(__float_as_int(v) & __builtin_amdgcn_sbfe((int)m, 31 - k, 1));
}

// Layer 0 as ONE slab of the same product: image columns 0..2 hold the point (3..15 zero), the packed matrix the three xyz
// columns of W0 (its 64 code columns are folded into the per-hypothesis bias c0, the accumulators' initial value).
template <int NR, int NCB>
__device__ __forceinline__ void gemm_l0_h2(const _Float16* __restrict__ img, const float4* __restrict__ w_, f32x16 (&acc)[NR][NCB],
                                           f32x16 (&acc2)[NR][NCB], int lane) {
    gbytes w = (gbytes)w_;
    const uint32_t voff = 16u * lane;
    const _Float16* b_row = img + (lane & 31) * LDH + (lane >> 5) * 16;
    f16x8 bh[NR], bl[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        bh[r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH);
        bl[r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH + 8);
    }
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const f16x8 wh = as_f16x8(ldw(w + (size_t)(c * 2 + 0) * 64 * 16, voff)), wl = as_f16x8(ldw(w + (size_t)(c * 2 + 1) * 64 * 16, voff));
#pragma unroll
        for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc2[r][c], wl, bh[r]);
#pragma unroll
        for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc2[r][c], wh, bl[r]);
#pragma unroll
        for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc[r][c], wh, bh[r]);
    }
}

// One slab of a packed matrix against one slab of the image: img_slab = the slab's first half of row 0, w = the slab's hi fragment
// of this wave's first column block, consecutive column blocks cs float4 apart.  (Narrow decoders: the xyz slab of the latent_in
// layer, which sits behind a gap of all-zero slabs.)
template <int NR, int NCB>
__device__ __forceinline__ void gemm_slab_h2(const _Float16* __restrict__ img_slab, const float4* __restrict__ w_, int cs,
                                             f32x16 (&acc)[NR][NCB], f32x16 (&acc2)[NR][NCB], int lane) {
    gbytes w = (gbytes)w_;
    const uint32_t voff = 16u * lane;
    const _Float16* b_row = img_slab + (lane & 31) * LDH + (lane >> 5) * 16;
    f16x8 bh[NR], bl[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        bh[r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH);
        bl[r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH + 8);
    }
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const f16x8 wh = as_f16x8(ldw(w + ((size_t)c * cs + 0) * 16, voff)), wl = as_f16x8(ldw(w + ((size_t)c * cs + 64) * 16, voff));
#pragma unroll
        for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc2[r][c], wl, bh[r]);
#pragma unroll
        for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc2[r][c], wh, bl[r]);
#pragma unroll
        for (int r = 0; r < NR; ++r) QSP_MFMA_H(acc[r][c], wh, bh[r]);
    }
}

// image (32 NR points x 512) x a matrix packed in three column blocks (96 padded columns), as 32x32 tiles.  NR = 2 (six tiles):
// waves 0, 1 take column blocks 0, 1 for both point blocks, waves 2, 3 column block 2 for one point block each.  NR = 1 (three
// tiles): waves 0..2 one column block each, wave 3 idle.  side_c0 / side_row / side_count say which tiles a wave holds;
// out[r][g] = the lane's register quad g of its r-th tile: columns 32 c0 + 8 g + 4 (lane >> 5) .. + 3.
// With eight waves (NW = 8): one tile per wave -- NR = 2: wave w < 6 holds column block w >> 1 of point block w & 1; NR = 1: waves 0..2.
template <int NR, int NW> __device__ __forceinline__ int side_c0(int wave) {
    if (NW == 8) return NR == 2 ? (wave < 6 ? wave >> 1 : 0) : (wave < 3 ? wave : 0);
    return NR == 2 ? (wave < 2 ? wave : 2) : wave;
}
template <int NR, int NW> __device__ __forceinline__ int side_count(int wave) {
    if (NW == 8) return NR == 2 ? (wave < 6 ? 1 : 0) : (wave < 3 ? 1 : 0);
    return NR == 2 ? (wave < 2 ? 2 : 1) : (wave < 3 ? 1 : 0);
}
template <int NR, int NW> __device__ __forceinline__ int side_row(int wave, int r) {
    if (NW == 8) return NR == 2 ? (wave & 1) : 0;
    return NR == 2 ? (wave < 2 ? r : wave - 2) : 0;
}
template <int PF, int NR, int NW, bool RT = false>
__device__ __forceinline__ void gemm_side_h2(const _Float16* __restrict__ img, const float4* __restrict__ wb, int wave, int lane,
                                             f32x4 (&out)[NR][4], int ks_run = HID / 16) {
    constexpr int KSH = HID / 16, CS = KSH * 2 * 64;
    const int c0 = side_c0<NR, NW>(wave);
    const float4* w0 = wb + (size_t)((c0 < 3 ? c0 : 0) * KSH * 2) * 64;
    WRingH<PF, 1> ring0;
    f32x16 g0[NR][1], g2[NR][1];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) { g0[r][0][i] = 0.f; g2[r][0][i] = 0.f; }
    if (side_count<NR, NW>(wave) == NR) {
        gemm_h2<KSH, PF, 1, NR, false, false, RT>(img, w0, CS, w0, CS, ring0, g0, g2, lane, ks_run);
    } else if (side_count<NR, NW>(wave) == 1) {          // (NR == 2 only: one of the two point blocks)
        f32x16 g01[1][1], g21[1][1];
#pragma unroll
        for (int i = 0; i < 16; ++i) { g01[0][0][i] = 0.f; g21[0][0][i] = 0.f; }
        gemm_h2<KSH, PF, 1, 1, false, false, RT>(img + side_row<NR, NW>(wave, 0) * 32 * LDH, w0, CS, w0, CS, ring0, g01, g21, lane, ks_run);
        g0[0][0] = g01[0][0];
        g2[0][0] = g21[0][0];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) out[r][g][q] = fmaf(g2[r][0][4 * g + q], 0.00048828125f, g0[r][0][4 * g + q]);
}

// Network on the split-fp16 pipe for the tile staged in s.code / s.xin / s.c0 / s.c4, 256 threads.  On return s.y[row] = sdf
// value and, with BWD, rows of d sdf / d [code | xyz] in s.act (row stride LDG) like mlp_tile<true>.  amax: running maximum of
// the magnitudes this thread has split (the caller compares it with H2_MAX once per kernel).
// NW: waves of the workgroup -- 4 (one per SIMD, 512 registers, 128 units x 32 NR points each) or 8 (two per SIMD, 256 registers,
// 64 units each: while one wave of a SIMD is in a write-out, which is VALU work, the other can still be feeding the matrix pipe).
// NARROW: the form for decoders much smaller than the 8 x 512 shape they are embedded in (MlpParams::narrow): identity slots are
// skipped (a ReLU output passes through relu(1 . h) unchanged, and in the backward pass the identity's mask is the producing
// layer's own), the k-loops stop behind the last slab that carries weights, and a wave whose column blocks do not exist runs a
// zero-trip GEMM (its write-out stores relu(0) = 0: every column of the image is rewritten by every executed layer, as in the
// full-width form, so nothing the caller's epilogue left in the image survives).  Everything skipped is a product with an exact zero, so the values are those of the embedded form
// (up to the f32 rounding of x_hi + 2^-11 x_lo' that an identity layer applies to a few values and the skip does not).  Every
// GEMM fetches its own first slabs (no hand-over: what follows a layer depends on the decoder).
template <bool BWD, int PF, bool HAND = !BWD, int NR = 2, int NW = 4, bool NARROW = false>     // NR: point blocks of 32
__device__ __forceinline__ void mlp_tile_h2(MlpSmem& s, const MlpParams* __restrict__ Pm, float& amax, bool stage = true) {
    constexpr int NCB = 16 / NW, NT = 64 * NW;      // column blocks of 32 units per wave; threads
    static_assert(!(BWD && HAND), "the forward+backward tile fetches each matrix's first slabs itself (layer 7 would hand over to the wrong one)");
    static_assert(!(NARROW && HAND), "the narrow form has no hand-over");
    int hts_n = 0;
    (void)hts_n;
    QSP_HTS()
    int tid = threadIdx.x;
    // opaque per tile: otherwise every LDS address below that depends on the lane is computed once per kernel, ahead of the
    // caller's tile loop, and held (spilled) across it
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5;
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const MlpParams& P = Pm[oz];
    constexpr int TP = 32 * NR;
    f32x16 acc[NR][NCB], acc2[NR][NCB];
    uint32_t mk[8][NR][NCB / 2];                            // ReLU masks [layer][row block][column-block pair]
    _Float16* img = reinterpret_cast<_Float16*>(s.act);
    float* bias_sh = s.stash;                              // [7][512] biases of layers 1..7 (the stash is free until layer 4's backward)
    if (stage) {      // constants of the decoder: once per workgroup (the forward+backward tile, whose stash doubles as the bias
                      // store, re-stages the biases itself at its end)
        for (int i = tid; i < HID; i += NT) s.w8[i] = P.w8[i];
#pragma unroll
        for (int l = 1; l < 8; ++l)
            for (int i = tid; i < HID; i += NT) bias_sh[(l - 1) * HID + i] = P.bias[l][i];
    }
    if (tid < TP) {       // the point as slab 0 of the image: columns 0..2 = xyz split, 3..15 zero
        const f32x4 x = lds4(s.xin + 4 * tid);
        f16x8 hi, lo, z;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; lo[j] = (_Float16)0.f; z[j] = (_Float16)0.f; }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            hi[j] = (_Float16)x[j];
            lo[j] = (_Float16)((x[j] - (float)hi[j]) * 2048.f);
        }
        amax = fmaxf(amax, fmaxf(fabsf(x[0]), fmaxf(fabsf(x[1]), fabsf(x[2]))));
        _Float16* d = img + tid * LDH;
        *reinterpret_cast<f16x8*>(d) = hi;
        *reinterpret_cast<f16x8*>(d + 8) = lo;
        *reinterpret_cast<f16x8*>(d + 16) = z;
        *reinterpret_cast<f16x8*>(d + 24) = z;
    }
    const int cb0 = NCB * wave;
    constexpr int KSH = HID / 16, KS4 = K4 / 16, CS = KSH * 2 * 64, CS4 = KS4 * 2 * 64;
#define QSP_WH(L, KS_) (P.wfh[L] + (size_t)(cb0 * (KS_) * 2) * 64)
#define QSP_WBH(L) (P.wbh[L] + (size_t)(cb0 * KSH * 2) * 64)
    WRingH<PF, NCB> ring;
    if (!NARROW) ringh_prime(ring, QSP_WH(1, KSH), CS, lane);      // (layer 1's first slabs: in flight behind layer 0)
    QSP_HTS()
    __syncthreads();
    QSP_HTS()
    // (both accumulator sets stay in the Acc file until a write-out reads them, one register quad pair at a time)
#define QSP_PIN_ACC()                                                                                                    \
    _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_)                 \
        asm volatile("" : "+a"(acc[r_][c_]), "+a"(acc2[r_][c_]));
    // one hidden layer: accumulators start from the bias, GEMM, then (barrier) main + 2^-11 cross, ReLU, split, (barrier)
#define QSP_FWDH(L, BIASPTR, GEMM_STMT, PRIME_NEXT)                                                                      \
  if (NARROW && P.skip[L]) {       /* identity slot: the activations stand; its ReLU mask is the producing layer's */      \
    if (BWD) {                                                                                                           \
        _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) _Pragma("unroll") for (int cp_ = 0; cp_ < NCB / 2; ++cp_)      \
            mk[L][r_][cp_] = mk[(L) > 0 ? (L) - 1 : 0][r_][cp_];                                                         \
    }                                                                                                                    \
  } else {                                                                                                               \
    /* (narrow: a wave whose column blocks do not exist runs a zero-trip GEMM and writes relu(0) = 0: straight-line code) */  \
    const bool act_ = !NARROW || cb0 < (int)P.ncb_out[L];                                                                 \
    _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                \
        const f32x4 bv_ = lds4((BIASPTR) + 32 * NCB * wave + 32 * c_ + 8 * g_ + 4 * h);                                     \
        _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {             \
            acc[r_][c_][4 * g_ + q_] = bv_[q_];                                                                          \
            acc2[r_][c_][4 * g_ + q_] = 0.f;                                                                             \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                    \
    QSP_HTS()                                                                                                            \
    GEMM_STMT;                                                                                                           \
    QSP_HTS()                                                                                                            \
    QSP_PIN_ACC()                                                                                                        \
    __syncthreads();                                                                                                     \
    QSP_HTS()                                                                                                            \
    {                                                                                                                    \
        uint32_t m_[NR][NCB / 2] = {};                                                                                   \
        _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {            \
            const int u0_ = 32 * NCB * wave + 32 * c_ + 8 * g_ + 4 * h;                                                     \
            _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) {                                                          \
                f32x4 v_;                                                                                                \
                _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                       \
                    const float x_ = fmaf(acc2[r_][c_][4 * g_ + q_], 0.00048828125f, acc[r_][c_][4 * g_ + q_]);         \
                    if (BWD) h2_mask_push(m_[r_][c_ >> 1], x_);                                                          \
                    v_[q_] = x_ > 0.f ? x_ : 0.f;                                                                        \
                }                                                                                                        \
                h2_store4(img, 32 * r_ + (lane & 31), u0_, v_, amax);                                                    \
            }                                                                                                            \
            __builtin_amdgcn_sched_barrier(0);     /* one register quad pair at a time: bounded temporaries */            \
        }                                                                                                                \
        _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) _Pragma("unroll") for (int cp_ = 0; cp_ < NCB / 2; ++cp_) {    \
            asm volatile("" : "+v"(m_[r_][cp_]));                                                                        \
            mk[L][r_][cp_] = m_[r_][cp_];                                                                                \
        }                                                                                                                \
    }                                                                                                                    \
    if (!HAND && !NARROW) { PRIME_NEXT; }     /* the next GEMM's first slabs, behind the barrier and the accumulator initialisation */  \
    QSP_HTS()                                                                                                            \
    __syncthreads();                                                                                                     \
    QSP_HTS()                                                                                                            \
  }
#define QSP_GEMMF(L, KS_, NXW, NKS)                                                                                     \
    if constexpr (NARROW) {                                                                                              \
        gemm_h2<KS_, PF, NCB, NR, false, false, true>(img, QSP_WH(L, KS_), (KS_) * 2 * 64, QSP_WH(L, KS_), (KS_) * 2 * 64, ring, acc, \
                                                      acc2, lane, act_ ? (int)P.ks_in[L] : 0);                           \
        if ((L) == 4 && act_ && (int)P.ks_in[4] < KS4)      /* the xyz slab of the latent_in layer, behind the all-zero slabs */ \
            gemm_slab_h2<NR, NCB>(img + 32 * (KS4 - 1), QSP_WH(4, KS4) + (size_t)((KS4 - 1) * 2) * 64, CS4, acc, acc2, lane); \
    } else {                                                                                                             \
        gemm_h2<KS_, PF, NCB, NR, HAND, true>(img, QSP_WH(L, KS_), (KS_) * 2 * 64, NXW, (NKS) * 2 * 64, ring, acc, acc2, lane); \
    }
    QSP_FWDH(0, s.c0, (gemm_l0_h2<NR, NCB>(img, QSP_WH(0, 1), acc, acc2, lane)), (void)0)      // (narrow: every wave -- zero weights beyond the width)
    QSP_FWDH(1, bias_sh + 0 * HID, QSP_GEMMF(1, KSH, QSP_WH(2, KSH), KSH), ringh_prime(ring, QSP_WH(2, KSH), CS, lane))
    QSP_FWDH(2, bias_sh + 1 * HID, QSP_GEMMF(2, KSH, QSP_WH(3, KSH), KSH), ringh_prime(ring, QSP_WH(3, KSH), CS, lane))
    QSP_FWDH(3, bias_sh + 2 * HID, QSP_GEMMF(3, KSH, QSP_WH(4, KS4), KS4), ringh_prime(ring, QSP_WH(4, KS4), CS4, lane))
    if (tid < TP * 3) {        // the skip connection's xyz into columns 445..447 (zeros from layer 3's write-out until now)
        const int row = tid / 3, ci = tid - row * 3;
        const float x = s.xin[row * 4 + ci];
        const _Float16 xh = (_Float16)x;
        _Float16* d = img + h2_at(row, SKIP_COL + ci);
        d[0] = xh;
        d[8] = (_Float16)((x - (float)xh) * 2048.f);
        amax = fmaxf(amax, fabsf(x));
    }
    __syncthreads();
    QSP_FWDH(4, s.c4, QSP_GEMMF(4, KS4, QSP_WH(5, KSH), KSH), ringh_prime(ring, QSP_WH(5, KSH), CS, lane))
    QSP_FWDH(5, bias_sh + 4 * HID, QSP_GEMMF(5, KSH, QSP_WH(6, KSH), KSH), ringh_prime(ring, QSP_WH(6, KSH), CS, lane))
    QSP_FWDH(6, bias_sh + 5 * HID, QSP_GEMMF(6, KSH, QSP_WH(7, KSH), KSH), ringh_prime(ring, QSP_WH(7, KSH), CS, lane))
    QSP_FWDH(7, bias_sh + 6 * HID, QSP_GEMMF(7, KSH, QSP_WH(1, KSH), KSH), (void)0)
#undef QSP_FWDH
#undef QSP_GEMMF
    // ---- layer 8: 512 -> 1, tanh (f32): wave = k segment of 128 (waves 0..3 whatever NW is: the same partial sums in the same
    // order, so the value does not depend on the number of waves), lane = row; a7 = hi + 2^-11 lo' from the two planes
    // (every layer's write-out is the same code: a special case for layer 7 costs the register allocator its footing) ---------
    if (NW == 4 || wave < 4) {
        const _Float16* a = img + lane * LDH + 16 * (16 * wave);
        const float* w = s.w8 + 128 * wave;
        float pa[4] = {0.f, 0.f, 0.f, 0.f};        // four chains: one dependent chain of 128 multiply-adds is latency-bound
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const f16x8 hi = *reinterpret_cast<const f16x8*>(a + 16 * g), lo = *reinterpret_cast<const f16x8*>(a + 16 * g + 8);
            const f32x4 w0 = lds4(w + 8 * g), w1 = lds4(w + 8 * g + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                pa[j & 3] = fmaf(fmaf((float)lo[j], 0.00048828125f, (float)hi[j]), j < 4 ? w0[j] : w1[j - 4], pa[j & 3]);
        }
        const float part = (pa[0] + pa[1]) + (pa[2] + pa[3]);
        if (lane < TP) s.red[wave * TP + lane] = part;
    }
    __syncthreads();
    if (tid < TP) {
        float t = P.b8;
#pragma unroll
        for (int q = 0; q < 4; ++q) t += s.red[q * TP + tid];
        float dy_;
        s.y[tid] = mlp_output(t, P.use_tanh, dy_);
        s.dy[tid] = dy_;
    }
    __syncthreads();
    QSP_HTS()
#ifdef QSP_H2_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0) qsp_h2_nts = hts_n;
#endif
    if constexpr (BWD) {
        // ---- backward seed: d y / d a7 = (1 - y^2) * w8[unit] * [a7 > 0] ---------------------------------------------
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int p = 32 * r + (lane & 31);
            const float dy = s.dy[p];
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int u0 = 32 * NCB * wave + 32 * c + 8 * g + 4 * h;
                    const f32x4 wv = lds4(s.w8 + u0);
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q] = h2_mask_sel(dy * wv[q], mk[7][r][c >> 1], (c & 1) * 16 + 4 * g + q);
                    }
                    h2_store4(img, p, u0, v, amax);
                }
        }
        if (!NARROW) ringh_prime(ring, QSP_WBH(7), CS, lane);
        __syncthreads();
        // ---- backward through layers 7..1: g_in = g_a . W_L, masked by layer L-1; the skip gradient of layer 4 to the stash ----
#define QSP_BWDH(L, KS_, NWB)                                                                                            \
  if (!(NARROW && P.skip[L])) {        /* (an identity slot passes the gradient on: its mask is applied by the layer below) */ \
    const bool act_ = !NARROW || cb0 < (int)P.ncb_in[L];                                                                  \
    _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_)                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { acc[r_][c_][i_] = 0.f; acc2[r_][c_][i_] = 0.f; }             \
    if constexpr (NARROW)                                                                                                \
        gemm_h2<KS_, PF, NCB, NR, false, false, true>(img, QSP_WBH(L), CS, QSP_WBH(L), CS, ring, acc, acc2, lane,             \
                                                      act_ ? min((int)P.ks_out[L], KS_) : 0);                            \
    else                                                                                                                 \
        gemm_h2<KS_, PF, NCB, NR, HAND, true>(img, QSP_WBH(L), CS, NWB, CS, ring, acc, acc2, lane);                        \
    QSP_PIN_ACC()                                                                                                        \
    __syncthreads();                                                                                                     \
    _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                \
        const int u0_ = 32 * NCB * wave + 32 * c_ + 8 * g_ + 4 * h;                                                      \
        _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) {                                                              \
            const int p_ = 32 * r_ + (lane & 31);                                                                      \
            f32x4 v_;                                                                                                    \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                           \
                const float x_ = fmaf(acc2[r_][c_][4 * g_ + q_], 0.00048828125f, acc[r_][c_][4 * g_ + q_]);             \
                const float m_ = h2_mask_sel(x_, mk[(L) - 1][r_][c_ >> 1], (c_ & 1) * 16 + 4 * g_ + q_);                       \
                v_[q_] = m_;                                                                                             \
            }                                                                                                            \
            h2_store4(img, p_, u0_, v_, amax);                                                                           \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                    \
    if (!NARROW && (L) > 1) ringh_prime(ring, NWB, CS, lane);     /* the next layer's first slabs (layer 0's product has its own ring) */ \
    __syncthreads();                                                                                                     \
  }
        QSP_BWDH(7, KSH, QSP_WBH(6))
        QSP_BWDH(6, KSH, QSP_WBH(5))
        QSP_BWDH(5, KSH, QSP_WBH(4))
        {   // the skip connection's gradient d y / d [code | xyz] = g_a4 . W4[:, 445:512]: its own 64 x 96 product, to the stash
            // in f32 (layer 4's write-out below masks those columns to zero like any dead unit: mk[3] has no bit set there)
            f32x4 sk[NR][4];
            gemm_side_h2<PF, NR, NW, NARROW>(img, P.wbh4s, wave, lane, sk, NARROW ? (int)P.ks_out[4] : KSH);
            const int c0 = side_c0<NR, NW>(wave);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r >= side_count<NR, NW>(wave)) break;
                const int p = 32 * side_row<NR, NW>(wave, r) + (lane & 31);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k0 = 32 * c0 + 8 * g + 4 * h;
                    if (k0 < NIN) *reinterpret_cast<f32x4*>(s.stash + p * LDST + k0) = sk[r][g];
                }
            }
        }
        QSP_BWDH(4, KSH, QSP_WBH(3))
        // layer 3 has 445 outputs: its backward contraction runs over K4 = 448 gradient columns (445..447 are zeros)
        QSP_BWDH(3, KS4, QSP_WBH(2))
        QSP_BWDH(2, KSH, QSP_WBH(1))
        QSP_BWDH(1, KSH, QSP_WBH(1))       // (hand-over fetch unused: layer 0's GEMM has its own shape and primes its own ring)
#undef QSP_BWDH
        // ---- backward through layer 0: 67 (padded 96) input columns = 2 x 3 output tiles: waves 0, 1 take column blocks 0, 1
        // for both point blocks, waves 2, 3 column block 2 (inputs 64..66) for one point block each -------------------------
        {
            // the biases of layers 1..7 for the NEXT tile of this workgroup (the stash that holds them in the forward pass is
            // about to be read for the last time): loads issued here, in flight behind the product below, stored at the very end
            constexpr int BPT = HID / NT;       // bias values per thread and layer
            float bnext[7 * BPT];
#pragma unroll
            for (int l = 1; l < 8; ++l)
#pragma unroll
                for (int i = 0; i < BPT; ++i) bnext[BPT * (l - 1) + i] = P.bias[l][tid + i * NT];
            __builtin_amdgcn_sched_barrier(0);
            f32x4 gl[NR][4];
            gemm_side_h2<PF, NR, NW, NARROW>(img, P.wbh[0], wave, lane, gl, NARROW ? (int)P.ks_out[0] : KSH);
            const int c0 = side_c0<NR, NW>(wave);
            __syncthreads();
            // D[i = input column within block c0][j = point]: four consecutive input columns per register quad
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (r >= side_count<NR, NW>(wave)) break;
                const int p = 32 * side_row<NR, NW>(wave, r) + (lane & 31);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k0 = 32 * c0 + 8 * g + 4 * h;
                    if (k0 < NIN) {     // 64..67 is the last useful quad (67 itself is padding inside both row strides)
                        const f32x4 st = lds4(s.stash + p * LDST + k0);
                        f32x4 v;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = gl[r][g][q] + st[q];
                        *reinterpret_cast<f32x4*>(s.act + p * LDG + k0) = v;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int l = 1; l < 8; ++l)
#pragma unroll
                for (int i = 0; i < BPT; ++i) bias_sh[(l - 1) * HID + tid + i * NT] = bnext[BPT * (l - 1) + i];
        }
    }
#undef QSP_WH
#undef QSP_WBH
#undef QSP_PIN_ACC
}

// ===================================================================================================================
// The SCREENING tile (round 3): the decoder on the fp16 matrix pipe with ONE term per operand -- x_hi . w_hi only, f32
// accumulation -- over 128 points per workgroup.  It is not a precision mode: its values are never used as SDF values.  The render
// term clamps (loss_utils.py:40-48, loss.py:84-122): a ray sample with |sdf| >= th contributes an occupancy of exactly 0 or 1, so
// away from the surface only the SIGN of the value matters.  k_mlp_fwd_h1 evaluates every ray sample with this tile and keeps a
// list of the samples with |s1| < th + margin; only those are evaluated again by the three-product tile (k_mlp_fwd_h2), whose
// values overwrite s1.  With margin >= |s1 - s3| every clamp and every band decision -- hence K, n_valid, H, b -- equals the
// all-three-product path's bit for bit (tests/test_gpu_screening.py); the margin is a multiple of the largest |s1 - s3| measured
// (profiles/r03_screen_margin.txt).
//   * one plane: the activation image is 2 bytes per value, so 128 points fit the LDS that holds 64 points of the split image
//     -- half the weight bytes per point, and only the hi planes of the packed weights are fetched: a quarter of the split
//     tile's weight stream per point;
//   * one accumulator set: a wave's 128 units x 128 points are 4 x 4 tiles of v_mfma_f32_32x32x16_f16 = the 256 AccVGPRs;
//   * a third of the matrix-pipe work per point.
// Same packed weights (MlpParams::wfh), same lane maps, same layer-0 / skip handling as mlp_tile_h2.
constexpr int H1_ROWS = 128;
constexpr int LDH1 = HID + 8;     // image row stride in halfs (1040 B: 16 rows of a ds_read_b128 lane group fall on 16 different slots)
struct __attribute__((aligned(16))) MlpSmemH1 {
    _Float16 img[H1_ROWS * LDH1 + 64];      // (+ tail: the operand prefetch of a row's last slab reads one slab further)
    float bias[7 * HID];                    // biases of layers 1..7
    float c0[HID];                          // layer 0 without its xyz part, per hypothesis (k_c0)
    float c4[HID];                          // layer 4's bias with the code part of the skip connection
    float w8[HID];
    float xin[H1_ROWS * 4];
    float y[H1_ROWS];
    float red[4 * H1_ROWS];
};

template <int PF, int NCB>
struct WRingH1 {
    f32x4 q[PF][NCB];       // hi plane of [slab in flight][column block]
};
template <int PF, int NCB>
__device__ __forceinline__ void ringh1_prime(WRingH1<PF, NCB>& R, const float4* __restrict__ w_, int cs, int lane) {
    gbytes w = (gbytes)w_;
    const uint32_t voff = 16u * lane;
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int c = 0; c < NCB; ++c) R.q[d][c] = ldw(w + ((size_t)c * cs + (d * 2) * 64) * 16, voff);
}

// acc[r][c] += image rows [32 r, 32 r + 32) x slabs [0, KS) * W_hi for NCB column blocks (packed like gemm_h2's, lo' planes
// skipped); the last PF slabs' refills fetch the first PF slabs of the NEXT matrix (nw_, ncs).
template <int KS, int PF, int NCB, int NR>
__device__ __forceinline__ void gemm_h1(const _Float16* __restrict__ img, const float4* __restrict__ w_, int cs,
                                        const float4* __restrict__ nw_, int ncs, WRingH1<PF, NCB>& R, f32x16 (&acc)[NR][NCB], int lane) {
    static_assert(KS % PF == 0 && KS >= 2 * PF && PF % 2 == 0 && NR % NCB == 0, "shape");
    constexpr int RPC = NR / NCB;          // row blocks of the next slab's operands read behind each column block's MFMAs
    gbytes w = (gbytes)w_;
    gbytes nw = (gbytes)nw_;
    uint32_t voff = 16u * lane;
    int b_idx = (lane & 31) * LDH1 + (lane >> 5) * 8;
    asm volatile("" : "+v"(voff), "+v"(b_idx));
    __builtin_amdgcn_s_waitcnt(0x0F70);      // (see gemm_h2: no spill reload may be pending at the loop header)
    const _Float16* b_row = img + b_idx;
    f16x8 bh[2][NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) bh[0][r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH1);
#pragma nounroll
    for (int ks = 0; ks < KS; ks += PF) {
        const bool more = ks + PF < KS;
        gbytes fb = more ? w + (size_t)(ks + PF) * 2 * 64 * 16 : nw;
        const size_t fcs = (size_t)(more ? cs : ncs) * 16;
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const _Float16* nb = b_row + 16 * (ks + d + 1);
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const f16x8 wh = as_f16x8(R.q[d][c]);
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[d & 1][r], acc[r][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < RPC; ++rr)
                    bh[(d & 1) ^ 1][c * RPC + rr] = *reinterpret_cast<const f16x8*>(nb + (c * RPC + rr) * 32 * LDH1);
                if (c > 0) R.q[d][c - 1] = ldw(fb + (c - 1) * fcs + (size_t)(d * 2) * 64 * 16, voff);
                __builtin_amdgcn_sched_barrier(0);
            }
            R.q[d][NCB - 1] = ldw(fb + (NCB - 1) * fcs + (size_t)(d * 2) * 64 * 16, voff);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// four consecutive units of one row -> fp16, one 8-byte store
__device__ __forceinline__ void h1_store4(_Float16* img, int row, int u0, f32x4 v, float& amax) {
    amax = fmaxf(fmaxf(amax, fabsf(v[0])), fmaxf(fabsf(v[1]), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    asm volatile("" : "+v"(amax));
    f16x4 hi;
#pragma unroll
    for (int q = 0; q < 4; ++q) hi[q] = (_Float16)v[q];
    *reinterpret_cast<f16x4*>(img + row * LDH1 + u0) = hi;
}

// 128 points staged in s.xin / s.c0 / s.c4 (+ the decoder's constants in s.bias / s.w8) -> s.y[row] = screening value.
// NW waves: 4 (one per SIMD, 128 units each) or 8 (two per SIMD, 64 units each: a wave's write-out overlaps the other's GEMM).
template <int PF, int NW = 4>
__device__ __forceinline__ void mlp_tile_h1(MlpSmemH1& s, const MlpParams* __restrict__ Pm, float& amax) {
    constexpr int NR = 4, NCB = 16 / NW;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int h = lane >> 5;
    int oz;
    asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
    const MlpParams& P = Pm[oz];
    f32x16 acc[NR][NCB];
    _Float16* img = s.img;
    if (tid < H1_ROWS) {      // the point as slab 0 of the image: columns 0..2 = xyz, 3..15 zero
        const f32x4 x = lds4(s.xin + 4 * tid);
        f16x8 hi, z;
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; z[j] = (_Float16)0.f; }
#pragma unroll
        for (int j = 0; j < 3; ++j) hi[j] = (_Float16)x[j];
        amax = fmaxf(amax, fmaxf(fabsf(x[0]), fmaxf(fabsf(x[1]), fabsf(x[2]))));
        _Float16* d = img + tid * LDH1;
        *reinterpret_cast<f16x8*>(d) = hi;
        *reinterpret_cast<f16x8*>(d + 8) = z;
    }
    const int cb0 = NCB * wave;
    constexpr int KSH = HID / 16, KS4 = K4 / 16, CS = KSH * 2 * 64, CS4 = KS4 * 2 * 64;
#define QSP_WH(L, KS_) (P.wfh[L] + (size_t)(cb0 * (KS_) * 2) * 64)
    WRingH1<PF, NCB> ring;
    ringh1_prime(ring, QSP_WH(1, KSH), CS, lane);
    __syncthreads();
#define QSP_FWD1(L, BIASPTR, GEMM_STMT)                                                                                  \
    _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                \
        const f32x4 bv_ = lds4((BIASPTR) + 32 * NCB * wave + 32 * c_ + 8 * g_ + 4 * h);                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)               \
            acc[r_][c_][4 * g_ + q_] = bv_[q_];                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                    \
    GEMM_STMT;                                                                                                           \
    /* the accumulators stay in the Acc file until the write-out reads them one register quad at a time: without this the */ \
    /* register allocator forms VGPR copies of ~50 of them in the GEMM loop's latch, every iteration, and spills those     */ \
    _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_)                 \
        asm volatile("" : "+a"(acc[r_][c_]));                                                                          \
    __syncthreads();                                                                                                     \
    _Pragma("unroll") for (int c_ = 0; c_ < NCB; ++c_) _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                \
        const int u0_ = 32 * NCB * wave + 32 * c_ + 8 * g_ + 4 * h;                                                      \
        _Pragma("unroll") for (int r_ = 0; r_ < NR; ++r_) {                                                              \
            f32x4 v_;                                                                                                    \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                           \
                const float x_ = acc[r_][c_][4 * g_ + q_];                                                             \
                v_[q_] = x_ > 0.f ? x_ : 0.f;                                                                            \
            }                                                                                                            \
            h1_store4(img, 32 * r_ + (lane & 31), u0_, v_, amax);                                                        \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                    \
    __syncthreads();
#define QSP_GEMM1(L, KS_, NXW, NKS) gemm_h1<KS_, PF, NCB, NR>(img, QSP_WH(L, KS_), (KS_) * 2 * 64, NXW, (NKS) * 2 * 64, ring, acc, lane)
    {   // layer 0: one slab of the same product (xyz columns of W0; the code part is s.c0)
        gbytes w = (gbytes)QSP_WH(0, 1);
        const uint32_t voff = 16u * lane;
        const _Float16* b_row = img + (lane & 31) * LDH1 + (lane >> 5) * 8;
        QSP_FWD1(0, s.c0, {
            f16x8 bh0[NR];
            _Pragma("unroll") for (int r = 0; r < NR; ++r) bh0[r] = *reinterpret_cast<const f16x8*>(b_row + r * 32 * LDH1);
            _Pragma("unroll") for (int c = 0; c < NCB; ++c) {
                const f16x8 wh = as_f16x8(ldw(w + (size_t)(c * 2) * 64 * 16, voff));
                _Pragma("unroll") for (int r = 0; r < NR; ++r)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh0[r], acc[r][c], 0, 0, 0);
            }
        })
    }
    QSP_FWD1(1, s.bias + 0 * HID, QSP_GEMM1(1, KSH, QSP_WH(2, KSH), KSH))
    QSP_FWD1(2, s.bias + 1 * HID, QSP_GEMM1(2, KSH, QSP_WH(3, KSH), KSH))
    QSP_FWD1(3, s.bias + 2 * HID, QSP_GEMM1(3, KSH, QSP_WH(4, KS4), KS4))
    if (tid < H1_ROWS) {       // the skip connection's xyz into columns 445..447 (zeros from layer 3's write-out until now)
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float x = s.xin[tid * 4 + ci];
            img[tid * LDH1 + SKIP_COL + ci] = (_Float16)x;
        }
    }
    __syncthreads();
    QSP_FWD1(4, s.c4, QSP_GEMM1(4, KS4, QSP_WH(5, KSH), KSH))
    QSP_FWD1(5, s.bias + 4 * HID, QSP_GEMM1(5, KSH, QSP_WH(6, KSH), KSH))
    QSP_FWD1(6, s.bias + 5 * HID, QSP_GEMM1(6, KSH, QSP_WH(7, KSH), KSH))
    QSP_FWD1(7, s.bias + 6 * HID, QSP_GEMM1(7, KSH, QSP_WH(1, KSH), KSH))
#undef QSP_FWD1
#undef QSP_GEMM1
#undef QSP_WH
    // layer 8: 512 -> 1, tanh: wave = k segment of 128 (waves 0..3 whatever NW is: same partial sums, same order), lane = rows
    // lane and lane + 64
    if (NW == 4 || wave < 4) {
        const float* w = s.w8 + 128 * wave;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const _Float16* a = img + (lane + 64 * rr) * LDH1 + 128 * wave;
            float pa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const f16x8 hi = *reinterpret_cast<const f16x8*>(a + 8 * g);
                const f32x4 w0 = lds4(w + 8 * g), w1 = lds4(w + 8 * g + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) pa[j & 3] = fmaf((float)hi[j], j < 4 ? w0[j] : w1[j - 4], pa[j & 3]);
            }
            s.red[wave * H1_ROWS + lane + 64 * rr] = (pa[0] + pa[1]) + (pa[2] + pa[3]);
        }
    }
    __syncthreads();
    if (tid < H1_ROWS) {
        float t = P.b8;
#pragma unroll
        for (int q = 0; q < 4; ++q) t += s.red[q * H1_ROWS + tid];
        float dy_;
        s.y[tid] = mlp_output(t, P.use_tanh, dy_);
    }
    __syncthreads();
}

}  // namespace qsp
