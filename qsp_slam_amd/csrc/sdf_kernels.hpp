// sdf_kernels.hpp -- the device code of path A (batched DeepSDF object refinement on gfx950): every kernel of the Gauss-Newton
// iteration that csrc/sdf_refine.hip launches, and the decode kernels.  Kept apart from the host side so that a one-kernel
// translation unit (tools/micro/, tests/test_isa_budget.py) can compile a single instantiation in seconds.
// Reference: reconstruct/optimizer.py:96-281, :47-93, reconstruct/loss.py:22-178, reconstruct/loss_utils.py:40-265.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "common.hpp"
#include "sdf_mlp.hpp"

namespace qsp {

constexpr int MAX_DEPTH = 64;
#ifndef QSP_NW_REND
#define QSP_NW_REND 16
#endif
#ifndef QSP_NW_SDF_MAX
#define QSP_NW_SDF_MAX 256      // surface slots per hypothesis: one 64-point tile per work item up to 16 k surface points
#endif
constexpr int NW_REND = QSP_NW_REND;     // render slots per hypothesis (work items looping over render-row tiles beyond 16 x 64 rows)
constexpr int NW_SDF_MAX = QSP_NW_SDF_MAX;  // work items per hypothesis looping over surface-point tiles
constexpr int NH = 71;          // 7 pose + 64 code unknowns
// One partial sum of J~^T J~ per (hypothesis, slot): the UPPER TRIANGLE of the 72 x 72 matrix, packed row by row (2628 floats,
// padded to a multiple of 32: 10.4 KB).  Rounds 1-3 stored the six 32 x 32 MFMA tiles of the padded 96 x 96 upper triangle whole
// (24 KB: 0.8 GB of writes per C4 launch, and k_solve pulled 1.9 MB per hypothesis through one compute unit).  Same values, same
// order of summation over the slots: same bits.
constexpr int PART_FLOATS = ((NJ * (NJ + 1) / 2 + 31) / 32) * 32;
__device__ __forceinline__ int tri72(int r, int c) { return r * NJ - (r * (r - 1)) / 2 + (c - r); }      // r <= c < NJ
// the accumulator tile (ta, tb) of a wave <-> its entries of the packed triangle (entries below the diagonal or beyond column 71
// are not stored: nobody reads them; a continued item restarts them from 0)
// (row r0 + o of the tile, o = acc_row(i, 0) a compile-time constant: tri72(r0 + o, c) = tri72(r0, c) + o (71 - r0) - o (o - 1) / 2 --
//  one multiply-add per element on two per-lane values; the f32 tile's kernel has no registers to spare in its epilogue)
__device__ __forceinline__ void part_store(float* __restrict__ slot, int ta, int tb, int lane, const f32x16& hacc) {
    const int r0 = 32 * ta + 4 * (lane >> 5), c = 32 * tb + (lane & 31);
    const int k = (NJ - 1) - r0, m = c < NJ ? c - r0 : -1;      // entry o is stored iff o <= m
    // ONE running index, stepped from row to row (idx(o + 1) - idx(o) = k - o) and pinned between the steps: sixteen independent
    // per-lane offsets cost the exact-f32 kernel 30 more spilled registers in an epilogue that has none to spare
    int idx = tri72(r0, c);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = (i & 3) + 8 * (i >> 2);
        if (o <= m) slot[idx] = hacc[i];
        if (i < 15) {
            const int o2 = ((i + 1) & 3) + 8 * ((i + 1) >> 2);      // next row: idx += sum_{q = o}^{o2 - 1} (k - q)
            idx += (o2 - o) * k - ((o2 - o) * (o + o2 - 1)) / 2;
            asm volatile("" : "+v"(idx));
        }
    }
}
__device__ __forceinline__ void part_load(const float* __restrict__ slot, int ta, int tb, int lane, bool first, f32x16& hacc) {
    const int r0 = 32 * ta + 4 * (lane >> 5), c = 32 * tb + (lane & 31);
    const int base = tri72(r0, c), k = (NJ - 1) - r0, m = c < NJ ? c - r0 : -1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = (i & 3) + 8 * (i >> 2);
        float v = 0.f;
        if (!first && o <= m) v = slot[base + o * k - (o * (o - 1)) / 2];
        hacc[i] = v;
    }
}

// per-hypothesis state, resident in HBM
struct HypState {
    float T_oc[16];     // camera -> object Sim3, row-major
    float code[CODE_LEN];
    float T_co[16];     // object -> camera (inverse), refreshed by k_sample
    float scale, d_min, d_max, loss;
    float loss_sdf, loss_render;
    int32_t alive;      // 1 while the reference would still be iterating
    int32_t n_valid;    // ray samples inside the unit ball
    int32_t n_render;   // render rows K
    int32_t obj;        // object index
    int32_t n_band;     // screened forward pass: ray samples whose screening value is within the band (k_mlp_fwd_h1)
    int32_t n_stage;    // depth-staged forward: samples on the current stage's list (k_stage_list)
    int32_t n_band_total, n_eval;   // (profile counters: band samples of the stages before this one; samples evaluated so far)
};

struct ObjView {            // per-object observation extents inside the concatenated arrays
    int64_t pts_off;        // in points
    int64_t ray_off;        // in rays
    int32_t n_pts, n_rays, n_fg, pad;
};

struct RefineCfg {
    float k1, k2, k3, k4, b1, b2, lr, s_damp, cut_off;
    int32_t n_depth;
    int32_t pose_only;      // estimate_pose_cam_obj mode
    int32_t iter;           // current iteration index (pose-only inlier filter)
    int32_t code_len;       // the decoder's code length L <= 64: code unknowns L..63 are padding (decoupled in k_solve)
    int32_t tile_p;         // points per MLP tile: 64, or 32 (QSP_DEC_OPT_TILE_POINTS, split-fp16 pipe only)
};

// ---------------------------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------------------------

// torch.linspace(d_min, d_max, D)[k] in f32 (two-sided form used by ATen's kernels)
__device__ __forceinline__ float depth_at(float d_min, float d_max, int k, int D) {
    const float step = (d_max - d_min) / (float)(D - 1);
    return (k < D / 2) ? d_min + step * (float)k : d_max - step * (float)(D - 1 - k);
}

__device__ __forceinline__ void xform(const float* T, float px, float py, float pz, float& x, float& y, float& z) {
    x = px * T[0] + py * T[1] + pz * T[2] + T[3];
    y = px * T[4] + py * T[5] + pz * T[6] + T[7];
    z = px * T[8] + py * T[9] + pz * T[10] + T[11];
}

// 4x4 inverse, Gauss-Jordan with partial pivoting in f64 from f32 input (reference: torch.inverse in f32)
__device__ void inv4(const float* A, float* Ainv) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = (double)A[4 * i + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); p = r; }
        if (p != c)
            for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) Ainv[4 * i + j] = (float)a[i][4 + j];
}

__device__ __forceinline__ float det3(const float* T) {   // of the upper-left 3x3 of a row-major 4x4
    const double a = T[0], b = T[1], c = T[2], d = T[4], e = T[5], f = T[6], g = T[8], h = T[9], i = T[10];
    return (float)(a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g));
}

// exclusive scan of one int per thread over the block (a multiple of 64 threads); returns the exclusive prefix, total in *total
__device__ int block_excl_scan_256(int v, int* smem /* >= one int per wave */, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) smem[wave] = x;
    __syncthreads();
    int base = 0, tot = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) {
        if (w < wave) base += smem[w];
        tot += smem[w];
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// Huber weight sqrt(rho(|r|))/|r| (loss_utils.py:236-247); |r| == 0 divides by 1
__device__ __forceinline__ float huber_w(float r, float b) {
    const float a = fabsf(r);
    const float rho = (a <= b) ? a * a : 2.f * b * a - b * b;
    return sqrtf(rho) / (a == 0.f ? 1.f : a);
}

// ---------------------------------------------------------------------------------------------------------------
// k_sample: per-hypothesis prologue + valid ray samples (loss.py:60-74, optimizer.py:144-153)
// ---------------------------------------------------------------------------------------------------------------
#ifndef QSP_PHASE_CLOCK
#define QSP_PHASE_CLOCK 0    // timing experiment only (tools/phase_clock.py): thread 0 of workgroup 0 of k_sample (0), k_scan (1) and k_solve
#endif                       // (2) stamps the constant 100 MHz counter at its marks; correct results; 0 in every build that ships
#if QSP_PHASE_CLOCK
__device__ unsigned long long qsp_phase_abs[3][16];      // the launch's stamps
__device__ unsigned long long qsp_phase_ticks[3][16];    // [kernel][i]: ticks between mark i-1 and mark i, summed over launches; [kernel][0]: launches
#define PHASE_MARK(kern, i)                                                                     \
    if (blockIdx.x == 0 && threadIdx.x == 0) qsp_phase_abs[kern][i] = __builtin_amdgcn_s_memrealtime();
#define PHASE_END(kern, n)                                                                      \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                                  \
        qsp_phase_abs[kern][n] = __builtin_amdgcn_s_memrealtime();                              \
        for (int i_ = 1; i_ <= (n); ++i_) qsp_phase_ticks[kern][i_] += qsp_phase_abs[kern][i_] - qsp_phase_abs[kern][i_ - 1]; \
        qsp_phase_ticks[kern][0] += 1;                                                          \
    }
#else
#define PHASE_MARK(kern, i)
#define PHASE_END(kern, n)
#endif
constexpr int SAMPLE_THREADS = 1024;      // k_sample's workgroup: latency, not throughput (a one-object call is ONE workgroup)
// (Measured in round 4 and not kept: a hypothesis's rays over up to 8 workgroups, the one that finishes last writing the lists from
//  what the others left in global memory -- k_sample 25.7 -> 23.5 us, k_scan 26.2 -> 30.6 us per launch of a one-object call, the
//  call unchanged at 2.52 ms: a wave walks its 64 rays in the same 11 us whether seven others walk beside it or not, and the
//  hand-over between workgroups -- fence, counter, fence, reads that miss -- costs the 3-5 us the shorter mask loop saves.)
__device__ __forceinline__ void sample_body(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                            const float* __restrict__ rays, const RefineCfg& cfg,
                                            int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                            int32_t* __restrict__ ray_voff, int64_t ray_stride,
                                            const MlpParams* __restrict__ Pm, float* __restrict__ c0_all) {
    const int h = blockIdx.x;
    HypState& S = st[h];
    if (!S.alive) return;
    __shared__ float T[16];
    __shared__ float dm[2];
    __shared__ int sc[SAMPLE_THREADS / 64];
    __shared__ float code_sh[CODE_LEN];
    __shared__ uint64_t mask_sh[SAMPLE_THREADS];     // one pass's rays: bit k = depth sample k is inside the unit ball
    __shared__ int off_sh[SAMPLE_THREADS];           // and where each ray's entries start in the pass's part of the list
    if (threadIdx.x < CODE_LEN) code_sh[threadIdx.x] = S.code[threadIdx.x];
    __syncthreads();
    PHASE_MARK(0, 1);
    // the per-hypothesis bias vectors of layers 0 and 4 (k_c0's arithmetic, folded in here: one launch less per iteration; the other
    // waves work on it while thread 0 inverts the pose)
    {   // (code_bias's sums, layer 0's on the first 512 threads and layer 4's on the other 512)
        static_assert(SAMPLE_THREADS == 2 * HID, "one bias entry per thread");
        const int u = threadIdx.x & (HID - 1);
        const bool l4 = threadIdx.x >= HID;
        const float* w = (l4 ? Pm->w4c : Pm->w0c) + u;
        float a = Pm->bias[l4 ? 4 : 0][u];
#pragma unroll
        for (int k = 0; k < CODE_LEN; ++k) a += w[(size_t)k * HID] * code_sh[k];      // (all 64 loads in flight: one round trip, not eight)
        c0_all[(size_t)h * 2 * HID + threadIdx.x] = a;
    }
    PHASE_MARK(0, 2);
    if (threadIdx.x == 0) {
        float Tco[16];
        inv4(S.T_oc, Tco);
        const float scale = powf(det3(Tco), (float)(1.0 / 3.0));
        for (int i = 0; i < 16; ++i) S.T_co[i] = Tco[i];
        S.scale = scale;
        S.d_min = Tco[11] - 1.0f * scale;
        S.d_max = Tco[11] + 1.0f * scale;
        dm[0] = S.d_min;
        dm[1] = S.d_max;
        for (int i = 0; i < 16; ++i) T[i] = S.T_oc[i];
    }
    __syncthreads();
    PHASE_MARK(0, 3);
    const ObjView ov = objs[S.obj];
    const int D = cfg.n_depth;
    const float* R = rays + 3 * ov.ray_off;
    int32_t* rk = valid_rk + h * rk_stride;
    int32_t* voff = ray_voff + h * ray_stride;
    // A wave per ray, a lane per depth sample: the ray's mask of samples inside the unit ball is one ballot, its list entries one
    // store of consecutive words.  (A thread per ray walked its 64 samples one after the other with one wave on each SIMD, and
    // stored its entries a line per lane: 25 of this kernel's 41 us on a one-object call.)  Same expressions per sample.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = SAMPLE_THREADS / 64;
    const float d = depth_at(dm[0], dm[1], lane, D);
    int carry = 0;
    for (int base = 0; base < ov.n_rays; base += SAMPLE_THREADS) {
        const int n_pass = min(SAMPLE_THREADS, ov.n_rays - base);
        // (lane i fetches the ray of its wave's i-th turn: three loads in flight per lane once, instead of a load's latency per turn)
        const int mine = wave + NW * lane;
        float fx = 0.f, fy = 0.f, fz = 0.f;
        if (mine < n_pass) fx = R[3 * (base + mine)], fy = R[3 * (base + mine) + 1], fz = R[3 * (base + mine) + 2];
        for (int rl = wave, turn = 0; rl < n_pass; rl += NW, ++turn) {
            const float rx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fx), turn));
            const float ry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fy), turn));
            const float rz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fz), turn));
            float x, y, z;
            xform(T, rx * d, ry * d, rz * d, x, y, z);
            const uint64_t m = __ballot(lane < D && sqrtf(x * x + y * y + z * z) < 1.0f);
            if (lane == 0) mask_sh[rl] = m;
        }
        __syncthreads();
        PHASE_MARK(0, 4);
        const int cnt = (threadIdx.x < n_pass) ? __popcll(mask_sh[threadIdx.x]) : 0;
        int tot;
        const int ex = block_excl_scan_256(cnt, sc, &tot);
        if (threadIdx.x < n_pass) {
            voff[base + threadIdx.x] = carry + ex;
            off_sh[threadIdx.x] = ex;
        }
        __syncthreads();
        PHASE_MARK(0, 5);
        for (int rl = wave; rl < n_pass; rl += NW) {
            const uint64_t m = mask_sh[rl];
            if (m >> lane & 1ull) rk[carry + off_sh[rl] + __popcll(m & ((1ull << lane) - 1ull))] = ((base + rl) << 6) | lane;
        }
        carry += tot;
        __syncthreads();        // (mask_sh and off_sh are the next pass's)
    }
    PHASE_MARK(0, 6);
    if (threadIdx.x == 0) {
        voff[ov.n_rays] = carry;
        S.n_valid = carry;
        S.n_render = 0;
        S.n_band = 0;
        S.n_stage = 0;
        S.n_band_total = 0;
        S.n_eval = carry;        // (every valid sample, unless the forward pass is depth-staged: k_stage_list counts then)
        if (!cfg.pose_only && carry < 10) S.alive = 0;   // loss.py:73-74 -> optimizer.py:171-172
    }
}

// The work-queue plan of the kernel that FOLLOWS (k_plan's arithmetic) in the tail of its producer: every workgroup of k_sample /
// k_scan counts itself done behind its last store to the hypothesis state, the one that finishes last builds the item list.  One
// launch and one kernel boundary less in front of each of the two MLP kernels (~9 us each, 10 per one-object call: round 3's
// one-object call spent 0.1 of its 2.8 ms there).  done[] is reset by its last reader.
struct PlanTail {
    int2* work;            // nullptr: no tail (the plan is a launch of its own)
    int* qctl;
    int* done;
    int n_hyp, nw_sdf, nw_rend, tile_p, mode;
};
template <int NT>
__device__ __forceinline__ void plan_body(int mode, const HypState* __restrict__ st, const ObjView* __restrict__ objs, int n_hyp, int nw_sdf,
                                          int nw_rend, int2* __restrict__ work, int* __restrict__ qctl, int tile_p);
template <int NT>
__device__ __forceinline__ void plan_tail(const PlanTail& pt, const HypState* st, const ObjView* objs) {
    if (!pt.work) return;
    __shared__ int s_last;
    __syncthreads();                                   // (thread 0's stores to the hypothesis state are behind us)
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(pt.done, 1) == pt.n_hyp - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                   // (acquire: the other workgroups' n_valid / n_render / alive)
    plan_body<NT>(pt.mode, st, objs, pt.n_hyp, pt.nw_sdf, pt.nw_rend, pt.work, pt.qctl, pt.tile_p);
    if (threadIdx.x == 0) *pt.done = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// staging of one tile's inputs
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_code_T(MlpSmem& s, const HypState& S, float* Tsh) {
    if (threadIdx.x < CODE_LEN) s.code[threadIdx.x] = S.code[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 80) Tsh[threadIdx.x - 64] = S.T_oc[threadIdx.x - 64];
}

// ---------------------------------------------------------------------------------------------------------------
// Work queues of the two MLP kernels.
// Both kernels are launched as ONE workgroup per CU and pull (hypothesis, slot) items from a list through an atomic
// counter until it is exhausted.  Why not a (slots, hypotheses) grid: workgroup ids are dealt round-robin to the 8 XCDs, a
// grid row holds slots with and without work (render slots beyond K, the ragged last surface slot), and for most row
// lengths the working slots of every hypothesis fall on the same XCDs -- measured 6-40 % of the chip idle depending on
// (slots mod 8).  A compacted list has no empty items, so whichever CU is free takes the next one.
// Partial sums stay addressed by the LOGICAL (hypothesis, slot), so results do not depend on who processed what.
//   qctl[0] = #items forward, qctl[1] = next forward item, qctl[2] = #items jtj, qctl[3] = next jtj item
// ---------------------------------------------------------------------------------------------------------------
// c0[h][u] = b0[u] + sum_k W0[u][k] code_h[k]: layer 0 without its xyz columns (see mlp_prepare), once per hypothesis
__global__ __launch_bounds__(MLP_THREADS) void k_c0(const HypState* __restrict__ st, const MlpParams* __restrict__ Pm,
                                                    float* __restrict__ c0_all) {
    __shared__ float code[CODE_LEN];
    const HypState& S = st[blockIdx.x];
    if (!S.alive) return;
    if (threadIdx.x < CODE_LEN) code[threadIdx.x] = S.code[threadIdx.x];
    __syncthreads();
    const int u = threadIdx.x;
    float a, a4;
    code_bias(Pm, u, code, a, a4);
    c0_all[(size_t)blockIdx.x * 2 * HID + u] = a;
    c0_all[(size_t)blockIdx.x * 2 * HID + HID + u] = a4;     // layer 4's bias with the skip connection's code part
}

// mode 0: forward items (h, tile) over the valid ray samples; mode 1: jtj items (h, slot), surface slots then render slots;
// mode 2: forward items (h, tile) over the hypothesis's band list (screened forward pass); mode 3: over its stage list (k_stage_list)
template <int NT>
__device__ __forceinline__ void plan_body(int mode, const HypState* __restrict__ st, const ObjView* __restrict__ objs, int n_hyp, int nw_sdf,
                                          int nw_rend, int2* __restrict__ work, int* __restrict__ qctl, int tile_p) {
    __shared__ int wsum[NT / 64];
    __shared__ int carry_sh;
    __shared__ int off_s[NT], na_s[NT], nb_s[NT];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry_sh = 0;
    __syncthreads();
    for (int base = 0; base < n_hyp; base += NT) {
        const int h = base + t;
        int n_a = 0, n_b = 0;
        if (h < n_hyp && st[h].alive) {
            if (mode == 0) n_a = (st[h].n_valid + tile_p - 1) / tile_p;
            else if (mode == 2) n_a = (st[h].n_band + tile_p - 1) / tile_p;      // second pass of the screened forward
            else if (mode == 3) n_a = (st[h].n_stage + tile_p - 1) / tile_p;     // one depth stage of the screened forward
            else {
                n_a = min(nw_sdf, (objs[st[h].obj].n_pts + tile_p - 1) / tile_p);
                n_b = min(nw_rend, (st[h].n_render + tile_p - 1) / tile_p);
            }
        }
        const int cnt = n_a + n_b;
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = carry_sh + inc - cnt;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        // the items of a hypothesis are written by a wave, 64 consecutive words per store (its own thread wrote them one by one:
        // ~400 stores in a row when the launch holds ONE hypothesis, most of the 10 us this tail took in k_sample)
        off_s[t] = off, na_s[t] = n_a, nb_s[t] = n_b;
        __syncthreads();
        const int n_here = min(NT, n_hyp - base);
        for (int i = wave; i < n_here; i += NT / 64) {
            const int o = off_s[i], a = na_s[i], b = nb_s[i];
            for (int j = lane; j < a + b; j += 64) work[o + j] = make_int2(base + i, j < a ? j : nw_sdf + (j - a));
        }
        if (t == NT - 1) carry_sh = off + cnt;
        __syncthreads();
    }
    if (t == 0) {
        const int q = mode >= 2 ? 0 : mode;      // (the band pass and the depth stages reuse the forward queue's control words and item list)
        qctl[2 * q] = carry_sh;
        qctl[2 * q + 1] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Depth-staged ray-sample forward (round 4).  The render term multiplies everything behind the first OPAQUE sample of a ray by
// an exact zero: occ = 0.5 - clamp(s, -th, th) / (2 th) is exactly 1 for s <= -th, the transmittance is a cumulative product of
// (1 - occ) (loss.py:101), so it is exactly 0 from that sample on -- every later term of the rendered depth is 0 x occ, the extra
// far bin 0, and d e / d o of a later in-band sample is 0 / (1 - o_k) = 0, below the 1e-2 keep threshold (loss.py:103-128).  The
// decoder values of those samples reach no output (K, H, b, the loss; n_valid is geometry).  So the samples are evaluated in two
// stages along the ray: depth indices [0, D/2) of every ray, then [D/2, D) of the rays that have no opaque sample yet -- a ray
// that hits the object is opaque before the centre plane of its depth range (the range is centred on the object).  Skipped
// samples get a benign finite value (+1: outside, occupancy 0).  On the synthetic scenes 21 % of the valid samples are skipped
// (C4; 19 % at C5); results are bit-identical to the unstaged pass (tests/test_gpu_screening.py).  The reference would differ
// only for a decoder that returns NaN inside the unit ball behind a surface (its NaN would poison the ray; here it is never
// computed).
// One workgroup per hypothesis; the ray's samples sit at [voff[r], voff[r+1]) of the valid list in ascending depth index.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stage_list(HypState* __restrict__ st, const ObjView* __restrict__ objs, RefineCfg cfg,
                                                    const int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                    const int32_t* __restrict__ ray_voff, int64_t ray_stride,
                                                    float* __restrict__ sdf_valid, uint8_t* __restrict__ ray_open,
                                                    int32_t* __restrict__ stage_idx, int k_prev, int k_lo, int k_hi, PlanTail pt) {
    const int h = blockIdx.x;
    HypState& S = st[h];
    __shared__ int sc[8];
    if (S.alive) {
        const ObjView ov = objs[S.obj];
        const int32_t* rk = valid_rk + h * rk_stride;
        const int32_t* voff = ray_voff + h * ray_stride;
        float* sdf = sdf_valid + h * rk_stride;
        uint8_t* open_r = ray_open + h * ray_stride;
        int32_t* out = stage_idx + h * rk_stride;
        const float th = cfg.cut_off;
        int carry = 0;
        for (int base = 0; base < ov.n_rays; base += 256) {
            const int r = base + threadIdx.x;
            int cnt = 0, v_lo = 0;
            if (r < ov.n_rays) {
                const int v0 = voff[r], v1 = voff[r + 1];
                bool open = k_lo == 0 ? true : open_r[r] != 0;
                int v = v0;
                if (open && k_lo > 0) {                     // did the stage before this one end the ray?
                    for (; v < v1; ++v) {
                        const int k = rk[v] & 63;
                        if (k >= k_lo) break;
                        if (k >= k_prev && sdf[v] <= -th) open = false;      // occupancy exactly 1 (a NaN keeps the ray open)
                    }
                    if (!open)
                        for (int q = v; q < v1; ++q) sdf[q] = 1.0f;           // never evaluated: any finite value (occupancy 0)
                } else {
                    while (v < v1 && (rk[v] & 63) < k_lo) ++v;
                }
                open_r[r] = open ? 1 : 0;
                v_lo = v;
                if (open)
                    while (v < v1 && (rk[v] & 63) < k_hi) { ++v; ++cnt; }
            }
            int tot;
            const int ex = block_excl_scan_256(cnt, sc, &tot);
            for (int q = 0; q < cnt; ++q) out[carry + ex + q] = v_lo + q;
            carry += tot;
        }
        if (threadIdx.x == 0) {
            S.n_band_total += S.n_band;                     // (the band list restarts with every stage)
            S.n_band = 0;
            S.n_eval = (k_lo == 0 ? 0 : S.n_eval) + carry;
            S.n_stage = carry;
        }
    }
    plan_tail<256>(pt, st, objs);                          // this stage's tiles (k_plan mode 3)
}

__global__ __launch_bounds__(1024) void k_plan(int mode, const HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                               int n_hyp, int nw_sdf, int nw_rend, int2* __restrict__ work, int* __restrict__ qctl,
                                               int tile_p) {
    plan_body<1024>(mode, st, objs, n_hyp, nw_sdf, nw_rend, work, qctl, tile_p);
}
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                                const float* __restrict__ rays, RefineCfg cfg,
                                                int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                int32_t* __restrict__ ray_voff, int64_t ray_stride,
                                                const MlpParams* __restrict__ Pm, float* __restrict__ c0_all, PlanTail pt) {
    PHASE_MARK(0, 0);
    sample_body(st, objs, rays, cfg, valid_rk, rk_stride, ray_voff, ray_stride, Pm, c0_all);
    PHASE_MARK(0, 7);
    plan_tail<SAMPLE_THREADS>(pt, st, objs);        // the forward kernel's item list (k_plan mode 0)
    PHASE_END(0, 8);
}

// ---------------------------------------------------------------------------------------------------------------
// k_mlp_fwd: decoder forward on the valid ray samples (loss.py:78)
// ---------------------------------------------------------------------------------------------------------------
template <bool BF3>
__global__ __launch_bounds__(MLP_THREADS, 2) void k_mlp_fwd(const HypState* __restrict__ st,
                                                            const ObjView* __restrict__ objs,
                                                            const float* __restrict__ rays, RefineCfg cfg, const MlpParams* __restrict__ P,
                                                            const int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                            float* __restrict__ sdf_valid, const int2* __restrict__ work,
                                                            int* __restrict__ qctl, const float* __restrict__ c0_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    __shared__ float Tsh[16];
    __shared__ int s_item;
    const int n_items = qctl[0];
    int h_cached = -1;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&qctl[1], 1);
        __syncthreads();                       // also: everybody is done with the previous item's LDS
        const int item = s_item;
        if (item >= n_items) break;            // the queue only grows towards n_items: every workgroup gets here
        const int h = work[item].x, t = work[item].y;
        const HypState& S = st[h];
        const int n = S.n_valid;
        const ObjView ov = objs[S.obj];
        const float* R = rays + 3 * ov.ray_off;
        const int32_t* rk = valid_rk + h * rk_stride;
        float* out = sdf_valid + h * rk_stride;
        if (h != h_cached) {                   // per-hypothesis staging: code, pose, layer-0 code part
            stage_code_T(s, S, Tsh);
            s.c0[threadIdx.x] = c0_all[(size_t)h * 2 * HID + threadIdx.x];
            s.c4[threadIdx.x] = c0_all[(size_t)h * 2 * HID + HID + threadIdx.x];
            h_cached = h;
        }
        const float d_min = S.d_min, d_max = S.d_max;
        __syncthreads();
        if (threadIdx.x < TILE_P) {
            const int v = t * TILE_P + threadIdx.x;
            float x = 0, y = 0, z = 0;
            if (v < n) {
                const int e = rk[v];
                const int r = e >> 6, k = e & 63;
                const float d = depth_at(d_min, d_max, k, cfg.n_depth);
                xform(Tsh, R[3 * r] * d, R[3 * r + 1] * d, R[3 * r + 2] * d, x, y, z);
            }
            s.xin[4 * threadIdx.x + 0] = x;
            s.xin[4 * threadIdx.x + 1] = y;
            s.xin[4 * threadIdx.x + 2] = z;
            s.xin[4 * threadIdx.x + 3] = 0.f;
        }
        __syncthreads();
        if (BF3) mlp_tile_bf3<QSP_BF3_PF>(s, P);
        else mlp_tile<false, 4>(s, P);
        if (threadIdx.x < TILE_P) {
            const int v = t * TILE_P + threadIdx.x;
            if (v < n) out[v] = s.y[threadIdx.x];
        }
    }
}

// The out-of-band AUDIT of the screened forward (QSP_DEC_OPT_SCREEN_AUDIT).  The band pass only ever sees the
// samples the screening pass put INSIDE the band; a sample it put outside (|s1| >= cut_off + margin) whose true value is inside the
// cut-off would be clamped wrongly and never looked at.  So the screening pass also lists a fixed pseudo-random one-in-N of the
// OUT-of-band samples (a hash of hypothesis, sample and iteration: the same samples on every run of the same state, other samples
// in the next iteration), flagged with bit 30; the band pass re-evaluates them with the rest, folds their |s1 - s3| into the same
// maximum, and counts every one whose split-fp16 value is inside the cut-off or has the other sign (a clamp the one-pass result
// would not have made) as a HARD failure: the host repeats the run in one pass.  Overwriting an audited sample's value with s3
// changes no bit downstream: k_scan reads such a value through clamp() and the band test only.
constexpr int32_t BAND_AUDIT_BIT = 1 << 30;
__device__ __forceinline__ bool screen_audit_pick(int h, int v, int iter, int one_in) {
    if (one_in <= 1) return one_in == 1;
    unsigned x = (unsigned)v * 0x9E3779B1u ^ (unsigned)h * 0x85EBCA77u ^ (unsigned)iter * 0xC2B2AE3Du;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return x % (unsigned)one_in == 0u;
}

// the same work queue on the split-fp16 tile: four waves per workgroup (mlp_tile_h2)
// NARROW: mlp_tile_h2's form for small decoders, run with NW = 8 waves so that the column blocks that exist spread over all SIMDs
template <int NR, bool NARROW = false, int NW = 4>      // NR point blocks of 32 per tile (QSP_DEC_OPT_TPOINTS)
__global__ __launch_bounds__(64 * NW) void k_mlp_fwd_h2(const HypState* __restrict__ st,
                                                            const ObjView* __restrict__ objs,
                                                            const float* __restrict__ rays, RefineCfg cfg, const MlpParams* __restrict__ P,
                                                            const int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                            float* __restrict__ sdf_valid, const int2* __restrict__ work,
                                                            int* __restrict__ qctl, const float* __restrict__ c0_all,
                                                            const int32_t* __restrict__ band_idx, unsigned int* __restrict__ screen_dmax) {
    // band_idx != nullptr: second pass of the screened forward -- the tiles run over the hypothesis's band list (indices into
    // its valid-sample list written by k_mlp_fwd_h1) and overwrite those samples' screening values.  On the way the largest
    // |s1 - s3| over the band samples is kept (*screen_dmax, the bits of a non-negative float): the quantity the screening margin
    // has to cover, measured on every run -- the host repeats a run unscreened if it ever comes near the margin.  Entries flagged
    // BAND_AUDIT_BIT are out-of-band samples under audit (above): screen_dmax[1] counts those the screening pass clamped wrongly,
    // (unsigned long long*)(screen_dmax + 2) how many were audited.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    __shared__ float Tsh[16];
    __shared__ int s_item;
    const int n_items = qctl[0];
    constexpr int TP = 32 * NR;
    bool staged = false;
    float amax = 0.f, dmax = 0.f;
    int h_cached = -1;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&qctl[1], 1);
        __syncthreads();                       // also: everybody is done with the previous item's LDS
        const int item = s_item;
        if (item >= n_items) break;            // the queue only grows towards n_items: every workgroup gets here
        const int h = work[item].x, t = work[item].y;
        const HypState& S = st[h];
        const int n = band_idx ? S.n_band : S.n_valid;
        const ObjView ov = objs[S.obj];
        const float* R = rays + 3 * ov.ray_off;
        const int32_t* rk = valid_rk + h * rk_stride;
        const int32_t* sel = band_idx ? band_idx + h * rk_stride : nullptr;
        float* out = sdf_valid + h * rk_stride;
        if (h != h_cached) {                   // per-hypothesis staging: code, pose, layer-0 code part
            stage_code_T(s, S, Tsh);
            for (int i = threadIdx.x; i < HID; i += 64 * NW) {
                s.c0[i] = c0_all[(size_t)h * 2 * HID + i];
                s.c4[i] = c0_all[(size_t)h * 2 * HID + HID + i];
            }
            h_cached = h;
        }
        const float d_min = S.d_min, d_max = S.d_max;
        __syncthreads();
        if (threadIdx.x < TP) {
            const int v = t * TP + threadIdx.x;
            float x = 0, y = 0, z = 0;
            if (v < n) {
                const int e = rk[sel ? (sel[v] & ~BAND_AUDIT_BIT) : v];
                const int r = e >> 6, k = e & 63;
                const float d = depth_at(d_min, d_max, k, cfg.n_depth);
                xform(Tsh, R[3 * r] * d, R[3 * r + 1] * d, R[3 * r + 2] * d, x, y, z);
            }
            s.xin[4 * threadIdx.x + 0] = x;
            s.xin[4 * threadIdx.x + 1] = y;
            s.xin[4 * threadIdx.x + 2] = z;
            s.xin[4 * threadIdx.x + 3] = 0.f;
        }
        __syncthreads();
        mlp_tile_h2<false, 2, !NARROW, NR, NW, NARROW>(s, P, amax, !staged);      // (the decoder's constants: staged by the first tile of the workgroup)
        staged = true;
        if (threadIdx.x < TP) {
            const int v = t * TP + threadIdx.x;
            bool audited = false, wrong = false;
            if (v < n) {
                const int ent = sel ? sel[v] : v;
                const int idx = ent & ~BAND_AUDIT_BIT;
                const float y = s.y[threadIdx.x];
                if (sel) {
                    const float s1 = out[idx];                         // (out[idx] still holds the screening value s1)
                    dmax = fmaxf(dmax, fabsf(s1 - y));
                    audited = (ent & BAND_AUDIT_BIT) != 0;
                    wrong = audited && (!(fabsf(y) >= cfg.cut_off) || (s1 < 0.f) != (y < 0.f));
                }
                out[idx] = y;
            }
            if (sel && screen_dmax && threadIdx.x < 64) {              // (counted per tile: nothing stays live across the tile loop)
                const unsigned long long ma = __ballot(audited), mw = __ballot(wrong);
                if (threadIdx.x == 0 && ma) atomicAdd(reinterpret_cast<unsigned long long*>(screen_dmax + 2), (unsigned long long)__popcll(ma));
                if (threadIdx.x == 0 && mw) atomicAdd(screen_dmax + 1, (unsigned int)__popcll(mw));
            }
        }
    }
    if (band_idx && screen_dmax && threadIdx.x < 64) {      // the first wave holds every row's difference (TP <= 64)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
        if (threadIdx.x == 0 && dmax > 0.f) atomicMax(screen_dmax, __float_as_uint(dmax));
    }
    if (!(amax <= H2_MAX)) *P->range_flag = 1;
}

// First pass of the screened forward (mlp_tile_h1, sdf_mlp.hpp): every valid ray sample on the one-product tile, 128 points per
// work item; writes the screening value and appends the samples with |s1| < band_th (or NaN) to the hypothesis's band list.
// The order of a band list depends on which workgroup finished first; nothing downstream does: the second pass writes each
// listed sample's value to its own slot, and a sample's value does not depend on its position in a tile.
template <int NW>      // waves per workgroup: 4 x 512 registers or 8 x 256 (mlp_tile_h1)
__global__ __launch_bounds__(64 * NW) void k_mlp_fwd_h1(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                                            const float* __restrict__ rays, RefineCfg cfg, const MlpParams* __restrict__ P,
                                                            const int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                            float* __restrict__ sdf_valid, const int2* __restrict__ work,
                                                            int* __restrict__ qctl, const float* __restrict__ c0_all,
                                                            int32_t* __restrict__ band_idx, float band_th, int audit_one_in,
                                                            const int32_t* __restrict__ stage_idx) {
    // stage_idx != nullptr: the tiles run over the hypothesis's stage list (positions in its valid-sample list, k_stage_list)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmemH1& s = *reinterpret_cast<MlpSmemH1*>(smem_raw);
    __shared__ float Tsh[16];
    __shared__ int s_item;
    const int n_items = qctl[0];
    float amax = 0.f;
    int h_cached = -1;
    {   // the decoder's constants, once per workgroup
        const MlpParams& Pp = *P;
        for (int i = threadIdx.x; i < HID; i += 64 * NW) s.w8[i] = Pp.w8[i];
#pragma unroll
        for (int l = 1; l < 8; ++l)
            for (int i = threadIdx.x; i < HID; i += 64 * NW) s.bias[(l - 1) * HID + i] = Pp.bias[l][i];
    }
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&qctl[1], 1);
        __syncthreads();                       // also: everybody is done with the previous item's LDS
        const int item = s_item;
        if (item >= n_items) break;            // the queue only grows towards n_items: every workgroup gets here
        const int h = work[item].x, t = work[item].y;
        HypState& S = st[h];
        const int n = stage_idx ? S.n_stage : S.n_valid;
        const int32_t* stg = stage_idx ? stage_idx + h * rk_stride : nullptr;
        const ObjView ov = objs[S.obj];
        const float* R = rays + 3 * ov.ray_off;
        const int32_t* rk = valid_rk + h * rk_stride;
        float* out = sdf_valid + h * rk_stride;
        if (h != h_cached) {                   // per-hypothesis staging: pose, code parts of layers 0 and 4
            if (threadIdx.x >= 64 && threadIdx.x < 80) Tsh[threadIdx.x - 64] = S.T_oc[threadIdx.x - 64];
            for (int i = threadIdx.x; i < HID; i += 64 * NW) {
                s.c0[i] = c0_all[(size_t)h * 2 * HID + i];
                s.c4[i] = c0_all[(size_t)h * 2 * HID + HID + i];
            }
            h_cached = h;
        }
        const float d_min = S.d_min, d_max = S.d_max;
        __syncthreads();
        if (threadIdx.x < H1_ROWS) {
            const int v = t * H1_ROWS + threadIdx.x;
            float x = 0, y = 0, z = 0;
            if (v < n) {
                const int e = rk[stg ? stg[v] : v];
                const int r = e >> 6, k = e & 63;
                const float d = depth_at(d_min, d_max, k, cfg.n_depth);
                xform(Tsh, R[3 * r] * d, R[3 * r + 1] * d, R[3 * r + 2] * d, x, y, z);
            }
            s.xin[4 * threadIdx.x + 0] = x;
            s.xin[4 * threadIdx.x + 1] = y;
            s.xin[4 * threadIdx.x + 2] = z;
            s.xin[4 * threadIdx.x + 3] = 0.f;
        }
        __syncthreads();
        mlp_tile_h1<2, NW>(s, P, amax);
        if (threadIdx.x < H1_ROWS) {           // (waves 0 and 1, all lanes)
            int v = t * H1_ROWS + threadIdx.x;
            bool in = false, audit = false;
            if (v < n) {
                if (stg) v = stg[v];           // (from here on v is the sample's position in the valid list)
                const float y = s.y[threadIdx.x];
                out[v] = y;
                in = !(fabsf(y) >= band_th);   // (a NaN goes to the second pass as well)
                audit = !in && screen_audit_pick(h, v, cfg.iter, audit_one_in);      // a sample of the OUT-of-band ones (above)
            }
            const unsigned long long m = __ballot(in || audit);
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == 0 && m) base = atomicAdd(&S.n_band, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (in || audit) band_idx[h * rk_stride + base + __popcll(m & ((1ull << lane) - 1ull))] = audit ? (v | BAND_AUDIT_BIT) : v;
        }
    }
    if (!(amax <= H2_MAX)) *P->range_flag = 1;
}

// ---------------------------------------------------------------------------------------------------------------
// k_scan: per ray render function and its derivative (loss.py:84-141)
// ---------------------------------------------------------------------------------------------------------------
struct RayScan {
    float d_u;
    int n_emit;
};

// One ray: `row` holds the SDF value of its depth sample k at row[k] (SCAN_NONE where the sample is outside the unit ball), staged
// in LDS by the workgroup (k_scan).  Every per-sample array is indexed by the unrolled loop counter only, so it lives in
// registers.  (Indexing them by the sample's k made them scratch memory, and reading the samples through a cursor made every
// load wait for the one before: 110 us per launch for 456 rays.)  Same operations in the same order as the reference's rows.
// Returns the number of kept rows n; their d e / d s go to row[0..n) (every sample has been read by then), their depth indices to
// the bits of `kept`, the ray's clamped residual to `res_out`: the caller writes them out once it knows where (one walk per ray --
// it used to walk every ray twice, once to count and once to write: 23 of k_scan's 39 us on a one-object call).
constexpr float SCAN_NONE = 1e30f;
constexpr int SCAN_RAYS = 512;                 // rays per pass = threads of k_scan
constexpr int SCAN_LD = MAX_DEPTH + 1;         // row stride in LDS: odd, so that the threads of a wave hit different banks
__device__ __forceinline__ int scan_ray(float* row, int D, float d_min, float d_max, float th, float depth_obs, uint64_t& kept,
                                        float& res_out) {
    float occ[MAX_DEPTH];          // occupancy row (zeros outside the unit ball)
    float Tl[MAX_DEPTH];           // transmittance T_l = prod_{j<=l} (1 - occ_j), then its suffix sums (loss.py:99-113)
    uint64_t inband = 0;           // bit k: a valid sample with |sdf| < th
    float acc = 1.f, d_u = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_DEPTH; ++k) {
        if (k < D) {
            const float s = row[k];
            float o = 0.f;
            if (s < 0.5f * SCAN_NONE) {
                const float c = fminf(fmaxf(s, -th), th);
                o = 0.5f - c / (2.f * th);
                if (s > -th && s < th) inband |= 1ull << k;
            }
            occ[k] = o;
            const float d = depth_at(d_min, d_max, k, D);
            d_u += d * (o * acc);
            acc *= (1.f - o);
            Tl[k] = acc;
        }
    }
    d_u += (1.1f * d_max) * acc;                       // the extra far bin
    float res = depth_obs - d_u;
    res = fminf(fmaxf(res, -0.30f), 0.30f);
    const float delta_d = (d_max - d_min) / (float)(D - 1);
    const float do_ds = -1.f / (2.f * th);
    float ssum = 0.f;                                  // suffix sums of T, walked from the far end
#pragma unroll
    for (int k = MAX_DEPTH - 1; k >= 0; --k)
        if (k < D) { ssum += Tl[k]; Tl[k] = ssum; }
    int n = 0;
    uint64_t km = 0;
#pragma unroll
    for (int k = 0; k < MAX_DEPTH; ++k) {              // kept rows in ascending k
        if (k < D && ((inband >> k) & 1ull)) {
            const float de_do = Tl[k] / (1.f - occ[k]);
            if (de_do > 1e-2f) {
                row[n] = de_do * delta_d * do_ds;
                km |= 1ull << k;
                ++n;
            }
        }
    }
    kept = km;
    res_out = res;
    return n;
}
__device__ __forceinline__ void scan_body(float* __restrict__ rows /*[SCAN_RAYS][SCAN_LD]*/, HypState* __restrict__ st,
                                          const ObjView* __restrict__ objs, const float* __restrict__ depth, const RefineCfg& cfg,
                                          const int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                          const int32_t* __restrict__ ray_voff, int64_t ray_stride,
                                          const float* __restrict__ sdf_valid, int32_t* __restrict__ rend_rk,
                                          float* __restrict__ rend_deds, float* __restrict__ rend_res) {
    const int h = blockIdx.x;
    HypState& S = st[h];
    if (!S.alive) return;
    __shared__ int sc[8];
    const ObjView ov = objs[S.obj];
    const int D = cfg.n_depth;
    const float* dep = depth + ov.ray_off;   // depth array is stored per ray (fg entries valid)
    const int32_t* rk = valid_rk + h * rk_stride;
    const int32_t* voff = ray_voff + h * ray_stride;
    const float* sdf = sdf_valid + h * rk_stride;
    int32_t* e_rk = rend_rk + h * rk_stride;
    float* e_deds = rend_deds + h * rk_stride;
    float* e_res = rend_res + h * rk_stride;
    const float d_min = S.d_min, d_max = S.d_max;
    int carry = 0;
    for (int base = 0; base < ov.n_rays; base += SCAN_RAYS) {        // one ray per thread: 456 rays in one pass
        // the pass's samples into a dense [ray][k] table: coalesced reads of the (ray, k)-sorted lists, one table row per thread
        if (base > 0) {      // (the first pass's table is cleared by k_scan, in front of the loads this function starts with)
            __syncthreads();
            for (int e = threadIdx.x; e < SCAN_RAYS * SCAN_LD; e += SCAN_RAYS) rows[e] = SCAN_NONE;
        }
        __syncthreads();
        PHASE_MARK(1, 1);
        const int r_end = min(base + SCAN_RAYS, ov.n_rays);
        const int v_beg = voff[base], v_end = voff[r_end];
        for (int v = v_beg + threadIdx.x; v < v_end; v += SCAN_RAYS) {
            const int e = rk[v];
            rows[((e >> 6) - base) * SCAN_LD + (e & 63)] = sdf[v];
        }
        __syncthreads();
        PHASE_MARK(1, 2);
        const int r = base + threadIdx.x;
        int n = 0;
        float dobs = 0.f, res = 0.f;
        uint64_t kept = 0;
        float* row = rows + threadIdx.x * SCAN_LD;
        if (r < ov.n_rays) {
            dobs = (r < ov.n_fg) ? dep[r] : 1.1f * d_max;   // optimizer.py:153
            if (voff[r + 1] > voff[r]) n = scan_ray(row, D, d_min, d_max, cfg.cut_off, dobs, kept, res);
        }
        int tot;
        PHASE_MARK(1, 3);
        const int ex = block_excl_scan_256(n, sc, &tot);
        PHASE_MARK(1, 4);
        {
            int w = carry + ex;
            for (uint64_t m = kept; m; m &= m - 1, ++w) {
                e_rk[w] = (r << 6) | (__ffsll((long long)m) - 1);
                e_deds[w] = row[w - (carry + ex)];
                e_res[w] = res;
            }
        }
        carry += tot;
    }
    PHASE_MARK(1, 5);
    if (threadIdx.x == 0) S.n_render = carry;
}
__global__ __launch_bounds__(SCAN_RAYS) void k_scan(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                                    const float* __restrict__ depth, RefineCfg cfg,
                                                    const int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                    const int32_t* __restrict__ ray_voff, int64_t ray_stride,
                                                    const float* __restrict__ sdf_valid, int32_t* __restrict__ rend_rk,
                                                    float* __restrict__ rend_deds, float* __restrict__ rend_res, PlanTail pt) {
    extern __shared__ __attribute__((aligned(16))) float rows[];     // [SCAN_RAYS][SCAN_LD]
    PHASE_MARK(1, 0);
    static_assert((SCAN_RAYS * SCAN_LD) % 4 == 0, "the table is cleared four words at a time");
    for (int e = threadIdx.x; e < SCAN_RAYS * SCAN_LD / 4; e += SCAN_RAYS)
        reinterpret_cast<float4*>(rows)[e] = make_float4(SCAN_NONE, SCAN_NONE, SCAN_NONE, SCAN_NONE);
    scan_body(rows, st, objs, depth, cfg, valid_rk, rk_stride, ray_voff, ray_stride, sdf_valid, rend_rk, rend_deds, rend_res);
    PHASE_MARK(1, 6);
    plan_tail<SCAN_RAYS>(pt, st, objs);      // the Jacobian kernel's item list (k_plan mode 1)
    PHASE_END(1, 7);
}

// ---------------------------------------------------------------------------------------------------------------
// k_mlp_jtj: surface points + kept render rows -> Jacobian rows -> J~^T J~ tile partials
// ---------------------------------------------------------------------------------------------------------------
template <bool B3>
__global__ __launch_bounds__(MLP_THREADS, 2) void k_mlp_jtj(const HypState* __restrict__ st,
                                                            const ObjView* __restrict__ objs,
                                                            const float* __restrict__ pts,
                                                            const float* __restrict__ rays, RefineCfg cfg, const MlpParams* __restrict__ P,
                                                            int nw_sdf, const int32_t* __restrict__ rend_rk,
                                                            const float* __restrict__ rend_deds,
                                                            const float* __restrict__ rend_res, int64_t rk_stride,
                                                            const uint8_t* __restrict__ pt_active, int64_t act_stride,
                                                            float* __restrict__ res_out, float* __restrict__ rows_out, int64_t rows_stride,
                                                            float* __restrict__ partials, int nw_total,
                                                            const int2* __restrict__ work, int* __restrict__ qctl,
                                                            const float* __restrict__ c0_all) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    __shared__ float Tsh[16];
    __shared__ int s_item;
    const int n_items = qctl[2];
    bool tsk_first = true;
    (void)tsk_first;
  for (;;) {                                   // work queue, see k_plan
    QSP_TSK(0)
    if (threadIdx.x == 0) s_item = atomicAdd(&qctl[3], 1);
    __syncthreads();                           // also: everybody is done with the previous item's LDS
    const int item = s_item;
    if (item >= n_items) break;                // the queue only grows towards n_items: every workgroup gets here
    QSP_TSK(1)
    const int h = work[item].x, slot = work[item].y;
    const HypState& S = st[h];
    const ObjView ov = objs[S.obj];
    const bool is_sdf = slot < nw_sdf;
    const int stride = is_sdf ? nw_sdf : nw_total - nw_sdf;
    const int j0 = is_sdf ? slot : slot - nw_sdf;
    const int n = is_sdf ? ov.n_pts : S.n_render;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // J~^T J~ accumulator of this wave's upper-triangular tile (waves 0..5)
    f32x16 hacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[i] = 0.f;
    const int ta = (wave < 3) ? 0 : (wave < 5 ? 1 : 2);
    const int tb = (wave < 3) ? wave : (wave < 5 ? wave - 2 : 2);

    stage_code_T(s, S, Tsh);
    s.c0[threadIdx.x] = c0_all[(size_t)h * 2 * HID + threadIdx.x];
    s.c4[threadIdx.x] = c0_all[(size_t)h * 2 * HID + HID + threadIdx.x];
    const float* Pc = pts + 3 * ov.pts_off;
    const float* R = rays + 3 * ov.ray_off;
    const int32_t* rk = rend_rk + h * rk_stride;
    const float* deds = rend_deds + h * rk_stride;
    const float* rres = rend_res + h * rk_stride;
    const uint8_t* active = pt_active ? pt_active + h * act_stride : nullptr;
    const float d_min = S.d_min, d_max = S.d_max;
    const float hub = is_sdf ? cfg.b2 : cfg.b1;

    for (int t = j0; t * TILE_P < n; t += stride) {
        __syncthreads();
        if (tid < TILE_P) {
            const int v = t * TILE_P + tid;
            float x = 0, y = 0, z = 0, sc = 0.f, rr = 0.f;
            if (v < n) {
                if (is_sdf) {
                    xform(Tsh, Pc[3 * v], Pc[3 * v + 1], Pc[3 * v + 2], x, y, z);
                    sc = (active && !active[v]) ? 0.f : 1.f;
                } else {
                    const int e = rk[v];
                    const int r = e >> 6, k = e & 63;
                    const float d = depth_at(d_min, d_max, k, cfg.n_depth);
                    xform(Tsh, R[3 * r] * d, R[3 * r + 1] * d, R[3 * r + 2] * d, x, y, z);
                    sc = deds[v];
                    rr = rres[v];
                }
            }
            s.xin[4 * tid + 0] = x;
            s.xin[4 * tid + 1] = y;
            s.xin[4 * tid + 2] = z;
            s.xin[4 * tid + 3] = (v < n) ? 1.f : 0.f;   // row-valid flag
            s.rscale[tid] = sc;
            s.rres[tid] = rr;
        }
        __syncthreads();
        QSP_TSK(2)
        mlp_tile<true, 4, !B3, B3>(s, P);      // (AccVGPR accumulators leave the split-bf16 tile too few ArchVGPRs)
        QSP_TSK(3)
        // ---- Jacobian rows: J~[p] = [ s*(g_x . [I | -x^ | x]) (7) | s*g_z (64) | r~ ] -------------------------------
        // G (gradient w.r.t. [code | xyz]) sits in s.act with row stride LDG; J~ goes behind it.
        float* G = s.act;
        float* Jt = s.act + TILE_P * LDG;     // [64][LDJ]
        {
            const int p = tid >> 3, sub = tid & 7;
            const float valid = s.xin[4 * p + 3];
            const float sc = s.rscale[p] * valid;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = sub + 8 * q;           // code column 0..63
                Jt[p * LDJ + 7 + c] = cfg.pose_only ? 0.f : sc * G[p * LDG + c];
            }
            if (sub == 0) {
                const float gx = sc * G[p * LDG + 64], gy = sc * G[p * LDG + 65], gz = sc * G[p * LDG + 66];
                const float x = s.xin[4 * p], y = s.xin[4 * p + 1], z = s.xin[4 * p + 2];
                // [I | -x^ | x]: columns t(3), omega(3), scale(1)   (loss_utils.py:166-185)
                Jt[p * LDJ + 0] = gx;
                Jt[p * LDJ + 1] = gy;
                Jt[p * LDJ + 2] = gz;
                Jt[p * LDJ + 3] = gz * y - gy * z;
                Jt[p * LDJ + 4] = gx * z - gz * x;
                Jt[p * LDJ + 5] = gy * x - gx * y;
                Jt[p * LDJ + 6] = cfg.pose_only ? 0.f : (gx * x + gy * y + gz * z);
                float r = is_sdf ? s.y[p] : s.rres[p];
                float w = cfg.pose_only ? 1.f : huber_w(r, hub);
                if (is_sdf && s.rscale[p] == 0.f) w = 0.f;      // filtered-out point (pose-only inlier mask)
                Jt[p * LDJ + 71] = valid * (w * r);
                if (res_out && is_sdf && valid != 0.f) res_out[h * act_stride + t * TILE_P + p] = r;
            }
            if (sub == 1) {
#pragma unroll
                for (int c = NJ; c < LDJ; ++c) Jt[p * LDJ + c] = 0.f;
            }
        }
        __syncthreads();
        if (rows_out) {   // parity-test tap: the augmented Jacobian rows exactly as the MFMA below consumes them
            float* ro = rows_out + (int64_t)h * rows_stride * NJ + (int64_t)(is_sdf ? 0 : ov.n_pts) * NJ;
            for (int e = tid; e < TILE_P * NJ; e += MLP_THREADS) {
                const int p = e / NJ, c = e - p * NJ;
                const int v = t * TILE_P + p;
                if (v < n) ro[(int64_t)v * NJ + c] = Jt[p * LDJ + c];
            }
        }
        if (wave < 6) {
            const float* A = Jt + (lane >> 5) * LDJ + 32 * ta + (lane & 31);
            const float* B = Jt + (lane >> 5) * LDJ + 32 * tb + (lane & 31);
#pragma unroll 8
            for (int ks = 0; ks < TILE_P / 2; ++ks) hacc = mfma32t<!B3>(A[2 * ks * LDJ], B[2 * ks * LDJ], hacc);
            mfma_acc_settle<!B3>(hacc);
        }
        QSP_TSK(4)
    }
    // partial slot [h][slot][packed upper triangle of 72 x 72]
    if (wave < 6) part_store(partials + ((int64_t)h * nw_total + slot) * PART_FLOATS, ta, tb, lane, hacc);
    QSP_TSK(5)
    tsk_first = false;
  }
}

// the same kernel on the split-fp16 tile: four waves per workgroup (mlp_tile_h2<true>); the six J~^T J~ tiles on waves 0..3
// (waves 0 and 1 carry two).
//
// Register discipline.  The tile needs all 512 registers of its wave (256 accumulators + weight ring + operand sets + ReLU
// masks), so NOTHING of the kernel around it may stay live across it.  Round 2's form kept the J~^T J~ accumulators, the
// per-item pointers and the spilled kernel arguments alive over the tile: 274 spilled VGPRs, 516 B of scratch per lane written
// and read back once per tile (5.2 GB of scratch writes per C4 launch).  Here
//   * the arguments are ONE by-value struct that is only ever read through the kernarg segment pointer, re-derived (opaquely)
//     at the start of each phase: the compiler cannot hoist those scalar loads over the tile, so no argument occupies a
//     register while the tile runs;
//   * everything a phase needs is recomputed from (item, t) -- two scalars -- after the tile;
//   * the J~^T J~ accumulators exist in the epilogue only: zero for an item's first tile, otherwise read back from the item's
//     partial slot (same values in the same order: the MFMA chain continues from the stored sum, bit for bit).  With one tile
//     per work item -- every surface slot below 16 k points, every render slot below 1024 rows -- that read never happens.
struct JtjArgs {
    const HypState* st;
    const ObjView* objs;
    const float* pts;
    const float* rays;
    RefineCfg cfg;
    const MlpParams* P;
    int nw_sdf, nw_total;
    const int32_t* rend_rk;
    const float* rend_deds;
    const float* rend_res;
    int64_t rk_stride;
    const uint8_t* pt_active;
    int64_t act_stride;
    float* res_out;
    float* rows_out;
    int64_t rows_stride;
    float* partials;
    const int2* work;
    int* qctl;
    const float* c0_all;
};
typedef const __attribute__((address_space(4))) JtjArgs* jtj_kargs_t;
__device__ __forceinline__ jtj_kargs_t jtj_kernargs() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));        // a new value as far as the optimiser knows: loads through it stay behind this point
    return (jtj_kargs_t)p;
}

// NW = 4: one wave of 512 registers per SIMD; NW = 8: two of 256 (mlp_tile_h2), the six J~^T J~ tiles on waves 0..5.
template <int NR, int NW = 4, bool NARROW = false>      // NR point blocks of 32 per tile (QSP_DEC_OPT_TPOINTS)
__global__ __launch_bounds__(64 * NW) void k_mlp_jtj_h2(JtjArgs /* read through jtj_kernargs() only */) {
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    __shared__ float Tsh[16];
    __shared__ int s_item;
    constexpr int TP = 32 * NR, SUBS = NT / TP;      // threads per Jacobian row
    bool staged = false;
    float amax = 0.f;
    int item = -1, t = 0;       // the work item in hand and its current tile; item < 0: pop the next one (see k_plan)
    for (;;) {
        // ---- phase 1: pop / stage.  Nothing computed here is used behind the tile. ------------------------------------------
        {
            const jtj_kargs_t A = jtj_kernargs();
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));     // opaque per phase: otherwise every LDS address that depends on the lane is computed once
                                              // per kernel, ahead of the item loop, and held (spilled) across the tile
            const bool fresh = item < 0;
            if (fresh) {
                if (tid == 0) s_item = atomicAdd(&A->qctl[3], 1);
                __syncthreads();                       // also: everybody is done with the previous item's LDS
                item = __builtin_amdgcn_readfirstlane(s_item);
                if (item >= A->qctl[2]) break;         // the queue only grows towards its length: every workgroup gets here
            } else {
                __syncthreads();                       // the previous tile's epilogue has read its Jacobian rows
            }
            const int2 wk = A->work[item];
            const int h = wk.x, slot = wk.y;
            const HypState& S = A->st[h];
            const ObjView ov = A->objs[S.obj];
            const int nw_sdf = A->nw_sdf;
            const bool is_sdf = slot < nw_sdf;
            const int n = is_sdf ? ov.n_pts : S.n_render;
            if (fresh) {
                t = is_sdf ? slot : slot - nw_sdf;
                if (tid < CODE_LEN) s.code[tid] = S.code[tid];
                if (tid >= 64 && tid < 80) Tsh[tid - 64] = S.T_oc[tid - 64];
                const float* c0 = A->c0_all + (size_t)h * 2 * HID;
                for (int i = tid; i < HID; i += NT) {
                    s.c0[i] = c0[i];
                    s.c4[i] = c0[HID + i];
                }
                __syncthreads();                       // Tsh is read below
            }
            if (tid < TP) {
                const int v = t * TP + tid;
                float x = 0, y = 0, z = 0, sc = 0.f, rr = 0.f;
                if (v < n) {
                    if (is_sdf) {
                        const float* Pc = A->pts + 3 * ov.pts_off;
                        const uint8_t* active = A->pt_active ? A->pt_active + h * A->act_stride : nullptr;
                        xform(Tsh, Pc[3 * v], Pc[3 * v + 1], Pc[3 * v + 2], x, y, z);
                        sc = (active && !active[v]) ? 0.f : 1.f;
                    } else {
                        const float* R = A->rays + 3 * ov.ray_off;
                        const int64_t ro = h * A->rk_stride;
                        const int e = A->rend_rk[ro + v];
                        const int r = e >> 6, k = e & 63;
                        const float d = depth_at(S.d_min, S.d_max, k, A->cfg.n_depth);
                        xform(Tsh, R[3 * r] * d, R[3 * r + 1] * d, R[3 * r + 2] * d, x, y, z);
                        sc = A->rend_deds[ro + v];
                        rr = A->rend_res[ro + v];
                    }
                }
                s.xin[4 * tid + 0] = x;
                s.xin[4 * tid + 1] = y;
                s.xin[4 * tid + 2] = z;
                s.xin[4 * tid + 3] = (v < n) ? 1.f : 0.f;   // row-valid flag
                s.rscale[tid] = sc;
                s.rres[tid] = rr;
            }
            __syncthreads();
        }
        // ---- the tile: live across it are item, t, staged (scalars) and amax ---------------------------------------------------
        {
            const MlpParams* Pm = jtj_kernargs()->P;
            mlp_tile_h2<true, 2, false, NR, NW, NARROW>(s, Pm, amax, !staged);
            staged = true;
        }
        // ---- phase 2: Jacobian rows  J~[p] = [ s*(g_x . [I | -x^ | x]) (7) | s*g_z (64) | r~ ],  J~^T J~ -----------------------
        // G (gradient w.r.t. [code | xyz]) sits in s.act with row stride LDG; J~ goes behind it.
        {
            asm volatile("" : "+s"(item), "+s"(t));
            const jtj_kargs_t A = jtj_kernargs();
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            const int2 wk = A->work[item];
            const int h = wk.x, slot = wk.y;
            const HypState& S = A->st[h];
            const ObjView ov = A->objs[S.obj];
            const int nw_sdf = A->nw_sdf, nw_total = A->nw_total;
            const bool is_sdf = slot < nw_sdf;
            const int n = is_sdf ? ov.n_pts : S.n_render;
            const int j0 = is_sdf ? slot : slot - nw_sdf;
            const int stride = is_sdf ? nw_sdf : nw_total - nw_sdf;
            const int pose_only = A->cfg.pose_only;
            const float hub = is_sdf ? A->cfg.b2 : A->cfg.b1;
            float* G = s.act;
            float* Jt = s.act + TILE_P * LDG;   /* (behind the 64-row G image whatever the tile size) */     // [64][LDJ]
            {
                const int p = tid / SUBS, sub = tid % SUBS;
                const float valid = s.xin[4 * p + 3];
                const float sc = s.rscale[p] * valid;
#pragma unroll
                for (int q = 0; q < CODE_LEN / SUBS; ++q) {
                    const int c = sub + SUBS * q;        // code column 0..63
                    Jt[p * LDJ + 7 + c] = pose_only ? 0.f : sc * G[p * LDG + c];
                }
                if (sub == 0) {
                    const float gx = sc * G[p * LDG + 64], gy = sc * G[p * LDG + 65], gz = sc * G[p * LDG + 66];
                    const float x = s.xin[4 * p], y = s.xin[4 * p + 1], z = s.xin[4 * p + 2];
                    // [I | -x^ | x]: columns t(3), omega(3), scale(1)   (loss_utils.py:166-185)
                    Jt[p * LDJ + 0] = gx;
                    Jt[p * LDJ + 1] = gy;
                    Jt[p * LDJ + 2] = gz;
                    Jt[p * LDJ + 3] = gz * y - gy * z;
                    Jt[p * LDJ + 4] = gx * z - gz * x;
                    Jt[p * LDJ + 5] = gy * x - gx * y;
                    Jt[p * LDJ + 6] = pose_only ? 0.f : (gx * x + gy * y + gz * z);
                    float r = is_sdf ? s.y[p] : s.rres[p];
                    float w = pose_only ? 1.f : huber_w(r, hub);
                    if (is_sdf && s.rscale[p] == 0.f) w = 0.f;      // filtered-out point (pose-only inlier mask)
                    Jt[p * LDJ + 71] = valid * (w * r);
                    float* res_out = A->res_out;
                    if (res_out && is_sdf && valid != 0.f) res_out[h * A->act_stride + t * TP + p] = r;
                }
                if (sub == 1) {
#pragma unroll
                    for (int c = NJ; c < LDJ; ++c) Jt[p * LDJ + c] = 0.f;
                }
            }
            __syncthreads();
            if (A->rows_out) {   // parity-test tap: the augmented Jacobian rows exactly as the MFMA below consumes them
                float* ro = A->rows_out + (int64_t)h * A->rows_stride * NJ + (int64_t)(is_sdf ? 0 : ov.n_pts) * NJ;
                for (int e = tid; e < TP * NJ; e += NT) {
                    const int p = e / NJ, c = e - p * NJ;
                    const int v = t * TP + p;
                    if (v < n) ro[(int64_t)v * NJ + c] = Jt[p * LDJ + c];
                }
            }
            // upper-triangular 32x32 tiles in the order (0,0) (0,1) (0,2) (1,1) (1,2) (2,2).  Four waves: tile w on every wave, tile
            // w + 4 on waves 0, 1; eight waves: tile w on waves 0..5.  Partial slot [h][slot][packed upper triangle].
            float* out = A->partials + ((int64_t)h * nw_total + slot) * PART_FLOATS;
            const bool first = t == j0;
            if (NW == 4 || wave < 6) {
                const int ta0 = wave < 3 ? 0 : (wave < 5 ? 1 : 2), tb0 = wave < 3 ? wave : (wave < 5 ? wave - 2 : 2);
                f32x16 hacc;
                part_load(out, ta0, tb0, lane, first, hacc);
                const float* Aj = Jt + (lane >> 5) * LDJ + 32 * ta0 + (lane & 31);
                const float* Bj = Jt + (lane >> 5) * LDJ + 32 * tb0 + (lane & 31);
#pragma unroll 8
                for (int ks = 0; ks < TP / 2; ++ks) hacc = mfma32t<false>(Aj[2 * ks * LDJ], Bj[2 * ks * LDJ], hacc);
                part_store(out, ta0, tb0, lane, hacc);
            }
            if (NW == 4 && wave < 2) {
                const int ta1 = wave == 0 ? 1 : 2, tb1 = 2;
                f32x16 hacc;
                part_load(out, ta1, tb1, lane, first, hacc);
                const float* Aj = Jt + (lane >> 5) * LDJ + 32 * ta1 + (lane & 31);
                const float* Bj = Jt + (lane >> 5) * LDJ + 32 * tb1 + (lane & 31);
#pragma unroll 8
                for (int ks = 0; ks < TP / 2; ++ks) hacc = mfma32t<false>(Aj[2 * ks * LDJ], Bj[2 * ks * LDJ], hacc);
                part_store(out, ta1, tb1, lane, hacc);
            }
            // the item's next tile (more than nw_sdf x TP surface points, or more than 16 x TP render rows), or the next item
            t += stride;
            if (t * TP >= n) item = -1;
        }
    }
    if (!(amax <= H2_MAX)) *jtj_kernargs()->P->range_flag = 1;
}

// ---------------------------------------------------------------------------------------------------------------
// k_solve: reduce partials, priors, damping, solve, update (optimizer.py:207-263)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tri_tile(int a, int b) {   // tile index of block (a<=b) in the 3x3 upper triangle
    return a == 0 ? b : (a == 1 ? 2 + b : 5);
}

__device__ void exp_sim3_dev(const float* x, float* T) {   // loss_utils.py:188-233, f32
    const float v0 = x[0], v1 = x[1], v2 = x[2], w0 = x[3], w1 = x[4], w2 = x[5], sg = x[6];
    const float W[9] = {0.f, -w2, w1, w2, 0.f, -w0, -w1, w0, 0.f};
    float W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    const float th = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    const float es = expf(sg);
    float Rm[9], J[9];
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (th <= 1e-8f) {
        const float c = (sg == 0.f) ? 1.f : (es - 1.f) / sg;
        for (int i = 0; i < 9; ++i) { Rm[i] = I[i]; J[i] = c * I[i]; }
    } else {
        const float sn = sinf(th), cs = cosf(th);
        const float a = es * sn, b = es * cs;
        const float c = (sg <= 1e-8f) ? 0.f : (es - 1.f) / sg;
        const float den = sg * sg + th * th;
        const float k1 = (a * sg + (1.f - b) * th) / den;
        const float k2 = c - ((b - 1.f) * sg + a * th) / den;
        for (int i = 0; i < 9; ++i) {
            Rm[i] = I[i] + W[i] * sn / th + W2[i] * (1.f - cs) / (th * th);
            J[i] = c * I[i] + k1 * W[i] / th + k2 * W2[i] / (th * th);
        }
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = es * Rm[3 * i + j];
        T[4 * i + 3] = J[3 * i] * v0 + J[3 * i + 1] * v1 + J[3 * i + 2] * v2;
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

__device__ void exp_se3_dev(const float* x, float* T) {   // loss_utils.py:129-163, f32
    const float v0 = x[0], v1 = x[1], v2 = x[2], w0 = x[3], w1 = x[4], w2 = x[5];
    const float W[9] = {0.f, -w2, w1, w2, 0.f, -w0, -w1, w0, 0.f};
    float W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    const float th = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float Rm[9], J[9];
    if (th <= 1e-8f) {
        for (int i = 0; i < 9; ++i) { Rm[i] = I[i]; J[i] = I[i]; }
    } else {
        const float sn = sinf(th), cs = cosf(th);
        const float th2 = th * th, th3 = th2 * th;
        for (int i = 0; i < 9; ++i) {
            Rm[i] = I[i] + W[i] * sn / th + W2[i] * (1.f - cs) / th2;
            J[i] = I[i] + ((1.f - cs) / th2) * W[i] + ((th - sn) / th3) * W2[i];
        }
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = Rm[3 * i + j];
        T[4 * i + 3] = J[3 * i] * v0 + J[3 * i + 1] * v1 + J[3 * i + 2] * v2;
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

#ifndef QSP_SOLVE_EXP
#define QSP_SOLVE_EXP 0      // timing experiments only (1: no elimination, 2: no slot reads); 0 in every build that ships
#endif
constexpr int SOLVE_THREADS = 1024;   // latency, not throughput: more loads in flight for the partial sums, shorter row strips per pivot
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                               RefineCfg cfg, const float* __restrict__ partials, int nw_sdf,
                                               int nw_total, const uint8_t* __restrict__ pt_active,
                                               int64_t act_stride, float* __restrict__ trH, float* __restrict__ trb,
                                               float* __restrict__ trdx, unsigned long long* __restrict__ counters,
                                               float* __restrict__ trrot) {
    const int h = blockIdx.x;
    HypState& S = st[h];
    if (!S.alive) return;
    PHASE_MARK(2, 0);
    if (threadIdx.x == 0 && counters) {   // work actually done this iteration (for the roofline figures)
        const ObjView o = objs[S.obj];
        atomicAdd(&counters[0], (unsigned long long)(o.n_pts + S.n_render));
        atomicAdd(&counters[1], (unsigned long long)S.n_eval);      // (= n_valid unless the forward pass was depth-staged)
        atomicAdd(&counters[2], (unsigned long long)((o.n_pts + cfg.tile_p - 1) / cfg.tile_p + (S.n_render + cfg.tile_p - 1) / cfg.tile_p));
        atomicAdd(&counters[3], (unsigned long long)((S.n_eval + TILE_P - 1) / TILE_P));
        atomicAdd(&counters[4], (unsigned long long)(S.n_band_total + S.n_band));
    }
    __shared__ double Hd[NH * (NH + 1)];  // augmented [H | b] in f64
    __shared__ float dxs[NH];
    __shared__ float loss_sh[2];
    __shared__ int n_act_sh;
    const ObjView ov = objs[S.obj];
    const int tid = threadIdx.x;
    const float* base = partials + (int64_t)h * nw_total * PART_FLOATS;
    const int n_sdf_slots = min(nw_sdf, (ov.n_pts + cfg.tile_p - 1) / cfg.tile_p);
    const int K = S.n_render;
    const int n_rend_slots = min(nw_total - nw_sdf, (K + cfg.tile_p - 1) / cfg.tile_p);
    // number of active surface points (pose-only inlier filter; otherwise n_pts)
    if (tid == 0) n_act_sh = ov.n_pts;
    __syncthreads();
    if (pt_active) {
        __shared__ int cnt_sh;
        if (tid == 0) cnt_sh = 0;
        __syncthreads();
        int c = 0;
        for (int i = tid; i < ov.n_pts; i += SOLVE_THREADS) c += pt_active[h * act_stride + i] ? 1 : 0;
        atomicAdd(&cnt_sh, c);
        __syncthreads();
        if (tid == 0) n_act_sh = cnt_sh;
        __syncthreads();
    }
    PHASE_MARK(2, 1);
    const float M = (float)n_act_sh;
    const float Kf = (float)K;
    const int N = cfg.pose_only ? 6 : NH;
    // The rotation prior's Jacobian and residual (loss.py:155-178) need the pose only: the LAST thread forms them while the others
    // sum the partial slots (its wave has no entry in the third turn of that loop), thread 0 applies them behind the barrier.
    // (Thread 0 used to do both behind the barrier: 3 us of one lane's divisions with 1023 threads waiting.)
    __shared__ float prior_sh[8];      // Jr[0..6], rr
    if (tid == SOLVE_THREADS - 1 && !cfg.pose_only) {
        float rco[9];
        const float sc = powf(det3(S.T_co), (float)(1.0 / 3.0));
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) rco[3 * i + j] = S.T_co[4 * i + j] / sc;
        // r_oc = inverse(r_co); for a rotation this is the transpose up to rounding -- invert generally (3x3, f64)
        double m[9];
        for (int i = 0; i < 9; ++i) m[i] = rco[i];
        const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                           m[2] * (m[3] * m[7] - m[4] * m[6]);
        float roc[9];
        roc[0] = (float)((m[4] * m[8] - m[5] * m[7]) / det);
        roc[1] = (float)((m[2] * m[7] - m[1] * m[8]) / det);
        roc[2] = (float)((m[1] * m[5] - m[2] * m[4]) / det);
        roc[3] = (float)((m[5] * m[6] - m[3] * m[8]) / det);
        roc[4] = (float)((m[0] * m[8] - m[2] * m[6]) / det);
        roc[5] = (float)((m[2] * m[3] - m[0] * m[5]) / det);
        roc[6] = (float)((m[3] * m[7] - m[4] * m[6]) / det);
        roc[7] = (float)((m[1] * m[6] - m[0] * m[7]) / det);
        roc[8] = (float)((m[0] * m[4] - m[1] * m[3]) / det);
        // ry = r_co e_y ; res = 1 - ry . n_g, n_g = (0,-1,0)
        const float res_rot = 1.f - (-(rco[4]));
        float Jr[7] = {0, 0, 0, 0, 0, 0, 0};
        float rr = 0.f;
        if (!(res_rot < 1e-7f)) {
            // (r_oc n_g) x e_y with n_g = (0,-1,0): a = -r_oc[:,1]; a x e_y = (-a_z, 0, a_x)
            const float ax = -roc[1], az = -roc[7];
            Jr[3] = -az;
            Jr[4] = 0.f;
            Jr[5] = ax;
            rr = res_rot;
        }
        if (trrot) {     // parity-test tap: the rotation prior's own Jacobian and residual (loss.py:155-178)
            trrot[4 * h + 0] = Jr[3];
            trrot[4 * h + 1] = Jr[4];
            trrot[4 * h + 2] = Jr[5];
            trrot[4 * h + 3] = rr;
        }
        for (int a = 0; a < 7; ++a) prior_sh[a] = Jr[a];
        prior_sh[7] = rr;
    }
    // Fixed-order sum of the tile partials (deterministic), then H, b in f32 exactly as optimizer.py:217-252 orders the
    // operations; entries are promoted to f64 only for the linear solve.
    // (one packed entry per thread and turn: 2628 entries are three turns of consecutive words -- walking the 72 x 72 square and
    //  skipping its lower half was five turns, 14 us of a one-object call's 58)
    for (int off = tid; off < NJ * (NJ + 1) / 2; off += SOLVE_THREADS) {
        int a = (int)(0.5f * ((float)(2 * NJ + 1) - sqrtf((float)((2 * NJ + 1) * (2 * NJ + 1) - 8 * off))));      // row of the packed entry,
        a = min(max(a, 0), NJ - 1);
        while (a + 1 < NJ && tri72(a + 1, a + 1) <= off) ++a;                                                     // made exact
        while (tri72(a, a) > off) --a;
        const int b = a + (off - tri72(a, a));
        float ss = 0.f, sr = 0.f;
        // same left-to-right order as ever; unrolled so that 32 of the (24 KiB-strided) loads are in flight at a time (a single
        // object's 2 k points are 63 slots of 32-point tiles: two batches instead of four in front of every entry)
#if QSP_SOLVE_EXP != 2
#pragma unroll 32
        for (int j = 0; j < n_sdf_slots; ++j) ss += base[(int64_t)j * PART_FLOATS + off];
#pragma unroll 16
        for (int j = 0; j < n_rend_slots; ++j) sr += base[(int64_t)(nw_sdf + j) * PART_FLOATS + off];
#else
        ss = 1.f + 0.001f * (float)off; sr = 1.f;      // (timing experiment: no slot reads)
#endif
        if (b < NH) {                      // normal-matrix entry
            if (a >= N || b >= N) continue;
            float v;
            if (cfg.pose_only) {
                v = ss / M;
                if (a == b) v += 1e-2f;                        // optimizer.py:75
            } else {
                v = (cfg.k1 * sr) / Kf + (cfg.k2 * ss) / M;
                if (a == b && a >= 7) v += cfg.k3;
                // code unknowns beyond the decoder's code length have zero Jacobian columns: unit diagonal, zero right-hand
                // side -> they stay 0 whatever k3 is and never mix into the other 7 + L unknowns
                if (a >= 7 + cfg.code_len || b >= 7 + cfg.code_len) v = (a == b) ? 1.f : 0.f;
            }
            Hd[a * (N + 1) + b] = (double)v;
            Hd[b * (N + 1) + a] = (double)v;
        } else if (a < NH) {               // right-hand side: column 71 of J~^T J~ is J^T r~
            if (a >= N) continue;
            float v;
            if (cfg.pose_only) v = -ss / M;
            else {
                v = -(cfg.k1 * sr) / Kf + (-(cfg.k2 * ss) / M);
                if (a >= 7) v -= cfg.k3 * S.code[a - 7];
                if (a >= 7 + cfg.code_len) v = 0.f;
            }
            Hd[a * (N + 1) + N] = (double)v;
        } else {                           // (71,71): sum of squared robust residuals
            loss_sh[0] = ss / M;                               // mean(robust_res^2)
            loss_sh[1] = cfg.pose_only ? 0.f : sr / Kf;
        }
    }
    __syncthreads();
    PHASE_MARK(2, 2);
    const float loss_s = loss_sh[0], loss_r = loss_sh[1];
    const bool bad = isnan(loss_s) || isnan(loss_r);           // optimizer.py:168-169,193-194
    __syncthreads();
    if (bad) {
        if (tid == 0) S.alive = 0;
        return;
    }
    if (tid == 0 && !cfg.pose_only) {
        // rotation prior on the pose block (formed above), then damping (optimizer.py:240-252)
        float Jr[7];
        for (int a = 0; a < 7; ++a) Jr[a] = prior_sh[a];
        const float rr = prior_sh[7];
        for (int a = 0; a < 7; ++a) {
            for (int b = 0; b < 7; ++b) {
                float v = (float)Hd[a * (N + 1) + b];
                v += cfg.k4 * (Jr[a] * Jr[b]);
                if (a == b) v += 1.0f;
                if (a == 6 && b == 6) v += cfg.s_damp;
                Hd[a * (N + 1) + b] = (double)v;
            }
            float bv = (float)Hd[a * (N + 1) + N];
            bv -= cfg.k4 * (-(Jr[a] * rr));
            Hd[a * (N + 1) + N] = (double)bv;
        }
    }
    __syncthreads();
    if (trH) {
        for (int e = tid; e < N * N; e += SOLVE_THREADS) trH[(int64_t)h * NH * NH + (e / N) * NH + (e % N)] = (float)Hd[(e / N) * (N + 1) + (e % N)];
        for (int a = tid; a < N; a += SOLVE_THREADS) trb[(int64_t)h * NH + a] = (float)Hd[a * (N + 1) + N];
    }
    __syncthreads();
    PHASE_MARK(2, 3);
#if QSP_PHASE_CLOCK
    const unsigned long long cyc0 = __builtin_readcyclecounter();
#endif
    // Gauss-Jordan elimination in f64 on the augmented system (reference: torch.inverse(H) @ b, f32).  H is symmetric
    // positive definite by construction (Gram matrices plus the identity damping of optimizer.py:240-252 / :75), so no
    // pivot search is needed: one barrier per column, every (row, column strip) pair on its own thread.
    // ~1200 cycles of the 2.4 GHz shader clock per column (tools/phase_clock.py), 36 of this kernel's 58 us on a one-object call.
    // Measured against it in round 4, none faster: the same loop on four waves (2.76 -> 2.94 ms per call); a column's reads issued
    // in front of the division (1650 cycles per column: six predicated read pairs every column instead of the 2.5 the strip
    // needs on average); the matrix in registers with the pivot row, the pivot column and ONE reciprocal published through LDS, on
    // sixteen waves with a runtime column loop (1640) and on eight waves with the loop unrolled (1280: a column is ~9 LDS reads and
    // 14 f64 operations per wave whichever way the entries are dealt, behind the pivot's division and a barrier).
    const int STR = SOLVE_THREADS / N;         // 14 strips for the 71 x 71 system
#if QSP_SOLVE_EXP == 1
    for (int c = 0; c < 0; ++c) {              // (timing experiment: no elimination)
#else
    for (int c = 0; c < N; ++c) {
#endif
        const double inv = 1.0 / Hd[c * (N + 1) + c];
        const int r = tid / STR, q = tid - r * STR;
        if (r < N && r != c) {
            const double f = Hd[r * (N + 1) + c] * inv;
            for (int j = c + 1 + q; j <= N; j += STR) Hd[r * (N + 1) + j] -= f * Hd[c * (N + 1) + j];
        }
        __syncthreads();
    }
#if QSP_PHASE_CLOCK
    if (blockIdx.x == 0 && tid == 0) qsp_phase_ticks[2][8] += __builtin_readcyclecounter() - cyc0;
#endif
    PHASE_MARK(2, 4);
    for (int a = tid; a < N; a += SOLVE_THREADS) dxs[a] = (float)(Hd[a * (N + 1) + N] / Hd[a * (N + 1) + a]);
    __syncthreads();
    if (trdx)
        for (int a = tid; a < N; a += SOLVE_THREADS) trdx[(int64_t)h * NH + a] = dxs[a];
    if (tid == 0) {
        float d[7], Td[16], Tn[16];
        if (cfg.pose_only) {
            for (int i = 0; i < 6; ++i) d[i] = dxs[i];
            exp_se3_dev(d, Td);
        } else {
            for (int i = 0; i < 7; ++i) d[i] = cfg.lr * dxs[i];
            exp_sim3_dev(d, Td);
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
                for (int k = 0; k < 4; ++k) a += Td[4 * i + k] * S.T_oc[4 * k + j];
                Tn[4 * i + j] = a;
            }
        for (int i = 0; i < 16; ++i) S.T_oc[i] = Tn[i];
        S.loss_sdf = loss_s;
        S.loss_render = loss_r;
        S.loss = cfg.k1 * loss_r + cfg.k2 * loss_s;    // optimizer.py:203
    }
    if (!cfg.pose_only && tid < CODE_LEN) S.code[tid] += cfg.lr * dxs[7 + tid];
    PHASE_END(2, 5);
}

// pose-only inlier filter after iteration index 4 (optimizer.py:80-82): |res| <= 0.05 on the residuals of THAT iteration
__global__ void k_inlier_filter(const HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                const float* __restrict__ res, int64_t act_stride, uint8_t* __restrict__ pt_active) {
    const int h = blockIdx.y;
    const ObjView ov = objs[st[h].obj];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ov.n_pts) {
        const bool keep = fabsf(res[h * act_stride + i]) <= 0.05f;
        pt_active[h * act_stride + i] = (pt_active[h * act_stride + i] && keep) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// generic decode kernels for the API-level entry points (loss_utils.py:51-103): points already in the object frame
// ---------------------------------------------------------------------------------------------------------------
template <bool GRAD, bool BF3 = false>
__global__ __launch_bounds__(MLP_THREADS, 2) void k_decode(const float* __restrict__ code, const float* __restrict__ xyz,
                                                           int64_t n, const MlpParams* __restrict__ P, float* __restrict__ y_out,
                                                           float* __restrict__ grad_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    if (threadIdx.x < CODE_LEN) s.code[threadIdx.x] = code[threadIdx.x];
    mlp_prepare(s, P);
    for (int64_t t = blockIdx.x; t * TILE_P < n; t += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < TILE_P) {
            const int64_t v = t * TILE_P + threadIdx.x;
            float x = 0, y = 0, z = 0;
            if (v < n) { x = xyz[3 * v]; y = xyz[3 * v + 1]; z = xyz[3 * v + 2]; }
            s.xin[4 * threadIdx.x + 0] = x;
            s.xin[4 * threadIdx.x + 1] = y;
            s.xin[4 * threadIdx.x + 2] = z;
            s.xin[4 * threadIdx.x + 3] = 0.f;
        }
        __syncthreads();
        if (BF3 && !GRAD) mlp_tile_bf3<QSP_BF3_PF>(s, P);
        else mlp_tile<GRAD, 4, false, BF3>(s, P);
        if (threadIdx.x < TILE_P) {
            const int64_t v = t * TILE_P + threadIdx.x;
            if (v < n) y_out[v] = s.y[threadIdx.x];
        }
        if (GRAD) {
            for (int e = threadIdx.x; e < TILE_P * NIN; e += MLP_THREADS) {
                const int p = e / NIN, c = e % NIN;
                const int64_t v = t * TILE_P + p;
                if (v < n) grad_out[v * NIN + c] = s.act[p * LDG + c];
            }
        }
    }
}

// decode on the split-fp16 tile (four waves per workgroup)
template <bool GRAD, bool NARROW = false>
__global__ __launch_bounds__(H2_THREADS) void k_decode_h2(const float* __restrict__ code, const float* __restrict__ xyz, int64_t n,
                                                          const MlpParams* __restrict__ P, float* __restrict__ y_out,
                                                          float* __restrict__ grad_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    if (threadIdx.x < CODE_LEN) s.code[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    for (int u = threadIdx.x; u < HID; u += H2_THREADS) {      // mlp_prepare for 256 threads
        float a, a4;
        code_bias(P, u, s.code, a, a4);
        s.c0[u] = a;
        s.c4[u] = a4;
    }
    bool staged = false;
    float amax = 0.f;
    for (int64_t t = blockIdx.x; t * TILE_P < n; t += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < TILE_P) {
            const int64_t v = t * TILE_P + threadIdx.x;
            float x = 0, y = 0, z = 0;
            if (v < n) { x = xyz[3 * v]; y = xyz[3 * v + 1]; z = xyz[3 * v + 2]; }
            s.xin[4 * threadIdx.x + 0] = x;
            s.xin[4 * threadIdx.x + 1] = y;
            s.xin[4 * threadIdx.x + 2] = z;
            s.xin[4 * threadIdx.x + 3] = 0.f;
        }
        __syncthreads();
        mlp_tile_h2<GRAD, 2, !GRAD && !NARROW, 2, 4, NARROW>(s, P, amax, !staged);
        staged = true;
        if (threadIdx.x < TILE_P) {
            const int64_t v = t * TILE_P + threadIdx.x;
            if (v < n) y_out[v] = s.y[threadIdx.x];
        }
        if (GRAD) {
            for (int e = threadIdx.x; e < TILE_P * NIN; e += H2_THREADS) {
                const int p = e / NIN, c = e % NIN;
                const int64_t v = t * TILE_P + p;
                if (v < n) grad_out[v * NIN + c] = s.act[p * LDG + c];
            }
        }
    }
    if (!(amax <= H2_MAX)) *P->range_flag = 1;
}

// the screening tile (mlp_tile_h1) on explicit query points: what the first pass of the screened forward computes, exposed for
// the tests and the margin measurement (qsp_decode_sdf_screen) -- these are NOT SDF values of the decoder's precision
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_decode_screen(const float* __restrict__ code, const float* __restrict__ xyz, int64_t n,
                                                              const MlpParams* __restrict__ P, float* __restrict__ y_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmemH1& s = *reinterpret_cast<MlpSmemH1*>(smem_raw);
    float* codes = s.red;                   // (free until the first tile's layer 8)
    if (threadIdx.x < CODE_LEN) codes[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    for (int u = threadIdx.x; u < HID; u += 64 * NW) {
        float a, a4;
        code_bias(P, u, codes, a, a4);
        s.c0[u] = a;
        s.c4[u] = a4;
        s.w8[u] = P->w8[u];
    }
#pragma unroll
    for (int l = 1; l < 8; ++l)
        for (int i = threadIdx.x; i < HID; i += 64 * NW) s.bias[(l - 1) * HID + i] = P->bias[l][i];
    float amax = 0.f;
    for (int64_t t = blockIdx.x; t * H1_ROWS < n; t += gridDim.x) {
        __syncthreads();
        if (threadIdx.x < H1_ROWS) {
            const int64_t v = t * H1_ROWS + threadIdx.x;
            float x = 0, y = 0, z = 0;
            if (v < n) { x = xyz[3 * v]; y = xyz[3 * v + 1]; z = xyz[3 * v + 2]; }
            s.xin[4 * threadIdx.x + 0] = x;
            s.xin[4 * threadIdx.x + 1] = y;
            s.xin[4 * threadIdx.x + 2] = z;
            s.xin[4 * threadIdx.x + 3] = 0.f;
        }
        __syncthreads();
        mlp_tile_h1<2, NW>(s, P, amax);
        if (threadIdx.x < H1_ROWS) {
            const int64_t v = t * H1_ROWS + threadIdx.x;
            if (v < n) y_out[v] = s.y[threadIdx.x];
        }
    }
    if (!(amax <= H2_MAX)) *P->range_flag = 1;
}

// 4x4 inverse as the reference's torch.inverse calls need it (optimizer.py:123,273): Gauss-Jordan with partial pivoting in
// double, rounded to f32.  One definition for host (set_state / get) and device (detections.hpp), no contraction, so both
// give the same bits.
__host__ __device__ inline void inv4_gj(const float* in, float* out) {
#pragma clang fp contract(off)
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = in[4 * i + j];
            a[i][4 + j] = i == j;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 8; ++j) {
                const double t = a[c][j];
                a[c][j] = a[p][j];
                a[p][j] = t;
            }
        const double inv = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= inv;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = (float)a[i][4 + j];
}

}  // namespace qsp
