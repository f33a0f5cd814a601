// sdf_refine.hip -- path A of the hot path: batched DeepSDF object refinement on gfx950.
//
// Replaces reconstruct/optimizer.py:96-281 (Optimizer.reconstruct_object), :47-93 (estimate_pose_cam_obj),
// reconstruct/loss.py:22-178 and reconstruct/loss_utils.py:40-265 of the reference.  One Gauss-Newton iteration of
// EVERY hypothesis in the batch is five launches with no host round trip:
//
//   k_sample      per hypothesis: T_co, scale, depth range; ray x depth samples inside the unit ball (loss.py:60-74)
//   k_mlp_fwd     decoder forward on the valid samples                                  (loss.py:78)
//   k_scan        per ray: occupancy, transmittance, rendered depth, de/ds; keeps rows   (loss.py:84-141)
//   k_mlp_jtj     decoder forward+backward on surface points and kept render rows, Jacobian rows, Huber weights and
//                 the 72x72 augmented normal-equation tile J~^T J~ (J~ = [J | r~]) by MFMA (loss.py:22-43,143-150,
//                 optimizer.py:217-226)
//   k_solve       fixed-order reduction of the tile partials, priors, damping, 71x71 solve, exp_sim3, state update
//                 (optimizer.py:231-263)
//
// (plus k_c0 and k_plan: per-hypothesis bias vectors and the work queues of the two MLP kernels.)  The two MLP kernels exist
// per decoder pipe (qsp_decoder_set_option): k_mlp_fwd<false> / k_mlp_jtj<false> on the exact-f32 matrix pipe, <true> with three
// bf16 terms per operand, k_mlp_fwd_h2 / k_mlp_jtj_h2 with two fp16 terms (four waves per workgroup, sdf_mlp.hpp).
//
// The reference does the same work as ~60 small torch launches and ~10 host synchronisations per iteration,
// per hypothesis, serially.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/qsp_hip.h"
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
constexpr int PART_FLOATS = HT_TILES * 1024;

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
    int32_t pad[2];
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

// exclusive scan of one int per thread over the block (a multiple of 64, at most 512 threads); returns the exclusive prefix,
// total in *total
__device__ int block_excl_scan_256(int v, int* smem /* >= 8 ints */, int* total) {
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
__global__ __launch_bounds__(256) void k_sample(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                                const float* __restrict__ rays, RefineCfg cfg,
                                                int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                int32_t* __restrict__ ray_voff, int64_t ray_stride) {
    const int h = blockIdx.x;
    HypState& S = st[h];
    if (!S.alive) return;
    __shared__ float T[16];
    __shared__ float dm[2];
    __shared__ int sc[8];
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
    const ObjView ov = objs[S.obj];
    const int D = cfg.n_depth;
    const float* R = rays + 3 * ov.ray_off;
    int32_t* rk = valid_rk + h * rk_stride;
    int32_t* voff = ray_voff + h * ray_stride;
    int carry = 0;
    for (int base = 0; base < ov.n_rays; base += 256) {
        const int r = base + threadIdx.x;
        uint64_t mask = 0;
        float rx = 0, ry = 0, rz = 0;
        if (r < ov.n_rays) {
            rx = R[3 * r], ry = R[3 * r + 1], rz = R[3 * r + 2];
            for (int k = 0; k < D; ++k) {
                const float d = depth_at(dm[0], dm[1], k, D);
                float x, y, z;
                xform(T, rx * d, ry * d, rz * d, x, y, z);
                if (sqrtf(x * x + y * y + z * z) < 1.0f) mask |= (1ull << k);
            }
        }
        const int cnt = __popcll(mask);
        int tot;
        const int ex = block_excl_scan_256(cnt, sc, &tot);
        if (r < ov.n_rays) {
            voff[r] = carry + ex;
            int w = carry + ex;
            for (int k = 0; k < D; ++k)
                if (mask >> k & 1ull) rk[w++] = (r << 6) | k;
        }
        carry += tot;
    }
    if (threadIdx.x == 0) {
        voff[ov.n_rays] = carry;
        S.n_valid = carry;
        S.n_render = 0;
        if (!cfg.pose_only && carry < 10) S.alive = 0;   // loss.py:73-74 -> optimizer.py:171-172
    }
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
    const float* w = Pm->w0c + (size_t)u * CODE_LEN;
    const float* w4 = Pm->w4c + (size_t)u * CODE_LEN;
    float a = Pm->bias[0][u], a4 = Pm->bias[4][u];
#pragma unroll 8
    for (int k = 0; k < CODE_LEN; ++k) {
        a += w[k] * code[k];
        a4 += w4[k] * code[k];
    }
    c0_all[(size_t)blockIdx.x * 2 * HID + u] = a;
    c0_all[(size_t)blockIdx.x * 2 * HID + HID + u] = a4;     // layer 4's bias with the skip connection's code part
}

// mode 0: forward items (h, tile) over the valid ray samples; mode 1: jtj items (h, slot), surface slots then render slots
__global__ __launch_bounds__(1024) void k_plan(int mode, const HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                               int n_hyp, int nw_sdf, int nw_rend, int2* __restrict__ work, int* __restrict__ qctl,
                                               int tile_p) {
    __shared__ int wsum[16];
    __shared__ int carry_sh;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry_sh = 0;
    __syncthreads();
    for (int base = 0; base < n_hyp; base += 1024) {
        const int h = base + t;
        int n_a = 0, n_b = 0;
        if (h < n_hyp && st[h].alive) {
            if (mode == 0) n_a = (st[h].n_valid + tile_p - 1) / tile_p;
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
        for (int j = 0; j < n_a; ++j) work[off + j] = make_int2(h, j);
        for (int j = 0; j < n_b; ++j) work[off + n_a + j] = make_int2(h, nw_sdf + j);
        __syncthreads();
        if (t == 1023) carry_sh = off + cnt;
        __syncthreads();
    }
    if (t == 0) {
        qctl[2 * mode] = carry_sh;
        qctl[2 * mode + 1] = 0;
    }
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

// the same work queue on the split-fp16 tile: four waves per workgroup (mlp_tile_h2)
template <int NR>      // NR point blocks of 32 per tile (QSP_DEC_OPT_TPOINTS)
__global__ __launch_bounds__(H2_THREADS) void k_mlp_fwd_h2(const HypState* __restrict__ st,
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
    constexpr int TP = 32 * NR;
    bool staged = false;
    float amax = 0.f;
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
            for (int i = threadIdx.x; i < HID; i += H2_THREADS) {
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
        mlp_tile_h2<false, 2, true, NR>(s, P, amax, !staged);      // (the decoder's constants: staged by the first tile of the workgroup)
        staged = true;
        if (threadIdx.x < TP) {
            const int v = t * TP + threadIdx.x;
            if (v < n) out[v] = s.y[threadIdx.x];
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

// walks one ray; if `emit` != nullptr writes the kept rows starting at emit index `w`
// One ray: `row` holds the SDF value of its depth sample k at row[k] (SCAN_NONE where the sample is outside the unit ball), staged
// in LDS by the workgroup (k_scan).  Every per-sample array is indexed by the unrolled loop counter only, so it lives in
// registers.  (Indexing them by the sample's k made them scratch memory, and reading the samples through a cursor made every
// load wait for the one before: 110 us per launch for 456 rays.)  Same operations in the same order as the reference's rows.
constexpr float SCAN_NONE = 1e30f;
constexpr int SCAN_RAYS = 512;                 // rays per pass = threads of k_scan
constexpr int SCAN_LD = MAX_DEPTH + 1;         // row stride in LDS: odd, so that the threads of a wave hit different banks
__device__ __forceinline__ int scan_ray(const float* __restrict__ row, int ray, int D, float d_min, float d_max, float th,
                                        float depth_obs, int32_t* e_rk, float* e_deds, float* e_res, int w) {
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
#pragma unroll
    for (int k = 0; k < MAX_DEPTH; ++k) {              // emission in ascending k
        if (k < D && ((inband >> k) & 1ull)) {
            const float de_do = Tl[k] / (1.f - occ[k]);
            if (de_do > 1e-2f) {
                if (e_rk) {
                    e_rk[w + n] = (ray << 6) | k;
                    e_deds[w + n] = de_do * delta_d * do_ds;
                    e_res[w + n] = res;
                }
                ++n;
            }
        }
    }
    return n;
}
__global__ __launch_bounds__(SCAN_RAYS) void k_scan(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                                    const float* __restrict__ depth, RefineCfg cfg,
                                                    const int32_t* __restrict__ valid_rk, int64_t rk_stride,
                                                    const int32_t* __restrict__ ray_voff, int64_t ray_stride,
                                                    const float* __restrict__ sdf_valid, int32_t* __restrict__ rend_rk,
                                                    float* __restrict__ rend_deds, float* __restrict__ rend_res) {
    const int h = blockIdx.x;
    HypState& S = st[h];
    if (!S.alive) return;
    extern __shared__ __attribute__((aligned(16))) float rows[];     // [SCAN_RAYS][SCAN_LD]
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
        __syncthreads();
        for (int e = threadIdx.x; e < SCAN_RAYS * SCAN_LD; e += SCAN_RAYS) rows[e] = SCAN_NONE;
        __syncthreads();
        const int r_end = min(base + SCAN_RAYS, ov.n_rays);
        const int v_beg = voff[base], v_end = voff[r_end];
        for (int v = v_beg + threadIdx.x; v < v_end; v += SCAN_RAYS) {
            const int e = rk[v];
            rows[((e >> 6) - base) * SCAN_LD + (e & 63)] = sdf[v];
        }
        __syncthreads();
        const int r = base + threadIdx.x;
        int n = 0;
        float dobs = 0.f;
        const float* row = rows + threadIdx.x * SCAN_LD;
        bool any = false;
        if (r < ov.n_rays) {
            any = voff[r + 1] > voff[r];
            dobs = (r < ov.n_fg) ? dep[r] : 1.1f * d_max;   // optimizer.py:153
            if (any) n = scan_ray(row, r, D, d_min, d_max, cfg.cut_off, dobs, nullptr, nullptr, nullptr, 0);
        }
        int tot;
        const int ex = block_excl_scan_256(n, sc, &tot);
        if (n > 0) scan_ray(row, r, D, d_min, d_max, cfg.cut_off, dobs, e_rk, e_deds, e_res, carry + ex);
        carry += tot;
    }
    if (threadIdx.x == 0) S.n_render = carry;
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
    // partial slot [h][slot][tile][32][32]
    if (wave < 6) {
        float* out = partials + ((int64_t)h * nw_total + slot) * PART_FLOATS + wave * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) out[acc_row(i, lane) * 32 + (lane & 31)] = hacc[i];
    }
    QSP_TSK(5)
    tsk_first = false;
  }
}

// the same kernel on the split-fp16 tile: four waves per workgroup (mlp_tile_h2<true>); the six J~^T J~ tiles on waves 0..3
// (waves 0 and 1 carry two)
template <int NR>      // NR point blocks of 32 per tile (QSP_DEC_OPT_TPOINTS)
__global__ __launch_bounds__(H2_THREADS) void k_mlp_jtj_h2(const HypState* __restrict__ st,
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
    constexpr int TP = 32 * NR, SUBS = H2_THREADS / TP;      // threads per Jacobian row
    bool staged = false;
    float amax = 0.f;
  for (;;) {                                   // work queue, see k_plan
    if (threadIdx.x == 0) s_item = atomicAdd(&qctl[3], 1);
    __syncthreads();                           // also: everybody is done with the previous item's LDS
    const int item = s_item;
    if (item >= n_items) break;                // the queue only grows towards n_items: every workgroup gets here
    const int h = work[item].x, slot = work[item].y;
    const HypState& S = st[h];
    const ObjView ov = objs[S.obj];
    const bool is_sdf = slot < nw_sdf;
    const int stride = is_sdf ? nw_sdf : nw_total - nw_sdf;
    const int j0 = is_sdf ? slot : slot - nw_sdf;
    const int n = is_sdf ? ov.n_pts : S.n_render;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // J~^T J~ accumulators of this wave's upper-triangular tiles: tile w on every wave, tile w + 4 on waves 0, 1
    // (tiles in the order (0,0) (0,1) (0,2) (1,1) (1,2) (2,2))
    f32x16 hacc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { hacc[0][i] = 0.f; hacc[1][i] = 0.f; }
    const int ta0 = wave < 3 ? 0 : 1, tb0 = wave < 3 ? wave : 1;
    const int ta1 = wave == 0 ? 1 : 2, tb1 = 2;

    stage_code_T(s, S, Tsh);
    for (int i = threadIdx.x; i < HID; i += H2_THREADS) {
        s.c0[i] = c0_all[(size_t)h * 2 * HID + i];
        s.c4[i] = c0_all[(size_t)h * 2 * HID + HID + i];
    }
    const float* Pc = pts + 3 * ov.pts_off;
    const float* R = rays + 3 * ov.ray_off;
    const int32_t* rk = rend_rk + h * rk_stride;
    const float* deds = rend_deds + h * rk_stride;
    const float* rres = rend_res + h * rk_stride;
    const uint8_t* active = pt_active ? pt_active + h * act_stride : nullptr;
    const float d_min = S.d_min, d_max = S.d_max;
    const float hub = is_sdf ? cfg.b2 : cfg.b1;

    for (int t = j0; t * TP < n; t += stride) {
        __syncthreads();
        if (tid < TP) {
            const int v = t * TP + tid;
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
        mlp_tile_h2<true, 2, false, NR>(s, P, amax, !staged);
        staged = true;
        // ---- Jacobian rows: J~[p] = [ s*(g_x . [I | -x^ | x]) (7) | s*g_z (64) | r~ ] -------------------------------
        // G (gradient w.r.t. [code | xyz]) sits in s.act with row stride LDG; J~ goes behind it.
        float* G = s.act;
        float* Jt = s.act + TILE_P * LDG;   /* (behind the 64-row G image whatever the tile size) */     // [64][LDJ]
        {
            const int p = tid / SUBS, sub = tid % SUBS;
            const float valid = s.xin[4 * p + 3];
            const float sc = s.rscale[p] * valid;
#pragma unroll
            for (int q = 0; q < CODE_LEN / SUBS; ++q) {
                const int c = sub + SUBS * q;        // code column 0..63
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
                if (res_out && is_sdf && valid != 0.f) res_out[h * act_stride + t * TP + p] = r;
            }
            if (sub == 1) {
#pragma unroll
                for (int c = NJ; c < LDJ; ++c) Jt[p * LDJ + c] = 0.f;
            }
        }
        __syncthreads();
        if (rows_out) {   // parity-test tap: the augmented Jacobian rows exactly as the MFMA below consumes them
            float* ro = rows_out + (int64_t)h * rows_stride * NJ + (int64_t)(is_sdf ? 0 : ov.n_pts) * NJ;
            for (int e = tid; e < TP * NJ; e += H2_THREADS) {
                const int p = e / NJ, c = e - p * NJ;
                const int v = t * TP + p;
                if (v < n) ro[(int64_t)v * NJ + c] = Jt[p * LDJ + c];
            }
        }
        {
            const float* A = Jt + (lane >> 5) * LDJ + 32 * ta0 + (lane & 31);
            const float* B = Jt + (lane >> 5) * LDJ + 32 * tb0 + (lane & 31);
#pragma unroll 8
            for (int ks = 0; ks < TP / 2; ++ks) hacc[0] = mfma32t<false>(A[2 * ks * LDJ], B[2 * ks * LDJ], hacc[0]);
        }
        if (wave < 2) {
            const float* A = Jt + (lane >> 5) * LDJ + 32 * ta1 + (lane & 31);
            const float* B = Jt + (lane >> 5) * LDJ + 32 * tb1 + (lane & 31);
#pragma unroll 8
            for (int ks = 0; ks < TP / 2; ++ks) hacc[1] = mfma32t<false>(A[2 * ks * LDJ], B[2 * ks * LDJ], hacc[1]);
        }
    }
    // partial slot [h][slot][tile][32][32]
    {
        float* out = partials + ((int64_t)h * nw_total + slot) * PART_FLOATS + wave * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) out[acc_row(i, lane) * 32 + (lane & 31)] = hacc[0][i];
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) out[4 * 1024 + acc_row(i, lane) * 32 + (lane & 31)] = hacc[1][i];
        }
    }
  }
    if (!(amax <= H2_MAX)) *P->range_flag = 1;
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

constexpr int SOLVE_THREADS = 1024;   // latency, not throughput: more loads in flight for the partial sums, shorter row strips per pivot
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(HypState* __restrict__ st, const ObjView* __restrict__ objs,
                                               RefineCfg cfg, const float* __restrict__ partials, int nw_sdf,
                                               int nw_total, const uint8_t* __restrict__ pt_active,
                                               int64_t act_stride, float* __restrict__ trH, float* __restrict__ trb,
                                               float* __restrict__ trdx, unsigned long long* __restrict__ counters) {
    const int h = blockIdx.x;
    HypState& S = st[h];
    if (!S.alive) return;
    if (threadIdx.x == 0 && counters) {   // work actually done this iteration (for the roofline figures)
        const ObjView o = objs[S.obj];
        atomicAdd(&counters[0], (unsigned long long)(o.n_pts + S.n_render));
        atomicAdd(&counters[1], (unsigned long long)S.n_valid);
        atomicAdd(&counters[2], (unsigned long long)((o.n_pts + cfg.tile_p - 1) / cfg.tile_p + (S.n_render + cfg.tile_p - 1) / cfg.tile_p));
        atomicAdd(&counters[3], (unsigned long long)((S.n_valid + TILE_P - 1) / TILE_P));
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
    const float M = (float)n_act_sh;
    const float Kf = (float)K;
    const int N = cfg.pose_only ? 6 : NH;
    // Fixed-order sum of the tile partials (deterministic), then H, b in f32 exactly as optimizer.py:217-252 orders the
    // operations; entries are promoted to f64 only for the linear solve.
    for (int e = tid; e < NJ * NJ; e += SOLVE_THREADS) {
        const int a = e / NJ, b = e % NJ;
        if (a > b) continue;
        const int off = tri_tile(a >> 5, b >> 5) * 1024 + (a & 31) * 32 + (b & 31);
        float ss = 0.f, sr = 0.f;
        // same left-to-right order as ever; unrolled so that 16 of the (24 KiB-strided) loads are in flight at a time
#pragma unroll 16
        for (int j = 0; j < n_sdf_slots; ++j) ss += base[(int64_t)j * PART_FLOATS + off];
#pragma unroll 4
        for (int j = 0; j < n_rend_slots; ++j) sr += base[(int64_t)(nw_sdf + j) * PART_FLOATS + off];
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
    const float loss_s = loss_sh[0], loss_r = loss_sh[1];
    const bool bad = isnan(loss_s) || isnan(loss_r);           // optimizer.py:168-169,193-194
    __syncthreads();
    if (bad) {
        if (tid == 0) S.alive = 0;
        return;
    }
    if (tid == 0 && !cfg.pose_only) {
        // rotation prior (loss.py:155-178) on the pose block, then damping (optimizer.py:240-252)
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
    // Gauss-Jordan elimination in f64 on the augmented system (reference: torch.inverse(H) @ b, f32).  H is symmetric
    // positive definite by construction (Gram matrices plus the identity damping of optimizer.py:240-252 / :75), so no
    // pivot search is needed: one barrier per column, every (row, column strip) pair on its own thread.
    const int STR = SOLVE_THREADS / N;         // 14 strips for the 71 x 71 system
    for (int c = 0; c < N; ++c) {
        const double inv = 1.0 / Hd[c * (N + 1) + c];
        const int r = tid / STR, q = tid - r * STR;
        if (r < N && r != c) {
            const double f = Hd[r * (N + 1) + c] * inv;
            for (int j = c + 1 + q; j <= N; j += STR) Hd[r * (N + 1) + j] -= f * Hd[c * (N + 1) + j];
        }
        __syncthreads();
    }
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
template <bool GRAD>
__global__ __launch_bounds__(H2_THREADS) void k_decode_h2(const float* __restrict__ code, const float* __restrict__ xyz, int64_t n,
                                                          const MlpParams* __restrict__ P, float* __restrict__ y_out,
                                                          float* __restrict__ grad_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    MlpSmem& s = *reinterpret_cast<MlpSmem*>(smem_raw);
    if (threadIdx.x < CODE_LEN) s.code[threadIdx.x] = code[threadIdx.x];
    __syncthreads();
    for (int u = threadIdx.x; u < HID; u += H2_THREADS) {      // mlp_prepare for 256 threads
        const float* w = P->w0c + (size_t)u * CODE_LEN;
        const float* w4 = P->w4c + (size_t)u * CODE_LEN;
        float a = P->bias[0][u], a4 = P->bias[4][u];
#pragma unroll 8
        for (int k = 0; k < CODE_LEN; ++k) {
            a += w[k] * s.code[k];
            a4 += w4[k] * s.code[k];
        }
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
        mlp_tile_h2<GRAD, 2>(s, P, amax, !staged);
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

// =================================================================================================================
// host side
// =================================================================================================================
using namespace qsp;

struct qsp_decoder {
    int device = 0;
    hipStream_t stream = nullptr;
    MlpParams P{};
    MlpParams* Pd = nullptr;   // device copy, read by the kernels
    std::vector<void*> allocs;
    double mac_per_point = 0;
    int code_len = CODE_LEN;   // the caller's code length L <= 64; the tile always works on 64 (columns L..63 are zero)
    int fwd_bf3 = 0;           // QSP_DEC_OPT_FORWARD_PRECISION: forward-only passes on the split-bf16 pipe (mlp_tile_bf3)
    int jac_bf3 = 0;           // QSP_DEC_OPT_JACOBIAN_PRECISION: the forward+backward pass (mlp_tile<true, .., B3>)
    bool fp16_ok = true;       // every weight of layers 0..7 fits fp16's range (split-fp16 planes are usable)
    int tile_p = 64;           // QSP_DEC_OPT_TILE_POINTS: points per MLP tile of the refinement batches created from now on
    int* range_flag_h = nullptr;   // host-mapped word the split-fp16 kernels set when a value left fp16's range (check_range)
};

// The family deep_sdf/deep_sdf_decoder.py:29-63 builds -- `dims` hidden layers of any width, one (or no) latent_in layer, any
// code length -- mapped EXACTLY onto the one network shape the tile kernels run (8 hidden layers x 512, code 64, skip into layer
// 4), so that specs.json decides the architecture, not the library:
//   * narrower layers: zero rows / columns (a unit with zero weights and zero bias outputs relu(0) = 0 and feeds nothing);
//   * fewer layers: identity layers.  Every hidden activation is a ReLU output, i.e. >= 0, and relu(1 * h + 0) == h bit for
//     bit (the other products of the row are exact zeros); in the backward pass the identity's ReLU mask (h > 0) only removes
//     gradient that the producing layer's own mask (pre-activation > 0, the same condition) removes anyway;
//   * shorter codes: zero code columns; the normal equations keep 64 code unknowns whose extra rows are decoupled (k_solve);
//   * the latent_in layer goes to slot 4, the layers before it to slots 0.., identity layers fill up to slot 3; the
//     layers after it to slots 5.., identity layers fill up to slot 7; the output layer is slot 8.
// Needs: at most 4 hidden layers before and at most 4 from the latent_in layer on, widths <= 512 (<= 445 in front of the skip).
// Cost: the arithmetic of the full 8 x 512 tile whatever the network's own size.
static int embed_family(const qsp_decoder_desc* desc, const std::vector<std::vector<float>>& W,
                        std::vector<std::vector<float>>& Wc, std::vector<std::vector<float>>& Bc) {
    const int nl = desc->n_layers, m = nl - 1, L = desc->code_len;
    if (nl < 3 || nl > 9 || L < 1 || L > CODE_LEN)
        return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: 2..8 hidden layers and a code of 1..64 are supported");
    int s = desc->latent_in_layer;
    const bool has_skip = s >= 0;
    if (!has_skip) s = std::max(1, m - 4);                 // no latent_in: any split with <= 4 layers on either side
    if (s < 1 || s > 4 || s > m - 1 || m - s > 4)
        return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: the latent_in layer needs 1..4 hidden layers in front of it and "
                                              "at most 4 from it on");
    if (desc->in_dim[0] != L + 3 || desc->out_dim[m] != 1)
        return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: first layer takes [code | xyz], last layer has one output");
    for (int l = 0; l < m; ++l) {
        const int lim = (l == s - 1 && has_skip) ? SKIP_COL : HID;
        if (desc->out_dim[l] < 1 || desc->out_dim[l] > lim)
            return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: hidden width above 512 (445 in front of the latent_in layer)");
    }
    for (int l = 1; l <= m; ++l) {
        const int expect = desc->out_dim[l - 1] + ((l == s && has_skip) ? L + 3 : 0);
        if (desc->in_dim[l] != expect) return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: layer input widths do not chain");
    }
    auto in_c = [](int c) { return c == 0 ? NIN : HID; };
    auto out_c = [](int c) { return c == 8 ? 1 : (c == 3 ? SKIP_COL : HID); };
    Wc.assign(9, std::vector<float>());
    Bc.assign(9, std::vector<float>());
    std::vector<int> src(9, -1);                            // canonical slot -> source layer (-1: identity)
    for (int l = 0; l < s; ++l) src[l] = l;
    for (int l = s; l < m; ++l) src[4 + (l - s)] = l;
    src[8] = m;
    for (int c = 0; c < 9; ++c) {
        const int ic = in_c(c), oc = out_c(c);
        Wc[c].assign((size_t)oc * ic, 0.f);
        Bc[c].assign((size_t)oc, 0.f);
        const int l = src[c];
        if (l < 0) {                                        // identity on the (<= 445 / 512 wide) activation
            for (int r = 0; r < oc; ++r) Wc[c][(size_t)r * ic + r] = 1.f;
            continue;
        }
        const int in = desc->in_dim[l], out = desc->out_dim[l];
        for (int o = 0; o < out; ++o) {
            Bc[c][o] = desc->bias[l][o];
            for (int k = 0; k < in; ++k) {
                int kc = k;
                if (c == 0) kc = (k < L) ? k : CODE_LEN + (k - L);                       // [code L | xyz] -> [code 64 | xyz]
                if (c == 4 && has_skip) {
                    const int w = desc->out_dim[l - 1];                                  // [prev w | code L | xyz]
                    kc = (k < w) ? k : (k < w + L ? SKIP_COL + (k - w) : SKIP_COL + CODE_LEN + (k - w - L));
                }
                Wc[c][(size_t)o * ic + kc] = W[l][(size_t)o * in + k];
            }
        }
    }
    return QSP_OK;
}

static inline uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);       // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_f32(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
// x = hi + mid + lo, each a bf16: 8 + 8 + 8 mantissa bits
static inline void bf16_split3(float v, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
    hi = bf16_rne(v);
    const float r1 = v - bf16_f32(hi);
    mid = bf16_rne(r1);
    const float r2 = r1 - bf16_f32(mid);
    lo = bf16_rne(r2);
}

static int pack_weights(qsp_decoder* d, const qsp_decoder_desc* desc) {
    if (desc->n_layers < 1 || desc->n_layers > 64) return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: layer count");
    // fold weight norm:  W = g * v / ||v||_row   (torch.nn.utils.weight_norm, dim=0)
    std::vector<std::vector<float>> Wsrc(desc->n_layers);
    for (int l = 0; l < desc->n_layers; ++l) {
        const int in = desc->in_dim[l], out = desc->out_dim[l];
        if (in < 1 || out < 1 || in > 4096 || out > 4096 || !desc->weight[l] || !desc->bias[l])
            return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: layer dims");
        Wsrc[l].assign((size_t)out * in, 0.f);
        const float* v = desc->weight[l];
        const float* g = (desc->weight_g && desc->weight_g[l]) ? desc->weight_g[l] : nullptr;
        for (int o = 0; o < out; ++o) {
            float sc = 1.f;
            if (g) {
                float ss = 0.f;
                for (int k = 0; k < in; ++k) ss += v[(size_t)o * in + k] * v[(size_t)o * in + k];
                sc = g[o] / sqrtf(ss);
            }
            for (int k = 0; k < in; ++k) Wsrc[l][(size_t)o * in + k] = v[(size_t)o * in + k] * sc;
        }
        d->mac_per_point += (double)in * out;
    }
    std::vector<std::vector<float>> W, Bias;
    {
        const int rc = embed_family(desc, Wsrc, W, Bias);
        if (rc) return rc;
    }
    d->code_len = desc->code_len;
    // from here on: the canonical 9-layer shape
    const int in_dim[9] = {NIN, HID, HID, HID, HID, HID, HID, HID, HID};
    const int out_dim[9] = {HID, HID, HID, SKIP_COL, HID, HID, HID, HID, 1};
    auto upload = [&](const std::vector<float>& h, const void** dst) -> int {
        void* p = nullptr;
        QSP_HIP(hipMalloc(&p, h.size() * sizeof(float) + 16384));   // + prefetch over-read slack (gemm_2x2)
        d->allocs.push_back(p);
        QSP_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        *dst = p;
        return QSP_OK;
    };
    for (int l = 0; l < 8; ++l) {
        const int in = in_dim[l], out = out_dim[l];
        int rc = QSP_OK;
        if (l == 0) {
            // layer 0 is evaluated directly (mlp_prepare + mlp_tile): code columns row-major, xyz columns per unit quad
            std::vector<float> wc((size_t)HID * CODE_LEN), wx((size_t)(HID / 4) * 3 * 4);
            for (int o = 0; o < HID; ++o)
                for (int k = 0; k < CODE_LEN; ++k) wc[(size_t)o * CODE_LEN + k] = W[0][(size_t)o * in + k];
            for (int q = 0; q < HID / 4; ++q)
                for (int a = 0; a < 3; ++a)
                    for (int e = 0; e < 4; ++e) wx[((size_t)q * 3 + a) * 4 + e] = W[0][(size_t)(4 * q + e) * in + CODE_LEN + a];
            rc = upload(wc, (const void**)&d->P.w0c);
            if (!rc) rc = upload(wx, (const void**)&d->P.w0x);
            d->P.wf[0] = nullptr;
            d->P.wf3[0] = nullptr;
            if (!rc) {   // split-fp16: the xyz columns as one slab of the forward product (gemm_l0_h2): k = 0..2 used of 16
                std::vector<_Float16> ph((size_t)16 * 2 * 64 * 8, (_Float16)0.f);
                for (int cb = 0; cb < 16; ++cb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int o = 32 * cb + (lane & 31), k = 8 * (lane >> 5) + j;
                            float v = 0.f;
                            if (o < out && k < 3) v = W[0][(size_t)o * in + CODE_LEN + k];
                            if (!(fabsf(v) < 65000.f)) d->fp16_ok = false;
                            const _Float16 hi = (_Float16)v;
                            const size_t base = ((size_t)cb * 2) * 64 * 8 + (size_t)lane * 8 + j;
                            ph[base] = hi;
                            ph[base + 64 * 8] = (_Float16)((v - (float)hi) * 2048.f);
                        }
                void* phd = nullptr;
                QSP_HIP(hipMalloc(&phd, ph.size() * sizeof(_Float16) + 16384));
                d->allocs.push_back(phd);
                QSP_HIP(hipMemcpy(phd, ph.data(), ph.size() * sizeof(_Float16), hipMemcpyHostToDevice));
                d->P.wfh[0] = (const float4*)phd;
            }
        } else {
            // forward: B[k][o]; column blocks over o (16 blocks of 32), k-groups of 8 over K = 512.
            // Layer 4 (latent_in): K = 448 = [h3 (445) | xyz (3)]; its 64 code columns go to w4c (folded into a bias per
            // hypothesis by k_c0 / mlp_prepare).
            const int KG = (l == 4) ? KG4 : HID / 8;
            std::vector<float> pf((size_t)16 * KG * 64 * 4, 0.f);
            for (int cb = 0; cb < 16; ++cb)
                for (int kg = 0; kg < KG; ++kg)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int o = 32 * cb + (lane & 31);
                            int k = 8 * kg + 4 * (lane >> 5) + e;
                            if (l == 4 && k >= SKIP_COL) k += CODE_LEN;      // xyz columns 509..511
                            float v = 0.f;
                            if (o < out && k < in) v = W[l][(size_t)o * in + k];
                            pf[(((size_t)cb * KG + kg) * 64 + lane) * 4 + e] = v;
                        }
            rc = upload(pf, (const void**)&d->P.wf[l]);
            if (!rc) {
                // split-bf16 planes for mlp_tile_bf3: [col block][slab of 16 k][plane][lane][8 bf16]; lane (r, h) holds
                // W[unit 32 cb + r][k = 16 s + 8 h + j] (the A-operand map of v_mfma_f32_32x32x16_bf16)
                const int KS = KG / 2;
                std::vector<uint16_t> p3((size_t)16 * KS * 3 * 64 * 8, 0);
                for (int cb = 0; cb < 16; ++cb)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int o = 32 * cb + (lane & 31);
                                int k = 16 * ks + 8 * (lane >> 5) + j;
                                if (l == 4 && k >= SKIP_COL) k += CODE_LEN;
                                float v = 0.f;
                                if (o < out && k < in) v = W[l][(size_t)o * in + k];
                                uint16_t hi, mid, lo;
                                bf16_split3(v, hi, mid, lo);
                                const size_t base = (((size_t)cb * KS + ks) * 3) * 64 * 8 + (size_t)lane * 8 + j;
                                p3[base] = hi;
                                p3[base + 64 * 8] = mid;
                                p3[base + 2 * 64 * 8] = lo;
                            }
                void* p3d = nullptr;
                QSP_HIP(hipMalloc(&p3d, p3.size() * sizeof(uint16_t) + 16384));
                d->allocs.push_back(p3d);
                QSP_HIP(hipMemcpy(p3d, p3.data(), p3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                d->P.wf3[l] = (const float4*)p3d;
            }
            if (!rc) {
                // split-fp16 planes for mlp_tile_h2: [col block][slab of 16 k][hi | lo' = (w - hi) 2^11][lane][8 fp16], same lane map
                const int KS = KG / 2;
                std::vector<_Float16> ph((size_t)16 * KS * 2 * 64 * 8, (_Float16)0.f);
                for (int cb = 0; cb < 16; ++cb)
                    for (int ks = 0; ks < KS; ++ks)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int o = 32 * cb + (lane & 31);
                                int k = 16 * ks + 8 * (lane >> 5) + j;
                                if (l == 4 && k >= SKIP_COL) k += CODE_LEN;
                                float v = 0.f;
                                if (o < out && k < in) v = W[l][(size_t)o * in + k];
                                if (!(fabsf(v) < 65000.f)) d->fp16_ok = false;
                                const _Float16 hi = (_Float16)v;
                                const _Float16 lo = (_Float16)((v - (float)hi) * 2048.f);
                                const size_t base = (((size_t)cb * KS + ks) * 2) * 64 * 8 + (size_t)lane * 8 + j;
                                ph[base] = hi;
                                ph[base + 64 * 8] = lo;
                            }
                void* phd = nullptr;
                QSP_HIP(hipMalloc(&phd, ph.size() * sizeof(_Float16) + 16384));
                d->allocs.push_back(phd);
                QSP_HIP(hipMemcpy(phd, ph.data(), ph.size() * sizeof(_Float16), hipMemcpyHostToDevice));
                d->P.wfh[l] = (const float4*)phd;
            }
            if (!rc && l == 4) {
                std::vector<float> wc((size_t)HID * CODE_LEN);
                for (int o = 0; o < HID; ++o)
                    for (int k = 0; k < CODE_LEN; ++k) wc[(size_t)o * CODE_LEN + k] = W[4][(size_t)o * in + SKIP_COL + k];
                rc = upload(wc, (const void**)&d->P.w4c);
            }
        }
        if (rc) return rc;
        // backward: B[o][k]; column blocks over the layer's inputs k, groups of 8 over o (padded to 512)
        const int NCB = (l == 0) ? 3 : 16;
        const int OG = HID / 8;
        std::vector<float> pb((size_t)NCB * OG * 64 * 4, 0.f);
        for (int cb = 0; cb < NCB; ++cb)
            for (int og = 0; og < OG; ++og)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int k = 32 * cb + (lane & 31);
                        const int o = 8 * og + 4 * (lane >> 5) + e;
                        float v = 0.f;
                        if (o < out && k < in) v = W[l][(size_t)o * in + k];
                        pb[(((size_t)cb * OG + og) * 64 + lane) * 4 + e] = v;
                    }
        rc = upload(pb, (const void**)&d->P.wb[l]);
        if (rc) return rc;
        {   // split-bf16 planes of the same matrix: [col block over inputs k][slab of 16 outputs o][plane][lane][8 bf16]
            const int KS = HID / 16;
            std::vector<uint16_t> p3((size_t)NCB * KS * 3 * 64 * 8, 0);
            for (int cb = 0; cb < NCB; ++cb)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int k = 32 * cb + (lane & 31);
                            const int o = 16 * ks + 8 * (lane >> 5) + j;
                            float v = 0.f;
                            if (o < out && k < in) v = W[l][(size_t)o * in + k];
                            uint16_t hi, mid, lo;
                            bf16_split3(v, hi, mid, lo);
                            const size_t base = (((size_t)cb * KS + ks) * 3) * 64 * 8 + (size_t)lane * 8 + j;
                            p3[base] = hi;
                            p3[base + 64 * 8] = mid;
                            p3[base + 2 * 64 * 8] = lo;
                        }
            void* p3d = nullptr;
            QSP_HIP(hipMalloc(&p3d, p3.size() * sizeof(uint16_t) + 16384));
            d->allocs.push_back(p3d);
            QSP_HIP(hipMemcpy(p3d, p3.data(), p3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            d->P.wb3[l] = (const float4*)p3d;
        }
        {   // split-fp16 planes of the same matrix: [col block over inputs k][slab of 16 outputs o][hi | lo'][lane][8 fp16]
            const int KS = HID / 16;
            std::vector<_Float16> ph((size_t)NCB * KS * 2 * 64 * 8, (_Float16)0.f);
            for (int cb = 0; cb < NCB; ++cb)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int k = 32 * cb + (lane & 31);
                            const int o = 16 * ks + 8 * (lane >> 5) + j;
                            float v = 0.f;
                            if (o < out && k < in) v = W[l][(size_t)o * in + k];
                            if (!(fabsf(v) < 65000.f)) d->fp16_ok = false;
                            const _Float16 hi = (_Float16)v;
                            const size_t base = (((size_t)cb * KS + ks) * 2) * 64 * 8 + (size_t)lane * 8 + j;
                            ph[base] = hi;
                            ph[base + 64 * 8] = (_Float16)((v - (float)hi) * 2048.f);
                        }
            void* phd = nullptr;
            QSP_HIP(hipMalloc(&phd, ph.size() * sizeof(_Float16) + 16384));
            d->allocs.push_back(phd);
            QSP_HIP(hipMemcpy(phd, ph.data(), ph.size() * sizeof(_Float16), hipMemcpyHostToDevice));
            d->P.wbh[l] = (const float4*)phd;
        }
        std::vector<float> bias(HID, 0.f);
        for (int o = 0; o < out; ++o) bias[o] = Bias[l][o];
        rc = upload(bias, (const void**)&d->P.bias[l]);
        if (rc) return rc;
    }
    {   // layer 4's skip columns (inputs 445..511 = [code | xyz]) as their own backward matrix, packed like wbh[0]:
        // [col block over ci = input - 445 (3 blocks, 67 used)][slab of 16 outputs o][hi | lo'][lane][8 fp16]
        const int KS = HID / 16, in4 = in_dim[4], out4 = out_dim[4];
        std::vector<_Float16> ph((size_t)3 * KS * 2 * 64 * 8, (_Float16)0.f);
        for (int cb = 0; cb < 3; ++cb)
            for (int ks = 0; ks < KS; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int ci = 32 * cb + (lane & 31);
                        const int o = 16 * ks + 8 * (lane >> 5) + j;
                        float v = 0.f;
                        if (o < out4 && ci < NIN) v = W[4][(size_t)o * in4 + SKIP_COL + ci];
                        const _Float16 hi = (_Float16)v;
                        const size_t base = (((size_t)cb * KS + ks) * 2) * 64 * 8 + (size_t)lane * 8 + j;
                        ph[base] = hi;
                        ph[base + 64 * 8] = (_Float16)((v - (float)hi) * 2048.f);
                    }
        void* phd = nullptr;
        QSP_HIP(hipMalloc(&phd, ph.size() * sizeof(_Float16) + 16384));
        d->allocs.push_back(phd);
        QSP_HIP(hipMemcpy(phd, ph.data(), ph.size() * sizeof(_Float16), hipMemcpyHostToDevice));
        d->P.wbh4s = (const float4*)phd;
    }
    std::vector<float> w8(W[8].begin(), W[8].end());
    int rc = upload(w8, (const void**)&d->P.w8);
    if (rc) return rc;
    d->P.b8 = Bias[8][0];
    QSP_HIP(hipHostMalloc((void**)&d->range_flag_h, sizeof(int), hipHostMallocMapped));
    *d->range_flag_h = 0;
    QSP_HIP(hipHostGetDevicePointer((void**)&d->P.range_flag, d->range_flag_h, 0));
    void* pd = nullptr;
    QSP_HIP(hipMalloc(&pd, sizeof(MlpParams)));
    d->allocs.push_back(pd);
    QSP_HIP(hipMemcpy(pd, &d->P, sizeof(MlpParams), hipMemcpyHostToDevice));
    d->Pd = (MlpParams*)pd;
    return QSP_OK;
}

// after a synchronisation of the decoder's stream: did a split-fp16 kernel meet a value it cannot represent?
static int check_range(qsp_decoder* d) {
    if (d->range_flag_h && *d->range_flag_h) {
        *d->range_flag_h = 0;
        return qsp_fail(QSP_ERR_UNSUPPORTED, "split fp16: an activation or gradient of this decoder left fp16's range (65504); "
                                             "use the split-bf16 or f32 precision for it");
    }
    return QSP_OK;
}

static int mlp_attr_once() {
    static bool done = false;
    if (done) return QSP_OK;
    const int bytes = (int)sizeof(MlpSmem);
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd_h2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_h2<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_h2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * SCAN_RAYS * SCAN_LD)));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done = true;
    return QSP_OK;
}

extern "C" int qsp_decoder_create(const qsp_decoder_desc* desc, int device, qsp_decoder** out) {
    if (!desc || !out || !desc->in_dim || !desc->out_dim || !desc->weight || !desc->bias)
        return qsp_fail(QSP_ERR_INVALID, "qsp_decoder_create: null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return qsp_fail(QSP_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return qsp_fail(QSP_ERR_INVALID, "device index out of range");
    QSP_HIP(hipSetDevice(device));
    qsp_decoder* d = new qsp_decoder();
    d->device = device;
    int rc = mlp_attr_once();
    if (!rc) rc = pack_weights(d, desc);
    if (!rc) {
        hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
        if (e != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
    }
    if (rc) {
        qsp_decoder_destroy(d);
        return rc;
    }
    *out = d;
    return QSP_OK;
}

extern "C" int qsp_decoder_set_option(qsp_decoder* d, int32_t option, int32_t value) {
    if (!d) return qsp_fail(QSP_ERR_INVALID, "qsp_decoder_set_option: null decoder");
    switch (option) {
        case QSP_DEC_OPT_FORWARD_PRECISION:
            if (value < 0 || value > 2) return qsp_fail(QSP_ERR_INVALID, "forward precision: 0 (f32 MFMA), 1 (split bf16) or 2 (split fp16)");
            if (value == 2 && !d->fp16_ok) return qsp_fail(QSP_ERR_UNSUPPORTED, "split fp16: a weight of this decoder is outside fp16's range");
            d->fwd_bf3 = value;
            return QSP_OK;
        case QSP_DEC_OPT_JACOBIAN_PRECISION:
            if (value < 0 || value > 2) return qsp_fail(QSP_ERR_INVALID, "jacobian precision: 0 (f32 MFMA), 1 (split bf16) or 2 (split fp16)");
            if (value == 2 && !d->fp16_ok) return qsp_fail(QSP_ERR_UNSUPPORTED, "split fp16: a weight of this decoder is outside fp16's range");
            d->jac_bf3 = value;
            return QSP_OK;
        case QSP_DEC_OPT_TILE_POINTS:
            if (value != 32 && value != 64) return qsp_fail(QSP_ERR_INVALID, "tile points: 64 (default) or 32");
            d->tile_p = value;
            return QSP_OK;
        default: return qsp_fail(QSP_ERR_INVALID, "qsp_decoder_set_option: unknown option");
    }
}

extern "C" void qsp_decoder_destroy(qsp_decoder* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    for (void* p : d->allocs) (void)hipFree(p);
    if (d->range_flag_h) (void)hipHostFree(d->range_flag_h);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

static int decode_common(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* y, float* grad) {
    if (!d || !code || !xyz || n < 0 || !y) return qsp_fail(QSP_ERR_INVALID, "decode: bad argument");
    if (n == 0) return QSP_OK;
    QSP_HIP(hipSetDevice(d->device));
    float *dc = nullptr, *dx = nullptr, *dy = nullptr, *dg = nullptr;
    QSP_HIP(hipMalloc((void**)&dc, CODE_LEN * sizeof(float)));
    QSP_HIP(hipMalloc((void**)&dx, n * 3 * sizeof(float)));
    QSP_HIP(hipMalloc((void**)&dy, n * sizeof(float)));
    if (grad) QSP_HIP(hipMalloc((void**)&dg, n * NIN * sizeof(float)));
    float code64[CODE_LEN] = {};                       // the caller's code has d->code_len entries
    memcpy(code64, code, sizeof(float) * d->code_len);
    QSP_HIP(hipMemcpyAsync(dc, code64, CODE_LEN * sizeof(float), hipMemcpyHostToDevice, d->stream));
    QSP_HIP(hipMemcpyAsync(dx, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, d->stream));
    const int64_t tiles = (n + TILE_P - 1) / TILE_P;
    const int grid = (int)std::min<int64_t>(tiles, 4096);
    if (grad && d->jac_bf3 == 2)
        hipLaunchKernelGGL(k_decode_h2<true>, dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy, dg);
    else if (grad && d->jac_bf3)
        hipLaunchKernelGGL((k_decode<true, true>), dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy, dg);
    else if (grad)
        hipLaunchKernelGGL(k_decode<true>, dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy, dg);
    else if (d->fwd_bf3 == 2)
        hipLaunchKernelGGL(k_decode_h2<false>, dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy,
                           (float*)nullptr);
    else if (d->fwd_bf3)
        hipLaunchKernelGGL((k_decode<false, true>), dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy,
                           (float*)nullptr);
    else
        hipLaunchKernelGGL(k_decode<false>, dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy,
                           (float*)nullptr);
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipMemcpyAsync(y, dy, n * sizeof(float), hipMemcpyDeviceToHost, d->stream));
    std::vector<float> g67;
    if (grad) {
        if (d->code_len == CODE_LEN) {
            QSP_HIP(hipMemcpyAsync(grad, dg, n * NIN * sizeof(float), hipMemcpyDeviceToHost, d->stream));
        } else {
            g67.resize((size_t)n * NIN);
            QSP_HIP(hipMemcpyAsync(g67.data(), dg, n * NIN * sizeof(float), hipMemcpyDeviceToHost, d->stream));
        }
    }
    QSP_HIP(hipStreamSynchronize(d->stream));
    {
        const int rc = check_range(d);
        if (rc) {
            (void)hipFree(dc);
            (void)hipFree(dx);
            (void)hipFree(dy);
            if (dg) (void)hipFree(dg);
            return rc;
        }
    }
    if (grad && d->code_len != CODE_LEN) {               // [code 64 | xyz] -> [code L | xyz]
        const int L = d->code_len;
        for (int64_t i = 0; i < n; ++i) {
            memcpy(grad + i * (L + 3), g67.data() + i * NIN, sizeof(float) * L);
            memcpy(grad + i * (L + 3) + L, g67.data() + i * NIN + CODE_LEN, sizeof(float) * 3);
        }
    }
    (void)hipFree(dc);
    (void)hipFree(dx);
    (void)hipFree(dy);
    if (dg) (void)hipFree(dg);
    return QSP_OK;
}

extern "C" int qsp_decode_sdf(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* sdf_out) {
    return decode_common(d, code, xyz, n, sdf_out, nullptr);
}

extern "C" int qsp_sdf_value_grad(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* y, float* grad) {
    if (!grad) return qsp_fail(QSP_ERR_INVALID, "qsp_sdf_value_grad: grad is null");
    return decode_common(d, code, xyz, n, y, grad);
}

// ---------------------------------------------------------------------------------------------------------------
// refinement batch
// ---------------------------------------------------------------------------------------------------------------
struct qsp_refine_batch {
    qsp_decoder* dec = nullptr;
    int device = 0;                 // of the decoder, cached: destroy must not touch a decoder that may already be gone
    RefineCfg cfg{};
    int code_len = CODE_LEN;        // dec->code_len, cached like `device`
    int n_iter_cfg = 5;
    int n_obj = 0, n_hyp = 0;
    int max_pts = 0, max_rays = 0;
    int nw_sdf = 1;
    int64_t rk_stride = 0, ray_stride = 0, act_stride = 0;
    std::vector<ObjView> objs_h;
    std::vector<int32_t> hyp_obj;
    // device
    HypState* st = nullptr;
    ObjView* objs = nullptr;
    float *pts = nullptr, *rays = nullptr, *depth = nullptr;
    int32_t *valid_rk = nullptr, *ray_voff = nullptr, *rend_rk = nullptr;
    float *sdf_valid = nullptr, *rend_deds = nullptr, *rend_res = nullptr, *partials = nullptr;
    float *trH = nullptr, *trb = nullptr, *trdx = nullptr;
    uint8_t* pt_active = nullptr;   // pose-only mode
    float* res_buf = nullptr;       // pose-only mode: per-point residual of the current iteration
    unsigned long long* counters = nullptr;   // [4] points/tiles processed (fwd+bwd, fwd-only)
    int2 *work_fwd = nullptr, *work_jtj = nullptr;   // work-queue items (k_plan)
    int* qctl = nullptr;            // [4] item counts / next-item counters
    float* c0_all = nullptr;        // [n_hyp][2][512] code part of layers 0 and 4 (bias included) per hypothesis (k_c0)
    int n_cu = 256;
    float* rows = nullptr;          // optional tap of the Jacobian rows (qsp_refine_batch_rows)
    int64_t rows_stride = 0;
    // profiling
    bool prof = false;
    qsp_refine_profile profile{};
    std::vector<hipEvent_t> ev;
};

static void batch_free(qsp_refine_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    void* ptrs[] = {b->st, b->objs, b->pts, b->rays, b->depth, b->valid_rk, b->ray_voff, b->rend_rk, b->sdf_valid,
                    b->rend_deds, b->rend_res, b->partials, b->trH, b->trb, b->trdx, b->pt_active, b->res_buf, b->rows, b->counters,
                    b->work_fwd, b->work_jtj, b->qctl, b->c0_all};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
    delete b;
    (void)hipGetLastError();   // errors are ignored here; do not leave one behind for the next call's launch check
}

static int batch_create(qsp_decoder* dec, const RefineCfg& cfg, int n_iter, int32_t n_obj, const float* const* pts,
                        const int32_t* n_pts, const float* const* rays, const int32_t* n_rays,
                        const float* const* depth, const int32_t* n_fg, int32_t n_hyp, const int32_t* hyp_obj,
                        qsp_refine_batch** out, bool device_fill = false) {
    // device_fill: only the extents are given, the observation arrays are written by a kernel (detections.hpp)
    if (!dec || !out || n_obj <= 0 || n_hyp <= 0 || (!pts && !device_fill) || !n_pts || !hyp_obj)
        return qsp_fail(QSP_ERR_INVALID, "refine batch: bad argument");
    if (!cfg.pose_only && ((!device_fill && (!rays || !depth)) || !n_rays || !n_fg))
        return qsp_fail(QSP_ERR_INVALID, "refine batch: rays missing");
    if (cfg.n_depth < 2 || cfg.n_depth > MAX_DEPTH) return qsp_fail(QSP_ERR_INVALID, "n_depth must be in [2, 64]");
    if (dec->tile_p == 32 && (dec->fwd_bf3 != 2 || dec->jac_bf3 != 2))
        return qsp_fail(QSP_ERR_UNSUPPORTED, "32-point tiles (QSP_DEC_OPT_TILE_POINTS) exist on the split-fp16 pipe only: set both "
                                             "precisions to 2 first");
    const int tile_p = dec->tile_p;
    QSP_HIP(hipSetDevice(dec->device));
    qsp_refine_batch* b = new qsp_refine_batch();
    b->dec = dec;
    b->device = dec->device;
    b->code_len = dec->code_len;
    b->cfg = cfg;
    b->cfg.code_len = dec->code_len;
    b->cfg.tile_p = dec->tile_p;
    b->n_iter_cfg = n_iter;
    b->n_obj = n_obj;
    b->n_hyp = n_hyp;
    b->objs_h.resize(n_obj);
    int64_t po = 0, ro = 0;
    for (int o = 0; o < n_obj; ++o) {
        const int nr = cfg.pose_only ? 0 : n_rays[o];
        const int nf = cfg.pose_only ? 0 : n_fg[o];
        if (n_pts[o] < 0 || nr < 0 || nf < 0 || nf > nr || nr >= (1 << 25)) {
            delete b;
            return qsp_fail(QSP_ERR_INVALID, "refine batch: bad per-object counts");
        }
        b->objs_h[o] = ObjView{po, ro, n_pts[o], nr, nf, 0};
        po += n_pts[o];
        ro += nr;
        b->max_pts = std::max(b->max_pts, (int)n_pts[o]);
        b->max_rays = std::max(b->max_rays, nr);
    }
    b->hyp_obj.assign(hyp_obj, hyp_obj + n_hyp);
    for (int h = 0; h < n_hyp; ++h)
        if (hyp_obj[h] < 0 || hyp_obj[h] >= n_obj) {
            delete b;
            return qsp_fail(QSP_ERR_INVALID, "refine batch: hyp_obj out of range");
        }
    std::vector<float> hp((size_t)std::max<int64_t>(po, 1) * 3), hr((size_t)std::max<int64_t>(ro, 1) * 3),
        hd((size_t)std::max<int64_t>(ro, 1), 0.f);
    for (int o = 0; o < n_obj && !device_fill; ++o) {
        const ObjView& v = b->objs_h[o];
        if (v.n_pts) memcpy(&hp[3 * v.pts_off], pts[o], sizeof(float) * 3 * v.n_pts);
        if (v.n_rays) memcpy(&hr[3 * v.ray_off], rays[o], sizeof(float) * 3 * v.n_rays);
        if (v.n_fg) memcpy(&hd[v.ray_off], depth[o], sizeof(float) * v.n_fg);
    }
    b->nw_sdf = std::max(1, std::min(NW_SDF_MAX, (b->max_pts + tile_p - 1) / tile_p));
    b->rk_stride = (int64_t)std::max(1, b->max_rays) * cfg.n_depth;
    b->ray_stride = b->max_rays + 1;
    b->act_stride = std::max(1, b->max_pts);
    const int nw_total = b->nw_sdf + (cfg.pose_only ? 0 : NW_REND);
    int rc = QSP_OK;
#define QSP_ALLOC(ptr, bytes)                                                                 \
    if (!rc) {                                                                                \
        hipError_t e_ = hipMalloc((void**)&(ptr), (size_t)(bytes));                           \
        if (e_ != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e_));           \
    }
    QSP_ALLOC(b->st, sizeof(HypState) * n_hyp);
    QSP_ALLOC(b->objs, sizeof(ObjView) * n_obj);
    QSP_ALLOC(b->pts, hp.size() * sizeof(float));
    QSP_ALLOC(b->rays, hr.size() * sizeof(float));
    QSP_ALLOC(b->depth, hd.size() * sizeof(float));
    QSP_ALLOC(b->partials, sizeof(float) * (size_t)n_hyp * nw_total * PART_FLOATS);
    QSP_ALLOC(b->trH, sizeof(float) * (size_t)n_hyp * NH * NH);
    QSP_ALLOC(b->trb, sizeof(float) * (size_t)n_hyp * NH);
    QSP_ALLOC(b->trdx, sizeof(float) * (size_t)n_hyp * NH);
    QSP_ALLOC(b->counters, sizeof(unsigned long long) * 4);
    QSP_ALLOC(b->qctl, sizeof(int) * 4);
    QSP_ALLOC(b->c0_all, sizeof(float) * (size_t)n_hyp * 2 * HID);
    QSP_ALLOC(b->work_jtj, sizeof(int2) * (size_t)n_hyp * nw_total);
    if (!cfg.pose_only) QSP_ALLOC(b->work_fwd, sizeof(int2) * (size_t)n_hyp * ((b->rk_stride + TILE_P - 1) / TILE_P));
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dec->device) == hipSuccess && ncu > 0) b->n_cu = ncu;
    }
    if (!cfg.pose_only) {
        QSP_ALLOC(b->valid_rk, sizeof(int32_t) * (size_t)n_hyp * b->rk_stride);
        QSP_ALLOC(b->ray_voff, sizeof(int32_t) * (size_t)n_hyp * b->ray_stride);
        QSP_ALLOC(b->rend_rk, sizeof(int32_t) * (size_t)n_hyp * b->rk_stride);
        QSP_ALLOC(b->sdf_valid, sizeof(float) * (size_t)n_hyp * b->rk_stride);
        QSP_ALLOC(b->rend_deds, sizeof(float) * (size_t)n_hyp * b->rk_stride);
        QSP_ALLOC(b->rend_res, sizeof(float) * (size_t)n_hyp * b->rk_stride);
    } else {
        QSP_ALLOC(b->pt_active, (size_t)n_hyp * b->act_stride);
        QSP_ALLOC(b->res_buf, sizeof(float) * (size_t)n_hyp * b->act_stride);
    }
#undef QSP_ALLOC
    if (!rc) {
        hipError_t e = hipMemcpy(b->objs, b->objs_h.data(), sizeof(ObjView) * n_obj, hipMemcpyHostToDevice);
        if (e == hipSuccess && !device_fill) e = hipMemcpy(b->pts, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess && !device_fill) e = hipMemcpy(b->rays, hr.data(), hr.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = device_fill ? hipMemset(b->depth, 0, hd.size() * sizeof(float))
                                             : hipMemcpy(b->depth, hd.data(), hd.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(b->trH, 0, sizeof(float) * (size_t)n_hyp * NH * NH);
        if (e == hipSuccess) e = hipMemset(b->trb, 0, sizeof(float) * (size_t)n_hyp * NH);
        if (e == hipSuccess) e = hipMemset(b->trdx, 0, sizeof(float) * (size_t)n_hyp * NH);
        if (e != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
    }
    if (rc) {
        batch_free(b);
        return rc;
    }
    *out = b;
    return QSP_OK;
}

extern "C" int qsp_refine_batch_create(qsp_decoder* dec, const qsp_joint_cfg* cfg, int32_t n_obj, const float* const* pts,
                                       const int32_t* n_pts, const float* const* rays, const int32_t* n_rays,
                                       const float* const* depth, const int32_t* n_fg, int32_t n_hyp,
                                       const int32_t* hyp_obj, qsp_refine_batch** out) {
    if (!cfg) return qsp_fail(QSP_ERR_INVALID, "cfg is null");
    if (cfg->code_len != dec->code_len) return qsp_fail(QSP_ERR_INVALID, "code_len of the optimizer config differs from the decoder's");
    RefineCfg c{cfg->k1, cfg->k2, cfg->k3, cfg->k4, cfg->b1, cfg->b2, cfg->lr, cfg->s_damp, cfg->cut_off, cfg->n_depth, 0, 0,
                dec->code_len};
    return batch_create(dec, c, cfg->n_iter, n_obj, pts, n_pts, rays, n_rays, depth, n_fg, n_hyp, hyp_obj, out);
}

extern "C" void qsp_refine_batch_destroy(qsp_refine_batch* b) { batch_free(b); }

extern "C" int qsp_refine_batch_set_state(qsp_refine_batch* b, const float* t_cam_obj, const float* code) {
    if (!b || !t_cam_obj) return qsp_fail(QSP_ERR_INVALID, "set_state: bad argument");
    QSP_HIP(hipSetDevice(b->dec->device));
    std::vector<HypState> hs(b->n_hyp);
    for (int h = 0; h < b->n_hyp; ++h) {
        HypState& S = hs[h];
        memset(&S, 0, sizeof(S));
        inv4_gj(t_cam_obj + 16 * h, S.T_oc);   // t_obj_cam = inverse(t_cam_obj)  (optimizer.py:123)
        if (code) memcpy(S.code, code + (size_t)h * b->code_len, sizeof(float) * b->code_len);   // rest stays 0
        S.alive = 1;
        S.obj = b->hyp_obj[h];
    }
    QSP_HIP(hipMemcpy(b->st, hs.data(), sizeof(HypState) * b->n_hyp, hipMemcpyHostToDevice));
    if (b->pt_active) QSP_HIP(hipMemset(b->pt_active, 1, (size_t)b->n_hyp * b->act_stride));
    return QSP_OK;
}

static hipEvent_t next_event(qsp_refine_batch* b, size_t& cursor) {
    if (cursor >= b->ev.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        b->ev.push_back(e);
    }
    hipEvent_t e = b->ev[cursor++];
    (void)hipEventRecord(e, b->dec->stream);
    return e;
}

extern "C" int qsp_refine_batch_run(qsp_refine_batch* b, int32_t n_iter) {
    if (!b) return qsp_fail(QSP_ERR_INVALID, "run: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    if (n_iter <= 0) n_iter = b->n_iter_cfg;
    hipStream_t s = b->dec->stream;
    const int nH = b->n_hyp;
    const int nw_total = b->nw_sdf + (b->cfg.pose_only ? 0 : NW_REND);
    size_t cur = 0;
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    hipEvent_t e_begin = nullptr, e_end = nullptr;
    QSP_HIP(hipMemsetAsync(b->counters, 0, sizeof(unsigned long long) * 4, s));
    if (b->prof) e_begin = next_event(b, cur);
    for (int it = 0; it < n_iter; ++it) {
        RefineCfg cfg = b->cfg;
        cfg.iter = it;
        hipEvent_t a = nullptr;
        hipLaunchKernelGGL(k_c0, dim3(nH), dim3(MLP_THREADS), 0, s, b->st, b->dec->Pd, b->c0_all);
        if (!cfg.pose_only) {
            if (b->prof) a = next_event(b, cur);
            hipLaunchKernelGGL(k_sample, dim3(nH), dim3(256), 0, s, b->st, b->objs, b->rays, cfg, b->valid_rk, b->rk_stride,
                               b->ray_voff, b->ray_stride);
            if (b->prof) spans.push_back({a, next_event(b, cur), 2});
            if (b->prof) a = next_event(b, cur);
            hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, 0, b->st, b->objs, nH, b->nw_sdf, nw_total - b->nw_sdf,
                               b->work_fwd, b->qctl, TILE_P);     // (the forward pass keeps 64-point tiles: tens of thousands
            if (b->dec->fwd_bf3 == 2)                                     //  of ray samples fill the chip either way)
                hipLaunchKernelGGL(k_mlp_fwd_h2<2>, dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                   cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all);
            else if (b->dec->fwd_bf3)
                hipLaunchKernelGGL(k_mlp_fwd<true>, dim3(b->n_cu), dim3(MLP_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                   cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all);
            else
                hipLaunchKernelGGL(k_mlp_fwd<false>, dim3(b->n_cu), dim3(MLP_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                   cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all);
            if (b->prof) spans.push_back({a, next_event(b, cur), 1});
            if (b->prof) a = next_event(b, cur);
            hipLaunchKernelGGL(k_scan, dim3(nH), dim3(SCAN_RAYS), sizeof(float) * SCAN_RAYS * SCAN_LD, s, b->st, b->objs, b->depth, cfg, b->valid_rk, b->rk_stride,
                               b->ray_voff, b->ray_stride, b->sdf_valid, b->rend_rk, b->rend_deds, b->rend_res);
            if (b->prof) spans.push_back({a, next_event(b, cur), 2});
        }
        if (b->prof) a = next_event(b, cur);
        hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, 1, b->st, b->objs, nH, b->nw_sdf, nw_total - b->nw_sdf,
                           b->work_jtj, b->qctl, cfg.tile_p);
        if (b->dec->jac_bf3 == 2 && cfg.tile_p == 32)
            hipLaunchKernelGGL(k_mlp_jtj_h2<1>, dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->pts,
                               b->rays, cfg, b->dec->Pd, b->nw_sdf, b->rend_rk, b->rend_deds, b->rend_res, b->rk_stride,
                               b->pt_active, b->act_stride, b->res_buf, b->rows, b->rows_stride, b->partials, nw_total,
                               b->work_jtj, b->qctl, b->c0_all);
        else if (b->dec->jac_bf3 == 2)
            hipLaunchKernelGGL(k_mlp_jtj_h2<2>, dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->pts,
                               b->rays, cfg, b->dec->Pd, b->nw_sdf, b->rend_rk, b->rend_deds, b->rend_res, b->rk_stride,
                               b->pt_active, b->act_stride, b->res_buf, b->rows, b->rows_stride, b->partials, nw_total,
                               b->work_jtj, b->qctl, b->c0_all);
        else if (b->dec->jac_bf3)
            hipLaunchKernelGGL(k_mlp_jtj<true>, dim3(b->n_cu), dim3(MLP_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->pts,
                               b->rays, cfg, b->dec->Pd, b->nw_sdf, b->rend_rk, b->rend_deds, b->rend_res, b->rk_stride,
                               b->pt_active, b->act_stride, b->res_buf, b->rows, b->rows_stride, b->partials, nw_total,
                               b->work_jtj, b->qctl, b->c0_all);
        else
            hipLaunchKernelGGL(k_mlp_jtj<false>, dim3(b->n_cu), dim3(MLP_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->pts,
                               b->rays, cfg, b->dec->Pd, b->nw_sdf, b->rend_rk, b->rend_deds, b->rend_res, b->rk_stride,
                               b->pt_active, b->act_stride, b->res_buf, b->rows, b->rows_stride, b->partials, nw_total,
                               b->work_jtj, b->qctl, b->c0_all);
        if (b->prof) spans.push_back({a, next_event(b, cur), 0});
        if (b->prof) a = next_event(b, cur);
        hipLaunchKernelGGL(k_solve, dim3(nH), dim3(SOLVE_THREADS), 0, s, b->st, b->objs, cfg, b->partials, b->nw_sdf, nw_total,
                           b->pt_active, b->act_stride, b->trH, b->trb, b->trdx, b->counters);
        if (b->prof) spans.push_back({a, next_event(b, cur), 2});
        if (cfg.pose_only && it == 4)   // optimizer.py:80-82
            hipLaunchKernelGGL(k_inlier_filter, dim3((b->max_pts + 255) / 256, nH), dim3(256), 0, s, b->st, b->objs,
                               b->res_buf, b->act_stride, b->pt_active);
    }
    if (b->prof) e_end = next_event(b, cur);
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipStreamSynchronize(s));
    {
        const int rc = check_range(b->dec);
        if (rc) return rc;
    }
    if (b->prof) {
        qsp_refine_profile& p = b->profile;
        memset(&p, 0, sizeof(p));
        (void)hipEventElapsedTime(&p.ms_total, e_begin, e_end);
        for (const Span& sp : spans) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, sp.a, sp.b);
            if (sp.kind == 0) { p.ms_mlp_jtj += ms; p.n_launch_jtj++; }
            else if (sp.kind == 1) { p.ms_mlp_fwd += ms; p.n_launch_fwd++; }
            else p.ms_other += ms;
        }
        unsigned long long c[4];
        QSP_HIP(hipMemcpy(c, b->counters, sizeof(c), hipMemcpyDeviceToHost));
        p.pts_jtj = (int64_t)c[0];
        p.pts_fwd = (int64_t)c[1];
        p.tiles_jtj = (int64_t)c[2];
        p.tiles_fwd = (int64_t)c[3];
    }
    return QSP_OK;
}

extern "C" int qsp_refine_batch_profile(qsp_refine_batch* b, int enable, qsp_refine_profile* out) {
    if (!b) return qsp_fail(QSP_ERR_INVALID, "profile: null batch");
    b->prof = enable != 0;
    if (out) *out = b->profile;
    return QSP_OK;
}

extern "C" int qsp_refine_batch_get(qsp_refine_batch* b, float* t_cam_obj_out, float* code_out, float* loss_out,
                                    uint8_t* is_good_out) {
    if (!b) return qsp_fail(QSP_ERR_INVALID, "get: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    std::vector<HypState> hs(b->n_hyp);
    QSP_HIP(hipMemcpy(hs.data(), b->st, sizeof(HypState) * b->n_hyp, hipMemcpyDeviceToHost));
    for (int h = 0; h < b->n_hyp; ++h) {
        const HypState& S = hs[h];
        if (t_cam_obj_out) {
            inv4_gj(S.T_oc, t_cam_obj_out + 16 * h);   // t_cam_obj = inverse(t_obj_cam)  (optimizer.py:273)
        }
        if (code_out) memcpy(code_out + (size_t)h * b->code_len, S.code, sizeof(float) * b->code_len);
        if (loss_out) loss_out[h] = S.loss;
        if (is_good_out) is_good_out[h] = S.alive ? 1 : 0;
    }
    return QSP_OK;
}

extern "C" int qsp_refine_batch_trace(qsp_refine_batch* b, float* H, float* rhs, float* dx, int32_t* n_valid,
                                      int32_t* n_render, float* loss_terms) {
    if (!b) return qsp_fail(QSP_ERR_INVALID, "trace: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    if (H) QSP_HIP(hipMemcpy(H, b->trH, sizeof(float) * (size_t)b->n_hyp * NH * NH, hipMemcpyDeviceToHost));
    if (rhs) QSP_HIP(hipMemcpy(rhs, b->trb, sizeof(float) * (size_t)b->n_hyp * NH, hipMemcpyDeviceToHost));
    if (dx) QSP_HIP(hipMemcpy(dx, b->trdx, sizeof(float) * (size_t)b->n_hyp * NH, hipMemcpyDeviceToHost));
    if (n_valid || n_render || loss_terms) {
        std::vector<HypState> hs(b->n_hyp);
        QSP_HIP(hipMemcpy(hs.data(), b->st, sizeof(HypState) * b->n_hyp, hipMemcpyDeviceToHost));
        for (int h = 0; h < b->n_hyp; ++h) {
            if (n_valid) n_valid[h] = hs[h].n_valid;
            if (n_render) n_render[h] = hs[h].n_render;
            if (loss_terms) {
                loss_terms[2 * h] = hs[h].loss_sdf;
                loss_terms[2 * h + 1] = hs[h].loss_render;
            }
        }
    }
    return QSP_OK;
}

extern "C" int qsp_refine_batch_rows(qsp_refine_batch* b, int enable, int32_t hyp, float* rows_sdf, float* rows_render) {
    if (!b) return qsp_fail(QSP_ERR_INVALID, "rows: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    if (enable && !b->rows) {
        b->rows_stride = b->act_stride + b->rk_stride;
        QSP_HIP(hipMalloc((void**)&b->rows, sizeof(float) * (size_t)b->n_hyp * b->rows_stride * NJ));
    }
    if (!enable && b->rows) {
        (void)hipFree(b->rows);
        b->rows = nullptr;
    }
    if (enable && (rows_sdf || rows_render)) {
        if (hyp < 0 || hyp >= b->n_hyp) return qsp_fail(QSP_ERR_INVALID, "rows: hyp out of range");
        std::vector<HypState> hs(b->n_hyp);
        QSP_HIP(hipMemcpy(hs.data(), b->st, sizeof(HypState) * b->n_hyp, hipMemcpyDeviceToHost));
        const ObjView& ov = b->objs_h[b->hyp_obj[hyp]];
        const float* base = b->rows + (size_t)hyp * b->rows_stride * NJ;
        if (rows_sdf && ov.n_pts)
            QSP_HIP(hipMemcpy(rows_sdf, base, sizeof(float) * (size_t)ov.n_pts * NJ, hipMemcpyDeviceToHost));
        if (rows_render && hs[hyp].n_render > 0)
            QSP_HIP(hipMemcpy(rows_render, base + (size_t)ov.n_pts * NJ, sizeof(float) * (size_t)hs[hyp].n_render * NJ,
                              hipMemcpyDeviceToHost));
    }
    return QSP_OK;
}

extern "C" int qsp_reconstruct_objects(qsp_decoder* dec, const qsp_joint_cfg* cfg, int32_t n_obj, const float* const* pts,
                                       const int32_t* n_pts, const float* const* rays, const int32_t* n_rays,
                                       const float* const* depth, const int32_t* n_fg, int32_t n_hyp,
                                       const int32_t* hyp_obj, const float* t_cam_obj, const float* code,
                                       float* t_cam_obj_out, float* code_out, float* loss_out, uint8_t* is_good_out) {
    qsp_refine_batch* b = nullptr;
    int rc = qsp_refine_batch_create(dec, cfg, n_obj, pts, n_pts, rays, n_rays, depth, n_fg, n_hyp, hyp_obj, &b);
    if (rc) return rc;
    rc = qsp_refine_batch_set_state(b, t_cam_obj, code);
    if (!rc) rc = qsp_refine_batch_run(b, 0);
    if (!rc) rc = qsp_refine_batch_get(b, t_cam_obj_out, code_out, loss_out, is_good_out);
    qsp_refine_batch_destroy(b);
    return rc;
}

extern "C" int qsp_estimate_pose(qsp_decoder* dec, int32_t n, const float* t_co_se3, const float* scale,
                                 const float* const* pts, const int32_t* n_pts, const float* code, int32_t n_iter,
                                 float* t_co_out) {
    if (!dec || n <= 0 || !t_co_se3 || !scale || !pts || !n_pts || !code || !t_co_out)
        return qsp_fail(QSP_ERR_INVALID, "qsp_estimate_pose: bad argument");
    if (n_iter <= 0) n_iter = 5;
    RefineCfg c{};
    c.n_depth = 2;
    c.pose_only = 1;
    std::vector<int32_t> hyp(n);
    for (int i = 0; i < n; ++i) hyp[i] = i;
    qsp_refine_batch* b = nullptr;
    int rc = batch_create(dec, c, n_iter, n, pts, n_pts, nullptr, nullptr, nullptr, nullptr, n, hyp.data(), &b);
    if (rc) return rc;
    // bake the scale into the pose: t_cam_obj[:3,:3] *= scale   (optimizer.py:57-58)
    std::vector<float> T((size_t)n * 16);
    for (int i = 0; i < n; ++i) {
        memcpy(&T[16 * i], t_co_se3 + 16 * i, 16 * sizeof(float));
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) T[16 * i + 4 * r + q] *= scale[i];
    }
    rc = qsp_refine_batch_set_state(b, T.data(), code);
    if (!rc) rc = qsp_refine_batch_run(b, n_iter);
    std::vector<float> To((size_t)n * 16);
    if (!rc) rc = qsp_refine_batch_get(b, To.data(), nullptr, nullptr, nullptr);
    if (!rc)
        for (int i = 0; i < n; ++i) {
            memcpy(t_co_out + 16 * i, &To[16 * i], 16 * sizeof(float));
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) t_co_out[16 * i + 4 * r + q] /= scale[i];   // optimizer.py:88
        }
    qsp_refine_batch_destroy(b);
    return rc;
}

#if (QSP_EXP_VARIANT & 16)
// timing experiment only (tools/phase_times.py): the stamps of the last launch
extern "C" int qsp_debug_timestamps(unsigned long long* out /*96*/, int* n, unsigned long long* rt /*96*/) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(qsp::qsp_dbg_ts), sizeof(unsigned long long) * 96);
    if (rt) (void)hipMemcpyFromSymbol(rt, HIP_SYMBOL(qsp::qsp_dbg_rt), sizeof(unsigned long long) * 96);
    (void)hipMemcpyFromSymbol(n, HIP_SYMBOL(qsp::qsp_dbg_n), sizeof(int));
    return 0;
}
#endif

#include "mesh_extract.hpp"
#include "detections.hpp"
