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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/qsp_hip.h"
#include "common.hpp"
#include "sdf_mlp.hpp"

#include "sdf_kernels.hpp"

#ifndef QSP_JTJ_WAVES_DEFAULT
#define QSP_JTJ_WAVES_DEFAULT 8
#endif
#ifndef QSP_SCREEN_WAVES_DEFAULT
#define QSP_SCREEN_WAVES_DEFAULT 4
#endif


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
    bool narrow_capable = false;   // the decoder is small enough for the NARROW tile (narrow_tables); QSP_DEC_OPT_NARROW_TILE toggles its use
    float screen_margin = 0.f;     // QSP_DEC_OPT_RENDER_SCREENING: > 0 = two-pass ray-sample forward with this band margin
    int32_t depth_staging = 1;     // QSP_DEC_OPT_DEPTH_STAGING: the screened forward in two depth stages (k_stage_list); same bits
    int32_t screen_audit = 100;    // QSP_DEC_OPT_SCREEN_AUDIT: one in this many OUT-of-band samples is re-evaluated too (0 = off, 1 = all)
    int64_t screen_min_samples = -1;   // QSP_DEC_OPT_SCREENING_MIN_SAMPLES: -1 = more than two rounds of 64-point tiles over the chip
    int range_fallback = 1;        // QSP_DEC_OPT_RANGE_FALLBACK: a call that left fp16's range is repeated on the f32 pipe
    int64_t n_range_fallbacks = 0; // QSP_DEC_CNT_RANGE_FALLBACKS
    int64_t n_screen_fallbacks = 0;    // QSP_DEC_CNT_SCREEN_FALLBACKS: runs repeated in one pass by the screening self-check
    // qsp_reconstruct_objects keeps ONE resident batch per decoder, sized by the high-water mark of the calls so far: the
    // reference's call pattern is one object per call (src/LocalMapping_util.cc:705-760), and creating / destroying ~25 device
    // buffers per call cost ~0.5 ms of a 3 ms call
    // One decoder = one stream, one resident batch (arena) and a few fields that calls rewrite for their own duration (the f32
    // override of a range fallback, the screening margin of a self-check repeat): entry points that launch on the decoder or
    // touch those serialise on this lock -- two host threads sharing a decoder get correct results, one call at a
    // time; threads that want to overlap use a decoder each (the weights are 15 MB).
    std::recursive_mutex mu;
    struct qsp_refine_batch* arena = nullptr;
    int64_t n_arena_reuse = 0, n_arena_create = 0;
};

// Runs the enclosed call with every decoder pass on the exact-f32 pipe (the range fallback of the split-fp16 modes).
struct F32Override {
    qsp_decoder* d;
    int fwd, jac;
    explicit F32Override(qsp_decoder* d_) : d(d_), fwd(d_->fwd_bf3), jac(d_->jac_bf3) { d->fwd_bf3 = 0; d->jac_bf3 = 0; }
    ~F32Override() { d->fwd_bf3 = fwd; d->jac_bf3 = jac; }
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
                        std::vector<std::vector<float>>& Wc, std::vector<std::vector<float>>& Bc, std::vector<int>& src) {
    const int nl = desc->n_layers, m = nl - 1, L = desc->code_len;
    if (nl < 3 || nl > 9 || L < 1 || L > CODE_LEN)
        return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: 2..8 hidden layers and a code of 1..64 are supported");
    int s = desc->latent_in_layer;
    const bool has_skip = s >= 0;
    if (!has_skip) {   // no latent_in: any split with <= 4 layers on either side whose front part ends on a layer that fits slot 3
        s = -1;
        for (int c = std::max(1, m - 4); c <= std::min(4, m - 1) && s < 0; ++c)
            if (desc->out_dim[c - 1] >= 1 && desc->out_dim[c - 1] <= SKIP_COL) s = c;
        if (s < 0)
            return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: without a latent_in layer one of the hidden layers 1..4 must be "
                                                  "at most 445 wide (it takes the place of the layer in front of the skip)");
    }
    if (s < 1 || s > 4 || s > m - 1 || m - s > 4)
        return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: the latent_in layer needs 1..4 hidden layers in front of it and "
                                              "at most 4 from it on");
    if (desc->in_dim[0] != L + 3 || desc->out_dim[m] != 1)
        return qsp_fail(QSP_ERR_UNSUPPORTED, "decoder family: first layer takes [code | xyz], last layer has one output");
    for (int l = 0; l < m; ++l) {
        // canonical slot 3 is 445 wide whether or not the network has a skip connection: the layer in front of slot 4 (or the
        // identity layers that carry its output there) must fit it -- columns 445..511 of slot 4's input are [code | xyz]
        const int lim = (l == s - 1) ? SKIP_COL : HID;
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
    src.assign(9, -1);                                      // canonical slot -> source layer (-1: identity)
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

// Layer gains.  A ReLU network is positively homogeneous layer by layer: with a_L' = c_L a_L the network
//     W_L' = W_L c_L / c_{L-1},  b_L' = b_L c_L   (layer 4's [code | xyz] columns: W c_4, their input is not a hidden activation),
//     w_8' = w_8 / c_7
// computes the same function, and for c_L a power of two every intermediate of the f32 evaluation is the original one times a power
// of two -- same bits in the result, forward and backward.  The split-fp16 planes hold a value in [6.1e-5, 65504] to 22 bits and
// coarser below: a decoder whose layer gains are skewed (one layer's weights x 1e-4, the next x 1e4: the same function) would put a
// whole layer's activations into fp16's subnormal range.  So every layer whose median row norm -- seen from the rescaled layer in
// front of it -- is outside [2^-8, 2^8] is brought back to ~1 by such a c_L.  Decoders with ordinary gains (every golden and
// fitted one) get c_L = 1 throughout and are packed exactly as before.
static void equalize_gains(std::vector<std::vector<float>>& W, std::vector<std::vector<float>>& B) {
    const int in_dim[9] = {NIN, HID, HID, HID, HID, HID, HID, HID, HID};
    const int out_dim[9] = {HID, HID, HID, SKIP_COL, HID, HID, HID, HID, 1};
    float c_prev = 1.f;
    for (int l = 0; l < 8; ++l) {
        const int in = in_dim[l], out = out_dim[l];
        const int n_hidden = (l == 0) ? 0 : (l == 4 ? SKIP_COL : in);     // input columns that are the previous layer's activations
        std::vector<double> norms;                  // of the rows that exist (embedded narrower layers have zero rows)
        for (int o = 0; o < out; ++o) {
            double ss = 0;
            for (int k = 0; k < in; ++k) {
                const double v = (double)W[l][(size_t)o * in + k] / (k < n_hidden ? (double)c_prev : 1.0);
                ss += v * v;
            }
            if (ss > 0 && std::isfinite(ss)) norms.push_back(sqrt(ss));
        }
        // the MEDIAN row norm: a single outlier row must not push every other row of its layer out of fp16's normal range
        // (an outlier beyond 65 504 is what QSP_ERR_UNSUPPORTED of the split-fp16 mode is for)
        double m = 0;
        if (!norms.empty()) {
            std::nth_element(norms.begin(), norms.begin() + norms.size() / 2, norms.end());
            m = norms[norms.size() / 2];
        }
        float c = 1.f;
        if (m > 0 && (m < 0x1p-8 || m > 0x1p8)) c = (float)ldexp(1.0, -(int)lrint(log2(m)));
        if (c != 1.f || c_prev != 1.f) {
            for (int o = 0; o < out; ++o) {
                for (int k = 0; k < in; ++k) W[l][(size_t)o * in + k] *= (k < n_hidden) ? c / c_prev : c;
                B[l][o] *= c;
            }
        }
        c_prev = c;
    }
    if (c_prev != 1.f)
        for (int k = 0; k < HID; ++k) W[8][k] /= c_prev;
}

// What the NARROW form of the split-fp16 tile may skip (MlpParams::skip / ks_in / ks_out / ncb_in / ncb_out), read off the embedded
// matrices themselves: a slot is skipped when embed_family made it an identity; an input slab / output block "exists" up to the
// last column / row that carries a non-zero weight or bias.  `narrow` is set when the multiply-adds that remain are less than
// half of the full shape's: only then the narrow kernels (no hand-over of weight fragments between layers, idle waves) pay.
static void narrow_tables(qsp_decoder* d, const std::vector<std::vector<float>>& W, const std::vector<std::vector<float>>& B,
                          const std::vector<int>& src) {
    const int in_dim[9] = {NIN, HID, HID, HID, HID, HID, HID, HID, HID};
    const int out_dim[9] = {HID, HID, HID, SKIP_COL, HID, HID, HID, HID, 1};
    auto even4 = [](int v, int cap) { v = std::max(4, (v + 1) & ~1); return std::min(v, cap); };
    double macs = 0, full = 0;
    for (int l = 0; l < 8; ++l) {
        const int in = in_dim[l], out = out_dim[l];
        const int n_hidden = (l == 0) ? 0 : (l == 4 ? SKIP_COL : in);        // input columns that are hidden activations
        int w_out = 0, w_in = 0;
        for (int o = 0; o < out; ++o) {
            bool any = B[l][o] != 0.f;
            for (int k = 0; k < in; ++k)
                if (W[l][(size_t)o * in + k] != 0.f) {
                    any = true;
                    if (k < n_hidden) w_in = std::max(w_in, k + 1);
                }
            if (any) w_out = o + 1;
        }
        const bool skip = src[l] < 0;
        d->P.skip[l] = skip ? 1 : 0;
        const int ks_cap = (l == 4) ? K4 / 16 : HID / 16;
        int ks_in = even4((w_in + 15) / 16, ks_cap);
        if (l == 4 && ks_in >= K4 / 16 - 1) ks_in = K4 / 16;               // (the xyz slab is the last one: no gap left to skip)
        d->P.ks_in[l] = (uint8_t)ks_in;
        d->P.ks_out[l] = (uint8_t)even4((w_out + 15) / 16, HID / 16);
        d->P.ncb_in[l] = (uint8_t)std::max(1, (w_in + 31) / 32);
        d->P.ncb_out[l] = (uint8_t)std::max(1, (w_out + 31) / 32);
        full += (double)in * out;
        if (!skip) macs += (double)(l == 0 ? in : 16 * ks_in + (l == 4 ? NIN : 0)) * (32.0 * d->P.ncb_out[l]);
    }
    d->narrow_capable = macs < 0.5 * full;
    d->P.narrow = d->narrow_capable ? 1 : 0;
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
    std::vector<int> slot_src;
    {
        const int rc = embed_family(desc, Wsrc, W, Bias, slot_src);
        if (rc) return rc;
    }
    d->code_len = desc->code_len;
    equalize_gains(W, Bias);
    narrow_tables(d, W, Bias, slot_src);
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
            // layer 0 is evaluated directly (mlp_prepare + mlp_tile): code columns [k][unit], xyz columns per unit quad
            std::vector<float> wc((size_t)HID * CODE_LEN), wx((size_t)(HID / 4) * 3 * 4);
            for (int o = 0; o < HID; ++o)
                for (int k = 0; k < CODE_LEN; ++k) wc[(size_t)k * HID + o] = W[0][(size_t)o * in + k];     // [k][unit]: code_bias
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
                    for (int k = 0; k < CODE_LEN; ++k) wc[(size_t)k * HID + o] = W[4][(size_t)o * in + SKIP_COL + k];
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

// after a synchronisation of the decoder's stream: did a split-fp16 kernel meet a value it cannot represent?  (clears the flag)
static bool range_hit(qsp_decoder* d) {
    if (d->range_flag_h && *d->range_flag_h) {
        *d->range_flag_h = 0;
        return true;
    }
    return false;
}
static int range_error() {
    return qsp_fail(QSP_ERR_UNSUPPORTED, "split fp16: an activation or gradient of this decoder left fp16's range (65504); "
                                         "use the split-bf16 or f32 precision for it, or leave QSP_DEC_OPT_RANGE_FALLBACK on");
}
// What a call does when range_hit(): with the fallback on (default) and a split-fp16 pass in use, the caller repeats itself
// under an F32Override and counts it; otherwise the call fails.  The reference never fails a call for a numeric condition
// (reconstruct/optimizer.py:161-194 returns is_good = False at worst), so neither does the default configuration.
static bool range_should_fall_back(qsp_decoder* d) { return d->range_fallback && (d->fwd_bf3 == 2 || d->jac_bf3 == 2); }
static int check_range(qsp_decoder* d) { return range_hit(d) ? range_error() : QSP_OK; }

// Waves per workgroup of the split-fp16 Jacobian kernel: 4 (one 512-register wave per SIMD) or 8 (two 256-register waves per
// SIMD).  QSP_JTJ_WAVES overrides the default for same-box A/B measurements (tools/ab_bench.sh).
static int jtj_waves() {
    static int w = 0;
    if (!w) {
        const char* e = getenv("QSP_JTJ_WAVES");
        w = (e && atoi(e) == 8) ? 8 : ((e && atoi(e) == 4) ? 4 : QSP_JTJ_WAVES_DEFAULT);
    }
    return w;
}

// ... and of the screening pass (QSP_SCREEN_WAVES; same values either way)
static int screen_waves() {
    static int w = 0;
    if (!w) {
        const char* e = getenv("QSP_SCREEN_WAVES");
        w = (e && atoi(e) == 8) ? 8 : ((e && atoi(e) == 4) ? 4 : QSP_SCREEN_WAVES_DEFAULT);
    }
    return w;
}
static int jtj_waves_t32() {
    static int w = 0;
    if (!w) {
        const char* e = getenv("QSP_JTJ_WAVES_T32");
        w = (e && atoi(e) == 8) ? 8 : 4;
    }
    return w;
}

static int mlp_attr_once() {
    static bool done = false;
    if (done) return QSP_OK;
    const int bytes = (int)sizeof(MlpSmem);
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd_h2<2, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd_h2<2, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_h2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_h2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd_h1<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpSmemH1)));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_fwd_h1<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpSmemH1)));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_screen<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpSmemH1)));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_screen<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpSmemH1)));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_h2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_decode_h2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    QSP_HIP(hipFuncSetAttribute((const void*)k_mlp_jtj_h2<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
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
    std::unique_lock<std::recursive_mutex> lk_d;
    if (d) lk_d = std::unique_lock<std::recursive_mutex>(d->mu);
    if (!d) return qsp_fail(QSP_ERR_INVALID, "qsp_decoder_set_option: null decoder");
    switch (option) {
        case QSP_DEC_OPT_FORWARD_PRECISION:
            if (value < 0 || value > 2) return qsp_fail(QSP_ERR_INVALID, "forward precision: 0 (f32 MFMA), 1 (split bf16) or 2 (split fp16)");
            if (value == 2 && !d->fp16_ok) return qsp_fail(QSP_ERR_UNSUPPORTED, "split fp16: a weight of this decoder is outside fp16's range");
            d->fwd_bf3 = value;
            if (value != 2) d->screen_margin = 0.f;      // (the screened forward pass exists on the split-fp16 pipe only)
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
        case QSP_DEC_OPT_RENDER_SCREENING:
            if (value < 0 || value > 50000) return qsp_fail(QSP_ERR_UNSUPPORTED, "render screening: margin in 1e-6 units, 0 (off) .. 50000");
            if (value > 0 && d->fwd_bf3 != 2)
                return qsp_fail(QSP_ERR_UNSUPPORTED, "render screening exists on the split-fp16 forward pass only: set "
                                                     "QSP_DEC_OPT_FORWARD_PRECISION to 2 first");
            d->screen_margin = 1e-6f * (float)value;
            return QSP_OK;
        case QSP_DEC_OPT_DEPTH_STAGING:
            if (value < 0 || value > 2) return qsp_fail(QSP_ERR_INVALID, "depth staging: 0 (off), 1 (large batches) or 2 (always)");
            d->depth_staging = (int32_t)value;
            return QSP_OK;
        case QSP_DEC_OPT_SCREEN_AUDIT:
            if (value < 0 || value > 1000000) return qsp_fail(QSP_ERR_INVALID, "screening audit: one in N out-of-band samples, N = 0 (off) .. 1e6");
            d->screen_audit = (int32_t)value;
            return QSP_OK;
        case QSP_DEC_OPT_SCREENING_MIN_SAMPLES:
            if (value < -1) return qsp_fail(QSP_ERR_INVALID, "screening threshold: -1 (automatic) or a sample count >= 0");
            d->screen_min_samples = value;
            return QSP_OK;
        case QSP_DEC_OPT_NARROW_TILE: {
            if (value != 0 && value != 1) return qsp_fail(QSP_ERR_INVALID, "narrow tile: 0 or 1");
            if (value && !d->narrow_capable) return qsp_fail(QSP_ERR_UNSUPPORTED, "narrow tile: this decoder fills most of the 8 x 512 shape");
            d->P.narrow = value;
            QSP_HIP(hipSetDevice(d->device));
            QSP_HIP(hipStreamSynchronize(d->stream));
            QSP_HIP(hipMemcpy(d->Pd, &d->P, sizeof(MlpParams), hipMemcpyHostToDevice));
            return QSP_OK;
        }
        case QSP_DEC_OPT_USE_TANH: {
            if (value != 0 && value != 1) return qsp_fail(QSP_ERR_INVALID, "use_tanh: 0 or 1");
            d->P.use_tanh = value;
            QSP_HIP(hipSetDevice(d->device));
            QSP_HIP(hipStreamSynchronize(d->stream));
            QSP_HIP(hipMemcpy(d->Pd, &d->P, sizeof(MlpParams), hipMemcpyHostToDevice));
            return QSP_OK;
        }
        case QSP_DEC_OPT_RANGE_FALLBACK:
            if (value != 0 && value != 1) return qsp_fail(QSP_ERR_INVALID, "range fallback: 0 or 1");
            d->range_fallback = value;
            return QSP_OK;
        default: return qsp_fail(QSP_ERR_INVALID, "qsp_decoder_set_option: unknown option");
    }
}

extern "C" int64_t qsp_decoder_get_counter(qsp_decoder* d, int32_t counter) {
    if (!d) return -1;
    switch (counter) {
        case QSP_DEC_CNT_RANGE_FALLBACKS: return d->n_range_fallbacks;
        case QSP_DEC_CNT_ARENA_REUSED: return d->n_arena_reuse;
        case QSP_DEC_CNT_ARENA_CREATED: return d->n_arena_create;
        case QSP_DEC_CNT_NARROW_TILE: return d->P.narrow ? 1 : 0;
        case QSP_DEC_CNT_SCREEN_FALLBACKS: return d->n_screen_fallbacks;
        default: return -1;
    }
}

static void batch_free(struct qsp_refine_batch* b);
extern "C" void qsp_decoder_destroy(qsp_decoder* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->arena) batch_free(d->arena);
    d->arena = nullptr;
    for (void* p : d->allocs) (void)hipFree(p);
    if (d->range_flag_h) (void)hipHostFree(d->range_flag_h);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

static int decode_once(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* y, float* grad, bool* hit) {
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
        if (d->P.narrow) hipLaunchKernelGGL((k_decode_h2<true, true>), dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy, dg);
        else hipLaunchKernelGGL((k_decode_h2<true, false>), dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy, dg);
    else if (grad && d->jac_bf3)
        hipLaunchKernelGGL((k_decode<true, true>), dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy, dg);
    else if (grad)
        hipLaunchKernelGGL(k_decode<true>, dim3(grid), dim3(MLP_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy, dg);
    else if (d->fwd_bf3 == 2)
        if (d->P.narrow) hipLaunchKernelGGL((k_decode_h2<false, true>), dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy,
                                            (float*)nullptr);
        else hipLaunchKernelGGL((k_decode_h2<false, false>), dim3(grid), dim3(H2_THREADS), sizeof(MlpSmem), d->stream, dc, dx, n, d->Pd, dy,
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
    if (range_hit(d)) {
        (void)hipFree(dc);
        (void)hipFree(dx);
        (void)hipFree(dy);
        if (dg) (void)hipFree(dg);
        *hit = true;
        return QSP_OK;
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

static int decode_common(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* y, float* grad) {
    if (!d || !code || !xyz || n < 0 || !y) return qsp_fail(QSP_ERR_INVALID, "decode: bad argument");
    if (n == 0) return QSP_OK;
    bool hit = false;
    int rc = decode_once(d, code, xyz, n, y, grad, &hit);
    if (rc || !hit) return rc;
    if (!range_should_fall_back(d)) return range_error();
    F32Override f32(d);
    d->n_range_fallbacks++;
    hit = false;
    return decode_once(d, code, xyz, n, y, grad, &hit);
}

extern "C" int qsp_decode_sdf(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* sdf_out) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (d) lk_d = std::unique_lock<std::recursive_mutex>(d->mu);
    return decode_common(d, code, xyz, n, sdf_out, nullptr);
}

// The screening tile's values on explicit points (diagnostic: tests and tools/screen_margin.py measure |s1 - s3| with it).
extern "C" int qsp_decode_sdf_screen(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* s1_out) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (d) lk_d = std::unique_lock<std::recursive_mutex>(d->mu);
    if (!d || !code || !xyz || n < 0 || !s1_out) return qsp_fail(QSP_ERR_INVALID, "decode_screen: bad argument");
    if (n == 0) return QSP_OK;
    if (!d->fp16_ok) return qsp_fail(QSP_ERR_UNSUPPORTED, "split fp16: a weight of this decoder is outside fp16's range");
    QSP_HIP(hipSetDevice(d->device));
    float *dc = nullptr, *dx = nullptr, *dy = nullptr;
    QSP_HIP(hipMalloc((void**)&dc, CODE_LEN * sizeof(float)));
    QSP_HIP(hipMalloc((void**)&dx, n * 3 * sizeof(float)));
    QSP_HIP(hipMalloc((void**)&dy, n * sizeof(float)));
    float code64[CODE_LEN] = {};
    memcpy(code64, code, sizeof(float) * d->code_len);
    hipError_t e = hipMemcpyAsync(dc, code64, CODE_LEN * sizeof(float), hipMemcpyHostToDevice, d->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dx, xyz, n * 3 * sizeof(float), hipMemcpyHostToDevice, d->stream);
    if (e == hipSuccess) {
        const int grid = (int)std::min<int64_t>((n + H1_ROWS - 1) / H1_ROWS, 4096);
        if (screen_waves() == 8)
            hipLaunchKernelGGL(k_decode_screen<8>, dim3(grid), dim3(512), sizeof(MlpSmemH1), d->stream, dc, dx, n, d->Pd, dy);
        else
            hipLaunchKernelGGL(k_decode_screen<4>, dim3(grid), dim3(H2_THREADS), sizeof(MlpSmemH1), d->stream, dc, dx, n, d->Pd, dy);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(s1_out, dy, n * sizeof(float), hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(dc);
    (void)hipFree(dx);
    (void)hipFree(dy);
    if (e != hipSuccess) return qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
    return check_range(d);
}

extern "C" int qsp_sdf_value_grad(qsp_decoder* d, const float* code, const float* xyz, int64_t n, float* y, float* grad) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (d) lk_d = std::unique_lock<std::recursive_mutex>(d->mu);
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
    // capacities the device buffers were allocated for (qsp_refine_batch_reload refills a batch whose capacities suffice)
    int cap_obj = 0, cap_hyp = 0;
    int64_t cap_pts_total = 0, cap_rays_total = 0;
    int tile_p_created = 64;
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
    float *trH = nullptr, *trb = nullptr, *trdx = nullptr, *trrot = nullptr;
    uint8_t* pt_active = nullptr;   // pose-only mode
    float* res_buf = nullptr;       // pose-only mode: per-point residual of the current iteration
    unsigned long long* counters = nullptr;   // [4] points/tiles processed (fwd+bwd, fwd-only)
    int2 *work_fwd = nullptr, *work_jtj = nullptr;   // work-queue items (k_plan)
    int* qctl = nullptr;            // [4] item counts / next-item counters
    float last_screen_dmax = 0.f;   // the last screened pass: largest |s1 - s3| seen, audited out-of-band samples, how many of them wrong
    int32_t last_audit_wrong = 0;
    int64_t last_audited = 0;
    int32_t* band_idx = nullptr;    // screened forward pass: per hypothesis, indices into its valid-sample list (k_mlp_fwd_h1)
    int32_t* stage_idx = nullptr;   // depth-staged forward: the current stage's samples, as positions in the valid list (k_stage_list)
    uint8_t* ray_open = nullptr;    //   per (hypothesis, ray): no opaque sample in the stages so far
    HypState* st_snap = nullptr;    // the hypotheses as a run found them (restored when the run is repeated on the f32 pipe)
    uint8_t* act_snap = nullptr;    // pose-only mode: pt_active likewise
    float* c0_all = nullptr;        // [n_hyp][2][512] code part of layers 0 and 4 (bias included) per hypothesis (k_c0)
    int n_cu = 256;
    float* rows = nullptr;          // optional tap of the Jacobian rows (qsp_refine_batch_rows)
    int64_t rows_stride = 0;
    // What the host provides per call -- hypothesis states, object table, depth, rays, surface points -- lives in ONE device arena
    // [st | objs | depth | rays | pts] (parts packed to what the fill holds, 256-byte aligned) with a pinned host mirror: fill and
    // set_state write the mirror, the next consumer (run, get, ...) uploads what changed as ONE asynchronous copy on the
    // decoder's stream.  (They were five blocking copies from pageable memory: 0.25 ms of idle GPU in front of every one-object
    // call of 2.7 ms, profiles/r04_call_timeline.txt.)
    char* in_dev = nullptr;
    char* in_host = nullptr;        // pinned
    HypState* out_host = nullptr;   // pinned: the states on their way back (get, trace)
    size_t in_bytes = 0;
    size_t st_bytes = 0;            // this fill's states
    size_t obs_lo = 0, obs_hi = 0;  // the part of the mirror behind the states that waits for its upload
    bool st_dirty = false;
    // profiling
    bool prof = false;
    qsp_refine_profile profile{};
    std::vector<hipEvent_t> ev;
};

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// uploads what fill / set_state left in the mirror (one copy; nothing when nothing changed)
static int batch_upload(qsp_refine_batch* b) {
    size_t lo = 0, hi = 0;
    if (b->st_dirty && b->obs_hi > b->obs_lo) lo = 0, hi = b->obs_hi;
    else if (b->st_dirty) lo = 0, hi = b->st_bytes;
    else if (b->obs_hi > b->obs_lo) lo = b->obs_lo, hi = b->obs_hi;
    b->st_dirty = false;
    b->obs_lo = b->obs_hi = 0;
    if (hi > lo) QSP_HIP(hipMemcpyAsync(b->in_dev + lo, b->in_host + lo, hi - lo, hipMemcpyHostToDevice, b->dec->stream));
    return QSP_OK;
}

static void batch_free(qsp_refine_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->in_host) (void)hipHostFree(b->in_host);
    if (b->out_host) (void)hipHostFree(b->out_host);
    void* ptrs[] = {b->in_dev, b->valid_rk, b->ray_voff, b->rend_rk, b->sdf_valid,
                    b->rend_deds, b->rend_res, b->partials, b->trH, b->trb, b->trdx, b->trrot, b->pt_active, b->res_buf, b->rows, b->counters,
                    b->work_fwd, b->work_jtj, b->c0_all, b->band_idx, b->stage_idx, b->ray_open, b->st_snap, b->act_snap};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
    delete b;
    (void)hipGetLastError();   // errors are ignored here; do not leave one behind for the next call's launch check
}

// Capacities of a batch: what its device buffers are allocated for.  A batch created for given inputs has exactly their extents;
// the resident batch of qsp_reconstruct_objects (the decoder's arena) is created with head-room and refilled by batch_fill.
struct BatchCaps {
    int obj = 0, hyp = 0, max_pts = 0, max_rays = 0;
    int64_t pts_total = 0, rays_total = 0;
};

static int batch_validate(const RefineCfg& cfg, int32_t n_obj, const int32_t* n_pts, const int32_t* n_rays, const int32_t* n_fg,
                          int32_t n_hyp, const int32_t* hyp_obj, BatchCaps* need) {
    BatchCaps c;
    c.obj = n_obj;
    c.hyp = n_hyp;
    for (int o = 0; o < n_obj; ++o) {
        const int nr = cfg.pose_only ? 0 : n_rays[o];
        const int nf = cfg.pose_only ? 0 : n_fg[o];
        if (n_pts[o] < 0 || nr < 0 || nf < 0 || nf > nr || nr >= (1 << 25)) return qsp_fail(QSP_ERR_INVALID, "refine batch: bad per-object counts");
        c.pts_total += n_pts[o];
        c.rays_total += nr;
        c.max_pts = std::max(c.max_pts, (int)n_pts[o]);
        c.max_rays = std::max(c.max_rays, nr);
    }
    for (int h = 0; h < n_hyp; ++h)
        if (hyp_obj[h] < 0 || hyp_obj[h] >= n_obj) return qsp_fail(QSP_ERR_INVALID, "refine batch: hyp_obj out of range");
    *need = c;
    return QSP_OK;
}

// (re)fills a batch whose capacities suffice: object extents, observations, hypothesis -> object map.  Strides, slot counts
// and buffer sizes stay those of the capacities; every per-object quantity the kernels use comes from the ObjView table.
static int batch_fill(qsp_refine_batch* b, int32_t n_obj, const float* const* pts, const int32_t* n_pts, const float* const* rays,
                      const int32_t* n_rays, const float* const* depth, const int32_t* n_fg, int32_t n_hyp, const int32_t* hyp_obj,
                      bool device_fill) {
    const RefineCfg& cfg = b->cfg;
    b->n_obj = n_obj;
    b->n_hyp = n_hyp;
    b->objs_h.resize(n_obj);
    int64_t po = 0, ro = 0;
    for (int o = 0; o < n_obj; ++o) {
        const int nr = cfg.pose_only ? 0 : n_rays[o];
        const int nf = cfg.pose_only ? 0 : n_fg[o];
        b->objs_h[o] = ObjView{po, ro, n_pts[o], nr, nf, 0};
        po += n_pts[o];
        ro += nr;
    }
    b->hyp_obj.assign(hyp_obj, hyp_obj + n_hyp);
    // this fill's layout of the input arena (sizes <= the capacities it was allocated for, part by part)
    const size_t n_ray_f = (size_t)std::max<int64_t>(ro, 1), n_pt_f = (size_t)std::max<int64_t>(po, 1);
    b->st_bytes = sizeof(HypState) * (size_t)n_hyp;
    const size_t o_objs = align256(b->st_bytes), o_depth = o_objs + align256(sizeof(ObjView) * (size_t)n_obj);
    const size_t o_rays = o_depth + align256(sizeof(float) * n_ray_f), o_pts = o_rays + align256(sizeof(float) * 3 * n_ray_f);
    const size_t o_end = o_pts + sizeof(float) * 3 * n_pt_f;
    if (o_end > b->in_bytes) return qsp_fail(QSP_ERR_INVALID, "refine batch: inputs exceed the batch's capacities");
    b->objs = (ObjView*)(b->in_dev + o_objs);
    b->depth = (float*)(b->in_dev + o_depth);
    b->rays = (float*)(b->in_dev + o_rays);
    b->pts = (float*)(b->in_dev + o_pts);
    memcpy(b->in_host + o_objs, b->objs_h.data(), sizeof(ObjView) * (size_t)n_obj);
    b->obs_lo = o_objs;
    b->obs_hi = o_depth;
    if (device_fill) {        // (the observation arrays are written by a kernel; depth entries beyond a ray's foreground stay 0)
        QSP_HIP(hipMemsetAsync(b->depth, 0, sizeof(float) * n_ray_f, b->dec->stream));
        return QSP_OK;
    }
    float* hp = (float*)(b->in_host + o_pts);
    float* hr = (float*)(b->in_host + o_rays);
    float* hd = (float*)(b->in_host + o_depth);
    memset(hd, 0, sizeof(float) * n_ray_f);
    for (int o = 0; o < n_obj; ++o) {
        const ObjView& v = b->objs_h[o];
        if (v.n_pts) memcpy(hp + 3 * v.pts_off, pts[o], sizeof(float) * 3 * v.n_pts);
        if (v.n_rays) memcpy(hr + 3 * v.ray_off, rays[o], sizeof(float) * 3 * v.n_rays);
        if (v.n_fg) memcpy(hd + v.ray_off, depth[o], sizeof(float) * v.n_fg);
    }
    b->obs_hi = o_end;
    return QSP_OK;
}

static bool batch_fits(const qsp_refine_batch* b, const BatchCaps& need) {
    return need.obj <= b->cap_obj && need.hyp <= b->cap_hyp && need.max_pts <= b->max_pts && need.max_rays <= b->max_rays &&
           need.pts_total <= b->cap_pts_total && need.rays_total <= b->cap_rays_total;
}

static int batch_create(qsp_decoder* dec, const RefineCfg& cfg, int n_iter, int32_t n_obj, const float* const* pts,
                        const int32_t* n_pts, const float* const* rays, const int32_t* n_rays,
                        const float* const* depth, const int32_t* n_fg, int32_t n_hyp, const int32_t* hyp_obj,
                        qsp_refine_batch** out, bool device_fill = false, const BatchCaps* caps_in = nullptr) {
    // device_fill: only the extents are given, the observation arrays are written by a kernel (detections.hpp)
    if (!dec || !out || n_obj <= 0 || n_hyp <= 0 || (!pts && !device_fill) || !n_pts || !hyp_obj)
        return qsp_fail(QSP_ERR_INVALID, "refine batch: bad argument");
    if (!cfg.pose_only && ((!device_fill && (!rays || !depth)) || !n_rays || !n_fg))
        return qsp_fail(QSP_ERR_INVALID, "refine batch: rays missing");
    if (cfg.n_depth < 2 || cfg.n_depth > MAX_DEPTH) return qsp_fail(QSP_ERR_INVALID, "n_depth must be in [2, 64]");
    if (dec->tile_p == 32 && (dec->fwd_bf3 != 2 || dec->jac_bf3 != 2))
        return qsp_fail(QSP_ERR_UNSUPPORTED, "32-point tiles (QSP_DEC_OPT_TILE_POINTS) exist on the split-fp16 pipe only: set both "
                                             "precisions to 2 first");
    BatchCaps need;
    {
        const int rc = batch_validate(cfg, n_obj, n_pts, n_rays, n_fg, n_hyp, hyp_obj, &need);
        if (rc) return rc;
    }
    BatchCaps caps = caps_in ? *caps_in : need;      // (caps_in >= need, the caller's business)
    const int tile_p = dec->tile_p;
    QSP_HIP(hipSetDevice(dec->device));
    qsp_refine_batch* b = new qsp_refine_batch();
    b->dec = dec;
    b->device = dec->device;
    b->code_len = dec->code_len;
    b->cfg = cfg;
    b->cfg.code_len = dec->code_len;
    b->cfg.tile_p = dec->tile_p;
    b->tile_p_created = dec->tile_p;
    b->n_iter_cfg = n_iter;
    b->cap_obj = caps.obj;
    b->cap_hyp = caps.hyp;
    b->cap_pts_total = caps.pts_total;
    b->cap_rays_total = caps.rays_total;
    b->max_pts = caps.max_pts;
    b->max_rays = caps.max_rays;
    const int64_t cp = std::max<int64_t>(caps.pts_total, 1), cr = std::max<int64_t>(caps.rays_total, 1);
    const int cap_hyp = caps.hyp;
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
    b->in_bytes = align256(sizeof(HypState) * (size_t)cap_hyp) + align256(sizeof(ObjView) * (size_t)caps.obj) +
                  align256(sizeof(float) * (size_t)cr) + align256(sizeof(float) * 3 * (size_t)cr) + align256(sizeof(float) * 3 * (size_t)cp);
    QSP_ALLOC(b->in_dev, b->in_bytes);
    b->st = (HypState*)b->in_dev;
    if (!rc) {
        hipError_t e_ = hipHostMalloc((void**)&b->in_host, b->in_bytes, hipHostMallocDefault);
        if (e_ == hipSuccess) e_ = hipHostMalloc((void**)&b->out_host, sizeof(HypState) * (size_t)cap_hyp, hipHostMallocDefault);
        if (e_ != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e_));
    }
    QSP_ALLOC(b->partials, sizeof(float) * (size_t)cap_hyp * nw_total * PART_FLOATS);
    QSP_ALLOC(b->trH, sizeof(float) * (size_t)cap_hyp * NH * NH);
    QSP_ALLOC(b->trb, sizeof(float) * (size_t)cap_hyp * NH);
    QSP_ALLOC(b->trdx, sizeof(float) * (size_t)cap_hyp * NH);
    QSP_ALLOC(b->trrot, sizeof(float) * (size_t)cap_hyp * 4);
    QSP_ALLOC(b->counters, sizeof(unsigned long long) * 8 + sizeof(int) * 8);      // [counters | qctl]: one fill clears both
    if (!rc) b->qctl = (int*)(b->counters + 8);
    QSP_ALLOC(b->st_snap, sizeof(HypState) * cap_hyp);
    // qctl: [0..3] the two queues, [4], [5] the done counters of the plans in k_sample's / k_scan's tail
    QSP_ALLOC(b->c0_all, sizeof(float) * (size_t)cap_hyp * 2 * HID);
    QSP_ALLOC(b->work_jtj, sizeof(int2) * (size_t)cap_hyp * nw_total);
    if (!cfg.pose_only) QSP_ALLOC(b->work_fwd, sizeof(int2) * (size_t)cap_hyp * ((b->rk_stride + TILE_P - 1) / TILE_P));
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dec->device) == hipSuccess && ncu > 0) b->n_cu = ncu;
    }
    if (!cfg.pose_only) {
        QSP_ALLOC(b->valid_rk, sizeof(int32_t) * (size_t)cap_hyp * b->rk_stride);
        QSP_ALLOC(b->ray_voff, sizeof(int32_t) * (size_t)cap_hyp * b->ray_stride);
        QSP_ALLOC(b->rend_rk, sizeof(int32_t) * (size_t)cap_hyp * b->rk_stride);
        QSP_ALLOC(b->sdf_valid, sizeof(float) * (size_t)cap_hyp * b->rk_stride);
        QSP_ALLOC(b->rend_deds, sizeof(float) * (size_t)cap_hyp * b->rk_stride);
        QSP_ALLOC(b->rend_res, sizeof(float) * (size_t)cap_hyp * b->rk_stride);
        QSP_ALLOC(b->band_idx, sizeof(int32_t) * (size_t)cap_hyp * b->rk_stride);
        QSP_ALLOC(b->stage_idx, sizeof(int32_t) * (size_t)cap_hyp * b->rk_stride);
        QSP_ALLOC(b->ray_open, (size_t)cap_hyp * b->ray_stride);
    } else {
        QSP_ALLOC(b->pt_active, (size_t)cap_hyp * b->act_stride);
        QSP_ALLOC(b->act_snap, (size_t)cap_hyp * b->act_stride);
        QSP_ALLOC(b->res_buf, sizeof(float) * (size_t)cap_hyp * b->act_stride);
    }
#undef QSP_ALLOC
    if (!rc) {
        hipError_t e = hipMemset(b->trH, 0, sizeof(float) * (size_t)cap_hyp * NH * NH);
        if (e == hipSuccess) e = hipMemset(b->trb, 0, sizeof(float) * (size_t)cap_hyp * NH);
        if (e == hipSuccess) e = hipMemset(b->trdx, 0, sizeof(float) * (size_t)cap_hyp * NH);
        if (e == hipSuccess) e = hipMemset(b->trrot, 0, sizeof(float) * (size_t)cap_hyp * 4);
        if (e != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
    }
    if (!rc) rc = batch_fill(b, n_obj, pts, n_pts, rays, n_rays, depth, n_fg, n_hyp, hyp_obj, device_fill);
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
    std::unique_lock<std::recursive_mutex> lk_d;
    if (b && b->dec) lk_d = std::unique_lock<std::recursive_mutex>(b->dec->mu);
    if (!b || !t_cam_obj) return qsp_fail(QSP_ERR_INVALID, "set_state: bad argument");
    QSP_HIP(hipSetDevice(b->dec->device));
    HypState* hs = (HypState*)b->in_host;      // (the mirror; uploaded by the next run / get)
    for (int h = 0; h < b->n_hyp; ++h) {
        HypState& S = hs[h];
        memset(&S, 0, sizeof(S));
        inv4_gj(t_cam_obj + 16 * h, S.T_oc);   // t_obj_cam = inverse(t_cam_obj)  (optimizer.py:123)
        if (code) memcpy(S.code, code + (size_t)h * b->code_len, sizeof(float) * b->code_len);   // rest stays 0
        S.alive = 1;
        S.obj = b->hyp_obj[h];
    }
    b->st_dirty = true;
    if (b->pt_active) QSP_HIP(hipMemsetAsync(b->pt_active, 1, (size_t)b->n_hyp * b->act_stride, b->dec->stream));
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

// the screened forward is trusted while the largest |s1 - s3| it sees on a band sample stays below this share of the margin
constexpr float SCREEN_TRUST = 0.5f;

// one pass over n_iter Gauss-Newton iterations on the decoder's current pipes; *hit = a split-fp16 kernel left fp16's range,
// *screen_hit = the screened forward saw |s1 - s3| above half its margin on a band sample (the margin's premise is in doubt)
static int run_once(qsp_refine_batch* b, int32_t n_iter, bool* hit, bool* screen_hit) {
    bool screened_any = false;
    hipStream_t s = b->dec->stream;
    const int nH = b->n_hyp;
    const int nw_total = b->nw_sdf + (b->cfg.pose_only ? 0 : NW_REND);
    size_t cur = 0;
    struct Span { hipEvent_t a, b; int kind; };
    std::vector<Span> spans;
    hipEvent_t e_begin = nullptr, e_end = nullptr;
    // (the work counters, and with them the queue words: the done counters of the plan tails may be left over from a run cut short)
    QSP_HIP(hipMemsetAsync(b->counters, 0, sizeof(unsigned long long) * 8 + sizeof(int) * 8, s));
    if (b->prof) e_begin = next_event(b, cur);
    for (int it = 0; it < n_iter; ++it) {
        RefineCfg cfg = b->cfg;
        cfg.iter = it;
        if (b->dec->jac_bf3 != 2) cfg.tile_p = TILE_P;      // (32-point tiles exist on the split-fp16 pipe only: the f32 repeat of a
                                                            //  batch created for them runs 64-point tiles over the same slots)
        hipEvent_t a = nullptr;
        if (cfg.pose_only) hipLaunchKernelGGL(k_c0, dim3(nH), dim3(MLP_THREADS), 0, s, b->st, b->dec->Pd, b->c0_all);
        if (!cfg.pose_only) {      // (k_sample also forms the bias vectors k_c0 forms in pose-only mode)
            if (b->prof) a = next_event(b, cur);
            // Two passes pay when the one-pass kernel would need more than one round of 64-point tiles over the chip; a batch that
            // fits one round (a single object per call) is faster in one pass: one tile deep either way, without the second
            // launch.  Both give the same bits, so the choice is free.  (~half of the ray samples are inside the unit ball.)
            int64_t ub_samples = 0;
            for (int h = 0; h < nH; ++h) ub_samples += (int64_t)b->objs_h[b->hyp_obj[h]].n_rays * cfg.n_depth;
            const int64_t min_samples = b->dec->screen_min_samples >= 0 ? b->dec->screen_min_samples : 2 * (int64_t)b->n_cu * TILE_P;
            // (a narrow decoder's one-pass forward on the NARROW tile is cheaper than the full-width screening pass: not screened)
            const bool screen = b->dec->fwd_bf3 == 2 && b->dec->screen_margin > 0.f && ub_samples > min_samples && !b->dec->P.narrow;
            // the forward kernel's item list is built in k_sample's tail (plan_tail; the forward pass keeps 64-point tiles: tens of
            // thousands of ray samples fill the chip either way) -- or, when the screened pass runs in depth stages, in the tail of
            // each stage's k_stage_list
            // (a stage costs a list kernel, a plan and a band pass that is one tile deep whatever its size: it pays once the screening
            //  pass is tens of tile rounds over the chip -- C4 and C5; measured break-even around C2, 0.7 M samples: 18.1 against
            //  19.1 ms per step; four yaw flips of one object: 3.8 against 4.8 ms per call)
            const bool staged = screen && cfg.n_depth >= 4 &&
                                (b->dec->depth_staging == 2 || (b->dec->depth_staging == 1 && ub_samples > 64 * (int64_t)b->n_cu * H1_ROWS));
            const PlanTail pt_fwd{staged ? nullptr : b->work_fwd, b->qctl, b->qctl + 4, nH, b->nw_sdf, nw_total - b->nw_sdf,
                                  screen ? H1_ROWS : TILE_P, 0};
            hipLaunchKernelGGL(k_sample, dim3(nH), dim3(SAMPLE_THREADS), 0, s, b->st, b->objs, b->rays, cfg, b->valid_rk, b->rk_stride,
                               b->ray_voff, b->ray_stride, b->dec->Pd, b->c0_all, pt_fwd);
            if (b->prof) spans.push_back({a, next_event(b, cur), 2});
            if (b->prof) a = next_event(b, cur);
            if (screen) {
                // two passes (QSP_DEC_OPT_RENDER_SCREENING): every sample on the one-product tile, then the band around the
                // surface on the split-fp16 tile; the queue's control words and item list are reused behind the first pass.
                // Depth-staged (QSP_DEC_OPT_DEPTH_STAGING): that pair once for the depth indices [0, D/2) of every ray and once
                // for [D/2, D) of the rays that have no opaque sample yet (k_stage_list).
                const int n_stage = staged ? 2 : 1, k_mid = cfg.n_depth / 2;
                for (int sg = 0; sg < n_stage; ++sg) {
                    const int32_t* stage_list = nullptr;
                    if (staged) {
                        const PlanTail pt_st{b->work_fwd, b->qctl, b->qctl + 4, nH, b->nw_sdf, nw_total - b->nw_sdf, H1_ROWS, 3};
                        hipLaunchKernelGGL(k_stage_list, dim3(nH), dim3(256), 0, s, b->st, b->objs, cfg, b->valid_rk, b->rk_stride, b->ray_voff,
                                           b->ray_stride, b->sdf_valid, b->ray_open, b->stage_idx, 0, sg == 0 ? 0 : k_mid,
                                           sg == 0 ? k_mid : cfg.n_depth, pt_st);
                        stage_list = b->stage_idx;
                    }
                    if (screen_waves() == 8)
                        hipLaunchKernelGGL(k_mlp_fwd_h1<8>, dim3(b->n_cu), dim3(512), sizeof(MlpSmemH1), s, b->st, b->objs, b->rays,
                                           cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all,
                                           b->band_idx, cfg.cut_off + b->dec->screen_margin, b->dec->screen_audit, stage_list);
                    else
                        hipLaunchKernelGGL(k_mlp_fwd_h1<4>, dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmemH1), s, b->st, b->objs, b->rays,
                                           cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all,
                                           b->band_idx, cfg.cut_off + b->dec->screen_margin, b->dec->screen_audit, stage_list);
                    hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, 2, b->st, b->objs, nH, b->nw_sdf, nw_total - b->nw_sdf,
                                       b->work_fwd, b->qctl, TILE_P);
                    hipLaunchKernelGGL((k_mlp_fwd_h2<2, false, 4>), dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                       cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all,
                                       (const int32_t*)b->band_idx, (unsigned int*)(b->counters + 5));
                }
                screened_any = true;
            } else if (b->dec->fwd_bf3 == 2 && b->dec->P.narrow)
                hipLaunchKernelGGL((k_mlp_fwd_h2<2, true, 8>), dim3(b->n_cu), dim3(512), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                   cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all,
                                   (const int32_t*)nullptr, (unsigned int*)nullptr);
            else if (b->dec->fwd_bf3 == 2)
                hipLaunchKernelGGL((k_mlp_fwd_h2<2, false, 4>), dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                   cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all,
                                   (const int32_t*)nullptr, (unsigned int*)nullptr);
            else if (b->dec->fwd_bf3)
                hipLaunchKernelGGL(k_mlp_fwd<true>, dim3(b->n_cu), dim3(MLP_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                   cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all);
            else
                hipLaunchKernelGGL(k_mlp_fwd<false>, dim3(b->n_cu), dim3(MLP_THREADS), sizeof(MlpSmem), s, b->st, b->objs, b->rays,
                                   cfg, b->dec->Pd, b->valid_rk, b->rk_stride, b->sdf_valid, b->work_fwd, b->qctl, b->c0_all);
            if (b->prof) spans.push_back({a, next_event(b, cur), 1});
            if (b->prof) a = next_event(b, cur);
            // (the Jacobian kernel's item list is built in k_scan's tail)
            const PlanTail pt_jtj{b->work_jtj, b->qctl, b->qctl + 5, nH, b->nw_sdf, nw_total - b->nw_sdf, cfg.tile_p, 1};
            hipLaunchKernelGGL(k_scan, dim3(nH), dim3(SCAN_RAYS), sizeof(float) * SCAN_RAYS * SCAN_LD, s, b->st, b->objs, b->depth, cfg, b->valid_rk, b->rk_stride,
                               b->ray_voff, b->ray_stride, b->sdf_valid, b->rend_rk, b->rend_deds, b->rend_res, pt_jtj);
            if (b->prof) spans.push_back({a, next_event(b, cur), 2});
        }
        if (b->prof) a = next_event(b, cur);
        if (cfg.pose_only)      // (no render pass in front of it: the plan is a launch of its own)
            hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, s, 1, b->st, b->objs, nH, b->nw_sdf, nw_total - b->nw_sdf,
                               b->work_jtj, b->qctl, cfg.tile_p);
        if (b->dec->jac_bf3 == 2) {
            const JtjArgs ja{b->st, b->objs, b->pts, b->rays, cfg, b->dec->Pd, b->nw_sdf, nw_total, b->rend_rk, b->rend_deds, b->rend_res,
                             b->rk_stride, b->pt_active, b->act_stride, b->res_buf, b->rows, b->rows_stride, b->partials, b->work_jtj,
                             b->qctl, b->c0_all};
            // eight waves of 256 registers (two per SIMD) or four of 512 (one per SIMD): same arithmetic, same bits; jtj_waves()
            // (32-point tiles are the latency option -- one tile deep: there the four-wave form is the shorter chain, 180 us
            //  against 248 per tile; QSP_JTJ_WAVES_T32=8 selects the other for measurements)
            if (b->dec->P.narrow && cfg.tile_p == 32)      // narrow decoders: eight waves, so that the column blocks that exist
                hipLaunchKernelGGL((k_mlp_jtj_h2<1, 8, true>), dim3(b->n_cu), dim3(512), sizeof(MlpSmem), s, ja);   // spread over all SIMDs
            else if (b->dec->P.narrow)
                hipLaunchKernelGGL((k_mlp_jtj_h2<2, 8, true>), dim3(b->n_cu), dim3(512), sizeof(MlpSmem), s, ja);
            else if (cfg.tile_p == 32 && jtj_waves_t32() == 8)
                hipLaunchKernelGGL((k_mlp_jtj_h2<1, 8>), dim3(b->n_cu), dim3(512), sizeof(MlpSmem), s, ja);
            else if (cfg.tile_p == 32)
                hipLaunchKernelGGL((k_mlp_jtj_h2<1, 4>), dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmem), s, ja);
            else if (jtj_waves() == 8)
                hipLaunchKernelGGL((k_mlp_jtj_h2<2, 8>), dim3(b->n_cu), dim3(512), sizeof(MlpSmem), s, ja);
            else
                hipLaunchKernelGGL((k_mlp_jtj_h2<2, 4>), dim3(b->n_cu), dim3(H2_THREADS), sizeof(MlpSmem), s, ja);
        }
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
                           b->pt_active, b->act_stride, b->trH, b->trb, b->trdx, b->counters, b->trrot);
        if (b->prof) spans.push_back({a, next_event(b, cur), 2});
        if (cfg.pose_only && it == 4)   // optimizer.py:80-82
            hipLaunchKernelGGL(k_inlier_filter, dim3((b->max_pts + 255) / 256, nH), dim3(256), 0, s, b->st, b->objs,
                               b->res_buf, b->act_stride, b->pt_active);
    }
    if (b->prof) e_end = next_event(b, cur);
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipStreamSynchronize(s));
    if (range_hit(b->dec)) {
        *hit = true;
        return QSP_OK;
    }
    float dmax = 0.f;
    if (screened_any) {      // (16 bytes, behind the synchronisation above; a one-object call is never screened and skips it)
        unsigned long long w[2] = {0, 0};
        QSP_HIP(hipMemcpy(w, b->counters + 5, sizeof(w), hipMemcpyDeviceToHost));
        const unsigned int bits = (unsigned int)w[0];
        memcpy(&dmax, &bits, sizeof(dmax));
        b->last_screen_dmax = dmax;
        b->last_audit_wrong = (int32_t)(w[0] >> 32);      // audited out-of-band samples the screening pass clamped wrongly
        b->last_audited = (int64_t)w[1];
        if (!(dmax <= SCREEN_TRUST * b->dec->screen_margin) || b->last_audit_wrong > 0) {
            *screen_hit = true;
            return QSP_OK;
        }
    }
    if (b->prof) {
        qsp_refine_profile& p = b->profile;
        memset(&p, 0, sizeof(p));
        p.screen_max_diff = dmax;
        p.pts_audit = screened_any ? b->last_audited : 0;
        (void)hipEventElapsedTime(&p.ms_total, e_begin, e_end);
        for (const Span& sp : spans) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, sp.a, sp.b);
            if (sp.kind == 0) { p.ms_mlp_jtj += ms; p.n_launch_jtj++; }
            else if (sp.kind == 1) { p.ms_mlp_fwd += ms; p.n_launch_fwd++; }
            else p.ms_other += ms;
        }
        unsigned long long c[8];
        QSP_HIP(hipMemcpy(c, b->counters, sizeof(c), hipMemcpyDeviceToHost));
        p.pts_jtj = (int64_t)c[0];
        p.pts_fwd = (int64_t)c[1];
        p.tiles_jtj = (int64_t)c[2];
        p.tiles_fwd = (int64_t)c[3];
        p.pts_band = (int64_t)c[4];
    }
    return QSP_OK;
}

extern "C" int qsp_refine_batch_run(qsp_refine_batch* b, int32_t n_iter) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (b && b->dec) lk_d = std::unique_lock<std::recursive_mutex>(b->dec->mu);
    if (!b) return qsp_fail(QSP_ERR_INVALID, "run: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    if (n_iter <= 0) n_iter = b->n_iter_cfg;
    qsp_decoder* d = b->dec;
    {
        const int rc_up = batch_upload(b);
        if (rc_up) return rc_up;
    }
    const bool may_fall_back = range_should_fall_back(d);
    if (may_fall_back) {      // the state this run starts from, in case it has to be repeated on the f32 pipe (n_hyp x 432 bytes)
        QSP_HIP(hipMemcpyAsync(b->st_snap, b->st, sizeof(HypState) * b->n_hyp, hipMemcpyDeviceToDevice, d->stream));
        if (b->pt_active) QSP_HIP(hipMemcpyAsync(b->act_snap, b->pt_active, (size_t)b->n_hyp * b->act_stride, hipMemcpyDeviceToDevice, d->stream));
    }
    // (the screened forward is a split-fp16 feature; its self-check needs the same snapshot)
    const bool may_screen = d->fwd_bf3 == 2 && d->screen_margin > 0.f;
    if (may_screen && !may_fall_back) {
        QSP_HIP(hipMemcpyAsync(b->st_snap, b->st, sizeof(HypState) * b->n_hyp, hipMemcpyDeviceToDevice, d->stream));
        if (b->pt_active) QSP_HIP(hipMemcpyAsync(b->act_snap, b->pt_active, (size_t)b->n_hyp * b->act_stride, hipMemcpyDeviceToDevice, d->stream));
    }
    auto restore = [&]() -> int {
        QSP_HIP(hipMemcpyAsync(b->st, b->st_snap, sizeof(HypState) * b->n_hyp, hipMemcpyDeviceToDevice, d->stream));
        if (b->pt_active) QSP_HIP(hipMemcpyAsync(b->pt_active, b->act_snap, (size_t)b->n_hyp * b->act_stride, hipMemcpyDeviceToDevice, d->stream));
        return QSP_OK;
    };
    // (what the last run reported must not outlive it -- with profiling off nothing else resets these)
    b->profile.range_fallbacks = b->profile.screen_fallbacks = b->profile.screen_audit_failures = 0;
    b->profile.screen_max_diff = 0.f;
    b->profile.pts_audit = 0;
    b->last_screen_dmax = 0.f;
    b->last_audit_wrong = 0;
    b->last_audited = 0;
    bool hit = false, screen_hit = false;
    int rc = run_once(b, n_iter, &hit, &screen_hit);
    if (rc || (!hit && !screen_hit)) return rc;
    int screen_fallbacks = 0, audit_failures = 0;
    float failed_dmax = 0.f;
    int64_t failed_audited = 0;
    if (screen_hit) {
        audit_failures = b->last_audit_wrong;
        failed_dmax = b->last_screen_dmax;
        failed_audited = b->last_audited;
        // The band samples' two values differ by more than half the margin: the premise "no sample outside the band could have
        // crossed the cut-off" no longer has its safety factor.  Repeat from the starting state in one pass (the path the
        // screened one is bit-identical to when the premise holds).
        if ((rc = restore())) return rc;
        const float margin = d->screen_margin;
        d->screen_margin = 0.f;
        d->n_screen_fallbacks++;
        screen_fallbacks = 1;
        hit = screen_hit = false;
        rc = run_once(b, n_iter, &hit, &screen_hit);
        d->screen_margin = margin;
        b->profile.screen_fallbacks = 1;
        b->profile.screen_audit_failures = audit_failures;      // what the screened attempt saw (the repeat is one-pass: it sees nothing)
        b->profile.screen_max_diff = failed_dmax;
        b->profile.pts_audit = failed_audited;
        if (rc || !hit) return rc;
    }
    if (!may_fall_back) return range_error();
    // A value left fp16's range.  The reference accepts such a decoder (it computes in float32) and never raises for a numeric
    // condition (reconstruct/optimizer.py:161-194), so the run is repeated from its starting state on the exact-f32 pipe.
    if ((rc = restore())) return rc;
    F32Override f32(d);
    d->n_range_fallbacks++;
    hit = screen_hit = false;
    rc = run_once(b, n_iter, &hit, &screen_hit);
    b->profile.range_fallbacks = 1;
    b->profile.screen_fallbacks = screen_fallbacks;
    b->profile.screen_audit_failures = audit_failures;
    return rc;
}


extern "C" int qsp_refine_batch_profile(qsp_refine_batch* b, int enable, qsp_refine_profile* out) {
    if (!b) return qsp_fail(QSP_ERR_INVALID, "profile: null batch");
    b->prof = enable != 0;
    if (out) *out = b->profile;
    return QSP_OK;
}

// the hypothesis states as the device holds them (pending uploads first), through the pinned buffer; valid until the next call
static int batch_states(qsp_refine_batch* b, const HypState** out) {
    const int rc = batch_upload(b);
    if (rc) return rc;
    QSP_HIP(hipMemcpyAsync(b->out_host, b->st, sizeof(HypState) * b->n_hyp, hipMemcpyDeviceToHost, b->dec->stream));
    QSP_HIP(hipStreamSynchronize(b->dec->stream));
    *out = b->out_host;
    return QSP_OK;
}

extern "C" int qsp_refine_batch_get(qsp_refine_batch* b, float* t_cam_obj_out, float* code_out, float* loss_out,
                                    uint8_t* is_good_out) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (b && b->dec) lk_d = std::unique_lock<std::recursive_mutex>(b->dec->mu);
    if (!b) return qsp_fail(QSP_ERR_INVALID, "get: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    const HypState* hs = nullptr;
    {
        const int rc = batch_states(b, &hs);
        if (rc) return rc;
    }
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
    std::unique_lock<std::recursive_mutex> lk_d;
    if (b && b->dec) lk_d = std::unique_lock<std::recursive_mutex>(b->dec->mu);
    if (!b) return qsp_fail(QSP_ERR_INVALID, "trace: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    if (H) QSP_HIP(hipMemcpy(H, b->trH, sizeof(float) * (size_t)b->n_hyp * NH * NH, hipMemcpyDeviceToHost));
    if (rhs) QSP_HIP(hipMemcpy(rhs, b->trb, sizeof(float) * (size_t)b->n_hyp * NH, hipMemcpyDeviceToHost));
    if (dx) QSP_HIP(hipMemcpy(dx, b->trdx, sizeof(float) * (size_t)b->n_hyp * NH, hipMemcpyDeviceToHost));
    if (n_valid || n_render || loss_terms) {
        const HypState* hs = nullptr;
        {
            const int rc = batch_states(b, &hs);
            if (rc) return rc;
        }
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

extern "C" int qsp_refine_batch_trace_rot(qsp_refine_batch* b, float* rot4) {
    if (!b || !rot4) return qsp_fail(QSP_ERR_INVALID, "trace_rot: bad argument");
    QSP_HIP(hipSetDevice(b->dec->device));
    QSP_HIP(hipMemcpy(rot4, b->trrot, sizeof(float) * (size_t)b->n_hyp * 4, hipMemcpyDeviceToHost));
    return QSP_OK;
}

extern "C" int qsp_refine_batch_rows(qsp_refine_batch* b, int enable, int32_t hyp, float* rows_sdf, float* rows_render) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (b && b->dec) lk_d = std::unique_lock<std::recursive_mutex>(b->dec->mu);
    if (!b) return qsp_fail(QSP_ERR_INVALID, "rows: null batch");
    QSP_HIP(hipSetDevice(b->dec->device));
    if (enable && !b->rows) {
        b->rows_stride = b->act_stride + b->rk_stride;
        QSP_HIP(hipMalloc((void**)&b->rows, sizeof(float) * (size_t)b->cap_hyp * b->rows_stride * NJ));
    }
    if (!enable && b->rows) {
        (void)hipFree(b->rows);
        b->rows = nullptr;
    }
    if (enable && (rows_sdf || rows_render)) {
        if (hyp < 0 || hyp >= b->n_hyp) return qsp_fail(QSP_ERR_INVALID, "rows: hyp out of range");
        const HypState* hs = nullptr;
        {
            const int rc = batch_states(b, &hs);
            if (rc) return rc;
        }
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
    std::unique_lock<std::recursive_mutex> lk_d;
    if (dec) lk_d = std::unique_lock<std::recursive_mutex>(dec->mu);
    if (!dec || !cfg) return qsp_fail(QSP_ERR_INVALID, "qsp_reconstruct_objects: null argument");
    if (cfg->code_len != dec->code_len) return qsp_fail(QSP_ERR_INVALID, "code_len of the optimizer config differs from the decoder's");
    if (n_obj <= 0 || n_hyp <= 0 || !pts || !n_pts || !rays || !n_rays || !depth || !n_fg || !hyp_obj)
        return qsp_fail(QSP_ERR_INVALID, "refine batch: bad argument");
    const RefineCfg c{cfg->k1, cfg->k2, cfg->k3, cfg->k4, cfg->b1, cfg->b2, cfg->lr, cfg->s_damp, cfg->cut_off, cfg->n_depth, 0, 0,
                      dec->code_len};
    BatchCaps need;
    int rc = batch_validate(c, n_obj, n_pts, n_rays, n_fg, n_hyp, hyp_obj, &need);
    if (rc) return rc;
    // The decoder's resident batch: reused when the call fits its capacities and the quantities its strides were derived from
    // (depth samples, tile size) are the same; the weights of the cost terms are launch arguments and simply replaced.  Otherwise
    // it is rebuilt with a quarter of head-room over the larger of (this call, what it held): a high-water mark, so a sequence
    // of objects of varying size settles after a few calls.  Results do not depend on the capacities (tests/test_gpu_latency.py).
    qsp_refine_batch* b = dec->arena;
    if (b && (b->cfg.n_depth != c.n_depth || b->tile_p_created != dec->tile_p || b->cfg.pose_only || !batch_fits(b, need))) {
        BatchCaps grow = need;
        grow.obj = std::max(need.obj, b->cap_obj);
        grow.hyp = std::max(need.hyp, b->cap_hyp);
        grow.max_pts = std::max(need.max_pts, b->max_pts);
        grow.max_rays = std::max(need.max_rays, b->max_rays);
        grow.pts_total = std::max(need.pts_total, b->cap_pts_total);
        grow.rays_total = std::max(need.rays_total, b->cap_rays_total);
        need = grow;
        batch_free(b);
        b = dec->arena = nullptr;
    }
    if (!b) {
        BatchCaps caps = need;
        caps.hyp = std::max(caps.hyp, 4);           // (one object x its four yaw flips: the other call shape of the reference)
        caps.max_pts += caps.max_pts / 4;
        caps.max_rays += caps.max_rays / 4;
        caps.pts_total += caps.pts_total / 4;
        caps.rays_total += caps.rays_total / 4;
        caps.pts_total = std::max<int64_t>(caps.pts_total, (int64_t)caps.max_pts);
        caps.rays_total = std::max<int64_t>(caps.rays_total, (int64_t)caps.max_rays);
        rc = batch_create(dec, c, cfg->n_iter, n_obj, pts, n_pts, rays, n_rays, depth, n_fg, n_hyp, hyp_obj, &b, false, &caps);
        if (rc) return rc;
        dec->arena = b;
        dec->n_arena_create++;
    } else {
        QSP_HIP(hipSetDevice(dec->device));
        const int tile_p = b->cfg.tile_p;
        b->cfg = c;
        b->cfg.code_len = dec->code_len;
        b->cfg.tile_p = tile_p;
        b->n_iter_cfg = cfg->n_iter;
        rc = batch_fill(b, n_obj, pts, n_pts, rays, n_rays, depth, n_fg, n_hyp, hyp_obj, false);
        if (rc) return rc;
        dec->n_arena_reuse++;
    }
    rc = qsp_refine_batch_set_state(b, t_cam_obj, code);
    if (!rc) rc = qsp_refine_batch_run(b, 0);
    if (!rc) rc = qsp_refine_batch_get(b, t_cam_obj_out, code_out, loss_out, is_good_out);
    return rc;
}

extern "C" int qsp_estimate_pose(qsp_decoder* dec, int32_t n, const float* t_co_se3, const float* scale,
                                 const float* const* pts, const int32_t* n_pts, const float* code, int32_t n_iter,
                                 float* t_co_out) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (dec) lk_d = std::unique_lock<std::recursive_mutex>(dec->mu);
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

#if QSP_PHASE_CLOCK
// timing experiment only (tools/phase_clock.py): ticks of the 100 MHz counter between the marks of k_sample / k_scan / k_solve, summed
// over the launches since the last call (which clears them)
extern "C" int qsp_debug_phase_ticks(unsigned long long* out /*48*/) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(qsp::qsp_phase_ticks), sizeof(unsigned long long) * 48);
    static const unsigned long long zero[48] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(qsp::qsp_phase_ticks), zero, sizeof(zero));
    return 0;
}
#endif

#include "mesh_extract.hpp"
#include "detections.hpp"
