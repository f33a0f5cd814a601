// detections.hpp -- caller-side marshalling of path A on the device.
//
// Replaces the per-detection host loops of LocalMapping::ProcessDetectedObjects, src/LocalMapping_util.cc:585-760, that sit
// either side of Optimizer.reconstruct_object: world -> camera transform of the object's map points, pixel -> ray
// back-projection, the yaw-flip initial poses and the keep-the-best rule.  Included at the end of sdf_refine.hip (it fills a
// qsp_refine_batch in place and reads its HypState array).
//
// The reference does this arithmetic with OpenCV (cv::Mat float products) and Eigen (Matrix3f inverse, Matrix4f products,
// AngleAxisf) -- system dependencies that are neither vendored in the reference tree nor installed in the build image, so
// the formulas below restate their published small-matrix code paths in plain f32 without contraction:
//   cv::Mat  Rcw * x + tcw   -> MatExpr folds it into gemm(Rcw, x, 1, tcw, 1); the 3x3 f32 special case sums a0*b0 + a1*b1 +
//                               a2*b2 in float and stores (float)(t*alpha + c*beta) evaluated in double
//   Eigen    Matrix3f::inverse() -> cofactors / determinant along column 0 (compute_inverse_size3)
//            fixed-size products -> coefficient-wise, sum over k in increasing order
//            AngleAxisf::toRotationMatrix() -> [c 0 s; 0 (1-c)+c 0; -s 0 c] for the axis e_y
// A build of the reference with FMA contraction (-march=native) can differ from this in the last bit; tests/test_gpu_detections.py
// states the tolerance against such a build and checks the restatement (oracle/detections_oracle.py) bit for bit.
#pragma once

namespace qsp {
namespace det {

struct Inputs {             // device pointers
    const float *T_cw, *K, *T_wo, *code;
    const int32_t *n_flip, *hyp_off;
    const int32_t *pts_off, *fg_off, *bg_off;
    const float *pts_world, *fg_px, *fg_world, *bg_rays;
    const float* Ry;        // (max_flip, 9) rotation about e_y by k * flip_angle, built on the host (libm cosf / sinf)
    int code_len;           // of the decoder (<= 64): row stride of `code`
};

// one row of cv::Mat (3x3 f32) * (3x1 f32) + (3x1 f32)
__device__ inline float cv_row(const float* T, int i, float x, float y, float z) {
#pragma clang fp contract(off)
    const float t0 = T[4 * i] * x + T[4 * i + 1] * y + T[4 * i + 2] * z;
    return (float)((double)t0 * 1.0 + (double)T[4 * i + 3] * 1.0);
}

// Eigen::Matrix3f::inverse() of K = [fx 0 cx; 0 fy cy; 0 0 1], row-major out
__device__ inline void eigen_inv_k(const float* k4, float* inv) {
#pragma clang fp contract(off)
    const float m[3][3] = {{k4[0], 0.f, k4[2]}, {0.f, k4[1], k4[3]}, {0.f, 0.f, 1.f}};
    auto cof = [&](int i, int j) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
    };
    const float c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const float det = (c00 * m[0][0] + c10 * m[1][0]) + c20 * m[2][0];
    const float invdet = 1.0f / det;
    inv[0] = c00 * invdet;
    inv[1] = c10 * invdet;
    inv[2] = c20 * invdet;
    inv[3] = cof(0, 1) * invdet;
    inv[4] = cof(1, 1) * invdet;
    inv[5] = cof(2, 1) * invdet;
    inv[6] = cof(0, 2) * invdet;
    inv[7] = cof(1, 2) * invdet;
    inv[8] = cof(2, 2) * invdet;
}

// blockIdx.y = detection; the three observation arrays of the resident batch are written in the reference's order
__global__ void __launch_bounds__(256) k_det_assemble(Inputs in, const ObjView* objs, float* pts, float* rays, float* depth) {
#pragma clang fp contract(off)
    const int o = blockIdx.y;
    const ObjView v = objs[o];
    const float* T = in.T_cw + 16 * o;
    const int stride = gridDim.x * blockDim.x;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // surface_points_cam (:610-628)
    {
        const float* w = in.pts_world + 3 * (int64_t)in.pts_off[o];
        float* d = pts + 3 * v.pts_off;
        for (int i = t; i < v.n_pts; i += stride) {
            const float x = w[3 * i], y = w[3 * i + 1], z = w[3 * i + 2];
            d[3 * i] = cv_row(T, 0, x, y, z);
            d[3 * i + 1] = cv_row(T, 1, x, y, z);
            d[3 * i + 2] = cv_row(T, 2, x, y, z);
        }
    }
    // depth_obs and fg_rays (:634-669)
    {
        float inv[9];
        eigen_inv_k(in.K + 4 * o, inv);
        const float* w = in.fg_world + 3 * (int64_t)in.fg_off[o];
        const float* px = in.fg_px + 2 * (int64_t)in.fg_off[o];
        float* r = rays + 3 * v.ray_off;
        float* dd = depth + v.ray_off;
        for (int i = t; i < v.n_fg; i += stride) {
            dd[i] = cv_row(T, 2, w[3 * i], w[3 * i + 1], w[3 * i + 2]);
            const float u = px[2 * i], vv = px[2 * i + 1];
            for (int c = 0; c < 3; ++c) r[3 * i + c] = (inv[3 * c] * u + inv[3 * c + 1] * vv) + inv[3 * c + 2] * 1.0f;
        }
    }
    // background rays appended (:671-672)
    {
        const float* g = in.bg_rays + 3 * (int64_t)in.bg_off[o];
        float* r = rays + 3 * (v.ray_off + v.n_fg);
        const int n = 3 * (v.n_rays - v.n_fg);
        for (int i = t; i < n; i += stride) r[i] = g[i];
    }
}

// blockIdx.x = detection, thread k = flip hypothesis: initial state of hypothesis hyp_off[d] + k  (:706,722-733)
__global__ void __launch_bounds__(64) k_det_init(Inputs in, HypState* st, float* t_init_tap) {
#pragma clang fp contract(off)
    const int d = blockIdx.x, k = threadIdx.x;
    const int nf = in.n_flip ? in.n_flip[d] : 1;
    if (k >= nf) return;
    const int h = in.hyp_off[d] + k;
    const float* A = in.T_cw + 16 * d;
    float F[16];
    for (int i = 0; i < 16; ++i) F[i] = in.T_wo[16 * d + i];
    if (k > 0) {   // flipped_Two.topLeftCorner(3,3) = flipped_Two.topLeftCorner(3,3) * Ry
        const float* Ry = in.Ry + 9 * k;
        float R[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = (F[4 * i] * Ry[j] + F[4 * i + 1] * Ry[3 + j]) + F[4 * i + 2] * Ry[6 + j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) F[4 * i + j] = R[3 * i + j];
    }
    float T0[16];   // SE3Tcw * flipped_Two
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            T0[4 * i + j] = ((A[4 * i] * F[j] + A[4 * i + 1] * F[4 + j]) + A[4 * i + 2] * F[8 + j]) + A[4 * i + 3] * F[12 + j];
    if (t_init_tap)
        for (int i = 0; i < 16; ++i) t_init_tap[16 * h + i] = T0[i];
    HypState& S = st[h];
    inv4_gj(T0, S.T_oc);   // optimizer.py:123
    for (int i = 0; i < CODE_LEN; ++i) S.code[i] = (in.code && i < in.code_len) ? in.code[(int64_t)in.code_len * d + i] : 0.f;
    for (int i = 0; i < 16; ++i) S.T_co[i] = 0.f;
    S.scale = S.d_min = S.d_max = S.loss = S.loss_sdf = S.loss_render = 0.f;
    S.alive = 1;
    S.n_valid = S.n_render = 0;
    S.obj = d;
    S.n_band = 0;
    S.n_stage = S.n_band_total = S.n_eval = 0;
}

// one thread per detection: the keep rule of :738-752 over its hypotheses, in order
__global__ void __launch_bounds__(64) k_det_select(const HypState* st, const int32_t* hyp_off, int n_det, float* t_cam_obj,
                                                   float* code, float* loss, uint8_t* good, int32_t* kept, float* losses,
                                                   int code_len) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_det) return;
    const int h0 = hyp_off[d], h1 = hyp_off[d + 1];
    int best = h0;
    losses[h0] = st[h0].loss;
    for (int h = h0 + 1; h < h1; ++h) {
        const bool state = st[best].alive != 0, state_f = st[h].alive != 0;
        const float l = st[best].loss, lf = st[h].loss;
        losses[h] = lf;
        if (!state || (l > lf && state_f)) best = h;
    }
    inv4_gj(st[best].T_oc, t_cam_obj + 16 * d);   // optimizer.py:273
    for (int i = 0; i < code_len; ++i) code[(int64_t)code_len * d + i] = st[best].code[i];
    loss[d] = st[best].loss;
    good[d] = st[best].alive ? 1 : 0;
    kept[d] = best - h0;
}

struct DevPool {            // the call's temporary device arrays
    std::vector<void*> p;
    ~DevPool() {
        for (void* q : p) (void)hipFree(q);
    }
    template <class T>
    int up(const T* host, size_t n, const T** out) {
        void* d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return qsp_fail(QSP_ERR_DEVICE, "detections: out of device memory");
        p.push_back(d);
        if (n && hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return qsp_fail(QSP_ERR_DEVICE, "detections: upload failed");
        *out = (const T*)d;
        return QSP_OK;
    }
    template <class T>
    int make(size_t n, T** out) {
        void* d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return qsp_fail(QSP_ERR_DEVICE, "detections: out of device memory");
        p.push_back(d);
        *out = (T*)d;
        return QSP_OK;
    }
};

}  // namespace det
}  // namespace qsp

extern "C" int qsp_refine_detections(qsp_decoder* dec, const qsp_joint_cfg* cfg, const qsp_detections* in,
                                     qsp_detection_results* out) {
    std::unique_lock<std::recursive_mutex> lk_d;
    if (dec) lk_d = std::unique_lock<std::recursive_mutex>(dec->mu);
    using namespace qsp::det;
    if (!dec || !cfg || !in || !out) return qsp_fail(QSP_ERR_INVALID, "qsp_refine_detections: null argument");
    const int n = in->n_det;
    if (n <= 0 || !in->T_cw || !in->K || !in->T_wo || !in->pts_off || !in->fg_off || !in->bg_off)
        return qsp_fail(QSP_ERR_INVALID, "qsp_refine_detections: bad argument");
    if (cfg->code_len != dec->code_len) return qsp_fail(QSP_ERR_INVALID, "code_len of the optimizer config differs from the decoder's");
    std::vector<int32_t> n_pts(n), n_rays(n), n_fg(n), hyp_off(n + 1, 0), hyp_obj;
    int max_flip = 1, max_items = 1;
    for (int d = 0; d < n; ++d) {
        const int np_ = in->pts_off[d + 1] - in->pts_off[d], nf = in->fg_off[d + 1] - in->fg_off[d],
                  nb = in->bg_off[d + 1] - in->bg_off[d];
        const int fl = in->n_flip ? in->n_flip[d] : 1;
        if (np_ < 0 || nf < 0 || nb < 0 || fl < 1 || fl > 64)
            return qsp_fail(QSP_ERR_INVALID, "qsp_refine_detections: offsets must not decrease, 1 <= n_flip <= 64");
        n_pts[d] = np_;
        n_fg[d] = nf;
        n_rays[d] = nf + nb;
        hyp_off[d + 1] = hyp_off[d] + fl;
        for (int k = 0; k < fl; ++k) hyp_obj.push_back(d);
        max_flip = std::max(max_flip, fl);
        max_items = std::max(max_items, std::max(np_, 3 * (nf + nb)));
    }
    const int n_hyp = hyp_off[n];
    const size_t tot_pts = in->pts_off[n] - in->pts_off[0], tot_fg = in->fg_off[n] - in->fg_off[0],
                 tot_bg = in->bg_off[n] - in->bg_off[0];
    if ((tot_pts && !in->pts_world) || (tot_fg && (!in->fg_px || !in->fg_world)) || (tot_bg && !in->bg_rays))
        return qsp_fail(QSP_ERR_INVALID, "qsp_refine_detections: observation array missing");
    if (in->pts_off[0] || in->fg_off[0] || in->bg_off[0]) return qsp_fail(QSP_ERR_INVALID, "qsp_refine_detections: offsets start at 0");

    // AngleAxisf(double(k) * flip_sample_angle, e_y).matrix(): the angle is narrowed to float, then libm's cosf / sinf
    std::vector<float> Ry((size_t)max_flip * 9, 0.f);
    for (int k = 0; k < max_flip; ++k) {
        const float a = (float)((double)k * in->flip_angle);
        const float c = cosf(a), s = sinf(a);
        float* R = &Ry[9 * k];
        R[0] = c;
        R[2] = s;
        R[4] = (1.0f - c) + c;
        R[6] = 0.0f - s;
        R[8] = c;
    }

    qsp_refine_batch* b = nullptr;
    RefineCfg c{cfg->k1, cfg->k2, cfg->k3, cfg->k4, cfg->b1, cfg->b2, cfg->lr, cfg->s_damp, cfg->cut_off, cfg->n_depth, 0, 0,
                dec->code_len};
    int rc = batch_create(dec, c, cfg->n_iter, n, nullptr, n_pts.data(), nullptr, n_rays.data(), nullptr, n_fg.data(), n_hyp,
                          hyp_obj.data(), &b, true);
    if (rc) return rc;
    struct Guard {
        qsp_refine_batch* b;
        ~Guard() { batch_free(b); }
    } guard{b};

    DevPool pool;
    Inputs I{};
    if ((rc = pool.up(in->T_cw, (size_t)n * 16, &I.T_cw)) || (rc = pool.up(in->K, (size_t)n * 4, &I.K)) ||
        (rc = pool.up(in->T_wo, (size_t)n * 16, &I.T_wo)) || (rc = pool.up(hyp_off.data(), (size_t)n + 1, &I.hyp_off)) ||
        (rc = pool.up(in->pts_off, (size_t)n + 1, &I.pts_off)) || (rc = pool.up(in->fg_off, (size_t)n + 1, &I.fg_off)) ||
        (rc = pool.up(in->bg_off, (size_t)n + 1, &I.bg_off)) || (rc = pool.up(in->pts_world, tot_pts * 3, &I.pts_world)) ||
        (rc = pool.up(in->fg_px, tot_fg * 2, &I.fg_px)) || (rc = pool.up(in->fg_world, tot_fg * 3, &I.fg_world)) ||
        (rc = pool.up(in->bg_rays, tot_bg * 3, &I.bg_rays)) || (rc = pool.up(Ry.data(), Ry.size(), &I.Ry)))
        return rc;
    if (in->code && (rc = pool.up(in->code, (size_t)n * dec->code_len, &I.code))) return rc;
    I.code_len = dec->code_len;
    if (in->n_flip && (rc = pool.up(in->n_flip, (size_t)n, &I.n_flip))) return rc;
    float *d_T = nullptr, *d_code = nullptr, *d_loss = nullptr, *d_losses = nullptr, *d_tinit = nullptr;
    uint8_t* d_good = nullptr;
    int32_t* d_kept = nullptr;
    if ((rc = pool.make((size_t)n * 16, &d_T)) || (rc = pool.make((size_t)n * CODE_LEN, &d_code)) ||
        (rc = pool.make((size_t)n, &d_loss)) || (rc = pool.make((size_t)n_hyp, &d_losses)) ||
        (rc = pool.make((size_t)n, &d_good)) || (rc = pool.make((size_t)n, &d_kept)))
        return rc;
    if (out->t_cam_obj_init && (rc = pool.make((size_t)n_hyp * 16, &d_tinit))) return rc;

    hipStream_t s = dec->stream;
    const int gx = std::min(64, (max_items + 255) / 256);
    if ((rc = batch_upload(b))) return rc;      // (the object table batch_fill left in the mirror)
    hipLaunchKernelGGL(k_det_assemble, dim3(gx, n), dim3(256), 0, s, I, b->objs, b->pts, b->rays, b->depth);
    hipLaunchKernelGGL(k_det_init, dim3(n), dim3(64), 0, s, I, b->st, d_tinit);
    QSP_HIP(hipGetLastError());
    if ((rc = qsp_refine_batch_run(b, 0))) return rc;
    hipLaunchKernelGGL(k_det_select, dim3((n + 63) / 64), dim3(64), 0, s, b->st, I.hyp_off, n, d_T, d_code, d_loss, d_good, d_kept,
                       d_losses, dec->code_len);
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipStreamSynchronize(s));
#define QSP_DOWN(dst, src, count)                                                                               \
    if ((dst) && (count)) QSP_HIP(hipMemcpy((dst), (src), sizeof(*(dst)) * (size_t)(count), hipMemcpyDeviceToHost));
    QSP_DOWN(out->t_cam_obj, d_T, (size_t)n * 16)
    QSP_DOWN(out->code, d_code, (size_t)n * dec->code_len)
    QSP_DOWN(out->loss, d_loss, n)
    QSP_DOWN(out->is_good, d_good, n)
    QSP_DOWN(out->kept_flip, d_kept, n)
    QSP_DOWN(out->losses, d_losses, n_hyp)
    QSP_DOWN(out->t_cam_obj_init, d_tinit, (size_t)n_hyp * 16)
    QSP_DOWN(out->pts_cam, b->pts, tot_pts * 3)
    QSP_DOWN(out->rays, b->rays, (tot_fg + tot_bg) * 3)
#undef QSP_DOWN
    if (out->depth_obs && tot_fg) {   // the resident depth array has one slot per ray; the first n_fg of each detection are observed
        std::vector<float> all(tot_fg + tot_bg);
        QSP_HIP(hipMemcpy(all.data(), b->depth, sizeof(float) * all.size(), hipMemcpyDeviceToHost));
        size_t w = 0, r = 0;
        for (int d = 0; d < n; ++d) {
            for (int i = 0; i < n_fg[d]; ++i) out->depth_obs[w++] = all[r + i];
            r += n_rays[d];
        }
    }
    return QSP_OK;
}
