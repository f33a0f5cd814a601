// ellipsoid_fit.hpp -- the single-ellipsoid Levenberg-Marquardt problems of the reference, batched.
//
// Replaces EllipsoidExtractor::OptimizeEllipsoidUsingPlanes, src/pca/EllipsoidExtractorLocalOptimization.cpp:16-85 -- one
// VertexEllipsoidXYZABC (translation + half-axes free, rotation fixed: src/core/Ellipsoid.cpp:62-76), one unary
// EdgeEllipsoidPlane per plane whose 1-D error is the distance from the plane to the nearest tangent point of the
// ellipsoid (src/pca/EllipsoidExtractorEdges.cpp:35-175), information 1, no robust kernel, g2o's NUMERIC Jacobian
// (Thirdparty/g2o/g2o/core/base_unary_edge.hpp:82-123: central differences, delta 1e-9), BlockSolverX + dense LDLT,
// OptimizationAlgorithmLevenberg, optimize(10).  Included at the end of ba_solver.hip.
//
// The quadric BA these problems once fed is dead code in the reference, and in this revision the function itself
// is compiled but has no live caller (its call sites, src/pca/EllipsoidExtractorMultiPlanes.cpp:692-693, are commented out).
// What is worth having on the device is the shape of the problem: thousands of independent 6-unknown LM problems with a dozen
// residuals each.  One WAVE owns one ellipsoid: lane = plane (strided), every sum is an xor-butterfly all-reduce so that all
// 64 lanes hold identical totals, and every lane then runs the 6x6 solve and the LM control flow redundantly in registers --
// no shared memory, no barrier, no divergence inside the wave, and ellipsoids of a batch never wait for each other.
//
// The reference evaluates the distance through two 4x4 inverses (plane into the ellipsoid frame, dual quadric -> primal); here
// those are written in closed form (R^T n, t.n + d; axes^2).  The two agree to ~1e-16 relative, which the delta = 1e-9
// difference quotient amplifies to ~1e-7 in a Jacobian entry -- the same spread two builds of the reference have.
#pragma once
#include <float.h>

namespace qsp {
namespace ell {

struct FitIn {
    int n, n_iter, direction;
    const double* ell;        // [n][10] t(3) q(x y z w) half-axes(3)   (g2o::ellipsoid::toVector)
    const int32_t* off;       // [n+1]
    const double* planes;     // [off[n]][4]
};

__device__ inline double wave_sum(double a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    return a;
}

// EdgeEllipsoidPlane::computeError (direction == 0) / EdgeSE3EllipsoidPlane::computeError with an identity camera
// (direction != 0: GetDistanceWithDirection + the NaN check, EllipsoidExtractorEdges.cpp:151-226)
__device__ inline double plane_error(const double* t, const double* R, const double* s, const double* pl, int direction) {
    const double n0 = pl[0], n1 = pl[1], n2 = pl[2];
    const double A = R[0] * n0 + R[3] * n1 + R[6] * n2, B = R[1] * n0 + R[4] * n1 + R[7] * n2,
                 C = R[2] * n0 + R[5] * n1 + R[8] * n2;
    const double D = t[0] * n0 + t[1] * n1 + t[2] * n2 + pl[3];
    const double a2 = s[0] * s[0], b2 = s[1] * s[1], c2 = s[2] * s[2];
    const double alpha = sqrt(4.0 / (A * A * a2 + B * B * b2 + C * C * c2));
    const double ex = alpha * (A * a2 / 2), ey = alpha * (B * b2 / 2), ez = alpha * (C * c2 / 2);
    const double den = sqrt(A * A + B * B + C * C);
    const double d1 = fabs((A * ex + B * ey + C * ez + D) / den), d2 = fabs((A * -ex + B * -ey + C * -ez + D) / den);
    const double nearest = d1 > d2 ? d2 : d1, farthest = d1 > d2 ? d1 : d2;
    if (!direction) return nearest;
    const double centre = (n0 * t[0] + n1 * t[1] + n2 * t[2] + pl[3]) / sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    double dis = centre > 0 ? nearest : farthest;
    if (isnan(dis)) dis = 0;
    return dis;
}

__global__ void __launch_bounds__(64) k_ellipsoid_fit(FitIn in, double* out_ell, double* out_chi2, int32_t* out_iters, double* trace) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const double* E = in.ell + 10 * (int64_t)w;
    const int p0 = in.off[w], np_ = in.off[w + 1] - p0;
    double est[6] = {E[0], E[1], E[2], E[7], E[8], E[9]};       // translation, half-axes
    double R[9];
    {   // Eigen::Quaterniond::toRotationMatrix
        const double x = E[3], y = E[4], z = E[5], qw = E[6];
        const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * x, txy = ty * x,
                     txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
        R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
        R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
        R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
    }
    auto chi2_at = [&](const double* v) {
        double a = 0;
        for (int k = lane; k < np_; k += 64) {
            const double e = plane_error(v, R, v + 3, in.planes + 4 * (int64_t)(p0 + k), in.direction);
            a += e * e;
        }
        return wave_sum(a);
    };
    int done = 0;
    double cur = 0, lambda = 0, ni = 2;
    int nbad = 0;
    if (np_ > 0) {
        for (int it = 0; it < in.n_iter; ++it) {
            // computeActiveErrors + buildSystem with the numeric Jacobian
            double v[28];
            for (int i = 0; i < 28; ++i) v[i] = 0;
            for (int k = lane; k < np_; k += 64) {
                const double* pl = in.planes + 4 * (int64_t)(p0 + k);
                const double e = plane_error(est, R, est + 3, pl, in.direction);
                double J[6];
                const double delta = 1e-9, scalar = 1.0 / (2 * delta);
                for (int d = 0; d < 6; ++d) {
                    double q[6];
                    for (int i = 0; i < 6; ++i) q[i] = est[i];
                    q[d] = est[d] + delta;
                    const double e1 = plane_error(q, R, q + 3, pl, in.direction);
                    q[d] = est[d] + -delta;
                    const double e2 = plane_error(q, R, q + 3, pl, in.direction);
                    J[d] = scalar * (e1 - e2);
                }
                int q = 0;
                for (int i = 0; i < 6; ++i) {
                    v[21 + i] += J[i] * -e;
                    for (int j = i; j < 6; ++j) v[q++] += J[i] * J[j];
                }
                v[27] += e * e;
            }
            double tot[28];
            for (int i = 0; i < 28; ++i) tot[i] = wave_sum(v[i]);
            cur = tot[27];
            const double ini = cur;
            if (it == 0) {                                           // computeLambdaInit
                double md = 0;
                int q = 0;
                for (int i = 0; i < 6; ++i) { md = fmax(fabs(tot[q]), md); q += 6 - i; }
                lambda = 1e-5 * md; ni = 2; nbad = 0;
            }
            int qmax = 0;
            double rho = 0;
            do {                                                     // LM trials
                double bk[6], x[6] = {0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 6; ++i) bk[i] = est[i];
                double Am[6][6];
                int q = 0;
                bool zero = true;
                for (int i = 0; i < 6; ++i)
                    for (int j = i; j < 6; ++j) { Am[i][j] = tot[q]; Am[j][i] = tot[q]; zero = zero && tot[q] == 0; ++q; }
                for (int i = 0; i < 6; ++i) Am[i][i] += lambda;
                bool ok = true;
                if (!(zero && lambda == 0)) {                        // (LDLT of the zero matrix: semi-definite, x = 0)
                    for (int j = 0; j < 6 && ok; ++j) {
                        double dd = Am[j][j];
                        for (int q2 = 0; q2 < j; ++q2) dd -= Am[j][q2] * Am[j][q2];
                        if (!(dd > 0) || !isfinite(dd)) { ok = false; break; }
                        const double l = sqrt(dd);
                        Am[j][j] = l;
                        for (int i = j + 1; i < 6; ++i) {
                            double s2 = Am[i][j];
                            for (int q2 = 0; q2 < j; ++q2) s2 -= Am[i][q2] * Am[j][q2];
                            Am[i][j] = s2 / l;
                        }
                    }
                    if (ok) {
                        double y[6];
                        for (int i = 0; i < 6; ++i) {
                            double s2 = tot[21 + i];
                            for (int q2 = 0; q2 < i; ++q2) s2 -= Am[i][q2] * y[q2];
                            y[i] = s2 / Am[i][i];
                        }
                        for (int i = 5; i >= 0; --i) {
                            double s2 = y[i];
                            for (int q2 = i + 1; q2 < 6; ++q2) s2 -= Am[q2][i] * x[q2];
                            x[i] = s2 / Am[i][i];
                        }
                    }
                }
                if (ok)
                    for (int i = 0; i < 6; ++i) est[i] = est[i] + x[i];          // exp_update_XYZABC
                double tempChi = chi2_at(est);
                if (!ok) tempChi = DBL_MAX;
                rho = cur - tempChi;
                double scale = 1e-3;
                for (int i = 0; i < 6; ++i) scale += x[i] * (lambda * x[i] + tot[21 + i]);
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow(2 * rho - 1, 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    cur = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    for (int i = 0; i < 6; ++i) est[i] = bk[i];
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            ++done;
            if (trace && lane == 0) {
                double* tr = trace + ((int64_t)w * in.n_iter + it) * 3;
                tr[0] = cur; tr[1] = lambda; tr[2] = (double)qmax;
            }
            if (qmax == 10 || rho == 0) break;
            if ((ini - cur) * 1e3 < ini) nbad++; else nbad = 0;
            if (nbad >= 3) break;
        }
    }
    if (lane == 0) {
        double* O = out_ell + 10 * (int64_t)w;
        for (int i = 0; i < 3; ++i) O[i] = est[i];
        for (int i = 3; i < 7; ++i) O[i] = E[i];
        for (int i = 0; i < 3; ++i) O[7 + i] = est[3 + i];
        out_chi2[w] = np_ > 0 ? cur : 0.0;
        out_iters[w] = done;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// priorInfer::infer's problem (src/core/PriorInfer.cpp:331-427), batched the same way: one wave per ellipsoid, lane = edge.
// One VertexEllipsoidXYZABCYaw (7 unknowns; oplus = exp_update_XYZABCYaw, src/core/Ellipsoid.cpp:78-106: the pose is multiplied
// from the right by SE3(zyx_euler_to_quat(0, 0, yaw), trans) -- se3quat.h:110-116,208-225 -- and the half-axes add), a fixed
// identity camera, and three edge kinds in g2o's id order: EdgeSE3EllipsoidPlaneWithNormal (2-D, Huber), EdgeSE3EllipsoidPlane with
// the normal-direction rule (1-D, Huber), EdgePri (2-D).  Numeric Jacobians (delta 1e-9), dense 7 x 7 Levenberg-Marquardt.
// ---------------------------------------------------------------------------------------------------------------------------
struct PriorIn {
    int n, n_iter;
    const double* ell;         // [n][10]
    const int32_t* off_pn;     // [n+1] planes with normal
    const double* planes_pn;
    const int32_t* off_pl;     // [n+1] planes (normal-direction rule)
    const double* planes_pl;
    const double* pri;         // [n][2] prior (d, e)
    const double* weight;      // [n] weight of the prior edge
    const double* ground_w;    // [n] weight of the FIRST plane of each list (bUseGroundPlaneWeight); < 0: 1
    double sigma;              // angle sigma in radians
};
struct EllState { double t[3], q[4], s[3]; };

__device__ inline void quat_R(const double* q, double* R) {      // Eigen::Quaterniond::toRotationMatrix, q = x y z w
    const double x = q[0], y = q[1], z = q[2], qw = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * x, txy = ty * x,
                 txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__device__ inline EllState yaw_update(const EllState& a, const double* u) {
    EllState o;
    double R[9];
    quat_R(a.q, R);
    for (int i = 0; i < 3; ++i) o.t[i] = a.t[i] + (R[3 * i] * u[0] + R[3 * i + 1] * u[1] + R[3 * i + 2] * u[2]);
    const double bz = sin(u[6] * 0.5), bw = cos(u[6] * 0.5);
    const double ax = a.q[0], ay = a.q[1], az = a.q[2], aw = a.q[3];
    double q[4] = {ax * bw + ay * bz, ay * bw - ax * bz, aw * bz + az * bw, aw * bw - az * bz};      // a (x) (0, 0, bz, bw)
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) o.q[i] = q[i] / nrm;
    for (int i = 0; i < 3; ++i) o.s[i] = a.s[i] + u[3 + i];
    return o;
}
// EdgeSE3EllipsoidPlaneWithNormal::calculateMinAngle (EllipsoidExtractorEdges.cpp:297-358); R^-1 n as R^T n
__device__ inline double min_angle(const double* n, const double* R) {
    const double c0 = R[0] * n[0] + R[3] * n[1] + R[6] * n[2], c1 = R[1] * n[0] + R[4] * n[1] + R[7] * n[2],
                 c2 = R[2] * n[0] + R[5] * n[1] + R[8] * n[2];
    const double az = acos(c2 / sqrt(c0 * c0 + c1 * c1 + c2 * c2));
    const double pi = 3.14159265358979323846;
    if (fmin(fabs(az), fabs(az - pi)) < pi / 180.0 * 30) return 0.0;
    const double ang = acos(c0 / sqrt(c0 * c0 + c1 * c1));
    return fmin(fmin(ang, fabs(ang - pi / 2)), fabs(ang - pi));
}
// residual of edge `k` (0 .. n_pn + n_pl: the last one is the prior) at state st; returns its dimension
__device__ inline int prior_edge(const PriorIn& in, int w, int k, int n_pn, int n_pl, const EllState& st, double* e) {
    double R[9];
    quat_R(st.q, R);
    if (k < n_pn) {
        const double* pl = in.planes_pn + 4 * (int64_t)(in.off_pn[w] + k);
        e[0] = plane_error(st.t, R, st.s, pl, 0);
        e[1] = min_angle(pl, R);
        return 2;
    }
    if (k < n_pn + n_pl) {
        e[0] = plane_error(st.t, R, st.s, in.planes_pl + 4 * (int64_t)(in.off_pl[w] + k - n_pn), 1);
        return 1;
    }
    double a = fabs(st.s[0]), b = fabs(st.s[1]), c = fabs(st.s[2]);      // Pri(ellipsoid): (mid / min, max / min)
    if (a > b) { const double t_ = a; a = b; b = t_; }
    if (b > c) { const double t_ = b; b = c; c = t_; }
    if (a > b) { const double t_ = a; a = b; b = t_; }
    e[0] = b / a - in.pri[2 * w];
    e[1] = c / a - in.pri[2 * w + 1];
    return 2;
}
__device__ inline void prior_omega(const PriorIn& in, int w, int k, int n_pn, int n_pl, double* om, bool* robust) {
    const double gw = in.ground_w[w];
    if (k < n_pn) {
        const double wk = (gw >= 0 && k == 0) ? gw : 1.0;
        om[0] = wk * wk; om[1] = (wk / in.sigma) * (wk / in.sigma); *robust = true;
    } else if (k < n_pn + n_pl) {
        const double wk = (gw >= 0 && k == n_pn) ? gw : 1.0;
        om[0] = wk * wk; om[1] = 0; *robust = true;
    } else {
        om[0] = om[1] = in.weight[w] * in.weight[w]; *robust = false;
    }
}
__device__ inline void huber1(double e2, double* r0, double* r1) {      // RobustKernelHuber, delta = 1 (robust_kernel_impl.cpp:78-91)
    if (e2 <= 1.0) { *r0 = e2; *r1 = 1.0; return; }
    const double r = sqrt(e2);
    *r0 = 2 * r - 1.0; *r1 = 1.0 / r;
}

__global__ void __launch_bounds__(64) k_ellipsoid_prior_fit(PriorIn in, double* out_ell, double* out_chi2, int32_t* out_iters, double* trace) {
    const int w = blockIdx.x, lane = threadIdx.x;
    const double* E = in.ell + 10 * (int64_t)w;
    const int n_pn = in.off_pn[w + 1] - in.off_pn[w], n_pl = in.off_pl[w + 1] - in.off_pl[w], n_e = n_pn + n_pl + 1;
    EllState st;
    for (int i = 0; i < 3; ++i) { st.t[i] = E[i]; st.s[i] = E[7 + i]; }
    for (int i = 0; i < 4; ++i) st.q[i] = E[3 + i];
    auto chi2_at = [&](const EllState& v) {
        double a = 0;
        for (int k = lane; k < n_e; k += 64) {
            double e[2] = {0, 0}, om[2];
            bool rob;
            const int dim = prior_edge(in, w, k, n_pn, n_pl, v, e);
            prior_omega(in, w, k, n_pn, n_pl, om, &rob);
            double c = om[0] * e[0] * e[0];
            if (dim == 2) c += om[1] * e[1] * e[1];
            double r0 = c, r1 = 1;
            if (rob) huber1(c, &r0, &r1);
            a += r0;
        }
        return wave_sum(a);
    };
    constexpr int N = 7, NT = N * (N + 1) / 2;
    int done = 0;
    double cur = 0, lambda = 0, ni = 2;
    int nbad = 0;
    for (int it = 0; it < in.n_iter; ++it) {
        double v[NT + N + 1];
        for (int i = 0; i < NT + N + 1; ++i) v[i] = 0;
        for (int k = lane; k < n_e; k += 64) {
            double e[2] = {0, 0}, om[2];
            bool rob;
            const int dim = prior_edge(in, w, k, n_pn, n_pl, st, e);
            prior_omega(in, w, k, n_pn, n_pl, om, &rob);
            double J[2][N];
            const double delta = 1e-9, scalar = 1.0 / (2 * delta);
            for (int d = 0; d < N; ++d) {
                double u[N] = {0, 0, 0, 0, 0, 0, 0}, e1[2] = {0, 0}, e2[2] = {0, 0};
                u[d] = delta;
                prior_edge(in, w, k, n_pn, n_pl, yaw_update(st, u), e1);
                u[d] = -delta;
                prior_edge(in, w, k, n_pn, n_pl, yaw_update(st, u), e2);
                J[0][d] = scalar * (e1[0] - e2[0]);
                J[1][d] = scalar * (e1[1] - e2[1]);
            }
            double c = om[0] * e[0] * e[0];
            if (dim == 2) c += om[1] * e[1] * e[1];
            double r0 = c, r1 = 1;
            if (rob) huber1(c, &r0, &r1);
            int q = 0;
            for (int i = 0; i < N; ++i) {
                for (int r = 0; r < dim; ++r) v[NT + i] -= J[r][i] * (r1 * om[r]) * e[r];
                for (int j = i; j < N; ++j) {
                    double a = 0;
                    for (int r = 0; r < dim; ++r) a += J[r][i] * (r1 * om[r]) * J[r][j];
                    v[q++] += a;
                }
            }
            v[NT + N] += r0;
        }
        double tot[NT + N + 1];
        for (int i = 0; i < NT + N + 1; ++i) tot[i] = wave_sum(v[i]);
        cur = tot[NT + N];
        const double ini = cur;
        if (it == 0) {
            double md = 0;
            int q = 0;
            for (int i = 0; i < N; ++i) { md = fmax(fabs(tot[q]), md); q += N - i; }
            lambda = 1e-5 * md; ni = 2; nbad = 0;
        }
        int qmax = 0;
        double rho = 0;
        do {
            const EllState bk = st;
            double x[N] = {0, 0, 0, 0, 0, 0, 0};
            double Am[N][N];
            int q = 0;
            bool zero = true;
            for (int i = 0; i < N; ++i)
                for (int j = i; j < N; ++j) { Am[i][j] = tot[q]; Am[j][i] = tot[q]; zero = zero && tot[q] == 0; ++q; }
            for (int i = 0; i < N; ++i) Am[i][i] += lambda;
            bool ok = true;
            if (!(zero && lambda == 0)) {
                for (int j = 0; j < N && ok; ++j) {
                    double dd = Am[j][j];
                    for (int q2 = 0; q2 < j; ++q2) dd -= Am[j][q2] * Am[j][q2];
                    if (!(dd > 0) || !isfinite(dd)) { ok = false; break; }
                    const double l = sqrt(dd);
                    Am[j][j] = l;
                    for (int i = j + 1; i < N; ++i) {
                        double s2 = Am[i][j];
                        for (int q2 = 0; q2 < j; ++q2) s2 -= Am[i][q2] * Am[j][q2];
                        Am[i][j] = s2 / l;
                    }
                }
                if (ok) {
                    double y[N];
                    for (int i = 0; i < N; ++i) {
                        double s2 = tot[NT + i];
                        for (int q2 = 0; q2 < i; ++q2) s2 -= Am[i][q2] * y[q2];
                        y[i] = s2 / Am[i][i];
                    }
                    for (int i = N - 1; i >= 0; --i) {
                        double s2 = y[i];
                        for (int q2 = i + 1; q2 < N; ++q2) s2 -= Am[q2][i] * x[q2];
                        x[i] = s2 / Am[i][i];
                    }
                }
            }
            if (ok) st = yaw_update(st, x);
            double tempChi = chi2_at(st);
            if (!ok) tempChi = DBL_MAX;
            rho = cur - tempChi;
            double scale = 1e-3;
            for (int i = 0; i < N; ++i) scale += x[i] * (lambda * x[i] + tot[NT + i]);
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                cur = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                st = bk;
            }
            qmax++;
        } while (rho < 0 && qmax < 10);
        ++done;
        if (trace && lane == 0) {
            double* tr = trace + ((int64_t)w * in.n_iter + it) * 3;
            tr[0] = cur; tr[1] = lambda; tr[2] = (double)qmax;
        }
        if (qmax == 10 || rho == 0) break;
        if ((ini - cur) * 1e3 < ini) nbad++; else nbad = 0;
        if (nbad >= 3) break;
    }
    if (lane == 0) {
        double* O = out_ell + 10 * (int64_t)w;
        for (int i = 0; i < 3; ++i) { O[i] = st.t[i]; O[7 + i] = st.s[i]; }
        for (int i = 0; i < 4; ++i) O[3 + i] = st.q[i];
        out_chi2[w] = cur;
        out_iters[w] = done;
    }
}

}  // namespace ell
}  // namespace qsp

extern "C" int qsp_ellipsoid_fit_planes(int device, int32_t n, const double* ellipsoid_in, const int32_t* plane_off,
                                        const double* planes, int32_t n_iter, int32_t normal_direction, double* ellipsoid_out,
                                        double* chi2_out, int32_t* iters_out, double* trace) {
    using namespace qsp;
    if (n <= 0 || !ellipsoid_in || !plane_off || !ellipsoid_out) return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_planes: bad argument");
    if (n_iter <= 0) n_iter = 10;
    if (plane_off[0] != 0) return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_planes: offsets start at 0");
    for (int i = 0; i < n; ++i)
        if (plane_off[i + 1] < plane_off[i]) return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_planes: offsets must not decrease");
    const size_t np_ = (size_t)plane_off[n];
    if (np_ && !planes) return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_planes: planes missing");
    QSP_HIP(hipSetDevice(device));
    struct Pool {
        std::vector<void*> p;
        ~Pool() { for (void* q : p) (void)hipFree(q); }
    } pool;
    auto dev = [&](size_t bytes, void** out) {
        hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 8));
        if (e == hipSuccess) pool.p.push_back(*out);
        return e;
    };
    double *d_ell, *d_pl, *d_out, *d_chi, *d_tr = nullptr;
    int32_t *d_off, *d_it;
    QSP_HIP(dev(sizeof(double) * 10 * n, (void**)&d_ell));
    QSP_HIP(dev(sizeof(double) * 4 * np_, (void**)&d_pl));
    QSP_HIP(dev(sizeof(double) * 10 * n, (void**)&d_out));
    QSP_HIP(dev(sizeof(double) * n, (void**)&d_chi));
    QSP_HIP(dev(sizeof(int32_t) * (n + 1), (void**)&d_off));
    QSP_HIP(dev(sizeof(int32_t) * n, (void**)&d_it));
    if (trace) {
        QSP_HIP(dev(sizeof(double) * 3 * (size_t)n * n_iter, (void**)&d_tr));
        QSP_HIP(hipMemset(d_tr, 0, sizeof(double) * 3 * (size_t)n * n_iter));
    }
    QSP_HIP(hipMemcpy(d_ell, ellipsoid_in, sizeof(double) * 10 * n, hipMemcpyHostToDevice));
    if (np_) QSP_HIP(hipMemcpy(d_pl, planes, sizeof(double) * 4 * np_, hipMemcpyHostToDevice));
    QSP_HIP(hipMemcpy(d_off, plane_off, sizeof(int32_t) * (n + 1), hipMemcpyHostToDevice));
    ell::FitIn in{n, n_iter, normal_direction ? 1 : 0, d_ell, d_off, d_pl};
    hipLaunchKernelGGL(ell::k_ellipsoid_fit, dim3(n), dim3(64), 0, 0, in, d_out, d_chi, d_it, d_tr);
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipDeviceSynchronize());
    QSP_HIP(hipMemcpy(ellipsoid_out, d_out, sizeof(double) * 10 * n, hipMemcpyDeviceToHost));
    if (chi2_out) QSP_HIP(hipMemcpy(chi2_out, d_chi, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (iters_out) QSP_HIP(hipMemcpy(iters_out, d_it, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (trace) QSP_HIP(hipMemcpy(trace, d_tr, sizeof(double) * 3 * (size_t)n * n_iter, hipMemcpyDeviceToHost));
    return QSP_OK;
}


extern "C" int qsp_ellipsoid_fit_prior(int device, int32_t n, const double* ellipsoid_in, const int32_t* off_normal, const double* planes_normal,
                                       const int32_t* off_plane, const double* planes, const double* pri, const double* weight,
                                       const double* ground_plane_weight, double angle_sigma_deg, int32_t n_iter, double* ellipsoid_out,
                                       double* chi2_out, int32_t* iters_out, double* trace) {
    using namespace qsp;
    if (n <= 0 || !ellipsoid_in || !off_normal || !off_plane || !pri || !weight || !ellipsoid_out)
        return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_prior: bad argument");
    if (n_iter <= 0) n_iter = 10;
    if (!(angle_sigma_deg > 0)) return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_prior: angle sigma must be positive");
    if (off_normal[0] != 0 || off_plane[0] != 0) return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_prior: offsets start at 0");
    for (int i = 0; i < n; ++i)
        if (off_normal[i + 1] < off_normal[i] || off_plane[i + 1] < off_plane[i])
            return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_prior: offsets must not decrease");
    const size_t n_pn = (size_t)off_normal[n], n_pl = (size_t)off_plane[n];
    if ((n_pn && !planes_normal) || (n_pl && !planes)) return qsp_fail(QSP_ERR_INVALID, "qsp_ellipsoid_fit_prior: planes missing");
    QSP_HIP(hipSetDevice(device));
    struct Pool {
        std::vector<void*> p;
        ~Pool() { for (void* q : p) (void)hipFree(q); }
    } pool;
    auto up = [&](const void* src, size_t bytes, void** out) -> hipError_t {
        hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 8));
        if (e != hipSuccess) return e;
        pool.p.push_back(*out);
        return (src && bytes) ? hipMemcpy(*out, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    std::vector<double> gw(n, -1.0);
    if (ground_plane_weight) gw.assign(ground_plane_weight, ground_plane_weight + n);
    double *d_ell, *d_pn, *d_pl, *d_pri, *d_w, *d_gw, *d_out, *d_chi, *d_tr = nullptr;
    int32_t *d_on, *d_op, *d_it;
    QSP_HIP(up(ellipsoid_in, sizeof(double) * 10 * n, (void**)&d_ell));
    QSP_HIP(up(planes_normal, sizeof(double) * 4 * n_pn, (void**)&d_pn));
    QSP_HIP(up(planes, sizeof(double) * 4 * n_pl, (void**)&d_pl));
    QSP_HIP(up(off_normal, sizeof(int32_t) * (n + 1), (void**)&d_on));
    QSP_HIP(up(off_plane, sizeof(int32_t) * (n + 1), (void**)&d_op));
    QSP_HIP(up(pri, sizeof(double) * 2 * n, (void**)&d_pri));
    QSP_HIP(up(weight, sizeof(double) * n, (void**)&d_w));
    QSP_HIP(up(gw.data(), sizeof(double) * n, (void**)&d_gw));
    QSP_HIP(up(nullptr, sizeof(double) * 10 * n, (void**)&d_out));
    QSP_HIP(up(nullptr, sizeof(double) * n, (void**)&d_chi));
    QSP_HIP(up(nullptr, sizeof(int32_t) * n, (void**)&d_it));
    if (trace) {
        QSP_HIP(up(nullptr, sizeof(double) * 3 * (size_t)n * n_iter, (void**)&d_tr));
        QSP_HIP(hipMemset(d_tr, 0, sizeof(double) * 3 * (size_t)n * n_iter));
    }
    ell::PriorIn in{n, n_iter, d_ell, d_on, d_pn, d_op, d_pl, d_pri, d_w, d_gw, angle_sigma_deg / 180.0 * 3.14159265358979323846};
    hipLaunchKernelGGL(ell::k_ellipsoid_prior_fit, dim3(n), dim3(64), 0, 0, in, d_out, d_chi, d_it, d_tr);
    QSP_HIP(hipGetLastError());
    QSP_HIP(hipDeviceSynchronize());
    QSP_HIP(hipMemcpy(ellipsoid_out, d_out, sizeof(double) * 10 * n, hipMemcpyDeviceToHost));
    if (chi2_out) QSP_HIP(hipMemcpy(chi2_out, d_chi, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (iters_out) QSP_HIP(hipMemcpy(iters_out, d_it, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (trace) QSP_HIP(hipMemcpy(trace, d_tr, sizeof(double) * 3 * (size_t)n * n_iter, hipMemcpyDeviceToHost));
    return QSP_OK;
}
