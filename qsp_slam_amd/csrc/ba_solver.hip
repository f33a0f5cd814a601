// ba_solver.hip -- path B of the hot path: the camera-point-object joint bundle adjustment on gfx950, FP64.
//
// Replaces the numeric core that src/Optimizer_util.cc:44-307,309-771 and src/Optimizer.cc:54-242,458-783 of the reference
// drive through g2o (Thirdparty/g2o: SparseOptimizer + BlockSolver_6_3 + OptimizationAlgorithmLevenberg +
// LinearSolverEigen, single-threaded, heap-allocated edges walked through pointers):
//
//   k_errors          computeActiveErrors + activeRobustChi2            (sparse_optimizer.cpp:61-114)
//   k_lin_points      per landmark: Jacobians of its edges, Hll, b_l (the 6x3 Hpl blocks are recomputed where needed)
//   k_lin_poses       per key-frame: segmented reduction of J^T W J / J^T W e over its edges (Hpp diagonal, b_p)
//   k_lin_objects     per object: its camera-object edges (Hpp diagonal, off-diagonal blocks, b_p)
//                                                                       (block_solver.hpp:502-560, base_binary_edge.hpp:55-120)
//   k_schur_prepare / k_schur_points   Hschur = Hpp + lambda I - sum_l B_l D_l^-1 B_l^T, b - B D^-1 b_l   (block_solver.hpp:381-432)
//   k_chol_*          blocked dense Cholesky of the reduced system (the reference: Eigen SimplicialLDLT, linear_solver_eigen.h:94-124)
//   k_trsv            forward/backward substitution
//   k_update_*        back-substitution x_l = D^-1 (b_l - B^T x_p), oplus on every vertex                 (block_solver.hpp:461-481)
//
// Layout: edges are physically re-ordered landmark-major at creation, so the landmark pass streams them; the key-frame
// pass gathers through a CSR.  Reductions are fixed-order (deterministic) except the Schur accumulation, which uses
// FP64 atomics into the dense reduced matrix (order-dependent only in the last bit).
// The Levenberg-Marquardt control flow (optimization_algorithm_levenberg.cpp:61-164) stays on the host: one 32-byte
// read-back per trial.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/qsp_hip.h"
#include "common.hpp"
#include "comm_rccl.hpp"

namespace qsp {
namespace ba {

constexpr int NB = 64;   // Cholesky block

// ---------------------------------------------------------------------------------------------------------------
// SE3 helpers on (tx ty tz qx qy qz qw), following g2o's SE3Quat (Thirdparty/g2o/g2o/types/se3quat.h)
// ---------------------------------------------------------------------------------------------------------------
__host__ __device__ inline void quat_to_R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                 tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void R_to_quat(const double* m, double* q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
__host__ __device__ inline void quat_normalize(double* q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__host__ __device__ inline void quat_mul(const double* a, const double* b, double* c) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    c[0] = aw * bx + ax * bw + ay * bz - az * by;
    c[1] = aw * by + ay * bw + az * bx - ax * bz;
    c[2] = aw * bz + az * bw + ax * by - ay * bx;
    c[3] = aw * bw - ax * bx - ay * by - az * bz;
}
__host__ __device__ inline void quat_rot(const double* q, const double* v, double* o) {
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    const double uvx = 2 * (uy * v[2] - uz * v[1]), uvy = 2 * (uz * v[0] - ux * v[2]), uvz = 2 * (ux * v[1] - uy * v[0]);
    o[0] = v[0] + w * uvx + (uy * uvz - uz * uvy);
    o[1] = v[1] + w * uvy + (uz * uvx - ux * uvz);
    o[2] = v[2] + w * uvz + (ux * uvy - uy * uvx);
}
__host__ __device__ inline void se3_mul(const double* a, const double* b, double* c) {
    double rt[3], q[4];
    quat_rot(a + 3, b, rt);
    quat_mul(a + 3, b + 3, q);
    c[0] = a[0] + rt[0]; c[1] = a[1] + rt[1]; c[2] = a[2] + rt[2];
    quat_normalize(q);
    c[3] = q[0]; c[4] = q[1]; c[5] = q[2]; c[6] = q[3];
}
__host__ __device__ inline void se3_inv(const double* a, double* c) {
    const double q[4] = {-a[3], -a[4], -a[5], a[6]}, mt[3] = {-a[0], -a[1], -a[2]};
    double t[3];
    quat_rot(q, mt, t);
    c[0] = t[0]; c[1] = t[1]; c[2] = t[2]; c[3] = q[0]; c[4] = q[1]; c[5] = q[2]; c[6] = q[3];
}
__host__ __device__ inline void se3_map(const double* a, const double* x, double* o) {
    double r[3];
    quat_rot(a + 3, x, r);
    o[0] = r[0] + a[0]; o[1] = r[1] + a[1]; o[2] = r[2] + a[2];
}
__host__ __device__ inline void skew3(const double* v, double* m) {
    m[0] = 0; m[1] = -v[2]; m[2] = v[1]; m[3] = v[2]; m[4] = 0; m[5] = -v[0]; m[6] = -v[1]; m[7] = v[0]; m[8] = 0;
}
__host__ __device__ inline void mat3mul(const double* a, const double* b, double* c) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
__device__ inline void se3_exp(const double* u, double* pose) {   // se3quat.h:273-305
    const double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    double Om[9], Om2[9], R[9], V[9];
    skew3(u, Om);
    mat3mul(Om, Om, Om2);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        const double s = sin(theta), c = cos(theta);
        const double a = s / theta, b = (1 - c) / (theta * theta), g = (theta - s) / pow(theta, 3);
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + a * Om[i] + b * Om2[i]; V[i] = I[i] + b * Om[i] + g * Om2[i]; }
    }
    double q[4];
    R_to_quat(R, q);
    quat_normalize(q);
    for (int i = 0; i < 3; ++i) pose[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    pose[3] = q[0]; pose[4] = q[1]; pose[5] = q[2]; pose[6] = q[3];
}
__device__ inline void se3_log(const double* pose, double* out) {   // se3quat.h:228-265
    double R[9];
    quat_to_R(pose + 3, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    const double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    double om[3], Om[9], Om2[9], Vi[9];
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (d > 0.99999) {
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        skew3(om, Om);
        mat3mul(Om, Om, Om2);
        for (int i = 0; i < 9; ++i) Vi[i] = I[i] - 0.5 * Om[i] + (1. / 12.) * Om2[i];
    } else {
        const double theta = acos(d);
        const double f = theta / (2 * sqrt(1 - d * d));
        for (int i = 0; i < 3; ++i) om[i] = f * dR[i];
        skew3(om, Om);
        mat3mul(Om, Om, Om2);
        const double g = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
        for (int i = 0; i < 9; ++i) Vi[i] = I[i] - 0.5 * Om[i] + g * Om2[i];
    }
    out[0] = om[0]; out[1] = om[1]; out[2] = om[2];
    for (int i = 0; i < 3; ++i) out[3 + i] = Vi[3 * i] * pose[0] + Vi[3 * i + 1] * pose[1] + Vi[3 * i + 2] * pose[2];
}

// Huber (g2o/core/robust_kernel_impl.cpp:78-91); delta <= 0: no kernel
__device__ inline void huber(double e, double delta, double& rho0, double& rho1) {
    if (delta <= 0 || e <= delta * delta) { rho0 = e; rho1 = 1.0; return; }
    const double s = sqrt(e);
    rho0 = 2 * s * delta - delta * delta;
    rho1 = delta / s;
}

// ---------------------------------------------------------------------------------------------------------------
// projection edges.  D = 2 (EdgeSE3ProjectXYZ) or 3 (EdgeStereoSE3ProjectXYZ); K = fx fy cx cy bf
// (Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:79-140, .cpp:103-234)
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__device__ inline double proj_error(const double* pose, const double* X, const double* K, const double* obs, double* e,
                                    double* p) {
    se3_map(pose, X, p);
    if (D == 2) {
        e[0] = obs[0] - (p[0] / p[2] * K[0] + K[2]);
        e[1] = obs[1] - (p[1] / p[2] * K[1] + K[3]);
    } else {
        const float invz = 1.0f / (float)p[2];          // cam_project(): `const float invz`, .cpp:150-157
        const float bf = (float)K[4];
        const double u = p[0] * invz * K[0] + K[2], v = p[1] * invz * K[1] + K[3];
        e[0] = obs[0] - u;
        e[1] = obs[1] - v;
        e[2] = obs[2] - (u - bf * invz);
    }
    return p[2];
}
template <int D>
__device__ inline void proj_jacobians(const double* pose, const double* p, const double* K, double* Jp /*Dx3*/,
                                      double* Jx /*Dx6*/) {
    double R[9];
    quat_to_R(pose + 3, R);
    // one FP64 reciprocal instead of g2o's ~20 divisions (an FP64 divide is ~30 instructions on CDNA); the results
    // differ from the reference's in the last bit only
    const double fx = K[0], fy = K[1], bf = K[4];
    const double x = p[0], y = p[1];
    const double iz = 1.0 / p[2], iz2 = iz * iz;
    if (D == 2) {
        const double t[6] = {fx, 0, -x * iz * fx, 0, fy, -y * iz * fy};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                Jp[3 * i + j] = -iz * (t[3 * i] * R[j] + t[3 * i + 1] * R[3 + j] + t[3 * i + 2] * R[6 + j]);
    } else {
        for (int j = 0; j < 3; ++j) {
            Jp[j] = -fx * R[j] * iz + fx * x * R[6 + j] * iz2;
            Jp[3 + j] = -fy * R[3 + j] * iz + fy * y * R[6 + j] * iz2;
            Jp[6 + j] = Jp[j] - bf * R[6 + j] * iz2;
        }
    }
    Jx[0] = x * y * iz2 * fx; Jx[1] = -(1 + (x * x * iz2)) * fx; Jx[2] = y * iz * fx;
    Jx[3] = -iz * fx; Jx[4] = 0; Jx[5] = x * iz2 * fx;
    Jx[6] = (1 + y * y * iz2) * fy; Jx[7] = -x * y * iz2 * fy; Jx[8] = -x * iz * fy;
    Jx[9] = 0; Jx[10] = -iz * fy; Jx[11] = y * iz2 * fy;
    if (D == 3) {
        Jx[12] = Jx[0] - bf * y * iz2; Jx[13] = Jx[1] + bf * x * iz2; Jx[14] = Jx[2];
        Jx[15] = Jx[3]; Jx[16] = 0; Jx[17] = Jx[5] - bf * iz2;
    }
}
// EdgeSE3LieAlgebra, include/ObjectPoseGraph.h:69-88
__device__ inline void obj_error(const double* Tcw, const double* Tow, const double* Z, double* e, double* Zi) {
    double Towi[7], a[7], b[7];
    se3_inv(Z, Zi);
    se3_inv(Tow, Towi);
    se3_mul(Zi, Tcw, a);
    se3_mul(a, Towi, b);
    se3_log(b, e);
}
__device__ inline void obj_jacobians(const double* e, const double* Zi, double* Ji, double* Jj) {
    double J[36], W[9], T[9], R[9], TR[9], Tz[9];
    skew3(e, W);
    skew3(e + 3, T);
    for (int i = 0; i < 36; ++i) J[i] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            J[6 * i + j] = 0.5 * W[3 * i + j];
            J[6 * (i + 3) + j] = 0.5 * T[3 * i + j];
            J[6 * (i + 3) + (j + 3)] = 0.5 * W[3 * i + j];
        }
    for (int i = 0; i < 6; ++i) J[7 * i] += 1.0;
    // Adj(Zi) = [[R, 0], [t^ R, R]]   (se3quat.h:307-316)
    quat_to_R(Zi + 3, R);
    skew3(Zi, Tz);
    mat3mul(Tz, R, TR);
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) {
                double a;
                if (k < 3) a = (j < 3) ? R[3 * k + j] : 0.0;
                else a = (j < 3) ? TR[3 * (k - 3) + j] : R[3 * (k - 3) + (j - 3)];
                s += J[6 * i + k] * a;
            }
            Ji[6 * i + j] = s;
            Jj[6 * i + j] = -J[6 * i + j];
        }
}

// ---------------------------------------------------------------------------------------------------------------
// device-resident problem
// ---------------------------------------------------------------------------------------------------------------
struct Edge {            // one projection edge, landmark-major order
    double obs[3];
    double info;
    int32_t pt, kf;
    int32_t stereo;      // 0 mono, 1 stereo
    int32_t user;        // index in the caller's mono / stereo array
};

struct Dev {
    int n_kf, n_pt, n_obj, n_edge, n_oe;
    double *kf_pose, *pt_xyz, *obj_pose, *kf_K;
    double *kf_bk, *pt_bk, *obj_bk;
    Edge* edge;
    uint8_t* edge_level;
    int32_t* edge_ha;               // per edge: hessian index of its key-frame, -1 if the edge is at level 1 or the key-frame fixed (k_edge_index, per optimize() call)
    double* edge_chi2;
    int32_t *pt_off;             // CSR landmark -> [first,last) in edge[]
    int32_t *kf_off, *kf_edge;   // CSR key-frame -> edge indices
    int32_t *chunk_pt;           // [n_chunk + 1] landmark ranges whose edges fit one 256-thread workgroup
    int32_t n_chunk;
    int32_t *ksp_kf, *ksp_begin, *ksp_end;   // key-frame pass splits: (kf, [begin,end) in kf_edge)
    int32_t *ksp_first;          // [n_kf + 1] first split of each key-frame
    int32_t n_ksplit;
    double* kpart;               // [n_ksplit][27] partial J^T W J / J^T W e
    // object edges
    int32_t *oe_kf, *oe_obj;
    double* oe_meas;
    uint8_t* oe_level;
    double* oe_chi2;
    int32_t *kfo_off, *kfo_edge; // CSR key-frame -> object-edge indices
    int32_t *obo_off, *obo_edge; // CSR object    -> object-edge indices
    double oe_info;
    // hessian indices
    int32_t *kf_h, *obj_h, *pt_h;
    uint8_t* kf_fixed;              // per key-frame: fixed vertex (never in the index)
    int32_t* pt_order;              // landmarks sorted by vertex id: the order in which g2o numbers them (k_tail_reindex)
    // system
    double *Hll, *bl, *Dinv, *xl;   // per landmark (indexed by landmark, not by hessian index)
    double *Hdiag;                  // per pose block (hessian index) 6x6
    double *Hoff;                   // per object edge 6x6 (row = key-frame, col = object)
    double *oe_rec;                 // per object edge: OE_REC doubles (see k_lin_objedges)
    double *oe_F;                   // per object edge: F = Hoff (Hoo + lambda I)^-1 (6x6), object elimination (k_obj_prepare)
    double *obj_G;                  // per object: (Hoo + lambda I)^-1 (36) then (Hoo + lambda I)^-1 b_o (6)
    double *bp, *bs, *xp;           // reduced rhs / solution
    double *Hs;                     // dense reduced matrix, ld = dimp
    double *Hpl;                    // deterministic mode only: materialised 6x3 blocks (k_hpl_fill)
    int32_t *pk_ka, *pk_kb, *pk_off; // deterministic mode: key-frame pairs with common landmarks, CSR into pk_ent
    int4* pk_ent;                    //   (edge in ka, edge in kb, landmark, 0) per common landmark, in landmark order
    int32_t n_pk;
    int32_t* pk_work;                // pair ids: the n_pk_big pairs with long lists first (a workgroup each), then the rest
    int32_t n_pk_big;                //   (a wave each, four to a workgroup)
    int32_t* pko_off;                // per pair: CSR into pko_ent, the common OBJECTS of the pair (object elimination)
    int4* pko_ent;                   //   (object edge at ka, object edge at kb, object, 0), in object order
    double *Uf, *Winv, *ych;        // Cholesky: off-diagonal factor blocks, inverse diagonal factors (transposed), L^-1 b
    double *partial;                // block partials for reductions
    double *scal;                   // [0] chi2, [1] scale, [2] maxdiag, [3] chol fail flag (as double)
};

struct Par {
    double delta_mono, delta_stereo, delta_obj;
    double lambda;
    int dim, dimp, n_pose;
    int is_root;   // landmark-sharded runs: only rank 0 contributes the (already all-reduced) pose blocks and padding
    int have_hpl;  // d.Hpl holds this build's 6x3 blocks (atomic-free Schur mode)
    int n_dense;   // pose blocks in the dense reduced system: n_pose, or the free key-frames only when the objects are
    int elim;      // eliminated first (second Schur complement, see k_obj_prepare)
};

// block-wide sum of one double per thread (256 threads), fixed tree -> thread 0 holds the result
__device__ inline double block_sum_256(double v, double* sh /*>=4*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return (threadIdx.x == 0) ? ((sh[0] + sh[1]) + (sh[2] + sh[3])) : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------
// k_errors: chi2 per active edge, block partials of the robust chi2
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_errors(Dev d, Par par) {
    __shared__ double sh[4];
    double acc = 0;
    const int n_tot = d.n_edge + d.n_oe;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_tot; i += gridDim.x * 256) {
        double r0, r1, e[6];
        if (i < d.n_edge) {
            if (d.edge_level[i]) continue;
            const Edge E = d.edge[i];
            double p[3];
            double c;
            if (E.stereo) {
                proj_error<3>(d.kf_pose + 7 * E.kf, d.pt_xyz + 3 * E.pt, d.kf_K + 5 * E.kf, E.obs, e, p);
                c = E.info * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
                huber(c, par.delta_stereo, r0, r1);
            } else {
                proj_error<2>(d.kf_pose + 7 * E.kf, d.pt_xyz + 3 * E.pt, d.kf_K + 5 * E.kf, E.obs, e, p);
                c = E.info * (e[0] * e[0] + e[1] * e[1]);
                huber(c, par.delta_mono, r0, r1);
            }
            d.edge_chi2[i] = c;
            acc += r0;
        } else {
            const int k = i - d.n_edge;
            if (d.oe_level[k]) continue;
            double Zi[7];
            obj_error(d.kf_pose + 7 * d.oe_kf[k], d.obj_pose + 7 * d.oe_obj[k], d.oe_meas + 7 * k, e, Zi);
            double c = 0;
            for (int q = 0; q < 6; ++q) c += e[q] * e[q];
            c *= d.oe_info;
            d.oe_chi2[k] = c;
            huber(c, par.delta_obj, r0, r1);
            if (par.is_root) acc += r0;          // camera-object edges are replicated on every rank of a sharded run
        }
    }
    const double s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) d.partial[blockIdx.x] = s;
}

// sums `n` partials in a fixed order into scal[slot]
__global__ __launch_bounds__(64) void k_finish_sum(Dev d, int n, int slot) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 64) s += d.partial[i];     // fixed order: lane-strided, then a fixed tree
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) d.scal[slot] = s;
}
// the same sum into scal[0], published to the host-visible copy together with scal[1..3] in the same launch (one rank: no
// reduction over ranks sits between the sum and the read-back); re-arms scal[3] like k_publish_scal
__global__ __launch_bounds__(64) void k_finish_sum_publish(Dev d, int n, double* __restrict__ host, int gp, int part_off, double seq) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 64) s += d.partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        // the rho denominator: the pose block's partial, then the landmark blocks' in order (k_update_poses' summation order)
        // (gp < 0: the update ran as two launches and k_update_poses has left the sum in scal[1])
        double tot = gp >= 0 ? d.partial[part_off + gp] : d.scal[1];
        for (int i = 0; i < gp; ++i) tot += d.partial[part_off + i];
        d.scal[0] = s;
        d.scal[1] = tot;
        host[0] = s;
        host[1] = tot;
        host[2] = d.scal[2];
        host[3] = d.scal[5] != 0.0 ? 2.0 : d.scal[3];      // (scal[5]: a flag wait of the chain factorisation expired -- sticky)
        d.scal[3] = 0.0;
        __threadfence_system();
        *reinterpret_cast<volatile double*>(host + 4) = seq;
    }
}

// scal[0..3] -> the host-visible copy; scal[3] (the "a block was not positive definite" flag of a trial) is re-armed for the
// next trial here, which saves a memset launch per trial
__global__ void k_publish_scal(Dev d, double* __restrict__ host, double seq) {
    if (threadIdx.x == 0) {
        for (int i = 0; i < 4; ++i) host[i] = d.scal[i];
        if (d.scal[5] != 0.0) host[3] = 2.0;
        d.scal[3] = 0.0;
        __threadfence_system();
        *reinterpret_cast<volatile double*>(host + 4) = seq;      // the host polls this word (wait_scal)
    }
}

// Sum over the 64 lanes of a wave on the VALU alone (DPP), result valid in lanes 48..63.  __shfl_xor goes through the LDS
// crossbar (two ds_bpermute per double and level), which all waves of a compute unit share: in k_chol_back 16 waves x 4 rows x
// 12 of them per step kept the LDS busy for 4.8 k of a step's 16 k cycles (tools/cb_stamps.py), in k_schur_pairs the butterfly
// over a pair's 42 sums was 504 of them per wave.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_term(double v) {
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, ROW_MASK, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, ROW_MASK, 0xf, false);
    return r.d;                                  // lanes outside ROW_MASK keep the `old` operand: +0.0
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_term<0xB1, 0xf>(v);                 // quad_perm [1,0,3,2]
    v += dpp_term<0x4E, 0xf>(v);                 // quad_perm [2,3,0,1]: every lane holds its quad's sum
    v += dpp_term<0x141, 0xf>(v);                // row_half_mirror: sums of 8
    v += dpp_term<0x140, 0xf>(v);                // row_mirror: every lane holds its row's (16 lanes) sum
    v += dpp_term<0x142, 0xa>(v);                // row_bcast15 into rows 1 and 3: rows 0+1, rows 2+3
    v += dpp_term<0x143, 0xc>(v);                // row_bcast31 into rows 2 and 3: row 3 holds the total
    return v;
}

// Sums of N <= 64 values over the 64 lanes at once: at each of the six levels a lane keeps the half of the values whose index has
// the lane's bit set like its own and adds the partner lane's copy of them, so the number of live values halves while the number
// of lanes that contributed doubles.  Afterwards lane L holds the complete sum of value L -- 1 + 1/2 + 1/4 + .. of the work of
// reducing every value on its own.  Partners: quad_perm (lane ^ 1, ^ 2), row_ror 4 / 8 (the lanes of a row that share lane & 3,
// then lane & 7), v_permlane16_swap / v_permlane32_swap (lane ^ 16, ^ 32).  VALU only, fixed order.
__device__ __forceinline__ double lane_xor16(double b, bool odd_row) {
    union { double d; unsigned u[2]; } x, r;
    x.d = b;
    const auto s0 = __builtin_amdgcn_permlane16_swap(x.u[0], x.u[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(x.u[1], x.u[1], false, false);
    r.u[0] = odd_row ? s0[0] : s0[1];
    r.u[1] = odd_row ? s1[0] : s1[1];
    return r.d;
}
__device__ __forceinline__ double lane_xor32(double b, bool upper) {
    union { double d; unsigned u[2]; } x, r;
    x.d = b;
    const auto s0 = __builtin_amdgcn_permlane32_swap(x.u[0], x.u[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(x.u[1], x.u[1], false, false);
    r.u[0] = upper ? s0[0] : s0[1];
    r.u[1] = upper ? s1[0] : s1[1];
    return r.d;
}
template <int LEVEL>
__device__ __forceinline__ double lane_partner(double b, int lane) {
    if (LEVEL == 0) return dpp_term<0xB1, 0xf>(b);            // quad_perm [1,0,3,2]
    if (LEVEL == 1) return dpp_term<0x4E, 0xf>(b);            // quad_perm [2,3,0,1]
    if (LEVEL == 2) return dpp_term<0x124, 0xf>(b);           // row_ror:4
    if (LEVEL == 3) return dpp_term<0x128, 0xf>(b);           // row_ror:8
    if (LEVEL == 4) return lane_xor16(b, (lane & 16) != 0);
    return lane_xor32(b, (lane & 32) != 0);
}
template <int N, int LEVEL>
struct LaneTranspose {
    static __device__ __forceinline__ void run(double* v, int lane) {      // v[0..N) in, lane L ends with the sum of value L in v[0]
        constexpr int H = (N + 1) / 2;
        const bool bit = (lane >> LEVEL) & 1;
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const double lo = v[2 * m], hi = (2 * m + 1 < N) ? v[2 * m + 1] : 0.0;
            const double keep = bit ? hi : lo, give = bit ? lo : hi;
            v[m] = keep + lane_partner<LEVEL>(give, lane);
        }
        LaneTranspose<H, LEVEL + 1>::run(v, lane);
    }
};
template <int N>
struct LaneTranspose<N, 6> {
    static __device__ __forceinline__ void run(double*, int) {}
};

// ---------------------------------------------------------------------------------------------------------------
// k_lin_points: edge-parallel.  A workgroup owns a chunk of consecutive landmarks whose edges (contiguous, landmark-
// major) number at most 256: thread = edge.  Each thread parks its J_p^T W J_p (6 unique)
// and J_p^T W e (3) in LDS; the first `n landmarks` threads then sum their landmark's segment in edge order.
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__device__ inline void lin_point_edge(const Dev& d, const Edge& E, int ei, double delta, double* out9) {
    double e[3], p[3], Jp[9], Jx[18], r0, r1;
    const double* pose = d.kf_pose + 7 * E.kf;
    proj_error<D>(pose, d.pt_xyz + 3 * E.pt, d.kf_K + 5 * E.kf, E.obs, e, p);
    proj_jacobians<D>(pose, p, d.kf_K + 5 * E.kf, Jp, Jx);
    double c = 0;
    for (int q = 0; q < D; ++q) c += e[q] * e[q];
    c *= E.info;
    huber(c, delta, r0, r1);
    const double w = r1 * E.info;
    int t = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            double s = 0;
            for (int q = 0; q < D; ++q) s += Jp[3 * q + i] * Jp[3 * q + j];
            out9[t++] = w * s;
        }
    for (int i = 0; i < 3; ++i) {
        double sb = 0;
        for (int q = 0; q < D; ++q) sb += Jp[3 * q + i] * (-E.info * e[q]) * r1;
        out9[6 + i] = sb;
    }
}

// The 6x3 block  w J_x^T J_p  of one edge (block_solver.hpp's Hpl) at the current estimates, zero for a fixed pose.
// It is NOT stored: the Schur complement and the back-substitution recompute it from the edge (56 B) instead of reading
// 144 B per edge per use -- the same code and inputs everywhere, so every consumer sees the same bits.
template <int D>
__device__ inline void edge_hpl_t(const Dev& d, const Edge& E, double delta, double* B /*18*/) {
    double e[3], p[3], Jp[9], Jx[18], r0, r1;
    const double* pose = d.kf_pose + 7 * E.kf;
    proj_error<D>(pose, d.pt_xyz + 3 * E.pt, d.kf_K + 5 * E.kf, E.obs, e, p);
    proj_jacobians<D>(pose, p, d.kf_K + 5 * E.kf, Jp, Jx);
    double c = 0;
    for (int q = 0; q < D; ++q) c += e[q] * e[q];
    c *= E.info;
    huber(c, delta, r0, r1);
    const double w = r1 * E.info;
    const bool free_pose = d.kf_h[E.kf] >= 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int q = 0; q < D; ++q) s += Jx[6 * q + i] * Jp[3 * q + j];
            B[3 * i + j] = free_pose ? w * s : 0.0;
        }
}
__device__ inline void edge_hpl(const Dev& d, const Edge& E, const Par& par, double* B) {
    if (E.stereo) edge_hpl_t<3>(d, E, par.delta_stereo, B);
    else edge_hpl_t<2>(d, E, par.delta_mono, B);
}

__device__ __forceinline__ void lin_points_block(const Dev& d, const Par& par, const int bid) {
    __shared__ double sh[256][9 + 1];
    const int p0 = d.chunk_pt[bid], p1 = d.chunk_pt[bid + 1];
    const int e0 = d.pt_off[p0], e1 = d.pt_off[p1];
    if (e1 - e0 > 256) {
        // a chunk of ONE landmark with more than 256 observations (qsp_ba_create gives such a landmark a chunk of its own):
        // windows of 256 edges, nine threads carry the nine running sums over the windows in edge order
        double run = 0;
        for (int w0 = e0; w0 < e1; w0 += 256) {
            const int ew = w0 + threadIdx.x;
            double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (ew < e1 && !d.edge_level[ew]) {
                const Edge E = d.edge[ew];
                if (E.stereo) lin_point_edge<3>(d, E, ew, par.delta_stereo, v);
                else lin_point_edge<2>(d, E, ew, par.delta_mono, v);
            }
            __syncthreads();                 // (the previous window's sums have been read)
            for (int i = 0; i < 9; ++i) sh[threadIdx.x][i] = v[i];
            __syncthreads();
            if (threadIdx.x < 9) {
                const int nq = min(256, e1 - w0);
                for (int q = 0; q < nq; ++q) run += sh[q][threadIdx.x];
            }
        }
        __syncthreads();
        if (threadIdx.x < 9) sh[0][threadIdx.x] = run;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double* a = sh[0];
            double* H = d.Hll + 9 * (size_t)p0;
            H[0] = a[0]; H[1] = a[1]; H[2] = a[2];
            H[3] = a[1]; H[4] = a[3]; H[5] = a[4];
            H[6] = a[2]; H[7] = a[4]; H[8] = a[5];
            double* bl = d.bl + 3 * (size_t)p0;
            bl[0] = a[6]; bl[1] = a[7]; bl[2] = a[8];
        }
        return;
    }
    const int ei = e0 + threadIdx.x;
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ei < e1 && !d.edge_level[ei]) {
        const Edge E = d.edge[ei];
        if (E.stereo) lin_point_edge<3>(d, E, ei, par.delta_stereo, v);
        else lin_point_edge<2>(d, E, ei, par.delta_mono, v);
    }
    for (int i = 0; i < 9; ++i) sh[threadIdx.x][i] = v[i];
    __syncthreads();
    const int pt = p0 + threadIdx.x;
    if (pt < p1) {
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = d.pt_off[pt] - e0; q < d.pt_off[pt + 1] - e0; ++q)
            for (int i = 0; i < 9; ++i) a[i] += sh[q][i];
        double* H = d.Hll + 9 * (size_t)pt;
        H[0] = a[0]; H[1] = a[1]; H[2] = a[2];
        H[3] = a[1]; H[4] = a[3]; H[5] = a[4];
        H[6] = a[2]; H[7] = a[4]; H[8] = a[5];
        double* bl = d.bl + 3 * (size_t)pt;
        bl[0] = a[6]; bl[1] = a[7]; bl[2] = a[8];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// k_lin_poses: one 256-thread workgroup per (key-frame, split of <= KSPLIT edges); upper triangle (21) + rhs (6)
// reduced in a fixed tree into kpart[split]; k_lin_poses_finish adds a key-frame's splits (and its camera-object
// edges) in order.
// ---------------------------------------------------------------------------------------------------------------
constexpr int KSPLIT = 2048;         // graphs that stream (>= KSPLIT_SMALL_BELOW edges): 4.4 TB/s on 2 M edges, 2.7 with 512
constexpr int KSPLIT_SMALL = 512;    // BASELINE-size graphs are latency-bound: shorter per-thread edge loops (C4 BA 3.92 -> 3.80 ms)
constexpr int KSPLIT_SMALL_BELOW = 1 << 18;

template <int D>
__device__ inline void lin_pose_edge(const Dev& d, const Edge& E, double delta, double* A /*21*/, double* b /*6*/) {
    double e[3], p[3], Jp[9], Jx[18], r0, r1;
    const double* pose = d.kf_pose + 7 * E.kf;
    proj_error<D>(pose, d.pt_xyz + 3 * E.pt, d.kf_K + 5 * E.kf, E.obs, e, p);
    proj_jacobians<D>(pose, p, d.kf_K + 5 * E.kf, Jp, Jx);
    double c = 0;
    for (int q = 0; q < D; ++q) c += e[q] * e[q];
    c *= E.info;
    huber(c, delta, r0, r1);
    const double w = r1 * E.info;
    int t = 0;
    for (int i = 0; i < 6; ++i) {
        double sb = 0;
        for (int q = 0; q < D; ++q) sb += Jx[6 * q + i] * (-E.info * e[q]) * r1;
        b[i] += sb;
        for (int j = i; j < 6; ++j) {
            double s = 0;
            for (int q = 0; q < D; ++q) s += Jx[6 * q + i] * Jx[6 * q + j];
            A[t++] += w * s;
        }
    }
}

__device__ __forceinline__ void lin_poses_block(const Dev& d, const Par& par, const int bid) {
    const int sp = bid;
    const int kf = d.ksp_kf[sp];
    if (d.kf_h[kf] < 0) return;
    __shared__ double sh[4][27];
    double A[27];
    for (int i = 0; i < 27; ++i) A[i] = 0;
    for (int q = d.ksp_begin[sp] + threadIdx.x; q < d.ksp_end[sp]; q += 256) {
        const int ei = d.kf_edge[q];
        if (d.edge_level[ei]) continue;
        const Edge E = d.edge[ei];
        if (E.stereo) lin_pose_edge<3>(d, E, par.delta_stereo, A, A + 21);
        else lin_pose_edge<2>(d, E, par.delta_mono, A, A + 21);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LaneTranspose<27, 0>::run(A, lane);                   // lane i < 27 holds the wave's sum of entry i (VALU only, fixed tree)
    if (lane < 27) sh[wave][lane] = A[0];
    __syncthreads();
    if (threadIdx.x < 27)
        d.kpart[27 * (size_t)sp + threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// camera-object edges, one wave per edge: lane 0 evaluates error and Jacobians (ObjectPoseGraph.h:57-89) into LDS, lanes
// 0..35 form entry (i,j) of w Ji^T Ji, w Jj^T Jj and w Ji^T Jj, lanes 36..47 the two right-hand sides.
// Record layout (OE_REC doubles per edge): [0,36) key-frame block, [36,72) object block, [72,78) b key-frame, [78,84) b object.
constexpr int OE_REC = 84;
__device__ __forceinline__ void lin_objedges_block(const Dev& d, const Par& par, const int bid) {
    __shared__ double J[4][80];     // Ji 36 | Jj 36 | -info e r1 (6) | w
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = bid * 4 + wave;
    const bool live = k < d.n_oe && !d.oe_level[k];
    if (live && lane == 0) {
        const int kf = d.oe_kf[k], ob = d.oe_obj[k];
        double e[6], Zi[7], Ji[36], Jj[36], r0, r1;
        obj_error(d.kf_pose + 7 * kf, d.obj_pose + 7 * ob, d.oe_meas + 7 * k, e, Zi);
        obj_jacobians(e, Zi, Ji, Jj);
        double c = 0;
        for (int i = 0; i < 6; ++i) c += e[i] * e[i];
        c *= d.oe_info;
        huber(c, par.delta_obj, r0, r1);
        for (int i = 0; i < 36; ++i) { J[wave][i] = Ji[i]; J[wave][36 + i] = Jj[i]; }
        for (int i = 0; i < 6; ++i) J[wave][72 + i] = (-d.oe_info * e[i]);
        J[wave][78] = r1 * d.oe_info;
        J[wave][79] = r1;
    }
    __syncthreads();
    if (!live) return;
    const double* Ji = J[wave];
    const double* Jj = J[wave] + 36;
    const double w = J[wave][78], r1 = J[wave][79];
    double* rec = d.oe_rec + (size_t)OE_REC * k;
    if (lane < 36) {
        const int i = lane / 6, j = lane % 6;
        double sii = 0, sjj = 0, sij = 0;
        for (int q = 0; q < 6; ++q) {
            sii += Ji[6 * q + i] * Ji[6 * q + j];
            sjj += Jj[6 * q + i] * Jj[6 * q + j];
            sij += Ji[6 * q + i] * Jj[6 * q + j];
        }
        rec[lane] = w * sii;
        rec[36 + lane] = w * sjj;
        d.Hoff[36 * (size_t)k + lane] = (d.kf_h[d.oe_kf[k]] >= 0) ? w * sij : 0.0;   // Ji^T W Jj at (kf row, obj col)
    } else if (lane < 48) {
        const int i = (lane - 36) % 6;
        const double* Jx = (lane < 42) ? Ji : Jj;
        double sb = 0;
        for (int q = 0; q < 6; ++q) sb += Jx[6 * q + i] * J[wave][72 + q] * r1;
        rec[72 + (lane - 36)] = sb;
    }
}

// one wave per key-frame: ordered sum of its splits, then of its camera-object edge records (vertex 0 side)
__device__ __forceinline__ void lin_poses_finish_block(const Dev& d, const Par& par, const int bid) {
    const int kf = bid;
    const int h = d.kf_h[kf];
    if (h < 0) return;
    const int t = threadIdx.x;
    if (t >= 42) return;
    int src, roff;          // index into a split's 27-vector / into an edge record
    if (t < 36) {
        const int i = t / 6, j = t % 6, lo = i < j ? i : j, hi = i < j ? j : i;
        src = lo * 6 - lo * (lo - 1) / 2 + (hi - lo);      // upper-triangle packing of k_lin_poses
        roff = t;
    } else {
        src = 21 + (t - 36);
        roff = 72 + (t - 36);
    }
    double a = 0;
    for (int sp = d.ksp_first[kf]; sp < d.ksp_first[kf + 1]; ++sp) a += d.kpart[27 * (size_t)sp + src];
    if (par.is_root)       // (sharded runs: the camera-object edges are linearised by every rank, counted by the root)
        for (int q = d.kfo_off[kf]; q < d.kfo_off[kf + 1]; ++q) {
            const int k = d.kfo_edge[q];
            if (!d.oe_level[k]) a += d.oe_rec[(size_t)OE_REC * k + roff];
        }
    if (t < 36) d.Hdiag[36 * (size_t)h + t] = a;
    else d.bp[6 * h + (t - 36)] = a;
}

// one wave per object: ordered sum of its edge records (vertex 1 side)
__device__ __forceinline__ void lin_objects_block(const Dev& d, const Par& par, const int bid) {
    const int ob = bid;
    const int hj = d.obj_h[ob];
    if (hj < 0) return;
    const int t = threadIdx.x;
    if (t >= 42) return;
    const int roff = (t < 36) ? 36 + t : 78 + (t - 36);
    double a = 0;
    if (par.is_root)
        for (int q = d.obo_off[ob]; q < d.obo_off[ob + 1]; ++q) {
            const int k = d.obo_edge[q];
            if (!d.oe_level[k]) a += d.oe_rec[(size_t)OE_REC * k + roff];
        }
    if (t < 36) d.Hdiag[36 * (size_t)hj + t] = a;
    else d.bp[6 * hj + (t - 36)] = a;
}

// buildSystem in TWO launches: the three edge passes are independent of each other (block ranges of one grid, the long
// key-frame splits first), and so are the two per-vertex sums that consume them.
__global__ __launch_bounds__(256) void k_lin_edges(Dev d, Par par, int nb_poses, int nb_points) {
    const int b = blockIdx.x;
    if (b < nb_poses) lin_poses_block(d, par, b);
    else if (b < nb_poses + nb_points) lin_points_block(d, par, b - nb_poses);
    else lin_objedges_block(d, par, b - nb_poses - nb_points);
}
__global__ __launch_bounds__(64) void k_lin_vertices(Dev d, Par par) {
    // (scal[2] collects k_maxdiag's atomicMax in the first iteration of a call: zeroed here, one launch earlier in the same stream,
    //  instead of by a memset node -- 13 us of idle device each, profiles/r03_ba_c4_timeline.txt)
    if (blockIdx.x == 0 && threadIdx.x == 0) d.scal[2] = 0.0;
    if ((int)blockIdx.x < d.n_kf) lin_poses_finish_block(d, par, blockIdx.x);
    else lin_objects_block(d, par, blockIdx.x - d.n_kf);
}

// max |diagonal| over all active vertices (computeLambdaInit, optimization_algorithm_levenberg.cpp:166-180)
__global__ __launch_bounds__(256) void k_maxdiag(Dev d, Par par) {
    // scal[2] is zeroed before the launch; |x| >= 0, so the bit patterns of the doubles order like unsigned integers
    __shared__ double sh[4];
    double m = 0;
    const int tid = blockIdx.x * 256 + threadIdx.x, nt = gridDim.x * 256;
    for (int i = tid; i < par.n_pose * 6; i += nt) m = fmax(m, fabs(d.Hdiag[36 * (size_t)(i / 6) + 7 * (i % 6)]));
    for (int i = tid; i < d.n_pt * 3; i += nt)
        if (d.pt_h[i / 3] >= 0) m = fmax(m, fabs(d.Hll[9 * (size_t)(i / 3) + 4 * (i % 3)]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
        atomicMax(reinterpret_cast<unsigned long long*>(&d.scal[2]), (unsigned long long)__double_as_longlong(m));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Schur complement
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_schur_prepare(Dev d, Par par) {
    // Hs (dimp x dimp, zeroed by a memset before) <- diagonal blocks + lambda, off-diagonal blocks; identity on the padding
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nd = par.n_dense * 36;
    if (i < nd) {
        if (par.is_root) {
            const int h = i / 36, r = (i % 36) / 6, c = i % 6;
            double v = d.Hdiag[i];
            if (r == c) v += par.lambda;
            d.Hs[(size_t)(6 * h + r) * par.dimp + 6 * h + c] = v;
        }
    } else if (i < nd + d.n_oe * 36) {
        const int k = (i - nd) / 36, r = ((i - nd) % 36) / 6, c = (i - nd) % 6;
        if (!d.oe_level[k] && !par.elim && par.is_root) {
            const int hi = d.kf_h[d.oe_kf[k]], hj = d.obj_h[d.oe_obj[k]];
            // only the upper block triangle of Hs is read: an object that precedes the key-frame in the hessian order
            // (possible with a caller's own vertex ids, never with the reference's) gets the transposed block
            if (hi >= 0 && hj >= 0) {
                const double v = d.Hoff[36 * (size_t)k + 6 * r + c];
                if (hi < hj) d.Hs[(size_t)(6 * hi + r) * par.dimp + 6 * hj + c] = v;
                else d.Hs[(size_t)(6 * hj + c) * par.dimp + 6 * hi + r] = v;
            }
        }
    } else if (i < nd + d.n_oe * 36 + (par.dimp - par.dim)) {
        const int q = par.dim + (i - nd - d.n_oe * 36);
        if (par.is_root) d.Hs[(size_t)q * par.dimp + q] = 1.0;
    }
    if (i < par.dimp) d.bs[i] = (i < par.dim && par.is_root) ? d.bp[i] : 0.0;
}

__device__ inline bool inv3(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], dd = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (dd * i - f * g) + c * (dd * h - e * g);
    const double id = 1.0 / det;
    o[0] = (e * i - f * h) * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = (f * g - dd * i) * id; o[4] = (a * i - c * g) * id; o[5] = (c * dd - a * f) * id;
    o[6] = (dd * h - e * g) * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * dd) * id;
    return true;
}

// one 64-lane wave per landmark   (block_solver.hpp:381-432).  D^-1 by lane 0; the landmark's hessian indices and
// 6x3 blocks are staged in LDS (one coalesced pass), then the k*k*36 (pair, entry) items are spread over the lanes so
// that no lane waits on a dependent global load; landmarks with more than SCH_K observations read from global memory.
constexpr int SCH_K = 48;
__global__ __launch_bounds__(256) void k_schur_points(Dev d, Par par) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wave;
    const bool live = pt < d.n_pt && d.pt_h[pt] >= 0;
    __shared__ double Dsh[4][12];
    __shared__ double Bsh[4][SCH_K * 18];
    __shared__ int hsh[4][SCH_K];
    const int e0 = live ? d.pt_off[pt] : 0, e1 = live ? d.pt_off[pt + 1] : 0;
    const int k = e1 - e0;
    const bool staged = k <= SCH_K;
    if (live) {
        if (lane == 0) {
            double Dm[9], Di[9];
            for (int i = 0; i < 9; ++i) Dm[i] = d.Hll[9 * (size_t)pt + i];
            Dm[0] += par.lambda; Dm[4] += par.lambda; Dm[8] += par.lambda;
            inv3(Dm, Di);
            const double* bl = d.bl + 3 * (size_t)pt;
            for (int i = 0; i < 9; ++i) { d.Dinv[9 * (size_t)pt + i] = Di[i]; Dsh[wave][i] = Di[i]; }
            for (int i = 0; i < 3; ++i) Dsh[wave][9 + i] = Di[3 * i] * bl[0] + Di[3 * i + 1] * bl[1] + Di[3 * i + 2] * bl[2];
        }
        if (staged) {
            for (int a = lane; a < k; a += 64) hsh[wave][a] = d.edge_level[e0 + a] ? -1 : d.kf_h[d.edge[e0 + a].kf];
            if (lane < k) edge_hpl(d, d.edge[e0 + lane], par, Bsh[wave] + 18 * lane);      // level-masked edges are never read
        }
    }
    __syncthreads();
    if (!live) return;
    const double* Di = Dsh[wave];
    const double* db = Dsh[wave] + 9;
    if (staged) {
        for (int q = lane; q < 6 * k; q += 64) {
            const int a = q / 6, r = q % 6, ha = hsh[wave][a];
            if (ha < 0) continue;
            const double* Ba = Bsh[wave] + 18 * a;
            atomicAdd(&d.bs[6 * ha + r], -(Ba[3 * r] * db[0] + Ba[3 * r + 1] * db[1] + Ba[3 * r + 2] * db[2]));
        }
        const int items = k * k * 36;
        for (int w = lane; w < items; w += 64) {
            const int en = w % 36, pr = w / 36, a = pr / k, b = pr % k;
            const int ha = hsh[wave][a], hb = hsh[wave][b];
            if (ha < 0 || hb < ha || (hb == ha && b != a)) continue;
            const int i = en / 6, j = en % 6;
            const double* Ba = Bsh[wave] + 18 * a;
            const double* Bb = Bsh[wave] + 18 * b;
            const double bd0 = Ba[3 * i] * Di[0] + Ba[3 * i + 1] * Di[3] + Ba[3 * i + 2] * Di[6];   // row i of B_a D^-1
            const double bd1 = Ba[3 * i] * Di[1] + Ba[3 * i + 1] * Di[4] + Ba[3 * i + 2] * Di[7];
            const double bd2 = Ba[3 * i] * Di[2] + Ba[3 * i + 1] * Di[5] + Ba[3 * i + 2] * Di[8];
            atomicAdd(&d.Hs[(size_t)(6 * ha + i) * par.dimp + 6 * hb + j],
                      -(bd0 * Bb[3 * j] + bd1 * Bb[3 * j + 1] + bd2 * Bb[3 * j + 2]));
        }
        return;
    }
    const int i = lane / 6, j = lane % 6;      // entry of the 6x6 block (lanes < 36)
    for (int a = e0; a < e1; ++a) {
        if (d.edge_level[a]) continue;
        const int ha = d.kf_h[d.edge[a].kf];
        if (ha < 0) continue;
        double Ba[18];
        edge_hpl(d, d.edge[a], par, Ba);
        if (lane >= 36 && lane < 42) {
            const int r = lane - 36;
            atomicAdd(&d.bs[6 * ha + r], -(Ba[3 * r] * db[0] + Ba[3 * r + 1] * db[1] + Ba[3 * r + 2] * db[2]));
        }
        if (lane < 36) {
            const double bd0 = Ba[3 * i] * Di[0] + Ba[3 * i + 1] * Di[3] + Ba[3 * i + 2] * Di[6];
            const double bd1 = Ba[3 * i] * Di[1] + Ba[3 * i + 1] * Di[4] + Ba[3 * i + 2] * Di[7];
            const double bd2 = Ba[3 * i] * Di[2] + Ba[3 * i + 1] * Di[5] + Ba[3 * i + 2] * Di[8];
            for (int b = e0; b < e1; ++b) {
                if (d.edge_level[b]) continue;
                const int hb = d.kf_h[d.edge[b].kf];
                if (hb < ha || (hb == ha && b != a)) continue;
                double Bb[18];
                edge_hpl(d, d.edge[b], par, Bb);
                atomicAdd(&d.Hs[(size_t)(6 * ha + i) * par.dimp + 6 * hb + j],
                          -(bd0 * Bb[3 * j] + bd1 * Bb[3 * j + 1] + bd2 * Bb[3 * j + 2]));
            }
        }
    }
}

// The same Schur complement by BLOCK ROW (the default): one workgroup per (key-frame, split of <= KSPLIT of its edges)
// keeps the key-frame's 6 x dimp row block and its 6 right-hand-side entries in LDS, accumulates
// -B_a D^-1 B_b^T for every other observation b of the landmark of each of its edges a (h_b >= h_a) with LDS atomics,
// and adds the non-zero part to Hs once at the end.  Global atomics drop from 36 per (a,b) pair to at most 6*dimp per
// workgroup: with few poses and many landmarks (64 key-frames, 2 M edges) the per-landmark kernel above spends 5.4 ms
// fighting over 147 k addresses.  D^-1 and D^-1 b_l come from k_schur_dinv.
constexpr size_t SCHUR_ROW_LDS_MAX = 160 * 1024;
constexpr int SCHUR_ROWS_MIN_EDGES = 65536;
__global__ __launch_bounds__(256) void k_schur_dinv(Dev d, Par par) {
    const int pt = blockIdx.x * 256 + threadIdx.x;
    if (pt >= d.n_pt || d.pt_h[pt] < 0) return;
    double Dm[9], Di[9];
    for (int i = 0; i < 9; ++i) Dm[i] = d.Hll[9 * (size_t)pt + i];
    Dm[0] += par.lambda; Dm[4] += par.lambda; Dm[8] += par.lambda;
    inv3(Dm, Di);
    const double* bl = d.bl + 3 * (size_t)pt;
    for (int i = 0; i < 9; ++i) d.Dinv[9 * (size_t)pt + i] = Di[i];
    for (int i = 0; i < 3; ++i) d.xl[3 * (size_t)pt + i] = Di[3 * i] * bl[0] + Di[3 * i + 1] * bl[1] + Di[3 * i + 2] * bl[2];
}

__global__ __launch_bounds__(256) void k_schur_rows(Dev d, Par par) {
    extern __shared__ __attribute__((aligned(16))) double srow[];     // [6][dimp] row block, then 6 rhs entries
    const int sp = blockIdx.x, t = threadIdx.x;
    const int ha = d.kf_h[d.ksp_kf[sp]];
    if (ha < 0) return;
    const int dimp = par.dimp;
    for (int i = t; i < 6 * dimp + 6; i += 256) srow[i] = 0.0;
    __syncthreads();
    for (int q = d.ksp_begin[sp] + t; q < d.ksp_end[sp]; q += 256) {
        const int a = d.kf_edge[q];
        if (d.edge_level[a]) continue;
        const int pt = d.edge[a].pt;
        if (d.pt_h[pt] < 0) continue;
        double Ba[18], BD[18];
        {
            const double* Di = d.Dinv + 9 * (size_t)pt;
            double Dl[9];
            edge_hpl(d, d.edge[a], par, Ba);
#pragma unroll
            for (int i = 0; i < 9; ++i) Dl[i] = Di[i];
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int c = 0; c < 3; ++c) BD[3 * i + c] = Ba[3 * i] * Dl[c] + Ba[3 * i + 1] * Dl[3 + c] + Ba[3 * i + 2] * Dl[6 + c];
            const double* db = d.xl + 3 * (size_t)pt;
            const double d0 = db[0], d1 = db[1], d2 = db[2];
#pragma unroll
            for (int r = 0; r < 6; ++r) atomicAdd(&srow[6 * dimp + r], -(Ba[3 * r] * d0 + Ba[3 * r + 1] * d1 + Ba[3 * r + 2] * d2));
        }
        const int e0 = d.pt_off[pt], e1 = d.pt_off[pt + 1];
        for (int b = e0; b < e1; ++b) {
            if (d.edge_level[b]) continue;
            const int hb = d.kf_h[d.edge[b].kf];
            if (hb < ha || (hb == ha && b != a)) continue;
            double Bb[18];
            edge_hpl(d, d.edge[b], par, Bb);
            double* dst = srow + 6 * hb;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    atomicAdd(&dst[i * dimp + j], -(BD[3 * i] * Bb[3 * j] + BD[3 * i + 1] * Bb[3 * j + 1] + BD[3 * i + 2] * Bb[3 * j + 2]));
        }
    }
    __syncthreads();
    for (int i = t; i < 6 * dimp; i += 256) {
        const double v = srow[i];
        if (v != 0.0) atomicAdd(&d.Hs[(size_t)(6 * ha + i / dimp) * dimp + i % dimp], v);
    }
    if (t < 6) atomicAdd(&d.bs[6 * ha + t], srow[6 * dimp + t]);
}

// ---------------------------------------------------------------------------------------------------------------
// Deterministic Schur complement (qsp_ba_set_deterministic): no atomics, every sum in a fixed order.
//   k_trial_stage1 materialises the 6x3 blocks once per trial (edge-parallel), with everything else that is independent;
//   k_schur_pairs one wave per pair of key-frames that share landmarks (list built on the host, in landmark order):
//                 lane (i,j) accumulates entry (i,j) of  sum_l B_a D_l^-1 B_b^T  and is the only writer of that entry;
//                 the diagonal pair (ka, ka) also carries the right-hand side sum over the key-frame's edges of B_a D^-1 b_l.
// Costs one sort-free pass over sum_l k_l (k_l+1)/2 pairs on the host at creation and 144 B per edge of extra storage;
// meant for reproducible runs and for the parity tests, not for the largest graphs.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_schur_pairs(Dev d, Par par) {
    // one workgroup per pair of key-frames (ka <= kb by scene index) that share landmarks or objects.  thread = list entry
    // (strided by 256): every thread accumulates the whole 6x6 of its entries, then a fixed butterfly over the lanes and a
    // fixed-order sum over the four waves -- the loads of 256 entries are in flight together and the order of every addition
    // depends on list positions only.  The diagonal pair (ka, ka) lists every edge of the key-frame once, so it also carries
    // the right-hand side  sum_l B_a D^-1 b_l  (and  sum_o F_e b_o  of the object elimination).
    __shared__ double sh[4][42];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // long lists (the diagonal pairs: every edge of a key-frame) get the whole workgroup, short ones a wave each
    const bool big = (int)blockIdx.x < d.n_pk_big;
    const int slot = big ? (int)blockIdx.x : d.n_pk_big + ((int)blockIdx.x - d.n_pk_big) * 4 + wave;
    if (slot >= d.n_pk) return;                           // (only in the last block of short pairs: no barrier there)
    const int q = d.pk_work[slot];
    const int tid = big ? t : lane, stride = big ? 256 : 64;
    const int ka = d.pk_ka[q], kb = d.pk_kb[q];
    const int ha = d.kf_h[ka], hb = d.kf_h[kb];
    if (ha < 0 || hb < 0) return;                         // (uniform per pair: a big pair leaves with all four waves)
    const bool diag = ka == kb;
    double acc[42];
#pragma unroll
    for (int i = 0; i < 42; ++i) acc[i] = 0.0;
    for (int u = d.pk_off[q] + tid; u < d.pk_off[q + 1]; u += stride) {
        const int4 en = d.pk_ent[u];                      // edge in ka, edge in kb, landmark
        if (d.edge_level[en.x] || d.edge_level[en.y] || d.pt_h[en.z] < 0) continue;
        double Ba[18], Bb[18], Di[9];
        const double* pa = d.Hpl + 18 * (size_t)en.x;
        const double* pb = d.Hpl + 18 * (size_t)en.y;
        const double* pd = d.Dinv + 9 * (size_t)en.z;
#pragma unroll
        for (int i = 0; i < 18; ++i) { Ba[i] = pa[i]; Bb[i] = pb[i]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) Di[i] = pd[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double bd0 = Ba[3 * i] * Di[0] + Ba[3 * i + 1] * Di[3] + Ba[3 * i + 2] * Di[6];
            const double bd1 = Ba[3 * i] * Di[1] + Ba[3 * i + 1] * Di[4] + Ba[3 * i + 2] * Di[7];
            const double bd2 = Ba[3 * i] * Di[2] + Ba[3 * i + 1] * Di[5] + Ba[3 * i + 2] * Di[8];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[6 * i + j] += bd0 * Bb[3 * j] + bd1 * Bb[3 * j + 1] + bd2 * Bb[3 * j + 2];
        }
        if (diag) {                                       // (diagonal lists hold (a, a, l) entries only)
            const double* db = d.xl + 3 * (size_t)en.z;
            const double d0 = db[0], d1 = db[1], d2 = db[2];
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[36 + r] += Ba[3 * r] * d0 + Ba[3 * r + 1] * d1 + Ba[3 * r + 2] * d2;
        }
    }
    if (par.elim && par.is_root) {
        for (int u = d.pko_off[q] + tid; u < d.pko_off[q + 1]; u += stride) {
            const int4 en = d.pko_ent[u];                 // object edge at ka, object edge at kb, object
            if (d.oe_level[en.x] || d.oe_level[en.y]) continue;
            const int ho = d.obj_h[en.z];
            if (ho < 0) continue;
            const double* F = d.oe_F + 36 * (size_t)en.x;     // F_e = H_ko (H_oo + lambda I)^-1
            const double* E = d.Hoff + 36 * (size_t)en.y;     // H_k'o
            double Fr[36], Er[36];
#pragma unroll
            for (int i = 0; i < 36; ++i) { Fr[i] = F[i]; Er[i] = E[i]; }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    double v = 0;
#pragma unroll
                    for (int m = 0; m < 6; ++m) v += Fr[6 * i + m] * Er[6 * j + m];
                    acc[6 * i + j] += v;
                }
            if (diag && en.x == en.y) {
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    double v = 0;
#pragma unroll
                    for (int m = 0; m < 6; ++m) v += Fr[6 * r + m] * d.bp[6 * ho + m];
                    acc[36 + r] += v;
                }
            }
        }
    }
    LaneTranspose<42, 0>::run(acc, lane);                 // lane L < 42 now holds the wave's sum of entry L (fixed tree, VALU only)
    if (big) {
        if (lane < 42) sh[wave][lane] = acc[0];
        __syncthreads();
        if (wave != 0) return;
    }
    // T = sum B_a D^-1 B_b^T belongs at (ha, hb); only the upper block triangle of Hs is used: transpose if ha > hb
    if (lane < 42) {
        double v = acc[0];
        if (big) v = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        if (lane < 36) {
            const int i = lane / 6, j = lane % 6;
            if (ha <= hb) d.Hs[(size_t)(6 * ha + i) * par.dimp + 6 * hb + j] -= v;
            else d.Hs[(size_t)(6 * hb + j) * par.dimp + 6 * ha + i] -= v;
        } else if (diag) {
            d.bs[6 * ha + (lane - 36)] -= v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Object elimination: a second Schur complement in front of the dense solve.  In g2o's graph an object vertex is a
// non-marginalised 6-DoF pose that is connected to key-frames only (EdgeSE3LieAlgebra, src/Optimizer_util.cc:548-584), never
// to another object or to a landmark: the object part of the reduced system is BLOCK DIAGONAL.  Eliminating it in closed
// form -- exactly what the fill-reducing ordering of the reference's sparse LDLT does implicitly
// (linear_solver_eigen.h:147-201) -- leaves a dense system over the free key-frames only (C4: 704 -> 320 unknowns, C5:
// 2752 -> 1216 after padding; the factorisation's serial pivot chain shrinks by the same factor):
//     G_o = (H_oo + lambda I)^-1,   F_e = H_ko G_o  for every edge e = (k, o)
//     Hs_kk' -= F_e H_k'o^T  (e = (k,o), e' = (k',o)),   bs_k -= F_e b_o,   x_o = G_o b_o - sum_e F_e^T x_k.
// Same solution as the joint factorisation to rounding (~1e-12 relative).  Used when every free key-frame precedes every
// object in the hessian order (always so with the reference's vertex ids); otherwise the objects stay in the dense system.
//   k_obj_prepare  one wave per object: lane 0 inverts the damped 6x6 block (Cholesky), lanes 0..35 form F_e per edge;
//   k_obj_rows     one wave per free key-frame: its row block of the update is accumulated in LDS in a fixed order (the
//                  key-frame's edges in list order, per object its edges in list order) and subtracted from Hs once:
//                  single writer per entry, no atomics, bit-reproducible;
//   the back-substitution of the objects is the tail of k_chol_back.
// In the atomic-free mode k_obj_prepare is a block range of k_trial_stage1 and k_obj_rows is replaced by object entries in the
// key-frame pair lists (k_schur_pairs).
// ---------------------------------------------------------------------------------------------------------------
__device__ inline bool inv6_spd(const double* A, double* G) {
    // Cholesky A = L L^T, then G = L^-T L^-1; A symmetric (only its upper triangle is read)
    double L[36], W[36];
    bool ok = true;
    for (int i = 0; i < 36; ++i) { L[i] = 0; W[i] = 0; }
    for (int j = 0; j < 6; ++j) {
        double dd = A[6 * j + j];
        for (int q = 0; q < j; ++q) dd -= L[6 * j + q] * L[6 * j + q];
        if (!(dd > 0) || !isfinite(dd)) { ok = false; dd = 1.0; }
        const double r = 1.0 / sqrt(dd);
        L[6 * j + j] = dd * r;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[6 * j + i];
            for (int q = 0; q < j; ++q) v -= L[6 * i + q] * L[6 * j + q];
            L[6 * i + j] = v * r;
        }
    }
    for (int c = 0; c < 6; ++c)            // W = L^-1 (lower), column by column
        for (int i = c; i < 6; ++i) {
            double v = (i == c) ? 1.0 : 0.0;
            for (int q = c; q < i; ++q) v -= L[6 * i + q] * W[6 * q + c];
            W[6 * i + c] = v / L[6 * i + i];
        }
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double v = 0;
            for (int q = (i > j ? i : j); q < 6; ++q) v += W[6 * q + i] * W[6 * q + j];
            G[6 * i + j] = v;
        }
    return ok;
}

__global__ __launch_bounds__(64) void k_obj_prepare(Dev d, Par par) {
    const int ob = blockIdx.x, lane = threadIdx.x;
    const int ho = d.obj_h[ob];
    if (ho < 0) return;
    __shared__ double G[42];
    double* out = d.obj_G + 42 * (size_t)ob;
    if (lane == 0) {
        double A[36];
        for (int i = 0; i < 36; ++i) A[i] = d.Hdiag[36 * (size_t)ho + i];
        for (int i = 0; i < 6; ++i) A[7 * i] += par.lambda;
        if (!inv6_spd(A, G)) d.scal[3] = 1.0;
        for (int i = 0; i < 6; ++i) {
            double v = 0;
            for (int j = 0; j < 6; ++j) v += G[6 * i + j] * d.bp[6 * ho + j];
            G[36 + i] = v;
        }
    }
    __syncthreads();
    if (lane < 42) out[lane] = G[lane];
    if (lane < 36) {
        const int r = lane / 6, c = lane % 6;
        for (int q = d.obo_off[ob]; q < d.obo_off[ob + 1]; ++q) {
            const int k = d.obo_edge[q];
            if (d.oe_level[k]) continue;
            const double* E = d.Hoff + 36 * (size_t)k;          // H_ko, zero when the key-frame is fixed
            double v = 0;
#pragma unroll
            for (int m = 0; m < 6; ++m) v += E[6 * r + m] * G[6 * m + c];
            d.oe_F[36 * (size_t)k + lane] = v;
        }
    }
}

__global__ __launch_bounds__(64) void k_obj_rows(Dev d, Par par) {
    extern __shared__ __attribute__((aligned(16))) double orow[];     // [6][dimp] row block
    const int kf = blockIdx.x, lane = threadIdx.x;
    const int ha = d.kf_h[kf];
    if (ha < 0 || !par.is_root) return;
    const int dimp = par.dimp;
    int n_live = 0;
    for (int q = d.kfo_off[kf]; q < d.kfo_off[kf + 1]; ++q) n_live += d.oe_level[d.kfo_edge[q]] ? 0 : 1;
    if (n_live == 0) return;
    for (int i = lane; i < 6 * dimp; i += 64) orow[i] = 0.0;
    __syncthreads();
    const int r = lane / 6, c = lane % 6;
    double rhs = 0;
    for (int q = d.kfo_off[kf]; q < d.kfo_off[kf + 1]; ++q) {
        const int e = d.kfo_edge[q];
        if (d.oe_level[e]) continue;
        const int ob = d.oe_obj[e];
        const int ho = d.obj_h[ob];
        if (ho < 0) continue;
        double Fr[6];                                             // row r of F_e
#pragma unroll
        for (int m = 0; m < 6; ++m) Fr[m] = (lane < 36) ? d.oe_F[36 * (size_t)e + 6 * r + m] : 0.0;
        if (lane < 36 && c == 0) {
#pragma unroll
            for (int m = 0; m < 6; ++m) rhs += Fr[m] * d.bp[6 * ho + m];
        }
        for (int q2 = d.obo_off[ob]; q2 < d.obo_off[ob + 1]; ++q2) {
            const int e2 = d.obo_edge[q2];
            if (d.oe_level[e2]) continue;
            const int hb = d.kf_h[d.oe_kf[e2]];
            if (hb < ha) continue;                                // upper block triangle only (fixed key-frames: hb = -1)
            if (lane < 36) {
                const double* E2 = d.Hoff + 36 * (size_t)e2;      // H_k'o: entry (r,c) of F_e H_k'o^T = sum_m F[r][m] E2[c][m]
                double v = 0;
#pragma unroll
                for (int m = 0; m < 6; ++m) v += Fr[m] * E2[6 * c + m];
                orow[r * dimp + 6 * hb + c] += v;                 // one lane per entry, program order: reproducible
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < 6 * dimp; i += 64) {
        const double v = orow[i];
        if (v != 0.0) d.Hs[(size_t)(6 * ha + i / dimp) * dimp + i % dimp] -= v;
    }
    if (lane < 36 && c == 0) d.bs[6 * ha + r] -= rhs;
}

// ---------------------------------------------------------------------------------------------------------------
// Everything a Levenberg-Marquardt trial needs before the pair kernel, in ONE launch (the pieces are independent of each
// other; at the BASELINE sizes each of them is a few microseconds of work behind a launch):
//   blocks [0, nb_hs)            Hs <- 0, diagonal blocks + lambda, identity on the padding; bs <- b_p   (was: memset + prepare)
//   blocks [.., + nb_pt)         D^-1 and D^-1 b_l per landmark                                          (k_schur_dinv)
//   blocks [.., + nb_edge)       the 6x3 blocks of this linearisation                                    (k_hpl_fill)
//   blocks [.., + nb_obj)        (H_oo + lambda I)^-1 and F_e per object                                  (k_obj_prepare)
//   blocks [.., + nb_bk)         backup of the estimates (g2o's push(), sparse_optimizer.cpp:519-527)     (was: 3 copies)
// ---------------------------------------------------------------------------------------------------------------
struct Stage1 { int nb_hs, nb_pt, nb_edge, nb_obj, nb_bk; };

__global__ __launch_bounds__(256) void k_trial_stage1(Dev d, Par par, Stage1 g) {
    __shared__ double G[4][42];
    int b = blockIdx.x;
    const int t = threadIdx.x;
    if (b < g.nb_hs) {
        const size_t n = (size_t)par.dimp * par.dimp;
        const size_t idx = (size_t)b * 256 + t;
        if (idx < n) {
            const int row = (int)(idx / par.dimp), col = (int)(idx % par.dimp);
            double v = 0.0;
            if (par.is_root) {
                if (row < par.dim && col < par.dim) {
                    if (row / 6 == col / 6) {
                        v = d.Hdiag[36 * (size_t)(row / 6) + 6 * (row % 6) + col % 6];
                        if (row == col) v += par.lambda;
                    }
                } else if (row == col) {
                    v = 1.0;
                }
            }
            d.Hs[idx] = v;
        } else if (idx < n + par.dimp) {
            const int i = (int)(idx - n);
            d.bs[i] = (i < par.dim && par.is_root) ? d.bp[i] : 0.0;
        }
        return;
    }
    b -= g.nb_hs;
    if (b < g.nb_pt) {
        const int pt = b * 256 + t;
        if (pt >= d.n_pt || d.pt_h[pt] < 0) return;
        double Dm[9], Di[9];
        for (int i = 0; i < 9; ++i) Dm[i] = d.Hll[9 * (size_t)pt + i];
        Dm[0] += par.lambda; Dm[4] += par.lambda; Dm[8] += par.lambda;
        inv3(Dm, Di);
        const double* bl = d.bl + 3 * (size_t)pt;
        for (int i = 0; i < 9; ++i) d.Dinv[9 * (size_t)pt + i] = Di[i];
        for (int i = 0; i < 3; ++i) d.xl[3 * (size_t)pt + i] = Di[3 * i] * bl[0] + Di[3 * i + 1] * bl[1] + Di[3 * i + 2] * bl[2];
        return;
    }
    b -= g.nb_pt;
    if (b < g.nb_edge) {
        const int e = b * 256 + t;
        if (e >= d.n_edge) return;
        double B[18];
        if (d.edge_level[e]) {
            for (int i = 0; i < 18; ++i) B[i] = 0.0;
        } else {
            edge_hpl(d, d.edge[e], par, B);
        }
        for (int i = 0; i < 18; ++i) d.Hpl[18 * (size_t)e + i] = B[i];
        return;
    }
    b -= g.nb_edge;
    if (b < g.nb_obj) {                                   // one wave per object (block-uniform branch: barriers are safe)
        const int wave = t >> 6, lane = t & 63;
        const int ob = b * 4 + wave;
        const int ho = (ob < d.n_obj) ? d.obj_h[ob] : -1;
        if (ho >= 0 && lane == 0) {
            double A[36];
            for (int i = 0; i < 36; ++i) A[i] = d.Hdiag[36 * (size_t)ho + i];
            for (int i = 0; i < 6; ++i) A[7 * i] += par.lambda;
            if (!inv6_spd(A, G[wave])) d.scal[3] = 1.0;
            for (int i = 0; i < 6; ++i) {
                double v = 0;
                for (int j = 0; j < 6; ++j) v += G[wave][6 * i + j] * d.bp[6 * ho + j];
                G[wave][36 + i] = v;
            }
        }
        __syncthreads();
        if (ho < 0) return;
        if (lane < 42) d.obj_G[42 * (size_t)ob + lane] = G[wave][lane];
        if (lane < 36) {
            const int r = lane / 6, c = lane % 6;
            for (int q = d.obo_off[ob]; q < d.obo_off[ob + 1]; ++q) {
                const int k = d.obo_edge[q];
                if (d.oe_level[k]) continue;
                const double* E = d.Hoff + 36 * (size_t)k;
                double v = 0;
#pragma unroll
                for (int m = 0; m < 6; ++m) v += E[6 * r + m] * G[wave][6 * m + c];
                d.oe_F[36 * (size_t)k + lane] = v;
            }
        }
        return;
    }
    b -= g.nb_obj;
    {
        const int i = b * 256 + t;
        const int nk = 7 * d.n_kf, no = 7 * d.n_obj, np = 3 * d.n_pt;
        if (i < nk) d.kf_bk[i] = d.kf_pose[i];
        else if (i < nk + no) d.obj_bk[i - nk] = d.obj_pose[i - nk];
        else if (i < nk + no + np) d.pt_bk[i - nk - no] = d.pt_xyz[i - nk - no];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Dense SPD solve of the reduced system  Hs x = bs  (dimp multiple of NB = 64), replacing g2o's LDLT
// (linear_solver_eigen.h:94-124).  Right-looking blocked Cholesky Hs = L L^T with ONE launch per block step:
//   * workgroup (i,j), k < i <= j, forms the two panel blocks it needs itself, P_i = L_kk^-1 A_ki and P_j = L_kk^-1 A_kj,
//     as products with the explicit inverse W_k = L_kk^-1 (no triangular solves anywhere), and updates A_ij -= P_i^T P_j;
//   * the first row of workgroups (i = k+1) also stores P_j as the factor's block row k and carries the right-hand side
//     along (forward substitution comes for free: y_k = W_k b_k, b_j -= P_j^T y_k);
//   * workgroup (k+1,k+1) goes on to factorise its freshly updated diagonal block in registers -- Gauss steps applied
//     to [S | I] give L^-1 beside the factor -- so the next launch finds W_{k+1} and y_{k+1} ready.
// The backward substitution x_k = W_k^T (y_k - sum_j U_kj x_j) is nb small launches of matrix-vector products.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CHOL_THREADS = 320;           // four waves of 4x4 tiles + the panel wave of factor_tile64
constexpr int CHOL_ROWBUF = 4 * 8 * NB;     // factor_tile64: published rows (rb) and rows handed to the panel wave (nx), double-buffered
constexpr int CHOL_LDS_DOUBLES = 3 * NB * NB + CHOL_ROWBUF + NB;
static_assert(NB * (NB + 1) <= 2 * NB * NB, "Wr (NB x 65) lies over X and the head of Yi");

// The 64x64x64 products of the factorisation on the FP64 matrix pipe: wave w of four forms rows 16w..16w+15 of the result as
// four 16x16 tiles, acc[tc] (+)= sum_m A(m, 16w + i) B[m][16 tc + j] with A(m, r) = X[m sk + r si] (so the left operand may be
// stored either way round, with any pitch) and B row-major with pitch NB.  v_mfma_f64_16x16x4_f64: lane l carries A[i = l & 15]
// [k = l >> 4] and B[k = l >> 4][j = l & 15]; its four results are rows (l >> 4) + 4 reg, column l & 15 of the tile.  The matrix
// pipe's FP64 rate equals the vector pipe's on this part; what it saves is LDS traffic -- one operand double per lane and 1024
// multiply-adds against eight doubles per 16 with 4x4 register tiles, which kept the four waves of a compute unit waiting on the
// LDS for ~60 % of a product (10.5 k cycles per product measured, 4.1 k of arithmetic).
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ inline void mfma_gemm64(const double* X, int sk, int si, const double* B, int wave, int lane, d4_t (&acc)[4]) {
    const double* xa = X + (lane >> 4) * sk + (16 * wave + (lane & 15)) * si;
    const double* yb = B + (lane >> 4) * NB + (lane & 15);
#pragma unroll 4
    for (int ks = 0; ks < NB / 4; ++ks) {
        const double a = xa[4 * ks * sk];
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) acc[tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yb[4 * ks * NB + 16 * tc], acc[tc], 0, 0, 0);
    }
}
// the wave's four result tiles into a row-major 64x64 LDS image
__device__ inline void mfma_store64(double* Z, const d4_t (&acc)[4], int wave, int lane) {
#pragma unroll
    for (int tc = 0; tc < 4; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[(16 * wave + (lane >> 4) + 4 * r) * NB + 16 * tc + (lane & 15)] = acc[tc][r];
}
// the factor's block U_kj = P (64 x 64) goes to memory TRANSPOSED, element (m, c) at Uf[(j NB + c) ld + k NB + m]: the backward
// substitution walks it by rows m with a lane per row.  Straight from the result registers (4 consecutive m per 16 lanes).
__device__ inline void mfma_store_factor(double* Uf, int ld, int j, int k, const d4_t (&acc)[4], int wave, int lane) {
#pragma unroll
    for (int tc = 0; tc < 4; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Uf[(size_t)(j * NB + 16 * tc + (lane & 15)) * ld + k * NB + 16 * wave + (lane >> 4) + 4 * r] = acc[tc][r];
}

// 1/sqrt(d): hardware estimate + two Newton steps (full double precision to an ulp or two; keeps the f64 divide and
// square-root sequences off the factorisation's serial chain)
__device__ inline double rsqrt_nr(double d) {
    double r = __builtin_amdgcn_rsq(d);
    r = r * (1.5 - 0.5 * d * r * r);
    r = r * (1.5 - 0.5 * d * r * r);
    return r;
}

// Factorisation of a 64x64 SPD block by 320 threads: four UPDATER waves hold S (and W = L^-1, grown from the identity by the same
// row operations) as 4x4 register tiles (thread owns rows r0.., cols c0..); a fifth PANEL wave (lane = column) owns the serial
// chain.  Phase p (one workgroup barrier each):
//   panel     takes rows 4p..4p+3 of [S | W] as the updaters left them one block earlier (nx), applies block p-1 itself (8 values
//             per lane), fetches the 4x4 diagonal values by v_readlane, factorises them (Newton rsqrt), finishes its column of
//             the four rows of U and W and publishes them (rb; the W rows also into Wr, pitch 65, for the tail);
//   updaters  apply block p-1 (rank 4) to every tile below it; the 16 threads owning rows 4(p+1).. then hand their tiles to the
//             panel wave (nx) for the next phase.
// The pivot chain and the rank-4 update overlap instead of alternating: 14.1 against 20.6 us per block on MI355X
// (tools/micro/chol_factor.hip, variants 0 and 5), every element going through the same operations in the same order (the two
// forms agree in every bit).  S is consumed.  rowbuf: CHOL_ROWBUF doubles of LDS (rb and nx, double-buffered); Wr: NB x 65.
// From phase FETCH_PHASE on the first two updater waves have no rows left (rows 0..31 are final): k_chol_chain gives them another
// job there -- fetching the next step's tiles (Fetch::phase(p), once per phase) -- in a loop of their own, so that what they keep
// in registers does not weigh on the loops of the waves still at work.
// (the phases exchange data through LDS only: their barrier waits for this wave's LDS traffic, not for global loads the fetching
//  waves have in flight across them)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int FETCH_PHASE = NB / 8;
struct NoFetch {
    static constexpr bool active = false;
    struct Regs {};
    __device__ __forceinline__ void phase(int, Regs&) const {}
    __device__ __forceinline__ void finish(Regs&) const {}
};

struct FactorLds { double *rbuf, *nx, *Wr; };
// what the panel wave carries from phase p-1 into phase p: its own column of the block it has just published (U and W rows) and
// that block's entries in columns 4p..4p+3 -- the latter read back from LDS right behind the stores, in FRONT of the barrier, so
// that behind it only the rows handed over by the updaters remain to be fetched
struct PanelCarry { double s[4], w[4], ur[4][4]; };
// the panel wave's phase p; returns whether a pivot was not positive
__device__ __forceinline__ bool factor_panel_phase(int p, const FactorLds& L, int lane, PanelCarry& pc) {
    bool bad = false;
    double* rb = L.rbuf + (p & 1) * 8 * NB;
    const double* in = L.nx + (p & 1) * 8 * NB;
    double s[4], w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { s[a] = in[a * NB + lane]; w[a] = in[(4 + a) * NB + lane]; }
    if (p > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                s[a] -= pc.ur[q][a] * pc.s[q];
                w[a] -= pc.ur[q][a] * pc.w[q];
            }
        }
    }
    double D[4][4], rs[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b) {      // (a uniform lane index: v_readlane to scalar registers, no LDS round trip)
            union { double d; int i[2]; } u, r;
            u.d = s[a];
            r.i[0] = __builtin_amdgcn_readlane(u.i[0], 4 * p + b);
            r.i[1] = __builtin_amdgcn_readlane(u.i[1], 4 * p + b);
            D[a][b] = r.d;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double dd = D[q][q];
        const bool isbad = !(dd > 0) || !isfinite(dd);
        bad |= isbad;
        if (isbad) dd = 1.0;
        rs[q] = rsqrt_nr(dd);
#pragma unroll
        for (int b = q + 1; b < 4; ++b) D[q][b] *= rs[q];
#pragma unroll
        for (int a = q + 1; a < 4; ++a)
#pragma unroll
            for (int b = a; b < 4; ++b) D[a][b] -= D[q][a] * D[q][b];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double sv = s[q], wv = w[q];
#pragma unroll
        for (int pp = 0; pp < q; ++pp) { sv -= D[pp][q] * s[pp]; wv -= D[pp][q] * w[pp]; }
        s[q] = sv * rs[q];
        w[q] = wv * rs[q];
        rb[q * NB + lane] = s[q];
        rb[(4 + q) * NB + lane] = w[q];
        L.Wr[(4 * p + q) * (NB + 1) + lane] = w[q];
        pc.s[q] = s[q];
        pc.w[q] = w[q];
    }
    if (p + 1 < NB / 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a) pc.ur[q][a] = rb[q * NB + 4 * (p + 1) + a];     // (this wave's own stores: LDS keeps their order)
    }
    return bad;
}
// an updater thread's phase p
__device__ __forceinline__ void factor_updater_phase(int p, const FactorLds& L, double (&S)[4][4], double (&W)[4][4], int r0, int c0) {
    const double* rbp = L.rbuf + ((p + 1) & 1) * 8 * NB;       // block p-1
    if (p > 0 && r0 > 4 * (p - 1)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double ur[4], uc[4], wc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { ur[a] = rbp[q * NB + r0 + a]; uc[a] = rbp[q * NB + c0 + a]; wc[a] = rbp[(4 + q) * NB + c0 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { S[a][b] -= ur[a] * uc[b]; W[a][b] -= ur[a] * wc[b]; }
        }
    }
    if (r0 == 4 * (p + 1)) {
        double* out = L.nx + ((p + 1) & 1) * 8 * NB;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) { out[a * NB + c0 + b] = S[a][b]; out[(4 + a) * NB + c0 + b] = W[a][b]; }
    }
}

template <class Fetch>
__device__ __forceinline__ bool factor_tile64(double (&S)[4][4], double* Wr, double* rowbuf, int r0, int c0, Fetch& fetch) {
    bool bad = false;
    const int t = threadIdx.x, lane = t & 63;
    const bool panel = t >= 256;
    const FactorLds L{rowbuf, rowbuf + 2 * 8 * NB, Wr};
    double W[4][4];
    if (!panel) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) W[a][b] = (r0 + a == c0 + b) ? 1.0 : 0.0;
        if (r0 == 0) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { L.nx[a * NB + c0 + b] = S[a][b]; L.nx[(4 + a) * NB + c0 + b] = W[a][b]; }
        }
    }
    lds_barrier();
    if (panel) {
        __builtin_amdgcn_s_setprio(3);              // (it shares a SIMD with an updater wave and everybody waits for it)
        PanelCarry pc;
#pragma nounroll
        for (int p = 0; p < NB / 4; ++p) {
            bad |= factor_panel_phase(p, L, lane, pc);
            lds_barrier();
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
#pragma nounroll
        for (int p = 0; p < FETCH_PHASE; ++p) {
            factor_updater_phase(p, L, S, W, r0, c0);
            lds_barrier();
        }
        if (Fetch::active && t < 128) {
            typename Fetch::Regs regs;               // (what is in flight lives in this loop only)
#pragma nounroll
            for (int p = FETCH_PHASE; p < NB / 4; ++p) {
                fetch.phase(p, regs);
                lds_barrier();
            }
            fetch.finish(regs);
        } else {
#pragma nounroll
            for (int p = FETCH_PHASE; p < NB / 4; ++p) {
                factor_updater_phase(p, L, S, W, r0, c0);
                lds_barrier();
            }
        }
    }
    return bad;
}

// tail shared by the first launch and workgroup (k+1,k+1), 320 threads: factorise block kb, publish W_kb^T and y_kb = W_kb b_kb.
// bvec (LDS, NB doubles) holds the fully updated right-hand-side block; Wr (LDS, NB x 65) receives the rows of W.
__device__ inline void factor_and_forward(double (&S)[4][4], int kb, double* Winv, double* y, double* Wr, double* rowbuf,
                                          const double* bvec, double* scal, int r0, int c0) {
    NoFetch nofetch;
    const bool bad = factor_tile64(S, Wr, rowbuf, r0, c0, nofetch);
    if (bad) scal[3] = 1.0;                                        // (the panel wave's lanes: one value, 64 writers)
    // (the loop's last barrier has made Wr complete.)  W^T to global: WT[m][q] = W[q][m]
    double* Wg = Winv + (size_t)kb * NB * NB;
    for (int e = threadIdx.x; e < NB * NB; e += CHOL_THREADS) Wg[e] = Wr[(e % NB) * (NB + 1) + e / NB];
    if (threadIdx.x < NB) {
        double v = 0;
        for (int m = 0; m < NB; ++m) v += Wr[threadIdx.x * (NB + 1) + m] * bvec[m];
        y[kb * NB + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(CHOL_THREADS) void k_chol_first(const double* A, double* Winv, const double* b, double* y, int ld,
                                                             double* scal) {
    extern __shared__ __attribute__((aligned(16))) double chol_lds[];
    double* X = chol_lds;
    double* rowbuf = chol_lds + 3 * NB * NB;
    double* bvec = rowbuf + CHOL_ROWBUF;
    const int t = threadIdx.x, r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    double S[4][4];
    if (t < 256) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[a][q] = A[(size_t)(r0 + a) * ld + c0 + q];
    }
    if (t < NB) bvec[t] = b[t];
    __syncthreads();
    factor_and_forward(S, 0, Winv, y, X, rowbuf, bvec, scal, r0, c0);
}

// grid (n, n), n = nb - k - 1: x = j - k - 1, y = i - k - 1; workgroups below the diagonal leave at once.  320 threads: the
// fifth wave is factor_tile64's panel wave and has work in workgroup (k+1,k+1) only (there it also carries the right-hand side
// while the others are in the second product).
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_step(double* A, double* Uf, double* Winv, double* b, double* y, int ld, int k,
                                                            double* scal) {
    const int j = k + 1 + blockIdx.x, i = k + 1 + blockIdx.y;
    if (i > j) return;
    const int t = threadIdx.x, r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    const bool first_row = (i == k + 1);
    const bool chain = first_row && i == j;               // this workgroup goes on to factorise block k+1
    const bool upd = t < 256;                             // (elsewhere the fifth wave only walks through the barriers)
    extern __shared__ __attribute__((aligned(16))) double chol_lds[];
    double* X = chol_lds;
    double* Yi = chol_lds + NB * NB;
    double* Yj = (i == j) ? Yi : chol_lds + 2 * NB * NB;
    double* rowbuf = chol_lds + 3 * NB * NB;
    double* bvec = rowbuf + CHOL_ROWBUF;
    const double* Wg = Winv + (size_t)k * NB * NB;
    double S[4][4];                                       // the tile this thread updates: in flight behind the panel loads
    if (upd) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[a][q] = A[(size_t)(i * NB + r0 + a) * ld + j * NB + c0 + q];
        for (int e = t; e < NB * NB; e += 256) {
            X[e] = Wg[e];
            Yi[e] = A[(size_t)(k * NB + e / NB) * ld + i * NB + e % NB];
            if (i != j) Yj[e] = A[(size_t)(k * NB + e / NB) * ld + j * NB + e % NB];
        }
    }
    __syncthreads();
    const int wave = t >> 6, lane = t & 63;
    d4_t pi[4] = {}, pj[4] = {};
    if (upd) {
        mfma_gemm64(X, NB, 1, Yi, wave, lane, pi);
        if (i != j) mfma_gemm64(X, NB, 1, Yj, wave, lane, pj);
    }
    __syncthreads();
    if (upd) {
        mfma_store64(Yi, pi, wave, lane);
        if (i != j) mfma_store64(Yj, pj, wave, lane);
    }
    __syncthreads();
    if (upd) {
        d4_t acc[4] = {};
        mfma_gemm64(Yi, NB, 1, Yj, wave, lane, acc);
        mfma_store64(X, acc, wave, lane);          // (X is free since the first products: the way back to the 4x4 register tiles)
    }
    __syncthreads();
    if (upd) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[a][q] -= X[(r0 + a) * NB + c0 + q];
    }
    if (first_row) {
        if (upd) {
            if (i == j) mfma_store_factor(Uf, ld, j, k, pi, wave, lane);
            else mfma_store_factor(Uf, ld, j, k, pj, wave, lane);
        }
        // b_j -= P_j^T y_k: by the panel wave where there is one (beside the second product), by the first wave elsewhere
        const int tb = chain ? t - 256 : t;
        if (tb >= 0 && tb < NB) {
            double v = b[j * NB + tb];
            for (int q = 0; q < NB; ++q) v -= Yj[q * NB + tb] * y[k * NB + q];
            b[j * NB + tb] = v;
            bvec[tb] = v;
        }
    }
    if (chain) {
        __syncthreads();
        factor_and_forward(S, k + 1, Winv, y, X, rowbuf, bvec, scal, r0, c0);
    } else if (upd) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) A[(size_t)(i * NB + r0 + a) * ld + j * NB + c0 + q] = S[a][q];
    }
}

// ---- the same factorisation as ONE launch: a resident chain workgroup and tile workgroups beside it (k_chol_solve) ----
// k_chol_step pays a kernel boundary on the serial chain of every step: kernel start, the reload of W_k and of two tiles another
// compute unit has just written, the drain of its own stores (~13 of a step's 31 us at C5).  Here the chain -- update of the next
// diagonal tile, its factorisation, y_k -- stays in one workgroup that never leaves its compute unit (chol_chain_body; W_k is used
// where factor_tile64 left it in LDS), and every other tile is taken by a workgroup that keeps it in registers through all its
// steps (chol_trail_tile).  They meet through epoch-valued flags in global memory (no reset between solves):
//   flag_w[k]         set by the chain once W_k^T and y_k are in global memory   -> awaited by every tile workgroup at its step k
//   tile_done[i nb+j] set by tile (i,j)'s workgroup once the tile is final in memory (for i = k+1 also: U_kj stored, b_j carried
//                     through step k)   -> awaited by the tiles of later rows that multiply with it and, for (s-1,s) and (s,s),
//                     by the chain before its step s
// Release: stores, workgroup barrier, then one thread's agent-scope fence and flag store; acquire: one thread polls (relaxed,
// agent scope), fences, workgroup barrier.  Every wait is bounded (CHOL_SPIN_MAX polls): on expiry scal[5] is raised, later waits
// give up at once and the kernel runs to its end on whatever it finds: the grid always drains; the host then repeats THAT trial
// on the one-launch-per-step form, which the problem keeps from there on (qsp_ba_optimize).  Every flag a workgroup
// waits for is set by the chain or by a tile of an EARLIER row -- a smaller TICKET, i.e. a workgroup that is running or done
// (k_chol_solve) -- and the chain waits for nothing a step ahead of it, so the scheme cannot wait on itself whichever workgroups
// the dispatcher starts first.  The chain's last steps have (transitively) awaited every tile_done of the solve: when the chain
// workgroup ends, every write of the tile workgroups is complete and visible to the kernels behind it in the stream.
// Same operations in the same order on every element as the k_chol_step path (the two agree in every bit).
#ifdef QSP_CB_STAMPS      // timing experiments only: shader-clock stamps of the chain workgroup's thread 0 at the phase boundaries of every step
__device__ unsigned long long qsp_chain_ts[64 * 8];
#define QSP_CHTS(s_, i_) { if (threadIdx.x == 0 && (s_) < 64) qsp_chain_ts[(s_) * 8 + (i_)] = __builtin_readcyclecounter(); }
#else
#define QSP_CHTS(s_, i_)
#endif
constexpr int CHOL_SPIN_MAX = 1 << 20;      // x (poll + s_sleep) ~ 1-2 us: a second or two
__device__ inline void chol_signal(unsigned* flag, unsigned epoch) {      // call after a workgroup barrier, one thread
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread: wait until *flag == epoch.  A wait that expires raises scal[5]; every later wait of the solve gives up at once
// (scal[5] is polled too), so a broken run costs one time-out, not one per step.
__device__ inline void chol_wait3(const unsigned* fa, const unsigned* fb, const unsigned* fc, unsigned epoch, double* scal) {
    const unsigned long long* dead = reinterpret_cast<const unsigned long long*>(scal + 5);
    for (int spin = 0; spin < CHOL_SPIN_MAX; ++spin) {
        // (the looks go out together: one round trip, not one per flag)
        const unsigned a = __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned b = fb ? __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        const unsigned c = fc ? __hip_atomic_load(fc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
        if (a == epoch && b == epoch && c == epoch) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return;
        }
        if ((spin & 255) == 0 && __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;
        __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(scal + 5), 0x3ff0000000000000ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 1.0
}
__device__ inline void chol_wait(const unsigned* flag, unsigned epoch, double* scal) { chol_wait3(flag, nullptr, nullptr, epoch, scal); }

// a wave-load of 64 consecutive doubles from a UNIFORM address: scalar base + this lane's 32-bit byte offset (one VGPR of address
// for all loads of the kernel instead of a 64-bit pointer per load in flight)
typedef const __attribute__((address_space(1))) char* gbytes_t;
__device__ __forceinline__ double ld_row(const double* uniform_ptr, uint32_t voff) {
    return *reinterpret_cast<const __attribute__((address_space(1))) double*>((gbytes_t)uniform_ptr + voff);
}

constexpr int CHAIN_LDS_DOUBLES = NB * (NB + 1) + 2 * NB * NB + CHOL_ROWBUF + 3 * NB + 2;
constexpr int TRAIL_LDS_DOUBLES = 3 * NB * NB + NB;
constexpr int CHOL_SOLVE_LDS_DOUBLES = CHAIN_LDS_DOUBLES > TRAIL_LDS_DOUBLES ? CHAIN_LDS_DOUBLES : TRAIL_LDS_DOUBLES;
constexpr int CHAIN_PROBE_GAP = 2;        // phases between a look at the flags and reading what it returned
constexpr int CHAIN_PARK_AFTER = 3;       // phases between issuing the next step's loads and parking them in LDS
// Fetching ahead (the first two updater waves in the second half of factor_tile64, 128 threads): thread 0 looks at the two flags
// the next step needs -- two relaxed loads, read CHAIN_PROBE_GAP phases later: they take 1-2 us to come back, a phase 0.8 us, and
// the thread must not sit in front of a phase barrier waiting for them -- and looks again until both are up; then it fences
// (acquire) and raises go[0]; every fetching thread that sees go[0] issues its share of the two tiles (2 x 32 doubles) and, the
// first wave, of b; CHAIN_PARK_AFTER phases later, or behind the last phase, the values are parked in LDS (Sn, Yp, bnext: all free
// during a factorisation).  go[1] = thread 0 has issued its loads (the others have by the
// phase after): read by the whole workgroup at the top of the next step.
struct ChainFetch {
    static constexpr bool active = true;
    const double* A;
    const double* b;
    const unsigned *fa, *fb;       // tile_done of (sn-1, sn) and (sn, sn)
    double *Sn, *Yp, *bnext;
    volatile int* go;
    unsigned epoch, f0, f1;
    int ld, sn, nb, t;
    int st;                        // thread 0: bit 0 a look is in flight, bit 1 flags up;  all: bit 2 loads issued, bit 3 parked
    int p_mark;
    struct Regs { double T[64], b2; };
    __device__ __forceinline__ void issue(int p, Regs& r) {
        // element e = t + 128 i of a tile: row (t >> 6) + 2 i, column t & 63 -- a uniform row base per i and ONE 32-bit offset per
        // thread (64 full addresses in flight would cost the fetching waves 128 registers)
        uint32_t voff = 8u * (uint32_t)((t >> 6) * ld + (t & 63));
        const double* baseS = A + (size_t)(sn * NB) * ld + sn * NB;
        const double* baseY = A + (size_t)((sn - 1) * NB) * ld + sn * NB;
        // (opaque to the optimiser: left alone it forms the 64 addresses in vector registers in front of the phase loop and keeps
        //  them there, 128 registers the loads' results then have to share with)
        asm volatile("" : "+s"(baseS), "+s"(baseY), "+v"(voff));
        const size_t step = (size_t)2 * ld;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            r.T[i] = ld_row(baseS, voff);
            r.T[32 + i] = ld_row(baseY, voff);
            baseS += step;
            baseY += step;
            asm volatile("" : "+s"(baseS), "+s"(baseY));
        }
        if (t < NB) r.b2 = b[sn * NB + t];
        st |= 4;
        p_mark = p;
    }
    __device__ __forceinline__ void park(Regs& r) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            Sn[t + 128 * i] = r.T[i];
            Yp[t + 128 * i] = r.T[32 + i];
        }
        if (t < NB) bnext[t] = r.b2;
        st |= 8;
    }
    // (loads issued in the last phases are parked behind the loop: the fetching waves then wait while the others publish W_s)
    __device__ __forceinline__ void finish(Regs& r) {
        if ((st & 4) && !(st & 8)) park(r);
    }
    __device__ __forceinline__ void phase(int p, Regs& r) {
        if (sn >= nb) return;
        if (st & 4) {
            if (!(st & 8) && p >= p_mark + CHAIN_PARK_AFTER) park(r);
            return;
        }
        if (sn >= 2) {
            if (t == 0 && !(st & 2) && p < NB / 4 - 1) {
                if (!(st & 1)) {
                    f0 = __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    f1 = __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    st |= 1;
                    p_mark = p;
                } else if (p >= p_mark + CHAIN_PROBE_GAP) {
                    if (f0 == epoch && f1 == epoch) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        go[0] = 1;
                        st |= 2;
                    } else {                        // look again
                        f0 = __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        f1 = __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        p_mark = p;
                    }
                }
            }
            if (go[0] == 0) return;
        }                                           // (step 1 reads the input itself: fetched at once)
        issue(p, r);
        if (t == 0) go[1] = 1;
    }
};

// One step s of the chain: [tiles (s-1,s), (s,s) and b_s -- normally fetched behind the previous factorisation] ->
// P = W_{s-1} A_{s-1,s} (the panel wave releases W_{s-1} to the tile workgroups meanwhile) -> S -= P^T P, U_{s-1,s} stored,
// b_s -= P^T y_{s-1} (panel wave) -> factor_tile64 -> W_s^T and y_s to memory.
__device__ __forceinline__ void chol_chain_body(double* chol_lds, const double* A, double* Uf, double* Winv, const double* b, double* y, int ld,
                                                int nb, double* scal, unsigned* flag_w, const unsigned* tile_done, unsigned epoch) {
    double* Wr = chol_lds;                          // rows of W_s (pitch 65): factor_tile64 writes them, the next step multiplies with them
    double* Yp = Wr + NB * (NB + 1);                // A_{s-1,s}, then P = W_{s-1} A_{s-1,s}
    double* Sn = Yp + NB * NB;                      // A_{s,s} as the fetching waves park it
    double* rowbuf = Sn + NB * NB;
    double* bvec = rowbuf + CHOL_ROWBUF;
    double* yprev = bvec + NB;
    double* bnext = yprev + NB;                     // b_s as the tile workgroups left it
    int* go_lds = reinterpret_cast<int*>(bnext + NB);
    const int t = threadIdx.x, r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    const int wave = t >> 6, lane = t & 63;
    const bool upd = t < 256;
    double S[4][4];
    if (upd) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[a][q] = A[(size_t)(r0 + a) * ld + c0 + q];
    }
    if (t < NB) bvec[t] = b[t];
    if (t == 0) go_lds[0] = go_lds[1] = 0;
    ChainFetch f;
    f.A = A; f.b = b; f.Sn = Sn; f.Yp = Yp; f.bnext = bnext; f.go = go_lds; f.epoch = epoch; f.ld = ld; f.nb = nb; f.t = t;
    f.st = 0; f.p_mark = 0; f.f0 = f.f1 = 0; f.sn = nb; f.fa = f.fb = tile_done;
    __syncthreads();
    for (int s = 0; s < nb; ++s) {
        QSP_CHTS(s, 0)
        if (s > 0) {
            // go[1] says for the whole workgroup whether the tiles are on their way: thread 0 issues in the phase in which it raises
            // go[0], every other fetching thread by the phase after, and go[0] is not raised in the last phase
            const bool have = go_lds[1] != 0;
            if (!have) {
                if (s >= 2) {                       // tiles (s-1,s) and (s,s) and b_s as their workgroups leave them
                    if (t == 0) {
                        chol_wait3(tile_done + (s - 1) * nb + s, tile_done + s * nb + s, nullptr, epoch, scal);
                    }
                    __syncthreads();
                }
                if (t < 128) {
                    ChainFetch::Regs r;
                    f.issue(0, r);
                    f.park(r);
                }
            }
            __syncthreads();                        // (everybody has read go[1]; Sn, Yp, bnext complete)
            QSP_CHTS(s, 1)
            if (upd) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) S[a][q] = Sn[(r0 + a) * NB + c0 + q];
            }
            if (t == 0) go_lds[0] = go_lds[1] = 0;
            QSP_CHTS(s, 2)
            // W_{s-1}^T and y_{s-1} have been in memory since the barrier that ended step s-1: released here, by the panel wave,
            // which has nothing else to do during the first product (the fence costs 1-2 k cycles)
            if (t == 256) chol_signal(flag_w + s - 1, epoch);
            d4_t pi[4] = {};
            if (upd) mfma_gemm64(Wr, 1, NB + 1, Yp, wave, lane, pi);      // P = W_{s-1} A_{s-1,s}: W's rows as factor_tile64 left them
            lds_barrier();                          // (LDS-only barriers inside a step: nobody waits for the stores of U here)
            QSP_CHTS(s, 3)
            if (upd) mfma_store64(Yp, pi, wave, lane);
            lds_barrier();
            if (upd) {
                d4_t acc[4] = {};
                mfma_gemm64(Yp, NB, 1, Yp, wave, lane, acc);
                mfma_store64(Wr, acc, wave, lane);      // (W_{s-1} has done its work; a plain 64x64 image: the way back to the 4x4 tiles)
                // the factor's block U_{s-1,s} = P, transposed (the tile workgroups of row s store the others)
                mfma_store_factor(Uf, ld, s, s - 1, pi, wave, lane);
            } else {                                // the panel wave, beside the second product: b_s -= P^T y_{s-1}
                double v = bnext[lane];
#pragma unroll 8
                for (int q = 0; q < NB; ++q) v -= Yp[q * NB + lane] * yprev[q];
                bvec[lane] = v;
            }
            lds_barrier();
            if (upd) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int q = 0; q < 4; ++q) S[a][q] -= Wr[(r0 + a) * NB + c0 + q];
            }                                       // (factor_tile64 writes W_s over the image behind its first barrier)
        }
        QSP_CHTS(s, 4)
        f.sn = s + 1;                               // the step to fetch for
        f.st = 0;
        f.fa = tile_done + (size_t)s * nb + s + 1;
        f.fb = tile_done + (size_t)(s + 1) * nb + s + 1;
        const bool bad = factor_tile64(S, Wr, rowbuf, r0, c0, f);
        if (bad) scal[3] = 1.0;
        QSP_CHTS(s, 5)
        double* Wg = Winv + (size_t)s * NB * NB;
        for (int e = t; e < NB * NB; e += CHOL_THREADS) Wg[e] = Wr[(e % NB) * (NB + 1) + e / NB];
        if (t >= 256) {                             // y_s = W_s b_s: the panel wave, a row each
            double v = 0;
            for (int m = 0; m < NB; ++m) v += Wr[lane * (NB + 1) + m] * bvec[m];
            y[s * NB + lane] = v;
            yprev[lane] = v;
        }
        __syncthreads();
        QSP_CHTS(s, 6)
    }
}

// everything off the chain: tile (i,j), 1 <= i <= j < nb, row-major in `tile` -- a tile only ever waits for the chain and for tiles
// of earlier rows, i.e. of SMALLER index.  The tile stays in registers across its steps k = 0 .. i-1 (a diagonal tile leaves its
// last step, k = i-1, to the chain); it goes back to memory once, final, with tile_done(i,j).  256 threads.
__device__ __forceinline__ void chol_trail_tile(double* chol_lds, int tile, double* A, double* Uf, const double* Winv, double* b, const double* y,
                                                int ld, int nb, double* scal, const unsigned* flag_w, unsigned* tile_done, unsigned epoch) {
    {
    int i = 1, rem = tile;
    while (rem >= nb - i) { rem -= nb - i; ++i; }
    const int j = i + rem;
    const int nsteps = (i == j) ? i - 1 : i;
    if (nsteps == 0) return;                          // (tile (1,1): the chain does its only step)
    double* X = chol_lds;
    double* Yi = chol_lds + NB * NB;
    double* Yj = (i == j) ? Yi : chol_lds + 2 * NB * NB;
    double* yk = chol_lds + 3 * NB * NB;
    const int t = threadIdx.x, r0 = (t >> 4) * 4, c0 = (t & 15) * 4;
    double S[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) S[a][q] = A[(size_t)(i * NB + r0 + a) * ld + j * NB + c0 + q];
    for (int k = 0; k < nsteps; ++k) {
        // row k is final once its workgroups have finished (row 0 is the input itself), W_k once the chain has got there: in the
        // steady state all three are up when this workgroup asks (the chain is a step ahead), so one look and ONE batch of loads
        if (t == 0)
            chol_wait3(flag_w + k, k > 0 ? tile_done + k * nb + i : nullptr, (k > 0 && i != j) ? tile_done + k * nb + j : nullptr, epoch, scal);
        __syncthreads();
        const double* Wg = Winv + (size_t)k * NB * NB;
        {
            double vx[16], vi[16], vj[16];            // (all 48 loads of a thread in flight before the first LDS store)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = t + 256 * u;
                vx[u] = Wg[e];
                vi[u] = A[(size_t)(k * NB + e / NB) * ld + i * NB + e % NB];
                vj[u] = (i != j) ? A[(size_t)(k * NB + e / NB) * ld + j * NB + e % NB] : 0.0;
            }
            if (t < NB) yk[t] = y[k * NB + t];        // (per-lane loads behind the acquire, not a uniform-address scalar load)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = t + 256 * u;
                X[e] = vx[u];
                Yi[e] = vi[u];
                if (i != j) Yj[e] = vj[u];
            }
        }
        __syncthreads();
        const int wave = t >> 6, lane = t & 63;
        d4_t pi[4] = {}, pj[4] = {};
        mfma_gemm64(X, NB, 1, Yi, wave, lane, pi);
        if (i != j) mfma_gemm64(X, NB, 1, Yj, wave, lane, pj);
        __syncthreads();
        mfma_store64(Yi, pi, wave, lane);
        if (i != j) mfma_store64(Yj, pj, wave, lane);
        __syncthreads();
        {
            d4_t acc[4] = {};
            mfma_gemm64(Yi, NB, 1, Yj, wave, lane, acc);
            mfma_store64(X, acc, wave, lane);        // (X is free since the first products: the way back to the 4x4 register tiles)
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) S[a][q] -= X[(r0 + a) * NB + c0 + q];
        if (i == k + 1) {                             // (j > i here) the factor's block U_kj and the right-hand side: b_j -= P_j^T y_k
            mfma_store_factor(Uf, ld, j, k, pj, wave, lane);
            if (t < NB) {
                double v = b[j * NB + t];
                for (int q = 0; q < NB; ++q) v -= Yj[q * NB + t] * yk[q];
                b[j * NB + t] = v;
            }
        }
        __syncthreads();                              // (the next step refills X, Yi, Yj, yk)
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) A[(size_t)(i * NB + r0 + a) * ld + j * NB + c0 + q] = S[a][q];
    __syncthreads();
    if (t == 0) chol_signal(tile_done + i * nb + j, epoch);
    }
}

// The factorisation in ONE launch on ONE stream (round 4).  Round 3 ran the chain and the tiles as two kernels on two streams
// that had to be seen running side by side (a handshake per device, an event + a cross-stream wait per solve: 7 us of idle device
// in front of every factorisation, and a way for two problems' stream pairs to block each other when streams share hardware
// queues).  Here every workgroup takes TICKETS from one counter: the first ticket of a solve is the chain, ticket n > 0 is tile
// n - 1.  A ticket is only ever held by a workgroup that is running, a tile waits for the chain and for tiles of smaller index
// only -- smaller tickets, i.e. running or finished workgroups -- so the grid makes progress whichever workgroups the dispatcher
// starts first and however few of them are resident (the blockIdx-strided loop of round 3 lost that guarantee beyond
// 23 block rows).  The counter is never reset: the host passes the value it had before the launch and knows what it will be
// afterwards (every workgroup draws exactly one ticket beyond the solve's range when it leaves).  The chain workgroup has 320
// threads; a workgroup that draws a tile first drops its fifth wave (s_barrier counts the waves that are left).
__global__ __launch_bounds__(CHOL_THREADS) void k_chol_solve(double* A, double* Uf, double* Winv, double* b, double* y, int ld, int nb, double* scal,
                                                             unsigned* flag_w, unsigned* tile_done, unsigned epoch, unsigned* ticket,
                                                             unsigned ticket_base, unsigned n_tiles) {
    extern __shared__ __attribute__((aligned(16))) double chol_lds[];
    __shared__ unsigned s_tk;
    if (threadIdx.x == 0) s_tk = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    unsigned tk = s_tk;
    if (tk == 0u) {
        chol_chain_body(chol_lds, A, Uf, Winv, b, y, ld, nb, scal, flag_w, tile_done, epoch);
        return;                                      // (the tiles are the other workgroups' -- the host launches at least one when there are any)
    }
    if (threadIdx.x >= 256) return;
    while (tk <= n_tiles) {
        chol_trail_tile(chol_lds, (int)tk - 1, A, Uf, Winv, b, y, ld, nb, scal, flag_w, tile_done, epoch);
        __syncthreads();                             // (everybody has read s_tk and is done with the tile's LDS)
        if (threadIdx.x == 0) s_tk = atomicAdd(ticket, 1u) - ticket_base;
        __syncthreads();
        tk = s_tk;
    }
}

#ifdef QSP_CB_STAMPS      // timing experiments only: shader-clock stamps of thread 0 at the phase boundaries of every step
__device__ unsigned long long qsp_cb_ts[64 * 5];
#define QSP_CBTS(j_, i_) { if (t == 0 && (j_) < 64) qsp_cb_ts[(j_) * 5 + (i_)] = __builtin_readcyclecounter(); }
#else
#define QSP_CBTS(j_, i_)
#endif
constexpr int CHOL_BACK_HELPERS = 4;     // workgroups on the solver's XCD that warm its L2 (k_chol_back)
// backward substitution x_k = W_k^T (y_k - sum_{j>k} U_kj x_j), k = nb-1 .. 0, in ONE launch of ONE workgroup of 1024
// threads (16 waves): y lives in LDS; at step j the waves form x_j = W_j^T y_j (4 rows each), then wave w takes the blocks
// k = w, w+16, ... < j and subtracts U_kj x_j from y_k with a lane per row (the factor is stored transposed, so a wave-load
// is 64 consecutive doubles).  Afterwards the same workgroup back-substitutes the eliminated objects.  2 nb workgroup
// barriers replace nb launches.  Measured alternatives at C5 (19 block rows), same box: this kernel 174 us; 512 threads with
// all 64 loads of a block in flight 201 us; one workgroup per block row chained by device-scope flags (all partial sums
// but the last off the critical path) 176-195 us -- a dependent round trip to another XCD's data costs ~3 us whichever way
// it is made, and every row needs three of them.  The simplest form was kept.
// two consecutive doubles from a UNIFORM address + this lane's byte offset (global_load_dwordx4: a compute unit pulls 148 GB/s out
// of its L2 with these against 76 GB/s with dwordx2 -- tools/micro/cu_stream.hip)
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2_t ld_row2(const double* uniform_ptr, uint32_t voff) {
    return *reinterpret_cast<const __attribute__((address_space(1))) d2_t*>((gbytes_t)uniform_ptr + voff);
}
__global__ __launch_bounds__(1024) void k_chol_back(Dev d, Par par, const double* __restrict__ Uf, const double* __restrict__ Winv,
                                                   const double* __restrict__ y, double* x, int n, int split) {
    extern __shared__ __attribute__((aligned(16))) double ysh[];       // [n] y, [NB] x_j, [n] x, split: [8 n] partial sums
    double* xj = ysh + n;
    double* xs = xj + NB;      // the solution stays in LDS until the end: a global store per step would sit in front of its barrier
    double* part = xs + n;     // split: part[((4 k + q) 2 + h) NB + m]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // uniform: block addresses stay in scalar registers
    const uint32_t voff = 8u * lane;
    const int nb = n / NB;
    if (blockIdx.x != 0) {
        // Helpers (speed only, no result): blocks b and b + 8 are observed to share an XCD, so blocks 8, 16, .. touch one double per
        // 128-byte line of the factor rows and diagonal inverses the solver is about to walk -- last block row first, which is the
        // order of use -- and leave.  The solver's loads then hit its XCD's L2 instead of travelling to the memory side.
        if (blockIdx.x & 7) return;
        const int h = (int)blockIdx.x / 8 - 1, H = ((int)gridDim.x - 1) / 8;
        double sink = 0;
        for (int R = n - 1 - h; R >= NB; R -= H) {            // row R of block row j = R / NB: columns [0, j NB) are read
            const double* row = Uf + (size_t)R * n;
            for (int q = t * 16; q < (R / NB) * NB; q += 1024 * 16) sink += row[q];
        }
        for (size_t q = ((size_t)h * 1024 + t) * 16; q < (size_t)nb * NB * NB; q += (size_t)H * 1024 * 16) sink += Winv[q];
        asm volatile("" ::"v"(sink));                        // (keeps the loads alive; nothing is written)
        return;
    }
    for (int i = t; i < n; i += 1024) ysh[i] = y[i];
    __syncthreads();
    // The product of block k with x_j: four chains of 16 columns (chain q: columns q, q + 4, ..).
    //  * split (the partial sums fit LDS beside y and x: up to ~2 000 unknowns): a UNIT is one chain of one block; the 4 j units of
    //    step j are dealt over the 16 waves one by one, and a unit is fetched with 16-byte loads -- a lane takes two adjacent rows,
    //    the lower half-wave the chain's even columns, the upper its odd ones: 1 KB per wave-load.  What bounds this kernel is what
    //    one compute unit can pull through its load path (round 2's form: whole blocks per wave, 8-byte loads, 41 GB/s).  A row's
    //    sum is ((e0 + o0) + (e1 + o1)) + ((e2 + o2) + (e3 + o3)), e / o the even / odd column sums of chain 0..3, each in ascending
    //    column order: fixed, so repeated runs agree in every bit.
    //  * otherwise: a wave takes whole blocks, a lane a row, 8-byte loads, (a0 + a1) + (a2 + a3) (round 2's arithmetic).
    // Nothing a step LOADS from global memory depends on the step before it, only what the loads are multiplied with does: W_{j-1}'s
    // rows are fetched while step j's products are being formed, and a wave's first batch before x_j exists.
    const int half = lane >> 5, r2 = lane & 31;
    const uint32_t voff2 = 8u * (uint32_t)(half * 4 * n + 2 * r2);
    double wt[NB / 16];
#pragma unroll
    for (int i = 0; i < NB / 16; ++i) wt[i] = ld_row(Winv + (size_t)(nb - 1) * NB * NB + (wave + 16 * i) * NB, voff);
    // a wave's first batch of step j: issued at the END of step j + 1 (in front of the loop for the first step), so that it is in
    // flight behind that step's tail and this step's W^T y; the barriers inside the loop wait for LDS traffic only (no global
    // store happens in it), so nothing waits for these loads before their first use
    double u[16];
    auto first_batch = [&](int j) {
        const double* Uj = Uf + (size_t)(j * NB) * n;
        if (split) {
            if (wave < 4 * j) {
                const int k0 = wave >> 2, q0 = wave & 3;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const d2_t v = ld_row2(Uj + k0 * NB + (size_t)(q0 + 8 * c) * n, voff2);
                    u[2 * c] = v.x; u[2 * c + 1] = v.y;
                }
            }
        } else if (wave < j) {
#pragma unroll
            for (int c = 0; c < 16; ++c) u[c] = ld_row(Uj + wave * NB + (size_t)c * n, voff);
        }
    };
    first_batch(nb - 1);
    for (int j = nb - 1; j >= 0; --j) {
        const double* Uj = Uf + (size_t)(j * NB) * n;     // element (m, c) of block k at Uj[k * NB + c * n + m]
        QSP_CBTS(j, 0)
        const int nunit = 4 * j;
        {                                                 // x_j[r] = sum_q W_j[q][r] y_j[q];  Winv holds WT[r][q] = W[q][r]
            const double yq = ysh[j * NB + lane];
            double pr[NB / 16];
#pragma unroll
            for (int i = 0; i < NB / 16; ++i) pr[i] = wt[i] * yq;
            LaneTranspose<NB / 16, 0>::run(pr, lane);         // lane i < 4 holds the sum of row wave + 16 i
            if (lane < NB / 16) {
                const int r = wave + 16 * lane;
                xj[r] = pr[0];
                xs[j * NB + r] = pr[0];
            }
            if (j > 0) {
#pragma unroll
                for (int i = 0; i < NB / 16; ++i) wt[i] = ld_row(Winv + (size_t)(j - 1) * NB * NB + (wave + 16 * i) * NB, voff);
            }
        }
        QSP_CBTS(j, 1)
        lds_barrier();
        QSP_CBTS(j, 2)
        if (split) {
            for (int uu = wave; uu < nunit; uu += 16) {
                const int k = uu >> 2, q = uu & 3;
                if (uu != wave) {                         // (the first unit's columns are already here)
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const d2_t v = ld_row2(Uj + k * NB + (size_t)(q + 8 * c) * n, voff2);
                        u[2 * c] = v.x; u[2 * c + 1] = v.y;
                    }
                }
                double alo = 0, ahi = 0;                  // rows 2 r2 and 2 r2 + 1, this half's columns q + 8 c + 4 half
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const double xc = xj[q + 8 * c + 4 * half];
                    alo += u[2 * c] * xc;
                    ahi += u[2 * c + 1] * xc;
                }
                {
                    d2_t pv;
                    pv.x = alo; pv.y = ahi;
                    *reinterpret_cast<d2_t*>(part + (size_t)(2 * uu + half) * NB + 2 * r2) = pv;
                }
            }
            if (j > 0) first_batch(j - 1);
            lds_barrier();
            for (int k = wave; k < j; k += 16) {
                const double* pk = part + (size_t)8 * k * NB + lane;
                ysh[k * NB + lane] -= ((pk[0] + pk[NB]) + (pk[2 * NB] + pk[3 * NB])) + ((pk[4 * NB] + pk[5 * NB]) + (pk[6 * NB] + pk[7 * NB]));
            }
        } else {
            for (int k = wave; k < j; k += 16) {
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 1
                for (int c0 = 0; c0 < NB; c0 += 16) {         // 16 independent wave-loads in flight, four accumulator chains
                    if (c0 > 0 || k > wave) {                 // (the first batch of the first block is already here)
#pragma unroll
                        for (int c = 0; c < 16; ++c) u[c] = ld_row(Uj + k * NB + (size_t)(c0 + c) * n, voff);
                    }
#pragma unroll
                    for (int c = 0; c < 16; c += 4) {
                        a0 += u[c + 0] * xj[c0 + c + 0];
                        a1 += u[c + 1] * xj[c0 + c + 1];
                        a2 += u[c + 2] * xj[c0 + c + 2];
                        a3 += u[c + 3] * xj[c0 + c + 3];
                    }
                }
                ysh[k * NB + lane] -= (a0 + a1) + (a2 + a3);
            }
            if (j > 0) first_batch(j - 1);
        }
        QSP_CBTS(j, 3)
        lds_barrier();
        QSP_CBTS(j, 4)
    }
    // (the last barrier of the loop has made xs complete.)  Only the rows that were solved: with the objects eliminated their
    // slots x[6 ho + c], ho >= n_dense, start at par.dim and would overlap the padding rows [dim, dimp) -- two writers, no
    // barrier in between.  Nobody reads the padding rows of x.
    for (int i = t; i < par.dim; i += 1024) x[i] = xs[i];
    if (par.elim) {                                       // x_o = G_o b_o - sum_e F_e^T x_k  (the dense x is complete in LDS)
        // A thread per (object, component); its object's edges eight at a time, each of the four dependent fetches (edge id ->
        // level, key-frame -> its index -> the column of F_e) issued for the whole batch before anything waits: edge after edge
        // this tail was a chain of 3 x (edges of an object) memory round trips -- ~20 us of the launch at C4 and C5 alike.  The
        // subtractions keep their order (edge by edge, m inside): the same bits.
        for (int i = t; i < 6 * d.n_obj; i += 1024) {
            const int ob = i / 6, c = i % 6;
            const int ho = d.obj_h[ob];
            const int q_end = d.obo_off[ob + 1];
            double v = d.obj_G[42 * (size_t)ob + 36 + c];
            if (ho < 0) continue;
            for (int q0 = d.obo_off[ob]; q0 < q_end; q0 += 8) {
                int e[8], kf[8], hk[8];
                uint8_t lv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = d.obo_edge[q0 + u < q_end ? q0 + u : q0];
#pragma unroll
                for (int u = 0; u < 8; ++u) { lv[u] = d.oe_level[e[u]]; kf[u] = d.oe_kf[e[u]]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) hk[u] = (q0 + u < q_end && !lv[u]) ? d.kf_h[kf[u]] : -1;
                double F[8][6];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double* Fe = d.oe_F + 36 * (size_t)e[u];
#pragma unroll
                    for (int m = 0; m < 6; ++m) F[u][m] = Fe[6 * m + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (hk[u] < 0) continue;
#pragma unroll
                    for (int m = 0; m < 6; ++m) v -= F[u][m] * xs[6 * hk[u] + m];
                }
            }
            x[6 * ho + c] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// back-substitution + oplus + the rho denominator   (block_solver.hpp:461-481, sparse_optimizer.cpp:423-435,
//                                                    optimization_algorithm_levenberg.cpp:182-189)
// ---------------------------------------------------------------------------------------------------------------
// edge -> hessian index of its key-frame (-1: edge at level 1 or key-frame fixed).  Levels and indices are fixed for the length of an
// optimize() call; the back-substitution of the landmarks reads this instead of level -> key-frame -> index (one dependent memory
// round trip instead of three per batch of observations).
// reset != 0: the call follows qsp_ba_set_levels(NULL, NULL, NULL) -- every edge active: the level arrays are written here instead of
// by two memset nodes (37 us of idle device in front of each, profiles/r04_ba_c4_timeline.txt)
__global__ __launch_bounds__(256) void k_edge_index(Dev d, int reset) {
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < d.n_edge) {
        if (reset) d.edge_level[a] = 0;
        d.edge_ha[a] = (!reset && d.edge_level[a]) ? -1 : d.kf_h[d.edge[a].kf];
    }
    if (reset && a < d.n_oe) d.oe_level[a] = 0;
}

__global__ __launch_bounds__(256) void k_update_points(Dev d, Par par) {
    __shared__ double sh[4];
    double sc = 0;
    for (int pt = blockIdx.x * 256 + threadIdx.x; pt < d.n_pt; pt += gridDim.x * 256) {
        if (d.pt_h[pt] < 0) continue;
        double c[3] = {d.bl[3 * (size_t)pt], d.bl[3 * (size_t)pt + 1], d.bl[3 * (size_t)pt + 2]};
        for (int a = d.pt_off[pt]; a < d.pt_off[pt + 1]; ++a) {
            if (d.edge_level[a]) continue;
            const int ha = d.kf_h[d.edge[a].kf];
            if (ha < 0) continue;
            double B[18];
            if (par.have_hpl) {                      // atomic-free mode keeps the blocks of this build (k_trial_stage1)
                const double* Bg = d.Hpl + 18 * (size_t)a;
                for (int i = 0; i < 18; ++i) B[i] = Bg[i];
            } else {
                edge_hpl(d, d.edge[a], par, B);
            }
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 6; ++i) c[j] -= B[3 * i + j] * d.xp[6 * ha + i];
        }
        const double* Di = d.Dinv + 9 * (size_t)pt;
        for (int i = 0; i < 3; ++i) {
            const double xl = Di[3 * i] * c[0] + Di[3 * i + 1] * c[1] + Di[3 * i + 2] * c[2];
            d.pt_xyz[3 * (size_t)pt + i] += xl;
            sc += xl * (par.lambda * xl + d.bl[3 * (size_t)pt + i]);
        }
    }
    const double s = block_sum_256(sc, sh);
    if (threadIdx.x == 0) d.partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_update_poses(Dev d, Par par, int n_partial) {
    // single workgroup: poses and objects, then the final sum of the scale partials
    __shared__ double sh[4];
    double sc = 0;
    for (int v = threadIdx.x; v < d.n_kf + d.n_obj; v += 256) {
        const bool is_kf = v < d.n_kf;
        const int h = is_kf ? d.kf_h[v] : d.obj_h[v - d.n_kf];
        if (h < 0) continue;
        double* pose = is_kf ? d.kf_pose + 7 * v : d.obj_pose + 7 * (v - d.n_kf);
        double dl[7], n[7];
        se3_exp(d.xp + 6 * h, dl);
        se3_mul(dl, pose, n);
        for (int i = 0; i < 7; ++i) pose[i] = n[i];
        if (par.is_root)
            for (int i = 0; i < 6; ++i) sc += d.xp[6 * h + i] * (par.lambda * d.xp[6 * h + i] + d.bp[6 * h + i]);
    }
    const double s = block_sum_256(sc, sh);
    if (threadIdx.x == 0) {
        double tot = s;
        for (int i = 0; i < n_partial; ++i) tot += d.partial[i];
        d.scal[1] = tot;
    }
}

// k_update_points and k_update_poses as ONE launch (one rank): blocks [0, gp) the landmarks, block gp the poses and objects; the
// scale partials go to the second half of `partial` (k_errors reuses the first) and are summed, in k_update_poses' order, by
// k_finish_sum_publish.
__global__ __launch_bounds__(256) void k_update_all(Dev d, Par par, int gp, int part_off) {
    __shared__ double sh[4];
    double sc = 0;
    if ((int)blockIdx.x < gp) {
        for (int pt = blockIdx.x * 256 + threadIdx.x; pt < d.n_pt; pt += gp * 256) {
            if (d.pt_h[pt] < 0) continue;
            double c[3] = {d.bl[3 * (size_t)pt], d.bl[3 * (size_t)pt + 1], d.bl[3 * (size_t)pt + 2]};
            const int a_end = d.pt_off[pt + 1];
            if (par.have_hpl) {
                // Eight observations' key-frame indices in one fetch (k_edge_index), then their 6x3 blocks and x_p four at a time,
                // each fetch issued for the whole batch before anything waits: one edge after the other this loop was a chain of
                // 3 x (observations) memory round trips per landmark, 20 of the launch's 23 us at C4.  The subtractions keep their
                // order (edge by edge, i inside), so every bit of the update is what it was.
                for (int a0 = d.pt_off[pt]; a0 < a_end; a0 += 8) {
                    int ha[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ha[u] = (a0 + u < a_end) ? d.edge_ha[a0 + u] : -1;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        if (a0 + 4 * g >= a_end) break;
                        double B[4][18], x[4][6];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int hv = ha[4 * g + u];
                            const int au = hv >= 0 ? a0 + 4 * g + u : a0, hu = hv >= 0 ? hv : 0;
                            const double* Bg = d.Hpl + 18 * (size_t)au;
#pragma unroll
                            for (int i = 0; i < 18; ++i) B[u][i] = Bg[i];
#pragma unroll
                            for (int i = 0; i < 6; ++i) x[u][i] = d.xp[6 * hu + i];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (ha[4 * g + u] < 0) continue;
                            for (int j = 0; j < 3; ++j)
                                for (int i = 0; i < 6; ++i) c[j] -= B[u][3 * i + j] * x[u][i];
                        }
                    }
                }
            } else {
                for (int a = d.pt_off[pt]; a < a_end; ++a) {
                    if (d.edge_level[a]) continue;
                    const int ha = d.kf_h[d.edge[a].kf];
                    if (ha < 0) continue;
                    double B[18];
                    edge_hpl(d, d.edge[a], par, B);
                    for (int j = 0; j < 3; ++j)
                        for (int i = 0; i < 6; ++i) c[j] -= B[3 * i + j] * d.xp[6 * ha + i];
                }
            }
            const double* Di = d.Dinv + 9 * (size_t)pt;
            for (int i = 0; i < 3; ++i) {
                const double xl = Di[3 * i] * c[0] + Di[3 * i + 1] * c[1] + Di[3 * i + 2] * c[2];
                d.pt_xyz[3 * (size_t)pt + i] += xl;
                sc += xl * (par.lambda * xl + d.bl[3 * (size_t)pt + i]);
            }
        }
    } else {
        for (int v = threadIdx.x; v < d.n_kf + d.n_obj; v += 256) {
            const bool is_kf = v < d.n_kf;
            const int h = is_kf ? d.kf_h[v] : d.obj_h[v - d.n_kf];
            if (h < 0) continue;
            double* pose = is_kf ? d.kf_pose + 7 * v : d.obj_pose + 7 * (v - d.n_kf);
            double dl[7], n[7];
            se3_exp(d.xp + 6 * h, dl);
            se3_mul(dl, pose, n);
            for (int i = 0; i < 7; ++i) pose[i] = n[i];
            if (par.is_root)
                for (int i = 0; i < 6; ++i) sc += d.xp[6 * h + i] * (par.lambda * d.xp[6 * h + i] + d.bp[6 * h + i]);
        }
    }
    const double s = block_sum_256(sc, sh);
    if (threadIdx.x == 0) d.partial[part_off + blockIdx.x] = s;
}

// The outlier classification between the two rounds of LocalBundleAdjustment (src/Optimizer_util.cc:621-654) on the device:
// level 1 = excluded from the second round.  chi2 is the value k_errors left for the last accepted estimates.
__global__ void k_classify_levels(Dev d, double th_mono, double th_stereo, double th_obj) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n_edge) {
        double p[3];
        se3_map(d.kf_pose + 7 * d.edge[i].kf, d.pt_xyz + 3 * d.edge[i].pt, p);
        d.edge_level[i] = (d.edge_chi2[i] > (d.edge[i].stereo ? th_stereo : th_mono) || !(p[2] > 0.0)) ? 1 : 0;
    }
    if (i < d.n_oe) d.oe_level[i] = d.oe_chi2[i] > th_obj ? 1 : 0;
}

// ---- The stage boundary of a local bundle adjustment WITHOUT the host (round 4; src/Optimizer_util.cc:598-661) -------------------------
// Between optimize(5) and optimize(10) the reference classifies every edge (chi2 of the last evaluated trial, depth at the current
// estimates), drops the outliers to level 1 and the robust kernels, and re-initialises: vertices without an active edge leave the
// index (sparse_optimizer.cpp:199-267).  Round 3 did that with a read-back of the levels, a host loop, an upload and two idle
// copy-engine hops (0.35 ms per BA, profiles/r03_ba_c4_timeline.txt).  Now the LAST iteration of the first stage enqueues, behind
// the trial's own publish and before the host has even seen its verdict: the classification into SECOND level arrays, the new
// landmark numbering, the edge -> key-frame index table, and the second stage's first chi2, linearisation and lambda seed on those
// -- published through a second pinned slot.  If the trial is accepted (it almost always is) the host swaps the buffers in and
// starts the second stage at its first trial; if it is rejected, or a key-frame / object vertex changed its activity (the reduced
// system would change shape), the speculative work is discarded and the host path of round 3 runs.
// act[]: one int per vertex [key-frames | objects | landmarks] holding the epoch of the last classification that saw an active edge
// at it (no clearing between runs).
__global__ __launch_bounds__(256) void k_tail_classify(Dev d, uint8_t* __restrict__ lvl_e, uint8_t* __restrict__ lvl_o, int32_t* __restrict__ edge_ha,
                                                       int32_t* __restrict__ act, int32_t epoch, double th_mono, double th_stereo, double th_obj) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d.n_edge) {
        const Edge E = d.edge[i];
        double p[3];
        se3_map(d.kf_pose + 7 * E.kf, d.pt_xyz + 3 * E.pt, p);
        const bool out = d.edge_chi2[i] > (E.stereo ? th_stereo : th_mono) || !(p[2] > 0.0);      // :621-641
        lvl_e[i] = out ? 1 : 0;
        edge_ha[i] = out ? -1 : d.kf_h[E.kf];                 // (k_edge_index of the second stage; valid while kf_h stands)
        if (!out) { act[E.kf] = epoch; act[d.n_kf + d.n_obj + E.pt] = epoch; }
    }
    if (i < d.n_oe) {
        const bool out = d.oe_chi2[i] > th_obj;                                                        // :650-654
        lvl_o[i] = out ? 1 : 0;
        if (!out) { act[d.oe_kf[i]] = epoch; act[d.n_kf + d.oe_obj[i]] = epoch; }
    }
}
// one workgroup: did a key-frame or object vertex change its activity (scal[6]); landmarks numbered in vertex-id order (pt_h2),
// their count (scal[7])
__global__ __launch_bounds__(1024) void k_tail_reindex(Dev d, const int32_t* __restrict__ act, int32_t epoch, int32_t* __restrict__ pt_h2) {
    __shared__ int s_cnt[1024];
    __shared__ int s_changed;
    const int t = threadIdx.x;
    if (t == 0) s_changed = 0;
    __syncthreads();
    int changed = 0;
    for (int i = t; i < d.n_kf; i += 1024) changed |= ((act[i] == epoch) && !d.kf_fixed[i]) != (d.kf_h[i] >= 0);
    for (int i = t; i < d.n_obj; i += 1024) changed |= (act[d.n_kf + i] == epoch) != (d.obj_h[i] >= 0);
    if (changed) s_changed = 1;
    const int per = (d.n_pt + 1023) / 1024, lo = min(t * per, d.n_pt), hi = min(lo + per, d.n_pt);
    const int32_t* pa = act + d.n_kf + d.n_obj;
    int c = 0;
    for (int q = lo; q < hi; ++q) c += pa[d.pt_order[q]] == epoch;
    s_cnt[t] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                     // inclusive scan
        const int v = t >= o ? s_cnt[t - o] : 0;
        __syncthreads();
        s_cnt[t] += v;
        __syncthreads();
    }
    int k = s_cnt[t] - c;
    for (int q = lo; q < hi; ++q) {
        const int pt = d.pt_order[q];
        pt_h2[pt] = (pa[pt] == epoch) ? k++ : -1;
    }
    if (t == 1023) { d.scal[7] = (double)s_cnt[1023]; d.scal[6] = s_changed ? 1.0 : 0.0; }
}
// the second stage's opening numbers to the host: chi2 of the new level set (partials of k_errors), lambda's seed, the two scalars above
__global__ __launch_bounds__(64) void k_tail_publish(Dev d, int n, double* __restrict__ host, double seq) {
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 64) s += d.partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        d.scal[0] = s;
        host[0] = s;
        host[1] = d.scal[2];
        host[2] = d.scal[6];
        host[3] = d.scal[7];
        __threadfence_system();
        *reinterpret_cast<volatile double*>(host + 4) = seq;
    }
}

__global__ void k_depth_positive(Dev d, uint8_t* pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n_edge) return;
    double p[3];
    se3_map(d.kf_pose + 7 * d.edge[i].kf, d.pt_xyz + 3 * d.edge[i].pt, p);
    pos[i] = p[2] > 0.0;
}


// ---------------------------------------------------------------------------------------------------------------
// Optimizer::PoseOptimization (reference src/Optimizer.cc:244-456) as ONE launch of ONE workgroup: a single free SE3
// vertex, unary only-pose edges (Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:266-358), 4 rounds of optimize(10)
// with g2o's Levenberg-Marquardt and the inlier / outlier classification after every round -- everything, including
// the LM control flow, stays on the device: the problem is a few hundred edges and 6 unknowns, so what matters is that
// no host round trip sits between its ~100 dependent steps.  Thread = edges k, k+256, ...; sums in a fixed tree.
// ---------------------------------------------------------------------------------------------------------------
struct PoseOptIn {
    int n;
    const double *X, *obs, *info;     // [n][3], [n][3] (u v u_right), [n]
    const uint8_t* stereo;            // [n]
    double K[5], pose0[7];
};
struct PoseOptOut {
    double pose[7];
    double trace[4][10][3];           // chi2, lambda, trials per (round, iteration)
    int iters[4];
    int n_inliers;
};

__device__ inline void po_residual(const double* pose, const double* X, const double* K, const double* obs, bool stereo,
                                   double* e, double* p) {
    se3_map(pose, X, p);
    if (!stereo) {                                   // project2d + cam_project, .cpp:290-296
        e[0] = obs[0] - (p[0] / p[2] * K[0] + K[2]);
        e[1] = obs[1] - (p[1] / p[2] * K[1] + K[3]);
        e[2] = 0.0;
    } else {                                         // .cpp:299-306: float 1/z, double bf
        const float invz = 1.0f / (float)p[2];
        const double u = p[0] * invz * K[0] + K[2], v = p[1] * invz * K[1] + K[3];
        e[0] = obs[0] - u;
        e[1] = obs[1] - v;
        e[2] = obs[2] - (u - K[4] * invz);
    }
}

__device__ inline void po_jacobian(const double* p, const double* K, bool stereo, double* J /*3x6*/) {   // .cpp:266-288,335-358
    const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz, fx = K[0], fy = K[1], bf = K[4];
    J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
    J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
    J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy;
    J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
    if (stereo) {
        J[12] = J[0] - bf * y * invz_2; J[13] = J[1] + bf * x * invz_2; J[14] = J[2];
        J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * invz_2;
    } else {
        for (int i = 12; i < 18; ++i) J[i] = 0;
    }
}

// sums v[0..N) over the 256 threads in a fixed tree; the totals are in tot[] for every thread after the call
template <int N>
__device__ inline void po_block_sum(double* v, double (*sh)[N], double* tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = v[i];
    LaneTranspose<N, 0>::run(w, lane);                    // lane i < N holds the wave's sum of v[i] (VALU only: this kernel is one
    if (lane < N) sh[wave][lane] = w[0];                  // workgroup running ~100 dependent steps, each with such a sum)
    __syncthreads();
    if (threadIdx.x < N) tot[threadIdx.x] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_pose_opt(PoseOptIn in, uint8_t* outlier, uint8_t* level, double* chi2, PoseOptOut* out) {
    __shared__ double sh[4][28];
    __shared__ double tot[28];
    __shared__ double pose[7], bk[7], x[6];
    __shared__ double s_lambda, s_ni, s_cur, s_ini, s_rho;
    __shared__ int s_qmax, s_nbad, s_again, s_stop, s_ok;
    const int t = threadIdx.x, n = in.n;
    const double dM = (double)(float)sqrt(5.991), dS = (double)(float)sqrt(7.815);
    for (int k = t; k < n; k += 256) { outlier[k] = 0; level[k] = 0; chi2[k] = 0.0; }
    if (t < 6) x[t] = 0.0;
    if (t == 0) {
        for (int r = 0; r < 4; ++r) out->iters[r] = 0;
        out->n_inliers = 0;
        for (int i = 0; i < 7; ++i) { pose[i] = in.pose0[i]; out->pose[i] = in.pose0[i]; }
    }
    __syncthreads();
    if (n < 3) return;                                               // :368-369
    bool robust = true;
    int n_bad_edges = 0;

    // robust chi2 of the active edges at `pose`; raw chi2 per edge kept (computeActiveErrors + activeRobustChi2)
    auto errors = [&](double* v /*[1]*/) {
        double a = 0;
        for (int k = t; k < n; k += 256) {
            if (level[k]) continue;
            double e[3], p[3], r0, r1;
            const bool st = in.stereo[k] != 0;
            po_residual(pose, in.X + 3 * k, in.K, in.obs + 3 * k, st, e, p);
            const double c = in.info[k] * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            chi2[k] = c;
            huber(c, robust ? (st ? dS : dM) : 0.0, r0, r1);
            a += r0;
        }
        v[0] = a;
    };

    for (int round = 0; round < 4; ++round) {
        if (t < 7) pose[t] = in.pose0[t];                            // vSE3->setEstimate(pFrame->mTcw), :381
        if (t == 0) { s_lambda = 0; s_ni = 2; s_nbad = 0; }
        __syncthreads();
        int done = 0;
        for (int it = 0; it < 10; ++it) {
            double v[28];
            errors(v + 27);
            // buildSystem: H (upper triangle, 21) and b (6)
            for (int i = 0; i < 27; ++i) v[i] = 0;
            for (int k = t; k < n; k += 256) {
                if (level[k]) continue;
                double e[3], p[3], J[18], r0, r1;
                const bool st = in.stereo[k] != 0;
                po_residual(pose, in.X + 3 * k, in.K, in.obs + 3 * k, st, e, p);
                po_jacobian(p, in.K, st, J);
                const double info = in.info[k];
                const double c = info * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
                huber(c, robust ? (st ? dS : dM) : 0.0, r0, r1);
                const double w = r1 * info;
                int q = 0;
                for (int i = 0; i < 6; ++i) {
                    double sb = 0;
                    for (int d = 0; d < 3; ++d) sb += J[6 * d + i] * (-info * e[d]) * r1;
                    v[21 + i] += sb;
                    for (int j = i; j < 6; ++j) {
                        double sum = 0;
                        for (int d = 0; d < 3; ++d) sum += J[6 * d + i] * J[6 * d + j];
                        v[q++] += w * sum;
                    }
                }
            }
            po_block_sum<28>(v, sh, tot);
            if (t == 0) {
                s_cur = tot[27];
                s_ini = tot[27];
                if (it == 0) {                                      // computeLambdaInit
                    double md = 0;
                    int q = 0;
                    for (int i = 0; i < 6; ++i) { md = fmax(fabs(tot[q]), md); q += 6 - i; }
                    s_lambda = 1e-5 * md; s_ni = 2; s_nbad = 0;
                }
                s_qmax = 0;
                s_rho = 0;
            }
            __syncthreads();
            while (true) {                                           // LM trials
                if (t == 0) {
                    for (int i = 0; i < 7; ++i) bk[i] = pose[i];
                    // (H + lambda I) x = b by Cholesky (upper-triangle storage, row i starts at 6i - i(i-1)/2)
                    double A[6][6];
                    int q = 0;
                    for (int i = 0; i < 6; ++i)
                        for (int j = i; j < 6; ++j) { A[i][j] = tot[q]; A[j][i] = tot[q]; ++q; }
                    for (int i = 0; i < 6; ++i) A[i][i] += s_lambda;
                    bool ok = true;
                    for (int j = 0; j < 6 && ok; ++j) {
                        double dd = A[j][j];
                        for (int q2 = 0; q2 < j; ++q2) dd -= A[j][q2] * A[j][q2];
                        if (!(dd > 0) || !isfinite(dd)) { ok = false; break; }
                        const double l = sqrt(dd);
                        A[j][j] = l;
                        for (int i = j + 1; i < 6; ++i) {
                            double s2 = A[i][j];
                            for (int q2 = 0; q2 < j; ++q2) s2 -= A[i][q2] * A[j][q2];
                            A[i][j] = s2 / l;
                        }
                    }
                    if (ok) {
                        double y[6];
                        for (int i = 0; i < 6; ++i) {
                            double s2 = tot[21 + i];
                            for (int q2 = 0; q2 < i; ++q2) s2 -= A[i][q2] * y[q2];
                            y[i] = s2 / A[i][i];
                        }
                        for (int i = 5; i >= 0; --i) {
                            double s2 = y[i];
                            for (int q2 = i + 1; q2 < 6; ++q2) s2 -= A[q2][i] * x[q2];
                            x[i] = s2 / A[i][i];
                        }
                        double dl[7], np[7];
                        se3_exp(x, dl);
                        se3_mul(dl, pose, np);
                        for (int i = 0; i < 7; ++i) pose[i] = np[i];
                    }
                    s_ok = ok ? 1 : 0;
                }
                __syncthreads();
                double tv[28];
                for (int i = 0; i < 28; ++i) tv[i] = 0;
                errors(tv);
                double* tsum = tot;                                   // keep H and b: reduce into a scratch row
                {
                    double a = tv[0];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
                    if ((t & 63) == 0) sh[t >> 6][0] = a;
                    __syncthreads();
                }
                if (t == 0) {
                    double tempChi = (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]);
                    if (!s_ok) tempChi = DBL_MAX;
                    double rho = s_cur - tempChi;
                    double scale = 1e-3;
                    for (int i = 0; i < 6; ++i) scale += x[i] * (s_lambda * x[i] + tsum[21 + i]);
                    rho /= scale;
                    if (rho > 0 && isfinite(tempChi)) {
                        double alpha = 1. - pow(2 * rho - 1, 3);
                        alpha = fmin(alpha, 2. / 3.);
                        s_lambda *= fmax(1. / 3., alpha);
                        s_ni = 2;
                        s_cur = tempChi;
                    } else {
                        s_lambda *= s_ni;
                        s_ni *= 2;
                        for (int i = 0; i < 7; ++i) pose[i] = bk[i];
                    }
                    s_qmax++;
                    s_rho = rho;
                    s_again = (rho < 0 && s_qmax < 10) ? 1 : 0;
                }
                __syncthreads();
                if (!s_again) break;
            }
            ++done;
            if (t == 0) {
                out->trace[round][it][0] = s_cur;
                out->trace[round][it][1] = s_lambda;
                out->trace[round][it][2] = (double)s_qmax;
                int stop = 0;
                if (s_qmax == 10 || s_rho == 0) stop = 1;
                else {
                    if ((s_ini - s_cur) * 1e3 < s_ini) s_nbad++; else s_nbad = 0;
                    if (s_nbad >= 3) stop = 1;
                }
                s_stop = stop;
            }
            __syncthreads();
            if (s_stop) break;
        }
        // classification, :384-437
        double nb[28];
        for (int i = 0; i < 28; ++i) nb[i] = 0;
        for (int k = t; k < n; k += 256) {
            const bool st = in.stereo[k] != 0;
            if (outlier[k]) {                                        // e->computeError() at the final pose
                double e[3], p[3];
                po_residual(pose, in.X + 3 * k, in.K, in.obs + 3 * k, st, e, p);
                chi2[k] = in.info[k] * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            }
            const float c = (float)chi2[k];
            if (c > (st ? 7.815f : 5.991f)) { outlier[k] = 1; level[k] = 1; nb[0] += 1.0; }
            else { outlier[k] = 0; level[k] = 0; }
        }
        po_block_sum<28>(nb, sh, tot);
        n_bad_edges = (int)tot[0];
        if (t == 0) out->iters[round] = done;
        if (round == 2) robust = false;                              // e->setRobustKernel(0)
        __syncthreads();
        if (n < 10) break;                                           // optimizer.edges().size() < 10
    }
    if (t == 0) {
        for (int i = 0; i < 7; ++i) out->pose[i] = pose[i];
        out->n_inliers = n - n_bad_edges;
    }
}

}  // namespace ba
}  // namespace qsp

// =================================================================================================================
// host side
// =================================================================================================================
using namespace qsp;
using namespace qsp::ba;

struct qsp_ba_problem {
    int device = 0;
    hipStream_t stream = nullptr;
    Dev d{};
    int n_mono = 0, n_stereo = 0;
    std::vector<void*> allocs;
    std::vector<size_t> alloc_bytes;      // (their sizes: freed chunks go to a per-device cache)
    size_t host_bytes[3] = {0, 0, 0};     // scal_host, lvl_host, idx_host as allocated
    char* pool_cur = nullptr;     // bump allocator over the chunks in `allocs`
    size_t pool_left = 0;
    // qsp_ba_create only: pinned staging of its ~30 uploads, so that they are asynchronous copies on the problem's stream with ONE
    // synchronisation at the end instead of a synchronous staged copy each (0.4 ms of a 1.3 ms create + destroy at C4)
    char* stage = nullptr;
    size_t stage_cap = 0, stage_used = 0, stage_alloc = 0;
    // staged uploads whose source AND destination follow each other (256-byte aligned both sides) leave as one copy (upload_flush)
    char* pend_dst = nullptr;
    size_t pend_src = 0, pend_bytes = 0;
    // host mirrors needed for index building
    std::vector<Edge> edge_h;
    std::vector<int32_t> pt_off_h, oe_kf_h, oe_obj_h;
    std::vector<uint8_t> kf_fixed_h, edge_level_h, oe_level_h;
    std::vector<int64_t> kf_id_h, pt_id_h, obj_id_h;
    std::vector<int32_t> mono_pos, st_pos;   // user index -> landmark-major position
    std::vector<int32_t> kf_h, obj_h, pt_h;
    int n_pose = 0, n_land = 0, dim = 0, dimp = 0, dimp_max = 0;   // dim / dimp: the DENSE reduced system (padded to NB)
    int n_dense = 0, dim_all = 0;      // pose blocks in the dense system; 6 * n_pose
    bool elim = false;                 // objects eliminated in front of the dense solve (k_obj_*)
    bool elim_allowed = true;          // qsp_ba_set_option(QSP_BA_OPT_OBJECT_ELIMINATION)
    int n_partial = 0;
    // LM state that persists inside one optimize call only
    qsp_ba_stats prof{};
    bool profiling = false;
    // the factorisation as ONE launch: a resident chain workgroup + tile workgroups that take tickets (k_chol_solve)
    unsigned* chol_flags = nullptr;        // [nb_max] flag_w, [nb_max^2] tile_done, then the ticket counter
    unsigned* chol_ticket = nullptr;       // (= chol_flags + nflag)
    unsigned chol_ticket_base = 0;         // the counter's value when everything enqueued so far has drained
    unsigned chol_epoch = 0;
    bool chol_chain_ok = false;            // the flags exist and QSP_BA_CHOL != steps
    bool chol_chain = false;               // QSP_BA_OPT_CHOLESKY_CHAIN
    bool chol_fault = false;               // (value 2 of the option: the tile workgroups are not launched)
    int chol_grid_max = 1;                 // tile workgroups launched at most: one per compute unit beside the chain's
    int chain_timeouts = 0;                // solves in which a flag wait expired: repeated on the one-launch-per-step form
    int n_boundary_device = 0, n_boundary_host = 0;      // local joint BAs by the path their stage boundary took (qsp_ba_stats)
    double* scal_host = nullptr; // pinned, device-visible copy of scal[0..3] (k_publish_scal): read back without a copy engine hop
    double* scal_host_dev = nullptr;
    double scal_seq = 0.0;       // sequence number of the last read-back enqueued
    bool speculate = true;       // enqueue the next iteration's linearisation before waiting for a trial's verdict
    // the device-side stage boundary of qsp_ba_local_joint (k_tail_*): second copies of what a re-initialisation rewrites
    uint8_t *edge_level2 = nullptr, *oe_level2 = nullptr;
    int32_t *edge_ha2 = nullptr, *pt_h2 = nullptr, *act = nullptr;
    int32_t act_epoch = 0;
    bool pt_order_uploaded = false;
    double tail_seq = 0.0;       // sequence number of the last k_tail_publish enqueued (scal_host[8..12])
    bool levels_reset_pending = false;   // qsp_ba_set_levels(NULL..): the device level arrays are zeroed by the next k_edge_index
    bool all_active = true;              // the host's level arrays are all zero (set_levels(NULL..) / a fresh problem)
    // the index of the all-active level set is a property of the problem: built once, restored by later first stages
    struct IndexCache { bool valid = false; std::vector<int32_t> kf_h, obj_h, pt_h; int n_pose, n_land, dim, dimp, n_dense, dim_all; bool elim, elim_allowed; } idx_cache;
    bool tail_ready = false;     // the last optimize() call left a valid speculated stage boundary behind
    bool host_stale = false;     // edge_level_h / oe_level_h / pt_h / n_land on the host lag behind the device (refresh_host_index)
    uint8_t* lvl_host = nullptr; // pinned: edge / object-edge levels classified on the device (qsp_ba_local_joint)
    int32_t* idx_host = nullptr; // pinned staging of the hessian indices [kf | obj | pt]
    std::vector<int32_t> idx_uploaded;   // what the device holds (an optimize() call re-uploads only what changed)
    std::vector<int> pose_order, pt_order;   // key-frame / object vertices and landmarks sorted by id (build_index)
    bool deterministic = false;   // qsp_ba_set_deterministic
    // landmark sharding across ranks: this rank linearises and marginalises the landmarks with
    // pt_id % world == rank (and the camera-object edges of objects with obj_id % world == rank); one SUM all-reduce of
    // the reduced system per LM trial
    int rank = 0, world = 1;
    qsp_allreduce_fn allreduce = nullptr;
    void* allreduce_ctx = nullptr;
    double* comm = nullptr;      // staging buffer for the small all-reduces
    size_t comm_cap = 0;
    std::vector<uint8_t> edge_foreign, oe_foreign;
    // qsp_ba_set_shard_rccl: the collectives are ncclAllReduce calls on `stream` (no host synchronisation per collective)
    ncclComm_t nccl = nullptr;
    uint8_t *d_pt_foreign = nullptr, *d_edge_foreign = nullptr, *d_oe_foreign = nullptr;   // device copies of the masks
};

static int upload_levels(qsp_ba_problem* p);
static void build_orders(qsp_ba_problem* p);

// Freed device chunks and pinned host buffers are kept per device for the next problem (bounded: 16 entries each, 512 MB of device
// memory): hipMalloc / hipFree and hipHostMalloc / hipHostFree cost 0.1-0.5 ms apiece, and a caller that builds a problem per bundle
// adjustment (the drop-in Optimizer) asks for the same sizes again and again.  Nothing relies on fresh memory being zero.
struct CachedBuf { void* p; size_t bytes; unsigned flags; };
static std::mutex g_buf_cache_mu;
static std::vector<CachedBuf> g_dev_cache[64], g_host_cache[64];
static void* buf_cache_take(std::vector<CachedBuf>* cache, int dev, size_t bytes, unsigned flags, size_t* got) {
    std::lock_guard<std::mutex> lk(g_buf_cache_mu);
    auto& v = cache[dev & 63];
    int best = -1;
    for (int i = 0; i < (int)v.size(); ++i)
        if (v[i].flags == flags && v[i].bytes >= bytes && v[i].bytes <= 2 * bytes + (1 << 16) && (best < 0 || v[i].bytes < v[best].bytes)) best = i;
    if (best < 0) return nullptr;
    void* q = v[best].p;
    *got = v[best].bytes;
    v.erase(v.begin() + best);
    return q;
}
// false: the cache is full, the caller frees the buffer
static bool buf_cache_put(std::vector<CachedBuf>* cache, int dev, void* q, size_t bytes, unsigned flags, size_t max_total) {
    std::lock_guard<std::mutex> lk(g_buf_cache_mu);
    auto& v = cache[dev & 63];
    size_t tot = bytes;
    for (const CachedBuf& c : v) tot += c.bytes;
    if (v.size() >= 16 || tot > max_total) return false;
    v.push_back({q, bytes, flags});
    return true;
}
static hipError_t host_alloc_cached(int dev, void** out, size_t bytes, unsigned flags, size_t* got) {
    *got = bytes;
    void* q = buf_cache_take(g_host_cache, dev, bytes, flags, got);
    if (q) { *out = q; return hipSuccess; }
    return hipHostMalloc(out, bytes, flags);
}

// device buffers come out of a few large chunks (a problem has ~50 of them; one hipMalloc each costs more than the
// uploads at the BASELINE sizes)
template <typename T>
static int dalloc(qsp_ba_problem* p, T** ptr, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
    if (bytes > p->pool_left) {
        size_t chunk = std::max<size_t>(bytes, (size_t)8 << 20);
        void* q = buf_cache_take(g_dev_cache, p->device, chunk, 0, &chunk);
        if (!q) QSP_HIP(hipMalloc(&q, chunk));
        p->allocs.push_back(q);
        p->alloc_bytes.push_back(chunk);
        if (chunk - bytes > p->pool_left) {          // keep whichever tail is larger for the next requests
            p->pool_cur = (char*)q + bytes;
            p->pool_left = chunk - bytes;
        }
        *ptr = (T*)q;
        return QSP_OK;
    }
    *ptr = (T*)p->pool_cur;
    p->pool_cur += bytes;
    p->pool_left -= bytes;
    return QSP_OK;
}
static int upload_flush(qsp_ba_problem* p) {
    if (!p->pend_bytes) return QSP_OK;
    const size_t n = p->pend_bytes;
    p->pend_bytes = 0;
    QSP_HIP(hipMemcpyAsync(p->pend_dst, p->stage + p->pend_src, n, hipMemcpyHostToDevice, p->stream));
    return QSP_OK;
}
template <typename T>
static int dupload(qsp_ba_problem* p, T** ptr, const T* src, size_t n) {
    int rc = dalloc(p, ptr, n);
    if (rc) return rc;
    const size_t bytes = n * sizeof(T);
    if (!n) return QSP_OK;
    if (p->stage && p->stream && bytes <= p->stage_cap - p->stage_used) {
        memcpy(p->stage + p->stage_used, src, bytes);
        const size_t padded = (bytes + 255) & ~(size_t)255;
        // qsp_ba_create uploads ~35 arrays: consecutive ones sit next to each other in the staging buffer and -- carved from the same
        // chunk by dalloc -- on the device, so they leave as ONE copy (0.28 ms of hipMemcpyAsync calls per problem otherwise)
        if (p->pend_bytes && (char*)*ptr == p->pend_dst + p->pend_bytes && p->stage_used == p->pend_src + p->pend_bytes) {
            p->pend_bytes += padded;
        } else {
            const int rc_f = upload_flush(p);
            if (rc_f) return rc_f;
            p->pend_dst = (char*)*ptr;
            p->pend_src = p->stage_used;
            p->pend_bytes = padded;
        }
        p->stage_used += padded;
    } else {
        const int rc_f = upload_flush(p);
        if (rc_f) return rc_f;
        QSP_HIP(hipMemcpy(*ptr, src, bytes, hipMemcpyHostToDevice));
    }
    return QSP_OK;
}

static void csr_build(int n_rows, const std::vector<int32_t>& row_of, std::vector<int32_t>& off, std::vector<int32_t>& idx) {
    off.assign(n_rows + 1, 0);
    for (int32_t r : row_of) off[r + 1]++;
    for (int i = 0; i < n_rows; ++i) off[i + 1] += off[i];
    idx.assign(row_of.size(), 0);
    std::vector<int32_t> cur(off.begin(), off.end() - 1);
    for (size_t e = 0; e < row_of.size(); ++e) idx[cur[row_of[e]]++] = (int32_t)e;
}

// pairs of key-frames (by scene index, ka <= kb) that observe a common landmark, and per pair the (edge in ka, edge in kb,
// landmark) entries in landmark order: counting sort over the keys ka * n_kf + kb.  Returns QSP_ERR_UNSUPPORTED when the
// lists would exceed PAIR_CAP entries.
constexpr int64_t PAIR_CAP = (int64_t)4 << 20;
static int64_t count_pairs(const qsp_ba_problem* p) {
    int64_t n = 0;
    for (int l = 0; l < p->d.n_pt; ++l) {
        const int64_t k = p->pt_off_h[l + 1] - p->pt_off_h[l];
        n += k * (k + 1) / 2;
    }
    return n;
}
static int build_pair_lists(qsp_ba_problem* p) {
    Dev& d = p->d;
    if (d.pk_off) return QSP_OK;
    if (count_pairs(p) > PAIR_CAP || (int64_t)d.n_kf * d.n_kf > ((int64_t)1 << 24))
        return qsp_fail(QSP_ERR_UNSUPPORTED, "deterministic mode: graph too large for the pair lists");
    {
        const int nk = d.n_kf;
        const std::vector<Edge>& E = p->edge_h;
        const std::vector<int32_t>& off = p->pt_off_h;
        std::vector<int64_t> cnt((size_t)nk * nk, 0), cnto((size_t)nk * nk, 0);
        auto for_pairs = [&](auto&& fn) {
            for (int l = 0; l < d.n_pt; ++l)
                for (int a = off[l]; a < off[l + 1]; ++a)
                    for (int b = a; b < off[l + 1]; ++b) {
                        const int ka = E[a].kf, kb = E[b].kf;
                        if (ka == kb && a != b) continue;          // (two edges of one landmark in one key-frame: skipped
                        if (ka <= kb) fn(ka, kb, a, b);            //  like in the atomic kernels)
                        else fn(kb, ka, b, a);
                    }
        };
        // pairs of key-frames that observe a common OBJECT (object elimination): per object every ordered choice of two of
        // its edges whose key-frames satisfy ka <= kb, the edge with itself included
        std::vector<std::vector<int32_t>> obj_edges(d.n_obj);
        for (int e = 0; e < d.n_oe; ++e) obj_edges[p->oe_obj_h[e]].push_back(e);
        auto for_obj_pairs = [&](auto&& fn) {
            for (int o = 0; o < d.n_obj; ++o)
                for (int32_t e : obj_edges[o])
                    for (int32_t e2 : obj_edges[o]) {
                        const int ka = p->oe_kf_h[e], kb = p->oe_kf_h[e2];
                        if (ka < kb || (ka == kb && e <= e2)) fn(ka, kb, e, e2, o);
                        if (ka == kb && e < e2) fn(ka, kb, e2, e, o);     // both orders land in the same diagonal block
                    }
        };
        for_pairs([&](int ka, int kb, int, int) { cnt[(size_t)ka * nk + kb]++; });
        for_obj_pairs([&](int ka, int kb, int, int, int) { cnto[(size_t)ka * nk + kb]++; });
        std::vector<int32_t> pk_ka, pk_kb, pk_off(1, 0), pko_off(1, 0);
        std::vector<int64_t> start((size_t)nk * nk, -1), starto((size_t)nk * nk, -1);
        int64_t tot = 0, toto = 0;
        for (int ka = 0; ka < nk; ++ka)
            for (int kb = ka; kb < nk; ++kb) {
                const int64_t c = cnt[(size_t)ka * nk + kb], co = cnto[(size_t)ka * nk + kb];
                if (!c && !co) continue;
                start[(size_t)ka * nk + kb] = tot;
                starto[(size_t)ka * nk + kb] = toto;
                tot += c;
                toto += co;
                pk_ka.push_back(ka); pk_kb.push_back(kb);
                pk_off.push_back((int32_t)tot); pko_off.push_back((int32_t)toto);
            }
        std::vector<int4> ent((size_t)std::max<int64_t>(tot, 1)), ento((size_t)std::max<int64_t>(toto, 1));
        // (the walk over the landmark pairs runs twice, count and fill: keeping the first walk's pairs in a flat array for the fill
        //  was slower -- 1.6 ns per step of the walk against three more arrays to write and read)
        for_pairs([&](int ka, int kb, int a, int b) { ent[(size_t)start[(size_t)ka * nk + kb]++] = make_int4(a, b, E[a].pt, 0); });
        for_obj_pairs([&](int ka, int kb, int e, int e2, int o) { ento[(size_t)starto[(size_t)ka * nk + kb]++] = make_int4(e, e2, o, 0); });
        d.n_pk = (int32_t)pk_ka.size();
        std::vector<int32_t> work, small;
        for (int q = 0; q < d.n_pk; ++q) {
            const int len = (pk_off[q + 1] - pk_off[q]) + (pko_off[q + 1] - pko_off[q]);
            (len > 192 ? work : small).push_back(q);
        }
        d.n_pk_big = (int32_t)work.size();
        work.insert(work.end(), small.begin(), small.end());
        int rc = dupload(p, &d.pk_ka, pk_ka.data(), pk_ka.size());
        if (!rc) rc = dupload(p, &d.pk_work, work.data(), work.size());
        if (!rc) rc = dupload(p, &d.pk_kb, pk_kb.data(), pk_kb.size());
        if (!rc) rc = dupload(p, &d.pk_off, pk_off.data(), pk_off.size());
        if (!rc) rc = dupload(p, &d.pk_ent, ent.data(), ent.size());
        if (!rc) rc = dupload(p, &d.pko_off, pko_off.data(), pko_off.size());
        if (!rc) rc = dupload(p, &d.pko_ent, ento.data(), ento.size());
        if (!rc) rc = dalloc(p, &d.Hpl, 18 * (size_t)std::max(d.n_edge, 1));
        if (rc) return rc;
    }
    return QSP_OK;
}

// Streams are kept for the life of the process: a problem takes a set (its stream, the second, high-priority one of the chain
// factorisation, their event) from a per-device pool and hands it back when it is destroyed.  Creating a stream costs
// milliseconds -- the hardware queue behind it comes into being with its first launch: 5.5 ms per create + destroy and 8 ms on
// the first solve of a fresh problem were measured with one pair per problem (tools/time_ba_create.py) -- and the drop-in
// Optimizer creates a problem per LocalJointBundleAdjustment call.
struct StreamSet { hipStream_t s = nullptr; };
static std::mutex g_stream_pool_mu;
static std::vector<StreamSet> g_stream_pool[64];
static bool stream_set_acquire(int dev, StreamSet* out) {
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        auto& v = g_stream_pool[dev & 63];
        if (!v.empty()) {
            *out = v.back();
            v.pop_back();
            return true;
        }
    }
    *out = StreamSet();
    return hipStreamCreateWithFlags(&out->s, hipStreamNonBlocking) == hipSuccess;
}
static void stream_set_release(int dev, const StreamSet& st) {
    if (!st.s) return;
    (void)hipStreamSynchronize(st.s);
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_stream_pool[dev & 63].push_back(st);
}

// Flags and ticket counter of the chain factorisation (k_chol_solve).  QSP_BA_CHOL=steps selects the one-launch-per-step form.
static int chol_chain_setup(qsp_ba_problem* p) {
    const char* env = getenv("QSP_BA_CHOL");
    if (env && !strcmp(env, "steps")) return QSP_OK;
    if (p->dimp_max < 2 * NB) return QSP_OK;          // one block: nothing to overlap
    const int nbm = p->dimp_max / NB;
    const size_t nflag = (size_t)nbm + (size_t)nbm * nbm;
    int rc = dalloc(p, &p->chol_flags, nflag + 8);
    if (rc) return rc;
    // From here on a failure is not an error of qsp_ba_create: the problem keeps the one-launch-per-step form.
    auto ok = [](hipError_t e) {
        if (e != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess;
    };
    int n_cu = 0;
    if (!ok(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, p->device)) || n_cu < 8) return QSP_OK;
    p->chol_grid_max = n_cu - 1;
    if (!ok(hipMemsetAsync(p->chol_flags, 0, sizeof(unsigned) * (nflag + 8), p->stream))) return QSP_OK;
    p->chol_ticket = p->chol_flags + nflag;
    p->chol_ticket_base = 0;
    p->chol_chain_ok = p->chol_chain = true;
    return QSP_OK;
}

extern "C" int qsp_ba_create(const qsp_ba_scene* s, int device, qsp_ba_problem** out) {
    // QSP_BA_CREATE_TIMING=1: host-side phases of this call on stderr (tools/time_ba_create.py)
    static const bool timing = getenv("QSP_BA_CREATE_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "qsp_ba_create: %-34s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_prev).count());
        t_prev = t;
    };
    if (!s || !out) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_create: null argument");
    if (s->n_kf <= 0 || s->n_pt < 0 || s->n_obj < 0 || s->n_mono < 0 || s->n_stereo < 0 || s->n_objedge < 0)
        return qsp_fail(QSP_ERR_INVALID, "qsp_ba_create: negative count");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return qsp_fail(QSP_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return qsp_fail(QSP_ERR_INVALID, "device index out of range");
    for (int e = 0; e < s->n_mono; ++e)
        if (s->mono_pt[e] < 0 || s->mono_pt[e] >= s->n_pt || s->mono_kf[e] < 0 || s->mono_kf[e] >= s->n_kf)
            return qsp_fail(QSP_ERR_INVALID, "qsp_ba_create: mono edge index out of range");
    for (int e = 0; e < s->n_stereo; ++e)
        if (s->stereo_pt[e] < 0 || s->stereo_pt[e] >= s->n_pt || s->stereo_kf[e] < 0 || s->stereo_kf[e] >= s->n_kf)
            return qsp_fail(QSP_ERR_INVALID, "qsp_ba_create: stereo edge index out of range");
    for (int e = 0; e < s->n_objedge; ++e)
        if (s->objedge_kf[e] < 0 || s->objedge_kf[e] >= s->n_kf || s->objedge_obj[e] < 0 || s->objedge_obj[e] >= s->n_obj)
            return qsp_fail(QSP_ERR_INVALID, "qsp_ba_create: object edge index out of range");
    QSP_HIP(hipSetDevice(device));
    qsp_ba_problem* p = new qsp_ba_problem();
    p->device = device;
    {
        StreamSet st;
        if (!stream_set_acquire(p->device, &st)) { delete p; return qsp_fail(QSP_ERR_DEVICE, "qsp_ba_create: no stream"); }
        p->stream = st.s;
        const size_t want = (size_t)24 << 20;
        if (host_alloc_cached(p->device, (void**)&p->stage, want, hipHostMallocDefault, &p->stage_alloc) == hipSuccess) p->stage_cap = p->stage_alloc;
        else { (void)hipGetLastError(); p->stage = nullptr; }
    }
    lap("checks, stream, staging buffer");
    Dev& d = p->d;
    d.n_kf = s->n_kf; d.n_pt = s->n_pt; d.n_obj = s->n_obj; d.n_oe = s->n_objedge;
    d.n_edge = s->n_mono + s->n_stereo;
    d.oe_info = s->objedge_info;
    p->n_mono = s->n_mono; p->n_stereo = s->n_stereo;
    // landmark-major edge order: stable by (landmark, then mono before stereo, then caller order)
    std::vector<int32_t> row(d.n_edge);
    for (int e = 0; e < s->n_mono; ++e) row[e] = s->mono_pt[e];
    for (int e = 0; e < s->n_stereo; ++e) row[s->n_mono + e] = s->stereo_pt[e];
    std::vector<int32_t> order;
    csr_build(d.n_pt, row, p->pt_off_h, order);
    p->edge_h.resize(d.n_edge);
    p->mono_pos.assign(s->n_mono, 0);
    p->st_pos.assign(s->n_stereo, 0);
    std::vector<int32_t> kf_of(d.n_edge);
    for (int pos = 0; pos < d.n_edge; ++pos) {
        const int src = order[pos];
        Edge& E = p->edge_h[pos];
        if (src < s->n_mono) {
            E.obs[0] = s->mono_obs[2 * src]; E.obs[1] = s->mono_obs[2 * src + 1]; E.obs[2] = 0;
            E.info = s->mono_info[src]; E.pt = s->mono_pt[src]; E.kf = s->mono_kf[src]; E.stereo = 0; E.user = src;
            p->mono_pos[src] = pos;
        } else {
            const int q = src - s->n_mono;
            E.obs[0] = s->stereo_obs[3 * q]; E.obs[1] = s->stereo_obs[3 * q + 1]; E.obs[2] = s->stereo_obs[3 * q + 2];
            E.info = s->stereo_info[q]; E.pt = s->stereo_pt[q]; E.kf = s->stereo_kf[q]; E.stereo = 1; E.user = q;
            p->st_pos[q] = pos;
        }
        kf_of[pos] = E.kf;
    }
    std::vector<int32_t> kf_off, kf_edge, kfo_off, kfo_edge, obo_off, obo_edge;
    csr_build(d.n_kf, kf_of, kf_off, kf_edge);
    p->oe_kf_h.assign(s->objedge_kf, s->objedge_kf + s->n_objedge);
    p->oe_obj_h.assign(s->objedge_obj, s->objedge_obj + s->n_objedge);
    csr_build(d.n_kf, p->oe_kf_h, kfo_off, kfo_edge);
    csr_build(d.n_obj, p->oe_obj_h, obo_off, obo_edge);
    // landmark chunks for k_lin_points (<= 256 edges and <= 256 landmarks each) and key-frame splits for k_lin_poses
    std::vector<int32_t> chunk_pt(1, 0);
    {
        int begin = 0;
        for (int pt = 0; pt < d.n_pt; ++pt) {
            // a landmark with more than 256 observations forms a chunk of its own (the kernel walks it in windows)
            const bool heavy = p->pt_off_h[pt + 1] - p->pt_off_h[pt] > 256;
            if (pt > begin && (heavy || p->pt_off_h[pt + 1] - p->pt_off_h[begin] > 256 || pt - begin >= 256)) {
                chunk_pt.push_back(pt);
                begin = pt;
            }
            if (heavy && pt + 1 < d.n_pt) {
                chunk_pt.push_back(pt + 1);
                begin = pt + 1;
            }
        }
        chunk_pt.push_back(d.n_pt);
        if (d.n_pt == 0) chunk_pt.assign(2, 0);
    }
    d.n_chunk = (d.n_pt == 0) ? 0 : (int)chunk_pt.size() - 1;
    std::vector<int32_t> ksp_kf, ksp_begin, ksp_end, ksp_first(d.n_kf + 1, 0);
    const int ksplit = d.n_edge < KSPLIT_SMALL_BELOW ? KSPLIT_SMALL : KSPLIT;
    for (int k = 0; k < d.n_kf; ++k) {
        ksp_first[k] = (int)ksp_kf.size();
        for (int b0 = kf_off[k]; b0 < kf_off[k + 1]; b0 += ksplit) {
            ksp_kf.push_back(k);
            ksp_begin.push_back(b0);
            ksp_end.push_back(std::min(b0 + ksplit, kf_off[k + 1]));
        }
    }
    ksp_first[d.n_kf] = (int)ksp_kf.size();
    d.n_ksplit = (int)ksp_kf.size();
    p->kf_fixed_h.assign(s->kf_fixed, s->kf_fixed + s->n_kf);
    p->kf_id_h.assign(s->kf_id, s->kf_id + s->n_kf);
    p->pt_id_h.assign(s->pt_id, s->pt_id + s->n_pt);
    p->obj_id_h.assign(s->obj_id, s->obj_id + s->n_obj);
    p->edge_level_h.assign(d.n_edge, 0);
    p->oe_level_h.assign(d.n_oe, 0);
    p->dimp_max = ((6 * (d.n_kf + d.n_obj) + NB - 1) / NB) * NB;
    lap("edge order, CSRs, chunks (host)");
    int rc = QSP_OK;
#define UP(field, src, n) if (!rc) rc = dupload(p, &d.field, src, (size_t)(n))
#define AL(field, n) if (!rc) rc = dalloc(p, &d.field, (size_t)(n))
    UP(kf_pose, s->kf_pose, 7 * d.n_kf);
    UP(pt_xyz, s->pt_xyz, 3 * d.n_pt);
    UP(obj_pose, s->obj_pose, 7 * d.n_obj);
    UP(kf_K, s->kf_K, 5 * d.n_kf);
    AL(kf_bk, 7 * d.n_kf); AL(pt_bk, 3 * d.n_pt); AL(obj_bk, 7 * d.n_obj);
    UP(edge, p->edge_h.data(), d.n_edge);
    AL(edge_level, d.n_edge); AL(edge_ha, std::max(d.n_edge, 1)); AL(edge_chi2, d.n_edge);
    UP(pt_off, p->pt_off_h.data(), d.n_pt + 1);
    UP(kf_off, kf_off.data(), d.n_kf + 1);
    UP(kf_edge, kf_edge.data(), d.n_edge);
    UP(chunk_pt, chunk_pt.data(), chunk_pt.size());
    UP(ksp_kf, ksp_kf.data(), ksp_kf.size());
    UP(ksp_begin, ksp_begin.data(), ksp_begin.size());
    UP(ksp_end, ksp_end.data(), ksp_end.size());
    UP(ksp_first, ksp_first.data(), ksp_first.size());
    AL(kpart, 27 * (size_t)std::max(d.n_ksplit, 1));
    UP(oe_kf, s->objedge_kf, d.n_oe);
    UP(oe_obj, s->objedge_obj, d.n_oe);
    UP(oe_meas, s->objedge_meas, 7 * d.n_oe);
    AL(oe_level, d.n_oe); AL(oe_chi2, d.n_oe);
    UP(kfo_off, kfo_off.data(), d.n_kf + 1);
    UP(kfo_edge, kfo_edge.data(), d.n_oe);
    UP(obo_off, obo_off.data(), d.n_obj + 1);
    UP(obo_edge, obo_edge.data(), d.n_oe);
    AL(kf_h, d.n_kf); AL(obj_h, d.n_obj); AL(pt_h, d.n_pt);
    UP(kf_fixed, p->kf_fixed_h.data(), d.n_kf);
    build_orders(p);
    {
        std::vector<int32_t> po(p->pt_order.begin(), p->pt_order.end());
        UP(pt_order, po.data(), d.n_pt);                   // (g2o's numbering order of the landmarks, for k_tail_reindex)
        p->pt_order_uploaded = true;
    }
    if (!rc) rc = dalloc(p, &p->edge_level2, (size_t)std::max(d.n_edge, 1));
    if (!rc) rc = dalloc(p, &p->oe_level2, (size_t)std::max(d.n_oe, 1));
    if (!rc) rc = dalloc(p, &p->edge_ha2, (size_t)std::max(d.n_edge, 1));
    if (!rc) rc = dalloc(p, &p->pt_h2, (size_t)std::max(d.n_pt, 1));
    if (!rc) rc = dalloc(p, &p->act, (size_t)(d.n_kf + d.n_obj + d.n_pt));
    AL(Hll, 9 * (size_t)d.n_pt); AL(bl, 3 * (size_t)d.n_pt); AL(Dinv, 9 * (size_t)d.n_pt); AL(xl, 3 * (size_t)d.n_pt);
    AL(Hdiag, 36 * (size_t)(d.n_kf + d.n_obj));
    AL(Hoff, 36 * (size_t)d.n_oe);
    AL(oe_rec, 84 * (size_t)d.n_oe);
    AL(oe_F, 36 * (size_t)d.n_oe);
    AL(obj_G, 42 * (size_t)d.n_obj);
    AL(bp, p->dimp_max); AL(bs, p->dimp_max); AL(xp, p->dimp_max);
    AL(Hs, (size_t)p->dimp_max * p->dimp_max + p->dimp_max);   // + room for bs right behind the matrix
    AL(Uf, (size_t)p->dimp_max * p->dimp_max); AL(Winv, (size_t)p->dimp_max * NB); AL(ych, p->dimp_max);
    p->n_partial = 1024;
    AL(partial, 2 * p->n_partial + 8);     // [0, n_partial): chi2 partials of k_errors / update partials; [n_partial, ..): k_update_all
    AL(scal, 8);
#undef UP
#undef AL
    if (!rc) {
        hipError_t e = hipMemsetAsync(d.edge_level, 0, std::max(d.n_edge, 1), p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d.scal, 0, sizeof(double) * 8, p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d.oe_level, 0, std::max(d.n_oe, 1), p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d.edge_chi2, 0, sizeof(double) * std::max(d.n_edge, 1), p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d.oe_chi2, 0, sizeof(double) * std::max(d.n_oe, 1), p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(p->act, 0, sizeof(int32_t) * (size_t)(d.n_kf + d.n_obj + d.n_pt), p->stream);
        if (e == hipSuccess) e = host_alloc_cached(p->device, (void**)&p->scal_host, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent, &p->host_bytes[0]);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&p->scal_host_dev, p->scal_host, 0);
        if (e == hipSuccess) p->scal_host[4] = p->scal_host[12] = 0.0;      // ([0..4] a trial's scalars + sequence word, [8..12] k_tail_publish's)
        if (e == hipSuccess) e = host_alloc_cached(p->device, (void**)&p->lvl_host, (size_t)std::max(d.n_edge + d.n_oe, 1), hipHostMallocDefault, &p->host_bytes[1]);
        if (e == hipSuccess) e = host_alloc_cached(p->device, (void**)&p->idx_host, sizeof(int32_t) * (size_t)std::max(d.n_kf + d.n_obj + d.n_pt, 1), hipHostMallocDefault, &p->host_bytes[2]);
        const int chol_lds = (int)(sizeof(double) * CHOL_LDS_DOUBLES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_chol_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * CHOL_SOLVE_LDS_DOUBLES));
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_chol_first, hipFuncAttributeMaxDynamicSharedMemorySize, chol_lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_chol_step, hipFuncAttributeMaxDynamicSharedMemorySize, chol_lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_schur_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCHUR_ROW_LDS_MAX);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_chol_back, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCHUR_ROW_LDS_MAX);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_obj_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCHUR_ROW_LDS_MAX);
        if (e != hipSuccess) rc = qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
    }
    lap("allocations + staged uploads");
    if (!rc) rc = chol_chain_setup(p);
    lap("chain set-up");
    if (rc) {
        qsp_ba_destroy(p);
        return rc;
    }
    // default: the atomic-free Schur complement whenever its pair lists stay small (measured as fast as the atomic kernels
    // at C2 / C4 / C5); larger graphs use the block-row kernel with LDS atomics.  qsp_ba_set_deterministic overrides.
    if (count_pairs(p) <= PAIR_CAP && (int64_t)d.n_kf * d.n_kf <= ((int64_t)1 << 24)) {
        rc = build_pair_lists(p);
        if (rc) {
            qsp_ba_destroy(p);
            return rc;
        }
        p->deterministic = true;
    }
    lap("pair lists (host) + uploads");
    {   // every upload and memset above was asynchronous on the problem's stream: one wait, then the staging buffer goes back
        const int rc_f = upload_flush(p);
        const hipError_t e = rc_f ? hipErrorUnknown : hipStreamSynchronize(p->stream);
        lap("wait for the uploads");
        if (p->stage && !buf_cache_put(g_host_cache, p->device, p->stage, p->stage_alloc, hipHostMallocDefault, (size_t)64 << 20)) (void)hipHostFree(p->stage);
        p->stage = nullptr;
        p->stage_cap = p->stage_used = 0;
        if (e != hipSuccess) {
            qsp_ba_destroy(p);
            return qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
        }
    }
    *out = p;
    return QSP_OK;
}

extern "C" void qsp_ba_destroy(qsp_ba_problem* p) {
    if (p && p->stage) {      // (a create that failed half-way)
        (void)hipSetDevice(p->device);
        if (p->stream) (void)hipStreamSynchronize(p->stream);
        (void)hipHostFree(p->stage);
        p->stage = nullptr;
    }
    if (!p) return;
    (void)hipSetDevice(p->device);
    {
        StreamSet st;
        st.s = p->stream;
        stream_set_release(p->device, st);           // (synchronised and kept for the next problem on this device)
    }
    // (behind the synchronisation: nothing of this problem is in flight any more)
    for (size_t i = 0; i < p->allocs.size(); ++i)
        if (i >= p->alloc_bytes.size() || !p->alloc_bytes[i] || !buf_cache_put(g_dev_cache, p->device, p->allocs[i], p->alloc_bytes[i], 0, (size_t)512 << 20))
            (void)hipFree(p->allocs[i]);
    void* hb[3] = {p->scal_host, p->lvl_host, p->idx_host};
    const unsigned hf[3] = {hipHostMallocMapped | hipHostMallocCoherent, hipHostMallocDefault, hipHostMallocDefault};
    for (int i = 0; i < 3; ++i)
        if (hb[i] && !buf_cache_put(g_host_cache, p->device, hb[i], p->host_bytes[i], hf[i], (size_t)64 << 20)) (void)hipHostFree(hb[i]);
    delete p;
}

extern "C" void qsp_ba_release_caches(void) {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int dev = 0; dev < 64; ++dev) {
        std::vector<CachedBuf> dv, hv;
        std::vector<StreamSet> sv;
        {
            std::lock_guard<std::mutex> lk(g_buf_cache_mu);
            dv.swap(g_dev_cache[dev]);
            hv.swap(g_host_cache[dev]);
        }
        {
            std::lock_guard<std::mutex> lk(g_stream_pool_mu);
            sv.swap(g_stream_pool[dev]);
        }
        if (dv.empty() && hv.empty() && sv.empty()) continue;
        if (hipSetDevice(dev) != hipSuccess) continue;
        for (const CachedBuf& c : dv) (void)hipFree(c.p);
        for (const CachedBuf& c : hv) (void)hipHostFree(c.p);
        for (const StreamSet& st : sv) {
            if (st.s) (void)hipStreamDestroy(st.s);
        }
    }
    if (have_cur) (void)hipSetDevice(cur);
}

extern "C" int qsp_ba_set_levels(qsp_ba_problem* p, const uint8_t* mono, const uint8_t* stereo, const uint8_t* obj) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_set_levels: null problem");
    QSP_HIP(hipSetDevice(p->device));
    p->host_stale = false;                 // (every level is rewritten below, on both sides)
    p->tail_ready = false;
    std::fill(p->edge_level_h.begin(), p->edge_level_h.end(), 0);
    if (mono) for (int e = 0; e < p->n_mono; ++e) p->edge_level_h[p->mono_pos[e]] = mono[e] ? 1 : 0;
    if (stereo) for (int e = 0; e < p->n_stereo; ++e) p->edge_level_h[p->st_pos[e]] = stereo[e] ? 1 : 0;
    for (int e = 0; e < p->d.n_oe; ++e) p->oe_level_h[e] = (obj && obj[e]) ? 1 : 0;
    p->all_active = !mono && !stereo && !obj;
    if (p->all_active && p->world == 1) {     // everything active: no host array has to travel, and the device arrays are zeroed
        p->levels_reset_pending = true;       // by the next optimize() call's k_edge_index
        return QSP_OK;
    }
    p->levels_reset_pending = false;
    return upload_levels(p);
}

extern "C" int qsp_ba_set_state(qsp_ba_problem* p, const double* kf_pose, const double* pt_xyz, const double* obj_pose) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_set_state: null problem");
    QSP_HIP(hipSetDevice(p->device));
    if (kf_pose) QSP_HIP(hipMemcpy(p->d.kf_pose, kf_pose, sizeof(double) * 7 * p->d.n_kf, hipMemcpyHostToDevice));
    if (pt_xyz && p->d.n_pt) QSP_HIP(hipMemcpy(p->d.pt_xyz, pt_xyz, sizeof(double) * 3 * p->d.n_pt, hipMemcpyHostToDevice));
    if (obj_pose && p->d.n_obj) QSP_HIP(hipMemcpy(p->d.obj_pose, obj_pose, sizeof(double) * 7 * p->d.n_obj, hipMemcpyHostToDevice));
    return QSP_OK;
}

extern "C" int qsp_ba_get_state(qsp_ba_problem* p, double* kf_pose, double* pt_xyz, double* obj_pose) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_get_state: null problem");
    QSP_HIP(hipSetDevice(p->device));
    if (kf_pose) QSP_HIP(hipMemcpy(kf_pose, p->d.kf_pose, sizeof(double) * 7 * p->d.n_kf, hipMemcpyDeviceToHost));
    if (pt_xyz && p->d.n_pt) QSP_HIP(hipMemcpy(pt_xyz, p->d.pt_xyz, sizeof(double) * 3 * p->d.n_pt, hipMemcpyDeviceToHost));
    if (obj_pose && p->d.n_obj) QSP_HIP(hipMemcpy(obj_pose, p->d.obj_pose, sizeof(double) * 7 * p->d.n_obj, hipMemcpyDeviceToHost));
    return QSP_OK;
}

// SparseOptimizer::initializeOptimization(level) + buildIndexMapping (sparse_optimizer.cpp:199-267,166-190)
// vertices in id order (g2o's hessian order): the argsort is a property of the problem, computed once (qsp_ba_create)
static void build_orders(qsp_ba_problem* p) {
    const Dev& d = p->d;
    if (p->pose_order.size() == (size_t)(d.n_kf + d.n_obj) && p->pt_order.size() == (size_t)d.n_pt) return;
    std::vector<std::pair<int64_t, int>> v;
    for (int i = 0; i < d.n_kf; ++i) v.push_back({p->kf_id_h[i], i});
    for (int i = 0; i < d.n_obj; ++i) v.push_back({p->obj_id_h[i], d.n_kf + i});
    std::sort(v.begin(), v.end());
    p->pose_order.clear();
    for (const auto& q : v) p->pose_order.push_back(q.second);
    v.clear();
    for (int i = 0; i < d.n_pt; ++i) v.push_back({p->pt_id_h[i], i});
    std::sort(v.begin(), v.end());
    p->pt_order.clear();
    for (const auto& q : v) p->pt_order.push_back(q.second);
}

static void build_index(qsp_ba_problem* p) {
    const Dev& d = p->d;
    std::vector<uint8_t> ka(d.n_kf, 0), oa(d.n_obj, 0), pa(d.n_pt, 0);
    for (int e = 0; e < d.n_edge; ++e)
        if (!p->edge_level_h[e]) { ka[p->edge_h[e].kf] = 1; pa[p->edge_h[e].pt] = 1; }
    for (int e = 0; e < d.n_oe; ++e)
        if (!p->oe_level_h[e]) { ka[p->oe_kf_h[e]] = 1; oa[p->oe_obj_h[e]] = 1; }
    build_orders(p);
    p->kf_h.assign(d.n_kf, -1); p->obj_h.assign(d.n_obj, -1); p->pt_h.assign(d.n_pt, -1);
    int k = 0;
    for (int i : p->pose_order) {
        if (i < d.n_kf) { if (ka[i] && !p->kf_fixed_h[i]) p->kf_h[i] = k++; }
        else if (oa[i - d.n_kf]) p->obj_h[i - d.n_kf] = k++;
    }
    p->n_pose = k;
    k = 0;
    for (int i : p->pt_order)
        if (pa[i]) p->pt_h[i] = k++;
    p->n_land = k;
    // object elimination (k_obj_*): possible when every free key-frame precedes every object in the hessian order
    int n_kfree = 0, max_kf_h = -1, n_obj_act = 0;
    for (int i = 0; i < d.n_kf; ++i)
        if (p->kf_h[i] >= 0) { ++n_kfree; max_kf_h = std::max(max_kf_h, p->kf_h[i]); }
    for (int i = 0; i < d.n_obj; ++i) n_obj_act += p->obj_h[i] >= 0;
    p->elim = p->elim_allowed && n_obj_act > 0 && n_kfree > 0 && max_kf_h == n_kfree - 1;
    // k_obj_rows (the atomic mode's object update) holds a 6 x dimp row block in LDS (48 dimp bytes of the 160 KB): beyond 3413
    // dense unknowns (~568 free key-frames) it does not fit and the objects stay in the dense system.  The atomic-free mode
    // carries the object entries in its pair lists and has no such limit, but the mode can be switched between optimize() calls
    // (qsp_ba_set_deterministic), so the bound is applied to both.
    if (p->elim && sizeof(double) * (size_t)6 * (size_t)(((6 * n_kfree + NB - 1) / NB) * NB) > SCHUR_ROW_LDS_MAX) p->elim = false;
    p->n_dense = p->elim ? n_kfree : p->n_pose;
    p->dim_all = 6 * p->n_pose;
    p->dim = 6 * p->n_dense;
    p->dimp = ((p->dim + NB - 1) / NB) * NB;
}

// The scalars of a trial (chi2, rho denominator, max diagonal, failure flag) reach the host through pinned memory written by a
// one-wave kernel; the host waits on an event recorded right behind it.  (A device-to-host hipMemcpyAsync goes through the copy
// path and sat 41 us (C4) / 124 us (C5) idle in front of every read-back, profiles/r02_ba_timeline.txt.)  Two phases so that
// work can be enqueued between the publish and the wait.
static int publish_scal(qsp_ba_problem* p) {
    p->scal_seq += 1.0;
    hipLaunchKernelGGL(k_publish_scal, dim3(1), dim3(64), 0, p->stream, p->d, p->scal_host_dev, p->scal_seq);
    return QSP_OK;
}
// The publishing kernel writes the sequence number of the read-back last (system-scope fence in front of it); the host polls
// that word in coherent pinned memory and sees the scalars a few microseconds after the kernel, with no event object or
// signal packet in the stream.  (The ~10 us in front of the kernel that follows the publish are the system-scope release of a
// kernel that wrote host memory: they were there with an event record as well.)
static int wait_words(qsp_ba_problem* p, int base, double want, double* out4) {
    volatile double* seq = p->scal_host + base + 4;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spin = 0; *seq != want; ++spin) {
        // the word normally arrives within tens of microseconds; past ~0.5 ms of spinning (a slow peer rank in a sharded run, a
        // preempted process) the core is handed back between polls instead of being burnt for up to the timeout
        if (spin > 0xFFFF) std::this_thread::yield();
        if ((spin & 0xFFFF) == 0xFFFF) {     // a faulted or wedged kernel must not hang the caller: ask the runtime now and then
            const hipError_t e = hipStreamQuery(p->stream);
            if (e != hipSuccess && e != hipErrorNotReady) return qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
            if (e == hipSuccess && *seq != want)     // the stream has drained: the word is visible at the latest now
                return qsp_fail(QSP_ERR_DEVICE, "BA scalar read-back: sequence word never arrived");
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30))
                return qsp_fail(QSP_ERR_DEVICE, "BA scalar read-back timed out");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    for (int i = 0; i < 4; ++i) out4[i] = p->scal_host[base + i];
    return QSP_OK;
}
static int wait_scal(qsp_ba_problem* p, double* out4) { return wait_words(p, 0, p->scal_seq, out4); }
static int read_scal(qsp_ba_problem* p, double* out4) {
    int rc = publish_scal(p);
    return rc ? rc : wait_scal(p, out4);
}

// After a device-side stage boundary (k_tail_*) the host's copies of the levels and of the landmark numbering lag behind the
// device; whoever needs them (the next index build, qsp_ba_get_index) fetches the levels and rebuilds the tables -- host work
// that is off the path of the bundle adjustment that produced them.
static void build_index(qsp_ba_problem* p);
static int refresh_host_index(qsp_ba_problem* p) {
    if (!p->host_stale) return QSP_OK;
    const Dev& d = p->d;
    p->all_active = false;
    QSP_HIP(hipStreamSynchronize(p->stream));
    if (d.n_edge) QSP_HIP(hipMemcpy(p->edge_level_h.data(), d.edge_level, d.n_edge, hipMemcpyDeviceToHost));
    if (d.n_oe) QSP_HIP(hipMemcpy(p->oe_level_h.data(), d.oe_level, d.n_oe, hipMemcpyDeviceToHost));
    p->host_stale = false;
    build_index(p);
    return QSP_OK;
}

static int upload_levels(qsp_ba_problem* p) {
    // device level = caller's level OR "belongs to another rank"
    const Dev& d = p->d;
    std::vector<uint8_t> le(p->edge_level_h), lo(p->oe_level_h);
    if (p->world > 1) {
        for (int e = 0; e < d.n_edge; ++e) le[e] |= p->edge_foreign[e];
        for (int e = 0; e < d.n_oe; ++e) lo[e] |= p->oe_foreign[e];
    }
    if (d.n_edge) QSP_HIP(hipMemcpy(d.edge_level, le.data(), d.n_edge, hipMemcpyHostToDevice));
    if (d.n_oe) QSP_HIP(hipMemcpy(d.oe_level, lo.data(), d.n_oe, hipMemcpyHostToDevice));
    return QSP_OK;
}

// in-place SUM over ranks of `n` doubles at `buf` (device); no-op for world == 1
static int allreduce(qsp_ba_problem* p, double* buf, int64_t n) {
    if (p->world <= 1 || n <= 0) return QSP_OK;
    if (p->nccl) {     // RCCL on the library's stream: ordered against the producing / consuming kernels by the stream itself
        ncclResult_t r = rccl_api()->all_reduce(buf, buf, (size_t)n, ncclDouble, ncclSum, p->nccl, p->stream);
        if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce");
        return QSP_OK;
    }
    if (p->allreduce(p->allreduce_ctx, buf, n, (void*)p->stream) != 0) return qsp_fail(QSP_ERR_DEVICE, "all-reduce callback failed");
    return QSP_OK;
}

// SUM all-reduce of up to 3 scattered device ranges through the staging buffer
static int allreduce_gather(qsp_ba_problem* p, double* a, size_t na, double* b, size_t nb, double* c, size_t nc) {
    if (p->world <= 1) return QSP_OK;
    hipStream_t s = p->stream;
    const size_t n = na + nb + nc;
    if (n > p->comm_cap) return qsp_fail(QSP_ERR_INVALID, "all-reduce staging buffer too small");
    if (na) QSP_HIP(hipMemcpyAsync(p->comm, a, na * 8, hipMemcpyDeviceToDevice, s));
    if (nb) QSP_HIP(hipMemcpyAsync(p->comm + na, b, nb * 8, hipMemcpyDeviceToDevice, s));
    if (nc) QSP_HIP(hipMemcpyAsync(p->comm + na + nb, c, nc * 8, hipMemcpyDeviceToDevice, s));
    int rc = allreduce(p, p->comm, (int64_t)n);
    if (rc) return rc;
    if (na) QSP_HIP(hipMemcpyAsync(a, p->comm, na * 8, hipMemcpyDeviceToDevice, s));
    if (nb) QSP_HIP(hipMemcpyAsync(b, p->comm + na, nb * 8, hipMemcpyDeviceToDevice, s));
    if (nc) QSP_HIP(hipMemcpyAsync(c, p->comm + na + nb, nc * 8, hipMemcpyDeviceToDevice, s));
    return QSP_OK;
}

static int set_shard_common(qsp_ba_problem* p, int32_t rank, int32_t world) {
    QSP_HIP(hipSetDevice(p->device));
    const Dev& d = p->d;
    p->rank = rank; p->world = world;
    p->edge_foreign.assign(std::max(d.n_edge, 1), 0);
    p->oe_foreign.assign(std::max(d.n_oe, 1), 0);
    if (world > 1) {
        for (int e = 0; e < d.n_edge; ++e) p->edge_foreign[e] = (p->pt_id_h[p->edge_h[e].pt] % world) != rank;
        // the camera-object edges (n_obj x ~10, a few thousand at most) are NOT sharded: every rank linearises all of them
        // and holds the blocks the object elimination / back-substitution needs; only rank 0 counts their sums
        std::vector<uint8_t> ptf(std::max(d.n_pt, 1), 0);
        for (int i = 0; i < d.n_pt; ++i) ptf[i] = (p->pt_id_h[i] % world) != rank;
        const size_t need = std::max<size_t>((size_t)36 * (d.n_kf + d.n_obj) + p->dimp_max + 64,
                                             std::max<size_t>((size_t)3 * d.n_pt + 8, (size_t)d.n_edge + d.n_oe + 8));
        if (need > p->comm_cap) {
            void* q = nullptr;
            QSP_HIP(hipMalloc(&q, need * sizeof(double)));
            p->allocs.push_back(q);
            p->alloc_bytes.push_back(0);             // (not a pool chunk: freed, not cached)
            p->comm = (double*)q;
            p->comm_cap = need;
        }
        if (!p->d_pt_foreign) {
            int rc = dalloc(p, &p->d_pt_foreign, (size_t)std::max(d.n_pt, 1));
            if (!rc) rc = dalloc(p, &p->d_edge_foreign, (size_t)std::max(d.n_edge, 1));
            if (!rc) rc = dalloc(p, &p->d_oe_foreign, (size_t)std::max(d.n_oe, 1));
            if (rc) return rc;
        }
        QSP_HIP(hipMemcpy(p->d_pt_foreign, ptf.data(), ptf.size(), hipMemcpyHostToDevice));
        QSP_HIP(hipMemcpy(p->d_edge_foreign, p->edge_foreign.data(), p->edge_foreign.size(), hipMemcpyHostToDevice));
        QSP_HIP(hipMemcpy(p->d_oe_foreign, p->oe_foreign.data(), p->oe_foreign.size(), hipMemcpyHostToDevice));
    }
    return upload_levels(p);
}

extern "C" int qsp_ba_set_shard(qsp_ba_problem* p, int32_t rank, int32_t world, qsp_allreduce_fn fn, void* ctx) {
    if (!p || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn))
        return qsp_fail(QSP_ERR_INVALID, "qsp_ba_set_shard: bad argument");
    p->allreduce = fn; p->allreduce_ctx = ctx; p->nccl = nullptr;
    return set_shard_common(p, rank, world);
}

extern "C" int qsp_ba_set_shard_rccl(qsp_ba_problem* p, int32_t rank, int32_t world, void* nccl_comm) {
    if (!p || world < 1 || rank < 0 || rank >= world || (world > 1 && !nccl_comm))
        return qsp_fail(QSP_ERR_INVALID, "qsp_ba_set_shard_rccl: bad argument");
    if (world > 1 && !rccl_api()) return qsp_fail(QSP_ERR_DEVICE, "qsp_ba_set_shard_rccl: librccl.so.1 could not be resolved");
    p->allreduce = nullptr; p->allreduce_ctx = nullptr; p->nccl = world > 1 ? (ncclComm_t)nccl_comm : nullptr;
    return set_shard_common(p, rank, world);
}

// zero the landmark / object-edge entries this rank does not own, then SUM over ranks: every rank ends with all of them
__global__ void k_mask_foreign_points(Dev d, const uint8_t* foreign) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n_pt && foreign[i]) { d.pt_xyz[3 * i] = 0; d.pt_xyz[3 * i + 1] = 0; d.pt_xyz[3 * i + 2] = 0; }
}
__global__ void k_mask_foreign_chi2(Dev d, const uint8_t* ef, const uint8_t* of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n_edge && ef[i]) d.edge_chi2[i] = 0;
    if (i < d.n_oe && of[i]) d.oe_chi2[i] = 0;
}

static int launch_errors(qsp_ba_problem* p, const Par& par) {
    const Dev& d = p->d;
    const int n_tot = d.n_edge + d.n_oe;
    const int grid = std::max(1, std::min(p->n_partial, (n_tot + 255) / 256));
    hipLaunchKernelGGL(k_errors, dim3(grid), dim3(256), 0, p->stream, d, par);
    hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(64), 0, p->stream, d, grid, 0);
    return QSP_OK;
}

// chi2 (scal[0]) and the rho denominator (scal[1]) summed over ranks
static int reduce_scalars(qsp_ba_problem* p) { return allreduce_gather(p, p->d.scal, 2, nullptr, 0, nullptr, 0); }

// want_tail: the call is the first stage of a local bundle adjustment -- its last iteration also enqueues the stage boundary
// (k_tail_*) and leaves p->tail_ready.  head: the call is the second stage behind a boundary that was committed -- levels, index
// tables, first chi2, first system and lambda's seed are on the device already.
struct StageHead { double chi2, maxdiag; };
static int ba_optimize(qsp_ba_problem* p, int32_t n_iter, double delta_mono, double delta_stereo, double delta_obj,
                       const volatile uint8_t* stop_flag, qsp_ba_trace* tr, bool want_tail, const StageHead* head) {
    QSP_HIP(hipSetDevice(p->device));
    Dev& d = p->d;
    hipStream_t s = p->stream;
    p->tail_ready = false;
    if (!head) {
        const int rc0 = refresh_host_index(p);
        if (rc0) return rc0;
        auto& ic = p->idx_cache;
        if (p->all_active && p->world == 1 && ic.valid && ic.elim_allowed == p->elim_allowed) {
            p->kf_h = ic.kf_h; p->obj_h = ic.obj_h; p->pt_h = ic.pt_h;
            p->n_pose = ic.n_pose; p->n_land = ic.n_land; p->dim = ic.dim; p->dimp = ic.dimp; p->n_dense = ic.n_dense;
            p->dim_all = ic.dim_all; p->elim = ic.elim;
        } else {
            build_index(p);
            if (p->all_active && p->world == 1) {
                ic.kf_h = p->kf_h; ic.obj_h = p->obj_h; ic.pt_h = p->pt_h;
                ic.n_pose = p->n_pose; ic.n_land = p->n_land; ic.dim = p->dim; ic.dimp = p->dimp; ic.n_dense = p->n_dense;
                ic.dim_all = p->dim_all; ic.elim = p->elim; ic.elim_allowed = p->elim_allowed;
                ic.valid = true;
            }
        }
    }
    if (!head) {   // hessian indices: through pinned staging (a pageable source makes the "async" copy a synchronous staged one), and only
        // the arrays that differ from what the device already holds (the second round of a local BA rarely changes them)
        const size_t nk = d.n_kf, no = d.n_obj, np_ = d.n_pt;
        if (p->idx_uploaded.size() != nk + no + np_) p->idx_uploaded.assign(nk + no + np_, INT32_MIN);
        QSP_HIP(hipStreamSynchronize(s));              // (the staging buffer of the previous call has been consumed)
        struct Part { const int32_t* src; int32_t* dst; size_t off, n; } parts[3] = {
            {p->kf_h.data(), d.kf_h, 0, nk}, {p->obj_h.data(), d.obj_h, nk, no}, {p->pt_h.data(), d.pt_h, nk + no, np_}};
        for (const Part& q : parts) {
            if (!q.n || !memcmp(q.src, p->idx_uploaded.data() + q.off, sizeof(int32_t) * q.n)) continue;
            memcpy(p->idx_host + q.off, q.src, sizeof(int32_t) * q.n);
            memcpy(p->idx_uploaded.data() + q.off, q.src, sizeof(int32_t) * q.n);
            QSP_HIP(hipMemcpyAsync(q.dst, p->idx_host + q.off, sizeof(int32_t) * q.n, hipMemcpyHostToDevice, s));
        }
    }
    if (sizeof(double) * (size_t)(2 * p->dimp + NB) > SCHUR_ROW_LDS_MAX)     // k_chol_back keeps y and x in LDS (dimp <= 10 208):
        return qsp_fail(QSP_ERR_UNSUPPORTED, "qsp_ba_optimize: reduced camera system too large for the dense solver");   // refused before anything is enqueued
    if (!head && (d.n_edge || d.n_oe)) {
        hipLaunchKernelGGL(k_edge_index, dim3((std::max(d.n_edge, d.n_oe) + 255) / 256), dim3(256), 0, s, d, p->levels_reset_pending ? 1 : 0);
        p->levels_reset_pending = false;
    }
    Par par{delta_mono, delta_stereo, delta_obj, 0.0, p->dim, p->dimp, p->n_pose, p->rank == 0 ? 1 : 0,
            (p->deterministic && p->dimp > 0) ? 1 : 0, p->n_dense, p->elim ? 1 : 0};
    if (tr) { tr->n = 0; tr->result = 0; tr->n_pose_blocks = p->n_pose; tr->n_landmarks = p->n_land; }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (p->profiling) {
        memset(&p->prof, 0, sizeof(p->prof));
        (void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
        (void)hipEventRecord(ev0, s);
    }
    const int gp = std::max(1, std::min(p->n_partial, (d.n_pt + 255) / 256));
    d.bs = d.Hs + (size_t)p->dimp * p->dimp;      // right behind the reduced matrix: one all-reduce covers both
    double lambda = 0, ni = 2, currentChi = 0;
    int nBad = 0, done = 0, result = 0;
    double sc[4];
    // buildSystem of the current estimates, enqueued.  (every entry of Hdiag and bp that is read is written by
    // k_lin_poses_finish / k_lin_objects: no memsets.)  Profiling: one event pair per call, read at the end.
    std::vector<hipEvent_t> lin_ev;
    auto enqueue_lin = [&](bool with_chi2) -> int {
        if (p->profiling) {
            hipEvent_t a = nullptr;
            (void)hipEventCreate(&a);
            (void)hipEventRecord(a, s);
            lin_ev.push_back(a);
        }
        const int nb_e = d.n_ksplit + d.n_chunk + (d.n_oe + 3) / 4;
        if (nb_e) hipLaunchKernelGGL(k_lin_edges, dim3(nb_e), dim3(256), 0, s, d, par, d.n_ksplit, d.n_chunk);
        hipLaunchKernelGGL(k_lin_vertices, dim3(d.n_kf + d.n_obj), dim3(64), 0, s, d, par);
        if (p->profiling) {
            hipEvent_t b = nullptr;
            (void)hipEventCreate(&b);
            (void)hipEventRecord(b, s);
            lin_ev.push_back(b);
        }
        return allreduce_gather(p, d.Hdiag, (size_t)36 * p->n_pose, d.bp, (size_t)p->dim_all, d.scal, with_chi2 ? 1 : 0);   // pose blocks, b_p, chi2
    };
    // The stage boundary, speculated behind the last iteration's trial (k_tail_*): classification into the second level arrays, new
    // landmark numbering, then the second stage's first chi2 / system / lambda seed on those buffers.
    const bool tail_possible = want_tail && p->world == 1 && p->pt_order_uploaded && (d.n_edge + d.n_oe) > 0;
    auto enqueue_tail = [&]() -> int {
        ++p->act_epoch;
        const int n = std::max(std::max(d.n_edge, d.n_oe), 1);
        hipLaunchKernelGGL(k_tail_classify, dim3((n + 255) / 256), dim3(256), 0, s, d, p->edge_level2, p->oe_level2, p->edge_ha2, p->act,
                           p->act_epoch, 5.991, 7.815, 1e3);
        hipLaunchKernelGGL(k_tail_reindex, dim3(1), dim3(1024), 0, s, d, (const int32_t*)p->act, p->act_epoch, p->pt_h2);
        Dev d2 = d;
        d2.edge_level = p->edge_level2; d2.oe_level = p->oe_level2; d2.edge_ha = p->edge_ha2; d2.pt_h = p->pt_h2;
        Par par2 = par;
        par2.delta_mono = par2.delta_stereo = par2.delta_obj = 0.0;      // robust kernels dropped, src/Optimizer_util.cc:628,643,655
        par2.lambda = 0.0;
        const int n_tot = d.n_edge + d.n_oe;
        const int grid = std::max(1, std::min(p->n_partial, (n_tot + 255) / 256));
        hipLaunchKernelGGL(k_errors, dim3(grid), dim3(256), 0, s, d2, par2);
        const int nb_e = d.n_ksplit + d.n_chunk + (d.n_oe + 3) / 4;
        if (nb_e) hipLaunchKernelGGL(k_lin_edges, dim3(nb_e), dim3(256), 0, s, d2, par2, d.n_ksplit, d.n_chunk);
        hipLaunchKernelGGL(k_lin_vertices, dim3(d.n_kf + d.n_obj), dim3(64), 0, s, d2, par2);
        const int gm = std::max(1, std::min(256, (d.n_pt * 3 + p->n_pose * 6 + 255) / 256));
        hipLaunchKernelGGL(k_maxdiag, dim3(gm), dim3(256), 0, s, d2, par2);
        p->tail_seq += 1.0;
        hipLaunchKernelGGL(k_tail_publish, dim3(1), dim3(64), 0, s, d2, grid, p->scal_host_dev + 8, p->tail_seq);
        return QSP_OK;
    };
    bool tail_ok = false;        // the last trial of the call was accepted with the stage boundary enqueued behind it
    bool lin_ready = false;      // the system of the current estimates is already enqueued (speculated behind an accepted trial)
    for (int it = 0; it < n_iter; ++it) {
        if (stop_flag && *stop_flag) { result = 2; break; }
        // computeActiveErrors + chi2 (sparse_optimizer.cpp:61-114).  After the first iteration the state is the one the last
        // accepted trial was evaluated on -- the same kernel on the same numbers -- so its chi2 is carried over instead of
        // being recomputed and read back (an iteration never starts after a rejected trial: that ends the call).
        if (it == 0 && !head) launch_errors(p, par);
        int rc = QSP_OK;
        if (!lin_ready && !(it == 0 && head)) rc = enqueue_lin(it == 0);
        if (rc) return rc;
        lin_ready = false;
        if (it == 0 && head) {      // chi2, system and lambda's seed of these estimates and levels are there (k_tail_*)
            sc[0] = head->chi2;
            sc[2] = head->maxdiag;
            currentChi = sc[0];
        } else if (it == 0) {
            const int gm = std::max(1, std::min(256, (d.n_pt * 3 + p->n_pose * 6 + 255) / 256));      // (scal[2] was zeroed by k_lin_vertices)
            hipLaunchKernelGGL(k_maxdiag, dim3(gm), dim3(256), 0, s, d, par);
            if (p->world > 1 && p->nccl) {   // max over ranks of the local maxima
                ncclResult_t r = rccl_api()->all_reduce(d.scal + 2, d.scal + 2, 1, ncclDouble, ncclMax, p->nccl, s);
                if (r != ncclSuccess) return rccl_fail(r, "ncclAllReduce(max)");
            }
            rc = read_scal(p, sc);               // (also arms scal[3] for the first trial)
            if (rc) return rc;
            if (p->world > 1 && !p->nccl) {   // callback hook (SUM only): max as a SUM over one-hot slots
                std::vector<double> slots(p->world, 0.0);
                slots[p->rank] = sc[2];
                QSP_HIP(hipMemcpyAsync(p->comm, slots.data(), 8 * p->world, hipMemcpyHostToDevice, s));
                rc = allreduce(p, p->comm, p->world);
                if (rc) return rc;
                QSP_HIP(hipMemcpyAsync(slots.data(), p->comm, 8 * p->world, hipMemcpyDeviceToHost, s));
                QSP_HIP(hipStreamSynchronize(s));
                for (double v : slots) sc[2] = std::max(sc[2], v);
            }
            currentChi = sc[0];
        }
        const double iniChi = currentChi;
        if (it == 0) { lambda = 1e-5 * sc[2]; ni = 2; nBad = 0; }
        double rho = 0;
        int qmax = 0, accepted = 0;
        do {
            par.lambda = lambda;
            // (scal[3], the trial's failure flag, was re-armed by the k_publish_scal of the previous read-back)
            const bool fused = p->deterministic && p->dimp > 0 && d.n_pk > 0;
            if (!fused) {   // push (g2o: sparse_optimizer.cpp:519-527); the fused path backs up inside k_trial_stage1
                QSP_HIP(hipMemcpyAsync(d.kf_bk, d.kf_pose, sizeof(double) * 7 * d.n_kf, hipMemcpyDeviceToDevice, s));
                if (d.n_obj) QSP_HIP(hipMemcpyAsync(d.obj_bk, d.obj_pose, sizeof(double) * 7 * d.n_obj, hipMemcpyDeviceToDevice, s));
                if (d.n_pt) QSP_HIP(hipMemcpyAsync(d.pt_bk, d.pt_xyz, sizeof(double) * 3 * d.n_pt, hipMemcpyDeviceToDevice, s));
            }
            // solve
            if (p->dimp > 0) {
                const int nprep = p->n_dense * 36 + d.n_oe * 36 + (p->dimp - p->dim);
                const size_t row_lds = sizeof(double) * ((size_t)6 * p->dimp + 6);
                if (fused) {
                    // atomic-free path, two launches: everything that is independent, then one workgroup per key-frame pair
                    Stage1 g;
                    g.nb_hs = (int)(((size_t)p->dimp * p->dimp + p->dimp + 255) / 256);
                    g.nb_pt = (d.n_pt + 255) / 256;
                    g.nb_edge = (d.n_edge + 255) / 256;
                    g.nb_obj = p->elim ? (d.n_obj + 3) / 4 : 0;
                    g.nb_bk = (7 * d.n_kf + 7 * d.n_obj + 3 * d.n_pt + 255) / 256;
                    hipLaunchKernelGGL(k_trial_stage1, dim3(g.nb_hs + g.nb_pt + g.nb_edge + g.nb_obj + g.nb_bk), dim3(256), 0, s, d, par, g);
                    if (!p->elim && d.n_oe)     // objects inside the dense system: their off-diagonal blocks
                        hipLaunchKernelGGL(k_schur_prepare, dim3((std::max(nprep, p->dimp) + 255) / 256), dim3(256), 0, s, d, par);
                    hipLaunchKernelGGL(k_schur_pairs, dim3(d.n_pk_big + (d.n_pk - d.n_pk_big + 3) / 4), dim3(256), 0, s, d, par);
                } else {
                    QSP_HIP(hipMemsetAsync(d.Hs, 0, sizeof(double) * (size_t)p->dimp * p->dimp, s));
                    hipLaunchKernelGGL(k_schur_prepare, dim3((std::max(nprep, p->dimp) + 255) / 256), dim3(256), 0, s, d, par);
                    // block rows pay off once the per-landmark kernel's global atomics collide or scatter (measured: C5 435 ->
                    // 157 us, 2 M edges 5.4 -> 0.8 ms); below ~64 k edges both are latency-bound and the single launch wins
                    if (d.n_pt && row_lds <= SCHUR_ROW_LDS_MAX && d.n_ksplit && d.n_edge >= SCHUR_ROWS_MIN_EDGES) {
                        hipLaunchKernelGGL(k_schur_dinv, dim3((d.n_pt + 255) / 256), dim3(256), 0, s, d, par);
                        hipLaunchKernelGGL(k_schur_rows, dim3(d.n_ksplit), dim3(256), row_lds, s, d, par);
                    } else if (d.n_pt) {
                        hipLaunchKernelGGL(k_schur_points, dim3((d.n_pt + 3) / 4), dim3(256), 0, s, d, par);
                    }
                    if (p->elim) {
                        hipLaunchKernelGGL(k_obj_prepare, dim3(d.n_obj), dim3(64), 0, s, d, par);
                        hipLaunchKernelGGL(k_obj_rows, dim3(d.n_kf), dim3(64), sizeof(double) * (size_t)6 * p->dimp, s, d, par);
                        QSP_HIP(hipGetLastError());      // (a rejected launch here would leave Hs without the object terms)
                    }
                }
                rc = allreduce(p, d.Hs, (int64_t)p->dimp * p->dimp + p->dimp);     // reduced matrix + right-hand side
                if (rc) return rc;
                const int nb = p->dimp / NB;
                const size_t lds = sizeof(double) * CHOL_LDS_DOUBLES;
                if (p->chol_chain && nb >= 2) {
                    // one launch: the first workgroup to start is the chain, the others take the tiles in ticket order (k_chol_solve)
                    if (++p->chol_epoch == 0) ++p->chol_epoch;
                    unsigned* flag_w = p->chol_flags;
                    unsigned* tile_done = p->chol_flags + p->dimp_max / NB;
                    const unsigned n_tiles = (nb >= 3 && !p->chol_fault) ? (unsigned)(nb * (nb - 1) / 2) : 0u;
                    const unsigned grid = 1u + std::min(n_tiles, (unsigned)p->chol_grid_max);
                    hipLaunchKernelGGL(k_chol_solve, dim3(grid), dim3(CHOL_THREADS), sizeof(double) * CHOL_SOLVE_LDS_DOUBLES, s, d.Hs, d.Uf, d.Winv,
                                       d.bs, d.ych, p->dimp, nb, d.scal, flag_w, tile_done, p->chol_epoch, p->chol_ticket, p->chol_ticket_base,
                                       n_tiles);
                    p->chol_ticket_base += 1u + n_tiles + (grid - 1u) * (n_tiles ? 1u : 0u);      // what the counter holds when the launch has drained
                    // (when the chain has ended every tile workgroup's writes are complete -- see chol_chain_body)
                } else {
                    hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(CHOL_THREADS), lds, s, d.Hs, d.Winv, d.bs, d.ych, p->dimp, d.scal);
                    for (int k = 0; k + 1 < nb; ++k)
                        hipLaunchKernelGGL(k_chol_step, dim3(nb - k - 1, nb - k - 1), dim3(CHOL_THREADS), lds, s, d.Hs, d.Uf, d.Winv, d.bs,
                                           d.ych, p->dimp, k, d.scal);
                }
                {   // the partial sums of a step's units beside y and x in LDS when they fit (k_chol_back)
                    const size_t lds_split = sizeof(double) * (size_t)(10 * p->dimp + NB);
                    const int split = lds_split <= SCHUR_ROW_LDS_MAX ? 1 : 0;
                    hipLaunchKernelGGL(k_chol_back, dim3(1 + 8 * CHOL_BACK_HELPERS), dim3(1024),
                                       split ? lds_split : sizeof(double) * (size_t)(2 * p->dimp + NB), s, d, par, d.Uf, d.Winv, d.ych, d.xp, p->dimp, split);
                }
            } else if (d.n_pt) {
                if (fused) { /* unreachable: fused needs dimp > 0 */ }
                hipLaunchKernelGGL(k_schur_dinv, dim3((d.n_pt + 255) / 256), dim3(256), 0, s, d, par);   // only D^-1 is needed
            }
            // update (oplus) + rho denominator
            if (p->world == 1) {     // one rank: update in one launch; chi2 sum, rho denominator and publish in one launch
                // (one launch only with the 6x3 blocks of this build in memory: recomputed blocks read the poses the pose
                //  block of the same launch is rewriting)
                const bool one = par.have_hpl != 0;
                if (one) {
                    hipLaunchKernelGGL(k_update_all, dim3(gp + 1), dim3(256), 0, s, d, par, gp, p->n_partial);
                } else {
                    hipLaunchKernelGGL(k_update_points, dim3(gp), dim3(256), 0, s, d, par);
                    hipLaunchKernelGGL(k_update_poses, dim3(1), dim3(256), 0, s, d, par, gp);
                }
                const int n_tot = d.n_edge + d.n_oe;
                const int grid = std::max(1, std::min(p->n_partial, (n_tot + 255) / 256));
                hipLaunchKernelGGL(k_errors, dim3(grid), dim3(256), 0, s, d, par);
                p->scal_seq += 1.0;
                hipLaunchKernelGGL(k_finish_sum_publish, dim3(1), dim3(64), 0, s, d, grid, p->scal_host_dev, one ? gp : -1, p->n_partial,
                                   p->scal_seq);
            } else {
                hipLaunchKernelGGL(k_update_points, dim3(gp), dim3(256), 0, s, d, par);
                hipLaunchKernelGGL(k_update_poses, dim3(1), dim3(256), 0, s, d, par, gp);
                launch_errors(p, par);
                rc = reduce_scalars(p);
                if (rc) return rc;
                rc = publish_scal(p);
                if (rc) return rc;
            }
            // A trial is accepted far more often than not, and the next iteration then linearises exactly the estimates the
            // device holds now: enqueue that system BEHIND the publish and only then wait for the verdict -- the read-back and
            // the host's turn-around hide behind ~50 us of device work.  If the trial is rejected the speculated system (of the
            // estimates about to be rolled back) is rebuilt after the restore.
            const bool spec = p->speculate && it + 1 < n_iter;
            const bool tail = tail_possible && it + 1 == n_iter;      // (spec is false then: both overwrite the system's buffers)
            if (spec) {
                rc = enqueue_lin(false);
                if (rc) return rc;
            }
            if (tail) {
                rc = enqueue_tail();
                if (rc) return rc;
            }
            tail_ok = false;
            rc = wait_scal(p, sc);
            if (rc) return rc;
            if (sc[3] == 2.0) {
                // A flag wait of the chain factorisation expired (its workgroups were kept from running together for ~1-2 s: another
                // process or thread holding the compute units).  Not an error of the problem: the one-launch-per-step form performs
                // the same operations in the same order.  Clear the sticky flag, keep that form for this problem from here on,
                // restore the estimates the trial started from and repeat the trial -- same lambda, not counted.
                if (!p->chol_chain) return qsp_fail(QSP_ERR_DEVICE, "BA: the factorisation reported an expired flag wait on the one-launch-per-step form");
                p->chol_chain = false;
                p->chain_timeouts++;
                p->prof.chain_timeouts = p->chain_timeouts;
                fprintf(stderr, "[qsp_hip] BA: a flag wait of the chain factorisation expired (#%d); this problem continues on the "
                                "one-launch-per-step form\n", p->chain_timeouts);
                QSP_HIP(hipMemsetAsync(d.scal + 5, 0, sizeof(double), s));
                QSP_HIP(hipMemcpyAsync(d.kf_pose, d.kf_bk, sizeof(double) * 7 * d.n_kf, hipMemcpyDeviceToDevice, s));
                if (d.n_obj) QSP_HIP(hipMemcpyAsync(d.obj_pose, d.obj_bk, sizeof(double) * 7 * d.n_obj, hipMemcpyDeviceToDevice, s));
                if (d.n_pt) QSP_HIP(hipMemcpyAsync(d.pt_xyz, d.pt_bk, sizeof(double) * 3 * d.n_pt, hipMemcpyDeviceToDevice, s));
                if (spec || tail) {      // the speculated system overwrote this iteration's
                    rc = enqueue_lin(false);
                    if (rc) return rc;
                }
                rho = -1.0;      // (stay in the trial loop)
                continue;
            }
            const bool ok2 = sc[3] == 0.0;
            double tempChi = ok2 ? sc[0] : DBL_MAX;
            rho = currentChi - tempChi;
            const double scale = sc[1] + 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = std::min(alpha, 2. / 3.);
                lambda *= std::max(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                accepted = 1;
                lin_ready = spec;
                tail_ok = tail;
            } else {
                lambda *= ni;
                ni *= 2;
                QSP_HIP(hipMemcpyAsync(d.kf_pose, d.kf_bk, sizeof(double) * 7 * d.n_kf, hipMemcpyDeviceToDevice, s));
                if (d.n_obj) QSP_HIP(hipMemcpyAsync(d.obj_pose, d.obj_bk, sizeof(double) * 7 * d.n_obj, hipMemcpyDeviceToDevice, s));
                if (d.n_pt) QSP_HIP(hipMemcpyAsync(d.pt_xyz, d.pt_bk, sizeof(double) * 3 * d.n_pt, hipMemcpyDeviceToDevice, s));
                accepted = 0;
                if (spec || tail) {      // the speculated system overwrote this iteration's: rebuild it for the restored estimates
                    rc = enqueue_lin(false);
                    if (rc) return rc;
                }
            }
            qmax++;
            p->prof.n_trials++;
        } while (rho < 0 && qmax < 10 && !(stop_flag && *stop_flag));
        done++;
        if (tr && tr->n < tr->cap) {
            tr->chi2[tr->n] = currentChi; tr->lambda[tr->n] = lambda; tr->trials[tr->n] = qmax;
            tr->accepted[tr->n] = accepted; tr->n++;
        }
        if (qmax == 10 || rho == 0) { result = 1; break; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { result = 1; break; }
    }
    if (p->world > 1 && d.n_pt) {   // every rank leaves with all landmarks: own values, zeros elsewhere, SUM
        hipLaunchKernelGGL(k_mask_foreign_points, dim3((d.n_pt + 255) / 256), dim3(256), 0, s, d, p->d_pt_foreign);
        int rc2 = allreduce(p, d.pt_xyz, (int64_t)3 * d.n_pt);
        if (rc2) return rc2;
    }
    QSP_HIP(hipStreamSynchronize(s));
    if (p->profiling) {
        (void)hipEventRecord(ev1, s);
        hipEventSynchronize(ev1);
        (void)hipEventElapsedTime(&p->prof.ms_total, ev0, ev1);
        (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
        for (size_t i = 0; i + 1 < lin_ev.size(); i += 2) {      // every system built, the speculated ones included
            float ms = 0;
            (void)hipEventElapsedTime(&ms, lin_ev[i], lin_ev[i + 1]);
            p->prof.ms_linearize += ms;
            p->prof.n_linearize++;
        }
        for (hipEvent_t e : lin_ev) (void)hipEventDestroy(e);
        // algorithmic bytes of one linearisation
        (void)refresh_host_index(p);      // (profiling only: the level counts come from the host's copies)
        int64_t nm = 0, ns = 0, no = 0;
        for (int e = 0; e < d.n_edge; ++e)
            if (!p->edge_level_h[e]) (p->edge_h[e].stereo ? ns : nm)++;
        for (int e = 0; e < d.n_oe; ++e) if (!p->oe_level_h[e]) no++;
        p->prof.bytes_linearize = 176 * nm + 184 * ns + 392 * (int64_t)p->n_pose + 120 * (int64_t)p->n_land + 352 * no;
    }
    if (tr) tr->result = result;
    QSP_HIP(hipGetLastError());
    if (tr) tr->iterations = done;
    p->tail_ready = tail_ok;
    return QSP_OK;
}

extern "C" int qsp_ba_optimize(qsp_ba_problem* p, int32_t n_iter, double delta_mono, double delta_stereo,
                               double delta_obj, const volatile uint8_t* stop_flag, qsp_ba_trace* tr) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_optimize: null problem");
    return ba_optimize(p, n_iter, delta_mono, delta_stereo, delta_obj, stop_flag, tr, false, nullptr);
}

extern "C" int qsp_ba_get_index(qsp_ba_problem* p, int32_t* kf_hidx, int32_t* obj_hidx, int32_t* pt_hidx) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_get_index: null problem");
    QSP_HIP(hipSetDevice(p->device));
    {
        const int rc = refresh_host_index(p);
        if (rc) return rc;
    }
    if (kf_hidx) memcpy(kf_hidx, p->kf_h.data(), sizeof(int32_t) * p->kf_h.size());
    if (obj_hidx) memcpy(obj_hidx, p->obj_h.data(), sizeof(int32_t) * p->obj_h.size());
    if (pt_hidx) memcpy(pt_hidx, p->pt_h.data(), sizeof(int32_t) * p->pt_h.size());
    return QSP_OK;
}

extern "C" int qsp_ba_get_edges(qsp_ba_problem* p, double* mono_chi2, double* stereo_chi2, double* obj_chi2,
                                uint8_t* mono_depth_pos, uint8_t* stereo_depth_pos) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_get_edges: null problem");
    QSP_HIP(hipSetDevice(p->device));
    const Dev& d = p->d;
    if (p->world > 1) {
        hipLaunchKernelGGL(k_mask_foreign_chi2, dim3((std::max(std::max(d.n_edge, d.n_oe), 1) + 255) / 256), dim3(256), 0, p->stream, d,
                           p->d_edge_foreign, p->d_oe_foreign);
        int rc = allreduce_gather(p, d.edge_chi2, (size_t)d.n_edge, nullptr, 0, nullptr, 0);   // (object edges: replicated)
        if (rc) return rc;
        QSP_HIP(hipStreamSynchronize(p->stream));
    }
    std::vector<double> c(std::max(d.n_edge, 1));
    std::vector<uint8_t> pos(std::max(d.n_edge, 1));
    if (d.n_edge) {
        QSP_HIP(hipMemcpy(c.data(), d.edge_chi2, sizeof(double) * d.n_edge, hipMemcpyDeviceToHost));
        if (mono_depth_pos || stereo_depth_pos) {
            uint8_t* dp = nullptr;
            QSP_HIP(hipMalloc((void**)&dp, d.n_edge));
            hipLaunchKernelGGL(k_depth_positive, dim3((d.n_edge + 255) / 256), dim3(256), 0, p->stream, d, dp);
            QSP_HIP(hipMemcpyAsync(pos.data(), dp, d.n_edge, hipMemcpyDeviceToHost, p->stream));
            QSP_HIP(hipStreamSynchronize(p->stream));
            (void)hipFree(dp);
        }
    }
    for (int e = 0; e < p->n_mono; ++e) {
        if (mono_chi2) mono_chi2[e] = c[p->mono_pos[e]];
        if (mono_depth_pos) mono_depth_pos[e] = pos[p->mono_pos[e]];
    }
    for (int e = 0; e < p->n_stereo; ++e) {
        if (stereo_chi2) stereo_chi2[e] = c[p->st_pos[e]];
        if (stereo_depth_pos) stereo_depth_pos[e] = pos[p->st_pos[e]];
    }
    if (obj_chi2 && d.n_oe) QSP_HIP(hipMemcpy(obj_chi2, d.oe_chi2, sizeof(double) * d.n_oe, hipMemcpyDeviceToHost));
    return QSP_OK;
}

extern "C" int qsp_ba_profile(qsp_ba_problem* p, int enable, qsp_ba_stats* out) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_profile: null problem");
    p->profiling = enable != 0;
    p->prof.cholesky_chain = p->chol_chain ? 1 : 0;
    p->prof.chain_timeouts = p->chain_timeouts;
    p->prof.boundary_device = p->n_boundary_device;
    p->prof.boundary_host = p->n_boundary_host;
    if (out) *out = p->prof;
    return QSP_OK;
}

// Optimizer::LocalJointBundleAdjustment schedule (src/Optimizer_util.cc:598-661)
extern "C" int qsp_ba_local_joint(qsp_ba_problem* p, const volatile uint8_t* stop_flag, qsp_ba_trace* t1, qsp_ba_trace* t2) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_local_joint: null problem");
    const float thMono = sqrtf(5.991f), thStereo = sqrtf(7.815f), thObj = sqrtf(1e3f);
    if (stop_flag && *stop_flag) return QSP_OK;                                   // :589-596
    static const bool tm = getenv("QSP_BA_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int rc = qsp_ba_set_levels(p, nullptr, nullptr, nullptr);
    const double t1s = now();
    static const bool no_tail = getenv("QSP_BA_HOST_BOUNDARY") != nullptr;      // (A/B: the host-side stage boundary of round 3)
    if (!rc) rc = ba_optimize(p, 5, (double)(float)sqrt(5.991), (double)(float)sqrt(7.815), (double)thObj, stop_flag, t1, !no_tail, nullptr);
    (void)thMono; (void)thStereo;
    if (rc) return rc;
    const double t2s = now();
    if (stop_flag && *stop_flag) return QSP_OK;                                   // :603-610 (no write-back by the caller)
    Dev& d = p->d;
    double t3s = t2s;
    if (p->tail_ready) {
        // The stage boundary ran on the device behind the first stage's last trial (k_tail_*): its four numbers are in (or on their
        // way to) the second pinned slot.  No pose vertex changed its activity -> the second level arrays, landmark numbering and
        // edge table become the current ones by a pointer swap, and the second stage starts at its first trial.
        double th[4];
        rc = wait_words(p, 8, p->tail_seq, th);
        if (rc) return rc;
        if (th[2] == 0.0) {
            std::swap(d.edge_level, p->edge_level2);
            std::swap(d.oe_level, p->oe_level2);
            std::swap(d.edge_ha, p->edge_ha2);
            std::swap(d.pt_h, p->pt_h2);
            p->n_land = (int)th[3];
            p->all_active = false;
            p->host_stale = true;                                             // (levels / landmark numbering: fetched when somebody asks)
            if (!p->idx_uploaded.empty()) std::fill(p->idx_uploaded.begin() + d.n_kf + d.n_obj, p->idx_uploaded.end(), INT32_MIN);
            const StageHead head{th[0], th[1]};
            p->n_boundary_device++;
            t3s = now();
            rc = ba_optimize(p, 10, 0.0, 0.0, 0.0, stop_flag, t2, false, &head);   // robust kernels dropped, :628,643,655
            if (tm)
                fprintf(stderr, "local_joint us: set_levels %.0f  optimize(5) %.0f  boundary (device) %.0f  optimize(10) %.0f\n",
                        t1s - t0, t2s - t1s, t3s - t2s, now() - t3s);
            return rc;
        }
        // (a key-frame or an object lost its last active edge: the reduced system changes shape -- the host path below)
    }
    p->n_boundary_host++;
    if (p->world == 1) {
        // one GPU: classify where the chi2 values are; the host needs the levels only (build_index), one byte per edge
        const int n = std::max(std::max(d.n_edge, d.n_oe), 1);
        hipLaunchKernelGGL(k_classify_levels, dim3((n + 255) / 256), dim3(256), 0, p->stream, d, 5.991, 7.815, 1e3);
        if (d.n_edge) QSP_HIP(hipMemcpyAsync(p->lvl_host, d.edge_level, d.n_edge, hipMemcpyDeviceToHost, p->stream));
        if (d.n_oe) QSP_HIP(hipMemcpyAsync(p->lvl_host + d.n_edge, d.oe_level, d.n_oe, hipMemcpyDeviceToHost, p->stream));
        QSP_HIP(hipStreamSynchronize(p->stream));
        t3s = now();
        if (d.n_edge) memcpy(p->edge_level_h.data(), p->lvl_host, d.n_edge);
        if (d.n_oe) memcpy(p->oe_level_h.data(), p->lvl_host + d.n_edge, d.n_oe);
        p->all_active = false;
    } else {
    std::vector<double> cm(std::max(p->n_mono, 1)), cs(std::max(p->n_stereo, 1)), co(std::max(d.n_oe, 1));
    std::vector<uint8_t> pm(std::max(p->n_mono, 1)), ps(std::max(p->n_stereo, 1));
    rc = qsp_ba_get_edges(p, cm.data(), cs.data(), co.data(), pm.data(), ps.data());
    if (rc) return rc;
    t3s = now();
    std::vector<uint8_t> lm(std::max(p->n_mono, 1)), ls(std::max(p->n_stereo, 1)), lo(std::max(d.n_oe, 1));
    for (int e = 0; e < p->n_mono; ++e) lm[e] = (cm[e] > 5.991 || !pm[e]) ? 1 : 0;   // :621-626
    for (int e = 0; e < p->n_stereo; ++e) ls[e] = (cs[e] > 7.815 || !ps[e]) ? 1 : 0; // :636-641
    for (int e = 0; e < d.n_oe; ++e) lo[e] = (co[e] > 1e3) ? 1 : 0;                  // :650-654
    rc = qsp_ba_set_levels(p, lm.data(), ls.data(), lo.data());
    if (rc) return rc;
    }
    const double t4s = now();
    const double t5s = t4s;
    rc = qsp_ba_optimize(p, 10, 0.0, 0.0, 0.0, stop_flag, t2);                       // robust kernels dropped, :628,643,655
    if (tm)
        fprintf(stderr, "local_joint us: set_levels %.0f  optimize(5) %.0f  get_edges %.0f  classify %.0f  set_levels %.0f  optimize(10) %.0f\n",
                t1s - t0, t2s - t1s, t3s - t2s, t4s - t3s, t5s - t4s, now() - t5s);
    return rc;
}


// ---------------------------------------------------------------------------------------------------------------
// pose-only optimisation (Optimizer::PoseOptimization)
// ---------------------------------------------------------------------------------------------------------------
struct qsp_pose_optimizer {
    int device = 0;
    int cap = 0;
    hipStream_t stream = nullptr;
    double *X = nullptr, *obs = nullptr, *info = nullptr, *chi2 = nullptr;
    uint8_t *stereo = nullptr, *outlier = nullptr, *level = nullptr;
    PoseOptOut* out = nullptr;
};

extern "C" void qsp_pose_optimizer_destroy(qsp_pose_optimizer* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    void* ptrs[] = {h->X, h->obs, h->info, h->chi2, h->stereo, h->outlier, h->level, h->out};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int qsp_pose_optimizer_create(int device, int32_t max_points, qsp_pose_optimizer** out) {
    if (!out || max_points < 1) return qsp_fail(QSP_ERR_INVALID, "qsp_pose_optimizer_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return qsp_fail(QSP_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return qsp_fail(QSP_ERR_INVALID, "qsp_pose_optimizer_create: device out of range");
    QSP_HIP(hipSetDevice(device));
    qsp_pose_optimizer* h = new qsp_pose_optimizer();
    h->device = device;
    h->cap = max_points;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    const size_t n = (size_t)max_points;
    if (e == hipSuccess) e = hipMalloc((void**)&h->X, 24 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&h->obs, 24 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&h->info, 8 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&h->chi2, 8 * n);
    if (e == hipSuccess) e = hipMalloc((void**)&h->stereo, n);
    if (e == hipSuccess) e = hipMalloc((void**)&h->outlier, n);
    if (e == hipSuccess) e = hipMalloc((void**)&h->level, n);
    if (e == hipSuccess) e = hipMalloc((void**)&h->out, sizeof(PoseOptOut));
    if (e != hipSuccess) {
        qsp_pose_optimizer_destroy(h);
        return qsp_fail(QSP_ERR_DEVICE, hipGetErrorString(e));
    }
    *out = h;
    return QSP_OK;
}

extern "C" int qsp_pose_optimize(qsp_pose_optimizer* h, int32_t n, const double* K, const double* pose_in, const double* X,
                                 const double* obs, const double* info, const uint8_t* stereo, double* pose_out,
                                 uint8_t* outlier, int32_t* n_inliers, qsp_pose_trace* trace) {
    if (!h || !K || !pose_in || !pose_out || n < 0) return qsp_fail(QSP_ERR_INVALID, "qsp_pose_optimize: bad argument");
    if (n > h->cap) return qsp_fail(QSP_ERR_INVALID, "qsp_pose_optimize: more correspondences than the optimiser was created for");
    if (n > 0 && (!X || !obs || !info || !stereo)) return qsp_fail(QSP_ERR_INVALID, "qsp_pose_optimize: null edge array");
    QSP_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    PoseOptIn in;
    in.n = n; in.X = h->X; in.obs = h->obs; in.info = h->info; in.stereo = h->stereo;
    for (int i = 0; i < 5; ++i) in.K[i] = K[i];
    for (int i = 0; i < 7; ++i) in.pose0[i] = pose_in[i];
    if (n) {
        QSP_HIP(hipMemcpyAsync(h->X, X, 24 * (size_t)n, hipMemcpyHostToDevice, s));
        QSP_HIP(hipMemcpyAsync(h->obs, obs, 24 * (size_t)n, hipMemcpyHostToDevice, s));
        QSP_HIP(hipMemcpyAsync(h->info, info, 8 * (size_t)n, hipMemcpyHostToDevice, s));
        QSP_HIP(hipMemcpyAsync(h->stereo, stereo, (size_t)n, hipMemcpyHostToDevice, s));
    }
    hipLaunchKernelGGL(k_pose_opt, dim3(1), dim3(256), 0, s, in, h->outlier, h->level, h->chi2, h->out);
    QSP_HIP(hipGetLastError());
    PoseOptOut o;
    QSP_HIP(hipMemcpyAsync(&o, h->out, sizeof(o), hipMemcpyDeviceToHost, s));
    if (outlier && n) QSP_HIP(hipMemcpyAsync(outlier, h->outlier, (size_t)n, hipMemcpyDeviceToHost, s));
    QSP_HIP(hipStreamSynchronize(s));
    for (int i = 0; i < 7; ++i) pose_out[i] = o.pose[i];
    if (n_inliers) *n_inliers = o.n_inliers;
    if (trace) {
        for (int r = 0; r < 4; ++r) trace->iters[r] = o.iters[r];
        memcpy(trace->trace, o.trace, sizeof(o.trace));
    }
    return QSP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// deterministic mode
// ---------------------------------------------------------------------------------------------------------------
extern "C" int qsp_ba_set_option(qsp_ba_problem* p, int32_t option, int32_t value) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_set_option: null problem");
    switch (option) {
        case QSP_BA_OPT_OBJECT_ELIMINATION: p->elim_allowed = value != 0; return QSP_OK;
        case QSP_BA_OPT_CHOLESKY_CHAIN:
            if (value && !p->chol_chain_ok) return qsp_fail(QSP_ERR_UNSUPPORTED, "cholesky chain: not set up for this problem (one block row, or QSP_BA_CHOL=steps)");
            p->chol_chain = value != 0;
            p->chol_fault = value == 2;     // tests: the chain without its tile workgroups -- every wait must expire, not hang
            return QSP_OK;
        default: return qsp_fail(QSP_ERR_INVALID, "qsp_ba_set_option: unknown option");
    }
}

extern "C" int qsp_ba_set_deterministic(qsp_ba_problem* p, int on) {
    if (!p) return qsp_fail(QSP_ERR_INVALID, "qsp_ba_set_deterministic: null problem");
    QSP_HIP(hipSetDevice(p->device));
    if (on) {
        const int rc = build_pair_lists(p);
        if (rc) return rc;
    }
    p->deterministic = on != 0;
    return QSP_OK;
}

#include "ellipsoid_fit.hpp"

#ifdef QSP_CB_STAMPS
extern "C" int qsp_debug_chain_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qsp::ba::qsp_chain_ts), sizeof(unsigned long long) * 64 * 8);
}
extern "C" int qsp_debug_cb_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qsp::ba::qsp_cb_ts), sizeof(unsigned long long) * 64 * 5);
}
#endif
