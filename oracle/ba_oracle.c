/* ba_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, double precision, single thread) of hot path B, the g2o-driven joint bundle adjustment of
 * QSP-SLAM.  It is the *checker*: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (qsp_slam_amd/) never links or calls it.
 *
 * Parity status: the reference's C++ path cannot be built here (g2o and src/Optimizer*.cc need Eigen >= 3.4, which the
 * image lacks -- SURVEY.md section 8c) and the reference has no tests or golden vectors of its own, so this restatement is
 * pinned by (1) oracle/ba_dense_check.py, an independent numpy/scipy formulation (full, un-marginalised normal equations
 * with central-difference Jacobians and its own Lie-group code) that must agree on every iteration's chi2, lambda and
 * accept/reject decision and on the final estimates, and (2) analytic self-checks (Jacobians vs central differences,
 * exp/log round trips) in tests/test_oracle_ba.py.  PARITY UNPINNED against the reference binary itself.
 *
 * Each function cites the reference lines it follows (paths relative to the reference root).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef struct {
    int n_kf, n_pt, n_obj, n_mono, n_stereo, n_oe;
    double* kf_pose;            /* [n_kf][7]  tx ty tz qx qy qz qw  of T_cw                         */
    const unsigned char* kf_fixed;
    const long long* kf_id;     /* g2o vertex id = KeyFrame::mnId                                    */
    const double* kf_K;         /* [n_kf][5]  fx fy cx cy bf                                         */
    double* pt_xyz;             /* [n_pt][3]                                                         */
    const long long* pt_id;     /* mnId + maxKFid + 1                                                */
    double* obj_pose;           /* [n_obj][7] T_ow                                                   */
    const long long* obj_id;    /* mnId + maxKFid + maxMPid + 2                                      */
    const int *mono_pt, *mono_kf;
    const double* mono_obs;     /* [n_mono][2]                                                       */
    const double* mono_info;    /* [n_mono]   invSigma2 (information = invSigma2 * I2)               */
    const int *st_pt, *st_kf;
    const double* st_obs;       /* [n_stereo][3]                                                     */
    const double* st_info;      /* [n_stereo]                                                        */
    const int *oe_kf, *oe_obj;
    const double* oe_meas;      /* [n_oe][7]  Z = det->SE3Tco                                        */
    double oe_info;             /* information = oe_info * I6                                        */
    /* per-edge state */
    unsigned char *mono_level, *st_level, *oe_level;   /* 0 active, 1 outlier (g2o edge level)       */
    double *mono_chi2, *st_chi2, *oe_chi2;             /* e^T Omega e of the last computeActiveErrors */
} ba_problem;

typedef struct {
    double delta_mono, delta_stereo, delta_obj;   /* Huber delta; <= 0: no robust kernel                */
    const volatile unsigned char* stop_flag;      /* may be NULL                                        */
} ba_opts;

typedef struct {          /* per outer iteration, capacity given by the caller */
    int cap, n;
    double* chi2;         /* currentChi after the iteration          */
    double* lambda;       /* lambda after the iteration               */
    int* trials;          /* LM trials in the iteration               */
    int* accepted;        /* 1 if the last trial was accepted         */
    int* kf_hidx; int* obj_hidx; int* pt_hidx;   /* hessian indices (-1 fixed/inactive), filled once */
    int result;           /* 0 OK (ran all iterations), 1 Terminate, 2 stopped by flag */
} ba_trace;

/* ------------------------------------------------------------------------------------------------------------ */
/* small fixed-size linear algebra                                                                               */
/* ------------------------------------------------------------------------------------------------------------ */
static void skew3(const double* v, double* m) {   /* g2o/types/se3_ops.hpp:27-38 */
    m[0] = 0; m[1] = -v[2]; m[2] = v[1];
    m[3] = v[2]; m[4] = 0; m[5] = -v[0];
    m[6] = -v[1]; m[7] = v[0]; m[8] = 0;
}
static void mat3mul(const double* a, const double* b, double* c) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
static void quat_to_R(const double* q, double* R) {   /* Eigen QuaternionBase::toRotationMatrix; q = x y z w */
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                 tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void R_to_quat(const double* m, double* q) {   /* Eigen quaternion-from-matrix (Shoemake) */
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
static void quat_normalize(double* q) {   /* SE3Quat::normalizeRotation, se3quat.h:328-333 */
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_mul(const double* a, const double* b, double* c) {   /* Eigen quaternion product, x y z w */
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    c[0] = aw * bx + ax * bw + ay * bz - az * by;
    c[1] = aw * by + ay * bw + az * bx - ax * bz;
    c[2] = aw * bz + az * bw + ax * by - ay * bx;
    c[3] = aw * bw - ax * bx - ay * by - az * bz;
}
static void quat_rot(const double* q, const double* v, double* o) {   /* Eigen: v + w*uv + u x uv, uv = 2 u x v */
    const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
    double uvx = 2 * (uy * v[2] - uz * v[1]), uvy = 2 * (uz * v[0] - ux * v[2]), uvz = 2 * (ux * v[1] - uy * v[0]);
    o[0] = v[0] + w * uvx + (uy * uvz - uz * uvy);
    o[1] = v[1] + w * uvy + (uz * uvx - ux * uvz);
    o[2] = v[2] + w * uvz + (ux * uvy - uy * uvx);
}

/* pose = tx ty tz qx qy qz qw */
static void se3_mul(const double* a, const double* b, double* c) {   /* se3quat.h:110-116 */
    double rt[3], q[4];
    quat_rot(a + 3, b, rt);
    quat_mul(a + 3, b + 3, q);
    c[0] = a[0] + rt[0]; c[1] = a[1] + rt[1]; c[2] = a[2] + rt[2];
    quat_normalize(q);
    c[3] = q[0]; c[4] = q[1]; c[5] = q[2]; c[6] = q[3];
}
static void se3_inv(const double* a, double* c) {   /* se3quat.h:128-133 */
    double q[4] = {-a[3], -a[4], -a[5], a[6]}, mt[3] = {-a[0], -a[1], -a[2]}, t[3];
    quat_rot(q, mt, t);
    c[0] = t[0]; c[1] = t[1]; c[2] = t[2];
    c[3] = q[0]; c[4] = q[1]; c[5] = q[2]; c[6] = q[3];
}
static void se3_map(const double* a, const double* x, double* o) {   /* se3quat.h:267 */
    double r[3];
    quat_rot(a + 3, x, r);
    o[0] = r[0] + a[0]; o[1] = r[1] + a[1]; o[2] = r[2] + a[2];
}
void ba_se3_exp(const double* u, double* pose) {   /* se3quat.h:273-305; u = (omega, upsilon) */
    const double* om = u;
    const double* up = u + 3;
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    double Om[9], Om2[9], R[9], V[9];
    skew3(om, Om);
    mat3mul(Om, Om, Om2);
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = I[i] + Om[i] + Om2[i]; V[i] = R[i]; }
    } else {
        const double s = sin(theta), c = cos(theta);
        for (int i = 0; i < 9; ++i) {
            R[i] = I[i] + s / theta * Om[i] + (1 - c) / (theta * theta) * Om2[i];
            V[i] = I[i] + (1 - c) / (theta * theta) * Om[i] + (theta - s) / pow(theta, 3) * Om2[i];
        }
    }
    double q[4];
    R_to_quat(R, q);
    quat_normalize(q);   /* SE3Quat(q, t) constructor normalises */
    pose[0] = V[0] * up[0] + V[1] * up[1] + V[2] * up[2];
    pose[1] = V[3] * up[0] + V[4] * up[1] + V[5] * up[2];
    pose[2] = V[6] * up[0] + V[7] * up[1] + V[8] * up[2];
    pose[3] = q[0]; pose[4] = q[1]; pose[5] = q[2]; pose[6] = q[3];
}
void ba_se3_log(const double* pose, double* out) {   /* se3quat.h:228-265 */
    double R[9];
    quat_to_R(pose + 3, R);
    const double d = 0.5 * (R[0] + R[4] + R[8] - 1);
    double dR[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};   /* deltaR, se3_ops.hpp:40-47 */
    double om[3], Om[9], Om2[9], Vi[9];
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (d > 0.99999) {
        for (int i = 0; i < 3; ++i) om[i] = 0.5 * dR[i];
        skew3(om, Om);
        mat3mul(Om, Om, Om2);
        for (int i = 0; i < 9; ++i) Vi[i] = I[i] - 0.5 * Om[i] + (1. / 12.) * Om2[i];
    } else {
        const double theta = acos(d);
        for (int i = 0; i < 3; ++i) om[i] = theta / (2 * sqrt(1 - d * d)) * dR[i];
        skew3(om, Om);
        mat3mul(Om, Om, Om2);
        for (int i = 0; i < 9; ++i)
            Vi[i] = I[i] - 0.5 * Om[i] + (1 - theta / (2 * tan(theta / 2))) / (theta * theta) * Om2[i];
    }
    out[0] = om[0]; out[1] = om[1]; out[2] = om[2];
    for (int i = 0; i < 3; ++i) out[3 + i] = Vi[3 * i] * pose[0] + Vi[3 * i + 1] * pose[1] + Vi[3 * i + 2] * pose[2];
}
static void se3_adj(const double* pose, double* A) {   /* se3quat.h:307-316: [[R,0],[t^ R, R]] */
    double R[9], T[9], TR[9];
    quat_to_R(pose + 3, R);
    skew3(pose, T);
    mat3mul(T, R, TR);
    memset(A, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[6 * i + j] = R[3 * i + j];
            A[6 * (i + 3) + (j + 3)] = R[3 * i + j];
            A[6 * (i + 3) + j] = TR[3 * i + j];
        }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* edges                                                                                                          */
/* ------------------------------------------------------------------------------------------------------------ */
/* EdgeSE3ProjectXYZ: error (types_six_dof_expmap.h:91-96), Jacobians (.cpp:103-139).  Returns z (depth). */
double ba_mono_edge(const double* pose, const double* X, const double* K, const double* obs, double* e, double* Jp /*2x3*/,
                    double* Jx /*2x6*/) {
    double p[3], R[9];
    se3_map(pose, X, p);
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double x = p[0], y = p[1], z = p[2], z2 = z * z;
    e[0] = obs[0] - (x / z * fx + cx);
    e[1] = obs[1] - (y / z * fy + cy);
    if (Jp) {
        quat_to_R(pose + 3, R);
        const double t[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j)
                Jp[3 * i + j] = -1. / z * (t[3 * i] * R[j] + t[3 * i + 1] * R[3 + j] + t[3 * i + 2] * R[6 + j]);
        Jx[0] = x * y / z2 * fx; Jx[1] = -(1 + (x * x / z2)) * fx; Jx[2] = y / z * fx;
        Jx[3] = -1. / z * fx; Jx[4] = 0; Jx[5] = x / z2 * fx;
        Jx[6] = (1 + y * y / z2) * fy; Jx[7] = -x * y / z2 * fy; Jx[8] = -x / z * fy;
        Jx[9] = 0; Jx[10] = -1. / z * fy; Jx[11] = y / z2 * fy;
    }
    return z;
}
/* EdgeStereoSE3ProjectXYZ: error with the float-precision 1/z of cam_project (.cpp:150-157), Jacobians (.cpp:198-234) */
double ba_stereo_edge(const double* pose, const double* X, const double* K, const double* obs, double* e, double* Jp /*3x3*/,
                      double* Jx /*3x6*/) {
    double p[3], R[9];
    se3_map(pose, X, p);
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const float bf_f = (float)K[4];                 /* edge member `double bf` is passed as `const float&` */
    const float invz = 1.0f / (float)p[2];
    const double u = p[0] * invz * fx + cx, v = p[1] * invz * fy + cy;
    const double ur = u - bf_f * invz;
    e[0] = obs[0] - u; e[1] = obs[1] - v; e[2] = obs[2] - ur;
    if (Jp) {
        const double bf = K[4];
        const double x = p[0], y = p[1], z = p[2], z2 = z * z;
        quat_to_R(pose + 3, R);
        for (int j = 0; j < 3; ++j) {
            Jp[j] = -fx * R[j] / z + fx * x * R[6 + j] / z2;
            Jp[3 + j] = -fy * R[3 + j] / z + fy * y * R[6 + j] / z2;
            Jp[6 + j] = Jp[j] - bf * R[6 + j] / z2;
        }
        Jx[0] = x * y / z2 * fx; Jx[1] = -(1 + (x * x / z2)) * fx; Jx[2] = y / z * fx;
        Jx[3] = -1. / z * fx; Jx[4] = 0; Jx[5] = x / z2 * fx;
        Jx[6] = (1 + y * y / z2) * fy; Jx[7] = -x * y / z2 * fy; Jx[8] = -x / z * fy;
        Jx[9] = 0; Jx[10] = -1. / z * fy; Jx[11] = y / z2 * fy;
        Jx[12] = Jx[0] - bf * y / z2; Jx[13] = Jx[1] + bf * x / z2; Jx[14] = Jx[2];
        Jx[15] = Jx[3]; Jx[16] = 0; Jx[17] = Jx[5] - bf / z2;
    }
    return p[2];
}
/* EdgeSE3LieAlgebra, include/ObjectPoseGraph.h:69-88: e = log(Z^-1 T_cw T_ow^-1); Ji = J Adj(Z^-1), Jj = -J */
void ba_obj_edge(const double* Tcw, const double* Tow, const double* Z, double* e, double* Ji, double* Jj) {
    double Zi[7], Towi[7], a[7], b[7];
    se3_inv(Z, Zi);
    se3_inv(Tow, Towi);
    se3_mul(Zi, Tcw, a);
    se3_mul(a, Towi, b);
    ba_se3_log(b, e);
    if (Ji) {
        double J[36], W[9], T[9], A[36];
        skew3(e, W);
        skew3(e + 3, T);
        memset(J, 0, sizeof(J));
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                J[6 * i + j] = 0.5 * W[3 * i + j];
                J[6 * (i + 3) + j] = 0.5 * T[3 * i + j];
                J[6 * (i + 3) + (j + 3)] = 0.5 * W[3 * i + j];
            }
        for (int i = 0; i < 6; ++i) J[7 * i] += 1.0;
        se3_adj(Zi, A);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) {
                double s = 0;
                for (int k = 0; k < 6; ++k) s += J[6 * i + k] * A[6 * k + j];
                Ji[6 * i + j] = s;
                Jj[6 * i + j] = -J[6 * i + j];
            }
    }
}

/* RobustKernelHuber::robustify, g2o/core/robust_kernel_impl.cpp:78-91; delta <= 0: identity */
static void huber(double e, double delta, double* rho0, double* rho1) {
    if (delta <= 0 || e <= delta * delta) { *rho0 = e; *rho1 = 1.0; return; }
    const double s = sqrt(e);
    *rho0 = 2 * s * delta - delta * delta;
    *rho1 = delta / s;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* the solver state                                                                                               */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    ba_problem* p;
    const ba_opts* o;
    int n_pose;          /* free key-frames + objects = non-marginalised unknown blocks                      */
    int n_land;          /* active points                                                                     */
    int *kf_h, *obj_h, *pt_h;   /* hessian indices                                                           */
    double *Hpp;         /* dense (6 n_pose)^2, row-major; only blocks that g2o would allocate are touched   */
    double *Hll;         /* [n_land][9]                                                                       */
    double *bp, *bl;     /* right-hand sides                                                                  */
    double *Hs, *bs, *xp, *xl, *Dinv;
    int dim_p;
} ba_state;

static int cmp_ll(const void* a, const void* b) {
    const long long x = ((const long long*)a)[0], y = ((const long long*)b)[0];
    return (x > y) - (x < y);
}

/* SparseOptimizer::initializeOptimization(level) + buildIndexMapping, sparse_optimizer.cpp:199-267,166-190:
 * a vertex is active iff some edge of the requested level touches it; pass 0 orders the free non-marginalised vertices
 * (key-frames, objects) by vertex id, pass 1 the marginalised ones (points) by vertex id; fixed -> -1. */
static void build_index(ba_state* s) {
    ba_problem* p = s->p;
    unsigned char* ka = calloc(p->n_kf + 1, 1);
    unsigned char* oa = calloc(p->n_obj + 1, 1);
    unsigned char* pa = calloc(p->n_pt + 1, 1);
    for (int e = 0; e < p->n_mono; ++e)
        if (!p->mono_level[e]) { ka[p->mono_kf[e]] = 1; pa[p->mono_pt[e]] = 1; }
    for (int e = 0; e < p->n_stereo; ++e)
        if (!p->st_level[e]) { ka[p->st_kf[e]] = 1; pa[p->st_pt[e]] = 1; }
    for (int e = 0; e < p->n_oe; ++e)
        if (!p->oe_level[e]) { ka[p->oe_kf[e]] = 1; oa[p->oe_obj[e]] = 1; }
    long long* tmp = malloc(sizeof(long long) * 2 * (p->n_kf + p->n_obj + p->n_pt + 1));
    int n = 0;
    for (int i = 0; i < p->n_kf; ++i) {
        s->kf_h[i] = -1;
        if (ka[i] && !p->kf_fixed[i]) { tmp[2 * n] = p->kf_id[i]; tmp[2 * n + 1] = i; ++n; }
    }
    for (int i = 0; i < p->n_obj; ++i) {
        s->obj_h[i] = -1;
        if (oa[i]) { tmp[2 * n] = p->obj_id[i]; tmp[2 * n + 1] = p->n_kf + i; ++n; }
    }
    qsort(tmp, n, 2 * sizeof(long long), cmp_ll);
    for (int k = 0; k < n; ++k) {
        const int v = (int)tmp[2 * k + 1];
        if (v < p->n_kf) s->kf_h[v] = k; else s->obj_h[v - p->n_kf] = k;
    }
    s->n_pose = n;
    n = 0;
    for (int i = 0; i < p->n_pt; ++i) {
        s->pt_h[i] = -1;
        if (pa[i]) { tmp[2 * n] = p->pt_id[i]; tmp[2 * n + 1] = i; ++n; }
    }
    qsort(tmp, n, 2 * sizeof(long long), cmp_ll);
    for (int k = 0; k < n; ++k) s->pt_h[(int)tmp[2 * k + 1]] = k;
    s->n_land = n;
    free(tmp); free(ka); free(oa); free(pa);
}

/* computeActiveErrors + activeRobustChi2, sparse_optimizer.cpp:61-114 (edge order: mono, stereo, object) */
static double compute_errors(ba_state* s) {
    ba_problem* p = s->p;
    double chi = 0, e[6], r0, r1;
    for (int k = 0; k < p->n_mono; ++k) {
        if (p->mono_level[k]) continue;
        ba_mono_edge(p->kf_pose + 7 * p->mono_kf[k], p->pt_xyz + 3 * p->mono_pt[k], p->kf_K + 5 * p->mono_kf[k],
                     p->mono_obs + 2 * k, e, NULL, NULL);
        const double c = p->mono_info[k] * (e[0] * e[0] + e[1] * e[1]);
        p->mono_chi2[k] = c;
        huber(c, s->o->delta_mono, &r0, &r1);
        chi += r0;
    }
    for (int k = 0; k < p->n_stereo; ++k) {
        if (p->st_level[k]) continue;
        ba_stereo_edge(p->kf_pose + 7 * p->st_kf[k], p->pt_xyz + 3 * p->st_pt[k], p->kf_K + 5 * p->st_kf[k],
                       p->st_obs + 3 * k, e, NULL, NULL);
        const double c = p->st_info[k] * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        p->st_chi2[k] = c;
        huber(c, s->o->delta_stereo, &r0, &r1);
        chi += r0;
    }
    for (int k = 0; k < p->n_oe; ++k) {
        if (p->oe_level[k]) continue;
        ba_obj_edge(p->kf_pose + 7 * p->oe_kf[k], p->obj_pose + 7 * p->oe_obj[k], p->oe_meas + 7 * k, e, NULL, NULL);
        double c = 0;
        for (int i = 0; i < 6; ++i) c += e[i] * e[i];
        c *= p->oe_info;
        p->oe_chi2[k] = c;
        huber(c, s->o->delta_obj, &r0, &r1);
        chi += r0;
    }
    return chi;
}

/* A += Ja^T (w) Jb for D x na and D x nb Jacobians into a dense block at (ra, rb) of an ld-wide matrix */
static void acc_block(double* M, int ld, int ra, int rb, const double* Ja, int na, const double* Jb, int nb, int D,
                      double w) {
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            double sum = 0;
            for (int d = 0; d < D; ++d) sum += Ja[d * na + i] * Jb[d * nb + j];
            M[(size_t)(ra + i) * ld + rb + j] += w * sum;
        }
}

/* BlockSolver::buildSystem -> linearizeOplus + constructQuadraticForm per edge
 * (block_solver.hpp:502-560, base_binary_edge.hpp:55-120).  Hpl is kept per edge (6x3). */
typedef struct { double* Hpl_mono; double* Hpl_st; } ba_hpl;

static void build_system(ba_state* s, ba_hpl* hpl) {
    ba_problem* p = s->p;
    const int dp = s->dim_p;
    memset(s->Hpp, 0, sizeof(double) * (size_t)dp * dp);
    memset(s->Hll, 0, sizeof(double) * 9 * (size_t)(s->n_land > 0 ? s->n_land : 1));
    memset(s->bp, 0, sizeof(double) * (dp > 0 ? dp : 1));
    memset(s->bl, 0, sizeof(double) * 3 * (size_t)(s->n_land > 0 ? s->n_land : 1));
    double e[6], Jp[9], Jx[18], r0, r1;
    for (int k = 0; k < p->n_mono; ++k) {
        if (p->mono_level[k]) continue;
        const int kf = p->mono_kf[k], pt = p->mono_pt[k];
        ba_mono_edge(p->kf_pose + 7 * kf, p->pt_xyz + 3 * pt, p->kf_K + 5 * kf, p->mono_obs + 2 * k, e, Jp, Jx);
        const double c = p->mono_info[k] * (e[0] * e[0] + e[1] * e[1]);
        huber(c, s->o->delta_mono, &r0, &r1);
        const double w = r1 * p->mono_info[k];
        const int hl = s->pt_h[pt], hp = s->kf_h[kf];
        /* vertex 0 = point (Xi), vertex 1 = pose (Xj) */
        for (int i = 0; i < 3; ++i) {
            double sb = 0;
            for (int d = 0; d < 2; ++d) sb += Jp[3 * d + i] * (-p->mono_info[k] * e[d]) * r1;
            s->bl[3 * hl + i] += sb;
            for (int j = 0; j < 3; ++j) {
                double sum = 0;
                for (int d = 0; d < 2; ++d) sum += Jp[3 * d + i] * Jp[3 * d + j];
                s->Hll[9 * hl + 3 * i + j] += w * sum;
            }
        }
        double* B = hpl->Hpl_mono + 18 * (size_t)k;
        memset(B, 0, 18 * sizeof(double));
        if (hp >= 0) {
            for (int i = 0; i < 6; ++i) {
                double sb = 0;
                for (int d = 0; d < 2; ++d) sb += Jx[6 * d + i] * (-p->mono_info[k] * e[d]) * r1;
                s->bp[6 * hp + i] += sb;
            }
            acc_block(s->Hpp, dp, 6 * hp, 6 * hp, Jx, 6, Jx, 6, 2, w);
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) {
                    double sum = 0;
                    for (int d = 0; d < 2; ++d) sum += Jx[6 * d + i] * Jp[3 * d + j];
                    B[3 * i + j] = w * sum;
                }
        }
    }
    for (int k = 0; k < p->n_stereo; ++k) {
        if (p->st_level[k]) continue;
        const int kf = p->st_kf[k], pt = p->st_pt[k];
        ba_stereo_edge(p->kf_pose + 7 * kf, p->pt_xyz + 3 * pt, p->kf_K + 5 * kf, p->st_obs + 3 * k, e, Jp, Jx);
        const double c = p->st_info[k] * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        huber(c, s->o->delta_stereo, &r0, &r1);
        const double w = r1 * p->st_info[k];
        const int hl = s->pt_h[pt], hp = s->kf_h[kf];
        for (int i = 0; i < 3; ++i) {
            double sb = 0;
            for (int d = 0; d < 3; ++d) sb += Jp[3 * d + i] * (-p->st_info[k] * e[d]) * r1;
            s->bl[3 * hl + i] += sb;
            for (int j = 0; j < 3; ++j) {
                double sum = 0;
                for (int d = 0; d < 3; ++d) sum += Jp[3 * d + i] * Jp[3 * d + j];
                s->Hll[9 * hl + 3 * i + j] += w * sum;
            }
        }
        double* B = hpl->Hpl_st + 18 * (size_t)k;
        memset(B, 0, 18 * sizeof(double));
        if (hp >= 0) {
            for (int i = 0; i < 6; ++i) {
                double sb = 0;
                for (int d = 0; d < 3; ++d) sb += Jx[6 * d + i] * (-p->st_info[k] * e[d]) * r1;
                s->bp[6 * hp + i] += sb;
            }
            acc_block(s->Hpp, dp, 6 * hp, 6 * hp, Jx, 6, Jx, 6, 3, w);
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) {
                    double sum = 0;
                    for (int d = 0; d < 3; ++d) sum += Jx[6 * d + i] * Jp[3 * d + j];
                    B[3 * i + j] = w * sum;
                }
        }
    }
    double Ji[36], Jj[36];
    for (int k = 0; k < p->n_oe; ++k) {
        if (p->oe_level[k]) continue;
        const int kf = p->oe_kf[k], ob = p->oe_obj[k];
        ba_obj_edge(p->kf_pose + 7 * kf, p->obj_pose + 7 * ob, p->oe_meas + 7 * k, e, Ji, Jj);
        double c = 0;
        for (int i = 0; i < 6; ++i) c += e[i] * e[i];
        c *= p->oe_info;
        huber(c, s->o->delta_obj, &r0, &r1);
        const double w = r1 * p->oe_info;
        const int hi = s->kf_h[kf], hj = s->obj_h[ob];
        if (hi >= 0) {
            for (int i = 0; i < 6; ++i) {
                double sb = 0;
                for (int d = 0; d < 6; ++d) sb += Ji[6 * d + i] * (-p->oe_info * e[d]) * r1;
                s->bp[6 * hi + i] += sb;
            }
            acc_block(s->Hpp, dp, 6 * hi, 6 * hi, Ji, 6, Ji, 6, 6, w);
            if (hj >= 0) {   /* off-diagonal block, stored in the upper triangle (hi < hj always: KF ids < object ids) */
                const int a = hi < hj ? hi : hj, b = hi < hj ? hj : hi;
                if (hi < hj) acc_block(s->Hpp, dp, 6 * a, 6 * b, Ji, 6, Jj, 6, 6, w);
                else acc_block(s->Hpp, dp, 6 * a, 6 * b, Jj, 6, Ji, 6, 6, w);
            }
        }
        if (hj >= 0) {
            for (int i = 0; i < 6; ++i) {
                double sb = 0;
                for (int d = 0; d < 6; ++d) sb += Jj[6 * d + i] * (-p->oe_info * e[d]) * r1;
                s->bp[6 * hj + i] += sb;
            }
            acc_block(s->Hpp, dp, 6 * hj, 6 * hj, Jj, 6, Jj, 6, 6, w);
        }
    }
}

static int inv3(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double id = 1.0 / det;
    o[0] = (e * i - f * h) * id; o[1] = (c * h - b * i) * id; o[2] = (b * f - c * e) * id;
    o[3] = (f * g - d * i) * id; o[4] = (a * i - c * g) * id; o[5] = (c * d - a * f) * id;
    o[6] = (d * h - e * g) * id; o[7] = (b * g - a * h) * id; o[8] = (a * e - b * d) * id;
    return isfinite(id);
}

/* dense Cholesky solve of the symmetric system given by its UPPER triangle (LinearSolverEigen uses a sparse LDL^T of
 * the upper triangle, linear_solver_eigen.h:94-124; same solution up to rounding).  Returns 0 if not positive definite. */
static int chol_solve_upper(double* A, int n, const double* b, double* x) {
    /* A = U^T U, U upper, in place */
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)k * n + j] * A[(size_t)k * n + j];
        if (!(d > 0) || !isfinite(d)) return 0;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = A[(size_t)j * n + i];
            for (int k = 0; k < j; ++k) v -= A[(size_t)k * n + j] * A[(size_t)k * n + i];
            A[(size_t)j * n + i] = v / d;
        }
    }
    for (int i = 0; i < n; ++i) {   /* U^T y = b */
        double v = b[i];
        for (int k = 0; k < i; ++k) v -= A[(size_t)k * n + i] * x[k];
        x[i] = v / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {   /* U x = y */
        double v = x[i];
        for (int k = i + 1; k < n; ++k) v -= A[(size_t)i * n + k] * x[k];
        x[i] = v / A[(size_t)i * n + i];
    }
    return 1;
}

/* ---- the linear solver the TIMED CPU baseline uses (bench.py cpu_baseline) ------------------------------
 * g2o's LinearSolverEigen factorises the reduced camera system with a sparse LDL^T whose elimination order comes from AMD on the
 * 6x6 BLOCK pattern (Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:94-124 solve, :147-201 computeSymbolicDecomposition).  The
 * dense solve above costs dim^3/3 whatever the pattern, which at 200 key-frames + 256 objects is a straw man.  This is the same
 * idea at block granularity: the 6x6 block pattern of the upper triangle, a greedy minimum-degree elimination order on it (with
 * the fill each elimination creates), then a right-looking block Cholesky that only ever touches blocks of the filled pattern.
 * An object vertex sees ~10 key-frames and is eliminated early; the key-frame part of a local window is nearly dense.  Same
 * solution as the dense solve to rounding (tests/test_oracle_ba.py); the PARITY oracle keeps the dense solve (its operation
 * order is what the GPU tests were pinned against).  Single thread, like g2o. */
static int g_sparse_solver = 0;
void ba_oracle_set_sparse_solver(int on) { g_sparse_solver = on; }

static int chol6(double* A /*6x6 row-major, upper used; overwritten by U*/) {
    for (int j = 0; j < 6; ++j) {
        double d = A[7 * j];
        for (int k = 0; k < j; ++k) d -= A[6 * k + j] * A[6 * k + j];
        if (!(d > 0) || !isfinite(d)) return 0;
        d = sqrt(d);
        A[7 * j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[6 * j + i];
            for (int k = 0; k < j; ++k) v -= A[6 * k + j] * A[6 * k + i];
            A[6 * j + i] = v / d;
        }
        for (int i = 0; i < j; ++i) A[6 * j + i] = 0.0;
    }
    return 1;
}

static int chol_solve_block_sparse(double* A, int n, const double* b, double* x) {
    const int nb = n / 6;
    if (nb * 6 != n || nb == 0) return chol_solve_upper(A, n, b, x);
    unsigned char* nz = calloc((size_t)nb * nb, 1);          /* symmetric block pattern */
    for (int i = 0; i < nb; ++i)
        for (int j = i; j < nb; ++j) {
            int any = i == j;
            for (int r = 0; r < 6 && !any; ++r)
                for (int c = 0; c < 6; ++c)
                    if (A[(size_t)(6 * i + r) * n + 6 * j + c] != 0.0) { any = 1; break; }
            nz[(size_t)i * nb + j] = nz[(size_t)j * nb + i] = (unsigned char)any;
        }
    /* greedy minimum degree on the elimination graph (ties: lowest index) */
    int* order = malloc(sizeof(int) * nb);
    int* pos = malloc(sizeof(int) * nb);
    unsigned char* done = calloc(nb, 1);
    unsigned char* g = malloc((size_t)nb * nb);
    memcpy(g, nz, (size_t)nb * nb);
    int* nbr = malloc(sizeof(int) * nb);
    for (int step = 0; step < nb; ++step) {
        int best = -1, bestdeg = nb + 1;
        for (int v = 0; v < nb; ++v) {
            if (done[v]) continue;
            int deg = 0;
            for (int u = 0; u < nb; ++u) deg += (!done[u] && u != v && g[(size_t)v * nb + u]);
            if (deg < bestdeg) { bestdeg = deg; best = v; }
        }
        int m = 0;
        for (int u = 0; u < nb; ++u)
            if (!done[u] && u != best && g[(size_t)best * nb + u]) nbr[m++] = u;
        for (int a = 0; a < m; ++a)
            for (int c = 0; c < m; ++c) g[(size_t)nbr[a] * nb + nbr[c]] = 1;     /* fill */
        done[best] = 1;
        order[step] = best;
        pos[best] = step;
    }
    /* permuted copy (upper triangle of P A P^T, block-wise), filled pattern in elimination order */
    double* M = calloc((size_t)n * n, sizeof(double));
    unsigned char* fp = calloc((size_t)nb * nb, 1);
    for (int i = 0; i < nb; ++i)
        for (int j = i; j < nb; ++j) {
            if (!nz[(size_t)i * nb + j]) continue;
            int pi = pos[i], pj = pos[j];
            const int tr = pi > pj;                           /* lands below the diagonal: store the transpose */
            if (tr) { const int t_ = pi; pi = pj; pj = t_; }
            fp[(size_t)pi * nb + pj] = 1;
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    const double v = (i == j && c < r) ? A[(size_t)(6 * i + c) * n + 6 * j + r] : A[(size_t)(6 * i + r) * n + 6 * j + c];
                    if (tr) M[(size_t)(6 * pi + c) * n + 6 * pj + r] = v;
                    else M[(size_t)(6 * pi + r) * n + 6 * pj + c] = v;
                }
        }
    int ok = 1;
    int* cols = malloc(sizeof(int) * nb);
    for (int k = 0; k < nb && ok; ++k) {
        double D[36];
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) D[6 * r + c] = M[(size_t)(6 * k + r) * n + 6 * k + c];
        if (!chol6(D)) { ok = 0; break; }
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) M[(size_t)(6 * k + r) * n + 6 * k + c] = D[6 * r + c];
        int m = 0;
        for (int j = k + 1; j < nb; ++j)
            if (fp[(size_t)k * nb + j]) cols[m++] = j;
        for (int a = 0; a < m; ++a) {                         /* U_kj = U_kk^-T A_kj */
            double* B = M + (size_t)(6 * k) * n + 6 * cols[a];
            for (int c = 0; c < 6; ++c)
                for (int r = 0; r < 6; ++r) {
                    double v = B[(size_t)r * n + c];
                    for (int q = 0; q < r; ++q) v -= D[6 * q + r] * B[(size_t)q * n + c];
                    B[(size_t)r * n + c] = v / D[7 * r];
                }
        }
        for (int a = 0; a < m; ++a)                           /* A_ij -= U_ki^T U_kj for i <= j among the row's blocks */
            for (int c2 = a; c2 < m; ++c2) {
                const int i = cols[a], j = cols[c2];
                fp[(size_t)i * nb + j] = 1;
                const double* Ui = M + (size_t)(6 * k) * n + 6 * i;
                const double* Uj = M + (size_t)(6 * k) * n + 6 * j;
                double* T = M + (size_t)(6 * i) * n + 6 * j;
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) {
                        double v = 0;
                        for (int q = 0; q < 6; ++q) v += Ui[(size_t)q * n + r] * Uj[(size_t)q * n + c];
                        T[(size_t)r * n + c] -= v;
                    }
            }
    }
    if (ok) {
        double* y = malloc(sizeof(double) * n);
        for (int i = 0; i < nb; ++i)
            for (int r = 0; r < 6; ++r) y[6 * pos[i] + r] = b[6 * i + r];
        for (int k = 0; k < nb; ++k) {                        /* U^T y = P b: forward, row k of U scatters into later rows */
            for (int r = 0; r < 6; ++r) {
                double v = y[6 * k + r];
                for (int q = 0; q < r; ++q) v -= M[(size_t)(6 * k + q) * n + 6 * k + r] * y[6 * k + q];
                y[6 * k + r] = v / M[(size_t)(6 * k + r) * n + 6 * k + r];
            }
            for (int j = k + 1; j < nb; ++j) {
                if (!fp[(size_t)k * nb + j]) continue;
                for (int c = 0; c < 6; ++c) {
                    double v = 0;
                    for (int q = 0; q < 6; ++q) v += M[(size_t)(6 * k + q) * n + 6 * j + c] * y[6 * k + q];
                    y[6 * j + c] -= v;
                }
            }
        }
        for (int k = nb - 1; k >= 0; --k) {                   /* U z = y: backward */
            for (int j = k + 1; j < nb; ++j) {
                if (!fp[(size_t)k * nb + j]) continue;
                for (int r = 0; r < 6; ++r) {
                    double v = 0;
                    for (int c = 0; c < 6; ++c) v += M[(size_t)(6 * k + r) * n + 6 * j + c] * y[6 * j + c];
                    y[6 * k + r] -= v;
                }
            }
            for (int r = 5; r >= 0; --r) {
                double v = y[6 * k + r];
                for (int c = r + 1; c < 6; ++c) v -= M[(size_t)(6 * k + r) * n + 6 * k + c] * y[6 * k + c];
                y[6 * k + r] = v / M[(size_t)(6 * k + r) * n + 6 * k + r];
            }
        }
        for (int i = 0; i < nb; ++i)
            for (int r = 0; r < 6; ++r) x[6 * i + r] = y[6 * pos[i] + r];
        free(y);
    }
    free(nz); free(order); free(pos); free(done); free(g); free(nbr); free(M); free(fp); free(cols);
    return ok;
}

/* BlockSolver::solve with Schur complement, block_solver.hpp:354-486, on the lambda-augmented system */
static int solve_schur(ba_state* s, const ba_hpl* hpl, double lambda) {
    ba_problem* p = s->p;
    const int dp = s->dim_p;
    /* Hs = Hpp (upper) + lambda on the diagonal; bs = bp */
    memcpy(s->Hs, s->Hpp, sizeof(double) * (size_t)dp * dp);
    for (int i = 0; i < dp; ++i) s->Hs[(size_t)i * dp + i] += lambda;
    memcpy(s->bs, s->bp, sizeof(double) * (dp > 0 ? dp : 1));
    /* per landmark: Dinv, then the (i1 <= i2) block updates over its observing free poses */
    for (int pt = 0; pt < p->n_pt; ++pt) {
        const int hl = s->pt_h[pt];
        if (hl < 0) continue;
        double D[9];
        memcpy(D, s->Hll + 9 * hl, sizeof(D));
        D[0] += lambda; D[4] += lambda; D[8] += lambda;
        inv3(D, s->Dinv + 9 * hl);
    }
    /* edge lists per landmark are implicit: loop edges twice (quadratic only in the per-landmark degree) */
    /* gather per landmark the (pose index, block) pairs */
    int* cnt = calloc(s->n_land + 1, sizeof(int));
    for (int k = 0; k < p->n_mono; ++k)
        if (!p->mono_level[k] && s->kf_h[p->mono_kf[k]] >= 0) cnt[s->pt_h[p->mono_pt[k]]]++;
    for (int k = 0; k < p->n_stereo; ++k)
        if (!p->st_level[k] && s->kf_h[p->st_kf[k]] >= 0) cnt[s->pt_h[p->st_pt[k]]]++;
    int* off = malloc(sizeof(int) * (s->n_land + 1));
    off[0] = 0;
    for (int i = 0; i < s->n_land; ++i) off[i + 1] = off[i] + cnt[i];
    const int tot = off[s->n_land];
    int* eh = malloc(sizeof(int) * (tot + 1));
    const double** eb = malloc(sizeof(double*) * (tot + 1));
    memset(cnt, 0, sizeof(int) * (s->n_land + 1));
    for (int k = 0; k < p->n_mono; ++k)
        if (!p->mono_level[k] && s->kf_h[p->mono_kf[k]] >= 0) {
            const int hl = s->pt_h[p->mono_pt[k]], q = off[hl] + cnt[hl]++;
            eh[q] = s->kf_h[p->mono_kf[k]];
            eb[q] = hpl->Hpl_mono + 18 * (size_t)k;
        }
    for (int k = 0; k < p->n_stereo; ++k)
        if (!p->st_level[k] && s->kf_h[p->st_kf[k]] >= 0) {
            const int hl = s->pt_h[p->st_pt[k]], q = off[hl] + cnt[hl]++;
            eh[q] = s->kf_h[p->st_kf[k]];
            eb[q] = hpl->Hpl_st + 18 * (size_t)k;
        }
    for (int hl = 0; hl < s->n_land; ++hl) {
        const double* Di = s->Dinv + 9 * hl;
        double db[3];
        for (int i = 0; i < 3; ++i) db[i] = Di[3 * i] * s->bl[3 * hl] + Di[3 * i + 1] * s->bl[3 * hl + 1] + Di[3 * i + 2] * s->bl[3 * hl + 2];
        for (int a = off[hl]; a < off[hl + 1]; ++a) {
            const double* Ba = eb[a];
            double BD[18];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) BD[3 * i + j] = Ba[3 * i] * Di[j] + Ba[3 * i + 1] * Di[3 + j] + Ba[3 * i + 2] * Di[6 + j];
            for (int i = 0; i < 6; ++i) s->bs[6 * eh[a] + i] -= Ba[3 * i] * db[0] + Ba[3 * i + 1] * db[1] + Ba[3 * i + 2] * db[2];
            for (int b = off[hl]; b < off[hl + 1]; ++b) {
                if (eh[b] < eh[a]) continue;           /* upper triangle, i2 >= i1 */
                if (eh[b] == eh[a] && b != a) continue; /* one edge per (pose, landmark) pair */
                const double* Bb = eb[b];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j)
                        s->Hs[(size_t)(6 * eh[a] + i) * dp + 6 * eh[b] + j] -=
                            BD[3 * i] * Bb[3 * j] + BD[3 * i + 1] * Bb[3 * j + 1] + BD[3 * i + 2] * Bb[3 * j + 2];
            }
        }
    }
    int ok = 1;
    if (dp > 0) ok = g_sparse_solver ? chol_solve_block_sparse(s->Hs, dp, s->bs, s->xp) : chol_solve_upper(s->Hs, dp, s->bs, s->xp);
    if (ok) {   /* x_l = Dinv (b_l - B^T x_p), block_solver.hpp:461-481 */
        for (int hl = 0; hl < s->n_land; ++hl) {
            double c[3] = {s->bl[3 * hl], s->bl[3 * hl + 1], s->bl[3 * hl + 2]};
            for (int a = off[hl]; a < off[hl + 1]; ++a)
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 6; ++i) c[j] -= eb[a][3 * i + j] * s->xp[6 * eh[a] + i];
            const double* Di = s->Dinv + 9 * hl;
            for (int i = 0; i < 3; ++i) s->xl[3 * hl + i] = Di[3 * i] * c[0] + Di[3 * i + 1] * c[1] + Di[3 * i + 2] * c[2];
        }
    }
    free(cnt); free(off); free(eh); free(eb);
    return ok;
}

/* SparseOptimizer::update -> oplus: poses exp(d) * T (types_six_dof_expmap.h:73-76), points += d (types_sba.h:52-56) */
static void apply_update(ba_state* s) {
    ba_problem* p = s->p;
    double d[7], n[7];
    for (int i = 0; i < p->n_kf; ++i)
        if (s->kf_h[i] >= 0) {
            ba_se3_exp(s->xp + 6 * s->kf_h[i], d);
            se3_mul(d, p->kf_pose + 7 * i, n);
            memcpy(p->kf_pose + 7 * i, n, sizeof(n));
        }
    for (int i = 0; i < p->n_obj; ++i)
        if (s->obj_h[i] >= 0) {
            ba_se3_exp(s->xp + 6 * s->obj_h[i], d);
            se3_mul(d, p->obj_pose + 7 * i, n);
            memcpy(p->obj_pose + 7 * i, n, sizeof(n));
        }
    for (int i = 0; i < p->n_pt; ++i)
        if (s->pt_h[i] >= 0)
            for (int j = 0; j < 3; ++j) p->pt_xyz[3 * i + j] += s->xl[3 * s->pt_h[i] + j];
}

/* SparseOptimizer::optimize(n_iter) with OptimizationAlgorithmLevenberg::solve
 * (sparse_optimizer.cpp:354-435, optimization_algorithm_levenberg.cpp:61-189).  Returns iterations done. */
int ba_oracle_optimize(ba_problem* p, int n_iter, const ba_opts* o, ba_trace* tr) {
    ba_state s;
    memset(&s, 0, sizeof(s));
    s.p = p; s.o = o;
    s.kf_h = malloc(sizeof(int) * (p->n_kf + 1));
    s.obj_h = malloc(sizeof(int) * (p->n_obj + 1));
    s.pt_h = malloc(sizeof(int) * (p->n_pt + 1));
    build_index(&s);
    if (tr && tr->kf_hidx) {
        memcpy(tr->kf_hidx, s.kf_h, sizeof(int) * p->n_kf);
        memcpy(tr->obj_hidx, s.obj_h, sizeof(int) * p->n_obj);
        memcpy(tr->pt_hidx, s.pt_h, sizeof(int) * p->n_pt);
    }
    const int dp = s.dim_p = 6 * s.n_pose;
    const size_t nl = s.n_land > 0 ? s.n_land : 1, dpp = dp > 0 ? dp : 1;
    s.Hpp = malloc(sizeof(double) * dpp * dpp);
    s.Hs = malloc(sizeof(double) * dpp * dpp);
    s.Hll = malloc(sizeof(double) * 9 * nl);
    s.Dinv = malloc(sizeof(double) * 9 * nl);
    s.bp = malloc(sizeof(double) * dpp); s.bs = malloc(sizeof(double) * dpp); s.xp = calloc(dpp, sizeof(double));
    s.bl = malloc(sizeof(double) * 3 * nl); s.xl = calloc(3 * nl, sizeof(double));
    ba_hpl hpl;
    hpl.Hpl_mono = malloc(sizeof(double) * 18 * (size_t)(p->n_mono + 1));
    hpl.Hpl_st = malloc(sizeof(double) * 18 * (size_t)(p->n_stereo + 1));
    double* bk_kf = malloc(sizeof(double) * 7 * (p->n_kf + 1));
    double* bk_ob = malloc(sizeof(double) * 7 * (p->n_obj + 1));
    double* bk_pt = malloc(sizeof(double) * 3 * (p->n_pt + 1));

    double lambda = 0, ni = 2;
    int nBad = 0, done = 0, result = 0;
    if (tr) tr->n = 0;
    for (int it = 0; it < n_iter; ++it) {
        if (o->stop_flag && *o->stop_flag) { result = 2; break; }
        double currentChi = compute_errors(&s);
        double tempChi = currentChi;
        const double iniChi = currentChi;
        build_system(&s, &hpl);
        if (it == 0) {   /* computeLambdaInit: tau * max |diagonal| over all active vertices, tau = 1e-5 */
            double md = 0;
            for (int i = 0; i < dp; ++i) md = fmax(fabs(s.Hpp[(size_t)i * dp + i]), md);
            for (int l = 0; l < s.n_land; ++l)
                for (int j = 0; j < 3; ++j) md = fmax(fabs(s.Hll[9 * l + 4 * j]), md);
            lambda = 1e-5 * md;
            ni = 2;
            nBad = 0;
        }
        double rho = 0;
        int qmax = 0, accepted = 0;
        do {
            memcpy(bk_kf, p->kf_pose, sizeof(double) * 7 * p->n_kf);          /* push */
            memcpy(bk_ob, p->obj_pose, sizeof(double) * 7 * p->n_obj);
            memcpy(bk_pt, p->pt_xyz, sizeof(double) * 3 * p->n_pt);
            const int ok2 = solve_schur(&s, &hpl, lambda);
            if (ok2) apply_update(&s);
            /* (when the solve fails g2o still applies the stale x of the previous trial, then rejects and pops;
               the popped state is identical, so skipping the stale update changes nothing observable but chi2[]) */
            tempChi = compute_errors(&s);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0;   /* computeScale: sum over ALL unknowns x_j (lambda x_j + b_j) */
            for (int i = 0; i < dp; ++i) scale += s.xp[i] * (lambda * s.xp[i] + s.bp[i]);
            for (int i = 0; i < 3 * s.n_land; ++i) scale += s.xl[i] * (lambda * s.xl[i] + s.bl[i]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                const double sf = fmax(1. / 3., alpha);
                lambda *= sf;
                ni = 2;
                currentChi = tempChi;
                accepted = 1;
            } else {
                lambda *= ni;
                ni *= 2;
                memcpy(p->kf_pose, bk_kf, sizeof(double) * 7 * p->n_kf);      /* pop */
                memcpy(p->obj_pose, bk_ob, sizeof(double) * 7 * p->n_obj);
                memcpy(p->pt_xyz, bk_pt, sizeof(double) * 3 * p->n_pt);
                accepted = 0;
            }
            qmax++;
        } while (rho < 0 && qmax < 10 && !(o->stop_flag && *o->stop_flag));
        done++;
        if (tr && tr->n < tr->cap) {
            tr->chi2[tr->n] = currentChi; tr->lambda[tr->n] = lambda; tr->trials[tr->n] = qmax;
            tr->accepted[tr->n] = accepted; tr->n++;
        }
        if (qmax == 10 || rho == 0) { result = 1; break; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) { result = 1; break; }
    }
    if (tr) tr->result = result;
    free(s.kf_h); free(s.obj_h); free(s.pt_h); free(s.Hpp); free(s.Hs); free(s.Hll); free(s.Dinv);
    free(s.bp); free(s.bs); free(s.xp); free(s.bl); free(s.xl); free(hpl.Hpl_mono); free(hpl.Hpl_st);
    free(bk_kf); free(bk_ob); free(bk_pt);
    return done;
}

/* isDepthPositive of every mono / stereo edge from the CURRENT estimates (types_six_dof_expmap.h:97-101,129-133) */
void ba_oracle_depth_positive(const ba_problem* p, unsigned char* mono_pos, unsigned char* st_pos) {
    double q[3];
    for (int k = 0; k < p->n_mono; ++k) {
        se3_map(p->kf_pose + 7 * p->mono_kf[k], p->pt_xyz + 3 * p->mono_pt[k], q);
        mono_pos[k] = q[2] > 0.0;
    }
    for (int k = 0; k < p->n_stereo; ++k) {
        se3_map(p->kf_pose + 7 * p->st_kf[k], p->pt_xyz + 3 * p->st_pt[k], q);
        st_pos[k] = q[2] > 0.0;
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* Optimizer::PoseOptimization (reference src/Optimizer.cc:244-456): one free SE3 vertex (the frame), unary edges   */
/* EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose with the map point fixed inside the edge            */
/* (Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:142-214, .cpp:266-358), 4 rounds of optimize(10) with the same  */
/* Levenberg-Marquardt as above, inlier / outlier classification after every round.                                */
/* ------------------------------------------------------------------------------------------------------------ */
static void po_edge(const double* pose, const double* X, const double* K, const double* obs, int stereo, double* e,
                    double* J /* D x 6, may be NULL */) {
    double p[3];
    se3_map(pose, X, p);
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], bf = K[4];
    if (!stereo) {                       /* project2d then cam_project, .cpp:290-296 */
        e[0] = obs[0] - (p[0] / p[2] * fx + cx);
        e[1] = obs[1] - (p[1] / p[2] * fy + cy);
    } else {                             /* .cpp:299-306: float 1/z, DOUBLE bf (the binary edge casts bf to float) */
        const float invz = 1.0f / (float)p[2];
        const double u = p[0] * invz * fx + cx, v = p[1] * invz * fy + cy;
        e[0] = obs[0] - u; e[1] = obs[1] - v; e[2] = obs[2] - (u - bf * invz);
    }
    if (!J) return;
    const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;   /* .cpp:266-288, 335-358 */
    J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
    J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
    J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy;
    J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
    if (stereo) {
        J[12] = J[0] - bf * y * invz_2; J[13] = J[1] + bf * x * invz_2; J[14] = J[2];
        J[15] = J[3]; J[16] = 0; J[17] = J[5] - bf * invz_2;
    }
}

static double po_errors(int n, const double* K, const double* pose, const double* X, const double* obs, const double* info,
                        const unsigned char* stereo, const unsigned char* level, double dm, double ds, double* chi2) {
    double chi = 0, e[3], r0, r1;
    for (int k = 0; k < n; ++k) {
        if (level[k]) continue;
        po_edge(pose, X + 3 * k, K, obs + 3 * k, stereo[k], e, NULL);
        const int D = stereo[k] ? 3 : 2;
        double c = 0;
        for (int d = 0; d < D; ++d) c += e[d] * e[d];
        c *= info[k];
        chi2[k] = c;
        huber(c, stereo[k] ? ds : dm, &r0, &r1);
        chi += r0;
    }
    return chi;
}

/* trace: [4 rounds][10 iterations][3] = chi2, lambda, trials (rows beyond iters[round] untouched) */
int ba_oracle_pose_optimization(int n, const double* K, const double* pose_in, const double* X, const double* obs,
                                const double* info, const unsigned char* stereo, double* pose_out, unsigned char* outlier,
                                double* trace, int* iters) {
    unsigned char* level = calloc(n + 1, 1);
    double* chi2 = calloc(n + 1, sizeof(double));
    double pose[7], bk[7];
    memset(outlier, 0, n);
    if (n < 3) { memcpy(pose_out, pose_in, sizeof(pose)); free(level); free(chi2); return 0; }   /* :368-369 */
    int nBadEdges = 0;
    int robust = 1;
    const double dM = (double)(float)sqrt(5.991), dS = (double)(float)sqrt(7.815);          /* :282-283 */
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;                                       /* :373-374 */
    memcpy(pose, pose_in, sizeof(pose));
    for (int round = 0; round < 4; ++round) {
        memcpy(pose, pose_in, sizeof(pose));            /* vSE3->setEstimate(pFrame->mTcw), :381 */
        const double dm = robust ? dM : 0.0, ds = robust ? dS : 0.0;
        double lambda = 0, ni = 2;
        int nBad = 0, done = 0;
        for (int it = 0; it < 10; ++it) {
            double currentChi = po_errors(n, K, pose, X, obs, info, stereo, level, dm, ds, chi2);
            double tempChi = currentChi;
            const double iniChi = currentChi;
            double H[36], b[6], x[6] = {0, 0, 0, 0, 0, 0};
            memset(H, 0, sizeof(H)); memset(b, 0, sizeof(b));
            for (int k = 0; k < n; ++k) {
                if (level[k]) continue;
                double e[3], J[18], r0, r1;
                po_edge(pose, X + 3 * k, K, obs + 3 * k, stereo[k], e, J);
                const int D = stereo[k] ? 3 : 2;
                double c = 0;
                for (int d = 0; d < D; ++d) c += e[d] * e[d];
                c *= info[k];
                huber(c, stereo[k] ? ds : dm, &r0, &r1);
                const double w = r1 * info[k];
                for (int i = 0; i < 6; ++i) {
                    double sb = 0;
                    for (int d = 0; d < D; ++d) sb += J[6 * d + i] * (-info[k] * e[d]) * r1;
                    b[i] += sb;
                    for (int j = 0; j < 6; ++j) {
                        double sum = 0;
                        for (int d = 0; d < D; ++d) sum += J[6 * d + i] * J[6 * d + j];
                        H[6 * i + j] += w * sum;
                    }
                }
            }
            if (it == 0) {
                double md = 0;
                for (int i = 0; i < 6; ++i) md = fmax(fabs(H[7 * i]), md);
                lambda = 1e-5 * md; ni = 2; nBad = 0;
            }
            double rho = 0;
            int qmax = 0;
            do {
                memcpy(bk, pose, sizeof(pose));
                double A[36];
                memcpy(A, H, sizeof(A));
                for (int i = 0; i < 6; ++i) A[7 * i] += lambda;
                const int ok2 = chol_solve_upper(A, 6, b, x);
                if (ok2) {
                    double dl[7], np[7];
                    ba_se3_exp(x, dl);
                    se3_mul(dl, pose, np);
                    memcpy(pose, np, sizeof(pose));
                }
                tempChi = po_errors(n, K, pose, X, obs, info, stereo, level, dm, ds, chi2);
                if (!ok2) tempChi = DBL_MAX;
                rho = currentChi - tempChi;
                double scale = 1e-3;
                for (int i = 0; i < 6; ++i) scale += x[i] * (lambda * x[i] + b[i]);
                rho /= scale;
                if (rho > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow(2 * rho - 1, 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    memcpy(pose, bk, sizeof(pose));
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (trace) { double* t = trace + 3 * (10 * round + it); t[0] = currentChi; t[1] = lambda; t[2] = qmax; }
            done++;
            if (qmax == 10 || rho == 0) break;
            if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
            if (nBad >= 3) break;
        }
        if (iters) iters[round] = done;
        nBadEdges = 0;
        for (int k = 0; k < n; ++k) {                    /* :384-437 */
            if (outlier[k]) {                            /* e->computeError() at the final pose */
                double e[3];
                po_edge(pose, X + 3 * k, K, obs + 3 * k, stereo[k], e, NULL);
                const int D = stereo[k] ? 3 : 2;
                double c = 0;
                for (int d = 0; d < D; ++d) c += e[d] * e[d];
                chi2[k] = c * info[k];
            }
            const float c = (float)chi2[k];              /* const float chi2 = e->chi2(); */
            if (c > (stereo[k] ? chi2Stereo : chi2Mono)) { outlier[k] = 1; level[k] = 1; nBadEdges++; }
            else { outlier[k] = 0; level[k] = 0; }
        }
        if (round == 2) robust = 0;                      /* e->setRobustKernel(0) */
        if (n < 10) break;                               /* optimizer.edges().size() < 10 */
    }
    memcpy(pose_out, pose, sizeof(pose));
    free(level); free(chi2);
    return n - nBadEdges;
}
