/* stub_qsp.c -- TEST STUB of the path-B C-ABI for the CPU-only check of the shim's graph flattening: qsp_ba_create dumps
 * the scene it receives to $QSP_STUB_DUMP; the "optimiser" returns the input state shifted by a fixed amount so that the
 * write-back path is visible.  It never computes anything. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qsp_hip.h"

struct qsp_ba_problem { qsp_ba_scene s; double *kf, *pt, *ob; };

/* failure injection: $QSP_STUB_FAIL names the entry point that reports QSP_ERR_DEVICE (create | local | optimize | pose) */
static int fails(const char* what) {
    const char* f = getenv("QSP_STUB_FAIL");
    return f && strcmp(f, what) == 0;
}
const char* qsp_last_error(void) { return "injected by the test stub"; }

static void wr(FILE* f, const void* p, size_t n) { if (n) fwrite(p, 1, n, f); }

int qsp_ba_create(const qsp_ba_scene* s, int device, qsp_ba_problem** out) {
    (void)device;
    if (fails("create")) return QSP_ERR_DEVICE;
    const char* path = getenv("QSP_STUB_DUMP");
    if (path) {
        FILE* f = fopen(path, "wb");
        int32_t hdr[6] = {s->n_kf, s->n_pt, s->n_obj, s->n_mono, s->n_stereo, s->n_objedge};
        wr(f, hdr, sizeof(hdr));
        wr(f, s->kf_pose, 56 * (size_t)s->n_kf); wr(f, s->kf_fixed, (size_t)s->n_kf); wr(f, s->kf_id, 8 * (size_t)s->n_kf);
        wr(f, s->kf_K, 40 * (size_t)s->n_kf); wr(f, s->pt_xyz, 24 * (size_t)s->n_pt); wr(f, s->pt_id, 8 * (size_t)s->n_pt);
        wr(f, s->obj_pose, 56 * (size_t)s->n_obj); wr(f, s->obj_id, 8 * (size_t)s->n_obj);
        wr(f, s->mono_pt, 4 * (size_t)s->n_mono); wr(f, s->mono_kf, 4 * (size_t)s->n_mono);
        wr(f, s->mono_obs, 16 * (size_t)s->n_mono); wr(f, s->mono_info, 8 * (size_t)s->n_mono);
        wr(f, s->stereo_pt, 4 * (size_t)s->n_stereo); wr(f, s->stereo_kf, 4 * (size_t)s->n_stereo);
        wr(f, s->stereo_obs, 24 * (size_t)s->n_stereo); wr(f, s->stereo_info, 8 * (size_t)s->n_stereo);
        wr(f, s->objedge_kf, 4 * (size_t)s->n_objedge); wr(f, s->objedge_obj, 4 * (size_t)s->n_objedge);
        wr(f, s->objedge_meas, 56 * (size_t)s->n_objedge); wr(f, &s->objedge_info, 8);
        fclose(f);
    }
    qsp_ba_problem* p = calloc(1, sizeof(*p));
    p->s = *s;
    /* empty vectors arrive as NULL with a zero count (legal on this ABI); memcpy(dst, NULL, 0) is not */
    p->kf = malloc(56 * (size_t)s->n_kf + 8); if (s->n_kf) memcpy(p->kf, s->kf_pose, 56 * (size_t)s->n_kf);
    p->pt = malloc(24 * (size_t)s->n_pt + 8); if (s->n_pt) memcpy(p->pt, s->pt_xyz, 24 * (size_t)s->n_pt);
    p->ob = malloc(56 * (size_t)s->n_obj + 8); if (s->n_obj) memcpy(p->ob, s->obj_pose, 56 * (size_t)s->n_obj);
    *out = p;
    return QSP_OK;
}
void qsp_ba_destroy(qsp_ba_problem* p) { free(p->kf); free(p->pt); free(p->ob); free(p); }
int qsp_ba_local_joint(qsp_ba_problem* p, const volatile uint8_t* stop, qsp_ba_trace* a, qsp_ba_trace* b) {
    (void)stop; (void)a; (void)b;
    if (fails("local")) return QSP_ERR_DEVICE;
    for (int i = 0; i < p->s.n_kf; ++i) if (!p->s.kf_fixed[i]) p->kf[7 * i] += 0.5;      /* visible fake update */
    for (int i = 0; i < p->s.n_pt; ++i) p->pt[3 * i + 1] += 0.25;
    for (int i = 0; i < p->s.n_obj; ++i) p->ob[7 * i + 2] += 0.125;
    return QSP_OK;
}
int qsp_ba_optimize(qsp_ba_problem* p, int32_t n, double a, double b, double c, const volatile uint8_t* s, qsp_ba_trace* t) {
    const char* path = getenv("QSP_STUB_DUMP");
    if (fails("optimize")) return QSP_ERR_DEVICE;
    if (path) {                       /* the call's arguments beside the scene dump: <dump>.args */
        char name[4096];
        snprintf(name, sizeof(name), "%s.args", path);
        FILE* f = fopen(name, "wb");
        double v[4] = {(double)n, a, b, c};
        wr(f, v, sizeof(v));
        fclose(f);
    }
    return qsp_ba_local_joint(p, s, t, t);
}
int qsp_ba_set_levels(qsp_ba_problem* p, const uint8_t* a, const uint8_t* b, const uint8_t* c) { (void)p; (void)a; (void)b; (void)c; return QSP_OK; }
int qsp_ba_get_state(qsp_ba_problem* p, double* kf, double* pt, double* ob) {
    memcpy(kf, p->kf, 56 * (size_t)p->s.n_kf); memcpy(pt, p->pt, 24 * (size_t)p->s.n_pt); memcpy(ob, p->ob, 56 * (size_t)p->s.n_obj);
    return QSP_OK;
}
int qsp_ba_get_edges(qsp_ba_problem* p, double* cm, double* cs, double* co, uint8_t* pm, uint8_t* ps) {
    /* mark the FIRST mono edge as an outlier so that the erase path runs */
    for (int i = 0; i < p->s.n_mono; ++i) { cm[i] = (i == 0) ? 100.0 : 0.0; pm[i] = 1; }
    for (int i = 0; i < p->s.n_stereo; ++i) { cs[i] = 0.0; ps[i] = 1; }
    for (int i = 0; i < p->s.n_objedge; ++i) co[i] = 0.0;
    return QSP_OK;
}

/* pose-only optimisation: dump the flattened correspondences to <dump>.pose, return a visible fake result */
struct qsp_pose_optimizer { int cap; };
int qsp_pose_optimizer_create(int device, int32_t max_points, qsp_pose_optimizer** out) {
    (void)device;
    *out = calloc(1, sizeof(**out));
    (*out)->cap = max_points;
    return QSP_OK;
}
void qsp_pose_optimizer_destroy(qsp_pose_optimizer* h) { free(h); }
int qsp_pose_optimize(qsp_pose_optimizer* h, int32_t n, const double* K, const double* pose_in, const double* X,
                      const double* obs, const double* info, const uint8_t* stereo, double* pose_out, uint8_t* outlier,
                      int32_t* n_inliers, qsp_pose_trace* trace) {
    (void)trace;
    if (n > h->cap) return QSP_ERR_INVALID;
    if (fails("pose")) return QSP_ERR_DEVICE;
    const char* path = getenv("QSP_STUB_DUMP");
    if (path) {
        char name[4096];
        snprintf(name, sizeof(name), "%s.pose", path);
        FILE* f = fopen(name, "wb");
        wr(f, &n, 4); wr(f, K, 40); wr(f, pose_in, 56); wr(f, X, 24 * (size_t)n); wr(f, obs, 24 * (size_t)n);
        wr(f, info, 8 * (size_t)n); wr(f, stereo, (size_t)n);
        fclose(f);
    }
    memcpy(pose_out, pose_in, 56);
    pose_out[1] += 0.75;                                  /* visible fake update */
    for (int i = 0; i < n; ++i) outlier[i] = (uint8_t)(i % 3 == 0);
    *n_inliers = n - (n + 2) / 3;
    return QSP_OK;
}
