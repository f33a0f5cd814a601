/* qsp_hip.h -- C-ABI of the MI355X-native joint object-optimisation hot path of QSP-SLAM.
 *
 * One shared library (libqsp_hip.so) exports every symbol declared here: plain pointers and sizes, int status
 * returns, no exceptions across the boundary, no torch / Eigen / OpenCV types.  Each entry point names the reference
 * interface it replaces (paths relative to the reference root, GetOverMassif/QSP-SLAM).
 *
 * Threading: a handle (decoder, batch, BA problem) may be used by one host thread at a time; distinct handles are
 * independent.  The reference calls both paths from the LocalMapping thread under the GIL (include/System.h:61-75).
 *
 * Path A  (DeepSDF object refinement)  replaces reconstruct/optimizer.py + reconstruct/loss.py + loss_utils.py
 * Path B  (joint bundle adjustment)    replaces src/Optimizer.cc, src/Optimizer_util.cc, include/ObjectPoseGraph.h
 *                                      on top of Thirdparty/g2o (BlockSolver_6_3 + LM + LinearSolverEigen)
 */
#ifndef QSP_HIP_H
#define QSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------------------------
 * status codes
 * ------------------------------------------------------------------------------------------------------------ */
enum {
    QSP_OK = 0,
    QSP_ERR_INVALID = 1,      /* bad argument (null pointer, negative size, inconsistent counts)            */
    QSP_ERR_UNSUPPORTED = 2,  /* decoder architecture outside the supported family (see qsp_decoder_create) */
    QSP_ERR_DEVICE = 3,       /* a HIP call failed; qsp_last_error() holds the text                          */
    QSP_ERR_NO_DEVICE = 4     /* no gfx950 device visible -- there is no CPU fallback                        */
};

const char* qsp_last_error(void);   /* thread-local text of the last failure */
int qsp_version(void);              /* ABI version, currently 1 */
int qsp_device_count(void);

/* ===============================================================================================================
 * Path A -- DeepSDF decoder and object refinement
 * ============================================================================================================ */

/* Decoder description.  Replaces deep_sdf/workspace.py:202-224 (config_decoder) + deep_sdf/deep_sdf_decoder.py:9-110.
 * weight[l] is (out_dim[l], in_dim[l]) row-major.  If weight_g != NULL and weight_g[l] != NULL, weight[l] is the
 * weight-norm direction v and the effective matrix is g * v / ||v||_row (old-style torch weight_norm, dim 0), folded
 * once at creation instead of on every call as the reference does.
 * Supported family: n_layers = 9, hidden width 512, code_len = 64, latent_in_layer = 4 (the DSP-SLAM "8 x 512"
 * decoder: 67->512, 512->512 x2, 512->445, 512->512 x4, 512->1, ReLU, final tanh).  Anything else: QSP_ERR_UNSUPPORTED. */
typedef struct {
    int32_t n_layers;
    int32_t code_len;
    int32_t latent_in_layer;
    const int32_t* in_dim;
    const int32_t* out_dim;
    const float* const* weight;
    const float* const* weight_g;   /* may be NULL */
    const float* const* bias;
} qsp_decoder_desc;

typedef struct qsp_decoder qsp_decoder;

/* Threads.  A decoder owns one HIP stream, one resident refinement batch and a few fields that a call rewrites for its own duration
 * (the f32 override of a range fallback, the screening margin of a self-check repeat).  Every entry point that launches on a
 * decoder -- qsp_decode_sdf*, qsp_sdf_value_grad, qsp_refine_batch_set_state / _run / _get, qsp_reconstruct_objects,
 * qsp_estimate_pose, qsp_refine_detections, qsp_mesh_extract / _from_volume, qsp_decoder_set_option -- takes the decoder's lock:
 * host threads may share a decoder and get correct results, one call at a time.  Threads that should overlap on the GPU use a
 * decoder each. */
int qsp_decoder_create(const qsp_decoder_desc* desc, int device, qsp_decoder** out);
/* Batches and mesh extractors created from a decoder use it until they are destroyed: destroy them first.  (Destroying one of
 * them after its decoder only frees its own memory and is harmless; any other call on it is undefined.) */
void qsp_decoder_destroy(qsp_decoder* dec);

/* Decoder options.  QSP_DEC_OPT_FORWARD_PRECISION selects the arithmetic of the FORWARD-ONLY decoder passes (qsp_decode_sdf,
 * the voxel-grid decode of qsp_mesh_extract, the ray-sample forward pass of the refinement, reconstruct/loss.py:78):
 *   0 (default)  exact float32 multiply-adds on the f32 matrix pipe (v_mfma_f32_32x32x2_f32);
 *   1            every f32 weight and activation as the exact sum of three bf16 values, six bf16 products per multiply-add on
 *                the bf16 matrix pipe with f32 accumulation (v_mfma_f32_32x32x16_bf16): float32-equivalent accuracy (2.2e-7
 *                relative on the SDF value against float64; plain f32: 1.8e-7) at up to 2.67 x the f32 pipe's rate.  Results
 *                differ from mode 0 in the last bits, as two float32 implementations with different summation orders do;
 *   2            every f32 weight and activation as two fp16 values, x = hi + 2^-11 lo' (22 significand bits), three fp16
 *                products per multiply-add on the fp16 matrix pipe with f32 accumulation (v_mfma_f32_32x32x16_f16): the same
 *                float32-equivalent accuracy (1.8e-7) at up to 5.3 x the f32 pipe's rate.  fp16's range is the precondition:
 *                qsp_decoder_set_option returns QSP_ERR_UNSUPPORTED for a decoder with a weight beyond +-65 000, and a call
 *                during which an activation or a back-propagated gradient exceeded 65 504 in magnitude FAILS with
 *                QSP_ERR_UNSUPPORTED instead of returning numbers (its outputs are then undefined; modes 0 and 1 have no such
 *                limit).  A DeepSDF decoder's activations are O(10).
 * QSP_DEC_OPT_JACOBIAN_PRECISION selects the same (0, 1, 2) for the forward+backward pass that builds the Jacobian rows and the
 * normal equations (qsp_sdf_value_grad and the fused kernel of the refinement, reconstruct/loss_utils.py:82-103): that pass has
 * no discrete decision besides the ReLU masks, so modes 1 and 2 move H, b by float32 rounding noise only.
 * QSP_DEC_OPT_TILE_POINTS (64, the default, or 32): points per workgroup tile of the Jacobian / normal-equation kernel of the
 * refinement batches created AFTER the call (the forward pass over the ray samples keeps 64).  A single object of a few
 * thousand surface points fills only tens of the 256 CUs with 64-point tiles; 32-point tiles spread it over twice as many work
 * items at ~0.67 of the time each -- the latency option for the reference's one-object-per-call pattern
 * (reconstruct/optimizer.py:96-281 called from src/LocalMapping_util.cc:705-760).  Exists on the split-fp16 pipe only (both
 * precisions = 2; qsp_refine_batch_create returns QSP_ERR_UNSUPPORTED otherwise).  The partition of the normal-equation
 * partial sums follows the tile size, so results are bit-reproducible per tile size, not across the two.
 * QSP_DEC_OPT_RENDER_SCREENING (value = margin in units of 1e-6 SDF, 0 = off, the default): the ray-sample forward pass of the
 * refinement (reconstruct/loss.py:78) in two passes.  The render term clamps -- a sample with |sdf| >= cut_off contributes an
 * occupancy of exactly 0 or 1 (reconstruct/loss_utils.py:40-48) -- so away from the surface only the SIGN of the value is used.
 * Pass 1 evaluates every sample with one fp16 term per operand (a third of the matrix-pipe work, 128-point tiles); pass 2
 * re-evaluates the samples with |s1| < cut_off + margin on the split-fp16 pipe and overwrites them.  With margin >= the
 * largest |s1 - s3| the result (K, n_valid, H, b, every later iterate) equals the unscreened split-fp16 result BIT FOR BIT;
 * 20 000 (0.02) is > 8 x the largest difference measured over the fixtures and the randomised sweep (profiles/r03_*).  Needs
 * QSP_DEC_OPT_FORWARD_PRECISION = 2 (QSP_ERR_UNSUPPORTED otherwise); margins above 5 x the usual cut_off (50 000) are refused:
 * at that point the second pass covers most samples and the option has no purpose.  The premise is checked on every run: the
 * second pass has both values of each band sample in hand and keeps the largest |s1 - s3|; a run in which it exceeds HALF the
 * margin is repeated from its starting state in one pass (qsp_decoder_get_counter(QSP_DEC_CNT_SCREEN_FALLBACKS) counts them,
 * qsp_refine_profile.screen_max_diff / .screen_fallbacks report the last run), so a decoder whose screening values are worse
 * than the fixtures' costs time, not bits.
 * QSP_DEC_OPT_SCREEN_AUDIT (N, default 100; 0 = off, 1 = every sample): the band check alone looks only where the screening pass
 * put a sample INSIDE the band.  The second pass therefore also re-evaluates a fixed pseudo-random one in N of the samples the
 * screening pass put OUTSIDE (a hash of hypothesis, sample and iteration), folds their |s1 - s3| into screen_max_diff, and counts
 * each one whose split-fp16 value lies inside the cut-off or has the other sign -- a clamp the one-pass result would not have
 * made -- as a hard failure (qsp_refine_profile.screen_audit_failures): the run is repeated in one pass like above.  An audited
 * sample's overwritten value is only ever read through the clamp, so the audit changes no bit of a passing run.
 * QSP_DEC_OPT_DEPTH_STAGING (1, the default: batches of more than 64 rounds of 128-sample tiles over the chip / 2: always / 0: never):
 * a screened run evaluates the ray samples in two depth stages -- indices [0, D/2)
 * of every ray, then [D/2, D) of the rays that have no OPAQUE sample (sdf <= -cut_off: occupancy exactly 1) yet.  Behind an
 * opaque sample the transmittance of reconstruct/loss.py:101 is exactly 0, so those samples' values reach no output (every term
 * they enter is multiplied by that zero; d e / d o of an in-band one is 0 and dropped by the 1e-2 rule, :103-128): they are not
 * evaluated.  ~20 % of the samples on the synthetic scenes; every output bit-identical to 0 (tests/test_gpu_screening.py).  Only
 * a decoder that returns NaN behind a surface would tell the two apart.
 * QSP_DEC_OPT_SCREENING_MIN_SAMPLES (-1, the default, or a count): a run is screened only when its batch holds more ray samples
 * (rays x depth samples, summed over the hypotheses) than this; -1 = more than two rounds of 64-point tiles over the chip.  A
 * batch that fits one round -- one object per call -- is one tile deep either way and faster in one pass.  The result is the
 * same bits whichever way a run goes.  0 = always screen (tests).
 * QSP_DEC_OPT_NARROW_TILE (1 / 0): decoders much smaller than the 8 x 512 shape they are embedded in (e.g. 4 x 256 / code 32: less
 * than half of its multiply-adds) run the NARROW form of the split-fp16 tile by default -- identity slots, all-zero k-slabs and
 * all-zero column blocks are skipped, exactly; 0 evaluates the embedded form at the full shape's cost (measurements, tests);
 * QSP_DEC_CNT_NARROW_TILE says which is in use.  Other pipes (f32, split bf16) always run the embedded form.
 * QSP_DEC_OPT_USE_TANH (0 / 1): NetworkSpecs.use_tanh of deep_sdf/deep_sdf_decoder.py:66-68,92-94 -- a tanh on the output
 * layer in front of the final tanh.
 * QSP_DEC_OPT_RANGE_FALLBACK (1, the default / 0): when a split-fp16 kernel meets an activation or gradient outside fp16's
 * range, 1 re-runs THAT call on the exact-f32 pipe inside the library and returns its result (counted by
 * qsp_decoder_get_counter(QSP_DEC_CNT_RANGE_FALLBACKS)); 0 fails the call with QSP_ERR_UNSUPPORTED as round 2 did. */
enum { QSP_DEC_OPT_FORWARD_PRECISION = 1, QSP_DEC_OPT_JACOBIAN_PRECISION = 2, QSP_DEC_OPT_TILE_POINTS = 3,
       QSP_DEC_OPT_RENDER_SCREENING = 4, QSP_DEC_OPT_USE_TANH = 5, QSP_DEC_OPT_RANGE_FALLBACK = 6,
       QSP_DEC_OPT_SCREENING_MIN_SAMPLES = 7, QSP_DEC_OPT_NARROW_TILE = 8, QSP_DEC_OPT_SCREEN_AUDIT = 9,
       QSP_DEC_OPT_DEPTH_STAGING = 10 };
enum { QSP_DEC_CNT_RANGE_FALLBACKS = 1, QSP_DEC_CNT_ARENA_REUSED = 2, QSP_DEC_CNT_ARENA_CREATED = 3, QSP_DEC_CNT_NARROW_TILE = 4,
       QSP_DEC_CNT_SCREEN_FALLBACKS = 5 };
/* lifetime counters of a decoder: calls that were re-run on the f32 pipe because a value left fp16's range; calls of
 * qsp_reconstruct_objects that refilled the decoder's resident batch / that had to (re)allocate it */
int64_t qsp_decoder_get_counter(qsp_decoder* dec, int32_t counter);
int qsp_decoder_set_option(qsp_decoder* dec, int32_t option, int32_t value);

/* decode_sdf, reconstruct/loss_utils.py:51-79.  Host pointers: code (code_len), xyz (n,3) row-major, sdf_out (n). */
int qsp_decode_sdf(qsp_decoder* dec, const float* code, const float* xyz, int64_t n, float* sdf_out);

/* Diagnostic: the values the FIRST pass of QSP_DEC_OPT_RENDER_SCREENING computes (one fp16 term per operand).  They are not
 * SDF values of the decoder's precision and nothing in the library returns them as such; tests and tools/screen_margin.py use
 * this entry to measure |s1 - s3|, the quantity the screening margin has to cover.  Host pointers. */
int qsp_decode_sdf_screen(qsp_decoder* dec, const float* code, const float* xyz, int64_t n, float* s1_out);

/* get_batch_sdf_jacobian, reconstruct/loss_utils.py:82-103 (out_dim = 1): y (n) and d y / d [code | xyz] (n, code_len+3),
 * without the weight gradients the reference's autograd also accumulates and never uses.  Host pointers. */
int qsp_sdf_value_grad(qsp_decoder* dec, const float* code, const float* xyz, int64_t n, float* y, float* grad);

/* The `optimizer` block of the detector-config JSON (configs/config_*.json:21-41) read by
 * reconstruct/optimizer.py:27-44. */
typedef struct {
    float k1, k2, k3, k4;      /* render, sdf, code-prior, rotation-prior weights                    */
    float b1, b2;              /* Huber thresholds: render, sdf                                       */
    float lr;                  /* learning_rate                                                       */
    float s_damp;              /* scale_damping                                                       */
    float cut_off;             /* cut_off_threshold (SDF -> occupancy)                                */
    int32_t n_iter;            /* joint_optim.num_iterations                                          */
    int32_t n_depth;           /* num_depth_samples (<= 64)                                           */
    int32_t code_len;          /* 64                                                                  */
} qsp_joint_cfg;

/* A resident batch of refinement hypotheses.  An *object* owns the observations of one detection (surface points,
 * rays, depths: the arguments src/LocalMapping_util.cc:585-672 assembles); a *hypothesis* is one call of
 * Optimizer.reconstruct_object (reconstruct/optimizer.py:96-281) on an object from its own initial pose/code -- the
 * reference makes flip_sample_num (=4) such calls per object, serially (LocalMapping_util.cc:705-760).  All
 * hypotheses of a batch advance together, one fused launch sequence per Gauss-Newton iteration, with no host
 * synchronisation inside the iteration loop. */
typedef struct qsp_refine_batch qsp_refine_batch;

/* pts[o]: (n_pts[o],3) row-major camera-frame points; rays[o]: (n_rays[o],3); depth[o]: (n_fg[o]) observed depth of
 * the first n_fg[o] rays (the rest are background rays).  hyp_obj[h] = object index of hypothesis h.  Host pointers;
 * everything is copied to the device here and stays resident. */
int qsp_refine_batch_create(qsp_decoder* dec, const qsp_joint_cfg* cfg, int32_t n_obj,
                            const float* const* pts, const int32_t* n_pts,
                            const float* const* rays, const int32_t* n_rays,
                            const float* const* depth, const int32_t* n_fg,
                            int32_t n_hyp, const int32_t* hyp_obj, qsp_refine_batch** out);
void qsp_refine_batch_destroy(qsp_refine_batch* b);

/* (re)initialise every hypothesis: t_cam_obj (n_hyp,4,4) row-major Sim3 object->camera, code (n_hyp,code_len) or NULL
 * for the zero code (reconstruct/optimizer.py:111-119).  Clears is_good/loss. */
int qsp_refine_batch_set_state(qsp_refine_batch* b, const float* t_cam_obj, const float* code);

/* Run n_iter Gauss-Newton iterations (n_iter <= 0: cfg.n_iter) and wait for completion. */
int qsp_refine_batch_run(qsp_refine_batch* b, int32_t n_iter);

/* Results per hypothesis: t_cam_obj_out (n_hyp,4,4), code_out (n_hyp,code_len), loss_out (n_hyp), is_good_out (n_hyp).
 * is_good = 0 reproduces the reference's early exits (fewer than 10 ray samples in the unit ball, NaN loss):
 * the pose/code outputs of such a hypothesis are unspecified, loss is that of the last completed iteration. */
int qsp_refine_batch_get(qsp_refine_batch* b, float* t_cam_obj_out, float* code_out, float* loss_out,
                         uint8_t* is_good_out);

/* Introspection of the LAST iteration run, for parity tests: H (n_hyp,71,71), rhs (n_hyp,71), dx (n_hyp,71),
 * n_valid (n_hyp) ray samples inside the unit ball, n_render (n_hyp) render rows K, loss terms (n_hyp,2) =
 * (sdf, render).  Any pointer may be NULL. */
int qsp_refine_batch_trace(qsp_refine_batch* b, float* H, float* rhs, float* dx, int32_t* n_valid,
                           int32_t* n_render, float* loss_terms);
/* ... and the rotation prior's own terms of that iteration (reconstruct/loss.py:155-178): rot4 (n_hyp,4) = J_rot (the three
 * rotation entries 3..5 of J_sim3) and res_rot -- lets a parity test separate the decoder part of rhs from the k4-weighted one. */
int qsp_refine_batch_trace_rot(qsp_refine_batch* b, float* rot4);

/* Parity-test tap.  enable != 0 (with NULL outputs) before a run makes the fused kernel also store every augmented Jacobian
 * row it feeds to the normal equations: [d e/d xi (7) | d e/d code (64) | robust residual] = 72 floats per row.  After
 * the run, a second call with outputs copies hypothesis `hyp`: rows_sdf (n_pts,72), rows_render (n_render,72) of the LAST
 * iteration.  Reference quantities: jac_toc / jac_code / robust residual of reconstruct/loss.py:22-43,143-150 and
 * loss_utils.py:250-265.  enable == 0 frees the tap buffer. */
int qsp_refine_batch_rows(qsp_refine_batch* b, int enable, int32_t hyp, float* rows_sdf, float* rows_render);

/* Timing of the last qsp_refine_batch_run, measured with HIP events on the library's own stream. */
typedef struct {
    float ms_total;          /* first launch -> last kernel complete                                  */
    float ms_mlp_jtj;        /* sum over launches of the fused MLP fwd+bwd+JtJ kernel                */
    float ms_mlp_fwd;        /* sum over launches of the forward-only MLP kernel (ray samples)       */
    float ms_other;          /* ray sampling, render scan, reduce+solve                              */
    int32_t n_launch_jtj, n_launch_fwd;
    int64_t pts_jtj;         /* points through fwd+bwd (sdf points + render rows), all iterations     */
    int64_t pts_fwd;         /* points through forward only (valid ray samples), all iterations       */
    int64_t tiles_jtj, tiles_fwd;   /* 64-point tiles actually executed (incl. padding rows)          */
    int64_t pts_band;        /* screened forward pass: samples re-evaluated by the second pass (0 when off) */
    int32_t range_fallbacks; /* 1 if this run was repeated on the f32 pipe (QSP_DEC_OPT_RANGE_FALLBACK)     */
    int32_t screen_fallbacks;/* 1 if this run was repeated in one pass (screening self-check, QSP_DEC_OPT_RENDER_SCREENING) */
    float screen_max_diff;   /* largest |s1 - s3| the second pass saw on a band or audited sample (0 when not screened)      */
    int32_t screen_audit_failures; /* audited OUT-of-band samples whose split-fp16 value was inside the cut-off or of the other
                                * sign: > 0 = the screened attempt was wrong somewhere and the run was repeated in one pass    */
    int64_t pts_audit;       /* out-of-band samples the second pass re-evaluated as the audit (QSP_DEC_OPT_SCREEN_AUDIT)      */
} qsp_refine_profile;
int qsp_refine_batch_profile(qsp_refine_batch* b, int enable, qsp_refine_profile* out);

/* One-shot entry point with the reference's per-call semantics, batched over hypotheses: fill + set_state + run + get on a
 * batch that stays resident with the decoder (sized by the high-water mark of the calls so far: no device allocation per call
 * once it has settled; released by qsp_decoder_destroy).  Optimizer.reconstruct_object, reconstruct/optimizer.py:96-281, as
 * src/LocalMapping_util.cc:705-760 calls it -- one object (x its yaw flips) per call.  Results do not depend on what the
 * resident batch held before. */
int qsp_reconstruct_objects(qsp_decoder* dec, const qsp_joint_cfg* cfg, int32_t n_obj,
                            const float* const* pts, const int32_t* n_pts,
                            const float* const* rays, const int32_t* n_rays,
                            const float* const* depth, const int32_t* n_fg,
                            int32_t n_hyp, const int32_t* hyp_obj, const float* t_cam_obj, const float* code,
                            float* t_cam_obj_out, float* code_out, float* loss_out, uint8_t* is_good_out);

/* Optimizer.estimate_pose_cam_obj, reconstruct/optimizer.py:47-93, batched: per item an SE3 t_co (4,4), a scale, surface
 * points and a code; n_iter SDF-only Gauss-Newton iterations on the 6 pose dimensions (1e-2 damping, raw residual,
 * inlier filter |res| <= 0.05 after iteration index 4).  t_co_out (n,4,4) SE3.  Host pointers. */
int qsp_estimate_pose(qsp_decoder* dec, int32_t n, const float* t_co_se3, const float* scale,
                      const float* const* pts, const int32_t* n_pts, const float* code, int32_t n_iter,
                      float* t_co_out);

/* ---------------------------------------------------------------------------------------------------------------
 * Mesh extraction: MeshExtractor.extract_mesh_from_code, reconstruct/optimizer.py:284-304
 * = decode_sdf over the voxel grid of create_voxel_grid (reconstruct/utils.py:98-117) + marching cubes at level 0
 * (convert_sdf_voxels_to_mesh, reconstruct/utils.py:120-141, skimage.measure.marching_cubes_lewiner there).
 * Both stages run on the device; only vertices and faces come back.
 *   voxel_points (dim^3, 3): the grid in the reference's order, point index = i0*dim^2 + i1*dim + i2 (host pointer, copied).
 *   Marching cubes = Lewiner's (Lewiner et al., JGT 2003) as scikit-image 0.18's marching_cubes_lewiner runs it -- the call of
 *   reconstruct/utils.py:131 --, reproduced on the device with its case tables (csrc/mesh_lewiner.hpp): one vertex per
 *   sign-changing grid edge (+ the extra vertex of some ambiguous configurations), numbered by first use in a z-outermost sweep
 *   of the cells; faces in that sweep's order, corners reversed (gradient_direction='descent').  Same vertices, same faces, same
 *   ORDER as scikit-image on the same volume (tests/golden/mc_lewiner_*.npz hold its output).
 *   vertices: qsp_mesh_fetch gives (V,3) float32 = float32(index coordinate) * float32(2/(dim-1)) - 1; qsp_mesh_fetch_f64 gives
 *   the reference's own float64 values, float32(index coordinate) * (2.0/(dim-1)) + (-1.0), bit for bit.
 *   qsp_mesh_extractor_set_method(m, 1) selects the triangulation of rounds 2-3 instead (face-consistent segments from a
 *   generated 256-case table, vertices ordered by owning grid point then axis: the same vertex set, other diagonals). */
typedef struct qsp_mesh_extractor qsp_mesh_extractor;
int qsp_mesh_extractor_create(qsp_decoder* dec, int32_t voxels_dim, const float* voxel_points, qsp_mesh_extractor** out);
void qsp_mesh_extractor_destroy(qsp_mesh_extractor* m);
/* decode the volume for `code` (64 floats) and run marching cubes; results stay on the device, counts are returned */
int qsp_mesh_extract(qsp_mesh_extractor* m, const float* code, int64_t* n_verts, int64_t* n_faces);
/* marching cubes on a caller-supplied (dim,dim,dim) volume (convert_sdf_voxels_to_mesh alone) */
int qsp_mesh_from_volume(qsp_mesh_extractor* m, const float* sdf_volume, int64_t* n_verts, int64_t* n_faces);
/* copy the last result out: verts (n_verts,3), faces (n_faces,3), optionally the (dim^3) volume; any pointer may be NULL */
int qsp_mesh_fetch(qsp_mesh_extractor* m, float* verts, int32_t* faces, float* sdf_volume);
/* the last result's vertices as float64 (n_verts,3), as convert_sdf_voxels_to_mesh returns them */
int qsp_mesh_fetch_f64(qsp_mesh_extractor* m, double* verts);
/* 0 (default): Lewiner's marching cubes, what the reference calls; 1: the face-consistent table of rounds 2-3 */
int qsp_mesh_extractor_set_method(qsp_mesh_extractor* m, int32_t method);
/* method 1's generated case table: ntri[256], tri[256][24] cube-edge ids (edge = 4*axis + u + 2v), -1 padded */
int qsp_mc_tables(int8_t* ntri, int8_t* tri);

/* ===============================================================================================================
 * Path B -- joint bundle adjustment (camera poses, map points, object poses)
 * ============================================================================================================ */

/* The g2o graph that Optimizer::{Local,}JointBundleAdjustment (src/Optimizer_util.cc:44-307,309-771) and
 * Optimizer::{Local,}BundleAdjustment (src/Optimizer.cc:54-242,458-783) build with heap-allocated vertices and edges,
 * flattened into arrays.  Poses are SE3Quat values laid out (tx ty tz qx qy qz qw) -- T_cw for key-frames
 * (VertexSE3Expmap, estimate = Converter::toSE3Quat(pKF->GetPose())), T_ow for objects (pMO->SE3Tow).
 * kf_id / pt_id / obj_id are the g2o VERTEX ids the reference assigns (mnId, mnId + maxKFid + 1,
 * mnId + maxKFid + maxMPid + 2): they define the order of the unknowns (hessian indices), which this library
 * reproduces exactly.  Edge arrays keep the caller's order; per-edge outputs come back in that order.
 * All arrays are copied at creation. */
typedef struct {
    int32_t n_kf, n_pt, n_obj, n_mono, n_stereo, n_objedge;
    const double* kf_pose;        /* [n_kf][7]                                                             */
    const uint8_t* kf_fixed;      /* [n_kf]  vSE3->setFixed(...)                                           */
    const int64_t* kf_id;         /* [n_kf]                                                                */
    const double* kf_K;           /* [n_kf][5]  fx fy cx cy bf (pKF->fx ... pKF->mbf)                      */
    const double* pt_xyz;         /* [n_pt][3]  VertexSBAPointXYZ, marginalised                            */
    const int64_t* pt_id;
    const double* obj_pose;       /* [n_obj][7]                                                            */
    const int64_t* obj_id;
    const int32_t* mono_pt;       /* EdgeSE3ProjectXYZ: vertex 0 = point, vertex 1 = key-frame             */
    const int32_t* mono_kf;
    const double* mono_obs;       /* [n_mono][2]  kpUn.pt                                                  */
    const double* mono_info;      /* [n_mono]     invSigma2 (information = invSigma2 * I)                  */
    const int32_t* stereo_pt;     /* EdgeStereoSE3ProjectXYZ                                               */
    const int32_t* stereo_kf;
    const double* stereo_obs;     /* [n_stereo][3]  u v u_right                                            */
    const double* stereo_info;
    const int32_t* objedge_kf;    /* EdgeSE3LieAlgebra (include/ObjectPoseGraph.h:57-89): vertex 0 = key-frame */
    const int32_t* objedge_obj;   /*                                                      vertex 1 = object   */
    const double* objedge_meas;   /* [n_objedge][7]  Z = det->SE3Tco                                        */
    double objedge_info;          /* information = objedge_info * I6 (1e3 in the reference)                 */
} qsp_ba_scene;

/* Per-iteration record of one optimize() call (G2OBatchStatistics-like, block_solver.hpp:441-453): the caller provides the
 * arrays with capacity `cap`. */
typedef struct {
    int32_t cap, n;
    double* chi2;            /* robust chi2 after the iteration                                             */
    double* lambda;          /* LM damping after the iteration                                              */
    int32_t* trials;         /* LM trials spent in the iteration (<= 10)                                    */
    int32_t* accepted;       /* 1 if the last trial was accepted                                            */
    int32_t result;          /* 0 ran all iterations, 1 terminated by the LM stop rules, 2 stop flag         */
    int32_t iterations;      /* iterations done                                                              */
    int32_t n_pose_blocks;   /* free key-frames + objects (reduced system = 6 x this)                        */
    int32_t n_landmarks;     /* active points                                                                */
} qsp_ba_trace;

typedef struct {
    float ms_total;          /* whole optimize() call on the GPU stream                                      */
    float ms_linearize;      /* sum over linearisations of the three J^T W J kernels (k_lin_*)               */
    float ms_schur, ms_solve, ms_update;
    int32_t n_linearize, n_trials;
    int64_t bytes_linearize; /* ALGORITHMIC bytes of one linearisation: 176 B/mono edge, 184 B/stereo edge,
                                392 B/free pose or object, 120 B/point, 352 B/object edge */
    int32_t cholesky_chain;  /* 1: the dense factorisation runs as one launch -- a resident chain workgroup and tile workgroups
                                beside it (QSP_BA_OPT_CHOLESKY_CHAIN), 0: one launch per block step                  */
    int32_t chain_timeouts;  /* solves of this problem in which a flag wait of that launch expired: the trial was repeated on
                                the one-launch-per-step form, which the problem keeps from there on (0 in a healthy run)  */
    int32_t boundary_device; /* qsp_ba_local_joint calls of this problem whose stage boundary (outlier classification, re-index,
                                the second stage's first system) ran on the device behind the first stage's last trial ...  */
    int32_t boundary_host;   /* ... and those that took the host path: the last trial was rejected, a key-frame or object
                                vertex lost its last active edge (the reduced system changes shape), or several ranks       */
} qsp_ba_stats;

typedef struct qsp_ba_problem qsp_ba_problem;

int qsp_ba_create(const qsp_ba_scene* scene, int device, qsp_ba_problem** out);
void qsp_ba_destroy(qsp_ba_problem* p);
/* qsp_ba_destroy keeps what is expensive to make for the next problem on the same device: the problem's streams (a hardware queue
 * costs milliseconds to bring up), up to 16 device chunks / 512 MB and its pinned staging buffers.  A caller that builds a problem
 * per bundle adjustment -- Optimizer::LocalJointBundleAdjustment, src/Optimizer_util.cc:309-771, as LocalMapping calls it --
 * pays 1.3 ms instead of 7 ms per create + destroy.  qsp_ba_release_caches() gives everything back (any thread, no problem of
 * the process may be inside a call). */
void qsp_ba_release_caches(void);

/* g2o edge levels: 1 = excluded from optimisation (e->setLevel(1), src/Optimizer_util.cc:621-654).  NULL = all active. */
int qsp_ba_set_levels(qsp_ba_problem* p, const uint8_t* mono, const uint8_t* stereo, const uint8_t* objedge);

/* optimizer.initializeOptimization(0); optimizer.optimize(n_iter) -- Levenberg-Marquardt with Schur complement over the
 * points (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-164, block_solver.hpp:354-486).
 * delta_* are the Huber deltas of RobustKernelHuber (<= 0: no kernel).  stop_flag mirrors setForceStopFlag(pbStopFlag):
 * polled before every iteration and LM trial; may be NULL.  trace may be NULL. */
int qsp_ba_optimize(qsp_ba_problem* p, int32_t n_iter, double delta_mono, double delta_stereo, double delta_obj,
                    const volatile uint8_t* stop_flag, qsp_ba_trace* trace);

/* The two-stage schedule of Optimizer::LocalJointBundleAdjustment (src/Optimizer_util.cc:598-661): optimize(5) with Huber
 * sqrt(5.991)/sqrt(7.815)/sqrt(1e3); edges with chi2 > 5.991 / 7.815 / 1e3 or non-positive depth go to level 1; kernels
 * dropped; optimize(10).  Returns early (state of stage 1 kept) when *stop_flag is set, as the reference does. */
int qsp_ba_local_joint(qsp_ba_problem* p, const volatile uint8_t* stop_flag, qsp_ba_trace* stage1, qsp_ba_trace* stage2);

int qsp_ba_set_state(qsp_ba_problem* p, const double* kf_pose, const double* pt_xyz, const double* obj_pose);
int qsp_ba_get_state(qsp_ba_problem* p, double* kf_pose, double* pt_xyz, double* obj_pose);

/* Per-edge chi2 = e^T Omega e of the LAST error evaluation (what e->chi2() returns in the reference's outlier checks,
 * which may belong to a rejected LM trial -- reproduced), and isDepthPositive() from the current estimates. */
int qsp_ba_get_edges(qsp_ba_problem* p, double* mono_chi2, double* stereo_chi2, double* objedge_chi2,
                     uint8_t* mono_depth_positive, uint8_t* stereo_depth_positive);

/* hessianIndex of every vertex as g2o's buildIndexMapping assigns it for the last optimize() call (-1: fixed or
 * inactive): free key-frames and objects in ascending vertex id, then points in ascending vertex id. */
int qsp_ba_get_index(qsp_ba_problem* p, int32_t* kf_hidx, int32_t* obj_hidx, int32_t* pt_hidx);

int qsp_ba_profile(qsp_ba_problem* p, int enable, qsp_ba_stats* out);

/* Multi-GPU: shard ONE scene's landmarks over `world` ranks (one process per GPU, every rank created from the same scene).
 * Rank r linearises and marginalises the landmarks with pt_id % world == r; the camera-object edges (a few thousand at
 * most) are linearised by every rank and counted by rank 0; the shared camera block is combined with ONE sum all-reduce of
 * the reduced system (dimp^2 + dimp doubles, dimp = 6 x free key-frames rounded up to 64) per Levenberg-Marquardt trial,
 * plus three tiny ones (pose blocks after linearisation, chi2 and rho scalars); every rank then solves the reduced system
 * redundantly and updates its own landmarks.  `fn` must sum
 * `count` doubles at `device_buf` in place over all ranks and be complete (or ordered on `hip_stream`) when it returns;
 * with torch.distributed this is all_reduce on RCCL ("nccl" backend) over xGMI -- see qsp_slam_amd/parallel.py.
 * The reference has no counterpart (single process). */
typedef int (*qsp_allreduce_fn)(void* ctx, double* device_buf, int64_t count, void* hip_stream);
int qsp_ba_set_shard(qsp_ba_problem* p, int32_t rank, int32_t world, qsp_allreduce_fn fn, void* ctx);

/* The same sharding with the collectives issued by the library itself: ncclAllReduce(ncclDouble, ncclSum) on the
 * problem's own HIP stream, in place on the device buffers -- no host synchronisation per collective, the stream orders
 * each reduction between the kernels that write and read its buffer.  `nccl_comm` is an ncclComm_t of `world` ranks (one per
 * GPU) that the caller owns: one made by qsp_comm_create below, or the embedding application's own.  Per LM iteration: one
 * reduction of the pose blocks + b_p + chi2 (36 n_pose + dim + 1 doubles) and, in the first iteration, a 1-double MAX for
 * lambda's initial value; per LM trial: the reduced system (dimp^2 + dimp doubles) and two scalars (chi2, rho
 * denominator); per optimize() call: the landmarks (3 n_pt doubles).  Sums are re-associated across ranks, so a sharded
 * solve follows the unsharded one to ~1e-9 relative, not bit for bit.  world == 1 (comm ignored) switches sharding off. */
int qsp_ba_set_shard_rccl(qsp_ba_problem* p, int32_t rank, int32_t world, void* nccl_comm);

/* ---------------------------------------------------------------------------------------------------------------
 * RCCL communicator owned by the library (one process per GPU; xGMI inside a node).  librccl.so.1 is resolved at run time
 * -- the copy already mapped into the process (PyTorch ships one) or the ROCm installation's -- so the library has no
 * link-time dependency on it and single-GPU users never load it.
 *   qsp_comm_unique_id: ncclGetUniqueId on ONE rank; the 128 bytes travel to the other ranks by any side channel
 *                       (torch.distributed broadcast, a file, MPI).
 *   qsp_comm_create:    ncclCommInitRank on `device`; collective over all ranks.
 *   qsp_comm_adopt:     wrap an ncclComm_t the application already has (not destroyed by qsp_comm_destroy).
 *   qsp_comm_allreduce_f64 / qsp_comm_allgather_f32: in-place SUM of `count` doubles / gather of `count_per_rank` floats
 *                       per rank into recv (world x count_per_rank) on `hip_stream` (NULL: the null stream); asynchronous. */
#define QSP_COMM_ID_BYTES 128
typedef struct qsp_comm qsp_comm;
int qsp_comm_unique_id(uint8_t* id_out /* [QSP_COMM_ID_BYTES] */);
int qsp_comm_create(const uint8_t* id /* [QSP_COMM_ID_BYTES] */, int32_t rank, int32_t world, int device, qsp_comm** out);
int qsp_comm_adopt(void* nccl_comm, int32_t rank, int32_t world, int device, qsp_comm** out);
void qsp_comm_destroy(qsp_comm* c);
void* qsp_comm_nccl(qsp_comm* c);     /* the ncclComm_t, for qsp_ba_set_shard_rccl */
/* librccl is resolved at run time: QSP_RCCL_LIB=<path> if set (an explicit library wins; a path that does not load is an error),
 * else the copy already mapped into the process, else librccl.so.1.  Test hook for the shared-memory stand-in of tests/stub_rccl
 * (which lets a one-GPU box execute the RCCL-on-stream path with two ranks): collectives that ran with world > 1 on this
 * communicator, out3 = [sum all-reduces, max all-reduces, all-gathers]; QSP_ERR_UNSUPPORTED with a real RCCL. */
int qsp_comm_stub_counts(qsp_comm* c, int64_t* out3);
int32_t qsp_comm_rank(qsp_comm* c);
int32_t qsp_comm_world(qsp_comm* c);
int qsp_comm_allreduce_f64(qsp_comm* c, double* device_buf, int64_t count, void* hip_stream);
int qsp_comm_allgather_f32(qsp_comm* c, const float* send, float* recv, int64_t count_per_rank, void* hip_stream);

/* Reproducible mode: the Schur complement is accumulated without atomics, every sum in a fixed order (per pair of
 * key-frames over their common landmarks in landmark order, pair lists built on the host from sum_l k_l (k_l+1)/2 entries,
 * 144 B per edge of extra device storage), so repeated runs give the same bits.  It is the DEFAULT whenever the lists stay
 * below 4 M entries (C2, C4, C5 of BASELINE.json all do; as fast as the atomic kernels there); larger graphs fall back to
 * FP64 atomics (run-to-run spread of the final chi2 3e-15 .. 8e-9 relative).  on = 1 forces it
 * (QSP_ERR_UNSUPPORTED when the graph is too large), on = 0 selects the atomic kernels. */
int qsp_ba_set_deterministic(qsp_ba_problem* p, int on);

/* Solver options.  QSP_BA_OPT_OBJECT_ELIMINATION (default 1): the object vertices -- connected to key-frames only, so their part
 * of the reduced system is block diagonal -- are eliminated in closed form in front of the dense factorisation, which then
 * covers the free key-frames only (what the fill-reducing ordering of the reference's sparse LDLT achieves,
 * Thirdparty/g2o/g2o/solvers/linear_solver_eigen.h:147-201).  0 keeps objects inside the dense system (same solution to
 * rounding; A/B tests).  Takes effect at the next optimize() call.
 * QSP_BA_OPT_CHOLESKY_CHAIN (default 1 where the reduced system has more than one block row): the blocked Cholesky of the reduced
 * system (linear_solver_eigen.h:94-124) as ONE launch on the problem's stream: the first workgroup to start carries the serial
 * chain -- update of the next diagonal block, its factorisation, the forward substitution -- and never leaves its compute unit, the
 * others take the remaining tiles by ticket and keep each in registers through all its steps; they meet through flags in device
 * memory.  A tile waits only for the chain and for tiles with smaller tickets, which running workgroups hold, so the launch makes
 * progress however few of its workgroups are resident.  Waits are bounded all the same: if one expires (compute units withheld
 * for ~1-2 s) the trial is repeated on the one-launch-per-step form -- same operations, same order, same bits -- and the problem
 * stays on it (qsp_ba_stats.chain_timeouts counts; a line on stderr).  QSP_BA_CHOL=steps in the environment selects that form
 * outright.  Value 2 is for tests: the chain is launched WITHOUT its tile workgroups, so its first wait must expire (about a
 * second); the call must neither hang nor fail, and must give the step form's bits. */
enum { QSP_BA_OPT_OBJECT_ELIMINATION = 1, QSP_BA_OPT_CHOLESKY_CHAIN = 2 };
int qsp_ba_set_option(qsp_ba_problem* p, int32_t option, int32_t value);

/* ---------------------------------------------------------------------------------------------------------------
 * Pose-only optimisation: Optimizer::PoseOptimization(Frame*), src/Optimizer.cc:244-456 --
 * one free SE3 vertex, one unary edge per matched map point (EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose),
 * 4 rounds of optimize(10) with inlier / outlier re-classification (chi2 5.991 / 7.815), robust kernel off from round 3.
 * The whole procedure is one kernel launch.  Host pointers.
 *   K (5) fx fy cx cy bf; pose (7) tx ty tz qx qy qz qw of T_cw = Converter::toSE3Quat(pFrame->mTcw);
 *   X (n,3) world points; obs (n,3) u v u_right (third entry ignored where stereo[i] == 0); info (n) invSigma2;
 *   outlier (n) out = pFrame->mvbOutlier of the matched points; *n_inliers = nInitialCorrespondences - nBad (0 and the
 *   pose unchanged when n < 3, :368-369). */
typedef struct qsp_pose_optimizer qsp_pose_optimizer;
typedef struct {
    int32_t iters[4];          /* LM iterations run in each round                       */
    double trace[4][10][3];    /* chi2, lambda, trials after each (round, iteration)    */
} qsp_pose_trace;
int qsp_pose_optimizer_create(int device, int32_t max_points, qsp_pose_optimizer** out);
void qsp_pose_optimizer_destroy(qsp_pose_optimizer* h);
int qsp_pose_optimize(qsp_pose_optimizer* h, int32_t n, const double* K, const double* pose_in, const double* X,
                      const double* obs, const double* info, const uint8_t* stereo, double* pose_out, uint8_t* outlier,
                      int32_t* n_inliers, qsp_pose_trace* trace);

/* ---------------------------------------------------------------------------------------------------------------
 * Caller-side marshalling on the device: what LocalMapping::ProcessDetectedObjects does
 * around Optimizer.reconstruct_object for every detection of a key frame, src/LocalMapping_util.cc:585-760 --
 *   surface_points_cam = Rcw * x3Dw + tcw over the object's map points                               (:610-628)
 *   depth_obs = z of the same transform over the detection's feature points, ray = invK * (u, v, 1)  (:634-669)
 *   rays = [fg_rays ; background_rays]                                                               (:671-672)
 *   hypothesis k: t_cam_obj = SE3Tcw * Sim3Two with the rotation block of Sim3Two right-multiplied by
 *   AngleAxisf(k * flip_sample_angle, e_y); k = 0 only when the object already has a good orientation (:706-733)
 *   keep the first result, replace it when it is not good or when the new one is good with a smaller loss (:748-752)
 * -- for ALL detections in one call: world-frame inputs go up once, three small kernels assemble the resident batch and
 * the initial states, the Gauss-Newton iterations run as in qsp_refine_batch_run, one kernel applies the selection rule, and
 * only the kept pose / code / loss per detection come back.  The filters on MapPoint flags (isBad, isOutlier, object_id:
 * :612-617,640-647) are pointer chasing over the map and stay with the caller: the arrays hold the points that passed.
 * Host pointers; ragged arrays are concatenated, *_off has n_det + 1 entries. */
typedef struct {
    int32_t n_det;
    const float* T_cw;         /* (n_det,4,4) row-major SE3 pose of the detection's key frame (KeyFrame::GetPose)        */
    const float* K;            /* (n_det,4) fx fy cx cy (Tracking::GetCameraIntrinsics)                                  */
    const float* T_wo;         /* (n_det,4,4) MapObject::Sim3Two                                                         */
    const float* code;         /* (n_det,code_len) MapObject::vShapeCode, or NULL for zero codes                         */
    const int32_t* n_flip;     /* (n_det) 1 if MapObject::findGoodOrientation else flip_sample_num; NULL = all 1         */
    double flip_angle;         /* flip_sample_angle = 2 pi / flip_sample_num (src/LocalMapping.cc:76)                    */
    const int32_t* pts_off;    /* map points on the object, world frame                                                  */
    const float* pts_world;    /* (pts_off[n_det],3)                                                                     */
    const int32_t* fg_off;     /* feature points of the detection that lie on the object                                 */
    const float* fg_px;        /* (fg_off[n_det],2) undistorted key-point pixel (mvKeysUn[idx].pt)                       */
    const float* fg_world;     /* (fg_off[n_det],3) world position of the matched map point                              */
    const int32_t* bg_off;     /* background rays of the detection (ObjectDetection::background_rays)                    */
    const float* bg_rays;      /* (bg_off[n_det],3)                                                                      */
} qsp_detections;

typedef struct {               /* any pointer may be NULL */
    float* t_cam_obj;          /* (n_det,4,4) Sim3Tco of the kept result (unspecified where is_good == 0)                */
    float* code;               /* (n_det,code_len)                                                                       */
    float* loss;               /* (n_det) loss of the kept result                                                        */
    uint8_t* is_good;          /* (n_det) 0: the reference's t_cam_obj would be None                                     */
    int32_t* kept_flip;        /* (n_det) index k of the kept hypothesis                                                 */
    float* losses;             /* (sum n_flip) loss of every hypothesis, detection-major (the "# Losses" line, :755-759) */
    /* parity taps: the assembled inputs as reconstruct_object would have received them */
    float* pts_cam;            /* (pts_off[n_det],3)                                                                     */
    float* rays;               /* (fg_off[n_det] + bg_off[n_det], 3), per detection foreground then background           */
    float* depth_obs;          /* (fg_off[n_det])                                                                        */
    float* t_cam_obj_init;     /* (sum n_flip,4,4)                                                                       */
} qsp_detection_results;

int qsp_refine_detections(qsp_decoder* dec, const qsp_joint_cfg* cfg, const qsp_detections* det,
                          qsp_detection_results* out);

/* ---------------------------------------------------------------------------------------------------------------
 * Single-ellipsoid fits, batched: EllipsoidExtractor::OptimizeEllipsoidUsingPlanes,
 * src/pca/EllipsoidExtractorLocalOptimization.cpp:16-85 -- per ellipsoid one VertexEllipsoidXYZABC (translation and half-axes
 * free, rotation fixed) and one unary EdgeEllipsoidPlane per plane (error = distance from the plane to the nearest tangent
 * point, src/pca/EllipsoidExtractorEdges.cpp:35-175; information 1; g2o's numeric Jacobian, delta 1e-9), dense
 * Levenberg-Marquardt, optimize(n_iter) (reference: 10).  One wave per ellipsoid, all ellipsoids in one launch.
 *   ellipsoid_in/out (n,10): translation (3), quaternion x y z w (4), half-axes (3) = g2o::ellipsoid::toVector()
 *   planes (plane_off[n],4): A B C D of the planes of ellipsoid i at [plane_off[i], plane_off[i+1]); no planes: unchanged
 *   normal_direction != 0: the residual of EdgeSE3EllipsoidPlane with setNormalDirection(true) and an identity camera
 *   (GetDistanceWithDirection + its NaN -> 0 rule, EllipsoidExtractorEdges.cpp:151-226) instead of EdgeEllipsoidPlane's
 *   chi2_out (n), iters_out (n), trace (n,n_iter,3) chi2 / lambda / trials per iteration: optional.  Host pointers. */
int qsp_ellipsoid_fit_planes(int device, int32_t n, const double* ellipsoid_in, const int32_t* plane_off,
                             const double* planes, int32_t n_iter, int32_t normal_direction, double* ellipsoid_out,
                             double* chi2_out, int32_t* iters_out, double* trace);

/* The OTHER single-ellipsoid problem of the reference: priorInfer::infer, src/core/PriorInfer.cpp:331-427 -- one
 * VertexEllipsoidXYZABCYaw (7 unknowns: translation in the ellipsoid's frame, half-axes, yaw; include/core/BasicEllipsoidEdges.h:46,
 * src/core/Ellipsoid.cpp:78-106), a fixed identity camera, and per ellipsoid
 *   planes_normal [off_normal[i], off_normal[i+1]): EdgeSE3EllipsoidPlaneWithNormal (2-D: nearest tangent distance, smallest angle
 *       between the plane normal and an ellipsoid axis; information diag(1, 1 / sigma^2) w^2; Huber delta 1),
 *       src/pca/EllipsoidExtractorEdges.h:52, .cpp:297-375;
 *   planes [off_plane[i], off_plane[i+1]): EdgeSE3EllipsoidPlane with setNormalDirection(true) (1-D; information w^2; Huber);
 *   pri (n,2), weight (n): EdgePri, error = (mid / min, max / min of the |half-axes|) - pri, information weight^2, no kernel;
 *   ground_plane_weight (n) or NULL: w of the FIRST plane of each of the two lists (bUseGroundPlaneWeight), otherwise w = 1;
 * g2o's numeric Jacobians (delta 1e-9), dense Levenberg-Marquardt, optimize(n_iter) (reference: 10).  One wave per ellipsoid.
 * Outputs as qsp_ellipsoid_fit_planes (the quaternion changes here).  Host pointers. */
int qsp_ellipsoid_fit_prior(int device, int32_t n, const double* ellipsoid_in, const int32_t* off_normal, const double* planes_normal,
                            const int32_t* off_plane, const double* planes, const double* pri, const double* weight,
                            const double* ground_plane_weight, double angle_sigma_deg, int32_t n_iter, double* ellipsoid_out,
                            double* chi2_out, int32_t* iters_out, double* trace);

#ifdef __cplusplus
}
#endif
#endif /* QSP_HIP_H */
