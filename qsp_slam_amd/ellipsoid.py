"""Batched single-ellipsoid plane fits over the C-ABI (qsp_ellipsoid_fit_planes, include/qsp_hip.h): the reference's
EllipsoidExtractor::OptimizeEllipsoidUsingPlanes (src/pca/EllipsoidExtractorLocalOptimization.cpp:16-85) for many
ellipsoids in one launch."""
import numpy as np

from . import _lib


def optimize_ellipsoids_using_planes(ellipsoids, planes, n_iter=10, normal_direction=False, device=0, trace=False):
    """ellipsoids (n,10) float64: translation, quaternion x y z w, half-axes (g2o::ellipsoid::toVector); planes: list of
    (P_i,4) arrays A B C D (mPlanesParam rows).  Returns (ellipsoids_out (n,10), chi2 (n,), iterations (n,)[, trace])."""
    E = np.ascontiguousarray(ellipsoids, dtype=np.float64).reshape(-1, 10)
    n = E.shape[0]
    if len(planes) != n:
        raise ValueError("one plane array per ellipsoid")
    arrs = [np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 4) for p in planes]
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum([a.shape[0] for a in arrs])
    flat = np.ascontiguousarray(np.concatenate(arrs, axis=0)) if off[-1] else np.zeros((1, 4))
    out = np.empty_like(E)
    chi2 = np.empty(n)
    iters = np.empty(n, np.int32)
    tr = np.zeros((n, max(int(n_iter), 1), 3)) if trace else None
    _lib.check(_lib.lib().qsp_ellipsoid_fit_planes(int(device), n, _lib.dptr(E), _lib.i32ptr(off), _lib.dptr(flat), int(n_iter),
                                                   1 if normal_direction else 0, _lib.dptr(out), _lib.dptr(chi2),
                                                   _lib.i32ptr(iters), _lib.dptr(tr) if trace else _lib.c_double_p()))
    return (out, chi2, iters, tr) if trace else (out, chi2, iters)


def infer_ellipsoids_with_prior(ellipsoids, planes_normal, planes, pri, weight, angle_sigma_deg=10.0, ground_plane_weight=None,
                                n_iter=10, device=0, trace=False):
    """priorInfer::infer's optimisation (src/core/PriorInfer.cpp:331-427) for many ellipsoids in one launch (qsp_ellipsoid_fit_prior).
    ellipsoids (n,10) as above; planes_normal / planes: lists of (P_i,4) arrays (the edges with / without the normal constraint);
    pri (n,2) the prior (d, e) = 1 : d : e; weight (n,) or scalar; ground_plane_weight (n,) / scalar / None.
    Returns (ellipsoids_out (n,10), chi2 (n,), iterations (n,)[, trace])."""
    E = np.ascontiguousarray(ellipsoids, dtype=np.float64).reshape(-1, 10)
    n = E.shape[0]

    def flat(lst):
        arrs = [np.ascontiguousarray(p, dtype=np.float64).reshape(-1, 4) for p in lst]
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum([a.shape[0] for a in arrs])
        return off, (np.ascontiguousarray(np.concatenate(arrs, axis=0)) if off[-1] else np.zeros((1, 4)))
    if len(planes_normal) != n or len(planes) != n:
        raise ValueError("one plane array of each kind per ellipsoid")
    on, fn = flat(planes_normal)
    op, fp = flat(planes)
    P = np.ascontiguousarray(np.broadcast_to(np.asarray(pri, np.float64), (n, 2)))
    W = np.ascontiguousarray(np.broadcast_to(np.asarray(weight, np.float64), (n,)))
    G = None if ground_plane_weight is None else np.ascontiguousarray(np.broadcast_to(np.asarray(ground_plane_weight, np.float64), (n,)))
    out = np.empty_like(E)
    chi2 = np.empty(n)
    iters = np.empty(n, np.int32)
    tr = np.zeros((n, max(int(n_iter), 1), 3)) if trace else None
    _lib.check(_lib.lib().qsp_ellipsoid_fit_prior(int(device), n, _lib.dptr(E), _lib.i32ptr(on), _lib.dptr(fn), _lib.i32ptr(op),
                                                  _lib.dptr(fp), _lib.dptr(P), _lib.dptr(W),
                                                  _lib.dptr(G) if G is not None else _lib.c_double_p(), float(angle_sigma_deg),
                                                  int(n_iter), _lib.dptr(out), _lib.dptr(chi2), _lib.i32ptr(iters),
                                                  _lib.dptr(tr) if trace else _lib.c_double_p()))
    return (out, chi2, iters, tr) if trace else (out, chi2, iters)
