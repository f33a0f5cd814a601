// qsp_optimizer_shim.h -- header-only C++ shim that keeps the ORB-SLAM2 `Optimizer` static API of QSP-SLAM
// (include/Optimizer.h:78-107 of the reference) and routes the bundle adjustments to libqsp_hip.so.
//
// It is compiled INSIDE the QSP-SLAM tree (it needs the reference's KeyFrame.h / MapPoint.h / MapObject.h / Map.h /
// ObjectDetection.h, hence Eigen + OpenCV, which this repository's build image does not have).  It contains no numerics:
// it walks the map exactly as src/Optimizer_util.cc:309-385,396-586 (local) and :44-250 (global) do, flattens the graph into
// a qsp_ba_scene with the reference's vertex-id scheme and edge insertion order, calls the C-ABI, and writes the results
// back as src/Optimizer_util.cc:686-769 / :252-305 do.  tests/shim_mock/ compiles it against stand-in types and checks the
// flattening.
//
// Usage in the reference tree: NOTHING changes at the call sites.  qsp_slam_amd/orbslam/Optimizer_hip.cc (this repository)
// takes the place of src/Optimizer.cc + src/Optimizer_util.cc in the CMake source list and defines the members of
// `class Optimizer` exactly as include/Optimizer.h:75-107 declares them, forwarding the bundle adjustments and
// PoseOptimization to OptimizerHip below, OptimizeSim3 / OptimizeEssentialGraph to the reference's own g2o code, and any call
// the GPU path reports an error for to that g2o code as well.  OptimizerHip's entry points
// return a qsp status (QSP_OK / QSP_ERR_*) instead of void so that the caller can tell; on error the map is left exactly as
// it was found (the BA bookkeeping marks mnBALocalForKF / mnBAFixedForKF are rolled back) and qsp_last_error() is logged.
#ifndef QSP_OPTIMIZER_SHIM_H
#define QSP_OPTIMIZER_SHIM_H

#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "qsp_hip.h"

#ifndef QSP_SHIM_MOCK_TYPES      // the real tree
#include "Converter.h"
#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapObject.h"
#include "MapPoint.h"
#include "ObjectDetection.h"
#endif

namespace ORB_SLAM2 {
namespace qsp_shim {

// Converter::toSE3Quat(cv::Mat float32 4x4) -> g2o::SE3Quat(R, t): Eigen's matrix->quaternion, then normalizeRotation()
// (src/Converter.cc:37-46, Thirdparty/g2o/g2o/types/se3quat.h:61-66,328-333).  Output (tx ty tz qx qy qz qw).
template <typename GetF>
inline void pose7_from_rt(GetF m, double* p) {
    const double R[9] = {m(0, 0), m(0, 1), m(0, 2), m(1, 0), m(1, 1), m(1, 2), m(2, 0), m(2, 1), m(2, 2)};
    double q[4];
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    p[0] = m(0, 3); p[1] = m(1, 3); p[2] = m(2, 3);
    p[3] = q[0] / n; p[4] = q[1] / n; p[5] = q[2] / n; p[6] = q[3] / n;
}

// SE3Quat -> 4x4 float (Converter::toCvMat(SE3Quat), src/Converter.cc:65-87): rotation matrix of the quaternion, float32
inline void pose7_to_mat(const double* p, float* T /*row-major 4x4*/) {
    const double x = p[3], y = p[4], z = p[5], w = p[6];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = (float)R[3 * i + j];
        T[4 * i + 3] = (float)p[i];
    }
    T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
}

// the reference's graph walk stamps key-frames / points / objects with the current key-frame id (src/Optimizer_util.cc:314-380);
// when the GPU path fails those stamps are rolled back: the map is left as found (and an opted-in g2o fallback walks the same sets)
struct Marks {
    std::vector<std::pair<long unsigned int*, long unsigned int>> saved;
    void set(long unsigned int& field, long unsigned int v) { saved.emplace_back(&field, field); field = v; }
    void rollback() {
        for (size_t i = saved.size(); i-- > 0;) *saved[i].first = saved[i].second;
        saved.clear();
    }
};

// What a failed library call does.  DEFAULT: fail loudly -- the error is logged, the BA marks are rolled
// back, the map is left exactly as found and the entry point returns (the reference's contract, include/Optimizer.h:78-107, is
// `void`, no throw: the embedding application sees `failure_count()` move and decides; PoseOptimization returns 0 inliers, which
// Tracking treats as a lost frame).  Nothing runs on the CPU unless the deployment asked for it:
//   QSP_SHIM_ALLOW_G2O_FALLBACK=1   hand the failed call to the reference's own g2o code compiled into the drop-in (counted in
//                                   `fallback_count()`, one stderr line per call);
//   QSP_SHIM_NO_FALLBACK=1          strictest: std::abort() after the message.
// OptimizeSim3 / OptimizeEssentialGraph are CPU pass-throughs by design and are not affected.
inline std::atomic<long>& failure_counter() {
    static std::atomic<long> n{0};
    return n;
}
inline std::atomic<long>& fallback_counter() {
    static std::atomic<long> n{0};
    return n;
}
inline long failure_count() { return failure_counter().load(); }     // library calls that returned an error
inline long fallback_count() { return fallback_counter().load(); }   // of those, entry points handed to g2o (opt-in only)
inline void reset_fallback_count() { fallback_counter().store(0); failure_counter().store(0); }

inline bool allow_g2o_fallback() {
    const char* e = std::getenv("QSP_SHIM_ALLOW_G2O_FALLBACK");
    return e && *e == '1';
}

inline int report(const char* where, int rc) {
    if (rc != QSP_OK) {
        failure_counter().fetch_add(1);
        const char* strict = std::getenv("QSP_SHIM_NO_FALLBACK");
        const bool fatal = strict && *strict == '1';
        if (!fatal && allow_g2o_fallback()) {
            fallback_counter().fetch_add(1);
            std::fprintf(stderr, "[qsp_hip] %s failed (%d): %s -- QSP_SHIM_ALLOW_G2O_FALLBACK=1: this call falls back to the "
                                 "reference's g2o path (fallback #%ld)\n", where, rc, qsp_last_error(), fallback_count());
        } else {
            std::fprintf(stderr, "[qsp_hip] %s failed (%d): %s -- the map is left untouched and the call returns (failure #%ld; "
                                 "QSP_SHIM_ALLOW_G2O_FALLBACK=1 would run the reference's g2o code instead)\n",
                         where, rc, qsp_last_error(), failure_count());
        }
        if (fatal) std::abort();
    }
    return rc;
}

struct Flat {   // the flattened graph + back-references for the write-back
    std::vector<KeyFrame*> kfs;
    std::vector<MapPoint*> pts;
    std::vector<MapObject*> objs;
    std::vector<double> kf_pose, kf_K, pt_xyz, obj_pose, mono_obs, mono_info, st_obs, st_info, oe_meas;
    std::vector<uint8_t> kf_fixed;
    std::vector<int64_t> kf_id, pt_id, obj_id;
    std::vector<int32_t> mono_pt, mono_kf, st_pt, st_kf, oe_kf, oe_obj;
    std::vector<KeyFrame*> mono_kfp, st_kfp, oe_kfp;
    std::vector<MapPoint*> mono_mp, st_mp;
    std::vector<MapObject*> oe_mo;
    unsigned long maxKFid = 0, maxMPid = 0;
    std::map<KeyFrame*, int> kf_index;

    int add_kf(KeyFrame* pKF, bool fixed) {
        const int i = (int)kfs.size();
        kfs.push_back(pKF);
        kf_index[pKF] = i;
        double p[7];
        const cv::Mat Tcw = pKF->GetPose();
        pose7_from_rt([&](int r, int c) { return (double)Tcw.at<float>(r, c); }, p);
        kf_pose.insert(kf_pose.end(), p, p + 7);
        kf_fixed.push_back(fixed ? 1 : 0);
        kf_id.push_back((int64_t)pKF->mnId);
        const double K[5] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy, pKF->mbf};
        kf_K.insert(kf_K.end(), K, K + 5);
        if (pKF->mnId > maxKFid) maxKFid = pKF->mnId;
        return i;
    }

    // one map point: vertex + one edge per observing key-frame that is in the graph (src/Optimizer_util.cc:454-541).
    // global_rules (Optimizer::BundleAdjustment / JointBundleAdjustment, src/Optimizer.cc:93-191, Optimizer_util.cc:87-173):
    // maxMPid counts every point offered, and a point without any edge is removed again (vbNotIncludedMP) -> false.
    bool add_point(MapPoint* pMP, bool global_rules = false) {
        const int ip = (int)pts.size();
        const size_t nm0 = mono_pt.size(), ns0 = st_pt.size();
        if (global_rules && pMP->mnId > maxMPid) maxMPid = pMP->mnId;
        pts.push_back(pMP);
        const cv::Mat X = pMP->GetWorldPos();
        for (int i = 0; i < 3; ++i) pt_xyz.push_back((double)X.at<float>(i));
        pt_id.push_back((int64_t)pMP->mnId);   // + maxKFid + 1, added in finish()
        const std::map<KeyFrame*, size_t> observations = pMP->GetObservations();
        for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
            KeyFrame* pKFi = mit->first;
            if (pKFi->isBad()) continue;
            std::map<KeyFrame*, int>::const_iterator f = kf_index.find(pKFi);
            if (f == kf_index.end()) continue;      // optimizer.vertex(pKFi->mnId) would be NULL in the reference
            const cv::KeyPoint& kpUn = pKFi->mvKeysUn[mit->second];
            const float invSigma2 = pKFi->mvInvLevelSigma2[kpUn.octave];
            if (pKFi->mvuRight[mit->second] < 0) {
                mono_pt.push_back(ip); mono_kf.push_back(f->second);
                mono_obs.push_back(kpUn.pt.x); mono_obs.push_back(kpUn.pt.y);
                mono_info.push_back(invSigma2);
                mono_kfp.push_back(pKFi); mono_mp.push_back(pMP);
            } else {
                st_pt.push_back(ip); st_kf.push_back(f->second);
                st_obs.push_back(kpUn.pt.x); st_obs.push_back(kpUn.pt.y); st_obs.push_back(pKFi->mvuRight[mit->second]);
                st_info.push_back(invSigma2);
                st_kfp.push_back(pKFi); st_mp.push_back(pMP);
            }
            if (pMP->mnId > maxMPid) maxMPid = pMP->mnId;
        }
        if (global_rules && mono_pt.size() == nm0 && st_pt.size() == ns0) {
            pts.pop_back();
            pt_xyz.resize(pt_xyz.size() - 3);
            pt_id.pop_back();
            return false;
        }
        return true;
    }

    // one static object: SE3 vertex (estimate SE3Tow) + one EdgeSE3LieAlgebra per observing key-frame in the graph
    // (src/Optimizer_util.cc:544-586)
    bool add_object(MapObject* pMO, bool global_rules = false) {
        const int io = (int)objs.size();
        const size_t ne0 = oe_kf.size();
        objs.push_back(pMO);
        double p[7];
        pose7_from_rt([&](int r, int c) { return (double)pMO->SE3Tow(r, c); }, p);
        obj_pose.insert(obj_pose.end(), p, p + 7);
        obj_id.push_back((int64_t)pMO->mnId);    // + maxKFid + maxMPid + 2, added in finish()
        const std::map<KeyFrame*, size_t> observations = pMO->GetObservations();
        for (std::map<KeyFrame*, size_t>::const_iterator it = observations.begin(); it != observations.end(); ++it) {
            KeyFrame* pKFi = it->first;
            std::map<KeyFrame*, int>::const_iterator f = kf_index.find(pKFi);
            if (f == kf_index.end() || pKFi->isBad()) continue;
            auto dets = pKFi->GetObjectDetections();
            auto det = dets[it->second];
            pose7_from_rt([&](int r, int c) { return (double)det->SE3Tco(r, c); }, p);
            oe_meas.insert(oe_meas.end(), p, p + 7);
            oe_kf.push_back(f->second); oe_obj.push_back(io);
            oe_kfp.push_back(pKFi); oe_mo.push_back(pMO);
        }
        if (global_rules && oe_kf.size() == ne0) {      // vbNotIncludedMO, src/Optimizer_util.cc:229-233
            objs.pop_back();
            obj_pose.resize(obj_pose.size() - 7);
            obj_id.pop_back();
            return false;
        }
        return true;
    }

    void finish(qsp_ba_scene* s) {
        for (size_t i = 0; i < pt_id.size(); ++i) pt_id[i] += (int64_t)maxKFid + 1;
        for (size_t i = 0; i < obj_id.size(); ++i) obj_id[i] += (int64_t)maxKFid + (int64_t)maxMPid + 2;
        s->n_kf = (int32_t)kfs.size(); s->n_pt = (int32_t)pts.size(); s->n_obj = (int32_t)objs.size();
        s->n_mono = (int32_t)mono_pt.size(); s->n_stereo = (int32_t)st_pt.size(); s->n_objedge = (int32_t)oe_kf.size();
        s->kf_pose = kf_pose.data(); s->kf_fixed = kf_fixed.data(); s->kf_id = kf_id.data(); s->kf_K = kf_K.data();
        s->pt_xyz = pt_xyz.data(); s->pt_id = pt_id.data(); s->obj_pose = obj_pose.data(); s->obj_id = obj_id.data();
        s->mono_pt = mono_pt.data(); s->mono_kf = mono_kf.data(); s->mono_obs = mono_obs.data(); s->mono_info = mono_info.data();
        s->stereo_pt = st_pt.data(); s->stereo_kf = st_kf.data(); s->stereo_obs = st_obs.data(); s->stereo_info = st_info.data();
        s->objedge_kf = oe_kf.data(); s->objedge_obj = oe_obj.data(); s->objedge_meas = oe_meas.data();
        s->objedge_info = 1e3;   // const float invSigmaObject = 1e3, src/Optimizer_util.cc:447
    }
};

}  // namespace qsp_shim

class OptimizerHip {
public:
    static int& nBAdone() { static int n = 0; return n; }
    static int& device() { static int d = 0; return d; }

    // Optimizer::LocalJointBundleAdjustment, src/Optimizer_util.cc:309-771.  with_objects = false gives
    // Optimizer::LocalBundleAdjustment (src/Optimizer.cc:458-783), whose only behavioural difference besides the missing
    // object vertices is that an abort after stage 1 still writes back (src/Optimizer.cc:667-673).
    static int LocalJointBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, bool with_objects = true) {
        using namespace qsp_shim;
        Marks marks;
        // ---- local key-frames, points, objects, fixed key-frames: :311-380 ------------------------------------------
        std::list<KeyFrame*> lLocalKeyFrames;
        lLocalKeyFrames.push_back(pKF);
        marks.set(pKF->mnBALocalForKF, pKF->mnId);
        const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
        for (size_t i = 0; i < vNeighKFs.size(); ++i) {
            KeyFrame* pKFi = vNeighKFs[i];
            marks.set(pKFi->mnBALocalForKF, pKF->mnId);
            if (!pKFi->isBad()) lLocalKeyFrames.push_back(pKFi);
        }
        std::list<MapPoint*> lLocalMapPoints;
        std::list<MapObject*> lLocalMapObjects;
        for (KeyFrame* k : lLocalKeyFrames) {
            for (MapPoint* pMP : k->GetMapPointMatches())
                if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) {
                    lLocalMapPoints.push_back(pMP);
                    marks.set(pMP->mnBALocalForKF, pKF->mnId);
                }
            if (with_objects)
                for (MapObject* pMO : k->GetMapObjectMatches())
                    if (pMO && pMO->mnBALocalForKF != pKF->mnId) {
                        lLocalMapObjects.push_back(pMO);
                        marks.set(pMO->mnBALocalForKF, pKF->mnId);
                    }
        }
        std::list<KeyFrame*> lFixedCameras;
        for (MapPoint* pMP : lLocalMapPoints) {
            const std::map<KeyFrame*, size_t> observations = pMP->GetObservations();
            for (std::map<KeyFrame*, size_t>::const_iterator mit = observations.begin(); mit != observations.end(); ++mit) {
                KeyFrame* pKFi = mit->first;
                if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                    marks.set(pKFi->mnBAFixedForKF, pKF->mnId);
                    if (!pKFi->isBad()) lFixedCameras.push_back(pKFi);
                }
            }
        }
        // ---- flatten: vertices and edges in the reference's insertion order (:396-586) ------------------------------
        Flat F;
        for (KeyFrame* k : lLocalKeyFrames) F.add_kf(k, k->mnId == 0);
        for (KeyFrame* k : lFixedCameras) F.add_kf(k, true);
        for (MapPoint* pMP : lLocalMapPoints) F.add_point(pMP);
        for (MapObject* pMO : lLocalMapObjects)
            if (!pMO->isDynamic()) F.add_object(pMO);
        qsp_ba_scene scene;
        F.finish(&scene);
        if (pbStopFlag && *pbStopFlag) return QSP_OK;                                 // :589-596
        qsp_ba_problem* prob = nullptr;
        int rc = report("qsp_ba_create", qsp_ba_create(&scene, device(), &prob));     // failure = the map is left as found
        if (rc != QSP_OK) { marks.rollback(); return rc; }
        const volatile uint8_t* stop = reinterpret_cast<const volatile uint8_t*>(pbStopFlag);
        rc = report("qsp_ba_local_joint", qsp_ba_local_joint(prob, stop, nullptr, nullptr));
        if (rc != QSP_OK) { qsp_ba_destroy(prob); marks.rollback(); return rc; }
        if (with_objects && pbStopFlag && *pbStopFlag) { qsp_ba_destroy(prob); return QSP_OK; }   // :603-610: no write-back
        // ---- outlier observations (:665-711), under the map mutex (:714-736) -------------------------------------------
        std::vector<double> cm(F.mono_pt.size() + 1), cs(F.st_pt.size() + 1), co(F.oe_kf.size() + 1);
        std::vector<uint8_t> pm(F.mono_pt.size() + 1), ps(F.st_pt.size() + 1);
        std::vector<double> kf(F.kf_pose.size()), pt(F.pt_xyz.size() + 1), ob(F.obj_pose.size() + 1);
        rc = report("qsp_ba_get_edges", qsp_ba_get_edges(prob, cm.data(), cs.data(), co.data(), pm.data(), ps.data()));
        if (rc == QSP_OK) rc = report("qsp_ba_get_state", qsp_ba_get_state(prob, kf.data(), pt.data(), ob.data()));
        qsp_ba_destroy(prob);
        if (rc != QSP_OK) { marks.rollback(); return rc; }
        std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);
        for (size_t i = 0; i < F.mono_pt.size(); ++i)
            if (!F.mono_mp[i]->isBad() && (cm[i] > 5.991 || !pm[i])) {
                F.mono_kfp[i]->EraseMapPointMatch(F.mono_mp[i]);
                F.mono_mp[i]->EraseObservation(F.mono_kfp[i]);
            }
        for (size_t i = 0; i < F.st_pt.size(); ++i)
            if (!F.st_mp[i]->isBad() && (cs[i] > 7.815 || !ps[i])) {
                F.st_kfp[i]->EraseMapPointMatch(F.st_mp[i]);
                F.st_mp[i]->EraseObservation(F.st_kfp[i]);
            }
        for (size_t i = 0; i < F.oe_kf.size(); ++i)
            if (co[i] > 1e3) {
                F.oe_kfp[i]->EraseMapObjectMatch(F.oe_mo[i]);
                F.oe_mo[i]->EraseObservation(F.oe_kfp[i]);
            }
        // ---- recover optimised data (:741-768): float32 round trip at the boundary -----------------------------------
        size_t n_local = lLocalKeyFrames.size();
        for (size_t i = 0; i < n_local; ++i) {
            float T[16];
            pose7_to_mat(&kf[7 * i], T);
            cv::Mat M(4, 4, CV_32F);
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) M.at<float>(r, c) = T[4 * r + c];
            F.kfs[i]->SetPose(M);
        }
        for (size_t i = 0; i < F.pts.size(); ++i) {
            cv::Mat X(3, 1, CV_32F);
            for (int r = 0; r < 3; ++r) X.at<float>(r) = (float)pt[3 * i + r];
            F.pts[i]->SetWorldPos(X);
            F.pts[i]->UpdateNormalAndDepth();
        }
        for (size_t i = 0; i < F.objs.size(); ++i) {
            MapObject* pMO = F.objs[i];
            if (pMO->isDynamic() || pMO->isBad()) continue;
            float T[16];
            pose7_to_mat(&ob[7 * i], T);
            Eigen::Matrix4f Tow;
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) Tow(r, c) = T[4 * r + c];
            pMO->SetObjectPoseSE3(Tow.inverse());            // Converter::toMatrix4f(SE3Tow).inverse(), :764-765
        }
        if (with_objects) nBAdone()++;       // Optimizer::nBAdone++ exists in the joint variant only (src/Optimizer_util.cc:769)
        return QSP_OK;
    }

    static int LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap) {
        return LocalJointBundleAdjustment(pKF, pbStopFlag, pMap, false);
    }

    // Optimizer::JointBundleAdjustment, src/Optimizer_util.cc:44-307 (and, with no objects, Optimizer::BundleAdjustment,
    // src/Optimizer.cc:54-242): the given key-frames (mnId 0 fixed), points and static objects; one optimize(nIterations);
    // Huber sqrt(5.99) / sqrt(7.815) / sqrt(0.1*1e3) only if bRobust; vertices left without an edge are dropped and not
    // written back; results go to the *GBA members when nLoopKF != 0.
    static int JointBundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP,
                                     const std::vector<MapObject*>& vpMO, int nIterations = 5, bool* pbStopFlag = nullptr,
                                     const unsigned long nLoopKF = 0, const bool bRobust = true) {
        using namespace qsp_shim;
        Flat F;
        for (KeyFrame* k : vpKFs)
            if (!k->isBad()) F.add_kf(k, k->mnId == 0);
        for (MapPoint* pMP : vpMP)
            if (!pMP->isBad()) F.add_point(pMP, true);
        for (MapObject* pMO : vpMO)
            if (pMO && !pMO->isDynamic() && !pMO->isBad()) F.add_object(pMO, true);
        qsp_ba_scene scene;
        F.finish(&scene);
        qsp_ba_problem* prob = nullptr;
        int rc = report("qsp_ba_create", qsp_ba_create(&scene, device(), &prob));
        if (rc != QSP_OK) return rc;                // nothing of the map has been touched yet (no *GBA member written)
        const volatile uint8_t* stop = reinterpret_cast<const volatile uint8_t*>(pbStopFlag);
        const double dm = bRobust ? (double)(float)std::sqrt(5.99) : 0.0;            // thHuber2D, :80
        const double ds = bRobust ? (double)(float)std::sqrt(7.815) : 0.0;           // thHuber3D, :81
        const double dobj = bRobust ? (double)(float)std::sqrt(0.1f * 1e3f) : 0.0;   // :82-83
        std::vector<double> kf(F.kf_pose.size()), pt(F.pt_xyz.size() + 1), ob(F.obj_pose.size() + 1);
        rc = report("qsp_ba_set_levels", qsp_ba_set_levels(prob, nullptr, nullptr, nullptr));
        if (rc == QSP_OK) rc = report("qsp_ba_optimize", qsp_ba_optimize(prob, nIterations, dm, ds, dobj, stop, nullptr));
        if (rc == QSP_OK) rc = report("qsp_ba_get_state", qsp_ba_get_state(prob, kf.data(), pt.data(), ob.data()));
        qsp_ba_destroy(prob);
        if (rc != QSP_OK) return rc;
        for (size_t i = 0; i < F.kfs.size(); ++i) {                                   // :252-270
            float T[16];
            pose7_to_mat(&kf[7 * i], T);
            cv::Mat M(4, 4, CV_32F);
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) M.at<float>(r, c) = T[4 * r + c];
            if (nLoopKF == 0) F.kfs[i]->SetPose(M);
            else { F.kfs[i]->mTcwGBA = M.clone(); F.kfs[i]->mnBAGlobalForKF = nLoopKF; }
        }
        for (size_t i = 0; i < F.pts.size(); ++i) {                                   // :272-290
            cv::Mat X(3, 1, CV_32F);
            for (int r = 0; r < 3; ++r) X.at<float>(r) = (float)pt[3 * i + r];
            if (nLoopKF == 0) { F.pts[i]->SetWorldPos(X); F.pts[i]->UpdateNormalAndDepth(); }
            else { F.pts[i]->mPosGBA = X.clone(); F.pts[i]->mnBAGlobalForKF = nLoopKF; }
        }
        for (size_t i = 0; i < F.objs.size(); ++i) {                                  // :292-305
            float T[16];
            pose7_to_mat(&ob[7 * i], T);
            Eigen::Matrix4f Tow;
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) Tow(r, c) = T[4 * r + c];
            if (nLoopKF == 0) F.objs[i]->SetObjectPoseSE3(Tow.inverse());
            else { F.objs[i]->mTwoGBA = Tow.inverse(); F.objs[i]->mnBAGlobalForKF = nLoopKF; }
        }
        return QSP_OK;
    }

    // Optimizer::PoseOptimization, src/Optimizer.cc:244-456: the frame's pose against its matched map points (fixed),
    // 4 x optimize(10) with inlier / outlier re-classification; sets pFrame->mvbOutlier and the pose, returns the number
    // of inliers.  One kernel launch (qsp_pose_optimize); the optimiser object is created once per thread.
    // *status (optional) receives the qsp status; on error nothing of the frame has been changed except that the outlier
    // flags of the matched slots were cleared, which the reference does first as well (:291).
    static int PoseOptimization(Frame* pFrame, int* status = nullptr) {
        using namespace qsp_shim;
        if (status) *status = QSP_OK;
        const int N = pFrame->N;
        std::vector<double> X, obs, info;
        std::vector<uint8_t> stereo;
        std::vector<int> index;
        {
            std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);
            for (int i = 0; i < N; ++i) {
                MapPoint* pMP = pFrame->mvpMapPoints[i];
                if (!pMP) continue;
                pFrame->mvbOutlier[i] = false;
                const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
                const bool st = !(pFrame->mvuRight[i] < 0);
                obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(st ? pFrame->mvuRight[i] : -1.0);
                info.push_back(pFrame->mvInvLevelSigma2[kpUn.octave]);
                stereo.push_back(st ? 1 : 0);
                const cv::Mat Xw = pMP->GetWorldPos();
                for (int r = 0; r < 3; ++r) X.push_back((double)Xw.at<float>(r));
                index.push_back(i);
            }
        }
        const int n = (int)index.size();
        if (n < 3) return 0;                                                     // :368-369
        static thread_local qsp_pose_optimizer* ctx = nullptr;
        static thread_local int ctx_cap = 0;
        if (!ctx || n > ctx_cap) {
            if (ctx) qsp_pose_optimizer_destroy(ctx);
            ctx = nullptr;
            ctx_cap = n > 4096 ? 2 * n : 4096;
            const int rc = report("qsp_pose_optimizer_create", qsp_pose_optimizer_create(device(), ctx_cap, &ctx));
            if (rc != QSP_OK) { ctx = nullptr; ctx_cap = 0; if (status) *status = rc; return 0; }
        }
        double pose[7], pose_out[7];
        const cv::Mat Tcw = pFrame->mTcw;
        pose7_from_rt([&](int r, int c) { return (double)Tcw.at<float>(r, c); }, pose);
        const double K[5] = {pFrame->fx, pFrame->fy, pFrame->cx, pFrame->cy, pFrame->mbf};
        std::vector<uint8_t> outlier(n, 0);
        int32_t n_inliers = 0;
        const int rc = report("qsp_pose_optimize", qsp_pose_optimize(ctx, n, K, pose, X.data(), obs.data(), info.data(),
                                                                       stereo.data(), pose_out, outlier.data(), &n_inliers, nullptr));
        if (rc != QSP_OK) { if (status) *status = rc; return 0; }
        for (int e = 0; e < n; ++e) pFrame->mvbOutlier[index[e]] = outlier[e] != 0;
        float T[16];
        pose7_to_mat(pose_out, T);
        cv::Mat M(4, 4, CV_32F);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) M.at<float>(r, c) = T[4 * r + c];
        pFrame->SetPose(M);
        return n_inliers;
    }

    // src/Optimizer.cc:54-242
    static int BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations = 5,
                                bool* pbStopFlag = nullptr, const unsigned long nLoopKF = 0, const bool bRobust = true) {
        return JointBundleAdjustment(vpKFs, vpMP, std::vector<MapObject*>(), nIterations, pbStopFlag, nLoopKF, bRobust);
    }

    // src/Optimizer.cc:46-51
    static int GlobalBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = nullptr,
                                      const unsigned long nLoopKF = 0, const bool bRobust = true) {
        return BundleAdjustment(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), nIterations, pbStopFlag, nLoopKF, bRobust);
    }

    // src/Optimizer_util.cc:36-42
    static int GlobalJointBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = nullptr,
                                           const unsigned long nLoopKF = 0, const bool bRobust = true) {
        return JointBundleAdjustment(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), pMap->GetAllMapObjects(), nIterations,
                              pbStopFlag, nLoopKF, bRobust);
    }
};

}  // namespace ORB_SLAM2
#endif  // QSP_OPTIMIZER_SHIM_H
