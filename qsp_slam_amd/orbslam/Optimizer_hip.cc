// Optimizer_hip.cc -- drop-in translation unit for the QSP-SLAM tree: it takes the place of src/Optimizer.cc and
// src/Optimizer_util.cc in the reference's CMake source list (CMakeLists.txt:107-108) and defines every member of
// `class ORB_SLAM2::Optimizer` that include/Optimizer.h:75-107 declares.  The header and every call site
// (src/LocalMapping.cc:235,239, src/LoopClosing.cc:336,577,660, src/LoopClosing_util.cc:200,299,
// src/Tracking.cc:810,899,1022,1064,1688-1719) stay as they are.
//
//   member                                                     runs on
//   BundleAdjustment / GlobalBundleAdjustemnt                  MI355X  (OptimizerHip, include/qsp_optimizer_shim.h -> libqsp_hip.so)
//   JointBundleAdjustment / GlobalJointBundleAdjustemnt        MI355X
//   LocalBundleAdjustment / LocalJointBundleAdjustment         MI355X
//   PoseOptimization                                           MI355X
//   OptimizeEssentialGraph / OptimizeSim3                      CPU: the reference's own g2o code (loop closing; out of the hot path)
//   Optimizer(), SetGroundPlane, nBAdone                       as src/Optimizer.cc:41-44, src/Optimizer_util.cc:34,773-776
//
// The reference's two source files are compiled INTO this unit, unchanged, under the class name OptimizerG2O (they stay on
// disk, they only leave the CMake list).  That gives the CPU pass-throughs above and the fallback: the reference's bundle
// adjustments cannot fail, the GPU path can (no device, out of device memory for the dense reduced system) -- when an
// OptimizerHip entry point reports an error it has left the map as it
// found it, the error text is logged, and the same call is handed to the g2o implementation.
//
// Build (in the reference tree):  copy this file to src/, replace the two entries of CMakeLists.txt:107-108 by
// src/Optimizer_hip.cc, add <this repository>/include to include_directories and qsp_hip to target_link_libraries.
// QSP_REF_OPTIMIZER_CC / QSP_REF_OPTIMIZER_UTIL_CC override where the two reference sources are found.
#include "Optimizer.h"               // the reference's header, unchanged: class Optimizer

#include "qsp_optimizer_shim.h"

#ifndef QSP_REF_OPTIMIZER_CC
#define QSP_REF_OPTIMIZER_CC "Optimizer.cc"
#endif
#ifndef QSP_REF_OPTIMIZER_UTIL_CC
#define QSP_REF_OPTIMIZER_UTIL_CC "Optimizer_util.cc"
#endif

namespace ORB_SLAM2 {

// Receives the definitions of the reference's two source files (every `Optimizer::member` there becomes
// `OptimizerG2O::member` through the macro below).  Only what those files define is declared.
class OptimizerG2O {
public:
    OptimizerG2O();
    void static BundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP, int nIterations = 5,
                                 bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true);
    void static JointBundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP,
                                      const std::vector<MapObject*>& vpMO, int nIterations = 5, bool* pbStopFlag = NULL,
                                      const unsigned long nLoopKF = 0, const bool bRobust = true);
    void static GlobalBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0,
                                       const bool bRobust = true);
    void static GlobalJointBundleAdjustemnt(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL,
                                            const unsigned long nLoopKF = 0, const bool bRobust = true);
    void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap);
    void static LocalJointBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap);
    int static PoseOptimization(Frame* pFrame);
    void static OptimizeEssentialGraph(Map* pMap, KeyFrame* pLoopKF, KeyFrame* pCurKF, const KeyFrameAndPose& NonCorrectedSim3,
                                       const KeyFrameAndPose& CorrectedSim3,
                                       const std::map<KeyFrame*, std::set<KeyFrame*>>& LoopConnections, const bool& bFixScale);
    static int OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12,
                            const float th2, const bool bFixScale);
    static int nBAdone;
    void SetGroundPlane(Vector4d& normal);

private:
    bool mbGroundPlaneSet;
    Vector4d mGroundPlaneNormal;
};

}  // namespace ORB_SLAM2

#define Optimizer OptimizerG2O
#include QSP_REF_OPTIMIZER_CC
#include QSP_REF_OPTIMIZER_UTIL_CC
#undef Optimizer

namespace ORB_SLAM2 {

int Optimizer::nBAdone = 0;

Optimizer::Optimizer() { mbGroundPlaneSet = false; }

void Optimizer::SetGroundPlane(Vector4d& normal) {
    mbGroundPlaneSet = true;
    mGroundPlaneNormal = normal;
}

// ---- the hot path: MI355X.  A failed call is logged by the shim and leaves the map untouched; the reference's g2o code runs in
// ---- its place only where the deployment opted in (QSP_SHIM_ALLOW_G2O_FALLBACK=1, include/qsp_optimizer_shim.h) ------------
static inline bool g2o_instead(int rc) { return rc != QSP_OK && qsp_shim::allow_g2o_fallback(); }

void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP, int nIterations,
                                 bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    if (g2o_instead(OptimizerHip::BundleAdjustment(vpKF, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust)))
        OptimizerG2O::BundleAdjustment(vpKF, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust);
}

void Optimizer::JointBundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP,
                                      const std::vector<MapObject*>& vpMO, int nIterations, bool* pbStopFlag,
                                      const unsigned long nLoopKF, const bool bRobust) {
    if (g2o_instead(OptimizerHip::JointBundleAdjustment(vpKF, vpMP, vpMO, nIterations, pbStopFlag, nLoopKF, bRobust)))
        OptimizerG2O::JointBundleAdjustment(vpKF, vpMP, vpMO, nIterations, pbStopFlag, nLoopKF, bRobust);
}

void Optimizer::GlobalBundleAdjustemnt(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF,
                                       const bool bRobust) {
    if (g2o_instead(OptimizerHip::GlobalBundleAdjustemnt(pMap, nIterations, pbStopFlag, nLoopKF, bRobust)))
        OptimizerG2O::GlobalBundleAdjustemnt(pMap, nIterations, pbStopFlag, nLoopKF, bRobust);
}

void Optimizer::GlobalJointBundleAdjustemnt(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF,
                                            const bool bRobust) {
    if (g2o_instead(OptimizerHip::GlobalJointBundleAdjustemnt(pMap, nIterations, pbStopFlag, nLoopKF, bRobust)))
        OptimizerG2O::GlobalJointBundleAdjustemnt(pMap, nIterations, pbStopFlag, nLoopKF, bRobust);
}

void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap) {
    if (g2o_instead(OptimizerHip::LocalBundleAdjustment(pKF, pbStopFlag, pMap)))
        OptimizerG2O::LocalBundleAdjustment(pKF, pbStopFlag, pMap);
}

void Optimizer::LocalJointBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap) {
    const int done = OptimizerHip::nBAdone();
    if (g2o_instead(OptimizerHip::LocalJointBundleAdjustment(pKF, pbStopFlag, pMap)))
        OptimizerG2O::LocalJointBundleAdjustment(pKF, pbStopFlag, pMap);       // counts in OptimizerG2O::nBAdone itself
    nBAdone += (OptimizerHip::nBAdone() - done);
    nBAdone += OptimizerG2O::nBAdone;                                          // src/Optimizer_util.cc:769
    OptimizerG2O::nBAdone = 0;
}

int Optimizer::PoseOptimization(Frame* pFrame) {
    int status = QSP_OK;
    const int nInliers = OptimizerHip::PoseOptimization(pFrame, &status);
    if (status == QSP_OK) return nInliers;
    return qsp_shim::allow_g2o_fallback() ? OptimizerG2O::PoseOptimization(pFrame) : 0;   // 0 inliers: Tracking sees a lost frame
}

// ---- loop closing: CPU pass-through to the reference's g2o code -----------------------------
void Optimizer::OptimizeEssentialGraph(Map* pMap, KeyFrame* pLoopKF, KeyFrame* pCurKF, const KeyFrameAndPose& NonCorrectedSim3,
                                       const KeyFrameAndPose& CorrectedSim3,
                                       const map<KeyFrame*, set<KeyFrame*>>& LoopConnections, const bool& bFixScale) {
    OptimizerG2O::OptimizeEssentialGraph(pMap, pLoopKF, pCurKF, NonCorrectedSim3, CorrectedSim3, LoopConnections, bFixScale);
}

int Optimizer::OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12,
                            const float th2, const bool bFixScale) {
    return OptimizerG2O::OptimizeSim3(pKF1, pKF2, vpMatches1, g2oS12, th2, bFixScale);
}

}  // namespace ORB_SLAM2

// For embedders that only see the reference-shaped Optimizer.h.  `qsp_optimizer_failure_count`: library calls that returned an
// error in this process -- each left the map untouched and was logged (0 in a healthy deployment).  `qsp_optimizer_fallback_count`:
// how many of them were handed to the reference's g2o path, which happens only with QSP_SHIM_ALLOW_G2O_FALLBACK=1
// (QSP_SHIM_NO_FALLBACK=1 makes the first failure fatal).  include/qsp_optimizer_shim.h:report.
extern "C" long qsp_optimizer_failure_count(void) { return ORB_SLAM2::qsp_shim::failure_count(); }
extern "C" long qsp_optimizer_fallback_count(void) { return ORB_SLAM2::qsp_shim::fallback_count(); }
