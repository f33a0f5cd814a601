// TEST STAND-IN for the reference's src/Optimizer_util.cc (the joint bundle adjustments, nBAdone, SetGroundPlane); see
// Optimizer.cc beside it.
#include "Optimizer.h"

namespace ORB_SLAM2 {

int Optimizer::nBAdone = 0;

void Optimizer::GlobalJointBundleAdjustemnt(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    g2o_log("GlobalJointBundleAdjustemnt", nIterations, (long)nLoopKF);
    JointBundleAdjustment(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), pMap->GetAllMapObjects(), nIterations, pbStopFlag,
                          nLoopKF, bRobust);
}

void Optimizer::JointBundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP,
                                      const std::vector<MapObject*>& vpMO, int, bool*, const unsigned long, const bool) {
    g2o_log("JointBundleAdjustment", (long)(vpKFs.size() + vpMP.size()), (long)vpMO.size());
}

void Optimizer::LocalJointBundleAdjustment(KeyFrame* pKF, bool*, Map*) {
    long clean = pKF->mnBALocalForKF != pKF->mnId;
    for (KeyFrame* k : pKF->GetVectorCovisibleKeyFrames()) clean &= (k->mnBALocalForKF != pKF->mnId && k->mnBAFixedForKF != pKF->mnId);
    for (MapPoint* p : pKF->GetMapPointMatches()) if (p) clean &= (p->mnBALocalForKF != pKF->mnId);
    for (MapObject* o : pKF->GetMapObjectMatches()) if (o) clean &= (o->mnBALocalForKF != pKF->mnId);
    g2o_log("LocalJointBundleAdjustment", (long)pKF->mnId, clean);
    Optimizer::nBAdone++;
}

void Optimizer::SetGroundPlane(Vector4d& normal) {
    mbGroundPlaneSet = true;
    mGroundPlaneNormal = normal;
}

}  // namespace ORB_SLAM2
