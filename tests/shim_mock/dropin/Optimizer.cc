// TEST STAND-IN for the reference's src/Optimizer.cc: defines the same `Optimizer::` members that file defines, each one
// appending a line to the file named by $QSP_G2O_LOG instead of running g2o.  Compiled into Optimizer_hip.cc under the name
// OptimizerG2O, exactly as the real file would be.
#include <cstdio>
#include <cstdlib>

#include "Optimizer.h"

namespace ORB_SLAM2 {

static void g2o_log(const char* what, long a = 0, long b = 0) {
    const char* p = getenv("QSP_G2O_LOG");
    if (!p) return;
    FILE* f = fopen(p, "a");
    fprintf(f, "g2o:%s %ld %ld\n", what, a, b);
    fclose(f);
}

Optimizer::Optimizer() { mbGroundPlaneSet = false; }

void Optimizer::GlobalBundleAdjustemnt(Map* pMap, int nIterations, bool* pbStopFlag, const unsigned long nLoopKF, const bool bRobust) {
    g2o_log("GlobalBundleAdjustemnt", nIterations, (long)nLoopKF);
    BundleAdjustment(pMap->GetAllKeyFrames(), pMap->GetAllMapPoints(), nIterations, pbStopFlag, nLoopKF, bRobust);
}

void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations,
                                 bool*, const unsigned long nLoopKF, const bool bRobust) {
    g2o_log("BundleAdjustment", (long)vpKFs.size(), (long)vpMP.size());
    (void)nIterations; (void)nLoopKF; (void)bRobust;
}

int Optimizer::PoseOptimization(Frame* pFrame) {
    g2o_log("PoseOptimization", pFrame->N);
    return -7;
}

void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool*, Map*) {
    // the walk of src/Optimizer.cc:461-470 starts from marks that differ from pKF->mnId: report whether it would find any
    long clean = pKF->mnBALocalForKF != pKF->mnId;
    for (KeyFrame* k : pKF->GetVectorCovisibleKeyFrames()) clean &= (k->mnBALocalForKF != pKF->mnId);
    g2o_log("LocalBundleAdjustment", (long)pKF->mnId, clean);
}

void Optimizer::OptimizeEssentialGraph(Map*, KeyFrame* pLoopKF, KeyFrame* pCurKF, const KeyFrameAndPose& NonCorrectedSim3,
                                       const KeyFrameAndPose& CorrectedSim3, const map<KeyFrame*, set<KeyFrame*>>& LoopConnections,
                                       const bool& bFixScale) {
    g2o_log("OptimizeEssentialGraph", (long)(NonCorrectedSim3.size() + CorrectedSim3.size() + LoopConnections.size()),
            (long)(pLoopKF->mnId * 100 + pCurKF->mnId * 10 + (bFixScale ? 1 : 0)));
}

int Optimizer::OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12, const float th2,
                            const bool bFixScale) {
    g2o_log("OptimizeSim3", (long)vpMatches1.size(), (long)th2);
    g2oS12.s = 42.0;
    (void)pKF1; (void)pKF2; (void)bFixScale;
    return 17;
}

}  // namespace ORB_SLAM2
