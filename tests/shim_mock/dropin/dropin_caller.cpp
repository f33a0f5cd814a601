// dropin_caller.cpp -- a caller written against the reference's Optimizer API only (include/Optimizer.h:75-107), using it
// the way the reference's call sites do (src/LocalMapping.cc:235-241, src/LoopClosing.cc:336,577,660,
// src/LoopClosing_util.cc:299, src/Tracking.cc:810,899).  It includes "Optimizer.h" and nothing of this repository: that it
// compiles and links against qsp_slam_amd/orbslam/Optimizer_hip.cc is the drop-in property under test.
//   usage: dropin_caller <scene.bin> <out.txt>
#include <cstdio>

#include "Optimizer.h"
#include "mock_scene.h"

using namespace ORB_SLAM2;

namespace ORB_SLAM2 {
struct OptimizerPeek {
    static bool ground_set(const Optimizer& o) { return o.mbGroundPlaneSet; }
    static double ground(const Optimizer& o, int i) { return o.mGroundPlaneNormal.v[i]; }
};
}  // namespace ORB_SLAM2

extern "C" long qsp_optimizer_failure_count(void);
extern "C" long qsp_optimizer_fallback_count(void);      // exported by Optimizer_hip.cc next to class Optimizer

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    mock::Scene S;
    if (!S.load(argv[1])) return 1;
    FILE* out = fopen(argv[2], "w");
    bool mbAbortBA = false, mbStopGBA = false;
    KeyFrame* mpCurrentKeyFrame = &S.kfs[0];
    Map* mpMap = &S.map;

    fprintf(out, "nBAdone0 %d\n", Optimizer::nBAdone);
    Optimizer::LocalJointBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpMap);        // src/LocalMapping.cc:235
    fprintf(out, "nBAdone1 %d\n", Optimizer::nBAdone);
    fprintf(out, "kf1_tx %.6f\n", S.kfs[1].Tcw.at<float>(0, 3));
    Optimizer::LocalBundleAdjustment(mpCurrentKeyFrame, &mbAbortBA, mpMap);             // src/LocalMapping.cc:239
    fprintf(out, "nBAdone2 %d\n", Optimizer::nBAdone);
    fprintf(out, "kf1_tx2 %.6f\n", S.kfs[1].Tcw.at<float>(0, 3));
    const unsigned long nLoopKF = 7;
    Optimizer::GlobalJointBundleAdjustemnt(mpMap, 10, &mbStopGBA, nLoopKF, false);      // src/LoopClosing_util.cc:299
    Optimizer::GlobalBundleAdjustemnt(mpMap, 10, &mbStopGBA, nLoopKF, false);           // src/LoopClosing.cc:660
    Optimizer::GlobalBundleAdjustemnt(mpMap, 20);                                       // src/Tracking.cc:810
    fprintf(out, "gba_marks %lu\n", S.kfs[1].mnBAGlobalForKF);

    Frame mCurrentFrame;                                                                // src/Tracking.cc:899
    KeyFrame& k = S.kfs[0];
    mCurrentFrame.fx = k.fx; mCurrentFrame.fy = k.fy; mCurrentFrame.cx = k.cx; mCurrentFrame.cy = k.cy; mCurrentFrame.mbf = k.mbf;
    mCurrentFrame.mTcw = k.Tcw.clone();
    mCurrentFrame.mvInvLevelSigma2 = S.sig;
    for (size_t i = 0; i < k.mvKeysUn.size(); ++i) {
        mCurrentFrame.mvKeysUn.push_back(k.mvKeysUn[i]);
        mCurrentFrame.mvuRight.push_back(k.mvuRight[i]);
        mCurrentFrame.mvpMapPoints.push_back(k.mps[i]);
        mCurrentFrame.mvbOutlier.push_back(false);
    }
    mCurrentFrame.N = (int)mCurrentFrame.mvKeysUn.size();
    int nGood = Optimizer::PoseOptimization(&mCurrentFrame);
    fprintf(out, "pose_inliers %d of %d\n", nGood, mCurrentFrame.N);

    std::vector<MapPoint*> vpMapPointMatches(5, &S.pts[0]);                             // src/LoopClosing.cc:336
    g2o::Sim3 gScm{};
    const int nInliers = Optimizer::OptimizeSim3(mpCurrentKeyFrame, &S.kfs[1], vpMapPointMatches, gScm, 10, true);
    fprintf(out, "sim3 %d %.1f\n", nInliers, gScm.s);
    KeyFrameAndPose NonCorrectedSim3, CorrectedSim3;                                    // src/LoopClosing.cc:577
    CorrectedSim3[&S.kfs[1]] = gScm;
    map<KeyFrame*, set<KeyFrame*>> LoopConnections;
    LoopConnections[&S.kfs[0]].insert(&S.kfs[1]);
    const bool mbFixScale = true;
    Optimizer::OptimizeEssentialGraph(mpMap, &S.kfs[1], mpCurrentKeyFrame, NonCorrectedSim3, CorrectedSim3, LoopConnections,
                                      mbFixScale);

    Optimizer* mpOptimizer = new Optimizer();                                           // src/System.cc (SetOptimizer users)
    fprintf(out, "ground0 %d\n", OptimizerPeek::ground_set(*mpOptimizer) ? 1 : 0);
    Vector4d n{{0.0, -1.0, 0.0, 1.5}};
    mpOptimizer->SetGroundPlane(n);
    fprintf(out, "ground1 %d %.1f %.1f\n", OptimizerPeek::ground_set(*mpOptimizer) ? 1 : 0, OptimizerPeek::ground(*mpOptimizer, 1),
            OptimizerPeek::ground(*mpOptimizer, 3));
    delete mpOptimizer;
    fprintf(out, "failures %ld\n", qsp_optimizer_failure_count());         // library calls that failed (map left untouched)
    fprintf(out, "fallbacks %ld\n", qsp_optimizer_fallback_count());      // what a deployment asserts to be zero
    fclose(out);
    return 0;
}
