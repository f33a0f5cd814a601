// TEST STAND-IN for the reference's include/Optimizer.h (lines 21-133): the same class name, the same members with the same
// signatures and default arguments, on top of the stand-in map types of ../mock_orbslam.h.  A caller compiled against this
// header (dropin_caller.cpp) is source-compatible with one compiled against the reference's header.
#ifndef OPTIMIZER_H
#define OPTIMIZER_H
#include <map>
#include <set>
#include <vector>

#include "mock_orbslam.h"

using std::map;      // the reference's headers rely on `using namespace std` from their includes
using std::set;
using std::vector;

struct Vector4d { double v[4]; };
namespace g2o { struct Sim3 { double r[4], t[3], s; }; }

namespace ORB_SLAM2 {

typedef map<KeyFrame*, g2o::Sim3, std::less<KeyFrame*>> KeyFrameAndPose;

class Optimizer {
public:
    Optimizer();
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
                                       const map<KeyFrame*, set<KeyFrame*>>& LoopConnections, const bool& bFixScale);
    static int OptimizeSim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches1, g2o::Sim3& g2oS12,
                            const float th2, const bool bFixScale);
    static int nBAdone;
    void SetGroundPlane(Vector4d& normal);

private:
    std::map<int, std::vector<float>> mMapObjectConstrain;
    bool mbGroundPlaneSet;
    Vector4d mGroundPlaneNormal;

    friend struct OptimizerPeek;     // test access to the private members
};

}  // namespace ORB_SLAM2
#endif
