// mock_orbslam.h -- TEST STAND-INS for the handful of ORB-SLAM2 / OpenCV / Eigen types that
// include/qsp_optimizer_shim.h touches, so that the shim can be compiled and exercised in an image that has neither
// Eigen nor OpenCV.  Only names and member signatures the shim uses exist; behaviour is the minimum a map needs
// (the real classes: include/KeyFrame.h, MapPoint.h, MapObject.h, ObjectDetection.h, Map.h of the reference).
#pragma once
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#define CV_32F 5
namespace cv {
struct Mat {
    int rows = 0, cols = 0;
    std::shared_ptr<std::vector<float>> d;
    Mat() {}
    Mat(int r, int c, int) : rows(r), cols(c), d(new std::vector<float>(r * c, 0.f)) {}
    template <typename T> T& at(int r, int c = 0) { return (*d)[r * cols + c]; }
    template <typename T> const T& at(int r, int c = 0) const { return (*d)[r * cols + c]; }
    Mat clone() const { Mat m(rows, cols, CV_32F); *m.d = *d; return m; }
};
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; int octave; };
}  // namespace cv

namespace Eigen {
struct Matrix4f {
    float m[16];
    Matrix4f() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.f : 0.f; }
    float& operator()(int r, int c) { return m[4 * r + c]; }
    float operator()(int r, int c) const { return m[4 * r + c]; }
    Matrix4f inverse() const {   // rigid inverse is enough for the mock: [R t]^-1 = [R^T, -R^T t]
        Matrix4f o;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) o(i, j) = (*this)(j, i);
        for (int i = 0; i < 3; ++i) o(i, 3) = -(o(i, 0) * (*this)(0, 3) + o(i, 1) * (*this)(1, 3) + o(i, 2) * (*this)(2, 3));
        return o;
    }
};
}  // namespace Eigen

namespace ORB_SLAM2 {
class KeyFrame; class MapPoint; class MapObject;

struct ObjectDetection { Eigen::Matrix4f SE3Tco; };

class MapPoint {
public:
    unsigned long mnId = 0, mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0;
    cv::Mat pos, mPosGBA;
    bool bad = false;
    int n_updates = 0;
    std::map<KeyFrame*, size_t> obs;
    cv::Mat GetWorldPos() { return pos.clone(); }
    void SetWorldPos(const cv::Mat& p) { pos = p.clone(); }
    void UpdateNormalAndDepth() { ++n_updates; }
    bool isBad() { return bad; }
    std::map<KeyFrame*, size_t> GetObservations() { return obs; }
    void EraseObservation(KeyFrame* k) { obs.erase(k); }
    static std::mutex mGlobalMutex;
};
inline std::mutex MapPoint::mGlobalMutex;

class Frame {     // include/Frame.h of the reference: what Optimizer::PoseOptimization reads and writes
public:
    int N = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    cv::Mat mTcw;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    std::vector<float> mvInvLevelSigma2;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    void SetPose(const cv::Mat& T) { mTcw = T.clone(); }
};

class MapObject {
public:
    unsigned long mnId = 0, mnBALocalForKF = ~0ul, mnBAGlobalForKF = 0;
    Eigen::Matrix4f SE3Tow, SE3Two, mTwoGBA;
    bool dynamic = false, bad = false;
    std::map<KeyFrame*, size_t> obs;
    std::map<KeyFrame*, size_t> GetObservations() { return obs; }
    bool isDynamic() { return dynamic; }
    bool isBad() { return bad; }
    void SetObjectPoseSE3(const Eigen::Matrix4f& Two) { SE3Two = Two; SE3Tow = Two.inverse(); }
    void EraseObservation(KeyFrame* k) { obs.erase(k); }
};

class KeyFrame {
public:
    unsigned long mnId = 0, mnBALocalForKF = ~0ul, mnBAFixedForKF = ~0ul, mnBAGlobalForKF = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    cv::Mat Tcw, mTcwGBA;
    bool bad = false;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    std::vector<float> mvInvLevelSigma2;
    std::vector<KeyFrame*> covis;
    std::vector<MapPoint*> mps;
    std::vector<MapObject*> mos;
    std::vector<std::shared_ptr<ObjectDetection>> dets;
    cv::Mat GetPose() { return Tcw.clone(); }
    void SetPose(const cv::Mat& T) { Tcw = T.clone(); }
    bool isBad() { return bad; }
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return covis; }
    std::vector<MapPoint*> GetMapPointMatches() { return mps; }
    std::vector<MapObject*> GetMapObjectMatches() { return mos; }
    std::vector<std::shared_ptr<ObjectDetection>> GetObjectDetections() { return dets; }
    void EraseMapPointMatch(MapPoint* p) { for (auto& q : mps) if (q == p) q = nullptr; }
    void EraseMapObjectMatch(MapObject* o) { for (auto& q : mos) if (q == o) q = nullptr; }
};

class Map {
public:
    std::mutex mMutexMapUpdate;
    std::vector<KeyFrame*> kfs;
    std::vector<MapPoint*> mps;
    std::vector<MapObject*> mos;
    std::vector<KeyFrame*> GetAllKeyFrames() { return kfs; }
    std::vector<MapPoint*> GetAllMapPoints() { return mps; }
    std::vector<MapObject*> GetAllMapObjects() { return mos; }
};
}  // namespace ORB_SLAM2
