// shim_driver.cpp -- builds a mock ORB-SLAM2 map from a flat scene file, runs the shim's LocalJointBundleAdjustment on it
// (against whatever libqsp the binary is linked with: the recording stub on CPU, the real library on the GPU box) and
// dumps the map state afterwards.
//   usage: shim_driver <scene.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define QSP_SHIM_MOCK_TYPES 1
#include "mock_scene.h"
#include "qsp_optimizer_shim.h"

using namespace ORB_SLAM2;

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    mock::Scene S;
    if (!S.load(argv[1])) return 1;
    Map& map = S.map;
    std::vector<KeyFrame>& kfs = S.kfs;
    std::vector<MapPoint>& pts = S.pts;
    std::vector<MapObject>& objs = S.objs;
    std::vector<float>& sig = S.sig;
    const int n_kf = S.n_kf, n_pt = S.n_pt, n_obj = S.n_obj;
    bool stop = false;
    const std::string mode = argc > 3 ? argv[3] : "local";
    const unsigned long nLoopKF = argc > 4 ? (unsigned long)atol(argv[4]) : 0;
    if (mode == "local") OptimizerHip::LocalJointBundleAdjustment(&kfs[0], &stop, &map);
    else if (mode == "global_joint") OptimizerHip::GlobalJointBundleAdjustemnt(&map, 10, &stop, nLoopKF, true);
    else if (mode == "global_points") OptimizerHip::GlobalBundleAdjustemnt(&map, 20, &stop, nLoopKF, false);
    else if (mode == "pose") {
        // a Frame made of key-frame 0's observations: slots in key-point order, every 5th slot without a map point
        KeyFrame& k = kfs[0];
        Frame fr;
        fr.fx = k.fx; fr.fy = k.fy; fr.cx = k.cx; fr.cy = k.cy; fr.mbf = k.mbf;
        fr.mTcw = k.Tcw.clone();
        fr.mvInvLevelSigma2 = sig;
        for (size_t i = 0; i < k.mvKeysUn.size(); ++i) {
            fr.mvKeysUn.push_back(k.mvKeysUn[i]);
            fr.mvuRight.push_back(k.mvuRight[i]);
            fr.mvpMapPoints.push_back(i % 5 == 4 ? nullptr : k.mps[i]);
            fr.mvbOutlier.push_back(true);                 // stale flags must be cleared for matched slots only
        }
        fr.N = (int)fr.mvKeysUn.size();
        const int ninl = OptimizerHip::PoseOptimization(&fr);
        FILE* o = fopen(argv[2], "wb");
        fwrite(fr.mTcw.d->data(), sizeof(float), 16, o);
        int32_t hdr[2] = {ninl, fr.N};
        fwrite(hdr, sizeof(hdr), 1, o);
        for (int i = 0; i < fr.N; ++i) { unsigned char b = fr.mvbOutlier[i] ? 1 : 0; fwrite(&b, 1, 1, o); }
        fclose(o);
        return 0;
    }
    else return 2;
    if (nLoopKF != 0) {   // loop-closing mode: results are parked in the *GBA members; move them over for the output
        for (int i = 0; i < n_kf; ++i)
            if (kfs[i].mnBAGlobalForKF == nLoopKF) kfs[i].Tcw = kfs[i].mTcwGBA.clone();
        for (int i = 0; i < n_pt; ++i)
            if (pts[i].mnBAGlobalForKF == nLoopKF) pts[i].pos = pts[i].mPosGBA.clone();
        for (int i = 0; i < n_obj; ++i)
            if (objs[i].mnBAGlobalForKF == nLoopKF) objs[i].SE3Tow = objs[i].mTwoGBA.inverse();
    }

    FILE* o = fopen(argv[2], "wb");
    for (int i = 0; i < n_kf; ++i) fwrite(kfs[i].Tcw.d->data(), sizeof(float), 16, o);
    for (int i = 0; i < n_pt; ++i) fwrite(pts[i].pos.d->data(), sizeof(float), 3, o);
    for (int i = 0; i < n_obj; ++i) fwrite(objs[i].SE3Tow.m, sizeof(float), 16, o);
    int32_t nobs = 0;
    for (int i = 0; i < n_pt; ++i) nobs += (int32_t)pts[i].obs.size();
    fwrite(&nobs, sizeof(nobs), 1, o);
    int32_t nba = OptimizerHip::nBAdone();
    fwrite(&nba, sizeof(nba), 1, o);
    fclose(o);
    return 0;
}
