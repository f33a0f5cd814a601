// mock_scene.h -- builds a mock ORB-SLAM2 map (the stand-in types of mock_orbslam.h) from the flat scene file that
// tests/test_shim.py writes.  Shared by shim_driver.cpp (calls OptimizerHip directly) and dropin_caller.cpp (calls
// `Optimizer::` as the reference's call sites do).
// scene.bin (all little-endian): int32 n_kf n_pt n_obj n_mono n_st n_oe; then the arrays written by tests/test_shim.py
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mock_orbslam.h"

namespace mock {
using namespace ORB_SLAM2;

template <typename T> static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

struct Scene {
    Map map;
    std::vector<KeyFrame> kfs;
    std::vector<MapPoint> pts;
    std::vector<MapObject> objs;
    std::vector<float> sig;
    int n_kf = 0, n_pt = 0, n_obj = 0;

    bool load(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    auto hdr = rd<int32_t>(f, 6);
    n_kf = hdr[0]; n_pt = hdr[1]; n_obj = hdr[2];
    const int n_mono = hdr[3], n_st = hdr[4], n_oe = hdr[5];
    auto kfT = rd<float>(f, 16 * n_kf);        // Tcw 4x4 float32
    auto kf_id = rd<int64_t>(f, n_kf);
    auto kfK = rd<float>(f, 5 * n_kf);
    auto kf_local = rd<int32_t>(f, n_kf);      // 1 = local key-frame (covisible with kf 0 of the list), 0 = fixed camera
    auto ptX = rd<float>(f, 3 * n_pt);
    auto pt_mn = rd<int64_t>(f, n_pt);
    auto objT = rd<float>(f, 16 * n_obj);      // Tow
    auto obj_mn = rd<int64_t>(f, n_obj);
    auto mono_pt = rd<int32_t>(f, n_mono); auto mono_kf = rd<int32_t>(f, n_mono);
    auto mono_obs = rd<float>(f, 2 * n_mono); auto mono_oct = rd<int32_t>(f, n_mono);
    auto st_pt = rd<int32_t>(f, n_st); auto st_kf = rd<int32_t>(f, n_st);
    auto st_obs = rd<float>(f, 3 * n_st); auto st_oct = rd<int32_t>(f, n_st);
    auto oe_kf = rd<int32_t>(f, n_oe); auto oe_obj = rd<int32_t>(f, n_oe);
    auto oe_Z = rd<float>(f, 16 * n_oe);
    fclose(f);

    kfs.resize(n_kf); pts.resize(n_pt); objs.resize(n_obj); sig.resize(8);
    for (int o = 0; o < 8; ++o) sig[o] = 1.0f / std::pow(1.2f, 2.0f * o);
    for (int i = 0; i < n_kf; ++i) {
        KeyFrame& k = kfs[i];
        k.mnId = (unsigned long)kf_id[i];
        k.Tcw = cv::Mat(4, 4, CV_32F);
        for (int e = 0; e < 16; ++e) (*k.Tcw.d)[e] = kfT[16 * i + e];
        k.fx = kfK[5 * i]; k.fy = kfK[5 * i + 1]; k.cx = kfK[5 * i + 2]; k.cy = kfK[5 * i + 3]; k.mbf = kfK[5 * i + 4];
        k.mvInvLevelSigma2 = sig;
        map.kfs.push_back(&k);
    }
    for (int i = 1; i < n_kf; ++i)
        if (kf_local[i]) kfs[0].covis.push_back(&kfs[i]);
    for (int i = 0; i < n_pt; ++i) {
        pts[i].mnId = (unsigned long)pt_mn[i];
        pts[i].pos = cv::Mat(3, 1, CV_32F);
        for (int e = 0; e < 3; ++e) (*pts[i].pos.d)[e] = ptX[3 * i + e];
        map.mps.push_back(&pts[i]);
    }
    auto add_obs = [&](int pt, int kf, float u, float v, float ur, int oct) {
        KeyFrame& k = kfs[kf];
        const size_t idx = k.mvKeysUn.size();
        k.mvKeysUn.push_back(cv::KeyPoint{{u, v}, oct});
        k.mvuRight.push_back(ur);
        k.mps.push_back(&pts[pt]);
        pts[pt].obs[&k] = idx;
    };
    for (int e = 0; e < n_mono; ++e) add_obs(mono_pt[e], mono_kf[e], mono_obs[2 * e], mono_obs[2 * e + 1], -1.f, mono_oct[e]);
    for (int e = 0; e < n_st; ++e) add_obs(st_pt[e], st_kf[e], st_obs[3 * e], st_obs[3 * e + 1], st_obs[3 * e + 2], st_oct[e]);
    for (int i = 0; i < n_obj; ++i) {
        objs[i].mnId = (unsigned long)obj_mn[i];
        for (int e = 0; e < 16; ++e) objs[i].SE3Tow.m[e] = objT[16 * i + e];
        map.mos.push_back(&objs[i]);
    }
    for (int e = 0; e < n_oe; ++e) {
        KeyFrame& k = kfs[oe_kf[e]];
        auto det = std::make_shared<ObjectDetection>();
        for (int q = 0; q < 16; ++q) det->SE3Tco.m[q] = oe_Z[16 * e + q];
        objs[oe_obj[e]].obs[&k] = k.dets.size();
        k.dets.push_back(det);
        bool have = false;
        for (auto* m : k.mos) have |= (m == &objs[oe_obj[e]]);
        if (!have) k.mos.push_back(&objs[oe_obj[e]]);
    }
    return true;
    }
};
}  // namespace mock
