// orbslam3_shim.hpp -- header-only C++ adapter from the C ABI (orbslam3_hip.h) to the reference's own call signatures,
// so that Tracking / LocalMapping / Frame compile and run unchanged.
//
// It is written against the reference's real types (cv::Mat, cv::KeyPoint, ORB_SLAM3::Frame, KeyFrame, MapPoint, Map,
// Sophus::SE3f) and is therefore only compilable INSIDE an ORB-SLAM3 tree that has OpenCV / Eigen / Sophus -- which this
// image lacks (that is why the repo's own tests drive the C ABI through ctypes instead).  Define ORBSLAM3_HIP_WITH_REFERENCE
// before including it from the reference tree; without the macro the header only provides the POD-level C++ wrappers that
// need nothing but the C ABI, which IS compiled by tests/test_shim_compiles.py.
//
// Reference signatures mirrored here:
//   int  ORBextractor::operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray,
//                                 std::vector<int>& vLappingArea)                       include/ORBextractor.h:57-59
//   int  ORBmatcher::SearchByBoW(KeyFrame*, Frame&, std::vector<MapPoint*>&)            include/ORBmatcher.h:56
//   int  ORBmatcher::SearchByProjection(Frame&, const std::vector<MapPoint*>&, float, bool, float)   :43
//   int  ORBmatcher::SearchByProjection(Frame&, const Frame&, float, bool)               :47
//   static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)            :40
//   static void Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*, int&, int&, int&, int&)     include/Optimizer.h:58
//   static void Optimizer::LocalInertialBA(KeyFrame*, bool*, Map*, int&, int&, int&, int&, bool, bool)   include/Optimizer.h (src/Optimizer.cc:2383)
//   static void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>&, const std::vector<MapPoint*>&, int, bool*, unsigned long, bool)   include/Optimizer.h:50-52
//   static void Optimizer::GlobalBundleAdjustemnt(Map*, int, bool*, unsigned long, bool)             include/Optimizer.h:53-54
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbslam3_hip.h"

namespace orbslam3_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c) : std::runtime_error(std::string("orbslam3_hip: ") + orbx_last_error()), code(c) {}
};
inline int check(int rc) { if (rc < 0 && rc != ORBX_ERR_EMPTY) throw Error(rc); return rc; }

// ---------------------------------------------------------------------------------------------------------------
// POD-level wrappers (need only the C ABI)
// ---------------------------------------------------------------------------------------------------------------
class Extractor {
public:
    Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
    { check(orbx_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, &h_)); }
    ~Extractor() { orbx_destroy(h_); }
    Extractor(const Extractor&) = delete;
    Extractor& operator=(const Extractor&) = delete;

    // returns the reference's return value (monoIndex, -1 for an empty image)
    int extract(const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                std::vector<OrbxKeyPoint>& kps, std::vector<uint8_t>& desc)
    {
        int cap = orbx_max_keypoints(h_);
        int n = 0, mono = -1, rc = ORBX_OK;
        for (int attempt = 0; attempt < 2; attempt++) {
            kps.resize(cap);
            desc.resize((size_t)cap * 32);
            rc = orbx_extract(h_, img, w, h, stride, lap0, lap1, kps.data(), desc.data(), cap, &n, &mono);
            // orbx_max_keypoints() bounds what an image of up to 8.5 : 1 returns; a wider one with a tiny per-level budget returns 4 key
            // points per octree root (src/ORBextractor.cc:606-672) and reports the count it needs: once more with that
            if (rc != ORBX_ERR_CAPACITY || n <= cap) break;
            cap = n;
        }
        if (rc == ORBX_ERR_EMPTY) { kps.clear(); desc.clear(); return -1; }
        check(rc);
        kps.resize(n);
        desc.resize((size_t)n * 32);
        return mono;
    }
    int GetLevels() const { return orbx_levels(h_); }
    float GetScaleFactor() const { return orbx_scale_factor(h_); }
    std::vector<float> GetScaleFactors() const { return table(0); }
    std::vector<float> GetInverseScaleFactors() const { return table(1); }
    std::vector<float> GetScaleSigmaSquares() const { return table(2); }
    std::vector<float> GetInverseScaleSigmaSquares() const { return table(3); }
    orbx_extractor* handle() { return h_; }

private:
    std::vector<float> table(int which) const
    {
        std::vector<float> t[4];
        for (auto& v : t) v.resize(orbx_levels(h_));
        orbx_scale_tables(h_, t[0].data(), t[1].data(), t[2].data(), t[3].data());
        return t[which];
    }
    orbx_extractor* h_ = nullptr;
};

// DBoW2::FeatureVector (std::map<unsigned, std::vector<unsigned>>-like) -> CSR
template <class FeatVecMap>
struct FlatFeatVec {
    std::vector<uint32_t> node, feat;
    std::vector<int32_t> off;
    OrbmFeatVec view;
    explicit FlatFeatVec(const FeatVecMap& fv)
    {
        off.push_back(0);
        for (auto it = fv.begin(); it != fv.end(); ++it) {
            node.push_back((uint32_t)it->first);
            for (unsigned f : it->second) feat.push_back(f);
            off.push_back((int32_t)feat.size());
        }
        view.n_nodes = (int32_t)node.size();
        view.node_id = node.data(); view.offset = off.data(); view.feat = feat.data();
    }
};

// One edge-SLAM packet (class SlamPktVI, include/Socket/slampkt_vi.h) through the device codec, POD in / POD out.
class EdgePacketCodec {
public:
    explicit EdgePacketCodec(int device = 0) { check(orbe_create(device, &h_)); }
    ~EdgePacketCodec() { orbe_destroy(h_); }
    EdgePacketCodec(const EdgePacketCodec&) = delete;
    EdgePacketCodec& operator=(const EdgePacketCodec&) = delete;

    // SlamPktVI(id, timestamp, kps, descriptors, imus) :127-167; head = getHead() :185-193
    std::vector<uint8_t> pack(int32_t frame_id, int64_t timestamp, const OrbxKeyPoint* kps, const uint8_t* desc, int n_pts,
                              const OrbeImuSample* imu, int n_imu, uint8_t head[2] = nullptr)
    {
        const int total = orbe_packet_bytes(n_pts, n_imu), stride = (total + 3) & ~3;
        std::vector<uint8_t> payload((size_t)stride);
        const int32_t off[2] = {0, n_imu};
        const OrbxKeyPoint none_k = {};
        const uint8_t none_d[32] = {0};
        int32_t n = n_pts, len = 0, status = 0;
        check(orbe_pack_batch(h_, n_pts ? kps : &none_k, n_pts ? desc : none_d, &n, 1, n_pts ? n_pts : 1, &frame_id, &timestamp,
                              n_imu ? imu : nullptr, n_imu ? off : nullptr, payload.data(), stride, &len, head, &status));
        if (status == ORBX_ERR_CAPACITY) throw Error(status);
        payload.resize((size_t)len);
        return payload;
    }

    // SlamPktVI(buffer, packet_size) :85-125
    void unpack(const uint8_t* buffer, int packet_size, int32_t& frame_id, int64_t& timestamp, std::vector<OrbxKeyPoint>& kps,
                std::vector<uint8_t>& desc, std::vector<OrbeImuSample>& imu)
    {
        if (packet_size < 16) throw Error(ORBX_ERR_ARG);
        const int n_pts = buffer[12] * 256 + buffer[13], n_imu = buffer[14] * 256 + buffer[15];      // capacities only
        const int stride = (packet_size + 3) & ~3;
        std::vector<uint8_t> padded((size_t)stride, 0);
        std::memcpy(padded.data(), buffer, (size_t)packet_size);
        kps.assign((size_t)(n_pts ? n_pts : 1), OrbxKeyPoint());
        desc.assign((size_t)(n_pts ? n_pts : 1) * 32, 0);
        imu.assign((size_t)(n_imu ? n_imu : 1), OrbeImuSample());
        int32_t len = packet_size, n = 0, m = 0, status = 0;
        check(orbe_unpack_batch(h_, padded.data(), stride, &len, 1, (int)kps.size(), (int)imu.size(), kps.data(), desc.data(), &n, &frame_id,
                                &timestamp, imu.data(), &m, &status));
        if (status < 0) throw Error(status);
        kps.resize((size_t)n); desc.resize((size_t)n * 32); imu.resize((size_t)m);
    }

private:
    orbe_codec* h_ = nullptr;
};

}  // namespace orbslam3_hip

// ---------------------------------------------------------------------------------------------------------------
// Reference-typed adapters (compile inside an ORB-SLAM3 tree only)
// ---------------------------------------------------------------------------------------------------------------
#ifdef ORBSLAM3_HIP_WITH_REFERENCE

#include <algorithm>
#include <cassert>
#include <cmath>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <opencv2/core/core.hpp>

#include "Frame.h"
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "ORBmatcher.h"     // the reference entry points the adapters fall back to for rigs outside the accelerated path
#include "Optimizer.h"

namespace ORB_SLAM3 {

// Drop-in for ORB_SLAM3::ORBextractor (same public surface incl. mvImagePyramid, filled lazily on request).
class ORBextractorHIP {
public:
    ORBextractorHIP(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : ex_(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) { mvImagePyramid.resize(nlevels); }

    int operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                   cv::OutputArray _descriptors, std::vector<int>& vLappingArea)
    {
        if (_image.empty()) return -1;
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);
        std::vector<OrbxKeyPoint> kps;
        std::vector<uint8_t> desc;
        const int mono = ex_.extract(image.data, image.cols, image.rows, (int)image.step, vLappingArea[0], vLappingArea[1], kps, desc);
        static_assert(sizeof(cv::KeyPoint) == sizeof(OrbxKeyPoint), "cv::KeyPoint layout");
        _keypoints.resize(kps.size());
        if (!kps.empty()) std::memcpy((void*)_keypoints.data(), kps.data(), kps.size() * sizeof(OrbxKeyPoint));
        if (kps.empty()) _descriptors.release();
        else {
            _descriptors.create((int)kps.size(), 32, CV_8U);
            std::memcpy(_descriptors.getMat().data, desc.data(), desc.size());
        }
        pyramid_valid_ = false;
        return mono;
    }
    // Frame::ComputeStereoMatches reads mvImagePyramid (src/Frame.cc:938,1028,1043): materialise it on demand, with the
    // EDGE_THRESHOLD border exactly as ComputePyramid lays it out (level image = ROI inside the bordered buffer).
    void SyncPyramid()
    {
        if (pyramid_valid_) return;
        for (int l = 0; l < ex_.GetLevels(); l++) {
            int w, h;
            orbslam3_hip::check(orbx_pyramid_level_size(ex_.handle(), l, &w, &h));
            cv::Mat temp(h + 38, w + 38, CV_8UC1);
            orbslam3_hip::check(orbx_pyramid_level(ex_.handle(), 0, l, 19, temp.data, (int)temp.step));
            mvImagePyramid[l] = temp(cv::Rect(19, 19, w, h));
        }
        pyramid_valid_ = true;
    }
    // void Frame::ComputeStereoMatches() (src/Frame.cc:931-1101) with this = mpORBextractorLeft: both pyramids stay in HBM
    // (no SyncPyramid), the frame gets mvuRight / mvDepth.
    void ComputeStereoMatches(ORBextractorHIP& right, Frame& F)
    {
        F.mvuRight.assign(F.N, -1.0f);
        F.mvDepth.assign(F.N, -1.0f);
        if (F.N == 0) return;
        static_assert(sizeof(cv::KeyPoint) == sizeof(OrbxKeyPoint), "cv::KeyPoint layout");
        orbslam3_hip::check(orbx_stereo_matches(ex_.handle(), right.ex_.handle(), 0,
                                                (const OrbxKeyPoint*)F.mvKeys.data(), F.mDescriptors.data, F.N,
                                                (const OrbxKeyPoint*)F.mvKeysRight.data(), F.mDescriptorsRight.data, (int)F.mvKeysRight.size(),
                                                F.mb, F.mbf, F.mvuRight.data(), F.mvDepth.data()));
    }
    int GetLevels() { return ex_.GetLevels(); }
    float GetScaleFactor() { return ex_.GetScaleFactor(); }
    std::vector<float> GetScaleFactors() { return ex_.GetScaleFactors(); }
    std::vector<float> GetInverseScaleFactors() { return ex_.GetInverseScaleFactors(); }
    std::vector<float> GetScaleSigmaSquares() { return ex_.GetScaleSigmaSquares(); }
    std::vector<float> GetInverseScaleSigmaSquares() { return ex_.GetInverseScaleSigmaSquares(); }
    std::vector<cv::Mat> mvImagePyramid;

private:
    orbslam3_hip::Extractor ex_;
    bool pyramid_valid_ = false;
};

// Drop-in for the ORBmatcher calls on the hot path.
class ORBmatcherHIP {
public:
    ORBmatcherHIP(float nnratio = 0.6, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri)
    { orbslam3_hip::check(orbm_create(0, &m_)); }
    ~ORBmatcherHIP() { orbm_destroy(m_); }

    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbm_hamming(a.ptr<uint8_t>(), b.ptr<uint8_t>()); }

    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
    {
        const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
        const int nKF = (int)vpMapPointsKF.size(), nF = F.N;
        std::vector<uint8_t> valid(nKF);
        std::vector<float> angKF(nKF), angF(nF);
        for (int i = 0; i < nKF; i++) { valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad(); angKF[i] = pKF->mvKeysUn[i].angle; }
        for (int i = 0; i < nF; i++) angF[i] = F.mvKeys[i].angle;
        orbslam3_hip::FlatFeatVec<DBoW2::FeatureVector> fvKF(pKF->mFeatVec), fvF(F.mFeatVec);
        std::vector<int32_t> match(nF, -1);
        const int n = orbslam3_hip::check(orbm_search_by_bow(m_, pKF->mDescriptors.data, nKF, valid.data(), angKF.data(), &fvKF.view,
                                                              F.mDescriptors.data, nF, angF.data(), &fvF.view,
                                                              mfNNratio, mbCheckOrientation, match.data()));
        vpMapPointMatches.assign(nF, static_cast<MapPoint*>(NULL));
        for (int f = 0; f < nF; f++) if (match[f] >= 0) vpMapPointMatches[f] = vpMapPointsKF[match[f]];
        return n;
    }

    // SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) (src/ORBmatcher.cc:43-213): monocular and
    // rectified-stereo / RGB-D frames (F.Nleft == -1; the mvuRight gate :92-98 runs on the device); stereo-fisheye frames keep
    // the reference.
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const bool bFarPoints, const float thFarPoints)
    {
        if (F.Nleft != -1) return ORBmatcher(mfNNratio, mbCheckOrientation).SearchByProjection(F, vpMapPoints, th, bFarPoints, thFarPoints);
        const int nMP = (int)vpMapPoints.size(), nF = F.N;
        std::vector<uint8_t> inView(nMP), hasObs(nMP), bad(nMP), desc((size_t)nMP * 32), occ(nF);
        std::vector<float> u(nMP), v(nMP), ur(nMP), vc(nMP), depth(nMP);
        std::vector<int32_t> lvl(nMP), assign(nF, -2);
        for (int i = 0; i < nMP; i++) {
            MapPoint* p = vpMapPoints[i];
            inView[i] = p->mbTrackInView; u[i] = p->mTrackProjX; v[i] = p->mTrackProjY; ur[i] = p->mTrackProjXR; lvl[i] = p->mnTrackScaleLevel;
            vc[i] = p->mTrackViewCos; depth[i] = p->mTrackDepth; bad[i] = p->isBad(); hasObs[i] = p->Observations() > 0;
            const cv::Mat d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        OrbmFrame f = frameView(F, occ);
        const int n = orbslam3_hip::check(orbm_search_by_projection(m_, &f, nMP, inView.data(), u.data(), v.data(), ur.data(), lvl.data(), vc.data(),
                                                                     depth.data(), desc.data(), hasObs.data(), bad.data(),
                                                                     th, bFarPoints, thFarPoints, mfNNratio, assign.data(), occ.data()));
        for (int i = 0; i < nF; i++) if (assign[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[assign[i]];
        return n;
    }

    // SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1676-1887): bMono is
    // honoured -- a stereo / RGB-D caller gets the forward / backward level windows (:1692-1693, :1728-1733) and the ur gate
    // (:1751-1757); stereo-fisheye frames (Nleft != -1) keep the reference.
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
    {
        if (CurrentFrame.Nleft != -1 || LastFrame.Nleft != -1) return ORBmatcher(mfNNratio, mbCheckOrientation).SearchByProjection(CurrentFrame, LastFrame, th, bMono);
        const int nL = LastFrame.N, nF = CurrentFrame.N;
        const Sophus::SE3f Tcw = CurrentFrame.GetPose();
        const Eigen::Vector3f twc = Tcw.inverse().translation();                   // :1686-1693
        const Sophus::SE3f Tlw = LastFrame.GetPose();
        const Eigen::Vector3f tlc = Tlw * twc;
        const bool bForward = tlc(2) > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc(2) > CurrentFrame.mb && !bMono;
        const int levelWindow = bForward ? ORBM_LEVELS_FORWARD : bBackward ? ORBM_LEVELS_BACKWARD : ORBM_LEVELS_AROUND;
        std::vector<uint8_t> valid(nL, 0), hasObs(nL, 0), desc((size_t)nL * 32, 0), occ(nF);
        std::vector<float> u(nL, 0.f), v(nL, 0.f), ur(nL, 0.f), ang(nL, 0.f);
        std::vector<int32_t> oct(nL, 0), assign(nF, -2);      // -2 = untouched, -1 = nulled by the rotation check (:1878)
        for (int i = 0; i < nL; i++) {
            MapPoint* p = LastFrame.mvpMapPoints[i];
            if (!p || LastFrame.mvbOutlier[i]) continue;
            const Eigen::Vector3f x3Dc = Tcw * p->GetWorldPos();           // the reference's own float expressions (:1701-1710)
            const float invzc = 1.0 / x3Dc(2);
            if (invzc < 0) continue;
            const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
            valid[i] = 1; u[i] = uv(0); v[i] = uv(1); ur[i] = uv(0) - CurrentFrame.mbf * invzc;      // :1753
            oct[i] = LastFrame.mvKeys[i].octave; ang[i] = LastFrame.mvKeysUn[i].angle; hasObs[i] = p->Observations() > 0;
            const cv::Mat d = p->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        OrbmFrame f = frameView(CurrentFrame, occ);
        const int n = orbslam3_hip::check(orbm_search_by_projection_last(m_, &f, nL, valid.data(), u.data(), v.data(), ur.data(), oct.data(), ang.data(),
                                                                          desc.data(), hasObs.data(), th, levelWindow, mbCheckOrientation,
                                                                          assign.data(), occ.data()));
        for (int i = 0; i < nF; i++) {
            if (assign[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[assign[i]];
            else if (assign[i] == -1) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);
        }
        return n;
    }

    // SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist) (src/ORBmatcher.cc:1889-2010)
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist)
    {
        const Sophus::SE3f Tcw = CurrentFrame.GetPose();
        const Eigen::Vector3f Ow = Tcw.inverse().translation();
        const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
        const int nP = (int)vpMPs.size(), nF = CurrentFrame.N;
        std::vector<uint8_t> valid(nP, 0), desc((size_t)nP * 32, 0), occ(nF);
        std::vector<float> u(nP, 0.f), v(nP, 0.f), ang(nP, 0.f);
        std::vector<int32_t> lvl(nP, 0), assign(nF, -2);
        for (int i = 0; i < nP; i++) {
            MapPoint* pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            const Eigen::Vector3f x3Dw = pMP->GetWorldPos();                   // :1911-1935, the reference's float expressions
            const Eigen::Vector3f x3Dc = Tcw * x3Dw;
            const Eigen::Vector2f uv = CurrentFrame.mpCamera->project(x3Dc);
            const float dist3D = (x3Dw - Ow).norm();
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            valid[i] = 1; u[i] = uv(0); v[i] = uv(1);                         // the image-bounds test runs on the device
            lvl[i] = pMP->PredictScale(dist3D, &CurrentFrame); ang[i] = pKF->mvKeysUn[i].angle;
            const cv::Mat d = pMP->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        OrbmFrame f = frameView(CurrentFrame, occ);
        for (int i = 0; i < nF; i++) occ[i] = CurrentFrame.mvpMapPoints[i] != NULL;   // :1953 tests the pointer only
        const int n = orbslam3_hip::check(orbm_search_by_projection_kf(m_, &f, nP, valid.data(), u.data(), v.data(), lvl.data(), ang.data(),
                                                                        desc.data(), th, ORBdist, mbCheckOrientation, assign.data(), occ.data()));
        for (int i = 0; i < nF; i++) {
            if (assign[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[assign[i]];
            else if (assign[i] == -1) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);
        }
        return n;
    }

    // Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight = false) (src/ORBmatcher.cc:1148-1338), conventional cameras
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false)
    {
        if (bRight || pKF->NLeft != -1) return ORBmatcher(mfNNratio, mbCheckOrientation).Fuse(pKF, vpMapPoints, th, bRight);
        const Sophus::SE3f Tcw = pKF->GetPose();
        const Eigen::Vector3f Ow = pKF->GetCameraCenter();
        const float bf = pKF->mbf;
        const int nMPs = (int)vpMapPoints.size(), nK = pKF->N;
        std::vector<uint8_t> valid(nMPs, 0), desc((size_t)nMPs * 32, 0);
        std::vector<float> u(nMPs, 0.f), v(nMPs, 0.f), ur(nMPs, 0.f);
        std::vector<int32_t> lvl(nMPs, 0), bestIdx(std::max(nMPs, 1)), bestDist(std::max(nMPs, 1));
        for (int i = 0; i < nMPs; i++) {                                          // prelude :1176-1239
            MapPoint* pMP = vpMapPoints[i];
            if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            const Eigen::Vector3f p3Dw = pMP->GetWorldPos();
            const Eigen::Vector3f p3Dc = Tcw * p3Dw;
            if (p3Dc(2) < 0.0f) continue;
            const float invz = 1 / p3Dc(2);
            const Eigen::Vector2f uv = pKF->mpCamera->project(p3Dc);
            if (!pKF->IsInImage(uv(0), uv(1))) continue;
            const Eigen::Vector3f PO = p3Dw - Ow;
            const float dist3D = PO.norm();
            if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
            if (PO.dot(pMP->GetNormal()) < 0.5 * dist3D) continue;
            valid[i] = 1; u[i] = uv(0); v[i] = uv(1); ur[i] = uv(0) - bf * invz;
            lvl[i] = pMP->PredictScale(dist3D, pKF);
            const cv::Mat d = pMP->GetDescriptor();
            std::memcpy(&desc[(size_t)i * 32], d.data, 32);
        }
        x_.resize(nK); y_.resize(nK); o_.resize(nK);
        for (int i = 0; i < nK; i++) { x_[i] = pKF->mvKeysUn[i].pt.x; y_[i] = pKF->mvKeysUn[i].pt.y; o_[i] = pKF->mvKeysUn[i].octave; }
        OrbmFrame f;
        f.n = nK; f.x = x_.data(); f.y = y_.data(); f.octave = o_.data(); f.angle = NULL; f.desc = pKF->mDescriptors.data;
        f.min_x = pKF->mnMinX; f.min_y = pKF->mnMinY; f.max_x = pKF->mnMaxX; f.max_y = pKF->mnMaxY;
        f.grid_cols = pKF->mnGridCols; f.grid_rows = pKF->mnGridRows;
        f.scale_factors = pKF->mvScaleFactors.data(); f.n_levels = (int)pKF->mvScaleFactors.size(); f.u_right = NULL;
        orbslam3_hip::check(orbm_fuse_search(m_, &f, pKF->mvuRight.data(), pKF->mvInvLevelSigma2.data(), nMPs, valid.data(), u.data(), v.data(),
                                             ur.data(), lvl.data(), desc.data(), th, 1, bestIdx.data(), bestDist.data()));
        int nFused = 0;
        for (int i = 0; i < nMPs; i++) {                                          // :1311-1333, in order
            if (!valid[i] || bestIdx[i] < 0 || bestDist[i] > 50 /* TH_LOW */) continue;
            MapPoint* pMP = vpMapPoints[i];
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;    // :1187-1196 see the surgery of earlier iterations (duplicates, Replace)
            MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[i]);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx[i]);
                pKF->AddMapPoint(pMP, bestIdx[i]);
            }
            nFused++;
        }
        return nFused;
    }

    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:648-763)
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10)
    {
        const int n1 = (int)F1.mvKeysUn.size(), n2 = F2.N;
        std::vector<int32_t> oct1(n1), m12(std::max(n1, 1), -1);
        std::vector<float> a1(n1), px(n1), py(n1);
        for (int i = 0; i < n1; i++) { oct1[i] = F1.mvKeysUn[i].octave; a1[i] = F1.mvKeysUn[i].angle; px[i] = vbPrevMatched[i].x; py[i] = vbPrevMatched[i].y; }
        std::vector<uint8_t> occ(n2);
        OrbmFrame f2 = frameView(F2, occ);
        const int n = orbslam3_hip::check(orbm_search_for_initialization(m_, F1.mDescriptors.data, n1, oct1.data(), a1.data(), px.data(), py.data(),
                                                                          &f2, windowSize, mfNNratio, mbCheckOrientation, m12.data()));
        vnMatches12.assign(m12.begin(), m12.begin() + n1);
        for (int i = 0; i < n1; i++) if (vnMatches12[i] >= 0) vbPrevMatched[i] = F2.mvKeysUn[vnMatches12[i]].pt;     // :757-760
        return n;
    }

    // SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:907-1146), conventional cameras
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const bool bOnlyStereo, const bool bCoarse = false)
    {
        if (pKF1->mpCamera2 || pKF2->mpCamera2) return ORBmatcher(mfNNratio, mbCheckOrientation).SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse);
        const Sophus::SE3f T1w = pKF1->GetPose(), T2w = pKF2->GetPose(), Tw2 = pKF2->GetPoseInverse();
        const Eigen::Vector3f C2 = T2w * pKF1->GetCameraCenter();
        const Eigen::Vector2f ep = pKF2->mpCamera->project(C2);
        const Sophus::SE3f T12 = T1w * Tw2;
        const Eigen::Matrix3f F12 = pKF1->mpCamera->toK_().transpose().inverse() * Sophus::SO3f::hat(T12.translation()) * T12.rotationMatrix() *
                                    pKF2->mpCamera->toK_().inverse();                  // Pinhole.cpp:109-112
        float F[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F[r * 3 + c] = F12(r, c);
        struct Side { std::vector<uint8_t> mp, st; std::vector<float> x, y, a; std::vector<int32_t> o; };
        Side sd[2];
        KeyFrame* kf[2] = {pKF1, pKF2};
        OrbmTriSide ts[2];
        orbslam3_hip::FlatFeatVec<DBoW2::FeatureVector> fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
        for (int q = 0; q < 2; q++) {
            const int n = kf[q]->N;
            Side& s = sd[q];
            s.mp.resize(n); s.st.resize(n); s.x.resize(n); s.y.resize(n); s.a.resize(n); s.o.resize(n);
            for (int i = 0; i < n; i++) {
                s.mp[i] = kf[q]->GetMapPoint(i) != NULL; s.st[i] = kf[q]->mvuRight[i] >= 0;
                const cv::KeyPoint& kp = kf[q]->mvKeysUn[i];
                s.x[i] = kp.pt.x; s.y[i] = kp.pt.y; s.a[i] = kp.angle; s.o[i] = kp.octave;
            }
            ts[q].n = n; ts[q].desc = kf[q]->mDescriptors.data; ts[q].has_mp = s.mp.data(); ts[q].stereo = s.st.data();
            ts[q].x = s.x.data(); ts[q].y = s.y.data(); ts[q].octave = s.o.data(); ts[q].angle = s.a.data();
            ts[q].fv = q ? fv2.view : fv1.view;
        }
        std::vector<int32_t> m12(std::max(pKF1->N, 1), -1);
        const int n = orbslam3_hip::check(orbm_search_for_triangulation(m_, &ts[0], &ts[1], ep(0), ep(1), F, pKF2->mvLevelSigma2.data(),
                                                                         pKF2->mvScaleFactors.data(), (int)pKF2->mvScaleFactors.size(),
                                                                         bOnlyStereo, bCoarse, mbCheckOrientation, m12.data()));
        vMatchedPairs.clear();
        vMatchedPairs.reserve(n);
        for (int i = 0; i < pKF1->N; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));     // :1133-1143
        return n;
    }

    // SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (src/ORBmatcher.cc:1457-1674) = the Fuse search core (level window,
    // first minimum, no chi2 gate) once per direction with TH_HIGH, followed by the agreement check.
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const Sophus::Sim3f& S12, const float th)
    {
        const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;
        const Sophus::SE3f T1w = pKF1->GetPose(), T2w = pKF2->GetPose();
        const Sophus::Sim3f S21 = S12.inverse();
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {                                              // :1480-1490
            MapPoint* pMP = vpMatches12[i];
            if (!pMP) continue;
            vbAlreadyMatched1[i] = true;
            const int idx2 = std::get<0>(pMP->GetIndexInKeyFrame(pKF2));
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
        // one direction: project the map points of `from` into `to` (prelude :1500-1540 / :1586-1626), search, accept <= TH_HIGH
        auto direction = [&](const std::vector<MapPoint*>& pts, const std::vector<bool>& done, const Sophus::SE3f& Tfw, const Sophus::Sim3f& Sto,
                             KeyFrame* to, std::vector<int>& vnMatch) {
            const int n = (int)pts.size(), nK = to->N;
            std::vector<uint8_t> valid(n, 0), desc((size_t)n * 32, 0);
            std::vector<float> u(n, 0.f), v(n, 0.f);
            std::vector<int32_t> lvl(n, 0), bestIdx(std::max(n, 1)), bestDist(std::max(n, 1));
            for (int i = 0; i < n; i++) {
                MapPoint* pMP = pts[i];
                if (!pMP || done[i] || pMP->isBad()) continue;
                const Eigen::Vector3f p3Dc = Sto * (Tfw * pMP->GetWorldPos());
                if (p3Dc(2) < 0.0) continue;
                const float invz = 1.0 / p3Dc(2);
                const float uu = fx * (p3Dc(0) * invz) + cx, vv = fy * (p3Dc(1) * invz) + cy;
                if (!to->IsInImage(uu, vv)) continue;
                const float dist3D = p3Dc.norm();
                if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) continue;
                valid[i] = 1; u[i] = uu; v[i] = vv; lvl[i] = pMP->PredictScale(dist3D, to);
                const cv::Mat d = pMP->GetDescriptor();
                std::memcpy(&desc[(size_t)i * 32], d.data, 32);
            }
            x_.resize(nK); y_.resize(nK); o_.resize(nK);
            for (int i = 0; i < nK; i++) { x_[i] = to->mvKeysUn[i].pt.x; y_[i] = to->mvKeysUn[i].pt.y; o_[i] = to->mvKeysUn[i].octave; }
            OrbmFrame f;
            f.n = nK; f.x = x_.data(); f.y = y_.data(); f.octave = o_.data(); f.angle = NULL; f.desc = to->mDescriptors.data;
            f.min_x = to->mnMinX; f.min_y = to->mnMinY; f.max_x = to->mnMaxX; f.max_y = to->mnMaxY;
            f.grid_cols = to->mnGridCols; f.grid_rows = to->mnGridRows;
            f.scale_factors = to->mvScaleFactors.data(); f.n_levels = (int)to->mvScaleFactors.size(); f.u_right = NULL;
            orbslam3_hip::check(orbm_fuse_search(m_, &f, NULL, NULL, n, valid.data(), u.data(), v.data(), NULL, lvl.data(), desc.data(), th, 0,
                                                 bestIdx.data(), bestDist.data()));
            vnMatch.assign(n, -1);
            for (int i = 0; i < n; i++) if (valid[i] && bestIdx[i] >= 0 && bestDist[i] <= 100 /* TH_HIGH */) vnMatch[i] = bestIdx[i];
        };
        std::vector<int> vnMatch1, vnMatch2;
        direction(vpMapPoints1, vbAlreadyMatched1, T1w, S21, pKF2, vnMatch1);
        direction(vpMapPoints2, vbAlreadyMatched2, T2w, S12, pKF1, vnMatch2);
        int nFound = 0;                                                             // agreement :1655-1670
        for (int i1 = 0; i1 < N1; i1++) {
            const int idx2 = vnMatch1[i1];
            if (idx2 >= 0 && vnMatch2[idx2] == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
        }
        return nFound;
    }

private:
    OrbmFrame frameView(Frame& F, std::vector<uint8_t>& occ)
    {
        const int n = F.N;
        x_.resize(n); y_.resize(n); o_.resize(n); a_.resize(n);
        for (int i = 0; i < n; i++) {
            x_[i] = F.mvKeysUn[i].pt.x; y_[i] = F.mvKeysUn[i].pt.y; o_[i] = F.mvKeysUn[i].octave; a_[i] = F.mvKeysUn[i].angle;
            occ[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;
        }
        OrbmFrame f;
        f.n = n; f.x = x_.data(); f.y = y_.data(); f.octave = o_.data(); f.angle = a_.data(); f.desc = F.mDescriptors.data;
        f.min_x = F.mnMinX; f.min_y = F.mnMinY; f.max_x = F.mnMaxX; f.max_y = F.mnMaxY;
        f.grid_cols = FRAME_GRID_COLS; f.grid_rows = FRAME_GRID_ROWS;
        f.scale_factors = F.mvScaleFactors.data(); f.n_levels = (int)F.mvScaleFactors.size();
        f.u_right = (int)F.mvuRight.size() == n && n > 0 ? F.mvuRight.data() : NULL;       // all -1 for a monocular frame (src/Frame.cc:320)
        return f;
    }
    float mfNNratio;
    bool mbCheckOrientation;
    orbm_matcher* m_ = nullptr;
    std::vector<float> x_, y_, a_;
    std::vector<int32_t> o_;
};

// Drop-in for Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*, int&, int&, int&, int&) (src/Optimizer.cc:1116-1498).
// The pointer-graph walk and the map write-back are the reference's own logic; only the g2o part is replaced.
//
// LbaGraph = what the walk produces: the reference's three lists plus the flattened problem in g2o's vertex / edge order
// (B1 of SURVEY.md 8(a)).  Kept separate from the solve so that the walk can be exercised without a device
// (tests/test_shim_reference_typed.py runs it on a toy map built from stand-in types).
struct LbaGraph {
    std::list<KeyFrame*> lLocalKeyFrames, lFixedCameras;
    std::list<MapPoint*> lLocalMapPoints;
    std::vector<KeyFrame*> kfs;                 // all pose vertices, ascending mnId (g2o sorts active vertices by id, sparse_optimizer.cpp:482-487)
    std::vector<MapPoint*> mps;                 // all point vertices, ascending mnId (vertex id = mnId + maxKFid + 1, :1287)
    std::map<KeyFrame*, int> kfIndex;
    std::map<MapPoint*, int> mpIndex;
    std::vector<double> q, t, X;                // poses [n][4] qx qy qz qw, [n][3]; points [n][3]
    std::vector<uint8_t> fixed;
    std::vector<int32_t> ePoint, ePose;         // edges in addEdge order: map points in list order, observations in map order (:1278-1401)
    std::vector<double> eObs, eW;
    std::vector<uint8_t> eStereo;
    std::vector<KeyFrame*> eKF;
    std::vector<MapPoint*> eMP;
    double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
    int num_fixedKF = 0, num_OptKF = 0, num_edges = 0;
};

// true when every camera of the window is a plain pinhole without a second (fisheye-rig) camera: the only edges the device
// builds are EdgeSE3ProjectXYZ with Pinhole::project and EdgeStereoSE3ProjectXYZ (src/Optimizer.cc:1305-1364); windows with
// EdgeSE3ProjectXYZToBody (:1366-1396) or KannalaBrandt8 cameras stay with the reference.
inline bool LbaWindowIsPinhole(KeyFrame* pKF)
{
    if (pKF->mpCamera2 || !pKF->mpCamera || pKF->mpCamera->GetType() != GeometricCamera::CAM_PINHOLE) return false;
    for (KeyFrame* pKFi : pKF->GetVectorCovisibleKeyFrames())
        if (pKFi->mpCamera2 || !pKFi->mpCamera || pKFi->mpCamera->GetType() != GeometricCamera::CAM_PINHOLE) return false;
    return true;
}

// src/Optimizer.cc:1118-1404.  Returns false on the reference's silent early return (no fixed key frame, :1182-1186); the
// counters are then what the reference leaves in its out-parameters at that point.
inline bool LocalBundleAdjustmentGraph(KeyFrame* pKF, Map* pMap, LbaGraph& g)
{
    g.lLocalKeyFrames.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    Map* pCurrentMap = pKF->GetMap();
    const std::vector<KeyFrame*> vNeighKFs = pKF->GetVectorCovisibleKeyFrames();
    for (KeyFrame* pKFi : vNeighKFs) {
        pKFi->mnBALocalForKF = pKF->mnId;
        if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) g.lLocalKeyFrames.push_back(pKFi);
    }
    g.num_fixedKF = 0;
    for (KeyFrame* pKFi : g.lLocalKeyFrames) {
        if (pKFi->mnId == pMap->GetInitKFid()) g.num_fixedKF = 1;
        for (MapPoint* pMP : pKFi->GetMapPointMatches())
            if (pMP && !pMP->isBad() && pMP->GetMap() == pCurrentMap && pMP->mnBALocalForKF != pKF->mnId) {
                g.lLocalMapPoints.push_back(pMP);
                pMP->mnBALocalForKF = pKF->mnId;
            }
    }
    for (MapPoint* pMP : g.lLocalMapPoints)
        for (auto& obs : pMP->GetObservations()) {
            KeyFrame* pKFi = obs.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad() && pKFi->GetMap() == pCurrentMap) g.lFixedCameras.push_back(pKFi);
            }
        }
    g.num_fixedKF = (int)g.lFixedCameras.size() + g.num_fixedKF;
    if (g.num_fixedKF == 0) return false;                                           // :1182-1186

    g.kfs.assign(g.lLocalKeyFrames.begin(), g.lLocalKeyFrames.end());
    g.kfs.insert(g.kfs.end(), g.lFixedCameras.begin(), g.lFixedCameras.end());
    std::sort(g.kfs.begin(), g.kfs.end(), [](KeyFrame* a, KeyFrame* b) { return a->mnId < b->mnId; });
    g.mps.assign(g.lLocalMapPoints.begin(), g.lLocalMapPoints.end());
    std::sort(g.mps.begin(), g.mps.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
    g.q.resize(g.kfs.size() * 4); g.t.resize(g.kfs.size() * 3); g.X.resize(g.mps.size() * 3);
    g.fixed.resize(g.kfs.size());
    for (size_t i = 0; i < g.kfs.size(); i++) {
        g.kfIndex[g.kfs[i]] = (int)i;
        const Sophus::SE3<float> Tcw = g.kfs[i]->GetPose();
        const Eigen::Quaterniond qd = Tcw.unit_quaternion().cast<double>();
        const Eigen::Vector3d td = Tcw.translation().cast<double>();
        g.q[4 * i] = qd.x(); g.q[4 * i + 1] = qd.y(); g.q[4 * i + 2] = qd.z(); g.q[4 * i + 3] = qd.w();
        g.t[3 * i] = td.x(); g.t[3 * i + 1] = td.y(); g.t[3 * i + 2] = td.z();
        g.fixed[i] = g.kfs[i]->mnBALocalForKF != pKF->mnId || g.kfs[i]->mnId == pMap->GetInitKFid();     // :1220, :1237
    }
    g.num_OptKF = (int)g.lLocalKeyFrames.size();                                    // :1227
    for (size_t i = 0; i < g.mps.size(); i++) {
        g.mpIndex[g.mps[i]] = (int)i;
        const Eigen::Vector3d Xd = g.mps[i]->GetWorldPos().cast<double>();
        g.X[3 * i] = Xd.x(); g.X[3 * i + 1] = Xd.y(); g.X[3 * i + 2] = Xd.z();
    }
    for (MapPoint* pMP : g.lLocalMapPoints)
        for (auto& obs : pMP->GetObservations()) {
            KeyFrame* pKFi = obs.first;
            if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap) continue;
            const int leftIndex = std::get<0>(obs.second);
            if (leftIndex == -1) continue;
            const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
            const float ur = pKFi->mvuRight[leftIndex];
            g.ePoint.push_back(g.mpIndex.at(pMP)); g.ePose.push_back(g.kfIndex.at(pKFi));      // .at(): an observer outside the window is a bug, not vertex 0
            g.eObs.push_back(kpUn.pt.x); g.eObs.push_back(kpUn.pt.y); g.eObs.push_back(ur >= 0 ? (double)ur : -1.0);
            g.eW.push_back((double)pKFi->mvInvLevelSigma2[kpUn.octave]);
            g.eStereo.push_back(ur >= 0);
            g.eKF.push_back(pKFi); g.eMP.push_back(pMP);
            g.fx = pKFi->fx; g.fy = pKFi->fy; g.cx = pKFi->cx; g.cy = pKFi->cy; g.bf = pKFi->mbf;
        }
    g.num_edges = (int)g.ePoint.size();                                             // :1404
    return true;
}

inline void LocalBundleAdjustmentHIP(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges)
{
    if (!LbaWindowIsPinhole(pKF)) { Optimizer::LocalBundleAdjustment(pKF, pbStopFlag, pMap, num_fixedKF, num_OptKF, num_MPs, num_edges); return; }
    LbaGraph g;
    const bool ok = LocalBundleAdjustmentGraph(pKF, pMap, g);
    // the fixed cameras -- key frames outside the covisibility list that see local map points -- only turn up in the walk: one
    // of them with a second camera or a non-pinhole model would need EdgeSE3ProjectXYZToBody edges (:1366-1396) the device does
    // not build.  The walk has only stamped mnBALocalForKF / mnBAFixedForKF, which the reference rewrites anyway.
    for (KeyFrame* pKFi : g.kfs)
        if (pKFi->mpCamera2 || !pKFi->mpCamera || pKFi->mpCamera->GetType() != GeometricCamera::CAM_PINHOLE) {
            Optimizer::LocalBundleAdjustment(pKF, pbStopFlag, pMap, num_fixedKF, num_OptKF, num_MPs, num_edges);
            return;
        }
    (void)num_MPs;      // never assigned by the reference overload either
    num_fixedKF = g.num_fixedKF;
    if (!ok) return;                                                                // :1182-1186: the other counters keep the caller's values
    num_OptKF = g.num_OptKF; num_edges = g.num_edges;
    if (pbStopFlag && *pbStopFlag) return;                                          // :1406-1408

    LbaProblem pr;
    pr.n_poses = (int)g.kfs.size(); pr.pose_q = g.q.data(); pr.pose_t = g.t.data(); pr.pose_fixed = g.fixed.data();
    pr.n_points = (int)g.mps.size(); pr.points = g.X.data();
    pr.n_edges = num_edges; pr.edge_point = g.ePoint.data(); pr.edge_pose = g.ePose.data(); pr.edge_obs = g.eObs.data();
    pr.edge_inv_sigma2 = g.eW.data(); pr.edge_stereo = g.eStereo.data();
    pr.fx = g.fx; pr.fy = g.fy; pr.cx = g.cx; pr.cy = g.cy; pr.bf = g.bf;
    const float thHuberMono = sqrt(5.991), thHuberStereo = sqrt(7.815);             // :1275-1276 (through float)
    pr.huber_mono = thHuberMono; pr.huber_stereo = thHuberStereo;
    static thread_local lba_solver* solver = nullptr;
    if (!solver) orbslam3_hip::check(lba_create(0, &solver));
    std::vector<double> qo(g.q.size()), to(g.t.size()), Xo(g.X.size()), chi2(num_edges);
    std::vector<uint8_t> depthPos(num_edges);
    LbaStats st;
    orbslam3_hip::check(lba_solve(solver, &pr, (const volatile uint8_t*)pbStopFlag, 10, pMap->IsInertial() ? 100.0 : 0.0,
                                  qo.data(), to.data(), Xo.data(), chi2.data(), depthPos.data(), &st));

    std::vector<std::pair<KeyFrame*, MapPoint*> > vToErase;                         // :1413-1460
    for (int e = 0; e < num_edges; e++) {
        if (g.eMP[e]->isBad()) continue;
        if (chi2[e] > (g.eStereo[e] ? 7.815 : 5.991) || !depthPos[e]) vToErase.push_back(std::make_pair(g.eKF[e], g.eMP[e]));
    }
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                       // :1464
    for (auto& er : vToErase) { er.first->EraseMapPointMatch(er.second); er.second->EraseObservation(er.first); }
    for (KeyFrame* pKFi : g.lLocalKeyFrames) {
        const int i = g.kfIndex.at(pKFi);
        const Eigen::Quaterniond qd(qo[4 * i + 3], qo[4 * i], qo[4 * i + 1], qo[4 * i + 2]);
        pKFi->SetPose(Sophus::SE3f(qd.cast<float>(), Eigen::Vector3d(to[3 * i], to[3 * i + 1], to[3 * i + 2]).cast<float>()));
    }
    for (MapPoint* pMP : g.lLocalMapPoints) {
        const int i = g.mpIndex.at(pMP);
        pMP->SetWorldPos(Eigen::Vector3d(Xo[3 * i], Xo[3 * i + 1], Xo[3 * i + 2]).cast<float>());
        pMP->UpdateNormalAndDepth();
    }
    pMap->IncreaseChangeIndex();
}

// ---------------------------------------------------------------------------------------------------------------
// Drop-in for Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust) (include/Optimizer.h:50-52,
// src/Optimizer.cc:60-390) and Optimizer::GlobalBundleAdjustemnt(pMap, ...) (:52-58; callers src/LoopClosing.cc:2288 with
// bRobust = false and src/Tracking.cc:2722 with 20 iterations).  Differences from the local window: EVERY non-bad key frame
// of vpKFs is a free pose except the map's initial one (:125), Huber deltas sqrt(5.99) / sqrt(7.815) (:130-131) or no robust
// kernel at all, a map point without a usable observation drops out (vbNotIncludedMP, :63, :269-277), and the results go to
// mTcwGBA / mPosGBA / mnBAGlobalForKF unless nLoopKF is the origin key frame (:297-303, :376-388).
// This is also the multi-GPU entry of the path: with a GbaSharding every rank -- one host thread per GPU of
// the process that owns the map -- walks the same graph, keeps the map points [lo, hi) of its rank with all their edges,
// and the reduced camera system is summed by ONE all-reduce per Levenberg trial (lba_shard_optimize).
// ---------------------------------------------------------------------------------------------------------------
struct GbaSharding {
    int rank = 0, world = 1;        // this thread's share of the map points; world == 1: everything on `device`
    int device = 0;
    lba_allreduce_fn allreduce = nullptr;   // e.g. ncclAllReduce on the stream it is given
    void* user = nullptr;
};

struct GbaGraph {
    std::vector<KeyFrame*> kfs;             // pose vertices: the non-bad key frames of vpKFs, ascending mnId (g2o's active-vertex order)
    std::vector<MapPoint*> mps;             // point vertices that kept an edge, ascending mnId (vertex id = mnId + maxKFid + 1, :142)
    std::vector<bool> vbNotIncludedMP;      // indexed like vpMP (:63)
    std::map<KeyFrame*, int> kfIndex;
    std::map<MapPoint*, int> mpIndex;
    std::vector<double> q, t, X;
    std::vector<uint8_t> fixed;
    std::vector<int32_t> ePoint, ePose;     // edges in addEdge order: vpMP order, observations in map order (:148-265)
    std::vector<double> eObs, eW;
    std::vector<uint8_t> eStereo;
    double fx = 0, fy = 0, cx = 0, cy = 0, bf = 0;
    unsigned long maxKFid = 0;
    bool accelerated = true;                // false: an EdgeSE3ProjectXYZToBody / non-pinhole edge would be needed -> the reference
};

// src/Optimizer.cc:62-279
inline void BundleAdjustmentGraph(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, GbaGraph& g)
{
    g.vbNotIncludedMP.assign(vpMP.size(), false);
    Map* pMap = vpKFs[0]->GetMap();
    std::map<unsigned long, KeyFrame*> vertexOfId;                                  // optimizer.vertex(pKF->mnId)
    for (KeyFrame* pKF : vpKFs) {                                                   // :116-129
        if (pKF->isBad()) continue;
        vertexOfId[pKF->mnId] = pKF;
        if (pKF->mnId > g.maxKFid) g.maxKFid = pKF->mnId;
        if (pKF->mpCamera2 || !pKF->mpCamera || pKF->mpCamera->GetType() != GeometricCamera::CAM_PINHOLE) g.accelerated = false;
    }
    for (auto& kv : vertexOfId) { g.kfIndex[kv.second] = (int)g.kfs.size(); g.kfs.push_back(kv.second); }
    g.q.resize(g.kfs.size() * 4); g.t.resize(g.kfs.size() * 3); g.fixed.resize(g.kfs.size());
    for (size_t i = 0; i < g.kfs.size(); i++) {
        const Sophus::SE3<float> Tcw = g.kfs[i]->GetPose();
        const Eigen::Quaterniond qd = Tcw.unit_quaternion().cast<double>();
        const Eigen::Vector3d td = Tcw.translation().cast<double>();
        g.q[4 * i] = qd.x(); g.q[4 * i + 1] = qd.y(); g.q[4 * i + 2] = qd.z(); g.q[4 * i + 3] = qd.w();
        g.t[3 * i] = td.x(); g.t[3 * i + 1] = td.y(); g.t[3 * i + 2] = td.z();
        g.fixed[i] = g.kfs[i]->mnId == pMap->GetInitKFid();                         // :125
    }
    // first pass: which map points keep an edge (their vertex order is by id, their edges are in vpMP order)
    struct Obs { KeyFrame* kf; int leftIndex; };
    std::vector<std::vector<Obs> > usable(vpMP.size());
    std::vector<MapPoint*> included;
    for (size_t i = 0; i < vpMP.size(); i++) {
        MapPoint* pMP = vpMP[i];
        if (pMP->isBad()) continue;                                                 // :137-138 (vbNotIncludedMP stays false)
        for (auto& obs : pMP->GetObservations()) {                                  // :151-265
            KeyFrame* pKF = obs.first;
            if (pKF->isBad() || pKF->mnId > g.maxKFid) continue;
            auto it = vertexOfId.find(pKF->mnId);
            if (it == vertexOfId.end()) continue;                                   // optimizer.vertex(pKF->mnId) == NULL
            const int leftIndex = std::get<0>(obs.second);
            if (leftIndex == -1) { g.accelerated = false; continue; }               // a right-camera-only observation (:231-263)
            usable[i].push_back(Obs{it->second, leftIndex});
        }
        if (usable[i].empty()) g.vbNotIncludedMP[i] = true;                         // :269-273 optimizer.removeVertex(vPoint)
        else included.push_back(pMP);
    }
    g.mps = included;
    std::sort(g.mps.begin(), g.mps.end(), [](MapPoint* a, MapPoint* b) { return a->mnId < b->mnId; });
    g.X.resize(g.mps.size() * 3);
    for (size_t i = 0; i < g.mps.size(); i++) {
        g.mpIndex[g.mps[i]] = (int)i;
        const Eigen::Vector3d Xd = g.mps[i]->GetWorldPos().cast<double>();
        g.X[3 * i] = Xd.x(); g.X[3 * i + 1] = Xd.y(); g.X[3 * i + 2] = Xd.z();
    }
    for (size_t i = 0; i < vpMP.size(); i++)
        for (const Obs& o : usable[i]) {
            KeyFrame* pKF = o.kf;
            const cv::KeyPoint& kpUn = pKF->mvKeysUn[o.leftIndex];
            const float ur = pKF->mvuRight[o.leftIndex];
            g.ePoint.push_back(g.mpIndex.at(vpMP[i])); g.ePose.push_back(g.kfIndex.at(pKF));
            g.eObs.push_back(kpUn.pt.x); g.eObs.push_back(kpUn.pt.y); g.eObs.push_back(ur >= 0 ? (double)ur : -1.0);
            g.eW.push_back((double)pKF->mvInvLevelSigma2[kpUn.octave]);
            g.eStereo.push_back(ur >= 0);                                           // :160 mono iff mvuRight < 0, :196 stereo
            g.fx = pKF->fx; g.fy = pKF->fy; g.cx = pKF->cx; g.cy = pKF->cy; g.bf = pKF->mbf;
        }
}

inline void BundleAdjustmentHIP(const std::vector<KeyFrame*>& vpKFs, const std::vector<MapPoint*>& vpMP, int nIterations = 5, bool* pbStopFlag = NULL,
                                const unsigned long nLoopKF = 0, const bool bRobust = true, const GbaSharding* shard = nullptr)
{
    GbaGraph g;
    BundleAdjustmentGraph(vpKFs, vpMP, g);
    if (!g.accelerated) { Optimizer::BundleAdjustment(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust); return; }
    Map* pMap = vpKFs[0]->GetMap();
    const int rank = shard ? shard->rank : 0, world = shard ? shard->world : 1;

    // this rank's map points [lo, hi) of the vertex order, with all their edges (poses are replicated)
    const int nMP = (int)g.mps.size();
    const int base = nMP / world, rem = nMP % world;
    const int lo = rank * base + std::min(rank, rem), hi = lo + base + (rank < rem ? 1 : 0);
    std::vector<int32_t> ePoint, ePose;
    std::vector<double> eObs, eW;
    std::vector<uint8_t> eStereo;
    for (size_t e = 0; e < g.ePoint.size(); e++)
        if (g.ePoint[e] >= lo && g.ePoint[e] < hi) {
            ePoint.push_back(g.ePoint[e] - lo); ePose.push_back(g.ePose[e]);
            eObs.insert(eObs.end(), g.eObs.begin() + 3 * e, g.eObs.begin() + 3 * e + 3);
            eW.push_back(g.eW[e]); eStereo.push_back(g.eStereo[e]);
        }
    LbaProblem pr;
    pr.n_poses = (int)g.kfs.size(); pr.pose_q = g.q.data(); pr.pose_t = g.t.data(); pr.pose_fixed = g.fixed.data();
    pr.n_points = hi - lo; pr.points = g.X.data() + 3 * (size_t)lo;
    pr.n_edges = (int)ePoint.size(); pr.edge_point = ePoint.data(); pr.edge_pose = ePose.data(); pr.edge_obs = eObs.data();
    pr.edge_inv_sigma2 = eW.data(); pr.edge_stereo = eStereo.data();
    pr.fx = g.fx; pr.fy = g.fy; pr.cx = g.cx; pr.cy = g.cy; pr.bf = g.bf;
    const float thHuber2D = sqrt(5.99), thHuber3D = sqrt(7.815);                    // :130-131 (through float; 5.99, not 5.991)
    pr.huber_mono = bRobust ? (double)thHuber2D : 0.0; pr.huber_stereo = bRobust ? (double)thHuber3D : 0.0;

    lba_shard* sh = nullptr;
    orbslam3_hip::check(lba_shard_create(shard ? shard->device : 0, &pr, &sh));
    std::vector<double> qo(g.q.size()), to(g.t.size()), Xo(3 * (size_t)(hi - lo));
    LbaStats st;
    int rc = lba_shard_optimize(sh, world > 1 ? shard->allreduce : nullptr, shard ? shard->user : nullptr, world, nIterations, 0.0,
                                (const volatile uint8_t*)pbStopFlag, &st);           // :281-283 optimize(nIterations)
    if (rc == ORBX_OK) rc = lba_shard_download(sh, qo.data(), to.data(), Xo.data(), nullptr, nullptr);
    lba_shard_destroy(sh);
    orbslam3_hip::check(rc);

    const bool toMap = nLoopKF == pMap->GetOriginKF()->mnId;
    if (rank == 0)                                                                  // key frames: identical on every rank, written once (:287-371)
        for (KeyFrame* pKF : vpKFs) {
            if (pKF->isBad()) continue;
            const int i = g.kfIndex.at(pKF);
            const Eigen::Quaterniond qd(qo[4 * i + 3], qo[4 * i], qo[4 * i + 1], qo[4 * i + 2]);
            const Eigen::Vector3d td(to[3 * i], to[3 * i + 1], to[3 * i + 2]);
            if (toMap) pKF->SetPose(Sophus::SE3f(qd.cast<float>(), td.cast<float>()));
            else {
                pKF->mTcwGBA = Sophus::SE3d(qd, td).cast<float>();
                pKF->mnBAGlobalForKF = nLoopKF;
                // (:304-369 count edges of key frames that moved by more than 1 m into local variables nothing reads: no effect)
            }
        }
    for (size_t i = 0; i < vpMP.size(); i++) {                                      // :374-389, every rank its own map points
        if (g.vbNotIncludedMP[i]) continue;
        MapPoint* pMP = vpMP[i];
        if (pMP->isBad()) continue;
        const int j = g.mpIndex.at(pMP);
        if (j < lo || j >= hi) continue;
        const Eigen::Vector3d Xd(Xo[3 * (size_t)(j - lo)], Xo[3 * (size_t)(j - lo) + 1], Xo[3 * (size_t)(j - lo) + 2]);
        if (toMap) { pMP->SetWorldPos(Xd.cast<float>()); pMP->UpdateNormalAndDepth(); }
        else { pMP->mPosGBA = Xd.cast<float>(); pMP->mnBAGlobalForKF = nLoopKF; }
    }
}

inline void GlobalBundleAdjustemntHIP(Map* pMap, int nIterations = 5, bool* pbStopFlag = NULL, const unsigned long nLoopKF = 0, const bool bRobust = true,
                                      const GbaSharding* shard = nullptr)
{
    const std::vector<KeyFrame*> vpKFs = pMap->GetAllKeyFrames();                   // :54-56
    const std::vector<MapPoint*> vpMP = pMap->GetAllMapPoints();
    BundleAdjustmentHIP(vpKFs, vpMP, nIterations, pbStopFlag, nLoopKF, bRobust, shard);
}

// void Optimizer::LocalInertialBA(KeyFrame*, bool* pbStopFlag, Map*, int&, int&, int&, int&, bool bLarge, bool bRecInit)
// (src/Optimizer.cc:2383-2958), conventional cameras (pKF->mpCamera2 == nullptr): the graph walk and the write-back follow the
// reference statement by statement; the g2o part is liba_solve.  Needs "G2oTypes.h" / "ImuTypes.h" of the reference.
inline void LocalInertialBAHIP(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges,
                               bool bLarge = false, bool bRecInit = false)
{
    (void)pbStopFlag; (void)num_MPs;                                                // the reference sets the stop flag only AFTER optimize() (:2846)
    if (pKF->mpCamera2) { Optimizer::LocalInertialBA(pKF, pbStopFlag, pMap, num_fixedKF, num_OptKF, num_MPs, num_edges, bLarge, bRecInit); return; }
    Map* pCurrentMap = pKF->GetMap();
    const int maxOpt = bLarge ? 25 : 10, opt_it = bLarge ? 4 : 10;
    const int Nd = std::min((int)pCurrentMap->KeyFramesInMap() - 2, maxOpt);
    std::vector<KeyFrame*> vpOptimizableKFs;                                        // :2397-2413
    vpOptimizableKFs.push_back(pKF);
    pKF->mnBALocalForKF = pKF->mnId;
    for (int i = 1; i < Nd; i++) {
        if (!vpOptimizableKFs.back()->mPrevKF) break;
        vpOptimizableKFs.push_back(vpOptimizableKFs.back()->mPrevKF);
        vpOptimizableKFs.back()->mnBALocalForKF = pKF->mnId;
    }
    std::list<MapPoint*> lLocalMapPoints;                                           // :2418-2435
    for (KeyFrame* pKFi : vpOptimizableKFs)
        for (MapPoint* pMP : pKFi->GetMapPointMatches())
            if (pMP && !pMP->isBad() && pMP->mnBALocalForKF != pKF->mnId) { lLocalMapPoints.push_back(pMP); pMP->mnBALocalForKF = pKF->mnId; }
    std::list<KeyFrame*> lFixedKeyFrames;                                           // :2438-2450
    if (vpOptimizableKFs.back()->mPrevKF) {
        lFixedKeyFrames.push_back(vpOptimizableKFs.back()->mPrevKF);
        vpOptimizableKFs.back()->mPrevKF->mnBAFixedForKF = pKF->mnId;
    } else {
        vpOptimizableKFs.back()->mnBALocalForKF = 0;
        vpOptimizableKFs.back()->mnBAFixedForKF = pKF->mnId;
        lFixedKeyFrames.push_back(vpOptimizableKFs.back());
        vpOptimizableKFs.pop_back();
    }
    // maxCovKF = 0: no optimisable covisible key frames (:2453-2484).  Fixed key frames seeing the local points (:2487-2507)
    const size_t maxFixKF = 200;
    for (MapPoint* pMP : lLocalMapPoints) {
        for (auto& obs : pMP->GetObservations()) {
            KeyFrame* pKFi = obs.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) {
                pKFi->mnBAFixedForKF = pKF->mnId;
                if (!pKFi->isBad()) { lFixedKeyFrames.push_back(pKFi); break; }
            }
        }
        if (lFixedKeyFrames.size() >= maxFixKF) break;
    }
    const int N = (int)vpOptimizableKFs.size();
    num_OptKF = N; num_fixedKF = (int)lFixedKeyFrames.size();

    std::vector<KeyFrame*> kfs(vpOptimizableKFs.begin(), vpOptimizableKFs.end());
    kfs.insert(kfs.end(), lFixedKeyFrames.begin(), lFixedKeyFrames.end());
    std::map<KeyFrame*, int> kfIndex;
    const int nKF = (int)kfs.size();
    std::vector<double> Rwb(9 * nKF), twb(3 * nKF), vel(3 * nKF, 0.0), bg(3 * nKF, 0.0), ba(3 * nKF, 0.0);
    std::vector<uint8_t> poseFixed(nKF), hasImu(nKF), imuFixed(nKF);
    for (int i = 0; i < nKF; i++) {
        KeyFrame* k = kfs[i];
        kfIndex[k] = i;
        const Eigen::Matrix3d R = k->GetImuRotation().cast<double>();
        const Eigen::Vector3d t = k->GetImuPosition().cast<double>();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rwb[9 * i + 3 * r + c] = R(r, c); twb[3 * i + r] = t(r); }
        poseFixed[i] = imuFixed[i] = i >= N;                                        // :2541-2600
        hasImu[i] = k->bImu;
        if (k->bImu) {
            const Eigen::Vector3d v = k->GetVelocity().cast<double>(), g = k->GetGyroBias().cast<double>(), a = k->GetAccBias().cast<double>();
            for (int r = 0; r < 3; r++) { vel[3 * i + r] = v(r); bg[3 * i + r] = g(r); ba[3 * i + r] = a(r); }
        }
    }
    std::vector<LibaLink> links;                                                    // :2603-2672
    for (int i = 0; i < N; i++) {
        KeyFrame* pKFi = vpOptimizableKFs[i];
        if (!pKFi->mPrevKF || !kfIndex.count(pKFi->mPrevKF)) continue;
        if (!(pKFi->bImu && pKFi->mPrevKF->bImu && pKFi->mpImuPreintegrated)) continue;
        IMU::Preintegrated* pInt = pKFi->mpImuPreintegrated;
        pInt->SetNewBias(pKFi->mPrevKF->GetImuBias());
        LibaLink L;
        std::memset(&L, 0, sizeof(L));
        L.kf1 = kfIndex.at(pKFi->mPrevKF); L.kf2 = kfIndex.at(pKFi);
        auto put3x3 = [](float* dst, const Eigen::Matrix3f& M) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) dst[3 * r + c] = M(r, c); };
        put3x3(L.dR, pInt->dR); put3x3(L.JRg, pInt->JRg); put3x3(L.JVg, pInt->JVg); put3x3(L.JVa, pInt->JVa); put3x3(L.JPg, pInt->JPg); put3x3(L.JPa, pInt->JPa);
        for (int r = 0; r < 3; r++) { L.dV[r] = pInt->dV(r); L.dP[r] = pInt->dP(r); }
        L.dT = pInt->dT;
        const IMU::Bias b = pInt->GetOriginalBias();
        L.bias0[0] = b.bax; L.bias0[1] = b.bay; L.bias0[2] = b.baz; L.bias0[3] = b.bwx; L.bias0[4] = b.bwy; L.bias0[5] = b.bwz;
        Eigen::Matrix<double, 9, 9> Info = pInt->C.block<9, 9>(0, 0).cast<double>().inverse();   // EdgeInertial ctor, G2oTypes.cc:510-518
        Info = (Info + Info.transpose()) / 2;
        Eigen::SelfAdjointEigenSolver<Eigen::Matrix<double, 9, 9> > es(Info);
        Eigen::Matrix<double, 9, 1> eigs = es.eigenvalues();
        for (int k = 0; k < 9; k++) if (eigs[k] < 1e-12) eigs[k] = 0;
        Info = es.eigenvectors() * eigs.asDiagonal() * es.eigenvectors().transpose();
        L.robust = (i == N - 1 || bRecInit);
        if (i == N - 1) Info *= 1e-2;                                               // :2651
        const Eigen::Matrix3d InfoG = pInt->C.block<3, 3>(9, 9).cast<double>().inverse(), InfoA = pInt->C.block<3, 3>(12, 12).cast<double>().inverse();
        for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) L.info9[9 * r + c] = Info(r, c);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { L.info_gyro[3 * r + c] = InfoG(r, c); L.info_acc[3 * r + c] = InfoA(r, c); }
        links.push_back(L);
    }
    std::vector<MapPoint*> mps(lLocalMapPoints.begin(), lLocalMapPoints.end());     // :2714-2840
    std::map<MapPoint*, int> mpIndex;
    std::vector<double> X(3 * mps.size());
    for (size_t i = 0; i < mps.size(); i++) {
        mpIndex[mps[i]] = (int)i;
        const Eigen::Vector3d Xd = mps[i]->GetWorldPos().cast<double>();
        X[3 * i] = Xd.x(); X[3 * i + 1] = Xd.y(); X[3 * i + 2] = Xd.z();
    }
    std::vector<int32_t> eKFi, ePt;
    std::vector<double> eObs, eW;
    std::vector<uint8_t> eStereo;
    std::vector<KeyFrame*> eKF;
    std::vector<MapPoint*> eMP;
    for (MapPoint* pMP : lLocalMapPoints)
        for (auto& obs : pMP->GetObservations()) {
            KeyFrame* pKFi = obs.first;
            if (pKFi->mnBALocalForKF != pKF->mnId && pKFi->mnBAFixedForKF != pKF->mnId) continue;
            if (pKFi->isBad() || pKFi->GetMap() != pCurrentMap || !kfIndex.count(pKFi)) continue;
            const int leftIndex = std::get<0>(obs.second);
            if (leftIndex == -1) continue;
            const cv::KeyPoint& kpUn = pKFi->mvKeysUn[leftIndex];
            const float ur = pKFi->mvuRight[leftIndex];
            Eigen::Matrix<double, 2, 1> o2; o2 << kpUn.pt.x, kpUn.pt.y;
            const float unc2 = pKFi->mpCamera->uncertainty2(o2);
            const float invSigma2 = pKFi->mvInvLevelSigma2[kpUn.octave] / unc2;
            eKFi.push_back(kfIndex.at(pKFi)); ePt.push_back(mpIndex.at(pMP));
            eObs.push_back(kpUn.pt.x); eObs.push_back(kpUn.pt.y); eObs.push_back(ur >= 0 ? (double)ur : -1.0);
            eW.push_back((double)invSigma2); eStereo.push_back(ur >= 0);
            eKF.push_back(pKFi); eMP.push_back(pMP);
        }
    num_edges = (int)eKFi.size();

    LibaProblem pr;
    std::memset(&pr, 0, sizeof(pr));
    pr.n_kf = nKF; pr.Rwb = Rwb.data(); pr.twb = twb.data(); pr.vel = vel.data(); pr.bg = bg.data(); pr.ba = ba.data();
    pr.pose_fixed = poseFixed.data(); pr.has_imu = hasImu.data(); pr.imu_fixed = imuFixed.data();
    const Eigen::Matrix3d Rcb = pKF->mImuCalib.mTcb.rotationMatrix().cast<double>();
    const Eigen::Vector3d tcb = pKF->mImuCalib.mTcb.translation().cast<double>(), tbc = pKF->mImuCalib.mTbc.translation().cast<double>();
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pr.Rcb[3 * r + c] = Rcb(r, c); pr.tcb[r] = tcb(r); pr.tbc[r] = tbc(r); }
    pr.fx = pKF->fx; pr.fy = pKF->fy; pr.cx = pKF->cx; pr.cy = pKF->cy; pr.bf = pKF->mbf;
    pr.n_points = (int)mps.size(); pr.points = X.data();
    pr.n_edges = num_edges; pr.edge_kf = eKFi.data(); pr.edge_point = ePt.data(); pr.edge_obs = eObs.data(); pr.edge_inv_sigma2 = eW.data(); pr.edge_stereo = eStereo.data();
    pr.n_links = (int)links.size(); pr.links = links.data();
    const float thHuberMono = sqrt(5.991), thHuberStereo = sqrt(7.815);             // :2694-2697 (through float)
    pr.huber_mono = thHuberMono; pr.huber_stereo = thHuberStereo; pr.huber_inertial = sqrt(16.92);
    pr.lambda_init = bLarge ? 1e-2 : 1e0; pr.max_iters = opt_it;
    static thread_local liba_solver* solver = nullptr;
    if (!solver) orbslam3_hip::check(liba_create(0, &solver));
    std::vector<double> Ro(Rwb.size()), to(twb.size()), vo(vel.size()), go(bg.size()), ao(ba.size()), Xo(X.size()), chi2(num_edges);
    std::vector<uint8_t> depthPos(num_edges);
    LbaStats st;
    orbslam3_hip::check(liba_solve(solver, &pr, Ro.data(), to.data(), vo.data(), go.data(), ao.data(), Xo.data(), chi2.data(), depthPos.data(), &st));
    const float err = (float)st.chi2_initial, err_end = (float)st.chi2_final;       // :2843-2845

    const float chi2Mono2 = 5.991, chi2Stereo2 = 7.815;                             // :2849-2884
    std::vector<std::pair<KeyFrame*, MapPoint*> > vToErase;
    for (int e = 0; e < num_edges; e++) {
        if (eMP[e]->isBad()) continue;
        if (eStereo[e]) { if (chi2[e] > chi2Stereo2) vToErase.push_back(std::make_pair(eKF[e], eMP[e])); continue; }
        const bool bClose = eMP[e]->mTrackDepth < 10.f;
        if ((chi2[e] > chi2Mono2 && !bClose) || (chi2[e] > 1.5f * chi2Mono2 && bClose) || !depthPos[e]) vToErase.push_back(std::make_pair(eKF[e], eMP[e]));
    }
    std::unique_lock<std::mutex> lock(pMap->mMutexMapUpdate);                       // :2887
    if ((2 * err < err_end || std::isnan(err) || std::isnan(err_end)) && !bLarge) return;        // :2891-2895
    for (auto& er : vToErase) { er.first->EraseMapPointMatch(er.second); er.second->EraseObservation(er.first); }
    for (KeyFrame* k : lFixedKeyFrames) k->mnBAFixedForKF = 0;
    for (int i = 0; i < N; i++) {                                                   // :2913-2934
        KeyFrame* pKFi = vpOptimizableKFs[i];
        Eigen::Matrix3d R; Eigen::Vector3d t;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R(r, c) = Ro[9 * i + 3 * r + c]; t(r) = to[3 * i + r]; }
        const Eigen::Matrix3d Rcw = Rcb * R.transpose();                            // ImuCamPose: Rcw = Rcb Rbw, tcw = Rcb tbw + tcb
        const Eigen::Vector3d tcw = Rcb * (-R.transpose() * t) + tcb;
        pKFi->SetPose(Sophus::SE3f(Rcw.cast<float>(), tcw.cast<float>()));
        pKFi->mnBALocalForKF = 0;
        if (pKFi->bImu) {
            pKFi->SetVelocity(Eigen::Vector3d(vo[3 * i], vo[3 * i + 1], vo[3 * i + 2]).cast<float>());
            pKFi->SetNewBias(IMU::Bias(ao[3 * i], ao[3 * i + 1], ao[3 * i + 2], go[3 * i], go[3 * i + 1], go[3 * i + 2]));
        }
    }
    for (MapPoint* pMP : lLocalMapPoints) {                                         // :2947-2954
        const int i = mpIndex[pMP];
        pMP->SetWorldPos(Eigen::Vector3d(Xo[3 * i], Xo[3 * i + 1], Xo[3 * i + 2]).cast<float>());
        pMP->UpdateNormalAndDepth();
    }
    pMap->IncreaseChangeIndex();
}

// int Optimizer::PoseInertialOptimizationLastKeyFrame(Frame*, bool bRecInit) (src/Optimizer.cc:4491-4873) and
// int Optimizer::PoseInertialOptimizationLastFrame(Frame*, bool bRecInit) (:4875-5285), conventional cameras (Nleft == -1): one
// LibaPoseProblem edge per feature holding a map point, in feature order.  The last-frame variant optimises the previous frame too,
// ties it to pFp->mpcpi and marginalises it afterwards with the reference's own Optimizer::Marginalize.
inline int PoseInertialOptimizationHIP(Frame* pFrame, bool bRecInit, bool lastFrame)
{
    if (pFrame->Nleft != -1)                                                        // stereo-fisheye rig: not on this path
        return lastFrame ? Optimizer::PoseInertialOptimizationLastFrame(pFrame, bRecInit) : Optimizer::PoseInertialOptimizationLastKeyFrame(pFrame, bRecInit);
    const int N = pFrame->N;
    std::vector<int> feat;
    std::vector<double> Xw, obs, w;
    std::vector<uint8_t> stereo, closePt;
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);                   // :4545
        for (int i = 0; i < N; i++) {
            MapPoint* pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
            const float ur = pFrame->mvuRight[i];
            Eigen::Matrix<double, 2, 1> o2; o2 << kpUn.pt.x, kpUn.pt.y;
            const float unc2 = pFrame->mpCamera->uncertainty2(o2);
            const float invSigma2 = pFrame->mvInvLevelSigma2[kpUn.octave] / unc2;
            const Eigen::Vector3d X = pMP->GetWorldPos().cast<double>();
            pFrame->mvbOutlier[i] = false;
            feat.push_back(i);
            Xw.push_back(X.x()); Xw.push_back(X.y()); Xw.push_back(X.z());
            obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(ur >= 0 ? (double)ur : -1.0);
            w.push_back((double)invSigma2); stereo.push_back(ur >= 0); closePt.push_back(pMP->mTrackDepth < 10.f);
        }
    }
    KeyFrame* pKF = pFrame->mpLastKeyFrame;
    Frame* pFp = pFrame->mpPrevFrame;
    IMU::Preintegrated* pInt = lastFrame ? pFrame->mpImuPreintegratedFrame : pFrame->mpImuPreintegrated;       // :5060 / :4673
    LibaPoseProblem pr;
    std::memset(&pr, 0, sizeof(pr));
    auto putState = [&](int i, const Eigen::Matrix3f& R, const Eigen::Vector3f& t, const Eigen::Vector3f& v, const IMU::Bias& b) {
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pr.Rwb[9 * i + 3 * r + c] = R(r, c); pr.twb[3 * i + r] = t(r); pr.vel[3 * i + r] = v(r); }
        pr.bg[3 * i] = b.bwx; pr.bg[3 * i + 1] = b.bwy; pr.bg[3 * i + 2] = b.bwz; pr.ba[3 * i] = b.bax; pr.ba[3 * i + 1] = b.bay; pr.ba[3 * i + 2] = b.baz;
    };
    if (lastFrame) putState(0, pFp->GetImuRotation(), pFp->GetImuPosition(), pFp->GetVelocity(), pFp->mImuBias);
    else putState(0, pKF->GetImuRotation(), pKF->GetImuPosition(), pKF->GetVelocity(), pKF->GetImuBias());
    putState(1, pFrame->GetImuRotation(), pFrame->GetImuPosition(), pFrame->GetVelocity(), pFrame->mImuBias);
    const Eigen::Matrix3d Rcb = pFrame->mImuCalib.mTcb.rotationMatrix().cast<double>();
    const Eigen::Vector3d tcb = pFrame->mImuCalib.mTcb.translation().cast<double>(), tbc = pFrame->mImuCalib.mTbc.translation().cast<double>();
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) pr.Rcb[3 * r + c] = Rcb(r, c); pr.tcb[r] = tcb(r); pr.tbc[r] = tbc(r); }
    pr.fx = pFrame->fx; pr.fy = pFrame->fy; pr.cx = pFrame->cx; pr.cy = pFrame->cy; pr.bf = pFrame->mbf;
    pr.n = (int)feat.size(); pr.Xw = Xw.data(); pr.obs = obs.data(); pr.inv_sigma2 = w.data(); pr.stereo = stereo.data(); pr.close_point = closePt.data();
    LibaLink& L = pr.link;
    L.kf1 = 0; L.kf2 = 1;
    auto put3x3 = [](float* dst, const Eigen::Matrix3f& M) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) dst[3 * r + c] = M(r, c); };
    put3x3(L.dR, pInt->dR); put3x3(L.JRg, pInt->JRg); put3x3(L.JVg, pInt->JVg); put3x3(L.JVa, pInt->JVa); put3x3(L.JPg, pInt->JPg); put3x3(L.JPa, pInt->JPa);
    for (int r = 0; r < 3; r++) { L.dV[r] = pInt->dV(r); L.dP[r] = pInt->dP(r); }
    L.dT = pInt->dT;
    const IMU::Bias b0 = pInt->GetOriginalBias();
    L.bias0[0] = b0.bax; L.bias0[1] = b0.bay; L.bias0[2] = b0.baz; L.bias0[3] = b0.bwx; L.bias0[4] = b0.bwy; L.bias0[5] = b0.bwz;
    Eigen::Matrix<double, 9, 9> Info = pInt->C.block<9, 9>(0, 0).cast<double>().inverse();       // EdgeInertial ctor, G2oTypes.cc:510-518
    Info = (Info + Info.transpose()) / 2;
    Eigen::SelfAdjointEigenSolver<Eigen::Matrix<double, 9, 9> > es(Info);
    Eigen::Matrix<double, 9, 1> eigs = es.eigenvalues();
    for (int k = 0; k < 9; k++) if (eigs[k] < 1e-12) eigs[k] = 0;
    Info = es.eigenvectors() * eigs.asDiagonal() * es.eigenvectors().transpose();
    // both variants take the random-walk informations from pFrame->mpImuPreintegrated (:4686, :5069)
    const Eigen::Matrix3d InfoG = pFrame->mpImuPreintegrated->C.block<3, 3>(9, 9).cast<double>().inverse();
    const Eigen::Matrix3d InfoA = pFrame->mpImuPreintegrated->C.block<3, 3>(12, 12).cast<double>().inverse();
    for (int r = 0; r < 9; r++) for (int c = 0; c < 9; c++) L.info9[9 * r + c] = Info(r, c);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { L.info_gyro[3 * r + c] = InfoG(r, c); L.info_acc[3 * r + c] = InfoA(r, c); }
    const float thHuberMono = sqrt(5.991), thHuberStereo = sqrt(7.815);
    pr.huber_mono = thHuberMono; pr.huber_stereo = thHuberStereo; pr.rec_init = bRecInit;
    if (lastFrame) {                                                                // EdgePriorPoseImu(pFp->mpcpi) (:5079-5090)
        const ConstraintPoseImu* c = pFp->mpcpi;
        pr.last_frame = 1;
        for (int r = 0; r < 3; r++) {
            for (int q = 0; q < 3; q++) pr.prior_Rwb[3 * r + q] = c->Rwb(r, q);
            pr.prior_twb[r] = c->twb(r); pr.prior_vel[r] = c->vwb(r); pr.prior_bg[r] = c->bg(r); pr.prior_ba[r] = c->ba(r);
        }
        for (int r = 0; r < 15; r++) for (int q = 0; q < 15; q++) pr.prior_H[15 * r + q] = c->H(r, q);
    }
    static thread_local liba_solver* solver = nullptr;
    if (!solver) orbslam3_hip::check(liba_create(0, &solver));
    double R[9], t[3], v[3], bg[3], ba[3], H[900];
    std::vector<uint8_t> outlier(feat.size() + 1);
    int32_t inliers = 0, nBad = 0;
    orbslam3_hip::check(liba_pose_optimize_batch(solver, &pr, 1, R, t, v, bg, ba, outlier.data(), H, &inliers, &nBad));
    for (size_t k = 0; k < feat.size(); k++) pFrame->mvbOutlier[feat[k]] = outlier[k] != 0;
    Eigen::Matrix3d Rwb; Eigen::Vector3d twb(t[0], t[1], t[2]), vwb(v[0], v[1], v[2]), vbg(bg[0], bg[1], bg[2]), vba(ba[0], ba[1], ba[2]);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rwb(r, c) = R[3 * r + c];
    pFrame->SetImuPoseVelocity(Rwb.cast<float>(), twb.cast<float>(), vwb.cast<float>());         // :4831-4834
    pFrame->mImuBias = IMU::Bias(ba[0], ba[1], ba[2], bg[0], bg[1], bg[2]);
    Eigen::Matrix<double, 15, 15> Hm;
    if (lastFrame) {                                                                // :5282-5287
        Eigen::MatrixXd H30(30, 30);
        for (int r = 0; r < 30; r++) for (int c = 0; c < 30; c++) H30(r, c) = H[30 * r + c];
        H30 = Optimizer::Marginalize(H30, 0, 14);
        Hm = H30.block<15, 15>(15, 15);
    } else
        for (int r = 0; r < 15; r++) for (int c = 0; c < 15; c++) Hm(r, c) = H[15 * r + c];
    pFrame->mpcpi = new ConstraintPoseImu(Rwb, twb, vwb, vbg, vba, Hm);                           // :4870 / :5284
    if (lastFrame) { delete pFp->mpcpi; pFp->mpcpi = NULL; }
    return inliers;
}
inline int PoseInertialOptimizationLastKeyFrameHIP(Frame* pFrame, bool bRecInit = false) { return PoseInertialOptimizationHIP(pFrame, bRecInit, false); }
inline int PoseInertialOptimizationLastFrameHIP(Frame* pFrame, bool bRecInit = false) { return PoseInertialOptimizationHIP(pFrame, bRecInit, true); }

// void Frame::ComputeBoW() (src/Frame.cc:825-832): mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) on the device.
// The tree is flattened once per vocabulary (TemplatedVocabulary::m_nodes is protected: reached through a derived type).
class VocabularyHIP {
public:
    explicit VocabularyHIP(const ORBVocabulary& voc)
    {
        struct Access : ORBVocabulary { using ORBVocabulary::m_nodes; using ORBVocabulary::m_L; };
        const Access& a = static_cast<const Access&>(voc);
        const int n = (int)a.m_nodes.size();
        std::vector<int32_t> off(n + 1, 0), word(n);
        std::vector<uint32_t> child;
        std::vector<uint8_t> desc((size_t)n * 32, 0);
        std::vector<double> weight(n);
        for (int i = 0; i < n; i++) {
            for (DBoW2::NodeId c : a.m_nodes[i].children) child.push_back(c);
            off[i + 1] = (int32_t)child.size();
            if (!a.m_nodes[i].descriptor.empty()) std::memcpy(&desc[(size_t)i * 32], a.m_nodes[i].descriptor.data, 32);
            weight[i] = a.m_nodes[i].weight;
            word[i] = a.m_nodes[i].isLeaf() ? (int32_t)a.m_nodes[i].word_id : -1;
        }
        OrbvVocabulary v;
        v.n_nodes = n; v.L = a.m_L; v.child_off = off.data(); v.child_id = child.data(); v.desc = desc.data();
        v.weight = weight.data(); v.word_id = word.data();
        orbslam3_hip::check(orbv_create(0, &v, &h_));
    }
    ~VocabularyHIP() { orbv_destroy(h_); }

    void ComputeBoW(Frame& F) const
    {
        if (!F.mBowVec.empty()) return;                                             // :827
        const int n = F.mDescriptors.rows;
        std::vector<uint32_t> bi(n + 1), fn(n + 1), ff(n + 1);
        std::vector<double> bv(n + 1);
        std::vector<int32_t> fo(n + 2);
        int32_t nb = 0, nf = 0;
        orbslam3_hip::check(orbv_transform(h_, F.mDescriptors.data, n, 4, bi.data(), bv.data(), &nb, fn.data(), fo.data(), ff.data(), &nf));
        DBoW2::BowVector::iterator bit = F.mBowVec.end();
        for (int k = 0; k < nb; k++) bit = F.mBowVec.insert(F.mBowVec.end(), std::make_pair(bi[k], bv[k]));      // ascending ids: O(1) hinted inserts
        for (int k = 0; k < nf; k++)
            F.mFeatVec.insert(F.mFeatVec.end(), std::make_pair(fn[k], std::vector<unsigned int>(ff.begin() + fo[k], ff.begin() + fo[k + 1])));
        (void)bit;
    }

private:
    orbv_vocab* h_ = nullptr;
};

// int Optimizer::PoseOptimization(Frame* pFrame) (src/Optimizer.cc:814-1115), conventional (non-rigid-body) cameras:
// one PoseProblem edge per feature holding a MapPoint, in feature order (= g2o's addEdge order).
inline int PoseOptimizationHIP(Frame* pFrame)
{
    if (pFrame->mpCamera2) return Optimizer::PoseOptimization(pFrame);              // rigid-body stereo-fisheye rig: not on this path
    const int N = pFrame->N;
    std::vector<double> Xw, obs, w;
    std::vector<uint8_t> stereo;
    std::vector<int> feat;
    {
        std::unique_lock<std::mutex> lock(MapPoint::mGlobalMutex);                  // :857
        for (int i = 0; i < N; i++) {
            MapPoint* pMP = pFrame->mvpMapPoints[i];
            if (!pMP) continue;
            pFrame->mvbOutlier[i] = false;                                          // :870, :898
            const cv::KeyPoint& kpUn = pFrame->mvKeysUn[i];
            const float ur = pFrame->mvuRight[i];
            const Eigen::Vector3d X = pMP->GetWorldPos().cast<double>();
            Xw.push_back(X.x()); Xw.push_back(X.y()); Xw.push_back(X.z());
            obs.push_back(kpUn.pt.x); obs.push_back(kpUn.pt.y); obs.push_back(ur < 0 ? -1.0 : (double)ur);
            w.push_back((double)pFrame->mvInvLevelSigma2[kpUn.octave]);
            stereo.push_back(ur < 0 ? 0 : 1);
            feat.push_back(i);
        }
    }
    PoseProblem pr;
    const Sophus::SE3<float> Tcw = pFrame->GetPose();
    const Eigen::Quaterniond qd = Tcw.unit_quaternion().cast<double>();
    const Eigen::Vector3d td = Tcw.translation().cast<double>();
    pr.q[0] = qd.x(); pr.q[1] = qd.y(); pr.q[2] = qd.z(); pr.q[3] = qd.w();
    pr.t[0] = td.x(); pr.t[1] = td.y(); pr.t[2] = td.z();
    pr.n = (int)feat.size(); pr.Xw = Xw.data(); pr.obs = obs.data(); pr.inv_sigma2 = w.data(); pr.stereo = stereo.data();
    pr.fx = pFrame->fx; pr.fy = pFrame->fy; pr.cx = pFrame->cx; pr.cy = pFrame->cy; pr.bf = pFrame->mbf;
    const float deltaMono = sqrt(5.991), deltaStereo = sqrt(7.815);                 // :838-839 (through float)
    pr.huber_mono = deltaMono; pr.huber_stereo = deltaStereo;
    static thread_local pose_solver* solver = nullptr;
    if (!solver) orbslam3_hip::check(pose_create(0, &solver));
    PoseResult res;
    std::vector<uint8_t> outlier(feat.size() + 1);
    orbslam3_hip::check(pose_optimize(solver, &pr, &res, outlier.data()));
    if (pr.n < 3) return 0;                                                         // :998-999 (pose untouched)
    for (size_t k = 0; k < feat.size(); k++) pFrame->mvbOutlier[feat[k]] = outlier[k] != 0;
    const Eigen::Quaterniond qo(res.q[3], res.q[0], res.q[1], res.q[2]);
    pFrame->SetPose(Sophus::SE3f(qo.cast<float>(), Eigen::Vector3d(res.t[0], res.t[1], res.t[2]).cast<float>()));   // :1107-1110
    return res.inliers;
}

}  // namespace ORB_SLAM3

// class SlamPktVI (include/Socket/slampkt_vi.h:14-212, global namespace there as well): same two constructors and
// getters, the byte shuffling done by the device codec.  Needs "Socket/imudata.h" (IMUData).
#include "Socket/imudata.h"
class SlamPktVIHIP {
public:
    SlamPktVIHIP(unsigned char* buffer, int packet_size) : total_len_(packet_size)                                   // :85
    {
        std::vector<OrbxKeyPoint> k; std::vector<uint8_t> d; std::vector<OrbeImuSample> im;
        int32_t id = 0; int64_t ts = 0;
        codec().unpack(buffer, packet_size, id, ts, k, d, im);
        frame_id_ = id; time_stamp_ = (long)ts;
        payload_.assign(buffer, buffer + packet_size);
        kps_.reserve(k.size());
        for (const OrbxKeyPoint& p : k) kps_.push_back(cv::KeyPoint(p.x, p.y, p.size, p.angle, p.response, p.octave, p.class_id));
        descriptors_ = cv::Mat((int)k.size(), 32, CV_8UC1);
        if (!k.empty()) std::memcpy(descriptors_.data, d.data(), d.size());
        for (const OrbeImuSample& s : im) {
            std::vector<float> g(s.gyro, s.gyro + 3), a(s.acce, s.acce + 3);
            imus_.push_back(IMUData((long)s.ts, g, a));
        }
    }
    SlamPktVIHIP(int id, long timestamp, std::vector<cv::KeyPoint>& kps, cv::Mat& descriptors, std::vector<IMUData>& imus)   // :127
        : frame_id_(id), time_stamp_(timestamp), kps_(kps), descriptors_(descriptors), imus_(imus)
    {
        std::vector<OrbxKeyPoint> k(kps.size());
        for (size_t i = 0; i < kps.size(); i++) { k[i] = OrbxKeyPoint(); k[i].x = kps[i].pt.x; k[i].y = kps[i].pt.y; }
        std::vector<OrbeImuSample> im(imus.size());
        for (size_t i = 0; i < imus.size(); i++) {
            im[i].ts = imus[i].ts_;
            for (int j = 0; j < 3; j++) { im[i].gyro[j] = imus[i].gyro_[j]; im[i].acce[j] = imus[i].acce_[j]; }
        }
        cv::Mat d = descriptors.isContinuous() ? descriptors : descriptors.clone();
        payload_ = codec().pack(id, timestamp, k.data(), d.data, (int)k.size(), im.data(), (int)im.size(), head_);
        total_len_ = (int)payload_.size();
    }
    std::vector<cv::KeyPoint> getKeyPoints() { return kps_; }
    cv::Mat getDescriptors() { return descriptors_; }
    std::vector<IMUData> getIMUData() { return imus_; }
    unsigned char* getPayload() { return payload_.data(); }
    unsigned char* getHead() { if (total_len_ > 65536) return nullptr; unsigned char* h = new unsigned char[2]; h[0] = head_[0]; h[1] = head_[1]; return h; }   // :185
    int getTotalLength() { return total_len_; }
    int getNumPoints() { return (int)kps_.size(); }
    int getNumIMUs() { return (int)imus_.size(); }
    int getFrameId() { return frame_id_; }
    long getTimeStamp() { return time_stamp_; }

private:
    static orbslam3_hip::EdgePacketCodec& codec() { static thread_local orbslam3_hip::EdgePacketCodec c; return c; }
    int total_len_ = 0, frame_id_ = 0;
    long time_stamp_ = 0;
    std::vector<cv::KeyPoint> kps_;
    cv::Mat descriptors_;

public:
    std::vector<IMUData> imus_;                                                     // public in the reference (src/Socket/client.cc:138)

private:
    std::vector<unsigned char> payload_;
    unsigned char head_[2] = {0, 0};
};

#endif  // ORBSLAM3_HIP_WITH_REFERENCE
