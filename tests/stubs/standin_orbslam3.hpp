// TEST INFRASTRUCTURE -- stand-ins for the ORB_SLAM3 classes that include/orbslam3_shim.hpp is written against, holding
// ONLY the members the shim touches (names and types as in the reference headers cited per class), with the few lines of
// behaviour a toy map needs.  They exist so that the reference-typed half of the shim is seen by a compiler and its glue
// (graph walk, flattening order, write-back, argument marshalling) can be run in an image without OpenCV / Eigen / Sophus.
// This is a compile/behaviour check of glue code -- not an oracle, not a build of the reference.
#pragma once

#include <list>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <tuple>
#include <vector>

#include <opencv2/core/core.hpp>

#include "standin_eigen.hpp"
#include "standin_sophus.hpp"

#define FRAME_GRID_ROWS 48      /* include/Frame.h:44-45 */
#define FRAME_GRID_COLS 64

namespace DBoW2 {
typedef unsigned int NodeId;
typedef unsigned int WordId;
typedef double WordValue;
class BowVector : public std::map<WordId, WordValue> {};                      // Thirdparty/DBoW2/DBoW2/BowVector.h
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > {}; // Thirdparty/DBoW2/DBoW2/FeatureVector.h
}  // namespace DBoW2

namespace ORB_SLAM3 {

class KeyFrame;
class MapPoint;
class Map;
class Frame;

// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:296-330, :411-423 (what VocabularyHIP flattens)
class ORBVocabulary {
public:
    struct Node {
        DBoW2::NodeId id = 0; DBoW2::WordValue weight = 0; std::vector<DBoW2::NodeId> children; DBoW2::NodeId parent = 0;
        cv::Mat descriptor; DBoW2::WordId word_id = 0;
        bool isLeaf() const { return children.empty(); }
    };
    int m_L = 0;
    std::vector<Node> m_nodes;
};

// include/CameraModels/GeometricCamera.h:61-95
class GeometricCamera {
public:
    virtual ~GeometricCamera() {}
    virtual Eigen::Vector2f project(const Eigen::Vector3f& v3D) = 0;
    virtual float uncertainty2(const Eigen::Matrix<double, 2, 1>& p2D) = 0;
    virtual Eigen::Matrix3f toK_() = 0;
    unsigned int GetType() { return mnType; }
    const static unsigned int CAM_PINHOLE = 0;
    const static unsigned int CAM_FISHEYE = 1;

protected:
    unsigned int mnType = CAM_PINHOLE;
};

// src/CameraModels/Pinhole.cpp:35-41, :99-103
class Pinhole : public GeometricCamera {
public:
    Pinhole(float fx, float fy, float cx, float cy) : fx_(fx), fy_(fy), cx_(cx), cy_(cy) { mnType = CAM_PINHOLE; }
    Eigen::Vector2f project(const Eigen::Vector3f& v) override { return Eigen::Vector2f(fx_ * v[0] / v[2] + cx_, fy_ * v[1] / v[2] + cy_); }
    float uncertainty2(const Eigen::Matrix<double, 2, 1>&) override { return 1.0f; }
    Eigen::Matrix3f toK_() override { Eigen::Matrix3f K; K(0, 0) = fx_; K(0, 2) = cx_; K(1, 1) = fy_; K(1, 2) = cy_; K(2, 2) = 1.f; return K; }

private:
    float fx_, fy_, cx_, cy_;
};

namespace IMU {
// include/ImuTypes.h:62-90
class Bias {
public:
    Bias() : bax(0), bay(0), baz(0), bwx(0), bwy(0), bwz(0) {}
    Bias(const float& ax, const float& ay, const float& az, const float& wx, const float& wy, const float& wz) : bax(ax), bay(ay), baz(az), bwx(wx), bwy(wy), bwz(wz) {}
    float bax, bay, baz, bwx, bwy, bwz;
};
// include/ImuTypes.h:92-125
class Calib {
public:
    Sophus::SE3<float> mTcb, mTbc;
};
// include/ImuTypes.h:140-230
class Preintegrated {
public:
    void SetNewBias(const Bias& b) { bu = b; }
    Bias GetOriginalBias() { return b; }
    float dT = 0;
    Eigen::Matrix<float, 15, 15> C;
    Eigen::Matrix3f dR, JRg, JVg, JVa, JPg, JPa;
    Eigen::Vector3f dV, dP;
    Bias b, bu;
};
}  // namespace IMU

// include/G2oTypes.h:706-730
class ConstraintPoseImu {
public:
    ConstraintPoseImu(const Eigen::Matrix3d& Rwb_, const Eigen::Vector3d& twb_, const Eigen::Vector3d& vwb_, const Eigen::Vector3d& bg_,
                      const Eigen::Vector3d& ba_, const Eigen::Matrix<double, 15, 15>& H_) : Rwb(Rwb_), twb(twb_), vwb(vwb_), bg(bg_), ba(ba_), H(H_) {}
    Eigen::Matrix3d Rwb;
    Eigen::Vector3d twb, vwb, bg, ba;
    Eigen::Matrix<double, 15, 15> H;
};

// include/Map.h (GetInitKFid, IsInertial, IncreaseChangeIndex, KeyFramesInMap, mMutexMapUpdate)
class Map {
public:
    long unsigned int GetInitKFid() { return mnInitKFid; }
    bool IsInertial() { return mbIsInertial; }
    void IncreaseChangeIndex() { mnMapChange++; }
    long unsigned int KeyFramesInMap() { return nKeyFrames; }
    KeyFrame* GetOriginKF() { return mpOriginKF; }
    std::vector<KeyFrame*> GetAllKeyFrames() { return mvpAllKeyFrames; }
    std::vector<MapPoint*> GetAllMapPoints() { return mvpAllMapPoints; }
    KeyFrame* mpOriginKF = nullptr;
    std::vector<KeyFrame*> mvpAllKeyFrames;
    std::vector<MapPoint*> mvpAllMapPoints;
    std::mutex mMutexMapUpdate;
    long unsigned int mnInitKFid = 0, nKeyFrames = 0;
    bool mbIsInertial = false;
    int mnMapChange = 0;
};

// include/MapPoint.h (tracking fields :142-160, BA markers :184, getters :100-135)
class MapPoint {
public:
    Eigen::Vector3f GetWorldPos() { return mWorldPos; }
    void SetWorldPos(const Eigen::Vector3f& p) { mWorldPos = p; }
    Eigen::Vector3f GetNormal() { return mNormal; }
    std::map<KeyFrame*, std::tuple<int, int> > GetObservations() { return mObservations; }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    Map* GetMap() { return mpMap; }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float&, KeyFrame*) { return mnPredicted; }
    int PredictScale(const float&, Frame*) { return mnPredicted; }
    bool IsInKeyFrame(KeyFrame* k) { return mObservations.count(k) != 0; }
    std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* k) { return mObservations.count(k) ? mObservations[k] : std::tuple<int, int>(-1, -1); }
    void AddObservation(KeyFrame* k, int idx) { mObservations[k] = std::tuple<int, int>(idx, -1); nObs++; }
    void EraseObservation(KeyFrame* k) { if (mObservations.erase(k)) nObs--; nErased++; }
    void Replace(MapPoint* p) { mpReplaced = p; mbBad = true; }
    void UpdateNormalAndDepth() { nNormalUpdates++; }

    long unsigned int mnId = 0, mnBALocalForKF = 0, mnBAGlobalForKF = 0;
    Eigen::Vector3f mPosGBA;
    float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackProjXR = 0, mTrackViewCos = 0;
    int mnTrackScaleLevel = 0;
    bool mbTrackInView = false, mbTrackInViewR = false;
    static std::mutex mGlobalMutex;

    // toy-map state (not reference members)
    Eigen::Vector3f mWorldPos, mNormal;
    std::map<KeyFrame*, std::tuple<int, int> > mObservations;
    int nObs = 0, nErased = 0, nNormalUpdates = 0, mnPredicted = 0;
    bool mbBad = false;
    Map* mpMap = nullptr;
    cv::Mat mDescriptor;
    float mfMinDistance = 0.f, mfMaxDistance = 1e9f;
    MapPoint* mpReplaced = nullptr;
};

// include/Frame.h (public data :188-330, pose accessors :108-120, IMU :100-130)
class Frame {
public:
    Sophus::SE3<float> GetPose() const { return mTcw; }
    void SetPose(const Sophus::SE3<float>& T) { mTcw = T; }
    Eigen::Matrix3f GetImuRotation() { return mRwb; }
    Eigen::Vector3f GetImuPosition() { return mtwb; }
    Eigen::Vector3f GetVelocity() const { return mVw; }
    void SetImuPoseVelocity(const Eigen::Matrix3f& R, const Eigen::Vector3f& t, const Eigen::Vector3f& v) { mRwb = R; mtwb = t; mVw = v; }

    int N = 0, Nleft = -1;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<float> mvuRight, mvDepth;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    std::vector<float> mvScaleFactors, mvInvLevelSigma2;
    float mb = 0, mbf = 0, fx = 0, fy = 0, cx = 0, cy = 0;
    GeometricCamera* mpCamera = nullptr;
    GeometricCamera* mpCamera2 = nullptr;
    DBoW2::BowVector mBowVec;
    DBoW2::FeatureVector mFeatVec;
    KeyFrame* mpLastKeyFrame = nullptr;
    Frame* mpPrevFrame = nullptr;
    IMU::Preintegrated* mpImuPreintegrated = nullptr;
    IMU::Preintegrated* mpImuPreintegratedFrame = nullptr;
    IMU::Bias mImuBias;
    IMU::Calib mImuCalib;
    ConstraintPoseImu* mpcpi = nullptr;

    Sophus::SE3<float> mTcw;
    Eigen::Matrix3f mRwb;
    Eigen::Vector3f mtwb, mVw;
};

// include/KeyFrame.h (ids / BA markers :312-330, grid :318-322, data :470-520, accessors :205-300)
class KeyFrame {
public:
    Sophus::SE3f GetPose() { return mTcw; }
    Sophus::SE3f GetPoseInverse() { return mTcw.inverse(); }
    void SetPose(const Sophus::SE3f& T) { mTcw = T; nPoseWrites++; }
    Eigen::Vector3f GetCameraCenter() { return mTcw.inverse().translation(); }
    Eigen::Matrix3f GetImuRotation() { return mRwb; }
    Eigen::Vector3f GetImuPosition() { return mtwb; }
    Eigen::Vector3f GetVelocity() { return mVw; }
    Eigen::Vector3f GetGyroBias() { return Eigen::Vector3f(mImuBias.bwx, mImuBias.bwy, mImuBias.bwz); }
    Eigen::Vector3f GetAccBias() { return Eigen::Vector3f(mImuBias.bax, mImuBias.bay, mImuBias.baz); }
    IMU::Bias GetImuBias() { return mImuBias; }
    void SetVelocity(const Eigen::Vector3f& v) { mVw = v; }
    void SetNewBias(const IMU::Bias& b) { mImuBias = b; }
    std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return mvpOrderedConnectedKeyFrames; }
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint* p, const size_t& idx) { mvpMapPoints[idx] = p; }
    void EraseMapPointMatch(MapPoint* p) { for (auto& q : mvpMapPoints) if (q == p) q = nullptr; }
    bool isBad() { return mbBad; }
    Map* GetMap() { return mpMap; }
    bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }

    long unsigned int mnId = 0, mnBALocalForKF = 0, mnBAFixedForKF = 0, mnBAGlobalForKF = 0;
    Sophus::SE3f mTcwGBA;
    int N = 0, NLeft = -1, mnGridCols = FRAME_GRID_COLS, mnGridRows = FRAME_GRID_ROWS;
    float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    std::vector<cv::KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    cv::Mat mDescriptors;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    DBoW2::FeatureVector mFeatVec;
    GeometricCamera* mpCamera = nullptr;
    GeometricCamera* mpCamera2 = nullptr;
    KeyFrame* mPrevKF = nullptr;
    bool bImu = false;
    IMU::Preintegrated* mpImuPreintegrated = nullptr;
    IMU::Calib mImuCalib;

    // toy-map state (not reference members)
    Sophus::SE3f mTcw;
    Eigen::Matrix3f mRwb;
    Eigen::Vector3f mtwb, mVw;
    IMU::Bias mImuBias;
    std::vector<KeyFrame*> mvpOrderedConnectedKeyFrames;
    std::vector<MapPoint*> mvpMapPoints;
    bool mbBad = false;
    Map* mpMap = nullptr;
    int nPoseWrites = 0;
};

// The reference entry points the adapters fall back to for camera rigs outside the accelerated path (declared as in
// include/ORBmatcher.h:40-106 and include/Optimizer.h:46-100; defined by the test translation unit to throw).
class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3, const bool bFarPoints = false, const float thFarPoints = 50.0f);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0, const bool bRight = false);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse = false);
};

class Optimizer {
public:
    static void LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges);
    static void BundleAdjustment(const std::vector<KeyFrame*>& vpKF, const std::vector<MapPoint*>& vpMP, int nIterations = 5, bool* pbStopFlag = NULL,
                                 const unsigned long nLoopKF = 0, const bool bRobust = true);
    static void LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF, int& num_OptKF, int& num_MPs, int& num_edges,
                                bool bLarge = false, bool bRecInit = false);
    static int PoseOptimization(Frame* pFrame);
    static int PoseInertialOptimizationLastKeyFrame(Frame* pFrame, bool bRecInit = false);
    static int PoseInertialOptimizationLastFrame(Frame* pFrame, bool bRecInit = false);
    static Eigen::MatrixXd Marginalize(const Eigen::MatrixXd& H, const int& start, const int& end);
};

}  // namespace ORB_SLAM3
