/*
 * orbslam3_hip.h -- C ABI of the MI355X (gfx950) implementation of ORB-SLAM3's per-frame hot path.
 *
 * The reference has no FFI layer: the boundary is three C++ signatures inside libORB_SLAM3.so
 * (SURVEY.md 8(b)).  This header is what a binding for that path links against; every entry
 * point cites the reference interface it replaces.  POD only: no cv::Mat / Eigen / torch types.
 * include/orbslam3_shim.hpp adapts these to the reference's own C++ signatures.
 *
 * Conventions: int status returns (0 ok, <0 error); caller-allocated outputs with capacity + count;
 * handles are NOT thread-safe (same as one reference ORBextractor instance); no global state.
 * All kernels are hand-written HIP for gfx950; there is no CPU fallback -- if no HIP device is
 * usable every compute entry point returns ORBX_ERR_NO_DEVICE.
 */
#ifndef ORBSLAM3_HIP_H
#define ORBSLAM3_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_OK 0
#define ORBX_ERR_EMPTY (-1)        /* empty image: reference returns -1 (src/ORBextractor.cc:1090-1091) */
#define ORBX_ERR_CAPACITY (-2)     /* caller-provided capacity too small; *n holds the required count */
#define ORBX_ERR_ARG (-3)
#define ORBX_ERR_NO_DEVICE (-4)
#define ORBX_ERR_HIP (-5)
#define ORBX_ERR_INTERNAL (-6)     /* device-side guard tripped (scratch capacity); never a silent wrong result */

const char* orbx_last_error(void);         /* thread-local message of the last failing call */
int orbx_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Extractor -- replaces ORB_SLAM3::ORBextractor (include/ORBextractor.h:44-109)
 * ------------------------------------------------------------------------------------------------ */

/* Field layout of cv::KeyPoint (28 bytes), so std::vector<cv::KeyPoint> can be filled with memcpy. */
typedef struct OrbxKeyPoint {
    float x, y;        /* pt, level-0 coordinates (scaled as src/ORBextractor.cc:1149-1151) */
    float size;        /* int(31 * scale[level])          (:880) */
    float angle;       /* degrees [0,360), IC_Angle        (:76-103) */
    float response;    /* FAST corner score */
    int32_t octave;
    int32_t class_id;  /* -1 */
} OrbxKeyPoint;

typedef struct orbx_extractor orbx_extractor;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) (src/ORBextractor.cc:409-469). */
int orbx_create(int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast,
                int device, orbx_extractor** out);
void orbx_destroy(orbx_extractor* h);

/* ORBextractor::operator()(image, mask [ignored], keypoints, descriptors, vLappingArea) (src/ORBextractor.cc:1086-1168).
 * Host buffers in, host buffers out.  kps/desc have room for `cap` keypoints (orbx_max_keypoints() is always enough).
 * *n = number of keypoints, *mono_index = the reference's return value.  Returns ORBX_ERR_EMPTY for an empty image. */
int orbx_extract(orbx_extractor* h, const uint8_t* img, int width, int height, int stride,
                 int lap0, int lap1, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n, int* mono_index);

/* Throughput path: `batch` independent frames of identical geometry per call (one reference operator() each).
 * Host variant: imgs[b] are host pointers; outputs are [batch][cap] host arrays. */
int orbx_extract_batch(orbx_extractor* h, const uint8_t* const* imgs, int batch, int width, int height, int stride,
                       int lap0, int lap1, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n, int* mono_index);

/* Device-resident variant: d_imgs, d_kps, d_desc, d_n, d_mono are DEVICE pointers (frame b at d_imgs + b*frame_stride);
 * the call only enqueues work on `stream` (a hipStream_t, NULL = default stream) and returns.
 * d_status[batch] (device, int32) receives ORBX_OK / ORBX_ERR_CAPACITY / ORBX_ERR_INTERNAL per frame. */
int orbx_extract_batch_device(orbx_extractor* h, const uint8_t* d_imgs, int batch, int width, int height,
                              int row_stride, size_t frame_stride, int lap0, int lap1,
                              OrbxKeyPoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_n, int32_t* d_mono,
                              int32_t* d_status, void* stream);

/* Upper bound on keypoints per frame: per level max(mnFeaturesPerLevel + 3, 32).  N + 3 is the octree's overshoot;
 * its first pass, however, divides every root without looking at N (src/ORBextractor.cc:606-672), so a level with a tiny budget
 * still returns up to 4 key points per root (nIni = round(width / height) of the level's bordered area).  The bound covers 8 roots;
 * orbx_max_keypoints_for() gives the bound for one image size (very wide images with tiny budgets need it: otherwise the calls
 * report ORBX_ERR_CAPACITY with *n = the count needed). */
int orbx_max_keypoints(const orbx_extractor* h);
int orbx_max_keypoints_for(const orbx_extractor* h, int width, int height);

/* Getters (include/ORBextractor.h:61-81).  Arrays of nlevels floats; NULL pointers are skipped. */
int orbx_levels(const orbx_extractor* h);
float orbx_scale_factor(const orbx_extractor* h);
int orbx_scale_tables(const orbx_extractor* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);
int orbx_features_per_level(const orbx_extractor* h, int* n_per_level);

/* Replaces the public member mvImagePyramid (include/ORBextractor.h:83; read by Frame::ComputeStereoMatches,
 * src/Frame.cc:938,1028,1043).  Copies level `level` of frame `frame` of the LAST extract call to host memory.
 * border=0: the WxH level image; border=19: with the EDGE_THRESHOLD reflect-101 border of ComputePyramid (:1170-1195). */
int orbx_pyramid_level_size(const orbx_extractor* h, int level, int* width, int* height);
int orbx_pyramid_level(orbx_extractor* h, int frame, int level, int border, uint8_t* dst, int dst_stride);

/* void Frame::ComputeStereoMatches() (src/Frame.cc:931-1101): for every left key point the best right
 * key point in its row band (Hamming), an 11x11 SAD sliding window on the pyramid level of the left key point, parabola
 * sub-pixel fit, disparity gates and the median cut.  `left` / `right` are the two extractor handles of the rectified
 * stereo pair (mpORBextractorLeft / Right); their pyramids of the LAST extract call stand for mvImagePyramid.  kps are
 * mvKeys / mvKeysRight, mb = baseline in metres, mbf = baseline * fx.  Outputs mvuRight / mvDepth (-1 where unmatched).
 * Host buffers, frame `frame` of the last batch: */
int orbx_stereo_matches(orbx_extractor* left, orbx_extractor* right, int frame,
                        const OrbxKeyPoint* kps_l, const uint8_t* desc_l, int n_l,
                        const OrbxKeyPoint* kps_r, const uint8_t* desc_r, int n_r, float mb, float mbf, float* u_right, float* depth);
/* Device-resident batch: the outputs of orbx_extract_batch_device of both handles ([batch][cap] arrays, d_n per frame);
 * d_u_right / d_depth are [batch][cap] floats.  One wave per left key point; only enqueues on `stream`. */
int orbx_stereo_matches_device(orbx_extractor* left, orbx_extractor* right, int batch,
                               const OrbxKeyPoint* d_kps_l, const uint8_t* d_desc_l, const int32_t* d_n_l,
                               const OrbxKeyPoint* d_kps_r, const uint8_t* d_desc_r, const int32_t* d_n_r, int cap,
                               float mb, float mbf, float* d_u_right, float* d_depth, void* stream);

/* Per-stage device time of the LAST extract call, measured with HIP events recorded on the launch stream
 * (used by bench.py for the roofline figure).  Stages: 0 level-0 copy, 1 pyramid resize, 2 FAST cells, 3 octree,
 * 4 output index, 5 blur, 6 orientation + descriptors.  Times in milliseconds. */
int orbx_profile_enable(orbx_extractor* h, int on);
int orbx_profile_read(orbx_extractor* h, float* stage_ms, int n_stages);

/* Stage-wise introspection for parity tests (results of the LAST extract call, frame `frame`). */
int orbx_debug_blurred_level(orbx_extractor* h, int frame, int level, uint8_t* dst, int dst_stride);
/* FAST candidates of a level in vToDistributeKeys order (src/ORBextractor.cc:787-872): x,y relative to minBorder, response. */
int orbx_debug_candidates(orbx_extractor* h, int frame, int level, OrbxKeyPoint* out, int cap, int* n);
/* per-level keypoints after DistributeOctTree + orientation, level coordinates, list order (:874-895). */
int orbx_debug_level_keypoints(orbx_extractor* h, int frame, int level, OrbxKeyPoint* out, int cap, int* n);
/* Test hook: capacity of the per-wave corner lists of the FAST kernel (cap <= 0: default).  A small value sends ordinary images
 * through the kernel's overflow path (every pixel of a wave's rows goes through non-maximum suppression and emission), which
 * otherwise only pathological images reach; results must not change. */
int orbx_debug_set_fast_corner_cap(orbx_extractor* h, int cap);
/* Test hook: a spin kernel of `microseconds` in front of the fused resize of the upper pyramid levels (which runs on a side
 * stream for batches >= 64 frames), so that a missing cross-stream dependency on those levels shows up as wrong results. */
int orbx_debug_set_tail_delay(orbx_extractor* h, int microseconds);
/* Test hook: the schedule the LAST extract call used -- bits 0-1 octree instantiation (0 node pool in HBM, 1 keys + nodes in
 * LDS, 2 keys in the L2-resident scratch, 3 per level and frame: keys in LDS when the level's candidates fit), 4 two octree launches, 8 level-0 octree started early, 16 level 0 read in place,
 * 32 resize tail on the side stream, 64 the whole resize chain on the side stream beside FAST on level 0. */
int orbx_debug_last_schedule(orbx_extractor* h);

/* ------------------------------------------------------------------------------------------------
 * Matcher -- replaces ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:40-106)
 * ------------------------------------------------------------------------------------------------ */

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:2058-2074).  Pure host helper (32-byte operands). */
int orbm_hamming(const uint8_t a[32], const uint8_t b[32]);

/* DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened to CSR, node ids ascending. */
typedef struct OrbmFeatVec {
    int32_t n_nodes;
    const uint32_t* node_id;   /* [n_nodes] ascending */
    const int32_t* offset;     /* [n_nodes+1] */
    const uint32_t* feat;      /* [offset[n_nodes]] feature indices, insertion order inside a node */
} OrbmFeatVec;

typedef struct orbm_matcher orbm_matcher;
int orbm_create(int device, orbm_matcher** out);
void orbm_destroy(orbm_matcher* m);

/* int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:223-425),
 * mono branch.  valid_kf[i] = (vpMapPointsKF[i] && !isBad()).  match_f2kf[nF] receives the KF feature index whose
 * MapPoint the shim stores in vpMapPointMatches[f], or -1.  Returns the reference's return value (>=0) or <0 on error. */
int orbm_search_by_bow(orbm_matcher* m,
                       const uint8_t* desc_kf, int n_kf, const uint8_t* valid_kf, const float* angle_kf, const OrbmFeatVec* fv_kf,
                       const uint8_t* desc_f, int n_f, const float* angle_f, const OrbmFeatVec* fv_f,
                       float nnratio, int check_orientation, int32_t* match_f2kf);

/* Batched variant: P independent (KF, F) pairs per launch, everything host-side SoA of per-pair pointers. */
typedef struct OrbmBowPair {
    const uint8_t* desc_kf; int32_t n_kf; const uint8_t* valid_kf; const float* angle_kf; OrbmFeatVec fv_kf;
    const uint8_t* desc_f;  int32_t n_f;  const float* angle_f;  OrbmFeatVec fv_f;
    int32_t* match_f2kf;    /* out [n_f] */
    int32_t n_matches;      /* out */
} OrbmBowPair;
int orbm_search_by_bow_batch(orbm_matcher* m, OrbmBowPair* pairs, int n_pairs, float nnratio, int check_orientation);

/* Device-resident batched SearchByBoW: the plan uploads the pairs once; run() only launches the kernel on `stream`
 * (hipStream_t, NULL = default stream); fetch() copies match_f2kf / n_matches back into the pairs. */
typedef struct orbm_bow_plan orbm_bow_plan;
int orbm_bow_plan_create(orbm_matcher* m, const OrbmBowPair* pairs, int n_pairs, orbm_bow_plan** out);
int orbm_bow_plan_run(orbm_bow_plan* plan, float nnratio, int check_orientation, void* stream);
int orbm_bow_plan_fetch(orbm_bow_plan* plan, OrbmBowPair* pairs, void* stream);
void orbm_bow_plan_destroy(orbm_bow_plan* plan);

/* Fully device-resident pairs: descriptors / key points as orbx_extract_batch_device wrote them, feature vectors as
 * orbv_transform_batch_device wrote them, counts read from device memory at run time -- extract -> transform -> SearchByBoW
 * without a host round trip.  `valid` may be NULL (every KF feature holds a good MapPoint).  fetch() is not used with such a
 * plan: match_f2kf[cap] / n_matches are the caller's device arrays. */
typedef struct OrbmBowSideDevice {
    const uint8_t* desc; const OrbxKeyPoint* kps; const int32_t* n; int32_t cap;      /* extractor outputs of one frame */
    const uint8_t* valid;                                                                /* KF side only, may be NULL */
    const uint32_t* fv_node; const int32_t* fv_off; const uint32_t* fv_feat; const int32_t* n_fv_nodes;   /* transform outputs */
} OrbmBowSideDevice;
typedef struct OrbmBowPairDevice {
    OrbmBowSideDevice kf, f;
    int32_t* match_f2kf;    /* device, f.cap entries */
    int32_t* n_matches;     /* device */
} OrbmBowPairDevice;
int orbm_bow_plan_create_device(orbm_matcher* m, const OrbmBowPairDevice* pairs, int n_pairs, orbm_bow_plan** out);

/* int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:765-905).
 * match12[n1] = feature index in KF2 or -1.  Strict `< TH_LOW` as the reference (:848). */
int orbm_search_by_bow_kfkf(orbm_matcher* m,
                            const uint8_t* desc1, int n1, const uint8_t* valid1, const float* angle1, const OrbmFeatVec* fv1,
                            const uint8_t* desc2, int n2, const uint8_t* valid2, const float* angle2, const OrbmFeatVec* fv2,
                            float nnratio, int check_orientation, int32_t* match12);

/* Frame keypoint grid inputs (Frame::AssignFeaturesToGrid / GetFeaturesInArea, src/Frame.cc:472-503,744-822). */
typedef struct OrbmFrame {
    int32_t n;                         /* F.N */
    const float* x; const float* y;    /* mvKeysUn[i].pt */
    const int32_t* octave;             /* mvKeysUn[i].octave */
    const float* angle;                /* mvKeysUn[i].angle (may be NULL when orientation is not checked) */
    const uint8_t* desc;               /* mDescriptors, n x 32 */
    float min_x, min_y, max_x, max_y;  /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    int32_t grid_cols, grid_rows;      /* FRAME_GRID_COLS, FRAME_GRID_ROWS (64, 48: include/Frame.h:44-45) */
    const float* scale_factors;        /* mvScaleFactors */
    int32_t n_levels;
    const float* u_right;              /* mvuRight (rectified stereo / RGB-D: right-image column of feature i, < 0 when the
                                        * feature has none); NULL for a monocular frame.  Only the two tracking searches
                                        * read it (src/ORBmatcher.cc:92-98, :1751-1757); frames are F.Nleft == -1. */
} OrbmFrame;

/* int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, float th, bool bFarPoints,
 *                                    float thFarPoints) (src/ORBmatcher.cc:43-213), F.Nleft == -1 (monocular and
 * rectified stereo / RGB-D; the stereo-fisheye branch :140-209 is out of scope).
 * Per map point i: in_view = mbTrackInView, proj_u/v = mTrackProjX/Y, proj_ur = mTrackProjXR (= u - mbf*invz,
 * src/Frame.cc:652; read only against features with f->u_right[idx] > 0, the gate of :92-98; may be NULL when
 * f->u_right is NULL), pred_level = mnTrackScaleLevel,
 * view_cos = mTrackViewCos, track_depth = mTrackDepth, bad = isBad(), has_obs = Observations()>0, desc = GetDescriptor().
 * occupied[n] (in/out): F.mvpMapPoints[i] != NULL && Observations()>0.  assign[n] (in/out): map point index now held by
 * feature i (unchanged where the call made no assignment).  Returns nmatches. */
int orbm_search_by_projection(orbm_matcher* m, const OrbmFrame* f,
                              int n_mp, const uint8_t* in_view, const float* proj_u, const float* proj_v, const float* proj_ur,
                              const int32_t* pred_level, const float* view_cos, const float* track_depth,
                              const uint8_t* desc_mp, const uint8_t* mp_has_obs, const uint8_t* mp_bad,
                              float th, int far_points, float th_far, float nnratio,
                              int32_t* assign, uint8_t* occupied);

/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th, bool bMono)
 * (src/ORBmatcher.cc:1676-1887), both frames Nleft == -1 (monocular and rectified stereo / RGB-D).  The caller projects
 * the last frame's map points (Tcw*x3Dw, Pinhole::project) with the reference's own float expressions and clears
 * last_valid where invzc<0, the point is NULL or an outlier.  proj_ur[i] = uv(0) - CurrentFrame.mbf*invzc (:1753, float;
 * read only against features with cur->u_right[i2] > 0; may be NULL when cur->u_right is NULL).
 * last_octave/last_angle = LastFrame.mvKeys[i].octave / mvKeysUn[i].angle.
 * level_window: ORBM_LEVELS_AROUND (neither bForward nor bBackward, always so for bMono: levels octave-1 .. octave+1, :1733),
 * ORBM_LEVELS_FORWARD (bForward = tlc(2) > mb && !bMono, :1692: levels >= octave, :1729) or ORBM_LEVELS_BACKWARD
 * (bBackward = -tlc(2) > mb && !bMono, :1693: levels 0 .. octave, :1731). */
#define ORBM_LEVELS_AROUND   0
#define ORBM_LEVELS_FORWARD  1
#define ORBM_LEVELS_BACKWARD 2
int orbm_search_by_projection_last(orbm_matcher* m, const OrbmFrame* cur,
                                   int n_last, const uint8_t* last_valid, const float* proj_u, const float* proj_v, const float* proj_ur,
                                   const int32_t* last_octave, const float* last_angle,
                                   const uint8_t* desc_mp, const uint8_t* mp_has_obs,
                                   float th, int level_window, int check_orientation,
                                   int32_t* assign, uint8_t* occupied);

/* Batched forms of the two tracking searches: n_frames independent (frame, projected points) problems -- e.g. one frame
 * of every client stream of a tracking server -- in ONE launch, a wave per frame.  Field meaning as in the single-frame
 * calls above (level = pred_level / last_octave, angle = last_angle; level_window is per query because bForward /
 * bBackward depend on each stream's own motion); n_matches is written per query. */
typedef struct OrbmProjQuery {
    const OrbmFrame* frame;
    int32_t n_pts;
    const uint8_t* valid; const float* proj_u; const float* proj_v; const float* proj_ur; const int32_t* level;
    int32_t level_window;                                                          /* last-frame search only: ORBM_LEVELS_* */
    const float* view_cos; const float* track_depth; const uint8_t* mp_bad;      /* Frame x map points only */
    const float* angle;                                                            /* last-frame search only */
    const uint8_t* desc_mp; const uint8_t* mp_has_obs;
    int32_t* assign; uint8_t* occupied;     /* in/out, frame->n entries */
    int32_t n_matches;                      /* out */
} OrbmProjQuery;
int orbm_search_by_projection_last_batch(orbm_matcher* m, OrbmProjQuery* queries, int n_frames, float th, int check_orientation);
int orbm_search_by_projection_batch(orbm_matcher* m, OrbmProjQuery* queries, int n_frames, float th, int far_points, float th_far, float nnratio);

/* Device-resident batch of SearchByProjection(CurrentFrame, LastFrame, th, bMono = true) (src/ORBmatcher.cc:1676-1887;
 * Tracking::TrackWithMotionModel, src/Tracking.cc:2975) for `batch` monocular streams, nothing visits the host:
 * the current frames are what orbx_extract_batch_device (or orbe_unpack_batch_device + orbe_undistort_batch_device) left in
 * HBM -- [batch][cap] key-point records (pt = mvKeysUn[i].pt: for a camera with distortion pass the undistorted records),
 * descriptors and counts; Frame::AssignFeaturesToGrid (src/Frame.cc:472-503) runs on the device.  One entry per LastFrame
 * feature i, [batch][cap] arrays: valid = LastFrame.mvpMapPoints[i] && !LastFrame.mvbOutlier[i] && invzc >= 0, (u, v) =
 * project(Tcw * x3Dw) (:1704-1709; the image-bounds test :1711-1714 is done here), octave / angle =
 * LastFrame.mvKeysUn[i], desc = pMP->GetDescriptor().  d_assign[batch][cur.cap] (in/out): index of the last-frame feature
 * whose map point feature i now holds; d_occupied likewise = CurrentFrame.mvpMapPoints[i] && Observations() > 0;
 * d_n_matches[batch].  Only enqueues on `stream`. */
typedef struct OrbmDeviceFrames {
    const OrbxKeyPoint* d_kps; const uint8_t* d_desc; const int32_t* d_n; int32_t cap;
    float min_x, min_y, max_x, max_y;          /* mnMinX .. mnMaxY */
    int32_t grid_cols, grid_rows;              /* FRAME_GRID_COLS, FRAME_GRID_ROWS */
    const float* scale_factors; int32_t n_levels;      /* HOST array, mvScaleFactors */
} OrbmDeviceFrames;
typedef struct OrbmDeviceLastPoints {
    const uint8_t* d_valid; const float* d_u; const float* d_v; const int32_t* d_octave; const float* d_angle;
    const uint8_t* d_desc; const int32_t* d_n; int32_t cap;
    const uint8_t* d_has_obs;                  /* pMP->Observations() > 0 per entry; NULL = all (monocular maps hold no temporal points) */
} OrbmDeviceLastPoints;
int orbm_search_by_projection_last_batch_device(orbm_matcher* m, const OrbmDeviceFrames* cur, const OrbmDeviceLastPoints* last, int batch,
                                                float th, int check_orientation, int32_t* d_assign, uint8_t* d_occupied,
                                                int32_t* d_n_matches, void* stream);
/* The same for SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints) (:43-213;
 * Tracking::SearchLocalPoints, src/Tracking.cc:3557) on monocular frames: `points` holds per local map point valid =
 * mbTrackInView, (u, v) = mTrackProjX / Y, octave = mnTrackScaleLevel, desc = GetDescriptor(), d_has_obs = Observations() > 0
 * (required here; d_angle is not read); `extras` the rest of what the reference reads per point and the call's parameters. */
typedef struct OrbmDeviceMapPointExtras {
    const float* d_view_cos;        /* mTrackViewCos, [batch][points->cap] */
    const float* d_track_depth;     /* mTrackDepth */
    const uint8_t* d_bad;           /* isBad() */
    float th_far, nnratio; int32_t far_points;
} OrbmDeviceMapPointExtras;
int orbm_search_by_projection_batch_device(orbm_matcher* m, const OrbmDeviceFrames* cur, const OrbmDeviceLastPoints* points,
                                           const OrbmDeviceMapPointExtras* extras, int batch, float th,
                                           int32_t* d_assign, uint8_t* d_occupied, int32_t* d_n_matches, void* stream);

/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, float th,
 *                                    int ORBdist) (src/ORBmatcher.cc:1889-2010; Tracking::Relocalization).
 * One entry per pKF->GetMapPointMatches()[i]: valid[i] = pMP && !isBad() && !sAlreadyFound.count(pMP) && dist3D inside
 * [GetMinDistanceInvariance, GetMaxDistanceInvariance]; proj_u/v = mpCamera->project(Tcw*x3Dw) (the image-bounds test
 * :1917-1920 is done on the device); pred_level = PredictScale(dist3D, &CurrentFrame); kf_angle[i] = pKF->mvKeysUn[i].angle.
 * occupied[n] (in/out) = CurrentFrame.mvpMapPoints[i2] != NULL; assign[n] as above.  Returns nmatches. */
int orbm_search_by_projection_kf(orbm_matcher* m, const OrbmFrame* cur,
                                 int n_pts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                 const int32_t* pred_level, const float* kf_angle, const uint8_t* desc_mp,
                                 float th, int orb_dist, int check_orientation, int32_t* assign, uint8_t* occupied);

/* int ORBmatcher::SearchByProjection(KeyFrame* pKF, Sim3f& Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>&
 *     vpMatched, int th, float ratioHamming) (:427-532) and the overload that also records the points' key frames
 * (:534-646; the caller maps assign[] through vpPointsKFs).  `kf` describes pKF (its grid is the same 64x48 CSR);
 * valid[i] = the prelude :447-487 passed; occupied[n] (in/out) = vpMatched[idx] != NULL. */
int orbm_search_by_projection_sim3(orbm_matcher* m, const OrbmFrame* kf,
                                   int n_pts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                   const int32_t* pred_level, const uint8_t* desc_mp, int th, float ratio_hamming,
                                   int32_t* assign, uint8_t* occupied);

/* Search core of int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, float th, bool bRight=false)
 * (:1148-1338, chi2_check = 1) and Fuse(KeyFrame*, Sim3f& Scw, vpPoints, th, vpReplacePoint) (:1340-1455, chi2_check = 0).
 * The candidate points do not interact, so the device returns, per point, the most similar key point of the window
 * (best_idx[i] or -1, best_dist[i]); the caller then runs the reference's `if(bestDist<=TH_LOW)` pointer surgery
 * (Replace / AddObservation / AddMapPoint) in order.  valid[i] = prelude passed (:1176-1239); proj_ur = u - bf*invz;
 * u_right = pKF->mvuRight, inv_level_sigma2 = pKF->mvInvLevelSigma2 (both may be NULL when chi2_check = 0).
 * One wave per candidate point: LocalMapping::SearchInNeighbors fuses thousands of points per key frame. */
int orbm_fuse_search(orbm_matcher* m, const OrbmFrame* kf, const float* u_right, const float* inv_level_sigma2,
                     int n_pts, const uint8_t* valid, const float* proj_u, const float* proj_v, const float* proj_ur,
                     const int32_t* pred_level, const uint8_t* desc_mp, float th, int chi2_check,
                     int32_t* best_idx, int32_t* best_dist);

/* int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<pair<size_t,size_t>>& vMatchedPairs,
 *                                        bool bOnlyStereo, bool bCoarse) (src/ORBmatcher.cc:907-1146; LocalMapping::
 * CreateNewMapPoints), conventional cameras (mpCamera2 == NULL, NLeft == -1).  Per side: has_mp[i] = GetMapPoint(i) != NULL,
 * stereo[i] = mvuRight[i] >= 0, x/y/octave/angle = mvKeysUn, fv = mFeatVec.  ep = mpCamera->project(T2w * Cw1) (:918-920);
 * F12 = K1^-T [t12]x R12 K2^-1, row-major (Pinhole::epipolarConstrain, src/CameraModels/Pinhole.cpp:109-112);
 * level_sigma2_2 / scale_factors_2 = pKF2->mvLevelSigma2 / mvScaleFactors.  match12[n1] = KF2 feature or -1 (the caller
 * builds vMatchedPairs from it in index order, :1133-1143).  Returns nmatches.  Every KF1 feature is matched in parallel:
 * the reference's loop never writes vbMatched2, so there is no order dependence. */
typedef struct OrbmTriSide {
    int32_t n;
    const uint8_t* desc; const uint8_t* has_mp; const uint8_t* stereo;
    const float* x; const float* y; const int32_t* octave; const float* angle;
    OrbmFeatVec fv;
} OrbmTriSide;
int orbm_search_for_triangulation(orbm_matcher* m, const OrbmTriSide* kf1, const OrbmTriSide* kf2, float ep_x, float ep_y, const float* F12,
                                  const float* level_sigma2_2, const float* scale_factors_2, int n_levels_2,
                                  int only_stereo, int coarse, int check_orientation, int32_t* match12);

/* int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12,
 *                                         int windowSize) (:648-763; monocular map initialisation).  prev_x/y = vbPrevMatched;
 * f2 describes F2 (grid, descriptors, angles).  match12[n1] = vnMatches12; the caller then refreshes vbPrevMatched from
 * it (:757-760).  Returns nmatches. */
int orbm_search_for_initialization(orbm_matcher* m, const uint8_t* desc1, int n1, const int32_t* octave1, const float* angle1,
                                   const float* prev_x, const float* prev_y, const OrbmFrame* f2,
                                   int window_size, float nnratio, int check_orientation, int32_t* match12);

/* Map-point upkeep that LocalMapping / Tracking run on every map point of a new key frame (src/LocalMapping.cc:708-710,
 * :810-822, src/Tracking.cc:2530, :3457) and the BA epilogues run on every optimised point (src/Optimizer.cc:1494), batched.
 *
 * orbm_distinctive_descriptors = MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-402): map point p owns the
 * descriptors desc[off[p] .. off[p+1]) in the reference's push order (observations in std::map order, left index before right
 * index, bad key frames skipped, :348-363).  best_idx[p] = BestIdx (row with the least median Hamming distance to the rest,
 * first one on ties, median = sorted row [0.5*(N-1)]), relative to off[p]; -1 for a point without descriptors (the reference
 * returns early, :365).  best_median (may be NULL) = BestMedian.
 *
 * orbm_update_normal_and_depth = MapPoint::UpdateNormalAndDepth (:433-493): pos[p] = mWorldPos; centers[off[p] .. off[p+1]) =
 * the camera centres of the observations in loop order (GetCameraCenter / GetRightCameraCenter, :451-468); ref_center[p] =
 * pRefKF->GetCameraCenter(); level_scale[p] = pRefKF->mvScaleFactors[level] (:471-484); last_level_scale =
 * mvScaleFactors[nLevels-1].  Outputs mNormalVector, mfMaxDistance, mfMinDistance in the reference's float expressions. */
int orbm_distinctive_descriptors(orbm_matcher* m, const uint8_t* desc, const int32_t* off, int n_points, int32_t* best_idx, int32_t* best_median);
int orbm_update_normal_and_depth(orbm_matcher* m, const float* pos, const float* centers, const int32_t* off, const float* ref_center,
                                 const float* level_scale, float last_level_scale, int n_points,
                                 float* normal, float* max_dist, float* min_dist);

/* ------------------------------------------------------------------------------------------------
 * Local bundle adjustment -- replaces the numerical core of
 * Optimizer::LocalBundleAdjustment(KeyFrame*, bool* pbStopFlag, Map*, int&, int&, int&, int&)
 * (include/Optimizer.h:58, src/Optimizer.cc:1116-1498) and, with robust=0 / other Huber deltas, of
 * Optimizer::BundleAdjustment (:60-390).  The pointer-graph walk (:1118-1186) and the map write-back
 * (:1464-1497) stay in the host shim.
 * ------------------------------------------------------------------------------------------------ */
typedef struct LbaProblem {
    int32_t n_poses;              /* local (optimisable) + fixed keyframes, in ascending vertex id */
    const double* pose_q;         /* [n_poses][4] qx qy qz qw of Tcw (SE3Quat ctor normalises, se3quat.h:59-61) */
    const double* pose_t;         /* [n_poses][3] */
    const uint8_t* pose_fixed;    /* [n_poses] setFixed() */
    int32_t n_points;
    const double* points;         /* [n_points][3] world positions, ascending vertex id */
    int32_t n_edges;              /* in optimizer.addEdge() order */
    const int32_t* edge_point;    /* [n_edges] */
    const int32_t* edge_pose;     /* [n_edges] */
    const double* edge_obs;       /* [n_edges][3] u, v, u_right (ignored for mono edges) */
    const double* edge_inv_sigma2;/* [n_edges] mvInvLevelSigma2[octave] promoted to double */
    const uint8_t* edge_stereo;   /* [n_edges] 0: EdgeSE3ProjectXYZ, 1: g2o::EdgeStereoSE3ProjectXYZ */
    double fx, fy, cx, cy, bf;    /* float camera parameters promoted to double */
    double huber_mono;            /* (double)(float)sqrt(5.991) for LocalBA; <= 0 disables the robust kernel */
    double huber_stereo;          /* (double)(float)sqrt(7.815) */
} LbaProblem;

typedef struct LbaStats {
    int32_t iterations;           /* outer iterations (OptimizationAlgorithmLevenberg::solve calls) */
    int32_t trials;               /* total LM trials */
    int32_t stop_reason;          /* 0 max iters, 1 trials exhausted / rho==0, 2 Raul's nBad>=3 rule, 3 stop flag, 4 solver failure */
    double lambda;
    double chi2_initial;
    double chi2_final;
    double chi2_trace[16];
} LbaStats;

typedef struct lba_solver lba_solver;
int lba_create(int device, lba_solver** out);
void lba_destroy(lba_solver* s);

/* optimizer.initializeOptimization(); optimizer.optimize(max_iters) (src/Optimizer.cc:1410-1411).
 * stop_flag: the caller's bool* pbStopFlag, polled (never written) between outer iterations and LM trials.
 * lambda_init: 0 => tau*max(diag) (g2o computeLambdaInit), >0 => setUserLambdaInit (src/Optimizer.cc:1197-1198).
 * Outputs: optimised poses/points (all n_poses / n_points entries; fixed ones unchanged up to normalisation),
 * chi2_per_edge = e->chi2() and depth_positive = e->isDepthPositive() as the epilogue (:1417-1460) reads them. */
int lba_solve(lba_solver* s, const LbaProblem* problem, const volatile uint8_t* stop_flag, int max_iters, double lambda_init,
              double* pose_q_out, double* pose_t_out, double* points_out,
              double* chi2_per_edge, uint8_t* depth_positive, LbaStats* stats);

/* Many windows per launch (one LocalBundleAdjustment per client session / map, src/LocalMapping.cc:158):
 * n_windows <= 64 independent problems, each solved exactly as lba_solve would solve it (bit-identical results), with ONE sequence
 * of kernel launches per Levenberg round for all of them (grid.y = window).  outputs[i] / stats[i] belong to problems[i]; any
 * pointer inside LbaOutputs may be NULL.  stop_flags: NULL, or one pointer per window (entries may be NULL) = that session's
 * bool* pbStopFlag.  Windows of more than 480 reduced unknowns (80 free key frames) are refused (ORBX_ERR_CAPACITY): use lba_solve. */
typedef struct LbaOutputs {
    double* pose_q; double* pose_t; double* points; double* chi2_per_edge; uint8_t* depth_positive;
} LbaOutputs;
typedef struct lba_batch lba_batch;
int lba_batch_create(int device, lba_batch** out);
void lba_batch_destroy(lba_batch* b);
int lba_solve_batch(lba_batch* b, const LbaProblem* problems, const LbaOutputs* outputs, int n_windows,
                    const volatile uint8_t* const* stop_flags, int max_iters, double lambda_init, LbaStats* stats);
double lba_batch_last_device_ms(const lba_batch* b);   /* device time of the last call's Levenberg rounds (HIP events) */

/* Sharded global BA: landmarks (with all their edges) are partitioned over ranks, poses replicated.
 * The same entry points also drive the single-GPU lba_solve().  One outer LM iteration on every rank:
 *   lba_shard_linearize()                      errors + buildSystem on the accepted state; chi2 and max diagonals
 *      -> caller: all-reduce SUM chi2, all-reduce MAX the diagonals (iteration 0: lambda = tau * max)
 *   repeat (LM trial):
 *     lba_shard_reduce(lambda)                 this rank's partial reduced camera system into the reduce buffer
 *        -> caller: all-reduce SUM of lba_shard_reduce_buffer() (a DEVICE pointer; RCCL over xGMI)
 *     lba_shard_finish(lambda, ...)            += lambda I, Cholesky, back-substitution of the local landmarks,
 *                                              trial state, errors -> chi2_local_new, scale terms
 *        -> caller: all-reduce SUM (chi2_local_new, scale_landmarks_local); rho test (levenberg.cpp:129-147)
 *     lba_shard_accept(accept)                 discardTop() / pop()
 * See INTEGRATION.md for the torch.distributed / RCCL side. */
typedef struct lba_shard lba_shard;
int lba_shard_create(int device, const LbaProblem* local_problem, lba_shard** out);
void lba_shard_destroy(lba_shard* s);
/* number of doubles in the reduce buffer: n*n + 3n with n = 6 * (number of non-fixed poses):
 * [ S (n x n, row-major) | b_schur (n) | b_p (n) | diag(Hpp) (n) ] -- every section is additive over shards. */
int64_t lba_shard_reduce_len(const lba_shard* s);
double* lba_shard_reduce_buffer(lba_shard* s);
/* optional: announce the lambda of the first trial after the NEXT lba_shard_linearize (known from the second LM iteration on);
 * the linearisation then also does the landmark side of the Schur complement and lba_shard_reduce skips that launch */
int lba_shard_hint_lambda(lba_shard* s, double lambda);
/* optional: use a caller-owned device buffer of lba_shard_reduce_len() doubles (e.g. a torch CUDA tensor) instead */
int lba_shard_set_reduce_buffer(lba_shard* s, double* device_buffer);
/* local != 0: world size 1, no all-reduce between reduce() and finish() -> lambda is folded into the Schur kernel, no sync */
int lba_shard_set_local(lba_shard* s, int local);
/* Stream hand-over for the exchange step without a host synchronisation: fence_out makes `other_stream` (a hipStream_t, the
 * collective's stream) wait for the shard's work enqueued so far, fence_in makes the shard's stream wait for `other_stream`.
 * After lba_shard_set_async_reduce(s, 1), lba_shard_reduce() only enqueues; the caller brackets its all-reduce of the reduce
 * buffer with the two fences. */
int lba_shard_fence_out(lba_shard* s, void* other_stream);
int lba_shard_fence_in(lba_shard* s, void* other_stream);
int lba_shard_set_async_reduce(lba_shard* s, int on);
/* Per-stage device time, HIP events on the shard's stream: enable(1) starts a fresh profile, read() returns the milliseconds
 * accumulated since per stage: {linearise, Schur complement, factorisation, substitution, update + errors, reductions,
 * gaps between the groups of launches}. */
int lba_shard_profile_enable(lba_shard* s, int on);
int lba_shard_profile_read(lba_shard* s, float* stage_ms, int n_stages);
int lba_shard_linearize(lba_shard* s, double* chi2_local, double* max_diag_poses_local, double* max_diag_landmarks_local);
int lba_shard_reduce(lba_shard* s, double lambda);
/* returns 1 if the reduced system was solved, 0 if it was not positive definite (step is rejected), <0 on error.
 * scale_poses is identical on every rank (count it once); scale_landmarks_local must be summed over ranks. */
int lba_shard_finish(lba_shard* s, double lambda, double* chi2_local_new, double* scale_poses, double* scale_landmarks_local);
int lba_shard_accept(lba_shard* s, int accept);
int lba_shard_reset(lba_shard* s);     /* back to the initial estimates (benchmarks re-run without re-uploading) */
/* The Levenberg-Marquardt loop above as ONE call, for C / C++ hosts: optimizer.optimize(max_iters) of
 * Optimizer::BundleAdjustment (src/Optimizer.cc:60-390, called by GlobalBundleAdjustemnt :52-58 from LoopClosing.cc:2288 and
 * Tracking.cc:2722) over a landmark shard.  Every rank (one process or thread per GPU) calls it on its own shard with the same
 * arguments; `allreduce` is called on every rank in the same order with a DEVICE buffer of `count` doubles that must be reduced
 * in place over all ranks (op: LBA_REDUCE_SUM / LBA_REDUCE_MAX), enqueued on `hip_stream` (the shard's stream: kernels before
 * and after are ordered by the stream, no host synchronisation is needed) -- with RCCL the whole callback is
 *     return ncclAllReduce(buf, buf, count, ncclDouble, op == LBA_REDUCE_MAX ? ncclMax : ncclSum, comm, (hipStream_t)hip_stream);
 * It returns 0 on success.  Per LM trial there is one call on the reduce buffer ([S | b_schur | b_p | diag Hpp], n*n + 3n
 * doubles) and two on packs of <= 3 scalars (chi2 / scale / solver-ok; the abort flag as a MAX so that all ranks stop together);
 * the first iteration makes one more exchange for g2o's lambda initialisation.  allreduce == NULL (world_size must be 1 then):
 * no exchange -- this is what lba_solve runs.  stop_flag, max_iters, lambda_init, stats: as lba_solve. */
enum { LBA_REDUCE_SUM = 0, LBA_REDUCE_MAX = 1 };
typedef int (*lba_allreduce_fn)(void* user, double* device_buffer, int64_t count, int op, void* hip_stream);
int lba_shard_optimize(lba_shard* s, lba_allreduce_fn allreduce, void* user, int world_size, int max_iters, double lambda_init,
                       const volatile uint8_t* stop_flag, LbaStats* stats);
/* estimates of the accepted state; chi2_per_edge = e->chi2() of the last computed errors, depth_positive = isDepthPositive() */
int lba_shard_download(lba_shard* s, double* pose_q, double* pose_t, double* points, double* chi2_per_edge, uint8_t* depth_positive);

/* ------------------------------------------------------------------------------------------------------------------
 * Motion-only BA.  Replaces  int Optimizer::PoseOptimization(Frame* pFrame)
 * (reference include/Optimizer.h:66, src/Optimizer.cc:814-1115) for the pinhole mono / stereo edges
 * (EdgeSE3ProjectXYZOnlyPose src/OptimizableTypes.cpp:49-63, g2o::EdgeStereoSE3ProjectXYZOnlyPose
 * Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:338-395).  The caller flattens the frame: one edge per feature i that
 * holds a MapPoint (pFrame->mvpMapPoints[i]), in feature order; after the call it writes outlier[] back to
 * pFrame->mvbOutlier[i], the pose to pFrame->SetPose() and returns `inliers` (= nInitialCorrespondences - nBad).
 * The whole 4-round Levenberg optimisation of a frame is one workgroup of one kernel launch; a batch is one launch. */
typedef struct PoseProblem {
    double q[4], t[3];              /* pFrame->GetPose() as Tcw: qx qy qz qw, t (normalised on entry like g2o::SE3Quat) */
    int32_t n;                      /* edges */
    const double* Xw;               /* n x 3 MapPoint::GetWorldPos() */
    const double* obs;              /* n x 3: kpUn.pt.x, kpUn.pt.y, mvuRight (ignored for mono edges) */
    const double* inv_sigma2;       /* n: pFrame->mvInvLevelSigma2[kpUn.octave] */
    const uint8_t* stereo;          /* n: 1 = stereo edge (mvuRight[i] >= 0) */
    double fx, fy, cx, cy, bf;      /* pinhole intrinsics, pFrame->mbf */
    double huber_mono, huber_stereo;/* deltaMono = sqrt(5.991), deltaStereo = sqrt(7.815) (:838-839) */
} PoseProblem;

typedef struct PoseResult {
    double q[4], t[3];              /* optimised Tcw */
    int32_t inliers;                /* the function's return value; 0 when n < 3 (:998-999) */
    int32_t n_bad;
    int32_t iterations[4], trials[4];   /* per round of optimize(10): outer iterations and Levenberg trials executed */
    double chi2[4];                 /* per round: activeRobustChi2 after the last trial */
} PoseResult;

typedef struct pose_solver pose_solver;
int  pose_create(int device, pose_solver** out);
void pose_destroy(pose_solver* s);
/* outlier: n bytes (may be NULL) */
int  pose_optimize(pose_solver* s, const PoseProblem* problem, PoseResult* result, uint8_t* outlier);
/* frames are independent: one workgroup each; outlier_out may be NULL or hold NULL entries */
int  pose_optimize_batch(pose_solver* s, const PoseProblem* problems, int n_problems, PoseResult* results, uint8_t* const* outlier_out);
/* device time of the kernel of the LAST call (HIP events on the solver's stream), milliseconds */
float pose_last_kernel_ms(const pose_solver* s);

/* Device-resident batch of Optimizer::PoseOptimization (src/Optimizer.cc:814-1115) for `batch` frames whose features and matches
 * are already in HBM (Tracking::TrackWithMotionModel, src/Tracking.cc:3053: right behind SearchByProjection(CurrentFrame,
 * LastFrame); TrackLocalMap :3115): nothing visits the host.  The edges of frame b are gathered on the device in feature order
 * (the order the reference walks pFrame->mvpMapPoints, :861-996): feature i holds a map point iff d_assign[b][i] >= 0 -- the row
 * of d_mp_xyz[b] with its GetWorldPos() (for the last-frame search: the index orbm_search_by_projection_last_batch_device wrote) --
 * observation = d_kps[b][i].pt (mvKeysUn: pass undistorted records) and d_u_right[b][i] (stereo edge iff >= 0; NULL = monocular
 * frames), information = inv_level_sigma2[octave]; floats become doubles exactly where the reference casts them.
 * Outputs (device, any may be NULL): d_pose_out[batch][7] = optimised Tcw (qx qy qz qw tx ty tz), d_inliers[batch] = the
 * function's return value, d_outlier[batch][cap] = pFrame->mvbOutlier (0 for features without a map point), d_results[batch].
 * Only enqueues on `stream`; one solver handle serves one stream at a time. */
typedef struct PoseDeviceFrames {
    const OrbxKeyPoint* d_kps; const int32_t* d_n; const float* d_u_right; int32_t cap;
    const int32_t* d_assign;                /* [batch][cap] */
    const float* d_mp_xyz; int32_t mp_cap;  /* [batch][mp_cap][3] */
    const double* d_pose;                   /* [batch][7] pFrame->GetPose() as Tcw, normalised on entry like g2o::SE3Quat */
    const float* inv_level_sigma2; int32_t n_levels;      /* HOST array, pFrame->mvInvLevelSigma2 (<= 16 levels) */
    double fx, fy, cx, cy, bf, huber_mono, huber_stereo;
} PoseDeviceFrames;
int  pose_optimize_batch_device(pose_solver* s, const PoseDeviceFrames* frames, int batch, double* d_pose_out, int32_t* d_inliers,
                                uint8_t* d_outlier, PoseResult* d_results, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Vocabulary transform.  Replaces, for FORB descriptors with TF_IDF weights and L1 scoring (what
 * ORBvoc.txt declares),  void TemplatedVocabulary::transform(const vector<TDescriptor>& features, BowVector& v,
 * FeatureVector& fv, int levelsup) const  (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1193, per-feature descent
 * :1216-1259), as called by Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:825-832, src/KeyFrame.cc:253-265) with
 * levelsup = 4.  The tree is handed over flattened; loading ORBvoc.txt stays with the reference's loader.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct OrbvVocabulary {     /* TemplatedVocabulary::m_nodes (:162-197), node 0 = root */
    int32_t n_nodes, L;             /* m_nodes.size(), m_L */
    const int32_t* child_off;       /* [n_nodes + 1]: children of node i are child_id[child_off[i] .. child_off[i+1]) */
    const uint32_t* child_id;       /* m_nodes[i].children, in order */
    const uint8_t* desc;            /* n_nodes x 32: m_nodes[i].descriptor */
    const double* weight;           /* m_nodes[i].weight */
    const int32_t* word_id;         /* m_nodes[i].word_id for leaves, -1 otherwise */
} OrbvVocabulary;

typedef struct orbv_vocab orbv_vocab;
int  orbv_create(int device, const OrbvVocabulary* voc, orbv_vocab** out);     /* uploads the tree (35 MB for ORBvoc) once */
void orbv_destroy(orbv_vocab* v);
/* transform(feature, word_id, weight, &nid, levelsup) for n features (host buffers). */
int  orbv_transform_features(orbv_vocab* v, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node);
/* transform(features, v, fv, levelsup) of one frame, host buffers of n entries (fv_off: n + 1).  BowVector = ascending
 * (bow_id, bow_val) pairs, L1-normalised, values bit-identical to the reference's order of double additions; FeatureVector
 * = CSR with ascending node ids and feature indices in insertion order.  Returns the number of features in fv or <0. */
int  orbv_transform(orbv_vocab* v, const uint8_t* desc, int n, int levelsup, uint32_t* bow_id, double* bow_val, int32_t* n_bow,
                    uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int32_t* n_fv_nodes);
/* Device-resident batch: d_desc is the extractor's descriptor output [batch][cap][32] with d_n[batch] live rows per frame;
 * every output is a device array with `cap` entries per frame (d_fv_off: cap + 1).  Only enqueues on `stream`. */
int  orbv_transform_batch_device(orbv_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int batch, int cap, int levelsup,
                                 uint32_t* d_bow_id, double* d_bow_val, int32_t* d_n_bow,
                                 uint32_t* d_fv_node, int32_t* d_fv_off, uint32_t* d_fv_feat, int32_t* d_n_fv, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Visual-inertial local BA -- the numerical core of
 * Optimizer::LocalInertialBA(KeyFrame*, bool*, Map*, int&, int&, int&, int&, bool bLarge, bool bRecInit)
 * (include/Optimizer.h, src/Optimizer.cc:2383-2958).  The graph walk (:2383-2500) and the map
 * write-back (:2862-2957) stay in the host shim, which also reads the pre-integrated terms off IMU::Preintegrated
 * (IntegrateNewMeasurement stays on the host) and forms the information matrices (G2oTypes.cc:510-518, Optimizer.cc:2651-2668).
 * Vertices: per key frame a body pose (VertexPose / ImuCamPose, 6), velocity, gyro bias, accelerometer bias (3 each);
 * landmarks marginalised.  Edges: EdgeMono / EdgeStereo, EdgeInertial, EdgeGyroRW, EdgeAccRW.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct LibaLink {           /* one pre-integrated link kf1 -> kf2: EdgeInertial + EdgeGyroRW + EdgeAccRW */
    int32_t kf1, kf2;
    float dR[9], dV[3], dP[3];      /* IMU::Preintegrated dR, dV, dP (row major) */
    float JRg[9], JVg[9], JVa[9], JPg[9], JPa[9];
    float dT;
    float bias0[6];                 /* the bias of the pre-integration: bax bay baz bwx bwy bwz */
    double info9[81];               /* EdgeInertial information, symmetrised and eigenvalue-clamped (x 1e-2 on the link to the fixed key frame) */
    double info_gyro[9], info_acc[9];
    uint8_t robust;                 /* Huber kernel on the inertial edge (i == N-1 || bRecInit, Optimizer.cc:2643-2653) */
} LibaLink;
typedef struct LibaProblem {
    int32_t n_kf;
    const double* Rwb;              /* [n_kf][9] GetImuRotation() */
    const double* twb;              /* [n_kf][3] GetImuPosition() */
    const double* vel;              /* [n_kf][3] */
    const double* bg;               /* [n_kf][3] */
    const double* ba;               /* [n_kf][3] */
    const uint8_t* pose_fixed;      /* VertexPose::setFixed */
    const uint8_t* has_imu;         /* pKFi->bImu: velocity / bias vertices exist */
    const uint8_t* imu_fixed;
    double Rcb[9], tcb[3], tbc[3];  /* mImuCalib.mTcb, mTbc translation */
    double fx, fy, cx, cy, bf;
    int32_t n_points;
    const double* points;
    int32_t n_edges;                /* in addEdge order */
    const int32_t* edge_kf;
    const int32_t* edge_point;
    const double* edge_obs;         /* [n_edges][3] */
    const double* edge_inv_sigma2;
    const uint8_t* edge_stereo;
    int32_t n_links;                /* <= 64 */
    const LibaLink* links;
    double huber_mono, huber_stereo, huber_inertial;    /* (float)sqrt(5.991), (float)sqrt(7.815), sqrt(16.92) */
    double lambda_init;             /* setUserLambdaInit: 1e0, or 1e-2 when bLarge */
    int32_t max_iters;              /* opt_it: 10, or 4 when bLarge */
} LibaProblem;
typedef struct liba_solver liba_solver;
int  liba_create(int device, liba_solver** out);
void liba_destroy(liba_solver* s);
/* outputs: per key frame Rwb / twb / velocity / biases (any may be NULL), points, per visual edge chi2 and depth sign */
int  liba_solve(liba_solver* s, const LibaProblem* problem, double* Rwb_out, double* twb_out, double* vel_out, double* bg_out,
                double* ba_out, double* points_out, double* chi2_per_edge, uint8_t* depth_positive, LbaStats* stats);

/* Many windows per launch (one LocalMapping window per client session or map): every stage of the
 * Levenberg loop is ONE launch for all windows (grid.y = window), each window keeps its own lambda / accept-reject state on the
 * host.  liba_solve is this path with one window: results of window i equal liba_solve(problems[i]) bit for bit.  At most 64
 * windows per call; any output pointer may be NULL. */
typedef struct LibaOutputs {
    double* Rwb;                    /* [n_kf][9] */
    double* twb;                    /* [n_kf][3] */
    double* vel;
    double* bg;
    double* ba;
    double* points;                 /* [n_points][3] */
    double* chi2_per_edge;          /* [n_edges] */
    uint8_t* depth_positive;        /* [n_edges] */
} LibaOutputs;
typedef struct liba_batch liba_batch;
int  liba_batch_create(int device, liba_batch** out);
void liba_batch_destroy(liba_batch* b);
int  liba_solve_batch(liba_batch* b, const LibaProblem* problems, const LibaOutputs* outputs, int n_windows, LbaStats* stats);
double liba_batch_last_device_ms(const liba_batch* b);  /* HIP-event time of the Levenberg rounds of the last call */

/* int Optimizer::PoseInertialOptimizationLastKeyFrame(Frame*, bool bRecInit) (src/Optimizer.cc:4491-4873): the per-frame optimisation
 * of the inertial tracker, for a batch of frames (one per client stream) in one launch.  Index [0] of the state arrays is the last key
 * frame (fixed), [1] the current frame.  Edges = the features holding a map point, in feature order; close_point = mTrackDepth < 10.
 * Outputs per frame: the optimised body pose / velocity / biases, mvbOutlier (frame b's flags start at the sum of the earlier frames'
 * n), the Hessian of the new ConstraintPoseImu (:4837-4870; 15 x 15, or 30 x 30 per frame for the last-frame variant), nInitialCorrespondences -
 * nBad (the return value) and nBad.
 * One camera / camera-body calibration per batch. */
typedef struct LibaPoseProblem {
    double Rwb[18], twb[6], vel[6], bg[6], ba[6];
    double Rcb[9], tcb[3], tbc[3];
    double fx, fy, cx, cy, bf;
    int32_t n;
    const double* Xw;               /* [n][3] pMP->GetWorldPos() */
    const double* obs;              /* [n][3] */
    const double* inv_sigma2;       /* mvInvLevelSigma2[octave] / uncertainty2(obs) */
    const uint8_t* stereo;
    const uint8_t* close_point;
    LibaLink link;                  /* pFrame->mpImuPreintegrated, kf1 = 0, kf2 = 1; robust unused */
    double huber_mono, huber_stereo;
    int32_t rec_init;
    /* last_frame != 0: Optimizer::PoseInertialOptimizationLastFrame (src/Optimizer.cc:4875-5285) -- index [0] is the PREVIOUS FRAME, optimised
     * too and tied to its prior pFp->mpcpi (EdgePriorPoseImu, Huber 5); link = pFrame->mpImuPreintegratedFrame; the Hessian output is then
     * 30 x 30 (previous frame 0-14, current frame 15-29), to be passed to Optimizer::Marginalize(H, 0, 14) by the caller (:5282). */
    int32_t last_frame;
    double prior_Rwb[9], prior_twb[3], prior_vel[3], prior_bg[3], prior_ba[3], prior_H[225];
} LibaPoseProblem;
int  liba_pose_optimize_batch(liba_solver* s, const LibaPoseProblem* problems, int batch, double* Rwb_out, double* twb_out, double* vel_out,
                              double* bg_out, double* ba_out, uint8_t* outlier_out, double* H15_out, int32_t* inliers_out, int32_t* n_bad_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Edge-SLAM wire format (the fork's client <-> server packets).  Replaces the two constructors
 * of class SlamPktVI, reference include/Socket/slampkt_vi.h:
 *   :127-167  SlamPktVI(id, timestamp, kps, descriptors, imus)  -> orbe_pack_batch(_device)   (edge client: packets are
 *             written straight from orbx_extract_batch_device's outputs);  getHead() :185-193 -> `head`
 *   :85-125   SlamPktVI(buffer, packet_size)                    -> orbe_unpack_batch(_device) (server: key points /
 *             descriptors land in the layout the matcher and vocabulary kernels read; src/Socket/client.cc:132-143)
 * Packet = 16-byte info block {int32 frame id, int64 time stamp at byte 4, u16 BE #points, u16 BE #imu}, 36 B per key
 * point {u16 BE (unsigned short)pt.x, u16 BE (unsigned short)pt.y, 32 descriptor bytes}, 32 B per IMU sample.
 * Unpacked key points are KeyPoint(x, y, 1): size 1, angle -1, response 0, octave 0, class_id -1 (:101).
 * Batch layout: frame b's key points at kps + b*cap, descriptors at desc + b*cap*32, packet at payload + b*stride
 * (stride a multiple of 4).  Per-frame status: ORBX_OK; ORBX_ERR_CAPACITY (packet does not fit stride / counts exceed
 * cap, imu_cap); ORBX_ERR_ARG (pack: packet larger than 65536 B, which getHead() cannot express; unpack: packet shorter
 * than its own counts -- the reference would read past the buffer).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct OrbeImuSample {      /* class IMUData, include/Socket/imudata.h:10-20; the 32-byte wire record (:152-161) */
    int64_t ts;
    float gyro[3];
    float acce[3];
} OrbeImuSample;

typedef struct orbe_codec orbe_codec;
int  orbe_packet_bytes(int n_pts, int n_imu);                     /* total_len_ = 16 + 36 n_pts + 32 n_imu (:130) */
int  orbe_create(int device, orbe_codec** out);
void orbe_destroy(orbe_codec* c);
/* imu / imu_off may both be NULL (no IMU samples); otherwise frame b owns imu[imu_off[b] .. imu_off[b+1]).  head
 * (2 bytes per frame, may be NULL) receives getHead().  Host buffers. */
int  orbe_pack_batch(orbe_codec* c, const OrbxKeyPoint* kps, const uint8_t* desc, const int32_t* n, int batch, int cap,
                     const int32_t* frame_id, const int64_t* timestamp, const OrbeImuSample* imu, const int32_t* imu_off,
                     uint8_t* payload, int stride, int32_t* len, uint8_t* head, int32_t* status);
int  orbe_unpack_batch(orbe_codec* c, const uint8_t* payload, int stride, const int32_t* len, int batch, int cap, int imu_cap,
                       OrbxKeyPoint* kps, uint8_t* desc, int32_t* n, int32_t* frame_id, int64_t* timestamp,
                       OrbeImuSample* imu /* [batch][imu_cap] */, int32_t* n_imu, int32_t* status);
/* Device-resident forms: every pointer is a device pointer; only enqueue on `stream`. */
int  orbe_pack_batch_device(orbe_codec* c, const OrbxKeyPoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int batch, int cap,
                            const int32_t* d_frame_id, const int64_t* d_timestamp, const OrbeImuSample* d_imu, const int32_t* d_imu_off,
                            uint8_t* d_payload, int stride, int32_t* d_len, uint8_t* d_head, int32_t* d_status, void* stream);
int  orbe_unpack_batch_device(orbe_codec* c, const uint8_t* d_payload, int stride, const int32_t* d_len, int batch, int cap, int imu_cap,
                              OrbxKeyPoint* d_kps, uint8_t* d_desc, int32_t* d_n, int32_t* d_frame_id, int64_t* d_timestamp,
                              OrbeImuSample* d_imu, int32_t* d_n_imu, int32_t* d_status, void* stream);

/* Frame::UndistortKeyPoints (src/Frame.cc:834-867) on the device, for key points that arrived in packets (the server-side Frame constructor,
 * src/Frame.cc:384-470, runs it before AssignFeaturesToGrid): cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK) per key point, every
 * other KeyPoint field copied.  k = (k1, k2, p1, p2, k3) as in mDistCoef; k[0] == 0 is the plain copy of :836-840.  Device pointers,
 * [batch][cap] layout with d_n live rows per frame; only enqueues on `stream`.  d_kps_un may equal d_kps. */
typedef struct OrbeCamera {
    float fx, fy, cx, cy;           /* Pinhole::toK() */
    float k[5];                     /* mDistCoef */
    float fx_new, fy_new, cx_new, cy_new;   /* mK */
} OrbeCamera;
int  orbe_undistort_batch_device(orbe_codec* c, const OrbxKeyPoint* d_kps, const int32_t* d_n, int batch, int cap, const OrbeCamera* cam,
                                 OrbxKeyPoint* d_kps_un, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORBSLAM3_HIP_H */
