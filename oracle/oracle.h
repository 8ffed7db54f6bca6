/*
 * oracle.h -- C interface of the CPU ORACLE (test infrastructure, NOT product).
 *
 * The oracle is a single-threaded CPU restatement of the reference hot path
 * (ORBextractor / ORBmatcher / Optimizer::LocalBundleAdjustment + the g2o code
 * it drives), written from reading the reference as text.  It exists to CHECK
 * the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; nothing under orb_slam3-1_amd/ may.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * for this path and cannot be compiled here (OpenCV / Eigen / Boost absent;
 * see SURVEY.md 8(c)).  OpenCV-4.4.0 primitive semantics (FAST, resize,
 * GaussianBlur, fastAtan2, cvRound) are restated from knowledge of the
 * upstream algorithms; fidelity to a real OpenCV build is not verified.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field layout as cv::KeyPoint (28 bytes). */
typedef struct OracleKeyPoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} OracleKeyPoint;

/* Frame::ComputeStereoMatches (reference src/Frame.cc:931-1101) on the pyramids of the last extract call of hL / hR.
 * Returns the number of matches before the median cut, or -2 if a patch left the image (cv::Mat would assert). */
int   orb_oracle_stereo_matches(void* hL, void* hR, const OracleKeyPoint* kpsL, const uint8_t* descL, int N,
                                const OracleKeyPoint* kpsR, const uint8_t* descR, int Nr, float mb, float mbf,
                                float* uRight, float* depth);

/* ---------------- extractor (reference src/ORBextractor.cc) ---------------- */
void* orb_oracle_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void  orb_oracle_destroy(void* h);
/* returns monoIndex (reference :1167) or -1 for an empty image (:1090). *n_out = number of keypoints. */
int   orb_oracle_extract(void* h, const uint8_t* img, int w, int hgt, int stride, int lap0, int lap1,
                         OracleKeyPoint* kps, uint8_t* desc, int cap, int* n_out);
/* tables of the constructor (reference :409-469) */
void  orb_oracle_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                        int* nfeat_per_level, int* umax16);
/* intermediates of the last extract call, for stage-wise parity */
int   orb_oracle_level_size(void* h, int level, int* w, int* hgt);
int   orb_oracle_level_image(void* h, int level, uint8_t* out /* w*h */);
int   orb_oracle_level_blurred(void* h, int level, uint8_t* out /* w*h, only if level had keypoints */);
int   orb_oracle_level_candidates(void* h, int level, OracleKeyPoint* out, int cap); /* vToDistributeKeys */
int   orb_oracle_level_keypoints(void* h, int level, OracleKeyPoint* out, int cap);  /* after octree + angle, level coords */

/* stand-alone primitives (OpenCV restatements), exposed for micro known-answer tests */
int   orb_oracle_fast(const uint8_t* img, int w, int hgt, int stride, int threshold, int nonmax,
                      OracleKeyPoint* out, int cap);
void  orb_oracle_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void  orb_oracle_gaussian7(const uint8_t* src, int w, int hgt, int sstride, uint8_t* dst, int dstride);
void  orb_oracle_gauss_taps(int ksize, double sigma, int* taps_q8);
float orb_oracle_fast_atan2(float y, float x);
void  orb_oracle_sincos(float angle_rad, float* c, float* s);
int   orb_oracle_cvround(double v);
/* std::sort with the reference's compareNodes ordering on (count, ULx) pairs; used to pin the device introsort */
void  orb_oracle_sort_nodes(int* count, int* ulx, int* tag, int n);

/* ---------------- matcher (reference src/ORBmatcher.cc) ---------------- */
int   orbm_oracle_hamming(const uint8_t* a, const uint8_t* b);   /* :2058-2074 */

/* FeatureVector as CSR: node ids ascending, offsets [n_nodes+1], feature indices. */
typedef struct OracleFeatVec {
    int32_t n_nodes;
    const uint32_t* node_id;
    const int32_t* offset;
    const uint32_t* feat;
} OracleFeatVec;

/* SearchByBoW(KeyFrame*, Frame&, ...) mono branch (:223-425). match_f2kf[nF] = KF feature index or -1. */
int   orbm_oracle_search_by_bow(const uint8_t* dKF, int nKF, const uint8_t* validKF, const float* angKF, const OracleFeatVec* fvKF,
                                const uint8_t* dF, int nF, const float* angF, const OracleFeatVec* fvF,
                                float nnratio, int checkOri, int32_t* match_f2kf);
/* SearchByBoW(KeyFrame*, KeyFrame*, ...) (:765-905). match12[n1] = index in KF2 or -1. */
int   orbm_oracle_search_by_bow_kfkf(const uint8_t* d1, int n1, const uint8_t* valid1, const float* ang1, const OracleFeatVec* fv1,
                                     const uint8_t* d2, int n2, const uint8_t* valid2, const float* ang2, const OracleFeatVec* fv2,
                                     float nnratio, int checkOri, int32_t* match12);

/* Frame grid (reference src/Frame.cc:472-503, 744-822): 64x48 cells, CSR built by the oracle from keypoints. */
typedef struct OracleFrameGrid {
    int32_t n;                 /* number of features */
    const float* x;            /* undistorted keypoint x */
    const float* y;
    const int32_t* octave;
    float min_x, min_y, max_x, max_y;   /* mnMinX .. mnMaxY */
    int32_t cols, rows;                 /* FRAME_GRID_COLS / ROWS (64, 48) */
    const float* u_right;               /* mvuRight (rectified stereo / RGB-D), NULL for a monocular frame; read by the two
                                         * tracking searches only */
} OracleFrameGrid;

/* SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFar, thFar), F.Nleft == -1 (:43-138).
 * Per point: in_view, proj u/v, proj_ur = mTrackProjXR (gate :92-98 against g->u_right), predicted level, view cos, track depth, 32B descriptor, has_obs (Observations()>0).
 * occupied[nF]: in/out, 1 if F.mvpMapPoints[i] holds a MapPoint with Observations()>0.
 * assign[nF]: in/out, index of map point held by feature i (-1 none). */
int   orbm_oracle_search_by_projection(const OracleFrameGrid* g, const uint8_t* dF, const float* scale_factors, int nlevels,
                                       int nMP, const uint8_t* in_view, const float* proj_u, const float* proj_v, const float* proj_ur,
                                       const int32_t* pred_level, const float* view_cos, const float* track_depth,
                                       const uint8_t* dMP, const uint8_t* mp_has_obs, const uint8_t* mp_bad,
                                       float th, int bFar, float thFar, float nnratio,
                                       int32_t* assign, uint8_t* occupied);

/* SearchByProjection(Frame& cur, const Frame& last, th, bMono), Nleft == -1 (:1676-1887).
 * The caller projects (Tcw * x3Dw, Pinhole::project), clears last_valid for invzc<0, computes proj_ur = uv(0) - mbf*invzc
 * (:1753) and level_window (0: octave-1..octave+1; 1: bForward, >= octave; 2: bBackward, 0..octave; :1692-1693,:1728-1733). */
int   orbm_oracle_search_by_projection_last(const OracleFrameGrid* g, const uint8_t* dF, const float* angF,
                                            const float* scale_factors, int nlevels,
                                            int nLast, const uint8_t* last_valid /* has MP && !outlier && invzc>=0 */,
                                            const float* proj_u, const float* proj_v, const float* proj_ur,
                                            const int32_t* last_octave, const float* last_angle,
                                            const uint8_t* dMP, const uint8_t* mp_has_obs,
                                            float th, int level_window, int checkOri,
                                            int32_t* assign, uint8_t* occupied);

/* SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist) (:1889-2010), relocalisation. */
int   orbm_oracle_search_by_projection_kf(const OracleFrameGrid* g, const uint8_t* dF, const float* angF, const float* scale_factors,
                                          int nPts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                          const int32_t* pred_level, const float* kf_angle, const uint8_t* dMP,
                                          float th, int ORBdist, int checkOri, int32_t* assign, uint8_t* occupied);
/* SearchByProjection(KeyFrame*, Sim3f&, vpPoints[, vpPointsKFs], vpMatched[, vpMatchedKF], th, ratioHamming) (:427-646). */
int   orbm_oracle_search_by_projection_sim3(const OracleFrameGrid* g, const uint8_t* dKF, const float* scale_factors,
                                            int nPts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                            const int32_t* pred_level, const uint8_t* dMP, int th, float ratioHamming,
                                            int32_t* assign, uint8_t* occupied);
/* search core of both ORBmatcher::Fuse overloads (:1148-1338 with the chi2 gate, :1340-1455 without). */
void  orbm_oracle_fuse_search(const OracleFrameGrid* g, const uint8_t* dKF, const float* scale_factors,
                              const float* u_right, const float* inv_level_sigma2,
                              int nPts, const uint8_t* valid, const float* proj_u, const float* proj_v, const float* proj_ur,
                              const int32_t* pred_level, const uint8_t* dMP, float th, int chi2_check,
                              int32_t* best_idx, int32_t* best_dist);

/* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:648-763). */
int   orbm_oracle_search_for_initialization(const uint8_t* d1, int n1, const int32_t* octave1, const float* ang1,
                                            const float* prev_x, const float* prev_y,
                                            const OracleFrameGrid* g2, const uint8_t* d2, const float* ang2,
                                            int windowSize, float nnratio, int checkOri, int32_t* match12);
/* SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (:907-1146), conventional cameras. */
int   orbm_oracle_search_for_triangulation(const uint8_t* d1, int n1, const uint8_t* has_mp1, const uint8_t* stereo1,
                                           const float* x1, const float* y1, const float* ang1, const OracleFeatVec* fv1,
                                           const uint8_t* d2, int n2, const uint8_t* has_mp2, const uint8_t* stereo2,
                                           const float* x2, const float* y2, const int32_t* octave2, const float* ang2, const OracleFeatVec* fv2,
                                           float ep_x, float ep_y, const float* F12, const float* sigma2_2, const float* scale2,
                                           int bOnlyStereo, int bCoarse, int checkOri, int32_t* match12);

void  orbm_oracle_three_maxima(const int* hist_counts, int L, int* ind1, int* ind2, int* ind3); /* :2012-2053 */

/* ---------------- local BA (reference src/Optimizer.cc:1116-1498 + g2o) ---------------- */
typedef struct OracleLbaProblem {
    int32_t n_poses;            /* optimisable + fixed */
    const double* pose_q;       /* n_poses x 4: qx qy qz qw  (unit quaternion, world->camera) */
    const double* pose_t;       /* n_poses x 3 */
    const uint8_t* pose_fixed;  /* n_poses */
    int32_t n_points;
    const double* points;       /* n_points x 3 */
    int32_t n_edges;
    const int32_t* edge_point;  /* n_edges */
    const int32_t* edge_pose;
    const double* edge_obs;     /* n_edges x 3: u, v, u_right (u_right ignored for mono) */
    const double* edge_inv_sigma2;
    const uint8_t* edge_stereo; /* 0 mono (EdgeSE3ProjectXYZ), 1 stereo (EdgeStereoSE3ProjectXYZ) */
    double fx, fy, cx, cy, bf;  /* camera parameters already promoted from float */
    double huber_mono, huber_stereo;   /* delta (already through float) ; <=0 => no robust kernel */
} OracleLbaProblem;

typedef struct OracleLbaStats {
    int32_t iterations;         /* outer iterations executed (calls of solve()) */
    int32_t trials;             /* total LM trials */
    int32_t stop_reason;        /* 0 max iters, 1 terminate(trials/rho==0), 2 nBad>=3, 3 stop flag, 4 solver fail */
    double lambda;
    double chi2_initial;
    double chi2_final;
    double chi2_trace[16];
} OracleLbaStats;

/* DBoW2 vocabulary tree, flattened (TemplatedVocabulary::m_nodes, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:162-197):
 * node 0 is the root; children of node i are child_id[child_off[i] .. child_off[i+1]) in m_nodes[i].children order. */
typedef struct OracleVocab {
    int32_t n_nodes, L;             /* m_L: depth levels */
    const int32_t* child_off;       /* [n_nodes + 1] */
    const uint32_t* child_id;
    const uint8_t* desc;            /* n_nodes x 32 (the root's is unused) */
    const double* weight;           /* node weight (idf for words) */
    const int32_t* word_id;         /* word id of a leaf, -1 for inner nodes */
} OracleVocab;
/* per feature: word id, word weight, node id `levelsup` levels above the leaves (TemplatedVocabulary.h:1216-1259) */
void  dbow_oracle_transform_features(const OracleVocab* V, const uint8_t* desc, int n, int levelsup,
                                     uint32_t* word, double* weight, uint32_t* node);
/* transform(features, BowVector&, FeatureVector&, levelsup), TF_IDF + L1 (:1127-1193): BowVector as ascending (id, value)
 * pairs, FeatureVector as CSR (ascending node ids; feature indices in insertion order).  Returns the CSR length. */
int   dbow_oracle_transform(const OracleVocab* V, const uint8_t* desc, int n, int levelsup,
                            uint32_t* bow_id, double* bow_val, int32_t* n_bow,
                            uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int32_t* n_fv_nodes);

/* Optimizer::PoseOptimization (reference src/Optimizer.cc:814-1115): one frame pose, unary reprojection edges. */
typedef struct OraclePoseProblem {
    double q[4], t[3];              /* frame pose Tcw: qx qy qz qw, t */
    int32_t n;                      /* edges = features holding a MapPoint */
    const double* Xw;               /* n x 3 world positions */
    const double* obs;              /* n x 3: u, v, u_right */
    const double* inv_sigma2;       /* n */
    const uint8_t* stereo;          /* n */
    double fx, fy, cx, cy, bf;
    double huber_mono, huber_stereo;
} OraclePoseProblem;
typedef struct OraclePoseStats {        /* per round of optimize(10): outer iterations, LM trials, robust chi2 after the last trial */
    int32_t iterations[4], trials[4];
    double chi2[4];
} OraclePoseStats;
int   pose_oracle_optimize(const OraclePoseProblem* p, double* q_out, double* t_out, uint8_t* outlier_out, int* n_bad_out,
                           OraclePoseStats* stats /* may be NULL */);

int   lba_oracle_solve(const OracleLbaProblem* p, const volatile uint8_t* stop_flag, int max_iters, double lambda_init,
                       double* poses_q_out, double* poses_t_out, double* points_out,
                       double* chi2_per_edge, uint8_t* depth_positive, OracleLbaStats* stats);

/* Edge-SLAM wire format: class SlamPktVI (reference include/Socket/slampkt_vi.h), IMUData (include/Socket/imudata.h). */
typedef struct OracleImuSample {
    int64_t ts;
    float gyro[3];
    float acce[3];
} OracleImuSample;
/* returns total_len_ (or -total_len_ when it exceeds capacity; nothing is written then); head = getHead() */
int   edge_oracle_pack(int32_t frame_id, int64_t timestamp, const OracleKeyPoint* kps, const uint8_t* desc, int n_pts,
                       const OracleImuSample* imu, int n_imu, uint8_t* payload, int capacity, uint8_t head[2]);
/* 0 ok; -1 packet shorter than its info block / its own counts; -2 counts exceed the caller's capacities */
int   edge_oracle_unpack(const uint8_t* payload, int packet_size, int32_t* frame_id, int64_t* timestamp,
                         OracleKeyPoint* kps, uint8_t* desc, int cap_pts, int* n_pts_out,
                         OracleImuSample* imu, int cap_imu, int* n_imu_out);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-402) and MapPoint::UpdateNormalAndDepth (:433-493), batched:
 * point p owns rows off[p] .. off[p+1) of desc / centers. */
void  map_oracle_distinctive(const uint8_t* desc, const int32_t* off, int n_points, int32_t* best_idx, int32_t* best_median);
void  map_oracle_normal_and_depth(const float* pos, const float* centers, const int32_t* off, const float* ref_center,
                                  const float* level_scale, float last_level_scale, int n_points,
                                  float* normal, float* max_dist, float* min_dist);

/* Optimizer::LocalInertialBA (reference src/Optimizer.cc:2383-2958), flattened.  GROUNDWORK: oracle only, no HIP path yet. */
typedef struct OracleInertialLink {     /* EdgeInertial + EdgeGyroRW + EdgeAccRW between key frames kf1 (earlier) and kf2 */
    int32_t kf1, kf2;
    float dR[9], dV[3], dP[3];          /* IMU::Preintegrated: dR, dV, dP */
    float JRg[9], JVg[9], JVa[9], JPg[9], JPa[9];
    float dT;
    float bias0[6];                     /* the bias the pre-integration was linearised at: bax bay baz bwx bwy bwz */
    double info9[81];                   /* EdgeInertial information (symmetrised, eigenvalue-clamped; x 1e-2 for the link to the fixed key frame) */
    double info_gyro[9], info_acc[9];   /* C.block<3,3>(9,9)^-1, C.block<3,3>(12,12)^-1 */
    uint8_t robust;                     /* Huber kernel on the inertial edge (i == N-1 || bRecInit) */
} OracleInertialLink;
typedef struct OracleInertialProblem {
    int32_t n_kf;
    const double* Rwb;                  /* n_kf x 9, row major */
    const double* twb;                  /* n_kf x 3 */
    const double* vel;                  /* n_kf x 3 */
    const double* bg;                   /* n_kf x 3 */
    const double* ba;                   /* n_kf x 3 */
    const uint8_t* pose_fixed;          /* VertexPose::setFixed */
    const uint8_t* has_imu;             /* velocity / bias vertices exist (pKFi->bImu) */
    const uint8_t* imu_fixed;
    double Rcb[9], tcb[3], tbc[3];      /* mImuCalib */
    double fx, fy, cx, cy, bf;
    int32_t n_points;
    const double* points;
    int32_t n_edges;                    /* EdgeMono / EdgeStereo in addEdge order */
    const int32_t* edge_kf;
    const int32_t* edge_point;
    const double* edge_obs;             /* n_edges x 3 */
    const double* edge_inv_sigma2;
    const uint8_t* edge_stereo;
    int32_t n_links;
    const OracleInertialLink* links;
    double huber_mono, huber_stereo, huber_inertial;     /* (float)sqrt(5.991), (float)sqrt(7.815), sqrt(16.92) */
    double lambda_init;                 /* setUserLambdaInit: 1e0, or 1e-2 when bLarge */
    int32_t max_iters;                  /* opt_it: 10, or 4 when bLarge */
} OracleInertialProblem;
int    inertial_oracle_solve(const OracleInertialProblem* P, double* Rwb_out, double* twb_out, double* vel_out, double* bg_out,
                             double* ba_out, double* points_out, double* chi2_per_edge, uint8_t* depth_positive, OracleLbaStats* stats);
double inertial_oracle_jacobian_check(const OracleInertialProblem* P, int link, double h);

/* Optimizer::PoseInertialOptimizationLastKeyFrame (reference src/Optimizer.cc:4491-4873).  GROUNDWORK: oracle only. */
typedef struct OraclePoseInertialProblem {
    double Rwb[18], twb[6], vel[6], bg[6], ba[6];    /* [0] the last key frame (fixed), [1] the current frame */
    double Rcb[9], tcb[3], tbc[3];
    double fx, fy, cx, cy, bf;
    int32_t n;                          /* features holding a map point, in feature order */
    const double* Xw;                   /* n x 3 */
    const double* obs;                  /* n x 3 */
    const double* inv_sigma2;           /* mvInvLevelSigma2[octave] / uncertainty2 */
    const uint8_t* stereo;
    const uint8_t* close_point;         /* pMP->mTrackDepth < 10 */
    OracleInertialLink link;            /* pFrame->mpImuPreintegrated: kf1 = 0, kf2 = 1 */
    double huber_mono, huber_stereo;
    int32_t rec_init;                   /* bRecInit */
    /* PoseInertialOptimizationLastFrame (:4875-5285): [0] is the PREVIOUS FRAME and is optimised too, tied to pFp->mpcpi */
    int32_t last_frame;
    double prior_Rwb[9], prior_twb[3], prior_vel[3], prior_bg[3], prior_ba[3], prior_H[225];
} OraclePoseInertialProblem;
/* returns nInitialCorrespondences - nBad; H15 = the 15 x 15 Hessian of the new ConstraintPoseImu */
int   pose_inertial_oracle_optimize(const OraclePoseInertialProblem* P, double* Rwb_out, double* twb_out, double* vel_out,
                                    double* bg_out, double* ba_out, uint8_t* outlier, double* H15_out, int* n_bad_out);

/* Frame::UndistortKeyPoints (src/Frame.cc:834-867): K = fx fy cx cy, dist = k1 k2 p1 p2 k3, Knew = mK */
void  edge_oracle_undistort(const OracleKeyPoint* kin, int n, const float K[4], const float dist[5], const float Knew[4], OracleKeyPoint* kout);

#ifdef __cplusplus
}
#endif
#endif
