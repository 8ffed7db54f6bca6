// TEST INFRASTRUCTURE -- drives the reference-typed half of include/orbslam3_shim.hpp on a toy map made of the stand-in
// types (tests/stubs/standin_*.hpp).  Modes:
//   graph <map.txt>   LocalBundleAdjustmentGraph only (no device needed): prints the counters and the flattened problem
//   solve <map.txt>   LocalBundleAdjustmentHIP end to end (needs a HIP device): also prints the written-back map
//   track <case.txt>  ORBmatcherHIP::SearchByProjection(Frame&, vpMapPoints, ...) and (CurrentFrame, LastFrame, th, bMono)
//   gba_graph <map.txt>                              BundleAdjustmentGraph only (no device needed)
//   gba <map.txt> <nLoopKF> <bRobust> <world> <its>  GlobalBundleAdjustemntHIP end to end; world > 1: one host thread per rank on
//                                                    the same device, the all-reduce callback sums through host memory (a stand-in
//                                                    for ncclAllReduce, which needs one GPU per rank)
// The python tests write the inputs, parse the output and compare with what the C ABI gives for the same data.
#define ORBSLAM3_HIP_WITH_REFERENCE
#include "orbslam3_shim.hpp"

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <thread>
#include <fstream>
#include <iostream>
#include <string>

using namespace ORB_SLAM3;

std::mutex MapPoint::mGlobalMutex;

// the reference fallbacks: never reached on the pinhole / Nleft == -1 inputs of these tests
static void fallback(const char* what) { std::fprintf(stderr, "reference fallback called: %s\n", what); std::exit(40); }
ORBmatcher::ORBmatcher(float, bool) {}
int ORBmatcher::SearchByProjection(Frame&, const std::vector<MapPoint*>&, const float, const bool, const float) { fallback("SearchByProjection(F, MPs)"); return 0; }
int ORBmatcher::SearchByProjection(Frame&, const Frame&, const float, const bool) { fallback("SearchByProjection(F, F)"); return 0; }
int ORBmatcher::Fuse(KeyFrame*, const std::vector<MapPoint*>&, const float, const bool) { fallback("Fuse"); return 0; }
int ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, std::vector<std::pair<size_t, size_t> >&, const bool, const bool) { fallback("SearchForTriangulation"); return 0; }
static bool g_reference_lba_called = false;
void Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*, int&, int&, int&, int&) { g_reference_lba_called = true; }
static bool g_reference_gba_called = false;
void Optimizer::BundleAdjustment(const std::vector<KeyFrame*>&, const std::vector<MapPoint*>&, int, bool*, const unsigned long, const bool) { g_reference_gba_called = true; }
void Optimizer::LocalInertialBA(KeyFrame*, bool*, Map*, int&, int&, int&, int&, bool, bool) { fallback("LocalInertialBA"); }
int Optimizer::PoseOptimization(Frame*) { fallback("PoseOptimization"); return 0; }
int Optimizer::PoseInertialOptimizationLastKeyFrame(Frame*, bool) { fallback("PoseInertialOptimizationLastKeyFrame"); return 0; }
int Optimizer::PoseInertialOptimizationLastFrame(Frame*, bool) { fallback("PoseInertialOptimizationLastFrame"); return 0; }
Eigen::MatrixXd Optimizer::Marginalize(const Eigen::MatrixXd& H, const int&, const int&) { fallback("Marginalize"); return H; }

struct ToyMap {
    Map map, otherMap;
    std::deque<KeyFrame> kfs;       // deque: stable addresses, and address order == index order inside one block is NOT guaranteed,
    std::vector<KeyFrame*> kfByAddr; // so the test reads the address order back instead of assuming it
    std::deque<MapPoint> mps;
    Pinhole* cam = nullptr;
    Pinhole* fisheyeLike = nullptr;
    int cur = 0;
};

static void load_map(const char* path, ToyMap& T)
{
    std::ifstream in(path);
    if (!in) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(41); }
    int nkf, npt, nobs, ncov;
    unsigned long initId;
    float fx, fy, cx, cy, bf, invs2[8];
    in >> nkf >> npt >> nobs >> T.cur >> initId >> ncov >> fx >> fy >> cx >> cy >> bf;
    for (float& v : invs2) in >> v;
    T.map.mnInitKFid = initId;
    T.cam = new Pinhole(fx, fy, cx, cy);
    T.kfs.resize(nkf); T.mps.resize(npt);
    for (int i = 0; i < nkf; i++) {
        KeyFrame& k = T.kfs[i];
        int bad, other;
        double q[4], t[3];
        in >> k.mnId >> bad >> other >> q[0] >> q[1] >> q[2] >> q[3] >> t[0] >> t[1] >> t[2];
        k.mbBad = bad; k.mpMap = other ? &T.otherMap : &T.map;
        k.mTcw = Sophus::SE3f(Eigen::Quaternionf((float)q[3], (float)q[0], (float)q[1], (float)q[2]), Eigen::Vector3f((float)t[0], (float)t[1], (float)t[2]));
        k.fx = fx; k.fy = fy; k.cx = cx; k.cy = cy; k.mbf = bf; k.mpCamera = T.cam;
        k.mvInvLevelSigma2.assign(invs2, invs2 + 8);
    }
    for (int c = 0; c < ncov; c++) { int j; in >> j; T.kfs[T.cur].mvpOrderedConnectedKeyFrames.push_back(&T.kfs[j]); }
    for (int i = 0; i < npt; i++) {
        MapPoint& p = T.mps[i];
        int bad, other;
        float x, y, z;
        in >> p.mnId >> bad >> other >> x >> y >> z;
        p.mbBad = bad; p.mpMap = other ? &T.otherMap : &T.map; p.mWorldPos = Eigen::Vector3f(x, y, z);
    }
    for (int e = 0; e < nobs; e++) {
        int ik, ip, oct;
        float u, v, ur;
        in >> ik >> ip >> u >> v >> ur >> oct;
        KeyFrame& k = T.kfs[ik];
        const int feat = (int)k.mvKeysUn.size();
        k.mvKeysUn.push_back(cv::KeyPoint(u, v, 31.f, 0.f, 1.f, oct));
        k.mvuRight.push_back(ur);
        k.mvpMapPoints.push_back(&T.mps[ip]);
        k.N = feat + 1;
        T.mps[ip].mObservations[&k] = std::tuple<int, int>(feat, -1);
        T.mps[ip].nObs++;
    }
    if (!in) { std::fprintf(stderr, "truncated map file\n"); std::exit(42); }
}

static void print_graph(const ToyMap& T, const LbaGraph& g)
{
    auto kfIdx = [&](KeyFrame* k) { for (size_t i = 0; i < T.kfs.size(); i++) if (&T.kfs[i] == k) return (int)i; return -1; };
    auto mpIdx = [&](MapPoint* p) { for (size_t i = 0; i < T.mps.size(); i++) if (&T.mps[i] == p) return (int)i; return -1; };
    std::printf("counters %d %d %d\n", g.num_fixedKF, g.num_OptKF, g.num_edges);
    std::printf("addr_order"); { std::vector<const KeyFrame*> a; for (auto& k : T.kfs) a.push_back(&k); std::sort(a.begin(), a.end()); for (auto* k : a) std::printf(" %d", kfIdx(const_cast<KeyFrame*>(k))); } std::printf("\n");
    std::printf("local"); for (KeyFrame* k : g.lLocalKeyFrames) std::printf(" %d", kfIdx(k)); std::printf("\n");
    std::printf("fixed_cams"); for (KeyFrame* k : g.lFixedCameras) std::printf(" %d", kfIdx(k)); std::printf("\n");
    std::printf("local_points"); for (MapPoint* p : g.lLocalMapPoints) std::printf(" %d", mpIdx(p)); std::printf("\n");
    std::printf("kfs"); for (KeyFrame* k : g.kfs) std::printf(" %d", kfIdx(k)); std::printf("\n");
    std::printf("mps"); for (MapPoint* p : g.mps) std::printf(" %d", mpIdx(p)); std::printf("\n");
    std::printf("pose_fixed"); for (uint8_t f : g.fixed) std::printf(" %d", (int)f); std::printf("\n");
    std::printf("pose_q"); for (double v : g.q) std::printf(" %.17g", v); std::printf("\n");
    std::printf("pose_t"); for (double v : g.t) std::printf(" %.17g", v); std::printf("\n");
    std::printf("points"); for (double v : g.X) std::printf(" %.17g", v); std::printf("\n");
    std::printf("edge_point"); for (int v : g.ePoint) std::printf(" %d", v); std::printf("\n");
    std::printf("edge_pose"); for (int v : g.ePose) std::printf(" %d", v); std::printf("\n");
    std::printf("edge_obs"); for (double v : g.eObs) std::printf(" %.17g", v); std::printf("\n");
    std::printf("edge_w"); for (double v : g.eW) std::printf(" %.17g", v); std::printf("\n");
    std::printf("edge_stereo"); for (uint8_t v : g.eStereo) std::printf(" %d", (int)v); std::printf("\n");
    std::printf("intrinsics %.17g %.17g %.17g %.17g %.17g\n", g.fx, g.fy, g.cx, g.cy, g.bf);
}

static int run_lba(const char* path, bool solve)
{
    ToyMap T;
    load_map(path, T);
    KeyFrame* pKF = &T.kfs[T.cur];
    if (!solve) {
        LbaGraph g;
        const bool ok = LocalBundleAdjustmentGraph(pKF, &T.map, g);
        std::printf("ok %d pinhole %d\n", (int)ok, (int)LbaWindowIsPinhole(pKF));
        print_graph(T, g);
        // a window with a second camera must go to the reference, not be optimised with dropped right-camera edges
        Pinhole second(1, 1, 0, 0);
        T.kfs[T.cur].mpCamera2 = &second;
        int a = -7, b = -7, c = -7, d = -7;
        LocalBundleAdjustmentHIP(pKF, nullptr, &T.map, a, b, c, d);
        std::printf("camera2_fallback %d untouched %d\n", (int)g_reference_lba_called, (int)(a == -7 && b == -7 && c == -7 && d == -7));
        return 0;
    }
    {   // the graph as the solve will see it (on a copy of the markers: the walk writes mnBALocalForKF / mnBAFixedForKF)
        ToyMap T2;
        load_map(path, T2);
        LbaGraph g;
        LocalBundleAdjustmentGraph(&T2.kfs[T2.cur], &T2.map, g);
        print_graph(T2, g);
    }
    int nFixed = -1, nOpt = -1, nMPs = -5, nEdges = -1;
    bool stop = false;
    try { LocalBundleAdjustmentHIP(pKF, &stop, &T.map, nFixed, nOpt, nMPs, nEdges); }
    catch (const orbslam3_hip::Error& e) { std::printf("error %d %s\n", e.code, e.what()); return e.code == ORBX_ERR_NO_DEVICE ? 3 : 4; }
    std::printf("out_params %d %d %d %d\n", nFixed, nOpt, nMPs, nEdges);
    std::printf("map_change %d\n", T.map.mnMapChange);
    for (size_t i = 0; i < T.kfs.size(); i++) {
        const Eigen::Matrix3f R = T.kfs[i].mTcw.rotationMatrix();
        const Eigen::Vector3f t = T.kfs[i].mTcw.translation();
        std::printf("kf %zu %d", i, T.kfs[i].nPoseWrites);
        for (int k = 0; k < 9; k++) std::printf(" %.9g", R[k]);
        for (int k = 0; k < 3; k++) std::printf(" %.9g", t[k]);
        std::printf("\n");
    }
    for (size_t i = 0; i < T.mps.size(); i++)
        std::printf("mp %zu %d %d %d %.9g %.9g %.9g\n", i, T.mps[i].nNormalUpdates, T.mps[i].nErased, T.mps[i].nObs, T.mps[i].mWorldPos[0], T.mps[i].mWorldPos[1], T.mps[i].mWorldPos[2]);
    return 0;
}

// ---- global BA on the toy map ----
// the HIP runtime calls the stand-in collective needs (declared here so that the test needs no HIP headers; libamdhip64 is
// already a dependency of liborbslam3_hip.so)
extern "C" {
int hipStreamSynchronize(void* stream);
int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);
}

struct HostAllReduce {          // all-reduce over `world` threads through host memory: D2H, barrier, sum / max, H2D
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, generation = 0;
    std::vector<std::vector<double> > part;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const int gen = generation;
        if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};
struct RankCtx { HostAllReduce* ar; int rank; long calls = 0; long doubles = 0; };

static int host_allreduce(void* user, double* dev, int64_t count, int op, void* stream)
{
    RankCtx* c = (RankCtx*)user;
    HostAllReduce& A = *c->ar;
    c->calls++; c->doubles += count;
    if (hipStreamSynchronize(stream)) return 1;
    A.part[c->rank].resize((size_t)count);
    if (hipMemcpy(A.part[c->rank].data(), dev, (size_t)count * sizeof(double), 2 /* hipMemcpyDeviceToHost */)) return 1;
    A.barrier();
    std::vector<double> acc(A.part[0]);
    for (int r = 1; r < A.world; r++)
        for (int64_t i = 0; i < count; i++) acc[(size_t)i] = op == LBA_REDUCE_MAX ? std::max(acc[(size_t)i], A.part[r][(size_t)i]) : acc[(size_t)i] + A.part[r][(size_t)i];
    A.barrier();                // everybody has read every part before anybody overwrites its own
    return hipMemcpy(dev, acc.data(), (size_t)count * sizeof(double), 1 /* hipMemcpyHostToDevice */) ? 1 : 0;
}

static void gba_lists(ToyMap& T, std::vector<KeyFrame*>& vpKFs, std::vector<MapPoint*>& vpMP)
{
    // Map::GetAllKeyFrames / GetAllMapPoints: everything of THIS map, bad ones included (the optimiser skips them itself)
    for (auto& k : T.kfs) if (k.mpMap == &T.map) vpKFs.push_back(&k);
    for (auto& p : T.mps) if (p.mpMap == &T.map) vpMP.push_back(&p);
    T.map.mvpAllKeyFrames = vpKFs; T.map.mvpAllMapPoints = vpMP;
    T.map.mpOriginKF = &T.kfs[T.cur];
}

static void print_gba_graph(const ToyMap& T, const std::vector<MapPoint*>& vpMP, const GbaGraph& g)
{
    auto kfIdx = [&](KeyFrame* k) { for (size_t i = 0; i < T.kfs.size(); i++) if (&T.kfs[i] == k) return (int)i; return -1; };
    auto mpIdx = [&](MapPoint* p) { for (size_t i = 0; i < T.mps.size(); i++) if (&T.mps[i] == p) return (int)i; return -1; };
    std::printf("accelerated %d\n", (int)g.accelerated);
    std::printf("addr_order"); { std::vector<const KeyFrame*> a; for (auto& k : T.kfs) a.push_back(&k); std::sort(a.begin(), a.end()); for (auto* k : a) std::printf(" %d", kfIdx(const_cast<KeyFrame*>(k))); } std::printf("\n");
    std::printf("vpmp"); for (MapPoint* p : vpMP) std::printf(" %d", mpIdx(p)); std::printf("\n");
    std::printf("not_included"); for (bool b : g.vbNotIncludedMP) std::printf(" %d", (int)b); std::printf("\n");
    std::printf("kfs"); for (KeyFrame* k : g.kfs) std::printf(" %d", kfIdx(k)); std::printf("\n");
    std::printf("mps"); for (MapPoint* p : g.mps) std::printf(" %d", mpIdx(p)); std::printf("\n");
    std::printf("pose_fixed"); for (uint8_t f : g.fixed) std::printf(" %d", (int)f); std::printf("\n");
    std::printf("pose_q"); for (double v : g.q) std::printf(" %.17g", v); std::printf("\n");
    std::printf("pose_t"); for (double v : g.t) std::printf(" %.17g", v); std::printf("\n");
    std::printf("points"); for (double v : g.X) std::printf(" %.17g", v); std::printf("\n");
    std::printf("edge_point"); for (int v : g.ePoint) std::printf(" %d", v); std::printf("\n");
    std::printf("edge_pose"); for (int v : g.ePose) std::printf(" %d", v); std::printf("\n");
    std::printf("edge_obs"); for (double v : g.eObs) std::printf(" %.17g", v); std::printf("\n");
    std::printf("edge_w"); for (double v : g.eW) std::printf(" %.17g", v); std::printf("\n");
    std::printf("edge_stereo"); for (uint8_t v : g.eStereo) std::printf(" %d", (int)v); std::printf("\n");
    std::printf("intrinsics %.17g %.17g %.17g %.17g %.17g\n", g.fx, g.fy, g.cx, g.cy, g.bf);
    std::printf("max_kf_id %lu\n", g.maxKFid);
}

static int run_gba(const char* path, bool solve, unsigned long nLoopKF, bool bRobust, int world, int its)
{
    ToyMap T;
    load_map(path, T);
    std::vector<KeyFrame*> vpKFs;
    std::vector<MapPoint*> vpMP;
    gba_lists(T, vpKFs, vpMP);
    {
        GbaGraph g;
        BundleAdjustmentGraph(vpKFs, vpMP, g);
        print_gba_graph(T, vpMP, g);
    }
    if (!solve) {
        // a key frame with a second camera sends the whole call to the reference
        Pinhole second(1, 1, 0, 0);
        T.kfs[T.cur].mpCamera2 = &second;
        BundleAdjustmentHIP(vpKFs, vpMP, 5, nullptr, 0, true);
        std::printf("camera2_fallback %d\n", (int)g_reference_gba_called);
        return 0;
    }
    std::printf("origin_id %lu\n", T.map.GetOriginKF()->mnId);
    bool stop = false;
    int err = 0;
    HostAllReduce ar;
    ar.world = world; ar.part.resize((size_t)world);
    std::vector<RankCtx> ctx((size_t)world);
    auto body = [&](int rank) {
        GbaSharding sh;
        sh.rank = rank; sh.world = world; sh.device = 0; sh.allreduce = host_allreduce; sh.user = &ctx[(size_t)rank];
        ctx[(size_t)rank].ar = &ar; ctx[(size_t)rank].rank = rank;
        try { GlobalBundleAdjustemntHIP(&T.map, its, &stop, nLoopKF, bRobust, world > 1 ? &sh : nullptr); }
        catch (const orbslam3_hip::Error& e) { std::printf("error %d %s\n", e.code, e.what()); err = e.code == ORBX_ERR_NO_DEVICE ? 3 : 4; }
    };
    if (world == 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < world; r++) th.emplace_back(body, r);
        for (auto& t : th) t.join();
        std::printf("allreduce_calls %ld doubles %ld\n", ctx[0].calls, ctx[0].doubles);
    }
    if (err) return err;
    for (size_t i = 0; i < T.kfs.size(); i++) {
        const KeyFrame& k = T.kfs[i];
        std::printf("kf %zu %d %lu", i, k.nPoseWrites, k.mnBAGlobalForKF);
        const Eigen::Matrix3f R = k.mTcw.rotationMatrix(), Rg = k.mTcwGBA.rotationMatrix();
        const Eigen::Vector3f t = k.mTcw.translation(), tg = k.mTcwGBA.translation();
        for (int j = 0; j < 9; j++) std::printf(" %.9g", R[j]);
        for (int j = 0; j < 3; j++) std::printf(" %.9g", t[j]);
        for (int j = 0; j < 9; j++) std::printf(" %.9g", Rg[j]);
        for (int j = 0; j < 3; j++) std::printf(" %.9g", tg[j]);
        std::printf("\n");
    }
    for (size_t i = 0; i < T.mps.size(); i++) {
        const MapPoint& p = T.mps[i];
        std::printf("mp %zu %d %lu %.9g %.9g %.9g %.9g %.9g %.9g\n", i, p.nNormalUpdates, p.mnBAGlobalForKF, p.mWorldPos[0], p.mWorldPos[1], p.mWorldPos[2],
                    p.mPosGBA[0], p.mPosGBA[1], p.mPosGBA[2]);
    }
    return 0;
}

// track <case.txt>: one frame + one set of map points / last-frame points; prints nmatches and the frame's mvpMapPoints
static int run_track(const char* path)
{
    std::ifstream in(path);
    if (!in) return 41;
    int mode, n, npts, bMono, nlev;
    float th, thFar, nnratio, mb, mbf, minx, miny, maxx, maxy;
    int bFar, checkOri;
    in >> mode >> n >> npts >> th >> bFar >> thFar >> nnratio >> checkOri >> bMono >> mb >> mbf >> minx >> miny >> maxx >> maxy >> nlev;
    Frame F;
    F.N = n; F.mnMinX = minx; F.mnMinY = miny; F.mnMaxX = maxx; F.mnMaxY = maxy; F.mb = mb; F.mbf = mbf;
    F.mvScaleFactors.resize(nlev);
    for (float& s : F.mvScaleFactors) in >> s;
    F.mvKeys.resize(n); F.mvKeysUn.resize(n); F.mvuRight.resize(n); F.mvpMapPoints.assign(n, nullptr); F.mvbOutlier.assign(n, false);
    F.mDescriptors.create(n, 32, CV_8U);
    std::deque<MapPoint> holders;       // map points already held by the frame (occupied features)
    for (int i = 0; i < n; i++) {
        float x, y, a, ur; int oct, occ;
        in >> x >> y >> oct >> a >> ur >> occ;
        F.mvKeysUn[i] = cv::KeyPoint(x, y, 31.f, a, 1.f, oct); F.mvKeys[i] = F.mvKeysUn[i]; F.mvuRight[i] = ur;
        for (int b = 0; b < 32; b++) { int v; in >> v; F.mDescriptors.ptr<uint8_t>(i)[b] = (uint8_t)v; }
        if (occ) { holders.emplace_back(); holders.back().nObs = 1; F.mvpMapPoints[i] = &holders.back(); }
    }
    std::deque<MapPoint> pts(npts);
    std::vector<MapPoint*> vp(npts);
    Frame Last;
    Pinhole cam(1.f, 1.f, 0.f, 0.f);     // fx = fy = 1, cx = cy = 0: project(X) = (x/z, y/z); with z = 1 the test dictates (u, v) directly
    F.mpCamera = &cam; Last.mpCamera = &cam;
    Last.N = npts; Last.mvKeys.resize(npts); Last.mvKeysUn.resize(npts); Last.mvpMapPoints.assign(npts, nullptr); Last.mvbOutlier.assign(npts, false);
    float tz = 0.f;
    for (int i = 0; i < npts; i++) {
        MapPoint& p = pts[i];
        int valid, lvl, hasObs, bad; float u, v, ur, vc, depth, ang;
        in >> valid >> u >> v >> ur >> lvl >> vc >> depth >> ang >> hasObs >> bad;
        p.mDescriptor.create(1, 32, CV_8U);
        for (int b = 0; b < 32; b++) { int q; in >> q; p.mDescriptor.data[b] = (uint8_t)q; }
        p.mbTrackInView = valid; p.mTrackProjX = u; p.mTrackProjY = v; p.mTrackProjXR = ur; p.mnTrackScaleLevel = lvl; p.mTrackViewCos = vc;
        p.mTrackDepth = depth; p.nObs = hasObs; p.mbBad = bad;
        vp[i] = &p;
        // last-frame form: the point sits at (u, v, 1) in the current camera frame (Tcw = identity), so that
        // project() returns (u, v), invzc = 1 and ur = u - mbf (the test generated ur = u - mbf for this mode)
        p.mWorldPos = Eigen::Vector3f(u, v, 1.f);
        Last.mvKeys[i] = cv::KeyPoint(0, 0, 31.f, ang, 1.f, lvl); Last.mvKeysUn[i] = Last.mvKeys[i];
        if (valid) Last.mvpMapPoints[i] = &p;
    }
    in >> tz;       // last-frame mode: z of the current camera centre in the last frame (drives bForward / bBackward)
    if (!in) return 42;
    int nm;
    try {
        ORBmatcherHIP matcher(nnratio, checkOri != 0);
        if (mode == 0) nm = matcher.SearchByProjection(F, vp, th, bFar != 0, thFar);
        else {
            Eigen::Matrix3f I; I(0, 0) = I(1, 1) = I(2, 2) = 1.f;
            Last.mTcw = Sophus::SE3f(I, Eigen::Vector3f(0.f, 0.f, tz));     // twc = 0 => tlc = Tlw * 0 = (0, 0, tz)
            nm = matcher.SearchByProjection(F, Last, th, bMono != 0);
        }
    } catch (const orbslam3_hip::Error& e) { std::printf("error %d %s\n", e.code, e.what()); return e.code == ORBX_ERR_NO_DEVICE ? 3 : 4; }
    std::printf("nmatches %d\nassign", nm);
    for (int i = 0; i < n; i++) {
        int a = -1;
        for (int k = 0; k < npts; k++) if (F.mvpMapPoints[i] == &pts[k]) a = k;
        if (a < 0 && F.mvpMapPoints[i]) a = -2;      // still holds its earlier map point
        std::printf(" %d", a);
    }
    std::printf("\n");
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    if (mode == "graph") return run_lba(argv[2], false);
    if (mode == "solve") return run_lba(argv[2], true);
    if (mode == "track") return run_track(argv[2]);
    if (mode == "gba_graph") return run_gba(argv[2], false, 0, true, 1, 5);
    if (mode == "gba" && argc >= 7) return run_gba(argv[2], true, std::strtoul(argv[3], nullptr, 10), std::atoi(argv[4]) != 0, std::atoi(argv[5]), std::atoi(argv[6]));
    return 2;
}
