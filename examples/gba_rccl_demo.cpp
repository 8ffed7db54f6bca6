// gba_rccl_demo.cpp -- the landmark-sharded global bundle adjustment driven from plain C++ with RCCL:
// one host thread per GPU, one lba_shard per thread, lba_shard_optimize() with an all-reduce callback that is a single
// ncclAllReduce on the stream the library hands over.  This is the shape of the change in
// LoopClosing::RunGlobalBundleAdjustment (reference src/LoopClosing.cc:2272-2288) for a server with several GPUs; the
// reference-typed form is GlobalBundleAdjustemntHIP(pMap, nIterations, pbStopFlag, nLoopKF, bRobust, &sharding) in
// include/orbslam3_shim.hpp.
//
//   gba_rccl_demo [n_gpus] [n_poses] [n_points]      (defaults: every visible GPU, 60 poses, 3000 points)
// Build:  g++ -std=c++17 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/gba_rccl_demo.cpp -Lorb_slam3-1_amd -lorbslam3_hip
//             -L/opt/rocm/lib -lrccl -lamdhip64 -lpthread -Wl,-rpath,$PWD/orb_slam3-1_amd -Wl,-rpath,/opt/rocm/lib -o gba_rccl_demo
// It solves the same synthetic map once on GPU 0 alone (lba_solve) and once sharded, and checks that both walked the same
// Levenberg path to the same estimates.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "orbslam3_hip.h"

static uint64_t rng_state = 88172645463325252ull;
static double rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }
static double gauss() { return std::sqrt(-2.0 * std::log(rnd() + 1e-300)) * std::cos(6.283185307179586 * rnd()); }

struct Problem {
    std::vector<double> q, t, X, obs, w;
    std::vector<uint8_t> fixed, stereo;
    std::vector<int32_t> ep, ek;
    LbaProblem view(int lo, int hi, std::vector<int32_t>& ep_l, std::vector<int32_t>& ek_l, std::vector<double>& obs_l, std::vector<double>& w_l, std::vector<uint8_t>& st_l) const
    {
        for (size_t e = 0; e < ep.size(); e++)
            if (ep[e] >= lo && ep[e] < hi) {
                ep_l.push_back(ep[e] - lo); ek_l.push_back(ek[e]);
                obs_l.insert(obs_l.end(), obs.begin() + 3 * e, obs.begin() + 3 * e + 3);
                w_l.push_back(w[e]); st_l.push_back(stereo[e]);
            }
        LbaProblem p;
        p.n_poses = (int)fixed.size(); p.pose_q = q.data(); p.pose_t = t.data(); p.pose_fixed = fixed.data();
        p.n_points = hi - lo; p.points = X.data() + 3 * (size_t)lo;
        p.n_edges = (int)ep_l.size(); p.edge_point = ep_l.data(); p.edge_pose = ek_l.data(); p.edge_obs = obs_l.data();
        p.edge_inv_sigma2 = w_l.data(); p.edge_stereo = st_l.data();
        p.fx = 458.654; p.fy = 457.296; p.cx = 367.215; p.cy = 248.375; p.bf = 47.9;
        p.huber_mono = 0.0; p.huber_stereo = 0.0;       // loop closing calls the global BA with bRobust = false (LoopClosing.cc:2288)
        return p;
    }
};

// cameras on a line looking along +z at a cloud 4..12 m away; identity rotations keep the generator short
static Problem make_problem(int n_poses, int n_points)
{
    Problem P;
    const double fx = 458.654, fy = 457.296, cx = 367.215, cy = 248.375;
    for (int i = 0; i < n_poses; i++) {
        P.q.insert(P.q.end(), {0.0, 0.0, 0.0, 1.0});
        const double cxw = 0.08 * i - 0.04 * n_poses;           // camera centre; t = -R c
        P.t.insert(P.t.end(), {-cxw + 0.01 * gauss(), 0.01 * gauss(), 0.01 * gauss()});
        P.fixed.push_back(i == 0);
    }
    for (int l = 0; l < n_points; l++) {
        const double X[3] = {6.0 * rnd() - 3.0, 3.0 * rnd() - 1.5, 4.0 + 8.0 * rnd()};
        const size_t first = P.ep.size();
        for (int k = 0; k < 10; k++) {
            const int i = (int)(rnd() * n_poses) % n_poses;
            bool dup = false;
            for (size_t e = first; e < P.ep.size(); e++) dup |= P.ek[e] == i;
            if (dup) continue;
            const double cxw = 0.08 * i - 0.04 * n_poses;
            const double xc = X[0] - cxw, yc = X[1], zc = X[2];
            const double u = fx * xc / zc + cx + 0.5 * gauss(), v = fy * yc / zc + cy + 0.5 * gauss();
            if (u < 0 || u >= 752 || v < 0 || v >= 480) continue;
            P.ep.push_back(l); P.ek.push_back(i);
            P.obs.insert(P.obs.end(), {(double)(float)u, (double)(float)v, -1.0});
            P.w.push_back(1.0); P.stereo.push_back(0);
        }
        if (P.ep.size() - first < 2) {      // a point needs two views: give it the first two cameras without the image check
            P.ep.resize(first); P.ek.resize(first); P.obs.resize(3 * first); P.w.resize(first); P.stereo.resize(first);
            for (int i = 0; i < 2; i++) {
                const double cxw = 0.08 * i - 0.04 * n_poses;
                P.ep.push_back(l); P.ek.push_back(i);
                P.obs.insert(P.obs.end(), {(double)(float)(fx * (X[0] - cxw) / X[2] + cx), (double)(float)(fy * X[1] / X[2] + cy), -1.0});
                P.w.push_back(1.0); P.stereo.push_back(0);
            }
        }
        P.X.insert(P.X.end(), {X[0] + 0.03 * gauss(), X[1] + 0.03 * gauss(), X[2] + 0.03 * gauss()});
    }
    return P;
}

// THE callback: the whole exchange step of the path
static int rccl_allreduce(void* user, double* buf, int64_t count, int op, void* stream)
{
    return ncclAllReduce(buf, buf, (size_t)count, ncclDouble, op == LBA_REDUCE_MAX ? ncclMax : ncclSum, *(ncclComm_t*)user, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

int main(int argc, char** argv)
{
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) { std::printf("no HIP device: the library has no CPU fallback\n"); return 77; }
    const int world = argc > 1 ? std::min(std::atoi(argv[1]), n_dev) : n_dev;
    const int n_poses = argc > 2 ? std::atoi(argv[2]) : 60, n_points = argc > 3 ? std::atoi(argv[3]) : 3000;
    const Problem P = make_problem(n_poses, n_points);

    // single GPU: lba_solve
    std::vector<double> q1(P.q.size()), t1(P.t.size()), X1(P.X.size());
    LbaStats s1;
    {
        std::vector<int32_t> ep, ek; std::vector<double> ob, w; std::vector<uint8_t> st;
        const LbaProblem pr = P.view(0, n_points, ep, ek, ob, w, st);
        lba_solver* sv = nullptr;
        if (lba_create(0, &sv) < 0 || lba_solve(sv, &pr, nullptr, 5, 0.0, q1.data(), t1.data(), X1.data(), nullptr, nullptr, &s1) < 0) {
            std::fprintf(stderr, "lba_solve failed: %s\n", orbx_last_error());
            return 1;
        }
        lba_destroy(sv);
    }

    // sharded: one thread per GPU, RCCL communicator over all of them
    std::vector<int> devs(world);
    for (int r = 0; r < world; r++) devs[r] = r;
    std::vector<ncclComm_t> comms(world);
    if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) { std::fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    std::vector<LbaStats> st(world);
    std::vector<std::vector<double> > qs(world), ts(world), Xs(world);
    std::vector<int> rc(world, 0), lo(world), hi(world);
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) {
        lo[r] = (int)((long long)n_points * r / world); hi[r] = (int)((long long)n_points * (r + 1) / world);
        th.emplace_back([&, r] {
            std::vector<int32_t> ep, ek; std::vector<double> ob, w; std::vector<uint8_t> stf;
            const LbaProblem pr = P.view(lo[r], hi[r], ep, ek, ob, w, stf);
            lba_shard* sh = nullptr;
            qs[r].resize(P.q.size()); ts[r].resize(P.t.size()); Xs[r].resize(3 * (size_t)(hi[r] - lo[r]));
            if ((rc[r] = lba_shard_create(devs[r], &pr, &sh)) < 0) return;
            rc[r] = lba_shard_optimize(sh, rccl_allreduce, &comms[r], world, 5, 0.0, nullptr, &st[r]);
            if (rc[r] == 0) rc[r] = lba_shard_download(sh, qs[r].data(), ts[r].data(), Xs[r].data(), nullptr, nullptr);
            lba_shard_destroy(sh);
        });
    }
    for (auto& t : th) t.join();
    for (int r = 0; r < world; r++) {
        if (rc[r] < 0) { std::fprintf(stderr, "rank %d failed (%d): %s\n", r, rc[r], orbx_last_error()); return 1; }
        ncclCommDestroy(comms[r]);
    }
    double dpose = 0, dpt = 0, upd = 0;
    for (size_t i = 0; i < t1.size(); i++) dpose = std::max(dpose, std::fabs(ts[0][i] - t1[i]));
    for (int r = 0; r < world; r++)
        for (size_t i = 0; i < Xs[r].size(); i++) dpt = std::max(dpt, std::fabs(Xs[r][i] - X1[3 * (size_t)lo[r] + i]));
    for (size_t i = 0; i < X1.size(); i++) upd = std::max(upd, std::fabs(X1[i] - P.X[i]));
    std::printf("%d poses, %d points, %zu edges; %d GPU(s)\n", n_poses, n_points, P.ep.size(), world);
    std::printf("single : %d iterations, %d trials, chi2 %.6f -> %.6f\n", s1.iterations, s1.trials, s1.chi2_initial, s1.chi2_final);
    std::printf("sharded: %d iterations, %d trials, chi2 %.6f -> %.6f   max |pose t diff| %.3g, max |point diff| %.3g (largest update %.3g)\n",
                st[0].iterations, st[0].trials, st[0].chi2_initial, st[0].chi2_final, dpose, dpt, upd);
    const bool ok = st[0].iterations == s1.iterations && st[0].trials == s1.trials && std::fabs(st[0].chi2_final - s1.chi2_final) <= 1e-6 * s1.chi2_final &&
                    dpose <= 1e-4 * upd && dpt <= 1e-4 * upd && s1.chi2_final < 0.5 * s1.chi2_initial;
    std::printf(ok ? "gba rccl demo OK\n" : "gba rccl demo MISMATCH\n");
    return ok ? 0 : 1;
}
