// frontend_demo.cpp -- the tracking front end through the C ABI from plain C++ (no Python, no torch): two synthetic frames
// -> ORB extraction -> DBoW2 transform -> SearchByBoW -> PoseOptimization on synthetic correspondences.
// Build:  g++ -std=c++17 -Iinclude examples/frontend_demo.cpp -Lorb_slam3-1_amd -lorbslam3_hip -Wl,-rpath,$PWD/orb_slam3-1_amd -o frontend_demo
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbslam3_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        const int rc_ = (call);                                                            \
        if (rc_ < 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, orbx_last_error()); return 1; } \
    } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main()
{
    if (orbx_device_count() < 1) { std::printf("no HIP device: %s\n", "the library has no CPU fallback"); return 77; }
    const int W = 640, H = 480;
    // a blocky random texture (corners at every block boundary) and the same texture moved by 4 px
    std::vector<uint8_t> a((size_t)W * H), b((size_t)W * H);
    std::vector<uint8_t> blocks((W / 8 + 2) * (H / 8 + 2));
    for (auto& v : blocks) v = (uint8_t)(rnd() & 0xFF);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            a[(size_t)y * W + x] = blocks[(y / 8) * (W / 8 + 2) + x / 8];
            const int xs = x + 4;
            b[(size_t)y * W + x] = blocks[(y / 8) * (W / 8 + 2) + xs / 8];
        }
    orbx_extractor* ex = nullptr;
    CHECK(orbx_create(1000, 1.2f, 8, 20, 7, 0, &ex));
    const int cap = orbx_max_keypoints(ex);
    std::vector<OrbxKeyPoint> kA(cap), kB(cap);
    std::vector<uint8_t> dA((size_t)cap * 32), dB((size_t)cap * 32);
    int nA = 0, nB = 0, mono = 0;
    CHECK(orbx_extract(ex, a.data(), W, H, W, 0, 1000, kA.data(), dA.data(), cap, &nA, &mono));
    CHECK(orbx_extract(ex, b.data(), W, H, W, 0, 1000, kB.data(), dB.data(), cap, &nB, &mono));
    std::printf("extracted %d and %d key points\n", nA, nB);

    // a small synthetic vocabulary: complete 10-ary tree of depth 3 with random centroids
    const int k = 10, L = 3;
    int n_nodes = 1, level_n = 1;
    for (int l = 0; l < L; l++) { level_n *= k; n_nodes += level_n; }
    const int n_inner = n_nodes - level_n;
    std::vector<int32_t> off(n_nodes + 1), word(n_nodes, -1);
    std::vector<uint32_t> child(n_nodes - 1);
    std::vector<uint8_t> cd((size_t)n_nodes * 32);
    std::vector<double> wt(n_nodes, 0.0);
    for (int i = 0; i <= n_nodes; i++) off[i] = (i < n_inner ? i : n_inner) * k;
    for (int i = 1; i < n_nodes; i++) child[i - 1] = (uint32_t)i;
    for (auto& v : cd) v = (uint8_t)(rnd() & 0xFF);
    for (int i = n_inner; i < n_nodes; i++) { word[i] = i - n_inner; wt[i] = 1.0 + (rnd() % 100) * 0.05; }
    OrbvVocabulary voc = {n_nodes, L, off.data(), child.data(), cd.data(), wt.data(), word.data()};
    orbv_vocab* vv = nullptr;
    CHECK(orbv_create(0, &voc, &vv));
    auto transform = [&](const std::vector<uint8_t>& d, int n, std::vector<uint32_t>& node, std::vector<int32_t>& o, std::vector<uint32_t>& feat, int32_t& nn) {
        std::vector<uint32_t> bi(n + 1); std::vector<double> bv(n + 1);
        node.assign(n + 1, 0); o.assign(n + 2, 0); feat.assign(n + 1, 0);
        int32_t nb = 0;
        return orbv_transform(vv, d.data(), n, 1, bi.data(), bv.data(), &nb, node.data(), o.data(), feat.data(), &nn);
    };
    std::vector<uint32_t> nodeA, nodeB, featA, featB;
    std::vector<int32_t> offA, offB;
    int32_t nnA = 0, nnB = 0;
    CHECK(transform(dA, nA, nodeA, offA, featA, nnA));
    CHECK(transform(dB, nB, nodeB, offB, featB, nnB));

    orbm_matcher* m = nullptr;
    CHECK(orbm_create(0, &m));
    std::vector<uint8_t> valid(nA, 1);
    std::vector<float> angA(nA), angB(nB);
    for (int i = 0; i < nA; i++) angA[i] = kA[i].angle;
    for (int i = 0; i < nB; i++) angB[i] = kB[i].angle;
    OrbmFeatVec fvA = {nnA, nodeA.data(), offA.data(), featA.data()}, fvB = {nnB, nodeB.data(), offB.data(), featB.data()};
    std::vector<int32_t> match(nB > 0 ? nB : 1, -1);
    const int nm = orbm_search_by_bow(m, dA.data(), nA, valid.data(), angA.data(), &fvA, dB.data(), nB, angB.data(), &fvB, 0.7f, 1, match.data());
    CHECK(nm);
    int consistent = 0;
    for (int f = 0; f < nB; f++)
        if (match[f] >= 0 && std::fabs((kA[match[f]].x - kB[f].x) - 4.0f) < 2.5f && std::fabs(kA[match[f]].y - kB[f].y) < 2.5f) consistent++;
    std::printf("SearchByBoW: %d matches, %d of them displaced by the true 4 px\n", nm, consistent);

    // motion-only BA on exact synthetic correspondences: the pose must come back to the truth
    const int ne = 200;
    std::vector<double> Xw(3 * ne), obs(3 * ne), w(ne, 1.0);
    std::vector<uint8_t> st(ne, 0), outl(ne);
    const double fx = 458.654, fy = 457.296, cx = 367.215, cy = 248.375, tx = 0.03, ty = -0.02, tz = 0.05;
    for (int i = 0; i < ne; i++) {
        const double X = (rnd() % 8000) / 1000.0 - 4.0, Y = (rnd() % 5000) / 1000.0 - 2.5, Z = 3.0 + (rnd() % 10000) / 1000.0;
        Xw[3 * i] = X; Xw[3 * i + 1] = Y; Xw[3 * i + 2] = Z;
        obs[3 * i] = fx * (X + tx) / (Z + tz) + cx; obs[3 * i + 1] = fy * (Y + ty) / (Z + tz) + cy; obs[3 * i + 2] = -1;
    }
    PoseProblem pr = {{0, 0, 0, 1}, {0, 0, 0}, ne, Xw.data(), obs.data(), w.data(), st.data(), fx, fy, cx, cy, 47.9, std::sqrt(5.991), std::sqrt(7.815)};
    pose_solver* ps = nullptr;
    CHECK(pose_create(0, &ps));
    PoseResult res;
    CHECK(pose_optimize(ps, &pr, &res, outl.data()));
    std::printf("PoseOptimization: %d inliers, t = (%.4f %.4f %.4f), truth (%.4f %.4f %.4f)\n", res.inliers, res.t[0], res.t[1], res.t[2], tx, ty, tz);
    const bool ok = nA > 300 && nB > 300 && nm > 50 && consistent > nm / 2 && res.inliers == ne &&
                    std::fabs(res.t[0] - tx) < 1e-6 && std::fabs(res.t[1] - ty) < 1e-6 && std::fabs(res.t[2] - tz) < 1e-6;
    pose_destroy(ps); orbm_destroy(m); orbv_destroy(vv); orbx_destroy(ex);
    std::printf(ok ? "frontend demo OK\n" : "frontend demo FAILED\n");
    return ok ? 0 : 2;
}
