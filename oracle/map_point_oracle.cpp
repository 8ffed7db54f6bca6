/*
 * map_point_oracle.cpp -- CPU ORACLE (test infrastructure, not the product) for the per-map-point upkeep the reference
 * runs after Fuse / triangulation / BA on every touched map point:
 *   map_oracle_distinctive      follows MapPoint::ComputeDistinctiveDescriptors   src/MapPoint.cc:329-402
 *                               (distance matrix :371-381, sort + median :388-390, strict first minimum :391-395)
 *   map_oracle_normal_and_depth follows MapPoint::UpdateNormalAndDepth            src/MapPoint.cc:433-493
 * The container walks (observations -> descriptor rows / camera centres) are the host shim's; the inputs here are what
 * those walks produce.  PARITY UNPINNED (no reference fixture).  The float part assumes Eigen's fixed-size reduction
 * order for Vector3f::norm(), x0*x0 + (x1*x1 + x2*x2) (redux_novec_unroller splits 3 as 1 + 2), true division for
 * vector / scalar, and no FMA contraction -- a -march=native build of the reference may contract.
 */
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {
int hamming256(const uint8_t* a, const uint8_t* b)          /* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2058-2074 */
{
    uint32_t x[8], y[8];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t v = x[i] ^ y[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}
}  // namespace

extern "C" {

void map_oracle_distinctive(const uint8_t* desc, const int32_t* off, int n_points, int32_t* best_idx, int32_t* best_median)
{
    for (int p = 0; p < n_points; p++) {
        const uint8_t* d = desc + (size_t)off[p] * 32;
        const size_t N = (size_t)(off[p + 1] - off[p]);
        if (N == 0) { best_idx[p] = -1; if (best_median) best_median[p] = -1; continue; }      /* :365-366 */
        std::vector<float> D(N * N);
        for (size_t i = 0; i < N; i++) {
            D[i * N + i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                const int dij = hamming256(d + i * 32, d + j * 32);
                D[i * N + j] = (float)dij;
                D[j * N + i] = (float)dij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> v(D.begin() + i * N, D.begin() + (i + 1) * N);
            std::sort(v.begin(), v.end());
            const int median = v[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
        }
        best_idx[p] = BestIdx;
        if (best_median) best_median[p] = BestMedian;
    }
}

void map_oracle_normal_and_depth(const float* pos, const float* centers, const int32_t* off, const float* ref_center,
                                 const float* level_scale, float last_level_scale, int n_points,
                                 float* normal, float* max_dist, float* min_dist)
{
    for (int p = 0; p < n_points; p++) {
        const float* P = pos + 3 * p;
        float n[3] = {0.f, 0.f, 0.f};
        int cnt = 0;
        for (int o = off[p]; o < off[p + 1]; o++) {                       /* :448-468 */
            const float d[3] = {P[0] - centers[3 * o], P[1] - centers[3 * o + 1], P[2] - centers[3 * o + 2]};
            const float nrm = std::sqrt(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2]));
            for (int k = 0; k < 3; k++) n[k] = n[k] + d[k] / nrm;
            cnt++;
        }
        const float c[3] = {P[0] - ref_center[3 * p], P[1] - ref_center[3 * p + 1], P[2] - ref_center[3 * p + 2]};
        const float dist = std::sqrt(c[0] * c[0] + (c[1] * c[1] + c[2] * c[2]));           /* :470-471 */
        max_dist[p] = dist * level_scale[p];                              /* :489 */
        min_dist[p] = max_dist[p] / last_level_scale;                     /* :490 */
        for (int k = 0; k < 3; k++) normal[3 * p + k] = n[k] / (float)cnt;                  /* :491 */
    }
}

}  // extern "C"
