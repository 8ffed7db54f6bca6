/*
 * ORACLE (test infrastructure, not product) -- CPU restatement of the ORBmatcher
 * searches on the hot path, over flattened (POD) views of Frame/KeyFrame/MapPoint.
 *
 * Follows (read as text; nothing copied):
 *   /root/reference/src/ORBmatcher.cc:35-41     TH_HIGH/TH_LOW/HISTO_LENGTH
 *   /root/reference/src/ORBmatcher.cc:43-221    SearchByProjection(Frame&, vector<MapPoint*>&, ...), RadiusByViewingCos
 *   /root/reference/src/ORBmatcher.cc:223-425   SearchByBoW(KeyFrame*, Frame&, ...)
 *   /root/reference/src/ORBmatcher.cc:765-905   SearchByBoW(KeyFrame*, KeyFrame*, ...)
 *   /root/reference/src/ORBmatcher.cc:1676-1887 SearchByProjection(Frame&, const Frame&, th, bMono)
 *   /root/reference/src/ORBmatcher.cc:2012-2074 ComputeThreeMaxima, DescriptorDistance
 *   /root/reference/src/Frame.cc:472-503,744-822 AssignFeaturesToGrid, GetFeaturesInArea, PosInGrid
 * Conventional cameras (F.Nleft == -1): monocular and rectified stereo / RGB-D (mvuRight gates :92-98, :1751-1757;
 * bForward / bBackward level windows :1692-1693, :1728-1733).  The stereo-fisheye branches (Nleft != -1) are out of scope.
 * PARITY UNPINNED (no reference tests).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

const int TH_HIGH = 100;
const int TH_LOW = 50;
const int HISTO_LENGTH = 30;

// DescriptorDistance (:2058-2074): SWAR popcount over 8 x u32
int hamming256(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t x, y;
        std::memcpy(&x, a + 4 * i, 4);
        std::memcpy(&y, b + 4 * i, 4);
        uint32_t v = x ^ y;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// ComputeThreeMaxima (:2012-2053)
void three_maxima(const int* counts, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = counts[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// rotation bin (:345-350): factor is 1/HISTO_LENGTH (quirk), C round() on a float
int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// lower_bound on the ascending node id array (std::map::lower_bound semantics)
int fv_lower_bound(const OracleFeatVec* fv, int from, uint32_t key)
{
    int lo = from, hi = fv->n_nodes;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (fv->node_id[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct Grid {
    const OracleFrameGrid* g;
    float winv, hinv;
    std::vector<std::vector<int> > cells;   // [col*rows + row], insertion order (Frame.cc:472-503)
    Grid(const OracleFrameGrid* gg) : g(gg)
    {
        winv = (float)g->cols / (g->max_x - g->min_x);     // mfGridElementWidthInv (Frame.cc ctor)
        hinv = (float)g->rows / (g->max_y - g->min_y);
        cells.resize((size_t)g->cols * g->rows);
        for (int i = 0; i < g->n; i++) {
            int px = (int)std::round((g->x[i] - g->min_x) * winv);      // PosInGrid (:812-822)
            int py = (int)std::round((g->y[i] - g->min_y) * hinv);
            if (px < 0 || px >= g->cols || py < 0 || py >= g->rows) continue;
            cells[(size_t)px * g->rows + py].push_back(i);
        }
    }
    // GetFeaturesInArea (:744-810), left image
    void in_area(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const
    {
        out.clear();
        const int nMinCellX = std::max(0, (int)std::floor((x - g->min_x - r) * winv));
        if (nMinCellX >= g->cols) return;
        const int nMaxCellX = std::min(g->cols - 1, (int)std::ceil((x - g->min_x + r) * winv));
        if (nMaxCellX < 0) return;
        const int nMinCellY = std::max(0, (int)std::floor((y - g->min_y - r) * hinv));
        if (nMinCellY >= g->rows) return;
        const int nMaxCellY = std::min(g->rows - 1, (int)std::ceil((y - g->min_y + r) * hinv));
        if (nMaxCellY < 0) return;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<int>& cell = cells[(size_t)ix * g->rows + iy];
                for (size_t j = 0; j < cell.size(); j++) {
                    const int idx = cell[j];
                    if (bCheckLevels) {
                        if (g->octave[idx] < minLevel) continue;
                        if (maxLevel >= 0 && g->octave[idx] > maxLevel) continue;
                    }
                    const float distx = g->x[idx] - x, disty = g->y[idx] - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(idx);
                }
            }
    }
};

}  // namespace

extern "C" {

int orbm_oracle_hamming(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }

void orbm_oracle_three_maxima(const int* counts, int L, int* i1, int* i2, int* i3)
{
    int a = -1, b = -1, c = -1;
    three_maxima(counts, L, a, b, c);
    *i1 = a; *i2 = b; *i3 = c;
}

int orbm_oracle_search_by_bow(const uint8_t* dKF, int nKF, const uint8_t* validKF, const float* angKF, const OracleFeatVec* fvKF,
                              const uint8_t* dF, int nF, const float* angF, const OracleFeatVec* fvF,
                              float nnratio, int checkOri, int32_t* match)
{
    (void)nKF;
    for (int i = 0; i < nF; i++) match[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    int k = 0, f = 0;
    while (k < fvKF->n_nodes && f < fvF->n_nodes) {
        if (fvKF->node_id[k] == fvF->node_id[f]) {
            for (int iKF = fvKF->offset[k]; iKF < fvKF->offset[k + 1]; iKF++) {
                const unsigned realIdxKF = fvKF->feat[iKF];
                if (!validKF[realIdxKF]) continue;          // !pMP || pMP->isBad()
                const uint8_t* d1 = dKF + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int iF = fvF->offset[f]; iF < fvF->offset[f + 1]; iF++) {
                    const unsigned realIdxF = fvF->feat[iF];
                    if (match[realIdxF] >= 0) continue;
                    const int dist = hamming256(d1, dF + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match[bestIdxF] = (int)realIdxKF;
                        if (checkOri) rotHist[rot_bin(angKF[realIdxKF], angF[bestIdxF])].push_back(bestIdxF);
                        nmatches++;
                    }
                }
            }
            k++; f++;
        } else if (fvKF->node_id[k] < fvF->node_id[f]) {
            k = fv_lower_bound(fvKF, k, fvF->node_id[f]);
        } else {
            f = fv_lower_bound(fvF, f, fvKF->node_id[k]);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orbm_oracle_search_by_bow_kfkf(const uint8_t* d1, int n1, const uint8_t* valid1, const float* ang1, const OracleFeatVec* fv1,
                                   const uint8_t* d2, int n2, const uint8_t* valid2, const float* ang2, const OracleFeatVec* fv2,
                                   float nnratio, int checkOri, int32_t* match12)
{
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<uint8_t> matched2(n2, 0);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    int a = 0, b = 0;
    while (a < fv1->n_nodes && b < fv2->n_nodes) {
        if (fv1->node_id[a] == fv2->node_id[b]) {
            for (int i1 = fv1->offset[a]; i1 < fv1->offset[a + 1]; i1++) {
                const int idx1 = (int)fv1->feat[i1];
                if (!valid1[idx1]) continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = fv2->offset[b]; i2 < fv2->offset[b + 1]; i2++) {
                    const int idx2 = (int)fv2->feat[i2];
                    if (matched2[idx2] || !valid2[idx2]) continue;
                    const int dist = hamming256(d1 + (size_t)idx1 * 32, d2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {       // strict here (:848)
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        match12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (checkOri) rotHist[rot_bin(ang1[idx1], ang2[bestIdx2])].push_back(idx1);
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (fv1->node_id[a] < fv2->node_id[b]) {
            a = fv_lower_bound(fv1, a, fv2->node_id[b]);
        } else {
            b = fv_lower_bound(fv2, b, fv1->node_id[a]);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

int orbm_oracle_search_by_projection(const OracleFrameGrid* g, const uint8_t* dF, const float* scale_factors, int nlevels,
                                     int nMP, const uint8_t* in_view, const float* proj_u, const float* proj_v, const float* proj_ur,
                                     const int32_t* pred_level, const float* view_cos, const float* track_depth,
                                     const uint8_t* dMP, const uint8_t* mp_has_obs, const uint8_t* mp_bad,
                                     float th, int bFar, float thFar, float nnratio,
                                     int32_t* assign, uint8_t* occupied)
{
    (void)nlevels;
    Grid grid(g);
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    std::vector<int> idxs;
    for (int iMP = 0; iMP < nMP; iMP++) {
        if (!in_view[iMP]) continue;                                // mbTrackInView (Nleft == -1: mbTrackInViewR is never set)
        if (bFar && track_depth[iMP] > thFar) continue;
        if (mp_bad[iMP]) continue;
        const int lvl = pred_level[iMP];
        float r = (view_cos[iMP] > 0.998) ? 2.5f : 4.0f;             // RadiusByViewingCos (:215-221)
        if (bFactor) r *= th;
        grid.in_area(proj_u[iMP], proj_v[iMP], r * scale_factors[lvl], lvl - 1, lvl, idxs);
        if (idxs.empty()) continue;
        const uint8_t* dmp = dMP + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t c = 0; c < idxs.size(); c++) {
            const int idx = idxs[c];
            if (occupied[idx]) continue;                            // F.mvpMapPoints[idx] && Observations()>0
            if (g->u_right && g->u_right[idx] > 0) {                // F.Nleft == -1 && F.mvuRight[idx]>0 (:92-98)
                const float er = std::fabs(proj_ur[iMP] - g->u_right[idx]);
                if (er > r * scale_factors[lvl]) continue;
            }
            const int dist = hamming256(dmp, dF + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = g->octave[idx];
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = g->octave[idx];
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            assign[bestIdx] = iMP;
            occupied[bestIdx] = mp_has_obs[iMP];
            nmatches++;
        }
    }
    return nmatches;
}

int orbm_oracle_search_by_projection_last(const OracleFrameGrid* g, const uint8_t* dF, const float* angF,
                                          const float* scale_factors, int nlevels,
                                          int nLast, const uint8_t* last_valid, const float* proj_u, const float* proj_v, const float* proj_ur,
                                          const int32_t* last_octave, const float* last_angle,
                                          const uint8_t* dMP, const uint8_t* mp_has_obs,
                                          float th, int level_window, int checkOri,
                                          int32_t* assign, uint8_t* occupied)
{
    const bool bForward = level_window == 1, bBackward = level_window == 2;      // :1692-1693, evaluated by the caller
    (void)nlevels;
    Grid grid(g);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> idxs;
    for (int i = 0; i < nLast; i++) {
        if (!last_valid[i]) continue;
        // The projection x3Dc = Tcw*x3Dw, invzc<0 rejection and Pinhole::project (:1701-1713) are done by
        // the caller with the reference's own float expressions; (u,v) arrive here already rounded to float.
        const float u = proj_u[i], v = proj_v[i];
        if (u < g->min_x || u > g->max_x) continue;
        if (v < g->min_y || v > g->max_y) continue;
        const int oct = last_octave[i];
        const float radius = th * scale_factors[oct];
        if (bForward) grid.in_area(u, v, radius, oct, -1, idxs);                 // :1728-1733
        else if (bBackward) grid.in_area(u, v, radius, 0, oct, idxs);
        else grid.in_area(u, v, radius, oct - 1, oct + 1, idxs);
        if (idxs.empty()) continue;
        const uint8_t* dmp = dMP + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t c = 0; c < idxs.size(); c++) {
            const int i2 = idxs[c];
            if (occupied[i2]) continue;
            if (g->u_right && g->u_right[i2] > 0) {            // CurrentFrame.Nleft == -1 && mvuRight[i2]>0 (:1751-1757)
                const float ur = proj_ur[i];                    // uv(0) - CurrentFrame.mbf*invzc, a float expression of the caller
                const float er = std::fabs(ur - g->u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = hamming256(dmp, dF + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            assign[bestIdx2] = i;
            occupied[bestIdx2] = mp_has_obs[i];
            nmatches++;
            if (checkOri) rotHist[rot_bin(last_angle[i], angF[bestIdx2])].push_back(bestIdx2);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                assign[rotHist[i][j]] = -1;         // CurrentFrame.mvpMapPoints[...] = NULL (:1878)
                occupied[rotHist[i][j]] = 0;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist) (:1889-2010),
// used by Tracking::Relocalization.  The caller flattens pKF->GetMapPointMatches(): valid[i] = pMP && !isBad() &&
// !sAlreadyFound.count(pMP) && dist3D inside [min,max]DistanceInvariance; (u,v) = project(Tcw*x3Dw); pred_level =
// PredictScale; kf_angle[i] = pKF->mvKeysUn[i].angle.  occupied[nF]: CurrentFrame.mvpMapPoints[i2] != NULL (in/out).
int orbm_oracle_search_by_projection_kf(const OracleFrameGrid* g, const uint8_t* dF, const float* angF, const float* scale_factors,
                                        int nPts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                        const int32_t* pred_level, const float* kf_angle, const uint8_t* dMP,
                                        float th, int ORBdist, int checkOri, int32_t* assign, uint8_t* occupied)
{
    Grid grid(g);
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> idxs;
    for (int i = 0; i < nPts; i++) {
        if (!valid[i]) continue;
        const float u = proj_u[i], v = proj_v[i];
        if (u < g->min_x || u > g->max_x) continue;                         // :1917-1920
        if (v < g->min_y || v > g->max_y) continue;
        const int lvl = pred_level[i];
        const float radius = th * scale_factors[lvl];                      // :1938
        grid.in_area(u, v, radius, lvl - 1, lvl + 1, idxs);                  // :1940
        if (idxs.empty()) continue;
        const uint8_t* dmp = dMP + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t c = 0; c < idxs.size(); c++) {
            const int i2 = idxs[c];
            if (occupied[i2]) continue;                                      // :1953
            const int dist = hamming256(dmp, dF + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {                                           // :1967
            assign[bestIdx2] = i;
            occupied[bestIdx2] = 1;
            nmatches++;
            if (checkOri) rotHist[rot_bin(kf_angle[i], angF[bestIdx2])].push_back(bestIdx2);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                assign[rotHist[i][j]] = -1;                                  // :2003
                occupied[rotHist[i][j]] = 0;
                nmatches--;
            }
        }
    }
    return nmatches;
}

// SearchByProjection(KeyFrame* pKF, Sim3f& Scw, vpPoints, vpMatched, th, ratioHamming) (:427-532) and its twin with
// vpPointsKFs / vpMatchedKF (:534-646), loop closing.  The caller does the prelude (:447-487: bad / already found /
// depth / IsInImage / distance invariance / viewing angle / PredictScale); occupied[nKF] = vpMatched[idx] != NULL.
int orbm_oracle_search_by_projection_sim3(const OracleFrameGrid* g, const uint8_t* dKF, const float* scale_factors,
                                          int nPts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                          const int32_t* pred_level, const uint8_t* dMP, int th, float ratioHamming,
                                          int32_t* assign, uint8_t* occupied)
{
    Grid grid(g);
    int nmatches = 0;
    std::vector<int> idxs;
    for (int i = 0; i < nPts; i++) {
        if (!valid[i]) continue;
        const int lvl = pred_level[i];
        const float radius = th * scale_factors[lvl];                      // :489 (int * float)
        grid.in_area(proj_u[i], proj_v[i], radius, -1, -1, idxs);            // KeyFrame::GetFeaturesInArea: no level filter
        if (idxs.empty()) continue;
        const uint8_t* dmp = dMP + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (size_t c = 0; c < idxs.size(); c++) {
            const int idx = idxs[c];
            if (occupied[idx]) continue;                                     // :504
            const int kpLevel = g->octave[idx];
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;                // :509
            const int dist = hamming256(dmp, dKF + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW * ratioHamming) {                             // :522 (int <= float)
            assign[bestIdx] = i;
            occupied[bestIdx] = 1;
            nmatches++;
        }
    }
    return nmatches;
}

// Search core of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th, bRight=false) (:1148-1338, chi2_check=1) and of
// Fuse(KeyFrame*, Sim3f& Scw, vpPoints, th, vpReplacePoint) (:1340-1455, chi2_check=0): for every candidate map point
// the most similar keypoint in the window.  Points are independent (the pointer surgery that follows -- Replace /
// AddObservation / AddMapPoint -- never feeds back into the search), so the result is two arrays: best_idx[i] (or -1)
// and best_dist[i].  The caller does the prelude (:1176-1239) and passes (u, v, ur = u - bf*invz, predicted level).
void orbm_oracle_fuse_search(const OracleFrameGrid* g, const uint8_t* dKF, const float* scale_factors,
                             const float* u_right /* mvuRight */, const float* inv_level_sigma2,
                             int nPts, const uint8_t* valid, const float* proj_u, const float* proj_v, const float* proj_ur,
                             const int32_t* pred_level, const uint8_t* dMP, float th, int chi2_check,
                             int32_t* best_idx, int32_t* best_dist)
{
    Grid grid(g);
    std::vector<int> idxs;
    for (int i = 0; i < nPts; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (!valid[i]) continue;
        const int lvl = pred_level[i];
        const float u = proj_u[i], v = proj_v[i];
        const float radius = th * scale_factors[lvl];                      // :1242
        grid.in_area(u, v, radius, -1, -1, idxs);
        const uint8_t* dmp = dMP + (size_t)i * 32;
        int bestDist = 256, bestIdx = -1;
        for (size_t c = 0; c < idxs.size(); c++) {
            const int idx = idxs[c];
            const int kpLevel = g->octave[idx];
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;                // :1265
            if (chi2_check) {
                const float kpx = g->x[idx], kpy = g->y[idx];
                if (u_right[idx] >= 0) {                                     // stereo keypoint (:1268-1281)
                    const float kpr = u_right[idx];
                    const float ex = u - kpx, ey = v - kpy, er = proj_ur[i] - kpr;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kpx, ey = v - kpy;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = hamming256(dmp, dKF + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
    }
}

// SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize) (:648-763), monocular map
// initialisation.  g2 = grid of F2; prev = vbPrevMatched (x,y per F1 feature).  Returns nmatches, match12[n1].
// (The caller refreshes vbPrevMatched from match12 afterwards, :757-760.)
int orbm_oracle_search_for_initialization(const uint8_t* d1, int n1, const int32_t* octave1, const float* ang1,
                                          const float* prev_x, const float* prev_y,
                                          const OracleFrameGrid* g2, const uint8_t* d2, const float* ang2,
                                          int windowSize, float nnratio, int checkOri, int32_t* match12)
{
    Grid grid(g2);
    const int n2 = g2->n;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    std::vector<int> vMatchedDistance(std::max(n2, 1), INT32_MAX), vnMatches21(std::max(n2, 1), -1), idxs;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = octave1[i1];
        if (level1 > 0) continue;
        grid.in_area(prev_x[i1], prev_y[i1], (float)windowSize, level1, level1, idxs);
        if (idxs.empty()) continue;
        const uint8_t* dd1 = d1 + (size_t)i1 * 32;
        int bestDist = INT32_MAX, bestDist2 = INT32_MAX, bestIdx2 = -1;
        for (size_t c = 0; c < idxs.size(); c++) {
            const int i2 = idxs[c];
            const int dist = hamming256(dd1, d2 + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { match12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                match12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) rotHist[rot_bin(ang1[i1], ang2[bestIdx2])].push_back(i1);
            }
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                const int idx1 = rotHist[i][j];
                if (match12[idx1] >= 0) { match12[idx1] = -1; nmatches--; }
            }
        }
    }
    return nmatches;
}

// SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (:907-1146), conventional
// cameras (mpCamera2 == NULL, NLeft == -1).  has_mp = GetMapPoint(idx) != NULL, stereo = mvuRight[idx] >= 0,
// (ep_x, ep_y) = epipole of KF1's centre in KF2 (:918-920), F12 = K1^-T [t12]x R12 K2^-1 row-major (Pinhole.cpp:109-112),
// sigma2_2 = pKF2->mvLevelSigma2, scale2 = pKF2->mvScaleFactors.  Note that vbMatched2 is never set by the reference loop.
int orbm_oracle_search_for_triangulation(const uint8_t* d1, int n1, const uint8_t* has_mp1, const uint8_t* stereo1,
                                         const float* x1, const float* y1, const float* ang1, const OracleFeatVec* fv1,
                                         const uint8_t* d2, int n2, const uint8_t* has_mp2, const uint8_t* stereo2,
                                         const float* x2, const float* y2, const int32_t* octave2, const float* ang2, const OracleFeatVec* fv2,
                                         float ep_x, float ep_y, const float* F12, const float* sigma2_2, const float* scale2,
                                         int bOnlyStereo, int bCoarse, int checkOri, int32_t* match12)
{
    (void)n2;
    int nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    int a = 0, b = 0;
    while (a < fv1->n_nodes && b < fv2->n_nodes) {
        if (fv1->node_id[a] == fv2->node_id[b]) {
            for (int e1 = fv1->offset[a]; e1 < fv1->offset[a + 1]; e1++) {
                const int idx1 = (int)fv1->feat[e1];
                if (has_mp1[idx1]) continue;
                const bool bStereo1 = stereo1[idx1] != 0;
                if (bOnlyStereo && !bStereo1) continue;
                const uint8_t* dd1 = d1 + (size_t)idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int e2 = fv2->offset[b]; e2 < fv2->offset[b + 1]; e2++) {
                    const int idx2 = (int)fv2->feat[e2];
                    if (has_mp2[idx2]) continue;                       // vbMatched2 stays all-false
                    const bool bStereo2 = stereo2[idx2] != 0;
                    if (bOnlyStereo && !bStereo2) continue;
                    const int dist = hamming256(dd1, d2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ep_x - x2[idx2], distey = ep_y - y2[idx2];
                        if (distex * distex + distey * distey < 100 * scale2[octave2[idx2]]) continue;
                    }
                    bool ok = bCoarse != 0;
                    if (!ok) {                                         // Pinhole::epipolarConstrain (Pinhole.cpp:107-129)
                        const float la = x1[idx1] * F12[0] + y1[idx1] * F12[3] + F12[6];
                        const float lb = x1[idx1] * F12[1] + y1[idx1] * F12[4] + F12[7];
                        const float lc = x1[idx1] * F12[2] + y1[idx1] * F12[5] + F12[8];
                        const float num = la * x2[idx2] + lb * y2[idx2] + lc;
                        const float den = la * la + lb * lb;
                        if (den != 0) {
                            const float dsqr = num * num / den;
                            ok = dsqr < 3.84 * sigma2_2[octave2[idx2]];
                        }
                    }
                    if (ok) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    match12[idx1] = bestIdx2;
                    nmatches++;
                    if (checkOri) rotHist[rot_bin(ang1[idx1], ang2[bestIdx2])].push_back(idx1);
                }
            }
            a++; b++;
        } else if (fv1->node_id[a] < fv2->node_id[b]) {
            a = fv_lower_bound(fv1, a, fv2->node_id[b]);
        } else {
            b = fv_lower_bound(fv2, b, fv1->node_id[a]);
        }
    }
    if (checkOri) {
        int counts[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) counts[i] = (int)rotHist[i].size();
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(counts, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) { match12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

}  // extern "C"
