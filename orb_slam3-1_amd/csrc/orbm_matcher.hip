// orbm_matcher.hip -- gfx950 kernels + C ABI for the ORBmatcher searches on the hot path
// (reference src/ORBmatcher.cc: SearchByBoW :223-425 / :765-905, SearchByProjection :43-213 / :1676-1887,
// ComputeThreeMaxima :2012-2053, DescriptorDistance :2058-2074).
//
// Hamming distance = 4 x popcount(u64 xor) per 256-bit descriptor pair.  The greedy "already matched"
// dependences of the reference are kept exactly:
//   * SearchByBoW: a Frame feature belongs to exactly one vocabulary node (DBoW2::FeatureVector::addFeature),
//     so nodes are independent work items: one lane per common node runs the reference's loops verbatim.
//     If a caller passes feature vectors that violate the invariant the kernel runs the nodes serially.
//   * SearchByProjection: later map points see earlier assignments (F.mvpMapPoints), so points are processed
//     in order by one wave; the candidate window of a point (GetFeaturesInArea) is searched by 64 lanes and
//     reduced with a lexicographic (distance, candidate order) top-2, which reproduces the sequential
//     best / second-best bookkeeping including ties.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orbslam3_hip.h"

namespace orbx {
extern thread_local std::string g_last_error;
int fail(int code, const char* fmt, ...);
}
using orbx::fail;

#define ORBM_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ORBX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace orbm {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;    // src/ORBmatcher.cc:35-37

__device__ __forceinline__ int hamming256(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b)
{
    const unsigned long long* pa = (const unsigned long long*)a;
    const unsigned long long* pb = (const unsigned long long*)b;
    return __popcll(pa[0] ^ pb[0]) + __popcll(pa[1] ^ pb[1]) + __popcll(pa[2] ^ pb[2]) + __popcll(pa[3] ^ pb[3]);
}

// rotation histogram bin (:345-350).  factor = 1/HISTO_LENGTH is the reference's quirk; round() = C round on a float.
__device__ __forceinline__ int rot_bin(float a1, float a2)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// ComputeThreeMaxima (:2012-2053)
__device__ inline void three_maxima(const int* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// ---- SearchByBoW -------------------------------------------------------------------------------------------
struct BowPairDev {
    // side 1 = KeyFrame (KF1), side 2 = Frame (KF2 for the KF-KF variant); all offsets index the pair blob
    const uint8_t* d1; const uint8_t* v1; const float* a1; const uint32_t* node1; const int32_t* off1; const uint32_t* feat1;
    const uint8_t* d2; const uint8_t* v2; const float* a2; const uint32_t* node2; const int32_t* off2; const uint32_t* feat2;
    int32_t n1, n2, nn1, nn2;
    // device-resident pairs (orbm_bow_plan_create_device): the four counts live in device memory (outputs of the extractor
    // and of the vocabulary transform), angles are read out of OrbxKeyPoint records (stride 7 floats), v1 may be NULL (all valid)
    const int32_t* n1p = nullptr; const int32_t* n2p = nullptr; const int32_t* nn1p = nullptr; const int32_t* nn2p = nullptr;
    int32_t a1_stride = 1, a2_stride = 1;
    int32_t cap1 = 0, cap2 = 0;
    int32_t* match;         // [n2] (KF-F: Frame feature -> KF feature) or [n1] (KF-KF: KF1 feature -> KF2 feature)
    uint8_t* matched2;      // [n2] KF-KF only
    int32_t* n_matches;
    int32_t serial;
};

template <bool KFKF>
__device__ void bow_node(const BowPairDev& P, int k, int f, float nnratio, int check_ori, int* s_hist)
{
    for (int i1 = P.off1[k]; i1 < P.off1[k + 1]; i1++) {
        const int idx1 = (int)P.feat1[i1];
        if (P.v1 && !P.v1[idx1]) continue;                   // !pMP || pMP->isBad()
        const uint8_t* da = P.d1 + (size_t)idx1 * 32;
        int best1 = 256, bestIdx = -1, best2 = 256;
        for (int i2 = P.off2[f]; i2 < P.off2[f + 1]; i2++) {
            const int idx2 = (int)P.feat2[i2];
            if (KFKF) { if (P.matched2[idx2] || !P.v2[idx2]) continue; }
            else { if (P.match[idx2] >= 0) continue; }
            const int dist = hamming256(da, P.d2 + (size_t)idx2 * 32);
            if (dist < best1) { best2 = best1; best1 = dist; bestIdx = idx2; }
            else if (dist < best2) best2 = dist;
        }
        const bool low = KFKF ? (best1 < TH_LOW) : (best1 <= TH_LOW);       // :848 is strict, :327 is not
        if (low && (float)best1 < nnratio * (float)best2) {
            if (KFKF) { P.match[idx1] = bestIdx; P.matched2[bestIdx] = 1; }
            else P.match[bestIdx] = idx1;
            if (check_ori) atomicAdd(&s_hist[rot_bin(P.a1[(size_t)idx1 * P.a1_stride], P.a2[(size_t)bestIdx * P.a2_stride])], 1);
        }
    }
}

// One vocabulary node of a pair handled by a 16-lane group: the KF features of the node are taken in order (a Frame
// feature consumed by one of them must be invisible to the next), the Frame features of the node are spread over the
// lanes.  (distance, position) keys make the 16-lane minimum the reference's first best / second best.  The lanes keep
// the "already consumed" state of the candidates they own in a register bitmask (positions lane, lane+16, ...).
constexpr int kBowGroup = 16, kBowThreads = 1024;
// The two smallest keys of a 16-lane group in every lane of it.  A group is one DPP row: four exchanges (lane ^ 1, lane ^ 2 by quad
// permutes, then the mirror of each half row and of the row -- every step joins two disjoint sets) at VALU speed; the same ladder through
// ds_bpermute (__shfl_xor) was eight dependent LDS-crossbar round trips per KF feature, most of the loop.  All 16 lanes must be active.
template <int CTRL>
__device__ __forceinline__ void bow_top2_step(unsigned& k1, unsigned& k2)
{
    const unsigned o1 = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)k1, CTRL, 0xF, 0xF, false);
    const unsigned o2 = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)k2, CTRL, 0xF, 0xF, false);
    if (o1 < k1) { k2 = min(k1, o2); k1 = o1; } else k2 = min(k2, o1);
}
__device__ __forceinline__ void bow_group_top2(unsigned& k1, unsigned& k2)
{
    static_assert(kBowGroup == 16, "a group is a DPP row");
    bow_top2_step<0xB1>(k1, k2);        // quad_perm [1, 0, 3, 2]
    bow_top2_step<0x4E>(k1, k2);        // quad_perm [2, 3, 0, 1]
    bow_top2_step<0x141>(k1, k2);       // row_half_mirror
    bow_top2_step<0x140>(k1, k2);       // row_mirror
}
template <bool KFKF>
__device__ void bow_node_group(const BowPairDev& P, int k, int f, float nnratio, int check_ori, int* s_hist, int sub)
{
    const int b2 = P.off2[f], n2 = P.off2[f + 1] - b2;
    unsigned taken = 0;                                         // bit j: candidate sub + 16*j is consumed
    for (int i1 = P.off1[k]; i1 < P.off1[k + 1]; i1++) {
        const int idx1 = (int)P.feat1[i1];
        if (P.v1 && !P.v1[idx1]) continue;                   // !pMP || pMP->isBad()   (uniform over the group)
        const unsigned long long* da = (const unsigned long long*)(P.d1 + (size_t)idx1 * 32);
        const unsigned long long a0 = da[0], a1 = da[1], a2 = da[2], a3 = da[3];
        unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        for (int c = sub, j = 0; c < n2; c += kBowGroup, j++) {
            if ((taken >> j) & 1u) continue;
            const int idx2 = (int)P.feat2[b2 + c];
            if (KFKF && !P.v2[idx2]) continue;
            const unsigned long long* db = (const unsigned long long*)(P.d2 + (size_t)idx2 * 32);
            const unsigned key = ((unsigned)(__popcll(a0 ^ db[0]) + __popcll(a1 ^ db[1]) + __popcll(a2 ^ db[2]) + __popcll(a3 ^ db[3])) << 16) | (unsigned)c;
            if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
        }
        bow_group_top2(k1, k2);
        if (k1 == 0xFFFFFFFFu) continue;
        const int best1 = (int)(k1 >> 16), best2 = (k2 == 0xFFFFFFFFu) ? 256 : (int)(k2 >> 16), cbest = (int)(k1 & 0xFFFFu);
        const bool low = KFKF ? (best1 < TH_LOW) : (best1 <= TH_LOW);       // :848 is strict, :327 is not
        if (low && (float)best1 < nnratio * (float)best2) {
            if ((cbest & (kBowGroup - 1)) == sub) taken |= 1u << (cbest / kBowGroup);
            if (sub == 0) {
                const int bestIdx = (int)P.feat2[b2 + cbest];
                if (KFKF) { P.match[idx1] = bestIdx; P.matched2[bestIdx] = 1; }
                else P.match[bestIdx] = idx1;
                if (check_ori) atomicAdd(&s_hist[rot_bin(P.a1[(size_t)idx1 * P.a1_stride], P.a2[(size_t)bestIdx * P.a2_stride])], 1);
            }
        }
    }
}

// The common case -- a node with at most 32 Frame features -- without a memory access inside the loop over the KF features:
// the lanes hold their (at most two) candidates' descriptors, indices and angles in registers, the KF features of the node are
// loaded 16 at a time (lane i holds feature i) and handed round by 16-lane shuffles.  The generic form above walks four
// dependent global round trips per KF feature (index, descriptor, candidate index, candidate descriptor), which made the kernel
// a chain of memory latencies.
__device__ __forceinline__ unsigned long long shfl16_u64(unsigned long long v, int src)
{
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src, kBowGroup), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src, kBowGroup);
    return ((unsigned long long)hi << 32) | lo;
}
// J = candidates per lane: 2 (a node of at most 32 Frame features), 4 (64), 6 (96; 8 spills registers at 1 024 threads).  Real frames through a vocabulary hold a few nodes
// beyond 32 features; the generic form took 175 us for such a pair where the rest of the pair takes 20.
template <bool KFKF, int J>
__device__ void bow_node_group_regs(const BowPairDev& P, int k, int f, float nnratio, int check_ori, int* s_hist, int sub)
{
    const int b2 = P.off2[f], n2 = P.off2[f + 1] - b2;         // n2 <= 16 J
    int idx2[J];
    bool ok2[J];
    unsigned long long db[J][4];
    float ang2[J];
#pragma unroll
    for (int j = 0; j < J; j++) { const int c = sub + kBowGroup * j; ok2[j] = c < n2; idx2[j] = 0; ang2[j] = 0.f; if (ok2[j]) idx2[j] = (int)P.feat2[b2 + c]; }
#pragma unroll
    for (int j = 0; j < J; j++) {
#pragma unroll
        for (int q = 0; q < 4; q++) db[j][q] = 0ull;
        if (ok2[j]) {
            if (KFKF && !P.v2[idx2[j]]) ok2[j] = false;
            const unsigned long long* dp = (const unsigned long long*)(P.d2 + (size_t)idx2[j] * 32);
#pragma unroll
            for (int q = 0; q < 4; q++) db[j][q] = dp[q];
            if (check_ori) ang2[j] = P.a2[(size_t)idx2[j] * P.a2_stride];
        }
    }
    unsigned taken = 0;                                         // bit j: this lane's candidate j is consumed
    const int b1 = P.off1[k], n1 = P.off1[k + 1] - b1;
    for (int i0 = 0; i0 < n1; i0 += kBowGroup) {
        const int ii = i0 + sub;
        int idx1m = 0, valid = 0;
        unsigned long long A[4] = {0ull, 0ull, 0ull, 0ull};
        float ang1 = 0.f;
        if (ii < n1) {
            idx1m = (int)P.feat1[b1 + ii];
            valid = (!P.v1 || P.v1[idx1m]) ? 1 : 0;              // !pMP || pMP->isBad()
            const unsigned long long* da = (const unsigned long long*)(P.d1 + (size_t)idx1m * 32);
#pragma unroll
            for (int q = 0; q < 4; q++) A[q] = da[q];
            if (check_ori) ang1 = P.a1[(size_t)idx1m * P.a1_stride];
        }
        // (the validity of the 16 features as a bit mask from one wave ballot: no LDS-crossbar round trip per feature for it)
        const unsigned vmask = (unsigned)((__ballot(valid != 0) >> (threadIdx.x & 48)) & 0xFFFFull);
        const int cnt = min(kBowGroup, n1 - i0);
        for (int i = 0; i < cnt; i++) {
            if (!((vmask >> i) & 1u)) continue;                 // uniform over the group
            const unsigned long long a0 = shfl16_u64(A[0], i), a1 = shfl16_u64(A[1], i), a2 = shfl16_u64(A[2], i), a3 = shfl16_u64(A[3], i);
            const int idx1 = __shfl(idx1m, i, kBowGroup);
            const float an1 = __shfl(ang1, i, kBowGroup);
            unsigned k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < J; j++) {
                if (!ok2[j] || ((taken >> j) & 1u)) continue;
                const unsigned key = ((unsigned)(__popcll(a0 ^ db[j][0]) + __popcll(a1 ^ db[j][1]) + __popcll(a2 ^ db[j][2]) + __popcll(a3 ^ db[j][3])) << 16) | (unsigned)(sub + kBowGroup * j);
                if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
            }
            bow_group_top2(k1, k2);
            if (k1 == 0xFFFFFFFFu) continue;
            const int best1 = (int)(k1 >> 16), best2 = (k2 == 0xFFFFFFFFu) ? 256 : (int)(k2 >> 16), cbest = (int)(k1 & 0xFFFFu);
            const bool low = KFKF ? (best1 < TH_LOW) : (best1 <= TH_LOW);       // :848 is strict, :327 is not
            if (low && (float)best1 < nnratio * (float)best2) {
                const int jb = cbest / kBowGroup, lb = cbest & (kBowGroup - 1);
                if (lb == sub) {            // the lane that owns the winner records the match: it holds the winner's index and angle, the key-frame side is uniform
                    taken |= 1u << jb;
                    int bestIdx = idx2[0]; float an2 = ang2[0];
#pragma unroll
                    for (int j = 1; j < J; j++) if (jb == j) { bestIdx = idx2[j]; an2 = ang2[j]; }
                    if (KFKF) { P.match[idx1] = bestIdx; P.matched2[bestIdx] = 1; }
                    else P.match[bestIdx] = idx1;
                    if (check_ori) atomicAdd(&s_hist[rot_bin(an1, an2)], 1);
                }
            }
        }
    }
}

template <bool KFKF>
__global__ __launch_bounds__(kBowThreads) void k_bow(const BowPairDev* __restrict__ pairs, float nnratio, int check_ori)
{
    __shared__ int s_hist[HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count;
    BowPairDev P = pairs[blockIdx.x];
    if (P.n1p) {        // device-resident pair: counts come from device memory, clamped to the arrays' capacity
        P.n1 = min(*P.n1p, P.cap1); P.n2 = min(*P.n2p, P.cap2); P.nn1 = min(*P.nn1p, P.cap1); P.nn2 = min(*P.nn2p, P.cap2);
    }
    const int tid = threadIdx.x;
    const int n_out = KFKF ? P.n1 : P.n2;
    for (int i = tid; i < n_out; i += kBowThreads) P.match[i] = -1;
    if (KFKF) for (int i = tid; i < P.n2; i += kBowThreads) P.matched2[i] = 0;
    if (tid < HISTO_LENGTH) s_hist[tid] = 0;
    if (tid == 0) s_count = 0;
    // (measured and dropped: the Frame's node ids in LDS for the groups' binary search -- no difference, 37 us per 256 pairs either way;
    // what a node costs is its chain of dependent global reads: offsets -> feature indices -> descriptors, on both sides)
    __threadfence_block();
    __syncthreads();
    if (P.serial) {
        if (tid == 0) {
            // merge-join of the two ascending node lists (:248-401)
            int k = 0, f = 0;
            while (k < P.nn1 && f < P.nn2) {
                if (P.node1[k] == P.node2[f]) { bow_node<KFKF>(P, k, f, nnratio, check_ori, s_hist); k++; f++; }
                else if (P.node1[k] < P.node2[f]) k++;
                else f++;
            }
        }
    } else {
        const int sub = tid & (kBowGroup - 1);
        for (int k = tid / kBowGroup; k < P.nn1; k += kBowThreads / kBowGroup) {
            const uint32_t key = P.node1[k];
            int lo = 0, hi = P.nn2;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.node2[mid] < key) lo = mid + 1; else hi = mid; }
            if (lo < P.nn2 && P.node2[lo] == key) {
                // a node with more than 512 Frame features does not fit the per-lane bitmask: one lane walks it
                const int n2 = P.off2[lo + 1] - P.off2[lo];
                if (n2 > 32 * kBowGroup) { if (sub == 0) bow_node<KFKF>(P, k, lo, nnratio, check_ori, s_hist); }
                else if (n2 <= 2 * kBowGroup) bow_node_group_regs<KFKF, 2>(P, k, lo, nnratio, check_ori, s_hist, sub);
                else if (n2 <= 4 * kBowGroup) bow_node_group_regs<KFKF, 4>(P, k, lo, nnratio, check_ori, s_hist, sub);
                else if (n2 <= 6 * kBowGroup) bow_node_group_regs<KFKF, 6>(P, k, lo, nnratio, check_ori, s_hist, sub);
                else bow_node_group<KFKF>(P, k, lo, nnratio, check_ori, s_hist, sub);
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    if (check_ori) {
        if (tid == 0) { int a, b, c; three_maxima(s_hist, HISTO_LENGTH, a, b, c); s_keep[0] = a; s_keep[1] = b; s_keep[2] = c; }
        __syncthreads();
    }
    int local = 0;
    for (int i = tid; i < n_out; i += kBowThreads) {
        const int m = P.match[i];
        if (m < 0) continue;
        if (check_ori) {
            const int bin = KFKF ? rot_bin(P.a1[(size_t)i * P.a1_stride], P.a2[(size_t)m * P.a2_stride])
                                 : rot_bin(P.a1[(size_t)m * P.a1_stride], P.a2[(size_t)i * P.a2_stride]);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) { P.match[i] = -1; continue; }
        }
        local++;
    }
    if (local) atomicAdd(&s_count, local);
    __syncthreads();
    if (tid == 0) *P.n_matches = s_count;
}

// ---- SearchByProjection ------------------------------------------------------------------------------------
struct ProjFrameDev {
    const float* x; const float* y; const int32_t* octave; const float* angle; const uint8_t* desc;
    const int32_t* cell_off;    // [cols*rows + 1], cell index = ix*rows + iy (mGrid[ix][iy])
    const int32_t* cell_feat;   // features of a cell in insertion order (AssignFeaturesToGrid, src/Frame.cc:472-503)
    const float* scale_factors;
    float min_x, min_y, max_x, max_y, winv, hinv;
    int32_t n, cols, rows;
    int32_t n_levels = 0;       // > 0: k_proj keeps the scale factors in LDS
    const float* u_right = nullptr;     // mvuRight (rectified stereo / RGB-D), nullptr for a monocular frame
};

constexpr unsigned long long kNoKey = ~0ull;
// key: dist (9 bits) | cell ordinal in the window (20) | position in the cell (14) | feature index (21)
__device__ __forceinline__ unsigned long long make_key(int dist, int cell_ord, int j, int idx)
{
    return ((unsigned long long)dist << 55) | ((unsigned long long)cell_ord << 35) | ((unsigned long long)j << 21) | (unsigned long long)idx;
}
__device__ __forceinline__ int key_dist(unsigned long long k) { return (int)(k >> 55); }
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)(k & 0x1FFFFFull); }

__device__ __forceinline__ unsigned long long read_lane64(unsigned long long v, int l)     // l is wave-uniform
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// Wave-wide minimum without LDS traffic: the classic gfx9 DPP ladder (row_shr 1/2/4/8, row_bcast 15, row_bcast 31) leaves
// the minimum in lane 63.  Lanes without a source keep `old` = identity, so no row / bank masks are needed.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x111, 0xF, 0xF, false));    // row_shr:1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x112, 0xF, 0xF, false));    // row_shr:2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x114, 0xF, 0xF, false));    // row_shr:4
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x118, 0xF, 0xF, false));    // row_shr:8
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x142, 0xF, 0xF, false));    // row_bcast:15
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, 0x143, 0xF, 0xF, false));    // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
    const unsigned hi = (unsigned)(k >> 32);
    const unsigned mh = wave_min_u32(hi);
    const unsigned ml = wave_min_u32(hi == mh ? (unsigned)k : 0xFFFFFFFFu);
    return ((unsigned long long)mh << 32) | ml;
}

__device__ __forceinline__ void top2_insert(unsigned long long& k1, unsigned long long& k2, unsigned long long k)
{
    if (k < k1) { k2 = k1; k1 = k; }
    else if (k < k2) k2 = k;
}

#ifdef ORBM_PROJ_TIMING      // cycle sums of job 0 (tools/proj_timing.py); never defined in the product build
__device__ unsigned long long d_proj_prof[8];
#define ORBM_PTICK(k) if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t_now = clock64(); d_proj_prof[k] += (unsigned long long)(t_now - t_prev); t_prev = t_now; }
#else
#define ORBM_PTICK(k)
#endif

// Frame::GetFeaturesInArea (src/Frame.cc:744-810) + the candidate loop of SearchByProjection: returns the two
// smallest candidates in (distance, visiting order) lexicographic order == the reference's best / second best.
// pt_ur: the point's predicted right-image column (mTrackProjXR, :94 / uv(0) - mbf*invzc, :1753); a candidate that has a
// right coordinate of its own (mvuRight > 0) is skipped when the two differ by more than the window radius.
// the LDS traffic of ONE wave executes in order: between its own writes and reads only the compiler has to be held back
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// WAVE_ONLY: the calling wave is one of several in its workgroup and works alone (k_proj_par): no workgroup barrier
template <bool WAVE_ONLY = false>
__device__ __forceinline__ void search_window(const ProjFrameDev& F, const uint8_t* s_occ, float x, float y, float r, int minLevel, int maxLevel,
                              float pt_ur, const unsigned long long* dq, int lane, int* s_col, unsigned long long& best, unsigned long long& second,
                              const uint4* __restrict__ recs = nullptr)     // recs: the key points in grid order {x, y, octave, feature} (k_proj_par's staged frames: they replace cell_feat / x / y / octave)
{
    best = second = kNoKey;
    const int nMinCellX = max(0, (int)floorf((x - F.min_x - r) * F.winv));
    if (nMinCellX >= F.cols) return;
    const int nMaxCellX = min(F.cols - 1, (int)ceilf((x - F.min_x + r) * F.winv));
    if (nMaxCellX < 0) return;
    const int nMinCellY = max(0, (int)floorf((y - F.min_y - r) * F.hinv));
    if (nMinCellY >= F.rows) return;
    const int nMaxCellY = min(F.rows - 1, (int)ceilf((y - F.min_y + r) * F.hinv));
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const int ny = nMaxCellY - nMinCellY + 1, nx = nMaxCellX - nMinCellX + 1;
    const int ncell = nx * ny;
    unsigned long long k1 = kNoKey, k2 = kNoKey;
#ifdef ORBM_PROJ_TIMING
    long long t_prev = clock64();
#endif
    if (nx <= 16 && F.n < 16384) {
        // The cells of one grid column are adjacent in the CSR (cell = ix * rows + iy), so a window is nx contiguous runs
        // of features -- and walking a run front to back IS the reference's visiting order.  The runs are flattened over
        // the lanes (exclusive prefix of the run lengths by a DPP ladder), which keeps the wave busy even when one cell
        // holds most of the candidates.
        int run_b = 0, run_n = 0;
        if (lane < nx) {
            const int cell0 = (nMinCellX + lane) * F.rows + nMinCellY;
            run_b = F.cell_off[cell0];
            run_n = F.cell_off[cell0 + ny] - run_b;
        }
        int inc = run_n;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, false);        // row_shr:1 .. 8: inclusive prefix over lanes 0..15
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, false);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, false);
        const int total = __builtin_amdgcn_readlane(inc, nx - 1);
        if (lane < nx) { s_col[lane] = inc - run_n; s_col[16 + lane] = run_b; }
        if (WAVE_ONLY) wave_lds_sync(); else __syncthreads();
        for (int t = lane; t < total; t += 64) {
            int cx = 0;
            for (int k = 1; k < nx; k++) cx += (t >= s_col[k]) ? 1 : 0;
            const int pos = t - s_col[cx];
            int idx; float distx, disty;
            if (recs) {
                const uint4 rc = recs[s_col[16 + cx] + pos];
                idx = (int)rc.w;
                const int oc = (int)rc.z;
                if (bCheckLevels && (oc < minLevel || (maxLevel >= 0 && oc > maxLevel))) continue;
                distx = __uint_as_float(rc.x) - x; disty = __uint_as_float(rc.y) - y;
            } else {
                idx = F.cell_feat[s_col[16 + cx] + pos];
                if (bCheckLevels) {
                    const int oc = F.octave[idx];
                    if (oc < minLevel) continue;
                    if (maxLevel >= 0 && oc > maxLevel) continue;
                }
                distx = F.x[idx] - x; disty = F.y[idx] - y;
            }
            if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
            if (s_occ[idx]) continue;                                        // mvpMapPoints[idx] && Observations()>0
            if (F.u_right) {                                                 // :92-98, :1751-1757
                const float ur = F.u_right[idx];
                if (ur > 0.f && fabsf(pt_ur - ur) > r) continue;
            }
            const unsigned long long* db = (const unsigned long long*)(F.desc + (size_t)idx * 32);
            const int dist = __popcll(dq[0] ^ db[0]) + __popcll(dq[1] ^ db[1]) + __popcll(dq[2] ^ db[2]) + __popcll(dq[3] ^ db[3]);
            if (dist >= 256) continue;
            top2_insert(k1, k2, make_key(dist, cx, pos, idx));               // (column, position in its run) = visiting order
        }
        __builtin_amdgcn_wave_barrier();
    } else {
    const float inv_ny = 1.0f / (float)ny;
    for (int c = lane; c < ncell; c += 64) {
        const int cq = (int)(((float)c + 0.5f) * inv_ny);                   // c / ny (exact: c < 2^12, ny <= 2^7)
        const int ix = nMinCellX + cq, iy = nMinCellY + (c - cq * ny);     // ix outer, iy inner (:774-778)
        const int cell = ix * F.rows + iy;
        const int e0 = F.cell_off[cell], e1 = F.cell_off[cell + 1];
        for (int e = e0; e < e1; e++) {
            int idx; float distx, disty;
            if (recs) {
                const uint4 rc = recs[e];
                idx = (int)rc.w;
                const int oc = (int)rc.z;
                if (bCheckLevels && (oc < minLevel || (maxLevel >= 0 && oc > maxLevel))) continue;
                distx = __uint_as_float(rc.x) - x; disty = __uint_as_float(rc.y) - y;
            } else {
                idx = F.cell_feat[e];
                if (bCheckLevels) {
                    const int oc = F.octave[idx];
                    if (oc < minLevel) continue;
                    if (maxLevel >= 0 && oc > maxLevel) continue;
                }
                distx = F.x[idx] - x; disty = F.y[idx] - y;
            }
            if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
            if (s_occ[idx]) continue;                                        // mvpMapPoints[idx] && Observations()>0
            if (F.u_right) {
                const float ur = F.u_right[idx];
                if (ur > 0.f && fabsf(pt_ur - ur) > r) continue;
            }
            const unsigned long long* db = (const unsigned long long*)(F.desc + (size_t)idx * 32);
            const int dist = __popcll(dq[0] ^ db[0]) + __popcll(dq[1] ^ db[1]) + __popcll(dq[2] ^ db[2]) + __popcll(dq[3] ^ db[3]);
            if (dist >= 256) continue;                                       // never beats the initial bestDist = 256
            top2_insert(k1, k2, make_key(dist, c, e - e0, idx));
        }
    }
    }
    ORBM_PTICK(1)
    // wave reduction of the (k1, k2) pairs with DPP minimum ladders (no LDS round trips): best, then second best
    // (keys are unique, so exactly one lane owns the minimum; it promotes its runner-up before the second pass)
    const unsigned long long b1 = wave_min_u64(k1);
    if (k1 == b1) { k1 = k2; k2 = kNoKey; }
    const unsigned long long b2 = (b1 == kNoKey) ? kNoKey : wave_min_u64(k1);
    best = b1; second = b2;
    ORBM_PTICK(2)
}

// ---- Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:472-503, 812-822) on the device ---------------------
// One workgroup per frame: key = (cell, feature index), bitonic sort in LDS -> the CSR the window searches walk
// (features of a cell in insertion order = ascending index), cell offsets by binary search over the sorted keys.
struct ProjArgs;
__global__ __launch_bounds__(256) void k_grid(const ProjArgs* __restrict__ jobs, int n_pow2);

struct ProjArgs {
    ProjFrameDev F;
    int32_t n_pts;
    const uint8_t* valid;       // in_view (M4) / last_valid (M5)
    const float* u; const float* v;
    const float* ur;            // mTrackProjXR (M4) / uv(0) - mbf*invzc (M5); read only when F.u_right != nullptr
    const int32_t* level;       // predicted level (M4) / last octave (M5)
    const float* view_cos; const float* depth; const uint8_t* bad;      // M4 only
    const float* angle;         // M5: last frame keypoint angle
    const uint8_t* desc; const uint8_t* has_obs;
    float th, th_far, nnratio, dist_th;
    int32_t far_points, check_ori;
    int32_t last_frame_mode;    // 0: Frame x map points (:43); 1: window around a projected feature, levels l-1..l+1, image-bounds
                                // check, rotation histogram (:1676 last frame, :1889 key frame); 2: Sim3 key-frame search (:427, :534)
    int32_t level_window;       // mode 1: ORBM_LEVELS_AROUND / _FORWARD (bForward, :1729) / _BACKWARD (bBackward, :1731)
    int32_t* assign; uint8_t* occupied;
    int32_t* log_feat; int32_t* log_bin;     // M5 rotation log, capacity n_pts
    int32_t* n_matches;
    int32_t lds_frame;                       // stage the frame (grid, key points, descriptors) in LDS
};

__global__ __launch_bounds__(256) void k_grid(const ProjArgs* __restrict__ jobs, int n_pow2_max)
{
    extern __shared__ __align__(16) unsigned long long s_gkey[];
    const ProjFrameDev F = jobs[blockIdx.x].F;
    const int tid = threadIdx.x, n = F.n, ncell = F.cols * F.rows;
    int32_t* cell_off = const_cast<int32_t*>(F.cell_off);
    int32_t* cell_feat = const_cast<int32_t*>(F.cell_feat);
    int n_pow2 = 2;                             // this frame's own sort size (the launch's LDS is sized for the largest frame / the arrays' capacity)
    while (n_pow2 < n) n_pow2 <<= 1;
    n_pow2 = min(n_pow2, n_pow2_max);
    for (int i = tid; i < n_pow2; i += 256) {
        unsigned long long key = ~0ull;
        if (i < n) {
            const int px = (int)roundf((F.x[i] - F.min_x) * F.winv), py = (int)roundf((F.y[i] - F.min_y) * F.hinv);      // PosInGrid
            if (px >= 0 && px < F.cols && py >= 0 && py < F.rows) key = ((unsigned long long)(px * F.rows + py) << 32) | (unsigned)i;
        }
        s_gkey[i] = key;
    }
    // bitonic sort by the 4 waves: wave w owns chunk w (n_pow2 / 4 keys); an exchange with a stride below the chunk size stays inside
    // it and needs only the wave's own program order, the three passes that reach across chunks a workgroup barrier (it was one per pass)
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int chunk = n_pow2 >> 2, pairs = n_pow2 >> 3;
        bool need_block = true;
        for (int k = 2; k <= n_pow2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool cross = j >= chunk || n_pow2 < 8;
                if (cross || need_block) __syncthreads();
                else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
                need_block = cross;
                const int t0 = n_pow2 >= 8 ? wave * pairs + lane : tid, t1 = n_pow2 >= 8 ? (wave + 1) * pairs : (n_pow2 >> 1), ts = n_pow2 >= 8 ? 64 : 256;
                for (int t = t0; t < t1; t += ts) {
                    const int lo = 2 * t - (t & (j - 1)), hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const unsigned long long a = s_gkey[lo], b = s_gkey[hi];
                    if ((a > b) == up) { s_gkey[lo] = b; s_gkey[hi] = a; }
                }
            }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) if (s_gkey[i] != ~0ull) cell_feat[i] = (int32_t)(s_gkey[i] & 0xFFFFFFFFull);
    for (int c = tid; c <= ncell; c += 256) {            // cell_off[c] = number of keys with cell < c
        int lo = 0, hi = n_pow2;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((s_gkey[mid] >> 32) < (unsigned long long)c) lo = mid + 1; else hi = mid; }
        cell_off[c] = lo;
    }
}

// STAGE: the frame fits in LDS (host decides for the whole launch); a compile-time switch so that every frame access in
// that instantiation is a ds_read (address space known) rather than a flat load through a may-be-LDS pointer.
template <bool STAGE>
__global__ __launch_bounds__(64) void k_proj(const ProjArgs* __restrict__ jobs)
{
    const ProjArgs A = jobs[blockIdx.x];        // one wave per job (frame): jobs are independent, a batch is one launch
    extern __shared__ __align__(16) uint8_t s_dyn[];
    __shared__ int s_hist[HISTO_LENGTH];
    const int lane = threadIdx.x;
    ProjFrameDev F = A.F;
    // The points are searched one after the other (greedy dependence), so a point costs a chain of dependent reads of
    // the frame's grid, key points and descriptors.  When the frame fits (49 B per feature + the cell table), it is staged
    // in LDS once and the chain runs at LDS latency instead of HBM / L2 latency.
    uint8_t* s_occ = s_dyn;
    if (STAGE) {
        const int n = F.n, ncell = F.cols * F.rows;
        size_t off = ((size_t)n + 15) & ~(size_t)15;
        uint8_t* s_desc = s_dyn + off; off += (size_t)n * 32;
        float* s_x = (float*)(s_dyn + off); off += (size_t)n * 4;
        float* s_y = (float*)(s_dyn + off); off += (size_t)n * 4;
        int32_t* s_oct = (int32_t*)(s_dyn + off); off += (size_t)n * 4;
        int32_t* s_cfeat = (int32_t*)(s_dyn + off); off += (size_t)n * 4;
        float* s_ur = (float*)(s_dyn + off); off += F.u_right ? (size_t)n * 4 : 0;
        int32_t* s_coff = (int32_t*)(s_dyn + off);
        for (int i = lane; i < n * 8; i += 64) ((uint32_t*)s_desc)[i] = ((const uint32_t*)F.desc)[i];
        for (int i = lane; i < n; i += 64) { s_x[i] = F.x[i]; s_y[i] = F.y[i]; s_oct[i] = F.octave[i]; }
        if (F.u_right) { for (int i = lane; i < n; i += 64) s_ur[i] = F.u_right[i]; F.u_right = s_ur; }
        const int nfeat_cells = F.cell_off[ncell];
        for (int i = lane; i < nfeat_cells; i += 64) s_cfeat[i] = F.cell_feat[i];
        for (int i = lane; i <= ncell; i += 64) s_coff[i] = F.cell_off[i];
        F.desc = s_desc; F.x = s_x; F.y = s_y; F.octave = s_oct; F.cell_feat = s_cfeat; F.cell_off = s_coff;
    }
    __shared__ float s_scale[32];
    __shared__ int s_col[32];                   // search_window: run starts / CSR bases of the window's grid columns
    if (F.n_levels > 0 && F.n_levels <= 32) {
        if (lane < F.n_levels) s_scale[lane] = F.scale_factors[lane];
        F.scale_factors = s_scale;
    }
    for (int i = lane; i < F.n; i += 64) s_occ[i] = A.occupied[i];
    if (lane < HISTO_LENGTH) s_hist[lane] = 0;
    __syncthreads();
    int nmatches = 0, nlog = 0;
    const bool bFactor = A.th != 1.0f;
    // point data is fetched one point ahead: the loads of point i+1 are in flight while point i walks its window
    struct Pt { int valid, level, bad; float u, v, ur, depth, vcos; unsigned long long d[4]; };
    auto load_pt = [&](int i) {
        Pt p;
        p.valid = A.valid[i]; p.level = A.level[i]; p.u = A.u[i]; p.v = A.v[i];
        p.ur = A.F.u_right ? A.ur[i] : 0.f;
        p.bad = 0; p.depth = 0.f; p.vcos = 0.f;
        if (!A.last_frame_mode) { p.bad = A.bad[i]; p.depth = A.depth[i]; p.vcos = A.view_cos[i]; }
        const unsigned long long* dp = (const unsigned long long*)(A.desc + (size_t)i * 32);
        p.d[0] = dp[0]; p.d[1] = dp[1]; p.d[2] = dp[2]; p.d[3] = dp[3];
        return p;
    };
    Pt nxt;
    if (A.n_pts > 0) nxt = load_pt(0);
#ifdef ORBM_PROJ_TIMING
    long long t_prev = clock64();
    const long long t_begin = t_prev;
#endif
    for (int i = 0; i < A.n_pts; i++) {
        const Pt cur = nxt;
        if (i + 1 < A.n_pts) nxt = load_pt(i + 1);
        if (!cur.valid) continue;
        float x = cur.u, y = cur.v, r;
        int minLevel, maxLevel;
        if (!A.last_frame_mode) {
            if (A.far_points && cur.depth > A.th_far) continue;
            if (cur.bad) continue;
            const int lvl = cur.level;
            r = ((double)cur.vcos > 0.998) ? 2.5f : 4.0f;             // RadiusByViewingCos (:215-221)
            if (bFactor) r *= A.th;
            r = r * F.scale_factors[lvl];
            minLevel = lvl - 1; maxLevel = lvl;
        } else if (A.last_frame_mode == 1) {
            if (x < F.min_x || x > F.max_x) continue;                  // :1711-1714, :1917-1920
            if (y < F.min_y || y > F.max_y) continue;
            const int oct = cur.level;
            r = A.th * F.scale_factors[oct];
            if (A.level_window == ORBM_LEVELS_FORWARD) { minLevel = oct; maxLevel = -1; }          // :1729
            else if (A.level_window == ORBM_LEVELS_BACKWARD) { minLevel = 0; maxLevel = oct; }     // :1731
            else { minLevel = oct - 1; maxLevel = oct + 1; }                                        // :1733, :1938
        } else {
            const int lvl = cur.level;
            r = A.th * F.scale_factors[lvl];                          // :489
            minLevel = lvl - 1; maxLevel = lvl;                        // :509 (KeyFrame::GetFeaturesInArea itself does not filter)
        }
        unsigned long long kb, ks;
        search_window(F, s_occ, x, y, r, minLevel, maxLevel, cur.ur, cur.d, lane, s_col, kb, ks);
        if (kb == kNoKey) continue;
        const int bestDist = key_dist(kb), bestIdx = key_idx(kb);
        if ((float)bestDist > A.dist_th) continue;                    // TH_HIGH (:122, :1844), ORBdist (:1967), TH_LOW*ratioHamming (:522)
        if (!A.last_frame_mode) {
            const int bestDist2 = (ks == kNoKey) ? 256 : key_dist(ks);
            const int bestLevel = F.octave[bestIdx];
            const int bestLevel2 = (ks == kNoKey) ? -1 : F.octave[key_idx(ks)];
            if (bestLevel == bestLevel2 && (float)bestDist > A.nnratio * (float)bestDist2) continue;   // :126-127
        }
        if (lane == 0) {
            A.assign[bestIdx] = i;
            s_occ[bestIdx] = A.has_obs ? A.has_obs[i] : (uint8_t)1;
            if (A.last_frame_mode == 1 && A.check_ori) {
                const int bin = rot_bin(A.angle[i], F.angle[bestIdx]);
                A.log_feat[nlog] = bestIdx; A.log_bin[nlog] = bin;
                s_hist[bin]++;
            }
        }
        nlog++;
        nmatches++;
        __syncthreads();        // the occupancy update must be seen by the next point's window search
    }
    __syncthreads();
#ifdef ORBM_PROJ_TIMING
    if (blockIdx.x == 0 && lane == 0) { d_proj_prof[0] += (unsigned long long)(clock64() - t_begin); d_proj_prof[7] += (unsigned long long)A.n_pts; }
#endif
    if (A.last_frame_mode == 1 && A.check_ori && lane == 0) {
        int i1, i2, i3;
        three_maxima(s_hist, HISTO_LENGTH, i1, i2, i3);
        for (int k = 0; k < nlog; k++) {
            const int b = A.log_bin[k];
            if (b != i1 && b != i2 && b != i3) {
                A.assign[A.log_feat[k]] = -1;           // CurrentFrame.mvpMapPoints[...] = NULL (:1878)
                s_occ[A.log_feat[k]] = 0;
                nmatches--;
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < F.n; i += 64) A.occupied[i] = s_occ[i];
    if (lane == 0) *A.n_matches = nmatches;
}

// ---- the same searches with the points taken 64 at a time (one lane per point) -----------------------------------------
// The reference walks the points in order and a point only sees the features no earlier point has taken (greedy dependence):
// k_proj above does exactly that, one point after the other with the wave spread over its window -- about 5.8 k cycles per
// point, 2.3 ms for 256 frames x 900 points.  Here a block of 64 points is searched SPECULATIVELY, every lane walking its own
// point's window against the occupancy at the start of the block; then the block is resolved in point order:
//   * taking features out of the pool can only change a point's outcome if its best (or, where the ratio test looks at it,
//     its second best) candidate is taken -- every other candidate is irrelevant to the result;
//   * so a lane is DIRTY iff its best / second best has been occupied since the block began or is claimed by an earlier lane
//     of the block that will occupy it.  All lanes in front of the first dirty lane d have final results: they commit
//     (assignment, occupancy, rotation log; same-feature writers in point order), lane d is searched again by the whole wave
//     against the now exact occupancy (search_window), commits, and the remaining lanes are re-examined.
// Results are those of the sequential loop, assignment for assignment (tests: every M4 / M5 / key-frame / Sim3 case and the goldens).
// Speculative search of a block of 64 points, load-balanced: the windows of the 64 points are flattened into ONE list of
// (point, candidate) items -- a window is at most 16 contiguous CSR runs, one per grid column, and the items of a point are its
// runs front to back = the reference's visiting order -- and every lane takes an equal, contiguous share of the list (a lane per
// point would make the wave wait for the largest window: 234 candidates against a mean of 51 on the synthetic frames).  A lane
// keeps the two best keys of the point it is walking in registers and merges them into the point's LDS slots when it moves on.
constexpr int kBlkRuns = 16;
constexpr int kBlkSlots = 64 * 8 + 64;   // threads of a k_proj_par workgroup + points of a block
struct ProjBlockTab {                   // the points of a block and their windows (two in LDS: one is searched / resolved while the next but one is set up)
    int run_start[64][kBlkRuns];        // CSR position of the first feature of a run
    int run_cum[64][kBlkRuns + 1];      // items of the point in front of the run
    int pt_cum[65];                     // items in front of the point
    float prm[64][4];                   // x, y, r, predicted right column
    int lvl[64][2];                     // minLevel, maxLevel
    unsigned long long dq[64][4];
    int flags[64];                      // bit 0 live, bit 1 irregular window, bits 8..15 the occupancy value a match writes
};
struct ProjBlockPart {                  // (best, second) a thread found for a point, slot = thread + point (unique: both only grow along the list).  Every slot
    unsigned long long part1[kBlkSlots], part2[kBlkSlots];  // block_merge reads was written by block_items for the same block (a thread stores a slot for every
};                                      // point it walked an item of, and block_merge reads exactly those threads): the slots need no initialisation.
// recs: the frame's key points in CSR (grid) order, 16 bytes each {x, y, octave, feature index}: one LDS read per item whose
// address does not depend on a previous read, so the record of item t + 1 is in flight while item t is examined (nullptr: built
// from the separate arrays, the path of frames too large to stage).
constexpr int kProjWaves = 8;           // waves of a k_proj_par workgroup: wave 0 resolves a block, wave 1 sets up the next but one, waves 2-7 walk the next one's item list
constexpr int kProjThreads = 64 * kProjWaves;
// phase 1 (one wave, lane = point): window -> runs; returns the point's item count
__device__ __forceinline__ int block_runs(const ProjFrameDev& F, ProjBlockTab& B, int lane, bool live,
                                          float x, float y, float r, int minLevel, int maxLevel, float pt_ur, bool& irregular)
{
    irregular = false;
    int cnt = 0, nx = 0;
    if (live) {
        const int nMinCellX = max(0, (int)floorf((x - F.min_x - r) * F.winv));
        const int nMaxCellX = min(F.cols - 1, (int)ceilf((x - F.min_x + r) * F.winv));
        const int nMinCellY = max(0, (int)floorf((y - F.min_y - r) * F.hinv));
        const int nMaxCellY = min(F.rows - 1, (int)ceilf((y - F.min_y + r) * F.hinv));
        if (nMinCellX < F.cols && nMaxCellX >= 0 && nMinCellY < F.rows && nMaxCellY >= 0) {
            nx = nMaxCellX - nMinCellX + 1;
            const int ny = nMaxCellY - nMinCellY + 1;
            if (nx > kBlkRuns) { irregular = true; nx = 0; }       // a window wider than 16 grid columns: left to the wave-wide search
            for (int k = 0; k < nx; k++) {
                const int cell0 = (nMinCellX + k) * F.rows + nMinCellY;
                const int b0 = F.cell_off[cell0];
                B.run_start[lane][k] = b0;
                B.run_cum[lane][k] = cnt;
                cnt += F.cell_off[cell0 + ny] - b0;
            }
        }
    }
    B.run_cum[lane][nx] = cnt;
    for (int k = nx + 1; k <= kBlkRuns; k++) B.run_cum[lane][k] = 0x7FFFFFFF;
    B.prm[lane][0] = x; B.prm[lane][1] = y; B.prm[lane][2] = r; B.prm[lane][3] = pt_ur;
    B.lvl[lane][0] = minLevel; B.lvl[lane][1] = maxLevel;
    int incl = cnt;                                      // inclusive prefix over the 64 lanes (DPP ladder)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
    B.pt_cum[lane] = incl - cnt;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    if (lane == 0) B.pt_cum[64] = total;
    return cnt;
}
__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src)
{
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// phase 2 (the searching waves, thread = an equal, contiguous share of the item list).  Whole waves only: it ends with wave-wide shuffles.
__device__ __forceinline__ void block_items(const ProjFrameDev& F, const uint4* __restrict__ recs, const uint8_t* s_occ, const ProjBlockTab& B, ProjBlockPart& S, int tid, int nthreads)
{
    const int total = B.pt_cum[64];
    if (total == 0) return;
    const int lane = tid;
    auto load_rec = [&](int e) -> uint4 {
        if (recs) return recs[e];
        const int idx = F.cell_feat[e];
        return make_uint4(__float_as_uint(F.x[idx]), __float_as_uint(F.y[idx]), (unsigned)F.octave[idx], (unsigned)idx);
    };
    const int chunk = (total + nthreads - 1) / nthreads;     // (nthreads: the threads that walk this block's list -- all 8 waves for the first block, 6 afterwards)
    const int t0 = lane * chunk, t1 = min(total, t0 + chunk);
    int cp = -1;                                        // point whose parameters are in registers
    unsigned long long k1 = kNoKey, k2 = kNoKey;
    if (t0 < t1) {
        int p = 0;                                      // last point with pt_cum[p] <= t0
        for (int step = 32; step > 0; step >>= 1) if (p + step < 64 && B.pt_cum[p + step] <= t0) p += step;
        int local = t0 - B.pt_cum[p];
        int k = 0;                                      // last run of p with run_cum <= local
        for (int step = 8; step > 0; step >>= 1) if (k + step < kBlkRuns && B.run_cum[p][k + step] <= local) k += step;
        int e = B.run_start[p][k] + (local - B.run_cum[p][k]);
        int run_end_local = B.run_cum[p][k + 1];        // first item of the next run (0x7FFFFFFF behind the last run)
        int pt_end = B.pt_cum[p + 1];
        int fp = p, ford = local;                       // point and visiting-order position of the record in flight
        uint4 rec = load_rec(e);
        float px = 0.f, py = 0.f, pr = 0.f, pur = 0.f;
        int mnl = 0, mxl = 0;
        unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        for (int t = t0; t < t1; t++) {
            const uint4 cur = rec;
            const int ip = fp, ord = ford;
            if (t + 1 < t1) {                           // position of item t + 1; its record's load goes out before item t is examined
                e++; local++;
                if (t + 1 >= pt_end) {
                    do { p++; pt_end = B.pt_cum[p + 1]; } while (t + 1 >= pt_end);
                    local = t + 1 - B.pt_cum[p]; k = 0;
                    e = B.run_start[p][0]; run_end_local = B.run_cum[p][1];
                }
                while (local >= run_end_local) { k++; e = B.run_start[p][k]; run_end_local = B.run_cum[p][k + 1]; }      // (empty runs are skipped)
                fp = p; ford = local;
                rec = load_rec(e);
            }
            if (ip != cp) {
                if (cp >= 0) { S.part1[lane + cp] = k1; S.part2[lane + cp] = k2; }     // plain stores: no atomic round trips in the loop
                k1 = k2 = kNoKey;
                cp = ip;
                px = B.prm[cp][0]; py = B.prm[cp][1]; pr = B.prm[cp][2]; pur = B.prm[cp][3];
                mnl = B.lvl[cp][0]; mxl = B.lvl[cp][1];
                q0 = B.dq[cp][0]; q1 = B.dq[cp][1]; q2 = B.dq[cp][2]; q3 = B.dq[cp][3];
            }
            const int oc = (int)cur.z, idx = (int)cur.w;
            const bool bCheckLevels = (mnl > 0) || (mxl >= 0);
            if (bCheckLevels && (oc < mnl || (mxl >= 0 && oc > mxl))) continue;
            const float distx = __uint_as_float(cur.x) - px, disty = __uint_as_float(cur.y) - py;
            if (!(fabsf(distx) < pr && fabsf(disty) < pr)) continue;
            // the rest only depends on idx: occupancy, right column and descriptor are requested together
            const unsigned char oc8 = s_occ[idx];
            const float ur = F.u_right ? F.u_right[idx] : 0.f;
            const unsigned long long* db = (const unsigned long long*)(F.desc + (size_t)idx * 32);
            const unsigned long long d0 = db[0], d1 = db[1], d2 = db[2], d3 = db[3];
            if (oc8) continue;
            if (F.u_right && ur > 0.f && fabsf(pur - ur) > pr) continue;
            const int dist = __popcll(q0 ^ d0) + __popcll(q1 ^ d1) + __popcll(q2 ^ d2) + __popcll(q3 ^ d3);
            if (dist >= 256) continue;
            const unsigned long long key = ((unsigned long long)dist << 55) | ((unsigned long long)ord << 21) | (unsigned long long)idx;
            if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
        }
    }
    // The threads whose share ENDS inside the same point are neighbours (a large window spans many of them): their results are combined
    // in the wave -- a segmented suffix reduction over the lanes: four DPP row shifts inside each row of 16 lanes, then the first lanes
    // of the up to three following rows (a run is contiguous: if the first lane of a later row belongs to it, so does everything in
    // between), whose reads are independent and in flight together -- and only the first lane of such a run stores them, the others
    // store "nothing", so that block_merge, which is on wave 0's serial path, reads a handful of slots per point instead of one per
    // thread that walked it.
    const int hl = threadIdx.x & 63;
#define ORBM_ROW_SHL64(v, ctrl) (((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((v) >> 32), ctrl, 0xF, 0xF, false) << 32) | \
                                 (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v), ctrl, 0xF, 0xF, false))
#define ORBM_ROW_STEP(ctrl) { \
        const int ocp = __builtin_amdgcn_update_dpp(-2, cp, ctrl, 0xF, 0xF, false);       /* lane + n of the same row, -2 beyond the row */ \
        const unsigned long long o1 = ORBM_ROW_SHL64(k1, ctrl), o2 = ORBM_ROW_SHL64(k2, ctrl); \
        if (cp >= 0 && ocp == cp) { if (o1 < k1) { k2 = min(k1, o2); k1 = o1; } else { k2 = min(k2, o1); } } }       /* (disjoint item sets: plain two-smallest merge) */
    ORBM_ROW_STEP(0x101) ORBM_ROW_STEP(0x102) ORBM_ROW_STEP(0x104) ORBM_ROW_STEP(0x108)
#undef ORBM_ROW_STEP
#undef ORBM_ROW_SHL64
    {
        const int row0 = hl & ~15;
        int ocp[3]; unsigned long long o1[3], o2[3];
#pragma unroll
        for (int j = 0; j < 3; j++) { const int src = min(row0 + 16 * (j + 1), 63); ocp[j] = __shfl(cp, src); o1[j] = shfl64(k1, src); o2[j] = shfl64(k2, src); }
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (row0 + 16 * (j + 1) < 64 && cp >= 0 && ocp[j] == cp) { if (o1[j] < k1) { k2 = min(k1, o2[j]); k1 = o1[j]; } else { k2 = min(k2, o1[j]); } }
    }
    const int pcp = __shfl(cp, max(hl - 1, 0));
    if (cp >= 0) {
        const bool head = hl == 0 || pcp != cp;
        S.part1[lane + cp] = head ? k1 : kNoKey; S.part2[lane + cp] = head ? k2 : kNoKey;
    }
}
// phase 3 (wave 0, lane = point again): the two smallest keys over the threads that walked a part of its items
__device__ __forceinline__ void block_merge(const ProjBlockTab& B, const ProjBlockPart& S, int lane, int cnt, int nthreads, unsigned long long& best, unsigned long long& second)
{
    best = kNoKey; second = kNoKey;
    if (cnt > 0) {
        const int total = B.pt_cum[64];
        const int chunk = (total + nthreads - 1) / nthreads;
        const int start = B.pt_cum[lane];
        const int l0 = start / chunk, l1 = (start + cnt - 1) / chunk;
        // slots that can hold results of this point: its first and its last thread, and the first thread of every wave in between
        // (block_items: the threads in between only walked this point and handed their results to the head of their run)
        // The first and the last thread always (two slots, in flight together); the wave starts in between only when the point spans
        // a wave boundary -- rare with small windows, where this merge is most of what is left on wave 0's serial path.
        const int q1 = max(l1, l0) + lane;
        unsigned long long k1 = S.part1[l0 + lane], k2 = S.part2[l0 + lane];
        unsigned long long b1 = S.part1[q1], b2 = S.part2[q1];
        if (l1 > l0) { if (b1 < k1) { k2 = min(k1, b2); k1 = b1; } else { k2 = min(k2, b1); } }
        for (int m_ = ((l0 >> 6) + 1) << 6; m_ < l1; m_ += 64) {
            b1 = S.part1[m_ + lane]; b2 = S.part2[m_ + lane];
            if (b1 < k1) { k2 = min(k1, b2); k1 = b1; } else { k2 = min(k2, b1); }
        }
        best = k1; second = k2;
    }
}

template <bool STAGE>
__global__ __launch_bounds__(kProjThreads) void k_proj_par(const ProjArgs* __restrict__ jobs)
{
    static_assert(kProjThreads + 64 <= kBlkSlots, "ProjBlockLds::part1 / part2");
    const ProjArgs A = jobs[blockIdx.x];        // one workgroup per job (frame)
    extern __shared__ __align__(16) uint8_t s_dyn[];
    __shared__ int s_hist[HISTO_LENGTH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = tid >> 6;                    // wave 0 resolves a block, wave 1 sets up the next but one, waves 2-7 walk the next one's item list
    const bool w0 = wv == 0;
    ProjFrameDev F = A.F;
    const int n = F.n;
    const size_t tab = ((size_t)n + 15) & ~(size_t)15;
    int* s_claim = (int*)s_dyn;                 // per feature: first pending lane that will take it out of the pool
    int* s_last = s_claim + tab;                // per feature: last lane of the committing range that writes it
    uint8_t* s_occ = s_dyn + 8 * tab;
    const uint4* s_recs = nullptr;              // key points in grid order (staged frames only)
    if (STAGE) {
        const int ncell = F.cols * F.rows;
        size_t off = tab;
        // (x, y and the CSR's feature list are NOT staged on their own: every search reads them from the 16-byte records in grid order)
        uint8_t* s_desc = s_occ + off; off += (size_t)n * 32;
        int32_t* s_oct = (int32_t*)(s_occ + off); off += (size_t)n * 4;
        float* s_ur = (float*)(s_occ + off); off += F.u_right ? (size_t)n * 4 : 0;
        int32_t* s_coff = (int32_t*)(s_occ + off);
        for (int i = tid; i < n * 8; i += kProjThreads) ((uint32_t*)s_desc)[i] = ((const uint32_t*)F.desc)[i];
        for (int i = tid; i < n; i += kProjThreads) s_oct[i] = F.octave[i];
        if (F.u_right) { for (int i = tid; i < n; i += kProjThreads) s_ur[i] = F.u_right[i]; F.u_right = s_ur; }
        const int nfeat_cells = F.cell_off[ncell];
        for (int i = tid; i <= ncell; i += kProjThreads) s_coff[i] = F.cell_off[i];
        off += ((size_t)ncell + 1) * 4;
        off = (off + 15) & ~(size_t)15;
        uint4* r4 = (uint4*)(s_occ + off);
        for (int e = tid; e < nfeat_cells; e += kProjThreads) {
            const int idx = F.cell_feat[e];
            r4[e] = make_uint4(__float_as_uint(F.x[idx]), __float_as_uint(F.y[idx]), (unsigned)F.octave[idx], (unsigned)idx);
        }
        s_recs = r4;
        F.desc = s_desc; F.octave = s_oct; F.cell_off = s_coff;
    }
    __shared__ float s_scale[32];
    __shared__ int s_col[32];
    __shared__ ProjBlockTab s_tab[2];
    __shared__ ProjBlockPart s_part;
    if (F.n_levels > 0 && F.n_levels <= 32) {
        if (tid < F.n_levels) s_scale[tid] = F.scale_factors[tid];
        F.scale_factors = s_scale;
    }
    for (int i = tid; i < n; i += kProjThreads) { s_occ[i] = A.occupied[i]; s_claim[i] = 0x7FFFFFFF; s_last[i] = -1; }
    if (tid < HISTO_LENGTH) s_hist[tid] = 0;
    __syncthreads();
    int nmatches = 0, nlog = 0;
    const bool bFactor = A.th != 1.0f;
    const int mode = A.last_frame_mode;
    const bool ori = mode == 1 && A.check_ori;
    // acceptance of a (best, second) pair: TH_HIGH / ORBdist / TH_LOW * ratio (:122, :1844, :1967, :522), ratio test of the map-point search (:126-127)
    auto accept = [&](unsigned long long kb, unsigned long long ks) -> bool {
        if (kb == kNoKey) return false;
        const int bestDist = key_dist(kb);
        if ((float)bestDist > A.dist_th) return false;
        if (!mode) {
            const int bestDist2 = (ks == kNoKey) ? 256 : key_dist(ks);
            const int bestLevel = F.octave[key_idx(kb)];
            const int bestLevel2 = (ks == kNoKey) ? -1 : F.octave[key_idx(ks)];
            if (bestLevel == bestLevel2 && (float)bestDist > A.nnratio * (float)bestDist2) return false;
        }
        return true;
    };
#ifdef ORBM_PROJ_TIMING      // (sums in registers, written once at the end: a global read-modify-write per tick would be most of what it measures)
    long long t_blk = clock64(), c_a = 0, c_b = 0, c_merge = 0, c_work = 0;
    int c_pass = 0, c_again = 0, c_walk = 0;
#endif
    // Software pipeline over the blocks, three stages deep: while wave 0 RESOLVES block b, waves 2-7 already walk the item list of block
    // b + 1 and wave 1 SETS UP block b + 2 (its points' global loads and window -> runs tables, into the table block b just left).
    // The early search is sound because the speculation never needed an exact snapshot: occupancy only grows while the blocks run, a
    // feature the search sees occupied stays so, and one it sees free but that is taken meanwhile turns up in the resolution as "best /
    // second best candidate occupied" -- the dirty rule -- and is searched again against the exact occupancy.  Only the merge of a block's
    // partial results (a few LDS reads per point) is left on wave 0's serial path between the two barriers of a block.
    struct PointRegs { bool live, irregular; float x, y, r, pur; int minLevel, maxLevel, occval, cnt; unsigned long long dq[4]; };
    // A block's set-up is two steps: load_block reads the points' arrays from global memory (unconditionally, at a clamped index: one
    // round of independent loads, no load behind the `valid` test), build_block turns them into the block's table.  Wave 1 issues the
    // loads of block b + 3 right after it has built block b + 2, so they are in flight across the barriers and no set-up waits for
    // global memory (with the loads inside the set-up it was the LONGEST job of the overlapped phase on TrackLocalMap's 2 000 points).
    struct RawPoint { bool in_range; unsigned char valid, bad, has_obs; float u, v, ur, view_cos, depth; int level; unsigned long long dq[4]; };
    auto load_block = [&](const int base) __attribute__((always_inline)) -> RawPoint {
        RawPoint R;
        const int i = base + lane, ic = min(i, A.n_pts - 1);          // (only called with n_pts > 0)
        R.in_range = i < A.n_pts;
        R.valid = A.valid[ic]; R.u = A.u[ic]; R.v = A.v[ic]; R.level = A.level[ic];
        R.ur = A.F.u_right ? A.ur[ic] : 0.f;
        R.bad = 0; R.view_cos = 0.f; R.depth = 0.f;
        if (!mode) { R.bad = A.bad[ic]; R.view_cos = A.view_cos[ic]; R.depth = A.far_points ? A.depth[ic] : 0.f; }
        R.has_obs = A.has_obs ? A.has_obs[ic] : (unsigned char)1;
        const unsigned long long* dp = (const unsigned long long*)(A.desc + (size_t)ic * 32);
        R.dq[0] = dp[0]; R.dq[1] = dp[1]; R.dq[2] = dp[2]; R.dq[3] = dp[3];
        return R;
    };
    auto build_block = [&](const RawPoint& R, ProjBlockTab& T) __attribute__((always_inline)) {       // one wave: the points of a block -> runs (block_runs)
        PointRegs P;
        P.irregular = false; P.x = 0.f; P.y = 0.f; P.r = 0.f; P.pur = 0.f; P.minLevel = 0; P.maxLevel = 0; P.occval = 1; P.cnt = 0;
        P.dq[0] = 0; P.dq[1] = 0; P.dq[2] = 0; P.dq[3] = 0;
        P.live = R.in_range && R.valid != 0;
        if (P.live) {
            P.x = R.u; P.y = R.v;
            const int lvl = R.level;
            P.pur = R.ur;
            if (!mode) {
                if ((A.far_points && R.depth > A.th_far) || R.bad) P.live = false;
                P.r = ((double)R.view_cos > 0.998) ? 2.5f : 4.0f;             // RadiusByViewingCos (:215-221)
                if (bFactor) P.r *= A.th;
                P.r = P.r * F.scale_factors[lvl];
                P.minLevel = lvl - 1; P.maxLevel = lvl;
            } else if (mode == 1) {
                if (P.x < F.min_x || P.x > F.max_x || P.y < F.min_y || P.y > F.max_y) P.live = false;      // :1711-1714, :1917-1920
                P.r = A.th * F.scale_factors[lvl];
                if (A.level_window == ORBM_LEVELS_FORWARD) { P.minLevel = lvl; P.maxLevel = -1; }          // :1729
                else if (A.level_window == ORBM_LEVELS_BACKWARD) { P.minLevel = 0; P.maxLevel = lvl; }     // :1731
                else { P.minLevel = lvl - 1; P.maxLevel = lvl + 1; }                                        // :1733, :1938
            } else {
                P.r = A.th * F.scale_factors[lvl];                          // :489
                P.minLevel = lvl - 1; P.maxLevel = lvl;                        // :509
            }
            P.dq[0] = R.dq[0]; P.dq[1] = R.dq[1]; P.dq[2] = R.dq[2]; P.dq[3] = R.dq[3];
            P.occval = R.has_obs;
        }
        T.dq[lane][0] = P.dq[0]; T.dq[lane][1] = P.dq[1]; T.dq[lane][2] = P.dq[2]; T.dq[lane][3] = P.dq[3];
        P.cnt = block_runs(F, T, lane, P.live, P.x, P.y, P.r, P.minLevel, P.maxLevel, P.pur, P.irregular);
        T.flags[lane] = (P.live ? 1 : 0) | (P.irregular ? 2 : 0) | ((P.occval & 0xFF) << 8);
    };
    RawPoint raw_next;          // wave 1: the loaded points of the block it builds next
    raw_next.in_range = false; raw_next.valid = 0; raw_next.bad = 0; raw_next.has_obs = 1; raw_next.u = 0.f; raw_next.v = 0.f; raw_next.ur = 0.f;
    raw_next.view_cos = 0.f; raw_next.depth = 0.f; raw_next.level = 0; raw_next.dq[0] = 0; raw_next.dq[1] = 0; raw_next.dq[2] = 0; raw_next.dq[3] = 0;
    int search_threads = kProjThreads;          // block 0 is walked by all eight waves, the later ones by six (wave 0 resolves, wave 1 sets up beside them)
    if (A.n_pts > 0) {
        if (wv == 0) build_block(load_block(0), s_tab[0]);
        else if (wv == 1) {
            if (64 < A.n_pts) build_block(load_block(64), s_tab[1]);
            if (128 < A.n_pts) raw_next = load_block(128);
        }
        __syncthreads();
        block_items(F, s_recs, s_occ, s_tab[0], s_part, tid, kProjThreads);
        __syncthreads();
    }
    int bi = 0;
    for (int base = 0; base < A.n_pts; base += 64, bi ^= 1) {
        const int i = base + lane;
        const bool has_next = base + 64 < A.n_pts, has_next2 = base + 128 < A.n_pts;
        ProjBlockTab& T = s_tab[bi];
        PointRegs cur;
        cur.live = false; cur.irregular = false; cur.cnt = 0;
        unsigned long long kb = kNoKey, ks = kNoKey;
#ifdef ORBM_PROJ_TIMING
        const long long t_m0 = clock64();
#endif
        if (w0) {               // the block's points back from its table (another wave set it up) and the merge of the partial results
            const int fl = T.flags[lane];
            cur.live = (fl & 1) != 0; cur.irregular = (fl & 2) != 0; cur.occval = (fl >> 8) & 0xFF;
            cur.x = T.prm[lane][0]; cur.y = T.prm[lane][1]; cur.r = T.prm[lane][2]; cur.pur = T.prm[lane][3];
            cur.minLevel = T.lvl[lane][0]; cur.maxLevel = T.lvl[lane][1];
            cur.dq[0] = T.dq[lane][0]; cur.dq[1] = T.dq[lane][1]; cur.dq[2] = T.dq[lane][2]; cur.dq[3] = T.dq[lane][3];
            cur.cnt = T.pt_cum[lane + 1] - T.pt_cum[lane];
            block_merge(T, s_part, lane, cur.cnt, search_threads, kb, ks);     // (ks: the ratio test's second best in mode 0, the stand-in for a taken best otherwise)
        }
#ifdef ORBM_PROJ_TIMING
        c_merge += clock64() - t_m0;
#endif
        __syncthreads();        // the table and the partial results have been read: both may be written again
#ifdef ORBM_PROJ_TIMING
        { const long long t_now = clock64(); c_a += t_now - t_blk; t_blk = t_now; }
#endif
        if (wv == 1) {
            if (has_next2) build_block(raw_next, T);
            if (base + 192 < A.n_pts) raw_next = load_block(base + 192);       // (in flight until the next block's turn)
        } else if (!w0) {
            // ---- speculative search of the NEXT block (waves 2-7), occupancy as it stands while wave 0 resolves this one ----
            if (has_next) block_items(F, s_recs, s_occ, s_tab[bi ^ 1], s_part, tid - 128, kProjThreads - 128);
        } else {
            const bool live = cur.live, irregular = cur.irregular;
            const float x = cur.x, y = cur.y, r = cur.r, pur = cur.pur;
            const int minLevel = cur.minLevel, maxLevel = cur.maxLevel, occval = cur.occval;
            const unsigned long long dq[4] = {cur.dq[0], cur.dq[1], cur.dq[2], cur.dq[3]};
            bool acc = live && accept(kb, ks);
            int bf = acc ? key_idx(kb) : -1;
            // ---- resolution in point order (this wave alone: its LDS traffic executes in order) ----
            int committed = 0;
            for (;;) {
                const bool pend = live && lane >= committed;
                const bool osets = pend && acc && occval != 0;
                if (osets) atomicMin(&s_claim[bf], lane);
                wave_lds_sync();
                bool dirty = pend && irregular;            // (a window the block search did not take: searched by the wave in its turn)
                if (pend && !dirty && kb != kNoKey) {
                    const int f1 = key_idx(kb);
                    dirty = s_occ[f1] != 0 || s_claim[f1] < lane;
                    if (!dirty && !mode && ks != kNoKey) { const int f2 = key_idx(ks); dirty = s_occ[f2] != 0 || s_claim[f2] < lane; }
                }
                const unsigned long long dm = __ballot(dirty);
                const int d = dm ? (int)__ffsll((long long)dm) - 1 : 64;
                wave_lds_sync();                            // every lane has read the claims
                if (osets) s_claim[bf] = 0x7FFFFFFF;
                const bool cm = pend && acc && lane < d;    // clean lanes in front of the first dirty one: final
                if (cm) atomicMax(&s_last[bf], lane);
                wave_lds_sync();
                const unsigned long long cmm = __ballot(cm);
                if (cm) {
                    if (s_last[bf] == lane) { A.assign[bf] = i; s_occ[bf] = (uint8_t)occval; }     // same-feature writers: the last in point order
                    if (ori) {          // the rotation log keeps (feature, point); bins and histogram follow after the blocks, by every thread
                        const int at = nlog + __popcll(cmm & ((1ull << lane) - 1ull));      // (two angle loads from global memory per commit were a third of a resolution pass)
                        A.log_feat[at] = bf; A.log_bin[at] = i;
                    }
                }
                wave_lds_sync();
                if (cm) s_last[bf] = -1;
                const int ncm = __popcll(cmm);
                nmatches += ncm; nlog += ncm;
#ifdef ORBM_PROJ_TIMING
                c_pass++;
#endif
                if (d == 64) break;
#ifdef ORBM_PROJ_TIMING
                c_again++;
#endif
                // ---- the first dirty point again, by the whole wave against the exact occupancy ----
                const float xd = __shfl(x, d), yd = __shfl(y, d), rd = __shfl(r, d), purd = __shfl(pur, d);
                const int minLd = __shfl(minLevel, d), maxLd = __shfl(maxLevel, d), occd = __shfl(occval, d);
                const unsigned long long dqd[4] = {shfl64(dq[0], d), shfl64(dq[1], d), shfl64(dq[2], d), shfl64(dq[3], d)};
                unsigned long long kbd = kNoKey, ksd = kNoKey;
                // Where only the best candidate counts (every mode but the map-point search with its ratio test) the speculative second
                // best IS the answer when it is still free: the candidates a search sees only shrink (occupancy grows), the best one has
                // been taken, so the smallest key of the rest is the second -- no window walk.  (No second at all: no candidate is left.)
                bool again = true;
                if (mode && !(__shfl((int)irregular, d) != 0)) {
                    const unsigned long long k2 = shfl64(ks, d);
                    if (k2 == kNoKey) again = false;
                    else if (s_occ[key_idx(k2)] == 0) { kbd = k2; again = false; }
                }
                if (again) search_window<true>(F, s_occ, xd, yd, rd, minLd, maxLd, purd, dqd, lane, s_col, kbd, ksd, s_recs);
#ifdef ORBM_PROJ_TIMING
                if (again) c_walk++;
#endif
                if (accept(kbd, ksd)) {
                    const int f = key_idx(kbd);
                    if (lane == 0) {
                        A.assign[f] = base + d;
                        s_occ[f] = (uint8_t)occd;
                        if (ori) { A.log_feat[nlog] = f; A.log_bin[nlog] = base + d; }
                    }
                    nmatches++; nlog++;
                }
                wave_lds_sync();
                committed = d + 1;
            }
        }
#ifdef ORBM_PROJ_TIMING
        c_work += clock64() - t_blk;        // this wave's own work of the overlapped phase (wave 0 resolution, wave 1 set-up, waves 2-7 item list)
#endif
        __syncthreads();            // the block's occupancy is final for everybody, the next block's list has been walked
#ifdef ORBM_PROJ_TIMING
        { const long long t_now = clock64(); c_b += t_now - t_blk; t_blk = t_now; }
#endif
        search_threads = kProjThreads - 128;
    }
#ifdef ORBM_PROJ_TIMING
    if (blockIdx.x == 0) {      // [0] wave 1 set-up, [1] wave 0 merge, [2] wave 0 resolution, [3] passes, [4] points searched again, [5] serial phase, [6] overlapped phase, [7] wave 2 item list
        if (tid == 0) { d_proj_prof[1] = c_merge; d_proj_prof[2] = c_work; d_proj_prof[3] = c_pass; d_proj_prof[4] = c_again | ((unsigned long long)c_walk << 32); d_proj_prof[5] = c_a; d_proj_prof[6] = c_b; }
        if (tid == 64) d_proj_prof[0] = c_work;
        if (tid >= 128 && lane == 0) atomicMax(&d_proj_prof[7], (unsigned long long)c_work);        // (the slowest of the searching waves; the slot is zeroed by the read-out)
    }
#endif
    if (ori) {
        // rotation consistency (:1868-1884) by the whole workgroup: bins of the logged matches, histogram, the three maxima, and the
        // matches outside them taken back (only -1 / 0 are written here, so the order of the log does not matter)
        __shared__ int s_rot[5];                // log length, the three maxima, matches taken back
        if (tid == 0) { s_rot[0] = nlog; s_rot[4] = 0; }
        __threadfence_block();
        __syncthreads();                        // (the log was written to global memory by wave 0: visible to the workgroup behind the barrier)
        const int n_log = s_rot[0];
        for (int k = tid; k < n_log; k += kProjThreads) {
            const int bin = rot_bin(A.angle[A.log_bin[k]], F.angle[A.log_feat[k]]);
            A.log_bin[k] = bin;                 // (the same thread reads it back below)
            atomicAdd(&s_hist[bin], 1);
        }
        __syncthreads();
        if (tid == 0) { int i1, i2, i3; three_maxima(s_hist, HISTO_LENGTH, i1, i2, i3); s_rot[1] = i1; s_rot[2] = i2; s_rot[3] = i3; }
        __syncthreads();
        const int i1 = s_rot[1], i2 = s_rot[2], i3 = s_rot[3];
        for (int k = tid; k < n_log; k += kProjThreads) {
            const int b_ = A.log_bin[k];
            if (b_ != i1 && b_ != i2 && b_ != i3) {
                const int f = A.log_feat[k];
                A.assign[f] = -1;               // CurrentFrame.mvpMapPoints[...] = NULL (:1878)
                s_occ[f] = 0;
                atomicAdd(&s_rot[4], 1);
            }
        }
        __syncthreads();
        nmatches -= s_rot[4];
    }
    __syncthreads();
    for (int i = tid; i < n; i += kProjThreads) A.occupied[i] = s_occ[i];
    if (tid == 0) *A.n_matches = nmatches;
}

// ---- device-resident batch of SearchByProjection(CurrentFrame, LastFrame) ------------------------------------------------
// The frames are what orbx_extract_batch_device left in HBM ([batch][cap] OrbxKeyPoint records + descriptors + counts): one
// setup launch turns them into the SoA arrays the search kernels read and writes the job table ON THE DEVICE (the counts never
// visit the host); k_grid builds Frame::AssignFeaturesToGrid, k_proj_par searches.
struct ProjDevSetup {
    const OrbxKeyPoint* kps; const uint8_t* desc; const int32_t* n; int32_t cap;
    const uint8_t* p_valid; const float* p_u; const float* p_v; const int32_t* p_oct; const float* p_angle; const uint8_t* p_desc; const int32_t* p_n; int32_t p_cap; const uint8_t* p_obs;
    const float* p_vcos; const float* p_depth; const uint8_t* p_bad; float th_far, nnratio; int32_t far_points, mode;      // Frame x map points (mode 0)
    float* x; float* y; int32_t* oct; float* ang; int32_t* cell_off; int32_t* cell_feat; int32_t* log_feat; int32_t* log_bin;
    const float* scale; int32_t n_levels, cols, rows;
    float min_x, min_y, max_x, max_y, th, dist_th;
    int32_t check_ori, lds_frame;
    int32_t* assign; uint8_t* occupied; int32_t* n_matches;
    ProjArgs* jobs;
};
__global__ __launch_bounds__(256) void k_proj_dev_setup(ProjDevSetup P)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(max(P.n[b], 0), P.cap);
    const OrbxKeyPoint* kp = P.kps + (size_t)b * P.cap;
    float* x = P.x + (size_t)b * P.cap; float* y = P.y + (size_t)b * P.cap; float* ang = P.ang + (size_t)b * P.cap;
    int32_t* oct = P.oct + (size_t)b * P.cap;
    for (int i = tid; i < n; i += 256) { const OrbxKeyPoint k = kp[i]; x[i] = k.x; y[i] = k.y; oct[i] = k.octave; ang[i] = k.angle; }
    if (tid == 0) {
        ProjArgs A;
        A.F.x = x; A.F.y = y; A.F.octave = oct; A.F.angle = ang; A.F.desc = P.desc + (size_t)b * P.cap * 32;
        A.F.cell_off = P.cell_off + (size_t)b * ((size_t)P.cols * P.rows + 1); A.F.cell_feat = P.cell_feat + (size_t)b * P.cap;
        A.F.scale_factors = P.scale;
        A.F.min_x = P.min_x; A.F.min_y = P.min_y; A.F.max_x = P.max_x; A.F.max_y = P.max_y;
        A.F.winv = (float)P.cols / (P.max_x - P.min_x); A.F.hinv = (float)P.rows / (P.max_y - P.min_y);        // mfGridElementWidthInv / HeightInv
        A.F.n = n; A.F.cols = P.cols; A.F.rows = P.rows; A.F.n_levels = P.n_levels; A.F.u_right = nullptr;
        A.n_pts = min(max(P.p_n[b], 0), P.p_cap);
        A.valid = P.p_valid + (size_t)b * P.p_cap; A.u = P.p_u + (size_t)b * P.p_cap; A.v = P.p_v + (size_t)b * P.p_cap; A.ur = nullptr;
        A.level = P.p_oct + (size_t)b * P.p_cap;
        A.view_cos = P.p_vcos ? P.p_vcos + (size_t)b * P.p_cap : nullptr; A.depth = P.p_depth ? P.p_depth + (size_t)b * P.p_cap : nullptr;
        A.bad = P.p_bad ? P.p_bad + (size_t)b * P.p_cap : nullptr;
        A.angle = P.p_angle ? P.p_angle + (size_t)b * P.p_cap : nullptr; A.desc = P.p_desc + (size_t)b * P.p_cap * 32; A.has_obs = P.p_obs ? P.p_obs + (size_t)b * P.p_cap : nullptr;
        A.th = P.th; A.th_far = P.th_far; A.nnratio = P.nnratio; A.dist_th = P.dist_th; A.far_points = P.far_points; A.check_ori = P.check_ori;
        A.last_frame_mode = P.mode; A.level_window = ORBM_LEVELS_AROUND;
        A.assign = P.assign + (size_t)b * P.cap; A.occupied = P.occupied + (size_t)b * P.cap;
        A.log_feat = P.log_feat + (size_t)b * P.p_cap; A.log_bin = P.log_bin + (size_t)b * P.p_cap;
        A.n_matches = P.n_matches + b;
        A.lds_frame = P.lds_frame;
        P.jobs[b] = A;
    }
}

// ---- search core of ORBmatcher::Fuse (src/ORBmatcher.cc:1148-1338 and :1340-1455) ------------------------------
// The candidate map points are independent of each other (nothing the loop writes feeds back into the search), so this
// is one wave per map point: 64 lanes walk the cells of the window, the first-minimum in visiting order comes out of the
// same lexicographic key reduction the sequential searches use.
struct FuseArgs {
    ProjFrameDev F;
    const float* u_right;           // pKF->mvuRight
    const float* inv_sigma2;        // pKF->mvInvLevelSigma2
    int32_t n_pts;
    const uint8_t* valid;
    const float* u; const float* v; const float* ur;
    const int32_t* level;
    const uint8_t* desc;
    float th;
    int32_t chi2_check;
    int32_t* best_idx; int32_t* best_dist;
};

__global__ __launch_bounds__(256) void k_fuse(FuseArgs A)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= A.n_pts) return;
    const ProjFrameDev& F = A.F;
    unsigned long long k1 = kNoKey;
    if (A.valid[i]) {
        const int lvl = A.level[i];
        const float x = A.u[i], y = A.v[i];
        const float r = A.th * F.scale_factors[lvl];                  // :1242
        const int nMinCellX = max(0, (int)floorf((x - F.min_x - r) * F.winv));
        const int nMaxCellX = min(F.cols - 1, (int)ceilf((x - F.min_x + r) * F.winv));
        const int nMinCellY = max(0, (int)floorf((y - F.min_y - r) * F.hinv));
        const int nMaxCellY = min(F.rows - 1, (int)ceilf((y - F.min_y + r) * F.hinv));
        if (nMinCellX < F.cols && nMaxCellX >= 0 && nMinCellY < F.rows && nMaxCellY >= 0) {
            const uint8_t* dmp = A.desc + (size_t)i * 32;
            const int ny = nMaxCellY - nMinCellY + 1, nx = nMaxCellX - nMinCellX + 1;
            for (int c = lane; c < nx * ny; c += 64) {
                const int ix = nMinCellX + c / ny, iy = nMinCellY + c % ny;
                const int cell = ix * F.rows + iy;
                const int e0 = F.cell_off[cell], e1 = F.cell_off[cell + 1];
                for (int e = e0; e < e1; e++) {
                    const int idx = F.cell_feat[e];
                    const float kpx = F.x[idx], kpy = F.y[idx];
                    const float distx = kpx - x, disty = kpy - y;
                    if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;   // KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:704-748)
                    const int kpLevel = F.octave[idx];
                    if (kpLevel < lvl - 1 || kpLevel > lvl) continue;        // :1265
                    if (A.chi2_check) {
                        const float kpr = A.u_right[idx];
                        const float ex = x - kpx, ey = y - kpy;
                        if (kpr >= 0) {
                            const float er = A.ur[i] - kpr;
                            const float e2 = ex * ex + ey * ey + er * er;
                            if ((double)(e2 * A.inv_sigma2[kpLevel]) > 7.8) continue;     // :1278
                        } else {
                            const float e2 = ex * ex + ey * ey;
                            if ((double)(e2 * A.inv_sigma2[kpLevel]) > 5.99) continue;    // :1289
                        }
                    }
                    const int dist = hamming256(dmp, F.desc + (size_t)idx * 32);
                    if (dist >= 256) continue;
                    k1 = min(k1, make_key(dist, c, e - e0, idx));
                }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) k1 = min(k1, (unsigned long long)__shfl_xor(k1, o));
    if (lane == 0) {
        A.best_idx[i] = (k1 == kNoKey) ? -1 : key_idx(k1);
        A.best_dist[i] = (k1 == kNoKey) ? 256 : key_dist(k1);
    }
}

// ---- SearchForTriangulation (src/ORBmatcher.cc:907-1146), conventional cameras ---------------------------------
// The reference never sets vbMatched2, so every KF1 feature is matched on its own: one thread per entry of KF1's feature
// vector walks the KF2 features of the same vocabulary node in order with the reference's `dist > bestDist -> continue`
// rule (the last of the equally good candidates that pass the epipolar test wins).
struct TriArgs {
    const uint8_t* d1; const uint8_t* mp1; const uint8_t* st1; const float* x1; const float* y1; const float* a1;
    const uint32_t* node1; const int32_t* off1; const uint32_t* feat1;
    const uint8_t* d2; const uint8_t* mp2; const uint8_t* st2; const float* x2; const float* y2; const int32_t* oct2; const float* a2;
    const uint32_t* node2; const int32_t* off2; const uint32_t* feat2;
    const float* sigma2_2; const float* scale2;
    float F12[9];
    float ep_x, ep_y;
    int32_t n1, nn1, nn2, only_stereo, coarse, check_ori;
    int32_t* match12; int32_t* n_matches;
};

__global__ __launch_bounds__(256) void k_triangulation(TriArgs A)
{
    __shared__ int s_hist[HISTO_LENGTH];
    __shared__ int s_keep[3];
    __shared__ int s_count;
    const int tid = threadIdx.x;
    for (int i = tid; i < A.n1; i += 256) A.match12[i] = -1;
    if (tid < HISTO_LENGTH) s_hist[tid] = 0;
    if (tid == 0) s_count = 0;
    __syncthreads();
    const int total1 = A.nn1 > 0 ? A.off1[A.nn1] : 0;
    for (int e1 = tid; e1 < total1; e1 += 256) {
        int lo = 0, hi = A.nn1;                                 // node of entry e1: last k with off1[k] <= e1
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.off1[mid] <= e1) lo = mid; else hi = mid; }
        const uint32_t key = A.node1[lo];
        int l2 = 0, h2 = A.nn2;
        while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (A.node2[mid] < key) l2 = mid + 1; else h2 = mid; }
        if (l2 >= A.nn2 || A.node2[l2] != key) continue;
        const int idx1 = (int)A.feat1[e1];
        if (A.mp1[idx1]) continue;                              // already a MapPoint (:968)
        const bool bStereo1 = A.st1[idx1] != 0;
        if (A.only_stereo && !bStereo1) continue;
        const uint8_t* dd1 = A.d1 + (size_t)idx1 * 32;
        const float la = A.x1[idx1] * A.F12[0] + A.y1[idx1] * A.F12[3] + A.F12[6];      // Pinhole::epipolarConstrain (Pinhole.cpp:115-117)
        const float lb = A.x1[idx1] * A.F12[1] + A.y1[idx1] * A.F12[4] + A.F12[7];
        const float lc = A.x1[idx1] * A.F12[2] + A.y1[idx1] * A.F12[5] + A.F12[8];
        const float den = la * la + lb * lb;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int e2 = A.off2[l2]; e2 < A.off2[l2 + 1]; e2++) {
            const int idx2 = (int)A.feat2[e2];
            if (A.mp2[idx2]) continue;
            const bool bStereo2 = A.st2[idx2] != 0;
            if (A.only_stereo && !bStereo2) continue;
            const int dist = hamming256(dd1, A.d2 + (size_t)idx2 * 32);
            if (dist > TH_LOW || dist > bestDist) continue;
            const float kx = A.x2[idx2], ky = A.y2[idx2];
            const int o2 = A.oct2[idx2];
            if (!bStereo1 && !bStereo2) {
                const float distex = A.ep_x - kx, distey = A.ep_y - ky;
                if (distex * distex + distey * distey < 100 * A.scale2[o2]) continue;  // too close to the epipole (:1011)
            }
            bool ok = A.coarse != 0;
            if (!ok && den != 0) {
                const float num = la * kx + lb * ky + lc;
                const float dsqr = num * num / den;
                ok = (double)dsqr < 3.84 * (double)A.sigma2_2[o2];
            }
            if (ok) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            A.match12[idx1] = bestIdx2;
            if (A.check_ori) atomicAdd(&s_hist[rot_bin(A.a1[idx1], A.a2[bestIdx2])], 1);
        }
    }
    __threadfence_block();
    __syncthreads();
    if (A.check_ori) {
        if (tid == 0) { int a, b, c; three_maxima(s_hist, HISTO_LENGTH, a, b, c); s_keep[0] = a; s_keep[1] = b; s_keep[2] = c; }
        __syncthreads();
    }
    int local = 0;
    for (int i = tid; i < A.n1; i += 256) {
        const int m = A.match12[i];
        if (m < 0) continue;
        if (A.check_ori) {
            const int bin = rot_bin(A.a1[i], A.a2[m]);
            if (bin != s_keep[0] && bin != s_keep[1] && bin != s_keep[2]) { A.match12[i] = -1; continue; }
        }
        local++;
    }
    if (local) atomicAdd(&s_count, local);
    __syncthreads();
    if (tid == 0) *A.n_matches = s_count;
}

// ---- SearchForInitialization (src/ORBmatcher.cc:648-763) ------------------------------------------------------
// Sequential by construction (vMatchedDistance / vnMatches21 feed back into later windows): one wave, F1 features in
// order, 64 lanes over the window; the state lives in LDS.
struct InitArgs {
    ProjFrameDev F2;
    const uint8_t* d1; const int32_t* oct1; const float* a1; const float* prev_x; const float* prev_y;
    int32_t n1, window, check_ori;
    float nnratio;
    int32_t* match12; int32_t* n_matches;
};

__global__ __launch_bounds__(64) void k_initialization(InitArgs A)
{
    extern __shared__ int s_state[];        // [n2] matched distance, [n2] vnMatches21
    __shared__ int s_hist[HISTO_LENGTH];
    const int lane = threadIdx.x;
    const ProjFrameDev& F = A.F2;
    int* s_mdist = s_state;
    int* s_m21 = s_state + F.n;
    for (int i = lane; i < F.n; i += 64) { s_mdist[i] = 0x7FFFFFFF; s_m21[i] = -1; }
    for (int i = lane; i < A.n1; i += 64) A.match12[i] = -1;
    if (lane < HISTO_LENGTH) s_hist[lane] = 0;
    __threadfence_block();
    __syncthreads();
    int nmatches = 0;
    const float r = (float)A.window;
    for (int i1 = 0; i1 < A.n1; i1++) {
        if (A.oct1[i1] > 0) continue;                                       // level1 > 0 (:666)
        const float x = A.prev_x[i1], y = A.prev_y[i1];
        const int nMinCellX = max(0, (int)floorf((x - F.min_x - r) * F.winv));
        if (nMinCellX >= F.cols) continue;
        const int nMaxCellX = min(F.cols - 1, (int)ceilf((x - F.min_x + r) * F.winv));
        if (nMaxCellX < 0) continue;
        const int nMinCellY = max(0, (int)floorf((y - F.min_y - r) * F.hinv));
        if (nMinCellY >= F.rows) continue;
        const int nMaxCellY = min(F.rows - 1, (int)ceilf((y - F.min_y + r) * F.hinv));
        if (nMaxCellY < 0) continue;
        const uint8_t* dd1 = A.d1 + (size_t)i1 * 32;
        const int ny = nMaxCellY - nMinCellY + 1, nx = nMaxCellX - nMinCellX + 1;
        unsigned long long k1 = kNoKey, k2 = kNoKey;
        for (int c = lane; c < nx * ny; c += 64) {
            const int ix = nMinCellX + c / ny, iy = nMinCellY + c % ny;
            const int cell = ix * F.rows + iy;
            const int e0 = F.cell_off[cell], e1 = F.cell_off[cell + 1];
            for (int e = e0; e < e1; e++) {
                const int idx = F.cell_feat[e];
                if (F.octave[idx] > 0) continue;                           // GetFeaturesInArea(…, level1, level1) with level1 = 0
                const float distx = F.x[idx] - x, disty = F.y[idx] - y;
                if (!(fabsf(distx) < r && fabsf(disty) < r)) continue;
                const int dist = hamming256(dd1, F.desc + (size_t)idx * 32);
                if (s_mdist[idx] <= dist) continue;                        // :686
                top2_insert(k1, k2, make_key(dist, c, e - e0, idx));
            }
        }
        {   // best / second best of the wave by DPP minimum ladders (see search_window)
            const unsigned long long b1 = wave_min_u64(k1);
            if (k1 == b1) { k1 = k2; k2 = kNoKey; }
            const unsigned long long b2 = (b1 == kNoKey) ? kNoKey : wave_min_u64(k1);
            k1 = b1; k2 = b2;
        }
        if (k1 == kNoKey) continue;
        const int bestDist = key_dist(k1), bestIdx2 = key_idx(k1);
        if (bestDist > TH_LOW) continue;
        const float second = (k2 == kNoKey) ? (float)0x7FFFFFFF : (float)key_dist(k2);
        if (!((float)bestDist < second * A.nnratio)) continue;             // :703
        const int prev = s_m21[bestIdx2];
        if (prev >= 0) nmatches--;
        nmatches++;
        if (lane == 0) {
            if (prev >= 0) A.match12[prev] = -1;
            A.match12[i1] = bestIdx2;
            s_m21[bestIdx2] = i1;
            s_mdist[bestIdx2] = bestDist;
            if (A.check_ori) s_hist[rot_bin(A.a1[i1], F.angle[bestIdx2])]++;
        }
        __threadfence_block();
        __syncthreads();
    }
    __syncthreads();
    if (A.check_ori) {
        // every F1 feature that ever won a window sits in the histogram, stolen or not (:708-718); only live matches are
        // cleared afterwards (:741-745).  A stolen feature keeps its bin count, so the bins are rebuilt from the log-free
        // state: counts come from s_hist (increments above), membership from the angles of the surviving matches.
        int i1a, i2a, i3a;
        three_maxima(s_hist, HISTO_LENGTH, i1a, i2a, i3a);
        int removed = 0;
        for (int i = lane; i < A.n1; i += 64) {
            const int m = A.match12[i];
            if (m < 0) continue;
            const int bin = rot_bin(A.a1[i], F.angle[m]);
            if (bin != i1a && bin != i2a && bin != i3a) { A.match12[i] = -1; removed++; }
        }
        for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
        nmatches -= removed;
    }
    if (lane == 0) *A.n_matches = nmatches;
}

// ---- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-402) for a batch of map points -------------------
// One wave per map point.  The N observed descriptors are staged in LDS; lane i owns row i of the N x N distance matrix
// (rows i, i + 64, ... when N > 64) and finds the row's median -- the reference's sort(vDists)[0.5 * (N - 1)] -- as the
// smallest value v with #{j : d(i, j) <= v} > k by bisection over v in [0, 256] (9 passes over the row, every descriptor
// read is an LDS broadcast), so no N x N buffer and no sort is needed.  The winner is the FIRST row with the least median
// (strict < at :391): a DPP minimum over (median << 16 | row) keys.
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ off, int n_points,
                                                    int32_t* __restrict__ best_idx, int32_t* __restrict__ best_median)
{
    extern __shared__ __align__(16) uint8_t s_dyn[];
    unsigned long long* s_d = (unsigned long long*)s_dyn;              // [N][4]
    const int p = blockIdx.x;
    const int o0 = off[p], N = off[p + 1] - o0;
    if (N <= 0) {
        if (threadIdx.x == 0) { best_idx[p] = -1; best_median[p] = -1; }
        return;
    }
    const unsigned long long* g = (const unsigned long long*)(desc + (size_t)o0 * 32);
    for (int j = threadIdx.x; j < N * 4; j += 64) s_d[j] = g[j];
    __syncthreads();
    const int k = (N - 1) >> 1;                                         // (size_t)(0.5 * (N - 1))
    unsigned key = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < N; i += 64) {
        const unsigned long long a0 = s_d[4 * i], a1 = s_d[4 * i + 1], a2 = s_d[4 * i + 2], a3 = s_d[4 * i + 3];
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < N; j++) {
                const int d = __popcll(a0 ^ s_d[4 * j]) + __popcll(a1 ^ s_d[4 * j + 1]) + __popcll(a2 ^ s_d[4 * j + 2]) + __popcll(a3 ^ s_d[4 * j + 3]);
                c += d <= mid;
            }
            if (c > k) hi = mid; else lo = mid + 1;
        }
        key = min(key, ((unsigned)lo << 16) | (unsigned)i);            // rows of one lane ascend, so min keeps the first
    }
    key = wave_min_u32(key);
    if (threadIdx.x == 0) { best_idx[p] = (int)(key & 0xFFFFu); best_median[p] = (int)(key >> 16); }
}

// ---- MapPoint::UpdateNormalAndDepth (src/MapPoint.cc:433-493) for a batch of map points ----------------------------
// One thread per map point: the observation loop is a sequential float accumulation (normal += normali / |normali|), kept
// in the reference's order and in its float expressions (no FMA contraction in this translation unit).
__global__ __launch_bounds__(256) void k_normal_depth(const float* __restrict__ pos, const float* __restrict__ centers, const int32_t* __restrict__ off,
                                                      const float* __restrict__ ref_center, const float* __restrict__ level_scale,
                                                      float last_scale, int n_points,
                                                      float* __restrict__ normal, float* __restrict__ max_dist, float* __restrict__ min_dist)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_points) return;
    const float px = pos[3 * p], py = pos[3 * p + 1], pz = pos[3 * p + 2];
    float nx = 0.f, ny = 0.f, nz = 0.f;
    const int o0 = off[p], o1 = off[p + 1];
    for (int o = o0; o < o1; o++) {
        const float dx = px - centers[3 * o], dy = py - centers[3 * o + 1], dz = pz - centers[3 * o + 2];
        const float nrm = sqrtf(dx * dx + (dy * dy + dz * dz));           // Eigen's fixed-size redux: x0 + (x1 + x2)
        nx = nx + dx / nrm; ny = ny + dy / nrm; nz = nz + dz / nrm;
    }
    const float cx = px - ref_center[3 * p], cy = py - ref_center[3 * p + 1], cz = pz - ref_center[3 * p + 2];
    const float dist = sqrtf(cx * cx + (cy * cy + cz * cz));
    const float mx = dist * level_scale[p];                            // mfMaxDistance (:489)
    max_dist[p] = mx;
    min_dist[p] = mx / last_scale;                                     // (:490)
    const float cnt = (float)(o1 - o0);
    normal[3 * p] = nx / cnt; normal[3 * p + 1] = ny / cnt; normal[3 * p + 2] = nz / cnt;     // normal / n (:491)
}

}  // namespace orbm

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// growable byte buffer in PINNED host memory: every host-buffer entry point packs its arrays here and moves them with one copy;
// from pageable memory that copy runs at a fraction of the link (24 MB for 256 frames of the last-frame search: 2 ms of its 2.5)
struct PinnedBytes {
    uint8_t* p = nullptr;
    size_t n = 0, cap = 0;
    ~PinnedBytes() { if (p) (void)hipHostFree(p); }
    PinnedBytes() = default;
    PinnedBytes(const PinnedBytes&) = delete;
    PinnedBytes& operator=(const PinnedBytes&) = delete;
    size_t size() const { return n; }
    uint8_t* data() { return p; }
    void clear() { n = 0; }
    void resize(size_t bytes)
    {
        if (bytes > cap) {
            const size_t want = std::max(bytes + bytes / 2, (size_t)1 << 16);
            uint8_t* q = nullptr;
            if (hipHostMalloc((void**)&q, want, hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
            if (p) { std::memcpy(q, p, n); (void)hipHostFree(p); }
            p = q; cap = want;
        }
        if (bytes > n) std::memset(p + n, 0, bytes - n);      // (std::vector semantics: new bytes are zero)
        n = bytes;
    }
};

struct orbm_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    uint8_t* d_blob = nullptr;
    size_t blob_cap = 0;
    PinnedBytes h_blob;
    uint8_t* d_ws = nullptr;            // workspace of the device-resident batch entries (SoA key points, grids, logs, job table)
    size_t ws_cap = 0;
    float* d_scale = nullptr;           // scale factors of the last device-resident call
    float h_scale[32] = {};             // ... staged here: the asynchronous copy must not read the caller's array after the call returned

    int ensure(size_t bytes)
    {
        if (bytes <= blob_cap) return ORBX_OK;
        if (d_blob) (void)hipFree(d_blob);
        d_blob = nullptr; blob_cap = 0;
        const size_t cap = std::max(bytes * 2, (size_t)1 << 20);
        ORBM_HIP(hipMalloc((void**)&d_blob, cap));
        blob_cap = cap;
        return ORBX_OK;
    }
};

namespace {

struct Blob {                   // host-side packer: every array is appended 16-byte aligned, device address = base + offset
    PinnedBytes& buf;
    explicit Blob(PinnedBytes& b) : buf(b) { buf.clear(); }
    size_t put(const void* src, size_t bytes)
    {
        const size_t off = (buf.size() + 15) & ~(size_t)15;
        buf.resize(off + bytes);
        if (src && bytes) std::memcpy(buf.data() + off, src, bytes);
        return off;
    }
    size_t reserve(size_t bytes) { return put(nullptr, bytes); }
};

bool features_unique(const OrbmFeatVec* fv, int n)
{
    std::vector<uint8_t> seen(std::max(n, 1), 0);
    const int total = fv->n_nodes > 0 ? fv->offset[fv->n_nodes] : 0;
    for (int i = 0; i < total; i++) {
        const uint32_t f = fv->feat[i];
        if ((int)f >= n || seen[f]) return false;
        seen[f] = 1;
    }
    return true;
}

int check_fv(const OrbmFeatVec* fv, int n, const char* name)
{
    if (!fv) return fail(ORBX_ERR_ARG, "%s is NULL", name);
    if (fv->n_nodes < 0) return fail(ORBX_ERR_ARG, "%s: negative node count", name);
    if (fv->n_nodes == 0) return ORBX_OK;
    if (!fv->node_id || !fv->offset || (!fv->feat && fv->offset[fv->n_nodes] > 0)) return fail(ORBX_ERR_ARG, "%s: NULL arrays", name);
    for (int k = 0; k < fv->n_nodes; k++) {
        if (fv->offset[k + 1] < fv->offset[k]) return fail(ORBX_ERR_ARG, "%s: offsets not monotone", name);
        if (k > 0 && fv->node_id[k] <= fv->node_id[k - 1]) return fail(ORBX_ERR_ARG, "%s: node ids not ascending", name);
    }
    const int total = fv->offset[fv->n_nodes];
    for (int i = 0; i < total; i++)
        if ((int)fv->feat[i] < 0 || (int)fv->feat[i] >= n) return fail(ORBX_ERR_ARG, "%s: feature index %u out of range", name, fv->feat[i]);
    return ORBX_OK;
}

struct PairOffsets {
    size_t d1, v1, a1, node1, off1, feat1, d2, v2, a2, node2, off2, feat2, match, matched2, nmatch;
    int n1, n2, nn1, nn2, serial, n_out;
};

}  // namespace

template <bool KFKF>
static int bow_batch(orbm_matcher* m, int n_pairs,
                     const uint8_t* const* d1, const int* n1, const uint8_t* const* v1, const float* const* a1, const OrbmFeatVec* const* fv1,
                     const uint8_t* const* d2, const int* n2, const uint8_t* const* v2, const float* const* a2, const OrbmFeatVec* const* fv2,
                     float nnratio, int check_ori, int32_t* const* match_out, int32_t* nmatches_out)
{
    if (!m) return fail(ORBX_ERR_ARG, "NULL matcher");
    ORBM_HIP(hipSetDevice(m->device));
    Blob blob(m->h_blob);
    std::vector<PairOffsets> po(n_pairs);
    const size_t desc_off = blob.reserve(sizeof(orbm::BowPairDev) * n_pairs);
    static const int32_t zero_off[1] = {0};
    for (int p = 0; p < n_pairs; p++) {
        int r;
        if (n1[p] < 0 || n2[p] < 0) return fail(ORBX_ERR_ARG, "negative feature count");
        if ((n1[p] > 0 && (!d1[p] || !v1[p])) || (n2[p] > 0 && !d2[p])) return fail(ORBX_ERR_ARG, "NULL descriptor/valid array");
        if (KFKF && n2[p] > 0 && !v2[p]) return fail(ORBX_ERR_ARG, "NULL valid2");
        if (check_ori && ((n1[p] > 0 && !a1[p]) || (n2[p] > 0 && !a2[p]))) return fail(ORBX_ERR_ARG, "NULL angle array");
        if ((r = check_fv(fv1[p], n1[p], "fv1")) || (r = check_fv(fv2[p], n2[p], "fv2"))) return r;
        PairOffsets& o = po[p];
        o.n1 = n1[p]; o.n2 = n2[p]; o.nn1 = fv1[p]->n_nodes; o.nn2 = fv2[p]->n_nodes;
        o.serial = !features_unique(fv2[p], n2[p]) || (KFKF && !features_unique(fv1[p], n1[p]));
        o.d1 = blob.put(d1[p], (size_t)o.n1 * 32); o.v1 = blob.put(v1[p], o.n1);
        o.a1 = blob.put(a1[p], a1[p] ? sizeof(float) * o.n1 : 0);
        o.node1 = blob.put(fv1[p]->node_id, sizeof(uint32_t) * o.nn1);
        o.off1 = blob.put(o.nn1 ? fv1[p]->offset : zero_off, sizeof(int32_t) * (o.nn1 + 1));
        o.feat1 = blob.put(fv1[p]->feat, sizeof(uint32_t) * (o.nn1 ? fv1[p]->offset[o.nn1] : 0));
        o.d2 = blob.put(d2[p], (size_t)o.n2 * 32); o.v2 = blob.put(KFKF ? v2[p] : nullptr, KFKF ? o.n2 : 0);
        o.a2 = blob.put(a2[p], a2[p] ? sizeof(float) * o.n2 : 0);
        o.node2 = blob.put(fv2[p]->node_id, sizeof(uint32_t) * o.nn2);
        o.off2 = blob.put(o.nn2 ? fv2[p]->offset : zero_off, sizeof(int32_t) * (o.nn2 + 1));
        o.feat2 = blob.put(fv2[p]->feat, sizeof(uint32_t) * (o.nn2 ? fv2[p]->offset[o.nn2] : 0));
        o.n_out = KFKF ? o.n1 : o.n2;
        o.match = blob.reserve(sizeof(int32_t) * std::max(o.n_out, 1));
        o.matched2 = blob.reserve(KFKF ? std::max(o.n2, 1) : 0);
        o.nmatch = blob.reserve(sizeof(int32_t));
    }
    int r = m->ensure(m->h_blob.size());
    if (r) return r;
    uint8_t* base = m->d_blob;
    orbm::BowPairDev* descs = (orbm::BowPairDev*)(m->h_blob.data() + desc_off);
    for (int p = 0; p < n_pairs; p++) {
        const PairOffsets& o = po[p];
        orbm::BowPairDev D;
        D.d1 = base + o.d1; D.v1 = base + o.v1; D.a1 = (const float*)(base + o.a1);
        D.node1 = (const uint32_t*)(base + o.node1); D.off1 = (const int32_t*)(base + o.off1); D.feat1 = (const uint32_t*)(base + o.feat1);
        D.d2 = base + o.d2; D.v2 = base + o.v2; D.a2 = (const float*)(base + o.a2);
        D.node2 = (const uint32_t*)(base + o.node2); D.off2 = (const int32_t*)(base + o.off2); D.feat2 = (const uint32_t*)(base + o.feat2);
        D.n1 = o.n1; D.n2 = o.n2; D.nn1 = o.nn1; D.nn2 = o.nn2;
        D.match = (int32_t*)(base + o.match); D.matched2 = base + o.matched2; D.n_matches = (int32_t*)(base + o.nmatch);
        D.serial = o.serial;
        descs[p] = D;
    }
    ORBM_HIP(hipMemcpyAsync(base, m->h_blob.data(), m->h_blob.size(), hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(orbm::k_bow<KFKF>, dim3(n_pairs), dim3(orbm::kBowThreads), 0, m->stream, (const orbm::BowPairDev*)(base + desc_off), nnratio, check_ori);
    ORBM_HIP(hipGetLastError());
    for (int p = 0; p < n_pairs; p++) {
        const PairOffsets& o = po[p];
        if (o.n_out > 0) ORBM_HIP(hipMemcpyAsync(match_out[p], base + o.match, sizeof(int32_t) * o.n_out, hipMemcpyDeviceToHost, m->stream));
        ORBM_HIP(hipMemcpyAsync(&nmatches_out[p], base + o.nmatch, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    }
    ORBM_HIP(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

// Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:472-503, 812-822): host-side flattening of mGrid into CSR.
static int build_grid(const OrbmFrame* f, std::vector<int32_t>& cell_off, std::vector<int32_t>& cell_feat, float& winv, float& hinv)
{
    if (!f || f->n < 0 || f->grid_cols < 1 || f->grid_rows < 1 || f->grid_cols * (int64_t)f->grid_rows > (1 << 20))
        return fail(ORBX_ERR_ARG, "bad frame grid");
    if (f->n > 0 && (!f->x || !f->y || !f->octave || !f->desc)) return fail(ORBX_ERR_ARG, "NULL frame arrays");
    if (!f->scale_factors || f->n_levels < 1) return fail(ORBX_ERR_ARG, "NULL scale factors");
    if (f->n >= (1 << 21)) return fail(ORBX_ERR_ARG, "too many features");
    winv = (float)f->grid_cols / (f->max_x - f->min_x);         // mfGridElementWidthInv (src/Frame.cc:338)
    hinv = (float)f->grid_rows / (f->max_y - f->min_y);
    const int ncell = f->grid_cols * f->grid_rows;
    std::vector<int32_t> cell_of(f->n);
    cell_off.assign(ncell + 1, 0);
    for (int i = 0; i < f->n; i++) {
        const int px = (int)std::round((f->x[i] - f->min_x) * winv);
        const int py = (int)std::round((f->y[i] - f->min_y) * hinv);
        if (px < 0 || px >= f->grid_cols || py < 0 || py >= f->grid_rows) { cell_of[i] = -1; continue; }
        cell_of[i] = px * f->grid_rows + py;
        cell_off[cell_of[i] + 1]++;
    }
    for (int c = 0; c < ncell; c++) {
        if (cell_off[c + 1] >= (1 << 14)) return fail(ORBX_ERR_ARG, "grid cell %d holds %d features (limit 16383)", c, cell_off[c + 1]);
        cell_off[c + 1] += cell_off[c];
    }
    cell_feat.assign(std::max(cell_off[ncell], 1), 0);
    std::vector<int32_t> cur(cell_off.begin(), cell_off.end() - 1);
    for (int i = 0; i < f->n; i++) if (cell_of[i] >= 0) cell_feat[cur[cell_of[i]]++] = i;
    return ORBX_OK;
}

// one projection-search problem (a frame and the points projected into it); host pointers
struct ProjJob {
    const OrbmFrame* f;
    int n_pts;
    const uint8_t* valid; const float* u; const float* v; const int32_t* level;
    const float* ur = nullptr;      // M4 / M5 with a rectified-stereo frame (f->u_right != NULL); other searches never gate on it
    int level_window = 0;           // M5: ORBM_LEVELS_*
    bool stereo_gate = false;       // the entry point is one of the two tracking searches (:92-98, :1751-1757)
    const float* view_cos; const float* depth; const uint8_t* bad;     // mode 0
    const float* angle;                                                  // mode 1
    const uint8_t* desc; const uint8_t* has_obs;
    int32_t* assign; uint8_t* occupied;
    int n_matches;      // out
};

// Packs all jobs into one blob, ONE k_proj launch with a wave per job, copies assign / occupied / counts back.
// Dynamic LDS a search kernel may ask for: the CU's 160 KB minus the kernel's own static LDS (block tables, histogram ...) and a
// margin.  (A fixed 150 KB ignored the static part: frames of about 1 700 .. 1 900 features -- staged size between the true budget and
// 150 KB -- made hipFuncSetAttribute fail; found by probing the feature counts around the limit.)
static size_t proj_dynamic_lds_budget()
{
    static size_t budget = 0;
    if (budget == 0) {
        size_t st = 0;
        const void* ks[4] = {(const void*)orbm::k_proj_par<true>, (const void*)orbm::k_proj_par<false>, (const void*)orbm::k_proj<true>, (const void*)orbm::k_proj<false>};
        for (const void* k : ks) {
            hipFuncAttributes fa;
            if (hipFuncGetAttributes(&fa, k) == hipSuccess) st = std::max(st, (size_t)fa.sharedSizeBytes);
        }
        (void)hipGetLastError();
        budget = 160 * 1024 - std::max(st, (size_t)4096) - 1024;
    }
    return budget;
}

static int run_projection_jobs(orbm_matcher* m, ProjJob* jobs, int n_jobs, int last_mode,
                               float th, int far_points, float th_far, float nnratio, int check_ori, float dist_th)
{
    if (!m) return fail(ORBX_ERR_ARG, "NULL matcher");
    if (!jobs || n_jobs < 1) return fail(ORBX_ERR_ARG, "no jobs");
    ORBM_HIP(hipSetDevice(m->device));
    (void)hipGetLastError();            // (an error an earlier, failed call left behind must not fail this one)
    Blob blob(m->h_blob);
    const size_t oargs = blob.reserve(sizeof(orbm::ProjArgs) * (size_t)n_jobs);
    struct Off { size_t x, y, oct, ang, desc, coff, cfeat, sf, uright, ur, valid, u, v, level, vc, dep, bad, angl, dmp, obs, assign, occ, logf, logb, nm; float winv, hinv; };
    std::vector<Off> offs(n_jobs);
    std::vector<int32_t> cell_off, cell_feat;
    size_t max_n = 0;
    bool device_grid = true;        // Frame::AssignFeaturesToGrid on the device (k_grid) when every frame fits its LDS sort
    for (int j = 0; j < n_jobs; j++) device_grid = device_grid && jobs[j].f && jobs[j].f->n <= 8192;
    for (int j = 0; j < n_jobs; j++) {
        const ProjJob& q = jobs[j];
        const OrbmFrame* f = q.f;
        const int n_pts = q.n_pts;
        if (n_pts < 0 || (n_pts > 0 && (!q.valid || !q.u || !q.v || !q.level || !q.desc))) return fail(ORBX_ERR_ARG, "job %d: NULL point arrays", j);
        if (last_mode == 0 && n_pts > 0 && (!q.has_obs || !q.view_cos || !q.depth || !q.bad)) return fail(ORBX_ERR_ARG, "job %d: NULL map point arrays", j);
        if (!q.assign || !q.occupied) return fail(ORBX_ERR_ARG, "job %d: NULL assign/occupied", j);
        Off& o = offs[j];
        int r;
        if (device_grid) {      // validation and the two reciprocals only: the CSR itself is built by k_grid
            if (!f || f->n < 0 || f->grid_cols < 1 || f->grid_rows < 1 || f->grid_cols * (int64_t)f->grid_rows > (1 << 20)) return fail(ORBX_ERR_ARG, "bad frame grid");
            if (f->n > 0 && (!f->x || !f->y || !f->octave || !f->desc)) return fail(ORBX_ERR_ARG, "NULL frame arrays");
            if (!f->scale_factors || f->n_levels < 1) return fail(ORBX_ERR_ARG, "NULL scale factors");
            o.winv = (float)f->grid_cols / (f->max_x - f->min_x);
            o.hinv = (float)f->grid_rows / (f->max_y - f->min_y);
            cell_off.assign((size_t)f->grid_cols * f->grid_rows + 1, 0);
            cell_feat.assign(std::max(f->n, 1), 0);
        } else if ((r = build_grid(f, cell_off, cell_feat, o.winv, o.hinv))) return r;
        if (last_mode == 1 && check_ori && n_pts > 0 && (!q.angle || !f->angle)) return fail(ORBX_ERR_ARG, "job %d: NULL angle arrays", j);
        const bool gate = q.stereo_gate && f->u_right != nullptr && f->n > 0;
        if (gate && n_pts > 0 && !q.ur) return fail(ORBX_ERR_ARG, "job %d: the frame has u_right but the points have no proj_ur", j);
        if (q.level_window < ORBM_LEVELS_AROUND || q.level_window > ORBM_LEVELS_BACKWARD) return fail(ORBX_ERR_ARG, "job %d: bad level_window %d", j, q.level_window);
        for (int i = 0; i < n_pts; i++)
            if (q.valid[i] && (q.level[i] < 0 || q.level[i] >= f->n_levels)) return fail(ORBX_ERR_ARG, "job %d point %d: level %d out of range", j, i, q.level[i]);
        if ((size_t)f->n + 256 > 150 * 1024) return fail(ORBX_ERR_ARG, "frame with %d features exceeds the LDS occupancy table", f->n);
        const int n = f->n;
        max_n = std::max(max_n, (size_t)n);
        o.x = blob.put(f->x, sizeof(float) * n); o.y = blob.put(f->y, sizeof(float) * n);
        o.oct = blob.put(f->octave, sizeof(int32_t) * n);
        o.ang = blob.put(f->angle, f->angle ? sizeof(float) * n : 0);
        o.desc = blob.put(f->desc, (size_t)n * 32);
        o.coff = blob.put(cell_off.data(), sizeof(int32_t) * cell_off.size());
        o.cfeat = blob.put(cell_feat.data(), sizeof(int32_t) * cell_feat.size());
        o.sf = blob.put(f->scale_factors, sizeof(float) * f->n_levels);
        o.uright = blob.put(gate ? f->u_right : nullptr, gate ? sizeof(float) * n : 0);
        o.ur = blob.put(gate ? q.ur : nullptr, gate ? sizeof(float) * n_pts : 0);
        o.valid = blob.put(q.valid, n_pts); o.u = blob.put(q.u, sizeof(float) * n_pts); o.v = blob.put(q.v, sizeof(float) * n_pts);
        o.level = blob.put(q.level, sizeof(int32_t) * n_pts);
        o.vc = blob.put(q.view_cos, q.view_cos ? sizeof(float) * n_pts : 0); o.dep = blob.put(q.depth, q.depth ? sizeof(float) * n_pts : 0);
        o.bad = blob.put(q.bad, q.bad ? n_pts : 0); o.angl = blob.put(q.angle, q.angle ? sizeof(float) * n_pts : 0);
        o.dmp = blob.put(q.desc, (size_t)n_pts * 32); o.obs = blob.put(q.has_obs, q.has_obs ? n_pts : 0);
        o.logf = blob.reserve(sizeof(int32_t) * std::max(n_pts, 1)); o.logb = blob.reserve(sizeof(int32_t) * std::max(n_pts, 1));
    }
    // in/out and output arrays of all jobs sit together at the end: one copy brings every result back
    const size_t out_begin = (m->h_blob.size() + 15) & ~(size_t)15;
    for (int j = 0; j < n_jobs; j++) {
        const ProjJob& q = jobs[j];
        const int n = q.f->n;
        Off& o = offs[j];
        o.assign = blob.put(q.assign, sizeof(int32_t) * n); o.occ = blob.put(q.occupied, n);
        o.nm = blob.reserve(sizeof(int32_t));
    }
    int r = m->ensure(m->h_blob.size());
    if (r) return r;
    uint8_t* base = m->d_blob;
    orbm::ProjArgs* args = (orbm::ProjArgs*)(m->h_blob.data() + oargs);
    for (int j = 0; j < n_jobs; j++) {
        const ProjJob& q = jobs[j];
        const OrbmFrame* f = q.f;
        const Off& o = offs[j];
        orbm::ProjArgs A;
        A.F.x = (const float*)(base + o.x); A.F.y = (const float*)(base + o.y); A.F.octave = (const int32_t*)(base + o.oct);
        A.F.angle = (const float*)(base + o.ang); A.F.desc = base + o.desc;
        A.F.cell_off = (const int32_t*)(base + o.coff); A.F.cell_feat = (const int32_t*)(base + o.cfeat);
        A.F.scale_factors = (const float*)(base + o.sf);
        A.F.min_x = f->min_x; A.F.min_y = f->min_y; A.F.max_x = f->max_x; A.F.max_y = f->max_y; A.F.winv = o.winv; A.F.hinv = o.hinv;
        A.F.n = f->n; A.F.cols = f->grid_cols; A.F.rows = f->grid_rows; A.F.n_levels = f->n_levels;
        const bool gate = q.stereo_gate && f->u_right != nullptr && f->n > 0;
        A.F.u_right = gate ? (const float*)(base + o.uright) : nullptr;
        A.ur = (const float*)(base + o.ur); A.level_window = q.level_window;
        A.n_pts = q.n_pts; A.valid = base + o.valid; A.u = (const float*)(base + o.u); A.v = (const float*)(base + o.v);
        A.level = (const int32_t*)(base + o.level); A.view_cos = (const float*)(base + o.vc); A.depth = (const float*)(base + o.dep);
        A.bad = base + o.bad; A.angle = (const float*)(base + o.angl); A.desc = base + o.dmp; A.has_obs = q.has_obs ? base + o.obs : nullptr;
        A.dist_th = dist_th;
        A.th = th; A.th_far = th_far; A.nnratio = nnratio; A.far_points = far_points; A.check_ori = check_ori; A.last_frame_mode = last_mode;
        A.assign = (int32_t*)(base + o.assign); A.occupied = base + o.occ;
        A.log_feat = (int32_t*)(base + o.logf); A.log_bin = (int32_t*)(base + o.logb); A.n_matches = (int32_t*)(base + o.nm);
        args[j] = A;
    }
    size_t max_cells = 0;
    for (int j = 0; j < n_jobs; j++) max_cells = std::max(max_cells, (size_t)jobs[j].f->grid_cols * jobs[j].f->grid_rows);
    // k_proj_par (64 points at a time) keeps two int tables per feature in LDS; frames too large for them keep the sequential k_proj
    static const bool force_seq = std::getenv("ORBM_PROJ_SEQUENTIAL") != nullptr;      // measurement knob (tools/proj_timing.py)
    const size_t tabs = 8 * ((max_n + 15) & ~(size_t)15);
    const size_t lds_budget = proj_dynamic_lds_budget();
    const bool par = !force_seq && tabs + ((max_n + 63) & ~(size_t)63) + 1024 <= lds_budget;
    const size_t lds_occ = std::max((max_n + 63) & ~(size_t)63, (size_t)64) + (par ? tabs : 0);
    const size_t lds_full = ((max_n + 15) & ~(size_t)15) + max_n * (par ? 40 : 52) + (max_cells + 1) * 4 + 64 + (par ? tabs + max_n * 16 + 32 : 0);     // k_proj: descriptor, x, y, octave, CSR entry, u_right; k_proj_par: descriptor, octave, u_right + the 16-byte records in grid order
    const bool stage = lds_full <= lds_budget;
    const size_t lds = stage ? lds_full : lds_occ;
    for (int j = 0; j < n_jobs; j++) args[j].lds_frame = stage ? 1 : 0;
    ORBM_HIP(hipMemcpyAsync(base, m->h_blob.data(), m->h_blob.size(), hipMemcpyHostToDevice, m->stream));
    if (device_grid) {
        int n_pow2 = 2;
        while ((size_t)n_pow2 < max_n) n_pow2 <<= 1;
        ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_grid, hipFuncAttributeMaxDynamicSharedMemorySize, n_pow2 * 8));
        hipLaunchKernelGGL(orbm::k_grid, dim3(n_jobs), dim3(256), (size_t)n_pow2 * 8, m->stream, (const orbm::ProjArgs*)(base + oargs), n_pow2);
    }
    if (par && stage) {
        ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_proj_par<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(orbm::k_proj_par<true>, dim3(n_jobs), dim3(orbm::kProjThreads), lds, m->stream, (const orbm::ProjArgs*)(base + oargs));
    } else if (par) {
        ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_proj_par<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(orbm::k_proj_par<false>, dim3(n_jobs), dim3(orbm::kProjThreads), lds, m->stream, (const orbm::ProjArgs*)(base + oargs));
    } else if (stage) {
        ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_proj<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(orbm::k_proj<true>, dim3(n_jobs), dim3(64), lds, m->stream, (const orbm::ProjArgs*)(base + oargs));
    } else {
        ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_proj<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(orbm::k_proj<false>, dim3(n_jobs), dim3(64), lds, m->stream, (const orbm::ProjArgs*)(base + oargs));
    }
    ORBM_HIP(hipGetLastError());
    ORBM_HIP(hipMemcpyAsync(m->h_blob.data() + out_begin, base + out_begin, m->h_blob.size() - out_begin, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipStreamSynchronize(m->stream));
    for (int j = 0; j < n_jobs; j++) {
        const Off& o = offs[j];
        const int n = jobs[j].f->n;
        if (n > 0) {
            std::memcpy(jobs[j].assign, m->h_blob.data() + o.assign, sizeof(int32_t) * n);
            std::memcpy(jobs[j].occupied, m->h_blob.data() + o.occ, n);
        }
        std::memcpy(&jobs[j].n_matches, m->h_blob.data() + o.nm, sizeof(int32_t));
    }
    return ORBX_OK;
}

static int run_projection(orbm_matcher* m, const OrbmFrame* f, int last_mode, int n_pts, const uint8_t* valid,
                          const float* u, const float* v, const int32_t* level, const float* view_cos, const float* depth,
                          const uint8_t* bad, const float* angle, const uint8_t* desc, const uint8_t* has_obs,
                          float th, int far_points, float th_far, float nnratio, int check_ori,
                          int32_t* assign, uint8_t* occupied, float dist_th = (float)orbm::TH_HIGH,
                          const float* ur = nullptr, int level_window = 0, bool stereo_gate = false)
{
    ProjJob q;
    q.ur = ur; q.level_window = level_window; q.stereo_gate = stereo_gate;
    q.f = f; q.n_pts = n_pts; q.valid = valid; q.u = u; q.v = v; q.level = level; q.view_cos = view_cos; q.depth = depth; q.bad = bad;
    q.angle = angle; q.desc = desc; q.has_obs = has_obs; q.assign = assign; q.occupied = occupied; q.n_matches = 0;
    const int r = run_projection_jobs(m, &q, 1, last_mode, th, far_points, th_far, nnratio, check_ori, dist_th);
    return r ? r : q.n_matches;
}

extern "C" {

int orbm_hamming(const uint8_t a[32], const uint8_t b[32])
{
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

int orbm_create(int device, orbm_matcher** out)
{
    if (!out) return fail(ORBX_ERR_ARG, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ORBX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(ORBX_ERR_ARG, "device %d out of range", device);
    ORBM_HIP(hipSetDevice(device));
    orbm_matcher* m = new orbm_matcher();
    m->device = device;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) { delete m; return fail(ORBX_ERR_HIP, "stream create failed"); }
    *out = m;
    return ORBX_OK;
}

void orbm_destroy(orbm_matcher* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    if (m->d_blob) (void)hipFree(m->d_blob);
    if (m->d_ws) (void)hipFree(m->d_ws);
    if (m->d_scale) (void)hipFree(m->d_scale);
    delete m;
}

int orbm_search_by_bow(orbm_matcher* m,
                       const uint8_t* desc_kf, int n_kf, const uint8_t* valid_kf, const float* angle_kf, const OrbmFeatVec* fv_kf,
                       const uint8_t* desc_f, int n_f, const float* angle_f, const OrbmFeatVec* fv_f,
                       float nnratio, int check_orientation, int32_t* match_f2kf)
{
    if (!match_f2kf && n_f > 0) return fail(ORBX_ERR_ARG, "match_f2kf is NULL");
    const uint8_t* v2 = nullptr;
    int32_t nm = 0;
    int r = bow_batch<false>(m, 1, &desc_kf, &n_kf, &valid_kf, &angle_kf, &fv_kf, &desc_f, &n_f, &v2, &angle_f, &fv_f,
                             nnratio, check_orientation, &match_f2kf, &nm);
    return r < 0 ? r : nm;
}

int orbm_search_by_bow_batch(orbm_matcher* m, OrbmBowPair* pairs, int n_pairs, float nnratio, int check_orientation)
{
    if (!pairs || n_pairs < 1) return fail(ORBX_ERR_ARG, "no pairs");
    std::vector<const uint8_t*> d1(n_pairs), v1(n_pairs), d2(n_pairs), v2(n_pairs, nullptr);
    std::vector<const float*> a1(n_pairs), a2(n_pairs);
    std::vector<const OrbmFeatVec*> f1(n_pairs), f2(n_pairs);
    std::vector<int> n1(n_pairs), n2(n_pairs);
    std::vector<int32_t*> mo(n_pairs);
    std::vector<int32_t> nm(n_pairs, 0);
    for (int p = 0; p < n_pairs; p++) {
        d1[p] = pairs[p].desc_kf; v1[p] = pairs[p].valid_kf; a1[p] = pairs[p].angle_kf; f1[p] = &pairs[p].fv_kf; n1[p] = pairs[p].n_kf;
        d2[p] = pairs[p].desc_f; a2[p] = pairs[p].angle_f; f2[p] = &pairs[p].fv_f; n2[p] = pairs[p].n_f;
        mo[p] = pairs[p].match_f2kf;
        if (!mo[p] && n2[p] > 0) return fail(ORBX_ERR_ARG, "pair %d: match_f2kf is NULL", p);
    }
    int r = bow_batch<false>(m, n_pairs, d1.data(), n1.data(), v1.data(), a1.data(), f1.data(), d2.data(), n2.data(), v2.data(), a2.data(), f2.data(),
                             nnratio, check_orientation, mo.data(), nm.data());
    if (r < 0) return r;
    for (int p = 0; p < n_pairs; p++) pairs[p].n_matches = nm[p];
    return ORBX_OK;
}

// ---- device-resident batched SearchByBoW: upload once (plan), launch many times, fetch when needed ----
struct orbm_bow_plan {
    orbm_matcher* m = nullptr;
    uint8_t* d_blob = nullptr;
    size_t desc_off = 0;
    int n_pairs = 0;
    std::vector<PairOffsets> po;
};

int orbm_bow_plan_create(orbm_matcher* m, const OrbmBowPair* pairs, int n_pairs, orbm_bow_plan** out)
{
    if (!m || !pairs || n_pairs < 1 || !out) return fail(ORBX_ERR_ARG, "bad plan arguments");
    *out = nullptr;
    ORBM_HIP(hipSetDevice(m->device));
    PinnedBytes host;            // (a plan is created once: its staging buffer lives for this call only)
    Blob blob(host);
    orbm_bow_plan* pl = new orbm_bow_plan();
    pl->m = m; pl->n_pairs = n_pairs; pl->po.resize(n_pairs);
    pl->desc_off = blob.reserve(sizeof(orbm::BowPairDev) * n_pairs);
    static const int32_t zero_off[1] = {0};
    for (int p = 0; p < n_pairs; p++) {
        const OrbmBowPair& q = pairs[p];
        int r;
        if (q.n_kf < 0 || q.n_f < 0 || (q.n_kf > 0 && (!q.desc_kf || !q.valid_kf || !q.angle_kf)) || (q.n_f > 0 && (!q.desc_f || !q.angle_f))) {
            delete pl; return fail(ORBX_ERR_ARG, "pair %d: NULL arrays", p);
        }
        if ((r = check_fv(&q.fv_kf, q.n_kf, "fv_kf")) || (r = check_fv(&q.fv_f, q.n_f, "fv_f"))) { delete pl; return r; }
        PairOffsets& o = pl->po[p];
        o.n1 = q.n_kf; o.n2 = q.n_f; o.nn1 = q.fv_kf.n_nodes; o.nn2 = q.fv_f.n_nodes;
        o.serial = !features_unique(&q.fv_f, q.n_f);
        o.d1 = blob.put(q.desc_kf, (size_t)o.n1 * 32); o.v1 = blob.put(q.valid_kf, o.n1); o.a1 = blob.put(q.angle_kf, sizeof(float) * o.n1);
        o.node1 = blob.put(q.fv_kf.node_id, sizeof(uint32_t) * o.nn1);
        o.off1 = blob.put(o.nn1 ? q.fv_kf.offset : zero_off, sizeof(int32_t) * (o.nn1 + 1));
        o.feat1 = blob.put(q.fv_kf.feat, sizeof(uint32_t) * (o.nn1 ? q.fv_kf.offset[o.nn1] : 0));
        o.d2 = blob.put(q.desc_f, (size_t)o.n2 * 32); o.v2 = blob.reserve(0); o.a2 = blob.put(q.angle_f, sizeof(float) * o.n2);
        o.node2 = blob.put(q.fv_f.node_id, sizeof(uint32_t) * o.nn2);
        o.off2 = blob.put(o.nn2 ? q.fv_f.offset : zero_off, sizeof(int32_t) * (o.nn2 + 1));
        o.feat2 = blob.put(q.fv_f.feat, sizeof(uint32_t) * (o.nn2 ? q.fv_f.offset[o.nn2] : 0));
        o.n_out = o.n2;
        o.match = blob.reserve(sizeof(int32_t) * std::max(o.n_out, 1));
        o.matched2 = blob.reserve(0);
        o.nmatch = blob.reserve(sizeof(int32_t));
    }
    if (hipMalloc((void**)&pl->d_blob, host.size()) != hipSuccess) { delete pl; return fail(ORBX_ERR_HIP, "hipMalloc(%zu) failed", host.size()); }
    uint8_t* base = pl->d_blob;
    orbm::BowPairDev* descs = (orbm::BowPairDev*)(host.data() + pl->desc_off);
    for (int p = 0; p < n_pairs; p++) {
        const PairOffsets& o = pl->po[p];
        orbm::BowPairDev D;
        D.d1 = base + o.d1; D.v1 = base + o.v1; D.a1 = (const float*)(base + o.a1);
        D.node1 = (const uint32_t*)(base + o.node1); D.off1 = (const int32_t*)(base + o.off1); D.feat1 = (const uint32_t*)(base + o.feat1);
        D.d2 = base + o.d2; D.v2 = base + o.v2; D.a2 = (const float*)(base + o.a2);
        D.node2 = (const uint32_t*)(base + o.node2); D.off2 = (const int32_t*)(base + o.off2); D.feat2 = (const uint32_t*)(base + o.feat2);
        D.n1 = o.n1; D.n2 = o.n2; D.nn1 = o.nn1; D.nn2 = o.nn2;
        D.match = (int32_t*)(base + o.match); D.matched2 = base + o.matched2; D.n_matches = (int32_t*)(base + o.nmatch);
        D.serial = o.serial;
        descs[p] = D;
    }
    if (hipMemcpy(base, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(pl->d_blob); delete pl; return fail(ORBX_ERR_HIP, "upload failed"); }
    *out = pl;
    return ORBX_OK;
}

int orbm_bow_plan_create_device(orbm_matcher* m, const OrbmBowPairDevice* pairs, int n_pairs, orbm_bow_plan** out)
{
    if (!m || !pairs || n_pairs < 1 || !out) return fail(ORBX_ERR_ARG, "bad plan arguments");
    *out = nullptr;
    ORBM_HIP(hipSetDevice(m->device));
    std::vector<orbm::BowPairDev> descs(n_pairs);
    for (int p = 0; p < n_pairs; p++) {
        const OrbmBowPairDevice& q = pairs[p];
        const OrbmBowSideDevice* sd[2] = {&q.kf, &q.f};
        for (int k = 0; k < 2; k++)
            if (!sd[k]->desc || !sd[k]->kps || !sd[k]->n || !sd[k]->fv_node || !sd[k]->fv_off || !sd[k]->fv_feat || !sd[k]->n_fv_nodes || sd[k]->cap < 1)
                return fail(ORBX_ERR_ARG, "pair %d: NULL device arrays", p);
        if (!q.match_f2kf || !q.n_matches) return fail(ORBX_ERR_ARG, "pair %d: NULL outputs", p);
        orbm::BowPairDev D;
        D.d1 = q.kf.desc; D.v1 = q.kf.valid; D.a1 = (const float*)q.kf.kps + 3;       // OrbxKeyPoint.angle
        D.node1 = q.kf.fv_node; D.off1 = q.kf.fv_off; D.feat1 = q.kf.fv_feat;
        D.d2 = q.f.desc; D.v2 = nullptr; D.a2 = (const float*)q.f.kps + 3;
        D.node2 = q.f.fv_node; D.off2 = q.f.fv_off; D.feat2 = q.f.fv_feat;
        D.n1 = D.n2 = D.nn1 = D.nn2 = 0;
        D.n1p = q.kf.n; D.n2p = q.f.n; D.nn1p = q.kf.n_fv_nodes; D.nn2p = q.f.n_fv_nodes;
        D.a1_stride = D.a2_stride = (int32_t)(sizeof(OrbxKeyPoint) / sizeof(float));
        D.cap1 = q.kf.cap; D.cap2 = q.f.cap;
        D.match = q.match_f2kf; D.matched2 = nullptr; D.n_matches = q.n_matches;
        D.serial = 0;       // the vocabulary transform puts every feature into exactly one node
        descs[p] = D;
    }
    orbm_bow_plan* pl = new orbm_bow_plan();
    pl->m = m; pl->n_pairs = n_pairs; pl->desc_off = 0;
    if (hipMalloc((void**)&pl->d_blob, sizeof(orbm::BowPairDev) * n_pairs) != hipSuccess) { delete pl; return fail(ORBX_ERR_HIP, "hipMalloc failed"); }
    if (hipMemcpy(pl->d_blob, descs.data(), sizeof(orbm::BowPairDev) * n_pairs, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(pl->d_blob); delete pl; return fail(ORBX_ERR_HIP, "upload failed");
    }
    *out = pl;
    return ORBX_OK;
}

int orbm_bow_plan_run(orbm_bow_plan* pl, float nnratio, int check_orientation, void* stream)
{
    if (!pl) return fail(ORBX_ERR_ARG, "NULL plan");
    ORBM_HIP(hipSetDevice(pl->m->device));
    hipLaunchKernelGGL(orbm::k_bow<false>, dim3(pl->n_pairs), dim3(orbm::kBowThreads), 0, (hipStream_t)stream,
                       (const orbm::BowPairDev*)(pl->d_blob + pl->desc_off), nnratio, check_orientation);
    ORBM_HIP(hipGetLastError());
    return ORBX_OK;
}

int orbm_bow_plan_fetch(orbm_bow_plan* pl, OrbmBowPair* pairs, void* stream)
{
    if (!pl || !pairs) return fail(ORBX_ERR_ARG, "NULL argument");
    if (pl->po.empty()) return fail(ORBX_ERR_ARG, "a device-resident plan writes into the caller's device arrays: nothing to fetch");
    ORBM_HIP(hipSetDevice(pl->m->device));
    hipStream_t st = (hipStream_t)stream;
    for (int p = 0; p < pl->n_pairs; p++) {
        const PairOffsets& o = pl->po[p];
        if (o.n_out > 0 && pairs[p].match_f2kf)
            ORBM_HIP(hipMemcpyAsync(pairs[p].match_f2kf, pl->d_blob + o.match, sizeof(int32_t) * o.n_out, hipMemcpyDeviceToHost, st));
        ORBM_HIP(hipMemcpyAsync(&pairs[p].n_matches, pl->d_blob + o.nmatch, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    ORBM_HIP(hipStreamSynchronize(st));
    return ORBX_OK;
}

void orbm_bow_plan_destroy(orbm_bow_plan* pl)
{
    if (!pl) return;
    (void)hipSetDevice(pl->m->device);
    if (pl->d_blob) (void)hipFree(pl->d_blob);
    delete pl;
}

int orbm_search_by_bow_kfkf(orbm_matcher* m,
                            const uint8_t* desc1, int n1, const uint8_t* valid1, const float* angle1, const OrbmFeatVec* fv1,
                            const uint8_t* desc2, int n2, const uint8_t* valid2, const float* angle2, const OrbmFeatVec* fv2,
                            float nnratio, int check_orientation, int32_t* match12)
{
    if (!match12 && n1 > 0) return fail(ORBX_ERR_ARG, "match12 is NULL");
    int32_t nm = 0;
    int r = bow_batch<true>(m, 1, &desc1, &n1, &valid1, &angle1, &fv1, &desc2, &n2, &valid2, &angle2, &fv2,
                            nnratio, check_orientation, &match12, &nm);
    return r < 0 ? r : nm;
}

int orbm_search_by_projection(orbm_matcher* m, const OrbmFrame* f,
                              int n_mp, const uint8_t* in_view, const float* proj_u, const float* proj_v, const float* proj_ur,
                              const int32_t* pred_level, const float* view_cos, const float* track_depth,
                              const uint8_t* desc_mp, const uint8_t* mp_has_obs, const uint8_t* mp_bad,
                              float th, int far_points, float th_far, float nnratio,
                              int32_t* assign, uint8_t* occupied)
{
    return run_projection(m, f, 0, n_mp, in_view, proj_u, proj_v, pred_level, view_cos, track_depth, mp_bad, nullptr,
                          desc_mp, mp_has_obs, th, far_points, th_far, nnratio, 0, assign, occupied, (float)orbm::TH_HIGH,
                          proj_ur, 0, true);
}

int orbm_search_by_projection_last(orbm_matcher* m, const OrbmFrame* cur,
                                   int n_last, const uint8_t* last_valid, const float* proj_u, const float* proj_v, const float* proj_ur,
                                   const int32_t* last_octave, const float* last_angle,
                                   const uint8_t* desc_mp, const uint8_t* mp_has_obs,
                                   float th, int level_window, int check_orientation,
                                   int32_t* assign, uint8_t* occupied)
{
    if (n_last > 0 && !mp_has_obs) return fail(ORBX_ERR_ARG, "NULL mp_has_obs");
    return run_projection(m, cur, 1, n_last, last_valid, proj_u, proj_v, last_octave, nullptr, nullptr, nullptr, last_angle,
                          desc_mp, mp_has_obs, th, 0, 0.f, 0.f, check_orientation, assign, occupied, (float)orbm::TH_HIGH,
                          proj_ur, level_window, true);
}

// Batched forms: n_frames independent (frame, points) problems in ONE launch, a wave per frame.
static int run_projection_batch(orbm_matcher* m, OrbmProjQuery* q, int n_frames, int mode, float th, int far_points, float th_far,
                                float nnratio, int check_ori, float dist_th)
{
    if (!q || n_frames < 1) return fail(ORBX_ERR_ARG, "no queries");
    std::vector<ProjJob> jobs(n_frames);
    for (int j = 0; j < n_frames; j++) {
        ProjJob& p = jobs[j];
        p.f = q[j].frame; p.n_pts = q[j].n_pts; p.valid = q[j].valid; p.u = q[j].proj_u; p.v = q[j].proj_v; p.level = q[j].level;
        p.ur = q[j].proj_ur; p.level_window = mode == 1 ? q[j].level_window : 0; p.stereo_gate = true;
        p.view_cos = q[j].view_cos; p.depth = q[j].track_depth; p.bad = q[j].mp_bad; p.angle = q[j].angle; p.desc = q[j].desc_mp;
        p.has_obs = q[j].mp_has_obs; p.assign = q[j].assign; p.occupied = q[j].occupied; p.n_matches = 0;
        if (!p.f) return fail(ORBX_ERR_ARG, "query %d: NULL frame", j);
        if (mode == 1 && p.n_pts > 0 && !p.has_obs) return fail(ORBX_ERR_ARG, "query %d: NULL mp_has_obs", j);
    }
    const int r = run_projection_jobs(m, jobs.data(), n_frames, mode, th, far_points, th_far, nnratio, check_ori, dist_th);
    if (r) return r;
    for (int j = 0; j < n_frames; j++) q[j].n_matches = jobs[j].n_matches;
    return ORBX_OK;
}

int orbm_search_by_projection_last_batch(orbm_matcher* m, OrbmProjQuery* queries, int n_frames, float th, int check_orientation)
{
    return run_projection_batch(m, queries, n_frames, 1, th, 0, 0.f, 0.f, check_orientation, (float)orbm::TH_HIGH);
}

int orbm_search_by_projection_batch(orbm_matcher* m, OrbmProjQuery* queries, int n_frames, float th, int far_points, float th_far, float nnratio)
{
    return run_projection_batch(m, queries, n_frames, 0, th, far_points, th_far, nnratio, 0, (float)orbm::TH_HIGH);
}

// shared body of the two device-resident tracking searches (mode 1: last frame, mode 0: Frame x map points)
static int projection_batch_device(orbm_matcher* m, const OrbmDeviceFrames* cur, const OrbmDeviceLastPoints* last, const OrbmDeviceMapPointExtras* mp, int mode,
                                   int batch, float th, int check_orientation, int32_t* d_assign, uint8_t* d_occupied, int32_t* d_n_matches, void* stream)
{
    if (!m || !cur || !last || !d_assign || !d_occupied || !d_n_matches) return fail(ORBX_ERR_ARG, "NULL argument");
    if (batch < 1 || cur->cap < 1 || last->cap < 1) return fail(ORBX_ERR_ARG, "bad batch / capacities");
    if (!cur->d_kps || !cur->d_desc || !cur->d_n || !cur->scale_factors || cur->n_levels < 1 || cur->n_levels > 32) return fail(ORBX_ERR_ARG, "bad current-frame description");
    if (!last->d_valid || !last->d_u || !last->d_v || !last->d_octave || !last->d_desc || !last->d_n || (mode == 1 && check_orientation && !last->d_angle)) return fail(ORBX_ERR_ARG, "bad point description");
    if (mode == 0 && (!mp || !mp->d_view_cos || !mp->d_track_depth || !mp->d_bad || !last->d_has_obs)) return fail(ORBX_ERR_ARG, "bad map-point description");
    if (cur->grid_cols < 1 || cur->grid_rows < 1 || cur->grid_cols * (int64_t)cur->grid_rows > (1 << 20) || !(cur->max_x > cur->min_x) || !(cur->max_y > cur->min_y)) return fail(ORBX_ERR_ARG, "bad frame grid");
    if (cur->cap > 8192) return fail(ORBX_ERR_CAPACITY, "at most 8192 features per frame (the grid is sorted in LDS)");
    ORBM_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t B = (size_t)batch, cap = (size_t)cur->cap, pcap = (size_t)last->cap, cells = (size_t)cur->grid_cols * cur->grid_rows;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t off = 0;
    const size_t o_x = off; off += al(B * cap * 4);
    const size_t o_y = off; off += al(B * cap * 4);
    const size_t o_oct = off; off += al(B * cap * 4);
    const size_t o_ang = off; off += al(B * cap * 4);
    const size_t o_coff = off; off += al(B * (cells + 1) * 4);
    const size_t o_cfeat = off; off += al(B * cap * 4);
    const size_t o_lf = off; off += al(B * pcap * 4);
    const size_t o_lb = off; off += al(B * pcap * 4);
    const size_t o_jobs = off; off += al(B * sizeof(orbm::ProjArgs));
    if (off > m->ws_cap) {
        if (m->d_ws) { ORBM_HIP(hipDeviceSynchronize()); (void)hipFree(m->d_ws); }
        m->d_ws = nullptr; m->ws_cap = 0;
        ORBM_HIP(hipMalloc((void**)&m->d_ws, off + off / 4));
        m->ws_cap = off + off / 4;
    }
    if (!m->d_scale) ORBM_HIP(hipMalloc((void**)&m->d_scale, 32 * sizeof(float)));
    if (std::memcmp(m->h_scale, cur->scale_factors, sizeof(float) * cur->n_levels) != 0) {      // (unchanged between the calls of a stream of frames)
        ORBM_HIP(hipStreamSynchronize(st));                 // an earlier copy out of h_scale may still be in flight
        std::memcpy(m->h_scale, cur->scale_factors, sizeof(float) * cur->n_levels);
        ORBM_HIP(hipMemcpyAsync(m->d_scale, m->h_scale, sizeof(float) * cur->n_levels, hipMemcpyHostToDevice, st));
    }
    // LDS of the search kernel, as in run_projection_jobs
    const size_t max_n = cap;
    const size_t tabs = 8 * ((max_n + 15) & ~(size_t)15);
    const size_t lds_budget = proj_dynamic_lds_budget();
    if (tabs + ((max_n + 63) & ~(size_t)63) + 1024 > lds_budget) return fail(ORBX_ERR_CAPACITY, "frames of %zu features exceed the search kernel's LDS tables", max_n);
    const size_t lds_occ = std::max((max_n + 63) & ~(size_t)63, (size_t)64) + tabs;
    const size_t lds_full = ((max_n + 15) & ~(size_t)15) + max_n * 40 + (cells + 1) * 4 + 64 + tabs + max_n * 16 + 32;
    const bool stage = lds_full <= lds_budget;
    const size_t lds = stage ? lds_full : lds_occ;
    orbm::ProjDevSetup P;
    P.kps = cur->d_kps; P.desc = cur->d_desc; P.n = cur->d_n; P.cap = cur->cap;
    P.p_valid = last->d_valid; P.p_u = last->d_u; P.p_v = last->d_v; P.p_oct = last->d_octave; P.p_angle = last->d_angle; P.p_desc = last->d_desc; P.p_n = last->d_n; P.p_cap = last->cap; P.p_obs = last->d_has_obs;
    uint8_t* w = m->d_ws;
    P.x = (float*)(w + o_x); P.y = (float*)(w + o_y); P.oct = (int32_t*)(w + o_oct); P.ang = (float*)(w + o_ang);
    P.cell_off = (int32_t*)(w + o_coff); P.cell_feat = (int32_t*)(w + o_cfeat); P.log_feat = (int32_t*)(w + o_lf); P.log_bin = (int32_t*)(w + o_lb);
    P.scale = m->d_scale; P.n_levels = cur->n_levels; P.cols = cur->grid_cols; P.rows = cur->grid_rows;
    P.min_x = cur->min_x; P.min_y = cur->min_y; P.max_x = cur->max_x; P.max_y = cur->max_y; P.th = th; P.dist_th = (float)orbm::TH_HIGH;
    P.check_ori = mode == 1 ? check_orientation : 0; P.lds_frame = stage ? 1 : 0;
    P.mode = mode;
    P.p_vcos = mp ? mp->d_view_cos : nullptr; P.p_depth = mp ? mp->d_track_depth : nullptr; P.p_bad = mp ? mp->d_bad : nullptr;
    P.th_far = mp ? mp->th_far : 0.f; P.nnratio = mp ? mp->nnratio : 0.f; P.far_points = mp ? mp->far_points : 0;
    P.assign = d_assign; P.occupied = d_occupied; P.n_matches = d_n_matches;
    P.jobs = (orbm::ProjArgs*)(w + o_jobs);
    hipLaunchKernelGGL(orbm::k_proj_dev_setup, dim3(batch), dim3(256), 0, st, P);
    int n_pow2 = 2;
    while ((size_t)n_pow2 < max_n) n_pow2 <<= 1;
    ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_grid, hipFuncAttributeMaxDynamicSharedMemorySize, n_pow2 * 8));
    hipLaunchKernelGGL(orbm::k_grid, dim3(batch), dim3(256), (size_t)n_pow2 * 8, st, (const orbm::ProjArgs*)P.jobs, n_pow2);
    if (stage) {
        ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_proj_par<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(orbm::k_proj_par<true>, dim3(batch), dim3(orbm::kProjThreads), lds, st, (const orbm::ProjArgs*)P.jobs);
    } else {
        ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_proj_par<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(orbm::k_proj_par<false>, dim3(batch), dim3(orbm::kProjThreads), lds, st, (const orbm::ProjArgs*)P.jobs);
    }
    ORBM_HIP(hipGetLastError());
    return ORBX_OK;
}

int orbm_search_by_projection_last_batch_device(orbm_matcher* m, const OrbmDeviceFrames* cur, const OrbmDeviceLastPoints* last, int batch,
                                                float th, int check_orientation, int32_t* d_assign, uint8_t* d_occupied, int32_t* d_n_matches, void* stream)
{
    return projection_batch_device(m, cur, last, nullptr, 1, batch, th, check_orientation, d_assign, d_occupied, d_n_matches, stream);
}

int orbm_search_by_projection_batch_device(orbm_matcher* m, const OrbmDeviceFrames* cur, const OrbmDeviceLastPoints* points, const OrbmDeviceMapPointExtras* extras,
                                           int batch, float th, int32_t* d_assign, uint8_t* d_occupied, int32_t* d_n_matches, void* stream)
{
    return projection_batch_device(m, cur, points, extras, 0, batch, th, 0, d_assign, d_occupied, d_n_matches, stream);
}

#ifdef ORBM_PROJ_TIMING
int orbm_debug_proj_prof(unsigned long long* out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(orbm::d_proj_prof), sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(orbm::d_proj_prof), z, sizeof(z)) != hipSuccess) return ORBX_ERR_HIP;
    return ORBX_OK;
}
#endif

int orbm_search_by_projection_kf(orbm_matcher* m, const OrbmFrame* cur,
                                 int n_pts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                 const int32_t* pred_level, const float* kf_angle, const uint8_t* desc_mp,
                                 float th, int orb_dist, int check_orientation, int32_t* assign, uint8_t* occupied)
{
    return run_projection(m, cur, 1, n_pts, valid, proj_u, proj_v, pred_level, nullptr, nullptr, nullptr, kf_angle,
                          desc_mp, nullptr, th, 0, 0.f, 0.f, check_orientation, assign, occupied, (float)orb_dist);
}

int orbm_search_by_projection_sim3(orbm_matcher* m, const OrbmFrame* kf,
                                   int n_pts, const uint8_t* valid, const float* proj_u, const float* proj_v,
                                   const int32_t* pred_level, const uint8_t* desc_mp, int th, float ratio_hamming,
                                   int32_t* assign, uint8_t* occupied)
{
    return run_projection(m, kf, 2, n_pts, valid, proj_u, proj_v, pred_level, nullptr, nullptr, nullptr, nullptr,
                          desc_mp, nullptr, (float)th, 0, 0.f, 0.f, 0, assign, occupied, (float)orbm::TH_LOW * ratio_hamming);
}

int orbm_search_for_triangulation(orbm_matcher* m, const OrbmTriSide* k1, const OrbmTriSide* k2, float ep_x, float ep_y, const float* F12,
                                  const float* level_sigma2_2, const float* scale_factors_2, int n_levels_2,
                                  int only_stereo, int coarse, int check_orientation, int32_t* match12)
{
    if (!m || !k1 || !k2 || !F12 || !level_sigma2_2 || !scale_factors_2 || n_levels_2 < 1) return fail(ORBX_ERR_ARG, "NULL argument");
    const OrbmTriSide* side[2] = {k1, k2};
    for (int q = 0; q < 2; q++) {
        const OrbmTriSide* k = side[q];
        if (k->n < 0 || (k->n > 0 && (!k->desc || !k->has_mp || !k->stereo || !k->x || !k->y || !k->octave))) return fail(ORBX_ERR_ARG, "NULL key-frame arrays");
        if (check_orientation && k->n > 0 && !k->angle) return fail(ORBX_ERR_ARG, "NULL angles");
        const int r = check_fv(&k->fv, k->n, q ? "fv2" : "fv1");
        if (r) return r;
        if (!features_unique(&k->fv, k->n)) return fail(ORBX_ERR_ARG, "a feature appears in two vocabulary nodes");
    }
    for (int i = 0; i < k2->n; i++)
        if (k2->octave[i] < 0 || k2->octave[i] >= n_levels_2) return fail(ORBX_ERR_ARG, "octave out of range");
    if (k1->n > 0 && !match12) return fail(ORBX_ERR_ARG, "NULL match12");
    if (k1->n == 0) return 0;
    ORBM_HIP(hipSetDevice(m->device));
    Blob blob(m->h_blob);
    size_t o[2][10];
    for (int q = 0; q < 2; q++) {
        const OrbmTriSide* k = side[q];
        const int n = k->n, nn = k->fv.n_nodes, tot = nn > 0 ? k->fv.offset[nn] : 0;
        o[q][0] = blob.put(k->desc, (size_t)n * 32); o[q][1] = blob.put(k->has_mp, n); o[q][2] = blob.put(k->stereo, n);
        o[q][3] = blob.put(k->x, sizeof(float) * n); o[q][4] = blob.put(k->y, sizeof(float) * n);
        o[q][5] = blob.put(k->octave, sizeof(int32_t) * n); o[q][6] = blob.put(k->angle, k->angle ? sizeof(float) * n : 0);
        o[q][7] = blob.put(k->fv.node_id, sizeof(uint32_t) * nn); o[q][8] = blob.put(k->fv.offset, sizeof(int32_t) * (nn + 1));
        o[q][9] = blob.put(k->fv.feat, sizeof(uint32_t) * tot);
    }
    const size_t osig = blob.put(level_sigma2_2, sizeof(float) * n_levels_2), osc = blob.put(scale_factors_2, sizeof(float) * n_levels_2);
    const size_t in_bytes = m->h_blob.size();
    const size_t omatch = blob.reserve(sizeof(int32_t) * k1->n), onm = blob.reserve(sizeof(int32_t));
    int r = m->ensure(m->h_blob.size());
    if (r) return r;
    uint8_t* b = m->d_blob;
    orbm::TriArgs A;
    A.d1 = b + o[0][0]; A.mp1 = b + o[0][1]; A.st1 = b + o[0][2]; A.x1 = (const float*)(b + o[0][3]); A.y1 = (const float*)(b + o[0][4]);
    A.a1 = (const float*)(b + o[0][6]); A.node1 = (const uint32_t*)(b + o[0][7]); A.off1 = (const int32_t*)(b + o[0][8]); A.feat1 = (const uint32_t*)(b + o[0][9]);
    A.d2 = b + o[1][0]; A.mp2 = b + o[1][1]; A.st2 = b + o[1][2]; A.x2 = (const float*)(b + o[1][3]); A.y2 = (const float*)(b + o[1][4]);
    A.oct2 = (const int32_t*)(b + o[1][5]); A.a2 = (const float*)(b + o[1][6]);
    A.node2 = (const uint32_t*)(b + o[1][7]); A.off2 = (const int32_t*)(b + o[1][8]); A.feat2 = (const uint32_t*)(b + o[1][9]);
    A.sigma2_2 = (const float*)(b + osig); A.scale2 = (const float*)(b + osc);
    for (int i = 0; i < 9; i++) A.F12[i] = F12[i];
    A.ep_x = ep_x; A.ep_y = ep_y;
    A.n1 = k1->n; A.nn1 = k1->fv.n_nodes; A.nn2 = k2->fv.n_nodes; A.only_stereo = only_stereo; A.coarse = coarse; A.check_ori = check_orientation;
    A.match12 = (int32_t*)(b + omatch); A.n_matches = (int32_t*)(b + onm);
    ORBM_HIP(hipMemcpyAsync(b, m->h_blob.data(), in_bytes, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(orbm::k_triangulation, dim3(1), dim3(256), 0, m->stream, A);
    ORBM_HIP(hipGetLastError());
    int nm = 0;
    ORBM_HIP(hipMemcpyAsync(match12, b + omatch, sizeof(int32_t) * k1->n, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipMemcpyAsync(&nm, b + onm, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipStreamSynchronize(m->stream));
    return nm;
}

int orbm_search_for_initialization(orbm_matcher* m, const uint8_t* desc1, int n1, const int32_t* octave1, const float* angle1,
                                   const float* prev_x, const float* prev_y, const OrbmFrame* f2,
                                   int window_size, float nnratio, int check_orientation, int32_t* match12)
{
    if (!m || !f2) return fail(ORBX_ERR_ARG, "NULL argument");
    if (n1 < 0 || (n1 > 0 && (!desc1 || !octave1 || !prev_x || !prev_y || !match12))) return fail(ORBX_ERR_ARG, "NULL F1 arrays");
    if (check_orientation && ((n1 > 0 && !angle1) || (f2->n > 0 && !f2->angle))) return fail(ORBX_ERR_ARG, "NULL angles");
    std::vector<int32_t> cell_off, cell_feat;
    float winv, hinv;
    int r = build_grid(f2, cell_off, cell_feat, winv, hinv);
    if (r) return r;
    if (n1 == 0) return 0;
    const int n = f2->n;
    if ((size_t)n * 8 + 256 > 150 * 1024) return fail(ORBX_ERR_ARG, "F2 with %d features exceeds the LDS state", n);
    ORBM_HIP(hipSetDevice(m->device));
    Blob blob(m->h_blob);
    const size_t ox = blob.put(f2->x, sizeof(float) * n), oy = blob.put(f2->y, sizeof(float) * n);
    const size_t ooct = blob.put(f2->octave, sizeof(int32_t) * n), oang = blob.put(f2->angle, f2->angle ? sizeof(float) * n : 0);
    const size_t odesc = blob.put(f2->desc, (size_t)n * 32);
    const size_t ocoff = blob.put(cell_off.data(), sizeof(int32_t) * cell_off.size()), ocfeat = blob.put(cell_feat.data(), sizeof(int32_t) * cell_feat.size());
    const size_t od1 = blob.put(desc1, (size_t)n1 * 32), oo1 = blob.put(octave1, sizeof(int32_t) * n1), oa1 = blob.put(angle1, angle1 ? sizeof(float) * n1 : 0);
    const size_t opx = blob.put(prev_x, sizeof(float) * n1), opy = blob.put(prev_y, sizeof(float) * n1);
    const size_t in_bytes = m->h_blob.size();
    const size_t omatch = blob.reserve(sizeof(int32_t) * n1), onm = blob.reserve(sizeof(int32_t));
    if ((r = m->ensure(m->h_blob.size()))) return r;
    uint8_t* b = m->d_blob;
    orbm::InitArgs A;
    A.F2.x = (const float*)(b + ox); A.F2.y = (const float*)(b + oy); A.F2.octave = (const int32_t*)(b + ooct); A.F2.angle = (const float*)(b + oang);
    A.F2.desc = b + odesc; A.F2.cell_off = (const int32_t*)(b + ocoff); A.F2.cell_feat = (const int32_t*)(b + ocfeat); A.F2.scale_factors = nullptr;
    A.F2.min_x = f2->min_x; A.F2.min_y = f2->min_y; A.F2.max_x = f2->max_x; A.F2.max_y = f2->max_y; A.F2.winv = winv; A.F2.hinv = hinv;
    A.F2.n = n; A.F2.cols = f2->grid_cols; A.F2.rows = f2->grid_rows;
    A.d1 = b + od1; A.oct1 = (const int32_t*)(b + oo1); A.a1 = (const float*)(b + oa1); A.prev_x = (const float*)(b + opx); A.prev_y = (const float*)(b + opy);
    A.n1 = n1; A.window = window_size; A.check_ori = check_orientation; A.nnratio = nnratio;
    A.match12 = (int32_t*)(b + omatch); A.n_matches = (int32_t*)(b + onm);
    const size_t lds = std::max((size_t)n * 8, (size_t)64);
    ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_initialization, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ORBM_HIP(hipMemcpyAsync(b, m->h_blob.data(), in_bytes, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(orbm::k_initialization, dim3(1), dim3(64), lds, m->stream, A);
    ORBM_HIP(hipGetLastError());
    int nm = 0;
    ORBM_HIP(hipMemcpyAsync(match12, b + omatch, sizeof(int32_t) * n1, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipMemcpyAsync(&nm, b + onm, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipStreamSynchronize(m->stream));
    return nm;
}

int orbm_fuse_search(orbm_matcher* m, const OrbmFrame* kf, const float* u_right, const float* inv_level_sigma2,
                     int n_pts, const uint8_t* valid, const float* proj_u, const float* proj_v, const float* proj_ur,
                     const int32_t* pred_level, const uint8_t* desc_mp, float th, int chi2_check,
                     int32_t* best_idx, int32_t* best_dist)
{
    if (!m) return fail(ORBX_ERR_ARG, "NULL matcher");
    if (n_pts < 0 || (n_pts > 0 && (!valid || !proj_u || !proj_v || !pred_level || !desc_mp || !best_idx || !best_dist)))
        return fail(ORBX_ERR_ARG, "NULL point arrays");
    if (chi2_check && (!u_right || !inv_level_sigma2 || (n_pts > 0 && !proj_ur))) return fail(ORBX_ERR_ARG, "NULL stereo / sigma arrays");
    std::vector<int32_t> cell_off, cell_feat;
    float winv, hinv;
    int r = build_grid(kf, cell_off, cell_feat, winv, hinv);
    if (r) return r;
    for (int i = 0; i < n_pts; i++)
        if (valid[i] && (pred_level[i] < 0 || pred_level[i] >= kf->n_levels)) return fail(ORBX_ERR_ARG, "point %d: level %d out of range", i, pred_level[i]);
    if (chi2_check)
        for (int i = 0; i < kf->n; i++)
            if (kf->octave[i] < 0 || kf->octave[i] >= kf->n_levels) return fail(ORBX_ERR_ARG, "key point %d: octave %d out of range", i, kf->octave[i]);
    if (n_pts == 0) return ORBX_OK;
    ORBM_HIP(hipSetDevice(m->device));
    Blob blob(m->h_blob);
    const int n = kf->n;
    const size_t ox = blob.put(kf->x, sizeof(float) * n), oy = blob.put(kf->y, sizeof(float) * n);
    const size_t ooct = blob.put(kf->octave, sizeof(int32_t) * n), odesc = blob.put(kf->desc, (size_t)n * 32);
    const size_t ocoff = blob.put(cell_off.data(), sizeof(int32_t) * cell_off.size());
    const size_t ocfeat = blob.put(cell_feat.data(), sizeof(int32_t) * cell_feat.size());
    const size_t osf = blob.put(kf->scale_factors, sizeof(float) * kf->n_levels);
    const size_t our = blob.put(u_right, chi2_check ? sizeof(float) * n : 0);
    const size_t osig = blob.put(inv_level_sigma2, chi2_check ? sizeof(float) * kf->n_levels : 0);
    const size_t ovalid = blob.put(valid, n_pts), ou = blob.put(proj_u, sizeof(float) * n_pts), ov = blob.put(proj_v, sizeof(float) * n_pts);
    const size_t opr = blob.put(proj_ur, (chi2_check && proj_ur) ? sizeof(float) * n_pts : 0);
    const size_t olevel = blob.put(pred_level, sizeof(int32_t) * n_pts), odmp = blob.put(desc_mp, (size_t)n_pts * 32);
    const size_t obi = blob.reserve(sizeof(int32_t) * n_pts), obd = blob.reserve(sizeof(int32_t) * n_pts);
    if ((r = m->ensure(m->h_blob.size()))) return r;
    uint8_t* base = m->d_blob;
    orbm::FuseArgs A;
    A.F.x = (const float*)(base + ox); A.F.y = (const float*)(base + oy); A.F.octave = (const int32_t*)(base + ooct);
    A.F.angle = nullptr; A.F.desc = base + odesc;
    A.F.cell_off = (const int32_t*)(base + ocoff); A.F.cell_feat = (const int32_t*)(base + ocfeat);
    A.F.scale_factors = (const float*)(base + osf);
    A.F.min_x = kf->min_x; A.F.min_y = kf->min_y; A.F.max_x = kf->max_x; A.F.max_y = kf->max_y; A.F.winv = winv; A.F.hinv = hinv;
    A.F.n = n; A.F.cols = kf->grid_cols; A.F.rows = kf->grid_rows;
    A.u_right = (const float*)(base + our); A.inv_sigma2 = (const float*)(base + osig);
    A.n_pts = n_pts; A.valid = base + ovalid; A.u = (const float*)(base + ou); A.v = (const float*)(base + ov); A.ur = (const float*)(base + opr);
    A.level = (const int32_t*)(base + olevel); A.desc = base + odmp; A.th = th; A.chi2_check = chi2_check;
    A.best_idx = (int32_t*)(base + obi); A.best_dist = (int32_t*)(base + obd);
    ORBM_HIP(hipMemcpyAsync(base, m->h_blob.data(), obi, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(orbm::k_fuse, dim3((n_pts + 3) / 4), dim3(256), 0, m->stream, A);
    ORBM_HIP(hipGetLastError());
    ORBM_HIP(hipMemcpyAsync(best_idx, base + obi, sizeof(int32_t) * n_pts, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipMemcpyAsync(best_dist, base + obd, sizeof(int32_t) * n_pts, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

int orbm_distinctive_descriptors(orbm_matcher* m, const uint8_t* desc, const int32_t* off, int n_points, int32_t* best_idx, int32_t* best_median)
{
    if (!m || n_points < 0 || !off || (n_points > 0 && !best_idx)) return fail(ORBX_ERR_ARG, "NULL argument");
    if (n_points == 0) return ORBX_OK;
    if (off[0] != 0) return fail(ORBX_ERR_ARG, "off[0] must be 0");
    int max_n = 0;
    for (int p = 0; p < n_points; p++) {
        const int n = off[p + 1] - off[p];
        if (n < 0) return fail(ORBX_ERR_ARG, "offsets of point %d decrease", p);
        max_n = std::max(max_n, n);
    }
    const int total = off[n_points];
    if (total > 0 && !desc) return fail(ORBX_ERR_ARG, "NULL descriptors");
    if (max_n > 4096) return fail(ORBX_ERR_CAPACITY, "a map point with %d observations exceeds the LDS staging (4096)", max_n);
    ORBM_HIP(hipSetDevice(m->device));
    Blob blob(m->h_blob);
    const size_t od = blob.put(desc, (size_t)total * 32), oo = blob.put(off, sizeof(int32_t) * (n_points + 1));
    const size_t in_bytes = m->h_blob.size();
    const size_t ob = blob.reserve(sizeof(int32_t) * n_points), om = blob.reserve(sizeof(int32_t) * n_points);
    int r = m->ensure(m->h_blob.size());
    if (r) return r;
    uint8_t* b = m->d_blob;
    const size_t lds = std::max((size_t)max_n * 32, (size_t)64);
    ORBM_HIP(hipFuncSetAttribute((const void*)orbm::k_distinctive, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ORBM_HIP(hipMemcpyAsync(b, m->h_blob.data(), in_bytes, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(orbm::k_distinctive, dim3(n_points), dim3(64), lds, m->stream, b + od, (const int32_t*)(b + oo), n_points,
                       (int32_t*)(b + ob), (int32_t*)(b + om));
    ORBM_HIP(hipGetLastError());
    ORBM_HIP(hipMemcpyAsync(best_idx, b + ob, sizeof(int32_t) * n_points, hipMemcpyDeviceToHost, m->stream));
    if (best_median) ORBM_HIP(hipMemcpyAsync(best_median, b + om, sizeof(int32_t) * n_points, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

int orbm_update_normal_and_depth(orbm_matcher* m, const float* pos, const float* centers, const int32_t* off, const float* ref_center,
                                 const float* level_scale, float last_level_scale, int n_points,
                                 float* normal, float* max_dist, float* min_dist)
{
    if (!m || n_points < 0 || !off || (n_points > 0 && (!pos || !ref_center || !level_scale || !normal || !max_dist || !min_dist)))
        return fail(ORBX_ERR_ARG, "NULL argument");
    if (n_points == 0) return ORBX_OK;
    if (off[0] != 0) return fail(ORBX_ERR_ARG, "off[0] must be 0");
    for (int p = 0; p < n_points; p++)
        if (off[p + 1] <= off[p]) return fail(ORBX_ERR_ARG, "point %d has no observation (the reference returns before touching it)", p);
    const int total = off[n_points];
    if (!centers) return fail(ORBX_ERR_ARG, "NULL camera centres");
    ORBM_HIP(hipSetDevice(m->device));
    Blob blob(m->h_blob);
    const size_t op = blob.put(pos, sizeof(float) * 3 * n_points), oc = blob.put(centers, sizeof(float) * 3 * total);
    const size_t oo = blob.put(off, sizeof(int32_t) * (n_points + 1)), orc = blob.put(ref_center, sizeof(float) * 3 * n_points);
    const size_t ol = blob.put(level_scale, sizeof(float) * n_points);
    const size_t in_bytes = m->h_blob.size();
    const size_t on = blob.reserve(sizeof(float) * 3 * n_points), omx = blob.reserve(sizeof(float) * n_points), omn = blob.reserve(sizeof(float) * n_points);
    int r = m->ensure(m->h_blob.size());
    if (r) return r;
    uint8_t* b = m->d_blob;
    ORBM_HIP(hipMemcpyAsync(b, m->h_blob.data(), in_bytes, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(orbm::k_normal_depth, dim3((n_points + 255) / 256), dim3(256), 0, m->stream, (const float*)(b + op), (const float*)(b + oc),
                       (const int32_t*)(b + oo), (const float*)(b + orc), (const float*)(b + ol), last_level_scale, n_points,
                       (float*)(b + on), (float*)(b + omx), (float*)(b + omn));
    ORBM_HIP(hipGetLastError());
    ORBM_HIP(hipMemcpyAsync(normal, b + on, sizeof(float) * 3 * n_points, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipMemcpyAsync(max_dist, b + omx, sizeof(float) * n_points, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipMemcpyAsync(min_dist, b + omn, sizeof(float) * n_points, hipMemcpyDeviceToHost, m->stream));
    ORBM_HIP(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

}  // extern "C"
