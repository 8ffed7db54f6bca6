// dbow_vocab.hip -- gfx950 kernels + C ABI for DBoW2's descriptor -> word transform: what
// Frame::ComputeBoW (reference src/Frame.cc:825-832) runs right before ORBmatcher::SearchByBoW,
//   mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)
// i.e. TemplatedVocabulary::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1193 and :1216-1259) with FORB
// descriptors (256-bit Hamming), TF_IDF weights and L1 normalisation (BowVector.cpp:33-84), FeatureVector.cpp:31-47.
//
// MI355X mapping.  k_vocab_descend: 16 lanes per feature, one lane per child of the current node (k = 10 in ORBvoc), so a
// tree level is ONE gather round trip per feature instead of k dependent ones; first-minimum tie-break through a
// (distance, child position) key.  The 1.1 M-node ORB vocabulary is 35 MB of centroids: its upper levels live in L2, the
// leaves stream from HBM / Infinity Cache.  k_vocab_assemble: one workgroup per frame turns the per-feature (word, weight,
// node) triples into the two std::map-ordered containers -- a bitonic sort of (id, feature) keys in LDS, ordered segment
// sums (the double additions happen in the reference's order, so BowVector values are bit-identical), CSR compaction.
// Device-resident in and out: it consumes the extractor's descriptor output in place.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/orbslam3_hip.h"

namespace orbx {
int fail(int code, const char* fmt, ...);
}
using orbx::fail;

#define ORBV_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(ORBX_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace orbv {

struct VocabDev {
    int32_t n_nodes, L;
    const int32_t* child_off;
    const uint32_t* child_id;
    const uint4* desc;          // 2 x uint4 per node
    const double* weight;
    const int32_t* word_id;
    // per CHILD SLOT (position in child_id): the child's centroid and its own child range, so that a tree level is ONE round trip
    // (slot -> id -> centroid and id -> range were three dependent ones)
    const uint4* slot_desc;     // 2 x uint4 per slot
    const int2* slot_range;     // (child_off[id], child_off[id + 1]) of the slot's child
};

// ---- per-feature descent (TemplatedVocabulary.h:1216-1259) ----
// frame-major descriptor array [batch][cap][32]; feature f of frame b is live when f < n[b] (n == nullptr: all live)
__global__ __launch_bounds__(256) void k_vocab_descend(VocabDev V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ n_per_frame,
                                                      int cap, int total, int levelsup,
                                                      uint32_t* __restrict__ word, double* __restrict__ weight, uint32_t* __restrict__ node)
{
    const int sub = threadIdx.x & 15;                           // child slot inside the 16-lane group
    const int f = blockIdx.x * 16 + (threadIdx.x >> 4);        // flat feature index = frame * cap + feature
    const bool in_range = f < total;
    const int frame = in_range ? f / cap : 0;
    const bool live = in_range && (n_per_frame == nullptr || (f - frame * cap) < n_per_frame[frame]);
    uint4 d0 = make_uint4(0, 0, 0, 0), d1 = d0;
    if (live) {
        const uint4* p = (const uint4*)(desc + (size_t)f * 32);
        d0 = p[0]; d1 = p[1];
    }
    const int nid_level = V.L - levelsup;
    uint32_t final_id = 0, nid = 0;
    bool nid_set = nid_level <= 0;
    int level = 0;
    bool going = live;
    // all 16 lanes of a group hold the same (final_id, c0, c1, going); groups of one wave leave the loop together via the ballot
    int c0 = V.child_off[0], c1 = V.child_off[1];          // the root's children
    while (__ballot(going) != 0ull) {
        uint32_t key = 0xFFFFFFFFu;
        uint32_t my_id = 0;
        int2 my_range = make_int2(0, 0);
        if (going) {
            const int c = c0 + sub;                             // the first 16 children: one load round for id, range and centroid
            if (c < c1) {
                my_id = V.child_id[c];
                my_range = V.slot_range[c];
                const uint4 a = V.slot_desc[2 * (size_t)c], b = V.slot_desc[2 * (size_t)c + 1];
                const int dist = __popc(a.x ^ d0.x) + __popc(a.y ^ d0.y) + __popc(a.z ^ d0.z) + __popc(a.w ^ d0.w) +
                                 __popc(b.x ^ d1.x) + __popc(b.y ^ d1.y) + __popc(b.z ^ d1.z) + __popc(b.w ^ d1.w);
                key = ((uint32_t)dist << 20) | (uint32_t)sub;          // first minimum in children order
            }
            for (int cc = c + 16; cc < c1; cc += 16) {          // (nodes with more than 16 children)
                const uint4 a = V.slot_desc[2 * (size_t)cc], b = V.slot_desc[2 * (size_t)cc + 1];
                const int dist = __popc(a.x ^ d0.x) + __popc(a.y ^ d0.y) + __popc(a.z ^ d0.z) + __popc(a.w ^ d0.w) +
                                 __popc(b.x ^ d1.x) + __popc(b.y ^ d1.y) + __popc(b.z ^ d1.z) + __popc(b.w ^ d1.w);
                key = min(key, ((uint32_t)dist << 20) | (uint32_t)(cc - c0));
            }
        }
        for (int o = 8; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o));
        const int best = (int)(key & 0xFFFFFu);
        // the winner's id and child range: from the lane that holds it (slots below 16), else one more load
        const uint32_t w_id = (uint32_t)__shfl((int)my_id, best & 15, 16);
        const int w_c0 = __shfl(my_range.x, best & 15, 16), w_c1 = __shfl(my_range.y, best & 15, 16);
        if (going) {
            if (best < 16) { final_id = w_id; c0 = w_c0; c1 = w_c1; }
            else { final_id = V.child_id[c0 + best]; const int2 r = V.slot_range[c0 + best]; c0 = r.x; c1 = r.y; }
            level++;
            if (level == nid_level) { nid = final_id; nid_set = true; }
            going = c1 > c0 && level < 64;                      // !isLeaf(); the bound ends a malformed (cyclic) tree
        }
    }
    if (live && sub == 0) {
        if (!nid_set) nid = final_id;       // leaf above the requested level: the reference leaves *nid unwritten
        word[f] = (uint32_t)V.word_id[final_id];
        weight[f] = V.weight[final_id];
        node[f] = nid;
    }
}

// ---- per-frame assembly of BowVector and FeatureVector ----
// Bitonic sort of n_pow2 64-bit keys in LDS by the 4 waves of the workgroup.  Wave w owns the pairs [w n/8, (w+1) n/8), i.e. the elements
// of chunk w (n/4 elements): every exchange with a stride below the chunk size stays inside the chunk, whose LDS accesses are the
// owning wave's own and therefore ordered (a wave-level fence keeps the compiler from reordering them) -- only the three passes whose
// stride reaches across chunks need a workgroup barrier (it was 55 barriers for 1024 keys).
__device__ __forceinline__ void bitonic_sort(unsigned long long* s, int n_pow2)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = n_pow2 >> 2, pairs = n_pow2 >> 3;               // elements / pairs per wave (n_pow2 >= 8: a lane may idle)
    bool need_block = true;                                           // the keys were written by other waves
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool cross = j >= chunk || n_pow2 < 8;
            if (cross || need_block) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
            need_block = cross;                                       // after a cross-chunk pass the next pass reads other waves' writes
            if (n_pow2 >= 8) {
                for (int tw = lane; tw < pairs; tw += 64) {
                    const int t = wave * pairs + tw;
                    const int lo = 2 * t - (t & (j - 1));             // index with bit j clear
                    const int hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const unsigned long long a = s[lo], b = s[hi];
                    if ((a > b) == up) { s[lo] = b; s[hi] = a; }
                }
            } else {
                for (int t = threadIdx.x; t < (n_pow2 >> 1); t += blockDim.x) {
                    const int lo = 2 * t - (t & (j - 1)), hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const unsigned long long a = s[lo], b = s[hi];
                    if ((a > b) == up) { s[lo] = b; s[hi] = a; }
                }
            }
        }
    __syncthreads();
}

// exclusive prefix sum of one flag per element (n <= 8192) -> ranks; returns the total.  s_cnt: [blockDim.x] ints.
__device__ __forceinline__ int block_rank(const unsigned long long* s, int n, bool by_high32_change, int* s_cnt, int* my_base_out, int per)
{
    // each thread owns a contiguous run of `per` elements: count heads in the run
    const int t = threadIdx.x;
    const int b = t * per, e = min(b + per, n);
    int c = 0;
    for (int i = b; i < e; i++) {
        const bool head = (s[i] != ~0ull) && (i == 0 || (s[i] >> 32) != (s[i - 1] >> 32));
        c += (by_high32_change ? head : (s[i] != ~0ull)) ? 1 : 0;
    }
    // scan inside the wave by shuffles, the four wave totals through LDS: two barriers (it was a 16-barrier Hillis-Steele scan)
    int incl = c;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if ((t & 63) >= o) incl += v; }
    if ((t & 63) == 63) s_cnt[t >> 6] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < (t >> 6); w++) base += s_cnt[w];
    const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    *my_base_out = base + incl - c;
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void k_vocab_assemble(const int32_t* __restrict__ n_per_frame, int n_fixed, int cap, int n_pow2_max,
                                                       const uint32_t* __restrict__ word, const double* __restrict__ weight,
                                                       const uint32_t* __restrict__ node,
                                                       uint32_t* __restrict__ bow_id, double* __restrict__ bow_val, int32_t* __restrict__ n_bow,
                                                       uint32_t* __restrict__ fv_node, int32_t* __restrict__ fv_off, uint32_t* __restrict__ fv_feat,
                                                       int32_t* __restrict__ n_fv)
{
    extern __shared__ __align__(16) unsigned long long s_key[];        // n_pow2_max keys, then n_pow2_max doubles: the BowVector values (for the norm)
    double* const s_val = (double*)(s_key + n_pow2_max);
    __shared__ int s_cnt[256];
    __shared__ double s_norm;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_per_frame ? n_per_frame[frame] : n_fixed, cap);
    const size_t fo = (size_t)frame * cap;
    int n_pow2 = 2;                             // this frame's own sort size (the launch's LDS is sized for the arrays' capacity)
    while (n_pow2 < n) n_pow2 <<= 1;
    n_pow2 = min(n_pow2, n_pow2_max);
    const int per = (n_pow2 + 255) / 256;

    // ---------- FeatureVector: keys (node id, feature index), only features whose word is not stopped (w > 0) ----------
    for (int i = tid; i < n_pow2; i += 256)
        s_key[i] = (i < n && weight[fo + i] > 0) ? (((unsigned long long)node[fo + i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    bitonic_sort(s_key, n_pow2);
    int base;
    const int n_nodes = block_rank(s_key, n_pow2, true, s_cnt, &base, per);
    {
        const int b = tid * per, e = min(b + per, n_pow2);
        int r = base;
        for (int i = b; i < e; i++) {
            if (s_key[i] == ~0ull) break;
            fv_feat[fo + i] = (uint32_t)(s_key[i] & 0xFFFFFFFFull);
            if (i == 0 || (s_key[i] >> 32) != (s_key[i - 1] >> 32)) {
                fv_node[fo + r] = (uint32_t)(s_key[i] >> 32);
                fv_off[(size_t)frame * (cap + 1) + r] = i;
                r++;
            }
        }
    }
    __syncthreads();
    int base2;
    const int n_used = block_rank(s_key, n_pow2, false, s_cnt, &base2, per);    // features that made it into the vectors
    if (tid == 0) { fv_off[(size_t)frame * (cap + 1) + n_nodes] = n_used; n_fv[frame] = n_nodes; }
    __syncthreads();

    // ---------- BowVector: keys (word id, feature index); a word's weights are added in feature order (addWeight) ----------
    for (int i = tid; i < n_pow2; i += 256)
        s_key[i] = (i < n && weight[fo + i] > 0) ? (((unsigned long long)word[fo + i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    bitonic_sort(s_key, n_pow2);
    const int n_words = block_rank(s_key, n_pow2, true, s_cnt, &base, per);
    {
        const int b = tid * per, e = min(b + per, n_pow2);
        int r = base;
        for (int i = b; i < e; i++) {
            if (s_key[i] == ~0ull) break;
            if (i == 0 || (s_key[i] >> 32) != (s_key[i - 1] >> 32)) {
                const unsigned long long w_id = s_key[i] >> 32;
                double acc = weight[fo + (s_key[i] & 0xFFFFFFFFull)];
                for (int j = i + 1; j < n_pow2 && (s_key[j] >> 32) == w_id; j++) acc += weight[fo + (s_key[j] & 0xFFFFFFFFull)];
                bow_id[fo + r] = (uint32_t)w_id;
                bow_val[fo + r] = acc;
                s_val[r] = acc;
                r++;
            }
        }
    }
    __syncthreads();
    // normalize(L1) (BowVector.cpp:62-84): the sum runs in word-id order, sequentially, to reproduce the reference's rounding -- one lane,
    // the values from LDS eight at a time (the additions are the dependent chain; reading them back from global memory one by one
    // was a third of this kernel)
    if (tid == 0) {
        double norm = 0.0;
        int k = 0;
        for (; k + 8 <= n_words; k += 8) {
            const double v0 = s_val[k], v1 = s_val[k + 1], v2 = s_val[k + 2], v3 = s_val[k + 3], v4 = s_val[k + 4], v5 = s_val[k + 5], v6 = s_val[k + 6], v7 = s_val[k + 7];
            norm += fabs(v0); norm += fabs(v1); norm += fabs(v2); norm += fabs(v3); norm += fabs(v4); norm += fabs(v5); norm += fabs(v6); norm += fabs(v7);
        }
        for (; k < n_words; k++) norm += fabs(s_val[k]);
        s_norm = norm;
        n_bow[frame] = n_words;
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
        for (int k = tid; k < n_words; k += 256) bow_val[fo + k] = s_val[k] / norm;
}

}  // namespace orbv

struct orbv_vocab {
    int device = 0;
    hipStream_t stream = nullptr;
    orbv::VocabDev V{};
    uint8_t* d_tree = nullptr;
    // scratch for the host-buffer entry points and the per-feature triples of the batched call
    uint8_t* d_scratch = nullptr;
    size_t scratch_cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= scratch_cap) return ORBX_OK;
        if (d_scratch) (void)hipFree(d_scratch);
        d_scratch = nullptr; scratch_cap = 0;
        const size_t cap = std::max(bytes * 2, (size_t)1 << 20);
        if (hipMalloc((void**)&d_scratch, cap) != hipSuccess) return fail(ORBX_ERR_HIP, "hipMalloc(%zu) failed", cap);
        scratch_cap = cap;
        return ORBX_OK;
    }
};

namespace {
inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

int pow2_at_least(int n) { int p = 2; while (p < n) p <<= 1; return p; }

// runs both kernels on device-resident descriptors; triples live at d_word/d_weight/d_node (batch*cap entries)
int enqueue_transform(orbv_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int n_fixed, int batch, int cap, int levelsup,
                      uint32_t* d_word, double* d_weight, uint32_t* d_node,
                      uint32_t* d_bow_id, double* d_bow_val, int32_t* d_n_bow, uint32_t* d_fv_node, int32_t* d_fv_off, uint32_t* d_fv_feat,
                      int32_t* d_n_fv, hipStream_t st, bool assemble)
{
    const long long total = (long long)batch * cap;
    if (total <= 0) return ORBX_OK;
    if (total > (1ll << 30)) return fail(ORBX_ERR_ARG, "batch x cap too large");
    // features beyond n[b] of a frame are not descended; a fixed count is expressed by cap == n
    hipLaunchKernelGGL(orbv::k_vocab_descend, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, v->V, d_desc, d_n, cap, (int)total, levelsup,
                       d_word, d_weight, d_node);
    if (assemble) {
        const int n_pow2 = pow2_at_least(cap);
        if (n_pow2 > 8192) return fail(ORBX_ERR_ARG, "more than 8192 features per frame are not supported by the assembly kernel");
        hipLaunchKernelGGL(orbv::k_vocab_assemble, dim3(batch), dim3(256), (size_t)n_pow2 * 16, st, d_n, n_fixed, cap, n_pow2, d_word, d_weight, d_node,
                           d_bow_id, d_bow_val, d_n_bow, d_fv_node, d_fv_off, d_fv_feat, d_n_fv);
    }
    ORBV_HIP(hipGetLastError());
    return ORBX_OK;
}
}  // namespace

extern "C" {

int orbv_create(int device, const OrbvVocabulary* voc, orbv_vocab** out)
{
    if (!out) return fail(ORBX_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!voc || voc->n_nodes < 2 || voc->L < 1 || !voc->child_off || !voc->child_id || !voc->desc || !voc->weight || !voc->word_id)
        return fail(ORBX_ERR_ARG, "bad vocabulary");
    const int nn = voc->n_nodes;
    if (voc->child_off[0] != 0 || voc->child_off[1] <= 0) return fail(ORBX_ERR_ARG, "the root has no children");
    for (int i = 0; i < nn; i++) {
        if (voc->child_off[i + 1] < voc->child_off[i]) return fail(ORBX_ERR_ARG, "child_off is not monotone at node %d", i);
        if (voc->child_off[i + 1] - voc->child_off[i] >= (1 << 20)) return fail(ORBX_ERR_ARG, "node %d has too many children", i);
    }
    const int n_child = voc->child_off[nn];
    for (int c = 0; c < n_child; c++)
        if (voc->child_id[c] == 0 || voc->child_id[c] >= (uint32_t)nn) return fail(ORBX_ERR_ARG, "child id %u out of range", voc->child_id[c]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ORBX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(ORBX_ERR_ARG, "device %d out of range", device);
    ORBV_HIP(hipSetDevice(device));
    orbv_vocab* v = new orbv_vocab();
    v->device = device;
    const size_t o_off = 0, o_cid = al(o_off + sizeof(int32_t) * ((size_t)nn + 1)), o_desc = al(o_cid + sizeof(uint32_t) * (size_t)std::max(n_child, 1)),
                 o_w = al(o_desc + (size_t)nn * 32), o_word = al(o_w + sizeof(double) * (size_t)nn),
                 o_sdesc = al(o_word + sizeof(int32_t) * (size_t)nn), o_srange = al(o_sdesc + 32 * (size_t)std::max(n_child, 1)),
                 total = al(o_srange + sizeof(int2) * (size_t)std::max(n_child, 1));
    // per child slot: centroid and child range of the slot's child (the descent then needs one round trip per level)
    std::vector<uint8_t> sdesc(32 * (size_t)std::max(n_child, 1));
    std::vector<int2> srange((size_t)std::max(n_child, 1));
    for (int c = 0; c < n_child; c++) {
        const uint32_t id = voc->child_id[c];
        std::memcpy(&sdesc[32 * (size_t)c], (const uint8_t*)voc->desc + 32 * (size_t)id, 32);
        srange[c] = make_int2(voc->child_off[id], voc->child_off[id + 1]);
    }
    if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&v->d_tree, total) != hipSuccess) {
        orbv_destroy(v);
        return fail(ORBX_ERR_HIP, "vocabulary allocation of %zu bytes failed", total);
    }
    bool ok = hipMemcpy(v->d_tree + o_off, voc->child_off, sizeof(int32_t) * ((size_t)nn + 1), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(v->d_tree + o_cid, voc->child_id, sizeof(uint32_t) * (size_t)n_child, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(v->d_tree + o_desc, voc->desc, (size_t)nn * 32, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(v->d_tree + o_w, voc->weight, sizeof(double) * (size_t)nn, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(v->d_tree + o_word, voc->word_id, sizeof(int32_t) * (size_t)nn, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(v->d_tree + o_sdesc, sdesc.data(), sdesc.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(v->d_tree + o_srange, srange.data(), sizeof(int2) * srange.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { orbv_destroy(v); return fail(ORBX_ERR_HIP, "vocabulary upload failed"); }
    if (hipFuncSetAttribute((const void*)orbv::k_vocab_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16) != hipSuccess) {
        orbv_destroy(v);
        return fail(ORBX_ERR_HIP, "hipFuncSetAttribute failed");
    }
    v->V.n_nodes = nn; v->V.L = voc->L;
    v->V.child_off = (const int32_t*)(v->d_tree + o_off); v->V.child_id = (const uint32_t*)(v->d_tree + o_cid);
    v->V.desc = (const uint4*)(v->d_tree + o_desc); v->V.weight = (const double*)(v->d_tree + o_w); v->V.word_id = (const int32_t*)(v->d_tree + o_word);
    v->V.slot_desc = (const uint4*)(v->d_tree + o_sdesc); v->V.slot_range = (const int2*)(v->d_tree + o_srange);
    *out = v;
    return ORBX_OK;
}

void orbv_destroy(orbv_vocab* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) { (void)hipStreamSynchronize(v->stream); (void)hipStreamDestroy(v->stream); }
    if (v->d_tree) (void)hipFree(v->d_tree);
    if (v->d_scratch) (void)hipFree(v->d_scratch);
    delete v;
}

int orbv_transform_features(orbv_vocab* v, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node)
{
    if (!v || n < 0 || (n > 0 && (!desc || !word || !weight || !node))) return fail(ORBX_ERR_ARG, "bad arguments");
    if (n == 0) return ORBX_OK;
    ORBV_HIP(hipSetDevice(v->device));
    const size_t o_d = 0, o_wd = al((size_t)n * 32), o_wt = al(o_wd + 4 * (size_t)n), o_nd = al(o_wt + 8 * (size_t)n), total = al(o_nd + 4 * (size_t)n);
    int r = v->ensure(total);
    if (r) return r;
    uint8_t* b = v->d_scratch;
    ORBV_HIP(hipMemcpyAsync(b + o_d, desc, (size_t)n * 32, hipMemcpyHostToDevice, v->stream));
    r = enqueue_transform(v, b + o_d, nullptr, n, 1, n, levelsup, (uint32_t*)(b + o_wd), (double*)(b + o_wt), (uint32_t*)(b + o_nd),
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, v->stream, false);
    if (r) return r;
    ORBV_HIP(hipMemcpyAsync(word, b + o_wd, 4 * (size_t)n, hipMemcpyDeviceToHost, v->stream));
    ORBV_HIP(hipMemcpyAsync(weight, b + o_wt, 8 * (size_t)n, hipMemcpyDeviceToHost, v->stream));
    ORBV_HIP(hipMemcpyAsync(node, b + o_nd, 4 * (size_t)n, hipMemcpyDeviceToHost, v->stream));
    ORBV_HIP(hipStreamSynchronize(v->stream));
    return ORBX_OK;
}

int orbv_transform(orbv_vocab* v, const uint8_t* desc, int n, int levelsup, uint32_t* bow_id, double* bow_val, int32_t* n_bow,
                   uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int32_t* n_fv_nodes)
{
    if (!v || n < 0 || !n_bow || !n_fv_nodes || !fv_off || (n > 0 && (!desc || !bow_id || !bow_val || !fv_node || !fv_feat)))
        return fail(ORBX_ERR_ARG, "bad arguments");
    if (n == 0) { *n_bow = 0; *n_fv_nodes = 0; fv_off[0] = 0; return 0; }
    if (n > 8192) return fail(ORBX_ERR_ARG, "more than 8192 features per frame are not supported");
    ORBV_HIP(hipSetDevice(v->device));
    const size_t N = (size_t)n;
    const size_t o_d = 0, o_wd = al(N * 32), o_wt = al(o_wd + 4 * N), o_nd = al(o_wt + 8 * N), o_bi = al(o_nd + 4 * N), o_bv = al(o_bi + 4 * N),
                 o_fn = al(o_bv + 8 * N), o_fo = al(o_fn + 4 * N), o_ff = al(o_fo + 4 * (N + 1)), o_cnt = al(o_ff + 4 * N), total = al(o_cnt + 8);
    int r = v->ensure(total);
    if (r) return r;
    uint8_t* b = v->d_scratch;
    ORBV_HIP(hipMemcpyAsync(b + o_d, desc, N * 32, hipMemcpyHostToDevice, v->stream));
    r = enqueue_transform(v, b + o_d, nullptr, n, 1, n, levelsup, (uint32_t*)(b + o_wd), (double*)(b + o_wt), (uint32_t*)(b + o_nd),
                          (uint32_t*)(b + o_bi), (double*)(b + o_bv), (int32_t*)(b + o_cnt), (uint32_t*)(b + o_fn), (int32_t*)(b + o_fo),
                          (uint32_t*)(b + o_ff), (int32_t*)(b + o_cnt) + 1, v->stream, true);
    if (r) return r;
    int32_t cnt[2] = {0, 0};
    ORBV_HIP(hipMemcpyAsync(cnt, b + o_cnt, 8, hipMemcpyDeviceToHost, v->stream));
    ORBV_HIP(hipStreamSynchronize(v->stream));
    *n_bow = cnt[0]; *n_fv_nodes = cnt[1];
    if (cnt[0] > 0) {
        ORBV_HIP(hipMemcpyAsync(bow_id, b + o_bi, 4 * (size_t)cnt[0], hipMemcpyDeviceToHost, v->stream));
        ORBV_HIP(hipMemcpyAsync(bow_val, b + o_bv, 8 * (size_t)cnt[0], hipMemcpyDeviceToHost, v->stream));
    }
    ORBV_HIP(hipMemcpyAsync(fv_off, b + o_fo, 4 * ((size_t)cnt[1] + 1), hipMemcpyDeviceToHost, v->stream));
    if (cnt[1] > 0) ORBV_HIP(hipMemcpyAsync(fv_node, b + o_fn, 4 * (size_t)cnt[1], hipMemcpyDeviceToHost, v->stream));
    ORBV_HIP(hipStreamSynchronize(v->stream));
    const int used = fv_off[cnt[1]];
    if (used > 0) {
        ORBV_HIP(hipMemcpyAsync(fv_feat, b + o_ff, 4 * (size_t)used, hipMemcpyDeviceToHost, v->stream));
        ORBV_HIP(hipStreamSynchronize(v->stream));
    }
    return used;
}

int orbv_transform_batch_device(orbv_vocab* v, const uint8_t* d_desc, const int32_t* d_n, int batch, int cap, int levelsup,
                                uint32_t* d_bow_id, double* d_bow_val, int32_t* d_n_bow,
                                uint32_t* d_fv_node, int32_t* d_fv_off, uint32_t* d_fv_feat, int32_t* d_n_fv, void* stream)
{
    if (!v || batch < 1 || cap < 1 || !d_desc || !d_n || !d_bow_id || !d_bow_val || !d_n_bow || !d_fv_node || !d_fv_off || !d_fv_feat || !d_n_fv)
        return fail(ORBX_ERR_ARG, "bad arguments");
    ORBV_HIP(hipSetDevice(v->device));
    const size_t T = (size_t)batch * cap;
    const size_t o_wd = 0, o_wt = al(4 * T), o_nd = al(o_wt + 8 * T), total = al(o_nd + 4 * T);
    const int r = v->ensure(total);
    if (r) return r;
    uint8_t* b = v->d_scratch;
    return enqueue_transform(v, d_desc, d_n, 0, batch, cap, levelsup, (uint32_t*)(b + o_wd), (double*)(b + o_wt), (uint32_t*)(b + o_nd),
                             d_bow_id, d_bow_val, d_n_bow, d_fv_node, d_fv_off, d_fv_feat, d_n_fv, (hipStream_t)stream, true);
}

}  // extern "C"
