/*
 * dbow_oracle.cpp -- CPU ORACLE (test infrastructure, not the product) for DBoW2's descriptor -> word transform,
 * the step that produces Frame::mBowVec / mFeatVec right before ORBmatcher::SearchByBoW
 * (reference src/Frame.cc:825-832 -> mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)).
 *
 * Restates, from the vendored sources under /root/reference/Thirdparty/DBoW2/DBoW2:
 *   TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)   TemplatedVocabulary.h:1216-1259
 *   TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup), TF_IDF branch   :1127-1193
 *   BowVector::addWeight / normalize(L1)                                      BowVector.cpp:33-84
 *   FeatureVector::addFeature                                                 FeatureVector.cpp:31-47
 *   FORB::distance (256-bit Hamming)                                          FORB.cpp:65-84
 * The ORB vocabulary (ORBvoc.txt: k = 10, L = 6, TF_IDF weighting, L1_NORM scoring) is not shipped with the reference,
 * so the tree itself is synthetic in the tests; PARITY UNPINNED (no fixture of the reference covers this path).
 *
 * One point where the reference is undefined: when a leaf is shallower than the level `L - levelsup`, *nid is never
 * written (:1252-1253).  Here, as in the kernels, the node id is then the leaf itself.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

#include "oracle.h"

namespace {
int hamming256(const uint8_t* a, const uint8_t* b)
{
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32);
    std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) + __builtin_popcountll(x[2] ^ y[2]) +
           __builtin_popcountll(x[3] ^ y[3]);
}

void transform_one(const OracleVocab* V, const uint8_t* f, int levelsup, uint32_t& word, double& weight, uint32_t& nid)
{
    const int nid_level = V->L - levelsup;
    bool nid_set = false;
    nid = 0;
    if (nid_level <= 0) nid_set = true;                                  // root (:1226)
    uint32_t final_id = 0;
    int current_level = 0;
    do {
        ++current_level;
        const int c0 = V->child_off[final_id], c1 = V->child_off[final_id + 1];
        final_id = V->child_id[c0];
        int best_d = hamming256(f, V->desc + (size_t)final_id * 32);
        for (int c = c0 + 1; c < c1; c++) {
            const uint32_t id = V->child_id[c];
            const int d = hamming256(f, V->desc + (size_t)id * 32);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (current_level == nid_level) { nid = final_id; nid_set = true; }
    } while (V->child_off[final_id + 1] > V->child_off[final_id]);      // !isLeaf()
    if (!nid_set) nid = final_id;
    word = (uint32_t)V->word_id[final_id];
    weight = V->weight[final_id];
}
}  // namespace

extern "C" {

void dbow_oracle_transform_features(const OracleVocab* V, const uint8_t* desc, int n, int levelsup,
                                    uint32_t* word, double* weight, uint32_t* node)
{
    for (int i = 0; i < n; i++) transform_one(V, desc + (size_t)i * 32, levelsup, word[i], weight[i], node[i]);
}

int dbow_oracle_transform(const OracleVocab* V, const uint8_t* desc, int n, int levelsup,
                          uint32_t* bow_id, double* bow_val, int32_t* n_bow,
                          uint32_t* fv_node, int32_t* fv_off, uint32_t* fv_feat, int32_t* n_fv_nodes)
{
    std::map<uint32_t, double> v;
    std::map<uint32_t, std::vector<uint32_t> > fv;
    for (int i = 0; i < n; i++) {
        uint32_t id, nid;
        double w;
        transform_one(V, desc + (size_t)i * 32, levelsup, id, w, nid);
        if (w > 0) {                                                     // not stopped (:1157)
            std::map<uint32_t, double>::iterator it = v.lower_bound(id);
            if (it != v.end() && !(id < it->first)) it->second += w;     // addWeight
            else v.insert(it, std::make_pair(id, w));
            fv[nid].push_back((uint32_t)i);                              // addFeature
        }
    }
    double norm = 0.0;                                                   // normalize(L1)
    for (std::map<uint32_t, double>::iterator it = v.begin(); it != v.end(); ++it) norm += std::fabs(it->second);
    if (norm > 0.0)
        for (std::map<uint32_t, double>::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
    int k = 0;
    for (std::map<uint32_t, double>::iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_id[k] = it->first; bow_val[k] = it->second; }
    *n_bow = k;
    int nn = 0, off = 0;
    for (std::map<uint32_t, std::vector<uint32_t> >::iterator it = fv.begin(); it != fv.end(); ++it, ++nn) {
        fv_node[nn] = it->first;
        fv_off[nn] = off;
        for (size_t j = 0; j < it->second.size(); j++) fv_feat[off++] = it->second[j];
    }
    fv_off[nn] = off;
    *n_fv_nodes = nn;
    return off;
}

}  // extern "C"
