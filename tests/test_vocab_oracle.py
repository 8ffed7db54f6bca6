"""DBoW2 transform restatement (oracle/dbow_oracle.cpp; reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259)
against a plain-Python descent of the same tree -- PARITY UNPINNED: ORBvoc.txt is not shipped, the trees are synthetic."""
import numpy as np

from oracle_api import oracle_transform, oracle_transform_features

_POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def _descend(voc, f, levelsup):
    nid_level = voc["L"] - levelsup
    node = 0; nid = 0; nid_set = nid_level <= 0; level = 0
    while True:
        level += 1
        ch = voc["child_id"][voc["child_off"][node]:voc["child_off"][node + 1]]
        d = _POP[voc["desc"][ch] ^ f].sum(1)
        node = int(ch[int(np.argmin(d))])                   # argmin = first minimum
        if level == nid_level:
            nid = node; nid_set = True
        if voc["child_off"][node + 1] == voc["child_off"][node]:
            break
    return int(voc["word_id"][node]), float(voc["weight"][node]), (nid if nid_set else node)


def test_transform_features_against_python(oracle, synth):
    for seed, L, levelsup, shuffle in ((0, 3, 1, False), (1, 4, 2, True), (2, 2, 4, False)):
        voc = synth.make_vocabulary(seed, k=6, L=L, shuffle_ids=shuffle)
        rs = np.random.RandomState(seed)
        # descriptors near random centroids so that deep nodes are actually reached
        src = rs.randint(1, voc["n_nodes"], 200)
        desc = voc["desc"][src] ^ (rs.uniform(size=(200, 32)) < 0.02).astype(np.uint8)
        w, wt, nd = oracle_transform_features(oracle, voc, desc, levelsup)
        for i in range(200):
            assert (int(w[i]), float(wt[i]), int(nd[i])) == _descend(voc, desc[i], levelsup)


def test_transform_containers(oracle, synth):
    voc = synth.make_vocabulary(3, k=8, L=3, stop_frac=0.15)
    rs = np.random.RandomState(5)
    desc = voc["desc"][rs.randint(1, voc["n_nodes"], 500)] ^ (rs.uniform(size=(500, 32)) < 0.03).astype(np.uint8)
    (bi, bv), (fn, fo, ff) = oracle_transform(oracle, voc, desc, 2)
    w, wt, nd = oracle_transform_features(oracle, voc, desc, 2)
    used = np.nonzero(wt > 0)[0]
    assert len(used) < 500                                   # some words are stopped
    # BowVector: ascending unique ids, L1-normalised tf-idf
    assert (np.diff(bi.astype(np.int64)) > 0).all() and set(bi) == set(w[used])
    assert abs(bv.sum() - 1.0) < 1e-12
    raw = np.array([wt[used][w[used] == i].sum() for i in bi])
    np.testing.assert_allclose(bv, raw / raw.sum(), rtol=1e-13)
    # FeatureVector: ascending nodes, features of a node ascending (insertion order), every used feature exactly once
    assert (np.diff(fn.astype(np.int64)) > 0).all() and fo[0] == 0 and fo[-1] == len(used) == len(ff)
    assert sorted(ff) == list(used)
    for k in range(len(fn)):
        seg = ff[fo[k]:fo[k + 1]]
        assert (np.diff(seg.astype(np.int64)) > 0).all() and (nd[seg] == fn[k]).all()
    # empty input
    (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(oracle, voc, desc[:0], 2)
    assert len(bi0) == 0 and len(fn0) == 0 and list(fo0) == [0]
