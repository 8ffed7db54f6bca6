"""HIP vocabulary transform (csrc/dbow_vocab.hip) against the CPU restatement of DBoW2's TemplatedVocabulary::transform
(reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259): word / node ids and the containers must be identical,
BowVector values bit-identical (the kernel adds and normalises in the reference's order)."""
import numpy as np
import pytest

from oracle_api import oracle_transform, oracle_transform_features

pytestmark = pytest.mark.gpu


def _descs(voc, n, seed, p=0.03):
    rs = np.random.RandomState(100 + seed)
    return np.ascontiguousarray(voc["desc"][rs.randint(1, voc["n_nodes"], n)] ^ (rs.uniform(size=(n, 32)) < p).astype(np.uint8))


@pytest.mark.parametrize("seed,k,L,levelsup,shuffle,n", [(0, 10, 3, 1, False, 1000), (1, 6, 4, 2, True, 1003), (2, 10, 2, 4, False, 17),
                                                         (3, 20, 2, 1, False, 500), (4, 3, 6, 4, True, 2000)])
def test_transform_features(pkg, oracle, synth, seed, k, L, levelsup, shuffle, n):
    voc = synth.make_vocabulary(seed, k=k, L=L, shuffle_ids=shuffle)
    desc = _descs(voc, n, seed)
    w0, wt0, nd0 = oracle_transform_features(oracle, voc, desc, levelsup)
    v = pkg.Vocabulary(voc)
    try:
        w1, wt1, nd1 = v.transform_features(desc, levelsup)
    finally:
        v.close()
    np.testing.assert_array_equal(w1, w0); np.testing.assert_array_equal(nd1, nd0); np.testing.assert_array_equal(wt1, wt0)


@pytest.mark.parametrize("seed,n", [(0, 1000), (1, 1), (2, 0), (3, 2048), (4, 5000), (5, 777)])
def test_transform_containers(pkg, oracle, synth, seed, n):
    voc = synth.make_vocabulary(seed + 10, k=8, L=3, stop_frac=0.08)
    desc = _descs(voc, n, seed)
    (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(oracle, voc, desc, 2)
    v = pkg.Vocabulary(voc)
    try:
        (bi1, bv1), (fn1, fo1, ff1) = v.transform(desc, 2)
    finally:
        v.close()
    np.testing.assert_array_equal(bi1, bi0); np.testing.assert_array_equal(bv1, bv0)      # bit-identical doubles
    np.testing.assert_array_equal(fn1, fn0); np.testing.assert_array_equal(fo1, fo0); np.testing.assert_array_equal(ff1, ff0)


def test_transform_batch_device_feeds_search_by_bow(pkg, oracle, synth):
    """extractor descriptors stay in HBM: ragged batch [B][cap][32] + n[B] -> per-frame containers, then SearchByBoW on them"""
    torch = pytest.importorskip("torch")
    voc = synth.make_vocabulary(21, k=10, L=3)
    B, cap = 5, 1200
    ns = np.array([1000, 0, 1200, 37, 512], np.int32)
    dev = torch.device("cuda", 0)
    host = np.zeros((B, cap, 32), np.uint8)
    for b in range(B):
        host[b, :ns[b]] = _descs(voc, int(ns[b]), 30 + b)
    d_desc = torch.from_numpy(host).to(dev); d_n = torch.from_numpy(ns).to(dev)
    z = lambda dt, m: torch.zeros(B * m, dtype=dt, device=dev)
    d_bi, d_bv, d_nb = z(torch.int32, cap), z(torch.float64, cap), torch.zeros(B, dtype=torch.int32, device=dev)
    d_fn, d_fo, d_ff, d_nf = z(torch.int32, cap), z(torch.int32, cap + 1), z(torch.int32, cap), torch.zeros(B, dtype=torch.int32, device=dev)
    v = pkg.Vocabulary(voc)
    try:
        v.transform_batch_device(d_desc.data_ptr(), d_n.data_ptr(), B, cap, 2, d_bi.data_ptr(), d_bv.data_ptr(), d_nb.data_ptr(),
                                 d_fn.data_ptr(), d_fo.data_ptr(), d_ff.data_ptr(), d_nf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        v.close()
    bi = d_bi.cpu().numpy().view(np.uint32).reshape(B, cap); bv = d_bv.cpu().numpy().reshape(B, cap); nb = d_nb.cpu().numpy()
    fn = d_fn.cpu().numpy().view(np.uint32).reshape(B, cap); fo = d_fo.cpu().numpy().reshape(B, cap + 1)
    ff = d_ff.cpu().numpy().view(np.uint32).reshape(B, cap); nf = d_nf.cpu().numpy()
    for b in range(B):
        (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(oracle, voc, host[b, :ns[b]], 2)
        assert nb[b] == len(bi0) and nf[b] == len(fn0)
        np.testing.assert_array_equal(bi[b, :nb[b]], bi0); np.testing.assert_array_equal(bv[b, :nb[b]], bv0)
        np.testing.assert_array_equal(fn[b, :nf[b]], fn0); np.testing.assert_array_equal(fo[b, :nf[b] + 1], fo0)
        np.testing.assert_array_equal(ff[b, :fo0[-1]], ff0)


def test_transform_full_size_tree(pkg, oracle, synth):
    """the shape of ORBvoc.txt (k = 10, L = 6: 1 111 111 nodes, 35 MB of centroids), levelsup = 4 as Frame::ComputeBoW uses it"""
    voc = synth.make_vocabulary_fast(1, k=10, L=6)
    rs = np.random.RandomState(9)
    desc = rs.randint(0, 256, size=(1500, 32)).astype(np.uint8)
    (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(oracle, voc, desc, 4)
    v = pkg.Vocabulary(voc)
    try:
        (bi1, bv1), (fn1, fo1, ff1) = v.transform(desc, 4)
        w1, wt1, nd1 = v.transform_features(desc[:64], 4)
    finally:
        v.close()
    np.testing.assert_array_equal(bi1, bi0); np.testing.assert_array_equal(bv1, bv0)
    np.testing.assert_array_equal(fn1, fn0); np.testing.assert_array_equal(fo1, fo0); np.testing.assert_array_equal(ff1, ff0)
    # size-independent properties: every word is a leaf, the node 4 levels up is its ancestor at depth 2, sum(bow) = 1
    n_inner = (10 ** 6 - 1) // 9
    assert (w1 + n_inner < voc["n_nodes"]).all() and abs(bv1.sum() - 1.0) < 1e-12
    leaf = w1.astype(np.int64) + n_inner
    anc = leaf
    for _ in range(4):
        anc = (anc - 1) // 10
    np.testing.assert_array_equal(anc, nd1.astype(np.int64))
