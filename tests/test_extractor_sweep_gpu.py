"""Seeded sweep over extractor configurations (image size, feature budget, pyramid depth, scale factor, FAST thresholds,
lapping area): every one must reproduce the oracle bit for bit -- key points, order, angles, responses, descriptors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases():
    rs = np.random.RandomState(20260)
    out = []
    for i in range(36):
        w = int(rs.randint(96, 900)); h = int(rs.randint(80, 620))
        nfeat = int(rs.choice([100, 300, 700, 1000, 1500, 2500, 4000]))
        scale = float(rs.choice([1.1, 1.2, 1.2, 1.25, 1.33, 1.5, 2.0]))
        nlev = int(rs.randint(1, 11))
        # keep the top level at least a few cells large (the reference itself breaks below EDGE_THRESHOLD-sized levels)
        while nlev > 1 and min(w, h) / scale ** (nlev - 1) < 60:
            nlev -= 1
        ini = int(rs.choice([20, 20, 12, 30])); mn = int(rs.choice([7, 7, 5, 10]))
        mn = min(mn, ini)
        lap = (0, 1000) if rs.uniform() < 0.5 else ((0, 0) if rs.uniform() < 0.5 else (int(w * 0.3), int(w * 0.6)))
        out.append((i, w, h, nfeat, scale, nlev, ini, mn, lap))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "%d_%dx%d_n%d_s%.2f_l%d" % (c[0], c[1], c[2], c[3], c[4], c[5]))
def test_extractor_sweep(pkg, oracle, synth, case):
    i, w, h, nfeat, scale, nlev, ini, mn, lap = case
    img = synth.make_frame(500 + i, w, h)
    r0, k0, d0 = oracle.extractor(nfeat, scale, nlev, ini, mn).extract(img, lap)
    ex = pkg.Extractor(nfeat, scale, nlev, ini, mn)
    try:
        r1, k1, d1 = ex(img, lap)
    finally:
        ex.close()
    assert r1 == r0 and len(k1) == len(k0)
    for f in k0.dtype.names:
        np.testing.assert_array_equal(k1[f], k0[f], err_msg=f)
    np.testing.assert_array_equal(d1, d0)


@pytest.mark.parametrize("args,size", [((12000, 1.2, 8, 20, 7), (640, 480)), ((4000, 1.2, 1, 20, 7), (320, 240)), ((9000, 1.5, 3, 12, 5), (752, 480))])
def test_feature_budgets_beyond_the_lds_node_pool(pkg, oracle, synth, args, size):
    """more than about 2500 features on ONE pyramid level do not fit the octree's LDS node pool: the pool then lives in an HBM
    scratch (the reference has no such limit; e.g. nfeatures = 12000 puts 2606 on level 0) -- same key points, bit for bit"""
    img = synth.make_frame(77, *size)
    r0, k0, d0 = oracle.extractor(*args).extract(img, (0, 1000))
    ex = pkg.Extractor(*args)
    try:
        assert ex.features_per_level().max() > 2500
        r1, k1, d1 = ex(img, (0, 1000))
    finally:
        ex.close()
    assert r1 == r0 and len(k1) == len(k0) > 500
    for f in k0.dtype.names:
        np.testing.assert_array_equal(k1[f], k0[f], err_msg=f)
    np.testing.assert_array_equal(d1, d0)
