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


@pytest.mark.parametrize("w,h,scale,nlev,B", [(333, 250, 2.0, 3, 1), (397, 301, 2.0, 2, 5), (640, 481, 2.5, 3, 70), (333, 250, 2.0, 3, 66)])
def test_scale_factors_of_two_and_more(pkg, oracle, synth, w, h, scale, nlev, B):
    """found by tools/soak_extractor.py: with scaleFactor >= 2 a quad of destination columns can span more than the 8 source bytes the
    table-driven resize kernel holds (e.g. 333 -> 166 columns: offsets 0, 2, 4, 7); such levels take k_resize_generic (and keep out of
    the fused resize tail of large batches) -- same bits as the oracle, small and large batches"""
    imgs = np.stack([synth.make_frame(700 + k, w, h) for k in range(min(B, 3))])
    oex = oracle.extractor(500, scale, nlev, 20, 7)
    ref = [oex.extract(im, (0, 1000)) for im in imgs]
    ex = pkg.Extractor(500, scale, nlev, 20, 7)
    try:
        mono, n, kps, desc = ex.extract_batch(np.ascontiguousarray(imgs[np.arange(B) % len(imgs)]))
    finally:
        ex.close()
    for b in range(B):
        r0, k0, d0 = ref[b % len(imgs)]
        assert mono[b] == r0 and n[b] == len(k0) > 20
        for f in k0.dtype.names:
            np.testing.assert_array_equal(kps[b, :n[b]][f], k0[f], err_msg=f)
        np.testing.assert_array_equal(desc[b, :n[b]], d0)


@pytest.mark.parametrize("nfeat,nlev,size,B,expect_over", [(100, 10, (756, 481), 1, True), (100, 10, (756, 481), 65, True), (40, 8, (640, 480), 3, False),
                                                             (12, 4, (900, 200), 2, True)])
def test_tiny_per_level_budgets_return_what_the_first_pass_makes(pkg, oracle, synth, nfeat, nlev, size, B, expect_over):
    """found by tools/soak_extractor.py: DistributeOctTree's first pass divides every root without looking at N (the size test follows
    the pass, src/ORBextractor.cc:606-672), so a level whose budget is below 4 x nIni still returns up to 4 nodes per root -- more than
    N + 3 (e.g. 100 features over 10 levels: the last level's budget is 1, it returns 8 key points).  The per-level selection capacity
    and orbx_max_keypoints cover that (it used to trip the device-side guard: ORBX_ERR_INTERNAL)."""
    w, h = size
    imgs = np.stack([synth.make_frame(720 + k, w, h) for k in range(min(B, 3))])
    oex = oracle.extractor(nfeat, 1.2, nlev, 20, 7)
    ref = [oex.extract(im, (0, 0)) for im in imgs]
    ex = pkg.Extractor(nfeat, 1.2, nlev, 20, 7)
    try:
        per_level = ex.features_per_level()
        mono, n, kps, desc = ex.extract_batch(np.ascontiguousarray(imgs[np.arange(B) % len(imgs)]), (0, 0))
    finally:
        ex.close()
    over = False
    for b in range(B):
        r0, k0, d0 = ref[b % len(imgs)]
        assert mono[b] == r0 and n[b] == len(k0)
        for f in k0.dtype.names:
            np.testing.assert_array_equal(kps[b, :n[b]][f], k0[f], err_msg=f)
        np.testing.assert_array_equal(desc[b, :n[b]], d0)
        over = over or bool((np.bincount(k0["octave"], minlength=nlev) > per_level + 3).any())
    assert over == expect_over, "the case should%s exercise a level that returns more than N + 3 key points" % ("" if expect_over else " not")


def test_capacity_contract_on_very_wide_images(pkg, oracle, synth):
    """found by tools/soak_extractor.py (extreme): orbx_max_keypoints() covers images of up to 8.5 : 1; a 700 x 98 image has 10 octree roots,
    and with a budget of ONE feature the level still returns 4 key points per root.  The contract for that: ORBX_ERR_CAPACITY with *n = the
    count needed, and the same call with that capacity gives the oracle's key points (the shim's Extractor::extract does exactly this)."""
    import ctypes as C
    img = synth.make_frame(3000 + 10 * 134, 700, 98)
    r0, k0, d0 = oracle.extractor(1, 1.1, 1, 20, 20).extract(img, (0, 1000))
    ex = pkg.Extractor(1, 1.1, 1, 20, 20)
    try:
        cap = ex.max_keypoints
        assert len(k0) > cap
        n, mono = C.c_int(), C.c_int()
        kps = np.zeros(cap, pkg.KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        rc = pkg.lib.orbx_extract(ex._h, img.ctypes.data_as(C.c_void_p), 700, 98, 700, 0, 1000, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                  C.byref(n), C.byref(mono))
        assert rc == -2 and n.value == len(k0)          # ORBX_ERR_CAPACITY, the count needed
        cap = n.value
        kps = np.zeros(cap, pkg.KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        rc = pkg.lib.orbx_extract(ex._h, img.ctypes.data_as(C.c_void_p), 700, 98, 700, 0, 1000, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p), cap,
                                  C.byref(n), C.byref(mono))
        assert rc == 0 and mono.value == r0 and n.value == len(k0)
        for f in k0.dtype.names:
            np.testing.assert_array_equal(kps[f], k0[f], err_msg=f)
        np.testing.assert_array_equal(desc, d0)
    finally:
        ex.close()


def test_max_keypoints_for_covers_wide_images(pkg, oracle, synth):
    """orbx_max_keypoints_for(): 888 x 141, 100 features over 4 levels at 1.25 -- the bordered area of the upper levels is 9 : 1, each
    level returns up to 36 key points, 146 in all against orbx_max_keypoints() = 133 -- sized with it the batch call succeeds"""
    img = synth.make_frame(5, 888, 141)
    r0, k0, d0 = oracle.extractor(100, 1.25, 4, 12, 10).extract(img, (0, 1000))
    ex = pkg.Extractor(100, 1.25, 4, 12, 10)
    try:
        assert ex.max_keypoints_for(888, 141) >= len(k0) and ex.max_keypoints_for(640, 480) == ex.max_keypoints
        mono, n, kps, desc = ex.extract_batch(np.stack([img] * 3))
    finally:
        ex.close()
    for b in range(3):
        assert mono[b] == r0 and n[b] == len(k0)
        for f in k0.dtype.names:
            np.testing.assert_array_equal(kps[b, :n[b]][f], k0[f], err_msg=f)
        np.testing.assert_array_equal(desc[b, :n[b]], d0)
