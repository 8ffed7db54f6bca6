"""Edge cases of the SURVEY 8(f) entry points: empty and degenerate inputs, capacity limits -- each either equals the oracle
or fails with an argument error, never with a fault."""
import numpy as np
import pytest

from oracle_api import oracle_pose_optimize, oracle_stereo_matches, oracle_transform

pytestmark = pytest.mark.gpu


def test_fuse_and_projection_with_nothing_to_do(pkg, oracle, sm):
    g, dKF, scale, u_right, inv_s2, pts = sm.make_fuse_case(50, n=300, n_pts=200)
    m = pkg.Matcher(0.6, True)
    try:
        none = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in pts.items()}
        none["valid"][:] = 0
        bi, bd = m.FuseSearch(g, dKF, scale, u_right, inv_s2, none, 3.0, True)
        assert (bi == -1).all() and (bd == 256).all()
        far = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in pts.items()}
        far["u"][:] = -500.0                                  # every window lies outside the grid
        bi0, bd0 = oracle.fuse_search(g, dKF, scale, u_right, inv_s2, far, 3.0, True)
        bi, bd = m.FuseSearch(g, dKF, scale, u_right, inv_s2, far, 3.0, True)
        np.testing.assert_array_equal(bi, bi0); np.testing.assert_array_equal(bd, bd0)
        assert (bi == -1).all()
        # projection search against a frame without features
        gg, dF, angF, sc, last, assign, occ = sm.make_last_frame_case(51, n=16, n_last=40)
        empty = dict(gg); empty["x"] = gg["x"][:0].copy(); empty["y"] = gg["y"][:0].copy(); empty["octave"] = gg["octave"][:0].copy()
        a = np.zeros(1, np.int32); o = np.zeros(1, np.uint8)
        assert m.SearchByProjection_last(empty, dF[:0].copy(), angF[:0].copy(), sc, last, 15.0, a[:0], o[:0]) == 0
    finally:
        m.close()


def test_triangulation_and_initialization_degenerate(pkg, oracle, sm):
    k1, k2, ep, F12, sigma2, scale = sm.make_triangulation_case(52, n=200)
    m = pkg.Matcher(0.6, True)
    try:
        allmp = dict(k1); allmp["has_mp"] = np.ones_like(k1["has_mp"])              # every KF1 feature already holds a map point
        n1, m1 = m.SearchForTriangulation(allmp, k2, ep, F12, sigma2, scale)
        assert n1 == 0 and (m1 == -1).all()
        zeroF = np.zeros(9, np.float32)                                             # den == 0: the epipolar test rejects everything
        n0, m0 = oracle.search_for_triangulation(k1, k2, ep, zeroF, sigma2, scale, False, False, True)
        n1, m1 = m.SearchForTriangulation(k1, k2, ep, zeroF, sigma2, scale)
        assert n1 == n0 == 0 and (m1 == -1).all()
        f1, g2, d2, a2, sc = sm.make_initialization_case(53, n=300)
        hi = dict(f1); hi["octave"] = np.full_like(f1["octave"], 2)                 # no level-0 feature in F1
        n1, m1 = m.SearchForInitialization(hi, g2, d2, a2, sc, 100)
        assert n1 == 0 and (m1 == -1).all()
    finally:
        m.close()


def test_vocabulary_capacity(pkg, oracle, synth):
    voc = synth.make_vocabulary(60, k=5, L=2)
    rs = np.random.RandomState(60)
    desc = rs.randint(0, 256, size=(8192, 32)).astype(np.uint8)
    v = pkg.Vocabulary(voc)
    try:
        (bi1, bv1), (fn1, fo1, ff1) = v.transform(desc, 1)                          # the largest frame the assembly kernel sorts in LDS
        (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(oracle, voc, desc, 1)
        np.testing.assert_array_equal(bi1, bi0); np.testing.assert_array_equal(bv1, bv0); np.testing.assert_array_equal(ff1, ff0)
        with pytest.raises(pkg.OrbxError) as e:
            v.transform(np.concatenate([desc, desc[:1]]), 1)
        assert e.value.code == -3
        w, wt, nd = v.transform_features(np.concatenate([desc, desc]), 1)           # the per-feature call has no such limit
        assert len(w) == 16384
    finally:
        v.close()


def test_pose_all_outliers_and_stereo_without_right_features(pkg, oracle, synth):
    w = synth.make_pose_problem(61, n=60, outlier_frac=1.0)                         # nothing consistent: the flags still match the oracle
    s = pkg.PoseSolver()
    try:
        r1 = s.optimize(w)
    finally:
        s.close()
    r0 = oracle_pose_optimize(oracle, w)
    np.testing.assert_array_equal(r1["outlier"], r0["outlier"])
    assert r1["inliers"] == r0["inliers"]
    left, right = synth.make_stereo_pair(62)
    oL, oR = oracle.extractor(), oracle.extractor()
    _, kL0, dL0 = oL.extract(left, (0, 0)); _, kR0, dR0 = oR.extract(right, (0, 0))
    exL, exR = pkg.Extractor(), pkg.Extractor()
    try:
        _, kL, dL = exL(left, (0, 0)); _, kR, dR = exR(right, (0, 0))
        ur, dp = exL.stereo_matches(exR, kL, dL, kR[:0], dR[:0], 0.11, 47.9)
        assert (ur == -1).all() and (dp == -1).all()
        # a baseline so short that maxD = mbf / mb admits no candidate beyond a few pixels
        _, ur0, dp0 = oracle_stereo_matches(oL, oR, kL0, dL0, kR0, dR0, 10.0, 47.9)
        ur1, dp1 = exL.stereo_matches(exR, kL, dL, kR, dR, 10.0, 47.9)
        np.testing.assert_array_equal(ur1, ur0); np.testing.assert_array_equal(dp1, dp0)
    finally:
        exL.close(); exR.close()
