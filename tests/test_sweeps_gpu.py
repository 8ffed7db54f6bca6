"""Seeded sweeps of the remaining entry points over sizes and parameters (the fixed cases of the other test files pin
behaviour; these look for configurations nobody thought of): every case must equal the oracle."""
import numpy as np
import pytest

from oracle_api import oracle_pose_optimize, oracle_transform

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_bow_sweep(pkg, oracle, synth, seed):
    rs = np.random.RandomState(900 + seed)
    n = int(rs.choice([1, 7, 64, 300, 1000, 1700, 3000]))
    ratio = float(rs.choice([0.6, 0.7, 0.75, 0.9])); ori = bool(rs.randint(0, 2))
    ms = synth.make_match_set(100 + seed, n=n, p_true=float(rs.uniform(0.2, 0.95)))
    n0, m0 = oracle.search_by_bow(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"], ratio, ori)
    k0, q0 = oracle.search_by_bow_kfkf(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], np.ones(n, np.uint8), ms["angF"], ms["fvF"], ratio, ori)
    m = pkg.Matcher(ratio, ori)
    try:
        n1, m1 = m.SearchByBoW(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"])
        k1, q1 = m.SearchByBoW_KFKF(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], np.ones(n, np.uint8), ms["angF"], ms["fvF"])
    finally:
        m.close()
    assert (n1, k1) == (n0, k0)
    np.testing.assert_array_equal(m1, m0); np.testing.assert_array_equal(q1, q0)


@pytest.mark.parametrize("seed", range(10))
def test_projection_sweep(pkg, oracle, sm, seed):
    rs = np.random.RandomState(1300 + seed)
    n = int(rs.choice([30, 200, 1000, 2500])); npts = int(rs.choice([1, 50, 900, 3000]))
    th = float(rs.choice([1.0, 3.0, 7.0, 15.0, 40.0])); ori = bool(rs.randint(0, 2))
    g, dF, angF, scale, last, assign, occ = sm.make_last_frame_case(60 + seed, n=n, n_last=npts)
    a0, o0 = assign.copy(), occ.copy()
    n0 = oracle.search_by_projection_last(g, dF, angF, scale, last, th, ori, a0, o0)
    g2, dF2, angF2, scale2, mp, assign2, occ2 = sm.make_projection_case(60 + seed, n=n, n_mp=npts)
    b0, p0 = assign2.copy(), occ2.copy()
    k0 = oracle.search_by_projection(g2, dF2, scale2, mp, th, 0.8, b0, p0, bool(seed & 1), 20.0)
    m = pkg.Matcher(0.8, ori)
    try:
        a1, o1 = assign.copy(), occ.copy()
        n1 = m.SearchByProjection_last(g, dF, angF, scale, last, th, a1, o1)
        b1, p1 = assign2.copy(), occ2.copy()
        k1 = m.SearchByProjection(g2, dF2, scale2, mp, th, b1, p1, far_points=bool(seed & 1), th_far=20.0)
    finally:
        m.close()
    assert (n1, k1) == (n0, k0)
    np.testing.assert_array_equal(a1, a0); np.testing.assert_array_equal(o1, o0)
    np.testing.assert_array_equal(b1, b0); np.testing.assert_array_equal(p1, p0)


@pytest.mark.parametrize("seed", range(10))
def test_lba_sweep(pkg, oracle, synth, seed):
    rs = np.random.RandomState(1700 + seed)
    kw = dict(n_opt=int(rs.randint(1, 40)), n_fixed=int(rs.randint(1, 8)), n_points=int(rs.randint(20, 900)),
              obs_per_point=int(rs.randint(2, 9)), stereo_frac=float(rs.choice([0.0, 0.0, 0.3, 1.0])), outlier_frac=float(rs.choice([0.0, 0.03, 0.1])))
    w = synth.make_ba_window(300 + seed, **kw)
    iters = int(rs.choice([1, 5, 10]))
    r0 = oracle.lba_solve(w, iters)
    s = pkg.LbaSolver()
    try:
        r1 = s.solve(w, iters)
    finally:
        s.close()
    s0, s1 = r0["stats"], r1["stats"]
    assert (s1["iterations"], s1["trials"], s1["stop_reason"]) == (s0["iterations"], s0["trials"], s0["stop_reason"]), kw
    # weakly constrained windows (3 observations per point) amplify the rounding of the reduced solve: 5.4e-9 relative is the
    # largest difference measured (seed 4, gpurun_out/lba_fused.log, round 1).  FROZEN at 1e-7: three decades inside the 1e-4
    # contract of BASELINE.json and twenty times the observed worst case -- a change that needs more is a regression.
    # (found by tools/soak_sweeps.py, seed 1034: an exactly solvable window ends at chi2 ~ 1e-23, pure rounding noise on both sides --
    # the absolute floor is 1e-18 of where the window started)
    np.testing.assert_allclose(s1["chi2_final"], s0["chi2_final"], rtol=1e-7, atol=1e-18 * max(s0["chi2_initial"], 1.0))
    d0, d1 = r0["points"] - w["points"], r1["points"] - w["points"]
    assert np.abs(d0 - d1).max() <= 1e-4 * max(np.abs(d0).max(), 1e-12), kw
    np.testing.assert_array_equal(r1["depth_positive"], r0["depth_positive"])


@pytest.mark.parametrize("seed", range(12))
def test_pose_and_vocab_sweep(pkg, oracle, synth, seed):
    rs = np.random.RandomState(2100 + seed)
    w = synth.make_pose_problem(500 + seed, n=int(rs.choice([3, 10, 11, 100, 256, 257, 512, 513, 1500])), outlier_frac=float(rs.uniform(0, 0.4)),
                                stereo_frac=float(rs.choice([0.0, 0.5, 1.0])))
    voc = synth.make_vocabulary(500 + seed, k=int(rs.randint(2, 17)), L=int(rs.randint(1, 5)), shuffle_ids=bool(rs.randint(0, 2)),
                                stop_frac=float(rs.choice([0.0, 0.1])))
    levelsup = int(rs.randint(0, 6))
    desc = np.ascontiguousarray(voc["desc"][rs.randint(1, voc["n_nodes"], 400)] ^ (rs.uniform(size=(400, 32)) < 0.05).astype(np.uint8))
    ps, vv = pkg.PoseSolver(), pkg.Vocabulary(voc)
    try:
        r1 = ps.optimize(w)
        (bi1, bv1), (fn1, fo1, ff1) = vv.transform(desc, levelsup)
    finally:
        ps.close(); vv.close()
    r0 = oracle_pose_optimize(oracle, w)
    np.testing.assert_array_equal(r1["outlier"], r0["outlier"])
    q0 = np.asarray(w["q"]) / np.linalg.norm(w["q"])
    assert np.abs(r1["t"] - r0["t"]).max() <= 1e-4 * np.abs(r0["t"] - w["t"]).max() + 1e-12
    assert np.abs(r1["q"] - r0["q"]).max() <= 1e-4 * np.abs(r0["q"] - q0).max() + 1e-12
    (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(oracle, voc, desc, levelsup)
    np.testing.assert_array_equal(bi1, bi0); np.testing.assert_array_equal(bv1, bv0)
    np.testing.assert_array_equal(fn1, fn0); np.testing.assert_array_equal(fo1, fo0); np.testing.assert_array_equal(ff1, ff0)


@pytest.mark.parametrize("seed", range(8))
def test_matcher_family_sweep(pkg, oracle, sm, seed):
    rs = np.random.RandomState(2500 + seed)
    n = int(rs.choice([40, 300, 1000, 2000])); npts = int(rs.choice([1, 100, 1000, 5000]))
    th = float(rs.choice([2.0, 3.0, 4.0, 10.0])); ori = bool(rs.randint(0, 2))
    m = pkg.Matcher(float(rs.choice([0.6, 0.8, 0.9])), ori)
    try:
        g, dKF, scale, u_right, inv_s2, pts = sm.make_fuse_case(70 + seed, n=n, n_pts=npts, stereo_frac=float(rs.uniform(0, 1)))
        chi2 = bool(rs.randint(0, 2))
        bi0, bd0 = oracle.fuse_search(g, dKF, scale, u_right, inv_s2, pts, th, chi2)
        bi1, bd1 = m.FuseSearch(g, dKF, scale, u_right, inv_s2, pts, th, chi2)
        np.testing.assert_array_equal(bi1, bi0); np.testing.assert_array_equal(bd1, bd0)
        g, dF, angF, sc, p, assign, occ = sm.make_kf_projection_case(70 + seed, n=n, n_pts=min(npts, 2000))
        a0, o0 = assign.copy(), occ.copy(); a1, o1 = assign.copy(), occ.copy()
        dist = int(rs.choice([50, 64, 100]))
        assert m.SearchByProjection_kf(g, dF, angF, sc, p, th, dist, a1, o1) == oracle.search_by_projection_kf(g, dF, angF, sc, p, th, dist, ori, a0, o0)
        np.testing.assert_array_equal(a1, a0); np.testing.assert_array_equal(o1, o0)
        a0, o0 = assign.copy(), occ.copy(); a1, o1 = assign.copy(), occ.copy()
        ith = int(rs.choice([3, 8, 30])); rh = float(rs.choice([1.0, 1.5]))
        assert m.SearchByProjection_sim3(g, dF, sc, p, ith, rh, a1, o1) == oracle.search_by_projection_sim3(g, dF, sc, p, ith, rh, a0, o0)
        np.testing.assert_array_equal(a1, a0); np.testing.assert_array_equal(o1, o0)
        k1, k2, ep, F12, sigma2, sc2 = sm.make_triangulation_case(70 + seed, n=n)
        only_st, coarse = bool(rs.randint(0, 2)), bool(rs.randint(0, 2))
        n0, m0 = oracle.search_for_triangulation(k1, k2, ep, F12, sigma2, sc2, only_st, coarse, ori)
        n1, m1 = m.SearchForTriangulation(k1, k2, ep, F12, sigma2, sc2, only_st, coarse)
        assert n1 == n0
        np.testing.assert_array_equal(m1, m0)
        f1, g2, d2, a2, sc3 = sm.make_initialization_case(70 + seed, n=n)
        win = int(rs.choice([10, 50, 100]))
        n0, m0 = oracle.search_for_initialization(f1, g2, d2, a2, win, m.nnratio, ori)
        n1, m1 = m.SearchForInitialization(f1, g2, d2, a2, sc3, win)
        assert n1 == n0
        np.testing.assert_array_equal(m1, m0)
    finally:
        m.close()


@pytest.mark.parametrize("seed", range(6))
def test_stereo_sweep(pkg, oracle, synth, seed):
    from oracle_api import oracle_stereo_matches
    rs = np.random.RandomState(2900 + seed)
    w, h = int(rs.choice([320, 640, 752])), int(rs.choice([240, 480]))
    nfeat = int(rs.choice([300, 1000, 2000])); nlev = int(rs.choice([4, 8])); scale = float(rs.choice([1.2, 1.3]))
    left, right = synth.make_stereo_pair(80 + seed, w, h, band=int(rs.choice([30, 60, 120])), dmin=1, dmax=int(rs.choice([10, 40, 80])))
    mb = float(rs.choice([0.05, 0.11, 0.5])); mbf = mb * float(rs.choice([300.0, 435.0]))
    oL, oR = oracle.extractor(nfeat, scale, nlev, 20, 7), oracle.extractor(nfeat, scale, nlev, 20, 7)
    _, kL0, dL0 = oL.extract(left, (0, 0)); _, kR0, dR0 = oR.extract(right, (0, 0))
    _, ur0, dp0 = oracle_stereo_matches(oL, oR, kL0, dL0, kR0, dR0, mb, mbf)
    exL, exR = pkg.Extractor(nfeat, scale, nlev, 20, 7), pkg.Extractor(nfeat, scale, nlev, 20, 7)
    try:
        _, kL, dL = exL(left, (0, 0)); _, kR, dR = exR(right, (0, 0))
        ur1, dp1 = exL.stereo_matches(exR, kL, dL, kR, dR, mb, mbf)
    finally:
        exL.close(); exR.close()
    np.testing.assert_array_equal(ur1, ur0); np.testing.assert_array_equal(dp1, dp0)
