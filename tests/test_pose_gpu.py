"""HIP PoseOptimization (pose_solver.hip, one workgroup per frame) against the CPU restatement of Optimizer::PoseOptimization
(reference src/Optimizer.cc:814-1115) through the C ABI.  Tolerance: BASELINE's 1e-4 relative on the pose UPDATE; the
inlier / outlier classification (float chi2 against 5.991 / 7.815) must be identical."""
import os

import numpy as np
import pytest

from oracle_api import oracle_pose_optimize

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(w, r, g):
    np.testing.assert_array_equal(r["outlier"], g["outlier"])
    assert (r["n_bad"], r["inliers"]) == (int(g["n_bad"]), int(g["inliers"]))
    # (Levenberg trial counts are NOT compared: at convergence the sign of rho is rounding noise on both sides)
    q0 = np.asarray(w["q"]) / np.linalg.norm(w["q"])
    dq = np.abs(np.asarray(g["q"]) - q0).max(); dt = np.abs(np.asarray(g["t"]) - w["t"]).max()
    assert np.abs(r["q"] - g["q"]).max() <= 1e-4 * dq + 1e-12
    assert np.abs(r["t"] - g["t"]).max() <= 1e-4 * dt + 1e-12


@pytest.mark.parametrize("seed,n,of,sf", [(0, 300, 0.1, 0.0), (1, 300, 0.1, 0.4), (2, 1000, 0.2, 1.0), (3, 50, 0.0, 0.0),
                                           (4, 9, 0.0, 0.0), (5, 2, 0.0, 0.0), (6, 0, 0.0, 0.0), (7, 3000, 0.3, 0.5),
                                           (8, 257, 0.05, 0.2), (9, 12, 0.3, 0.0)])
def test_pose_equals_oracle(pkg, oracle, synth, seed, n, of, sf):
    w = synth.make_pose_problem(seed, n=n, outlier_frac=of, stereo_frac=sf)
    s = pkg.PoseSolver()
    try:
        r = s.optimize(w)
    finally:
        s.close()
    _check(w, r, oracle_pose_optimize(oracle, w))


def test_pose_batch_equals_single(pkg, oracle, synth):
    """a batch is one launch with one workgroup per frame: results equal the per-frame oracle runs, in order"""
    ws = [synth.make_pose_problem(20 + i, n=100 + 37 * i, outlier_frac=0.1, stereo_frac=0.25 * (i % 3)) for i in range(24)]
    s = pkg.PoseSolver()
    try:
        rs = s.optimize_batch(ws)
        rs2 = s.optimize_batch(ws[::-1])[::-1]          # reuse of the solver's device blob
    finally:
        s.close()
    for w, r, r2 in zip(ws, rs, rs2):
        _check(w, r, oracle_pose_optimize(oracle, w))
        np.testing.assert_array_equal(r["q"], r2["q"]); np.testing.assert_array_equal(r["outlier"], r2["outlier"])


def test_pose_golden(pkg, synth):
    s = pkg.PoseSolver()
    try:
        for name, kw in (("pose_mono_300", dict(seed=2, n=300, outlier_frac=0.1, stereo_frac=0.0)),
                         ("pose_stereo_200", dict(seed=3, n=200, outlier_frac=0.15, stereo_frac=0.5))):
            w = synth.make_pose_problem(**kw)
            _check(w, s.optimize(w), np.load(os.path.join(GOLDEN, name + ".npz")))
    finally:
        s.close()


def test_pose_is_invariant_to_edge_order(pkg, synth):
    """a property that needs no oracle: permuting the edges permutes the outlier flags and leaves the pose alone (the block
    reductions are ordered, but the sum is the same set of terms)"""
    w = synth.make_pose_problem(40, n=700, outlier_frac=0.15, stereo_frac=0.3)
    rs = np.random.RandomState(1)
    p = rs.permutation(700)
    w2 = dict(w)
    for k in ("Xw", "obs", "inv_sigma2", "stereo"):
        w2[k] = np.ascontiguousarray(w[k][p])
    s = pkg.PoseSolver()
    try:
        r1, r2 = s.optimize_batch([w, w2])
    finally:
        s.close()
    np.testing.assert_array_equal(r1["outlier"][p], r2["outlier"])
    assert r1["inliers"] == r2["inliers"]
    np.testing.assert_allclose(r2["q"], r1["q"], rtol=0, atol=1e-9); np.testing.assert_allclose(r2["t"], r1["t"], rtol=0, atol=1e-9)
    Rerr = np.abs(r1["t"] - w["true_t"]).max()
    assert Rerr < 0.05
