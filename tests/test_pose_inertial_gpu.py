"""PoseInertialOptimizationLastKeyFrame on the device (liba_pose_optimize_batch; reference src/Optimizer.cc:4491-4873) against the
oracle: identical outlier flags / counters, states and the prior Hessian within 1e-4 relative (observed far tighter).  PARITY UNPINNED."""
import numpy as np
import pytest

from oracle_api import oracle_pose_inertial_optimize

pytestmark = pytest.mark.gpu


def _check(r0, r1, pr, tag):
    assert np.array_equal(r1["outlier"], r0["outlier"]), tag
    assert (r1["n_bad"], r1["inliers"]) == (r0["n_bad"], r0["inliers"]), tag
    for k, ini in (("twb", pr["twb"][1]), ("vel", pr["vel"][1])):
        d0, d1 = r0[k] - ini, r1[k] - ini
        assert np.abs(d0 - d1).max() <= 1e-4 * max(np.abs(d0).max(), 1e-9), (tag, k)
    assert np.abs(r0["Rwb"] - r1["Rwb"]).max() < 1e-7 and np.abs(r0["bg"] - r1["bg"]).max() < 1e-9 and np.abs(r0["ba"] - r1["ba"]).max() < 1e-9, tag
    assert np.abs(r0["H"] - r1["H"]).max() <= 1e-6 * np.abs(r0["H"]).max(), tag


def test_pose_inertial_batch_matches_oracle(pkg, oracle, synth):
    cases = [dict(n=300, outlier_frac=0.1), dict(n=120, outlier_frac=0.2, stereo_frac=0.5), dict(n=40, outlier_frac=0.0), dict(n=5, outlier_frac=0.0),
             dict(n=0), dict(n=700, outlier_frac=0.15, stereo_frac=1.0), dict(n=25, outlier_frac=0.3), dict(n=1000, outlier_frac=0.05)]
    probs = [synth.make_pose_inertial_problem(30 + i, **kw)[0] for i, kw in enumerate(cases)]
    probs[3]["rec_init"] = 1
    s = pkg.InertialSolver()
    try:
        res = s.pose_optimize_batch(probs)
        for i, (pr, r1) in enumerate(zip(probs, res)):
            _check(oracle_pose_inertial_optimize(oracle, pr), r1, pr, cases[i])
        # one frame alone gives the same result as inside the batch
        r_single = s.pose_optimize_batch([probs[0]])[0]
        assert np.array_equal(r_single["outlier"], res[0]["outlier"]) and np.array_equal(r_single["twb"], res[0]["twb"])
    finally:
        s.close()


@pytest.mark.parametrize("seed", range(6))
def test_pose_inertial_sweep(pkg, oracle, synth, seed):
    rs = np.random.RandomState(300 + seed)
    probs = [synth.make_pose_inertial_problem(1000 + 10 * seed + j, n=int(rs.randint(0, 600)), outlier_frac=float(rs.choice([0.0, 0.1, 0.3])),
                                              stereo_frac=float(rs.choice([0.0, 0.5, 1.0])), noise_px=float(rs.choice([0.3, 1.0])))[0] for j in range(6)]
    s = pkg.InertialSolver()
    try:
        for pr, r1 in zip(probs, s.pose_optimize_batch(probs)):
            _check(oracle_pose_inertial_optimize(oracle, pr), r1, pr, seed)
    finally:
        s.close()


def test_last_frame_variant_matches_oracle(pkg, oracle, synth):
    """PoseInertialOptimizationLastFrame: free previous frame + EdgePriorPoseImu, 30 x 30 Hessian"""
    cases = [dict(n=300, outlier_frac=0.1), dict(n=80, outlier_frac=0.2, stereo_frac=0.5), dict(n=4, outlier_frac=0.0), dict(n=0), dict(n=600, outlier_frac=0.05, stereo_frac=1.0),
             dict(n=150, outlier_frac=0.3)]
    probs = [synth.make_pose_inertial_problem(60 + i, last_frame=True, **kw)[0] for i, kw in enumerate(cases)]
    s = pkg.InertialSolver()
    try:
        res = s.pose_optimize_batch(probs)
        for i, (pr, r1) in enumerate(zip(probs, res)):
            r0 = oracle_pose_inertial_optimize(oracle, pr)
            assert r1["H"].shape == (30, 30)
            _check(r0, r1, pr, ("last frame", cases[i]))
        with pytest.raises(pkg.OrbxError):                   # one variant per batch
            s.pose_optimize_batch([probs[0], synth.make_pose_inertial_problem(1, n=50)[0]])
    finally:
        s.close()
