"""LocalInertialBA on the device (liba_solve; reference src/Optimizer.cc:2383-2958) against the oracle: same Levenberg control
flow (iterations, trials, stop reason), states within 1e-4 relative of the update, identical depth signs.  PARITY UNPINNED."""
import os

import numpy as np
import pytest

from oracle_api import oracle_inertial_solve

pytestmark = pytest.mark.gpu


def _compare(r0, r1, pr, tag):
    s0, s1 = r0["stats"], r1["stats"]
    assert (s1["iterations"], s1["trials"], s1["stop_reason"]) == (s0["iterations"], s0["trials"], s0["stop_reason"]), tag
    np.testing.assert_allclose(s1["chi2_initial"], s0["chi2_initial"], rtol=1e-9)
    np.testing.assert_allclose(s1["chi2_final"], s0["chi2_final"], rtol=1e-6)
    for key in ("twb", "vel", "bg", "ba", "points"):
        d0, d1 = r0[key] - np.asarray(pr[key]), r1[key] - np.asarray(pr[key])
        if d0.size:
            assert np.abs(d0 - d1).max() <= 1e-4 * max(np.abs(d0).max(), 1e-9), (tag, key)
    assert np.abs(r0["Rwb"] - r1["Rwb"]).max() < 1e-7, tag
    np.testing.assert_array_equal(r1["depth_positive"], r0["depth_positive"])
    np.testing.assert_allclose(r1["chi2"], r0["chi2"], rtol=1e-5, atol=1e-9)


def test_inertial_ba_matches_oracle(pkg, oracle, synth):
    s = pkg.InertialSolver()
    try:
        for seed, kw in ((0, dict(n_opt=6, n_points=200, obs_per_point=5)), (1, dict(n_opt=3, n_points=40)), (2, dict(n_opt=10, n_points=400, obs_per_point=4)),
                         (3, dict(n_opt=8, n_points=150, bias_error=0.002)), (4, dict(n_opt=25, n_points=600, obs_per_point=6))):
            pr, _ = synth.make_inertial_window(seed, **kw)
            if seed == 3:
                pr["lambda_init"] = 1e-2; pr["max_iters"] = 4          # bLarge
            _compare(oracle_inertial_solve(oracle, pr), s.solve(pr), pr, seed)
    finally:
        s.close()


@pytest.mark.parametrize("seed", range(8))
def test_inertial_ba_sweep(pkg, oracle, synth, seed):
    """seeded sweep over window size, observation count, stereo share, covisible fixed key frames, bLarge settings"""
    rs = np.random.RandomState(900 + seed)
    kw = dict(n_opt=int(rs.randint(2, 26)), n_points=int(rs.randint(30, 700)), obs_per_point=int(rs.randint(3, 8)),
              stereo_frac=float(rs.choice([0.0, 0.0, 0.4, 1.0])), n_covisible_fixed=int(rs.choice([0, 0, 3, 10])), bias_error=float(rs.choice([0.0, 0.001])))
    pr, _ = synth.make_inertial_window(50 + seed, **kw)
    if seed & 1:
        pr["lambda_init"] = 1e-2; pr["max_iters"] = 4
    s = pkg.InertialSolver()
    try:
        _compare(oracle_inertial_solve(oracle, pr), s.solve(pr), pr, kw)
    finally:
        s.close()


def test_inertial_ba_converges_and_keeps_the_fixed_key_frame(pkg, synth):
    pr, gt = synth.make_inertial_window(7, n_opt=6, n_points=200, obs_per_point=5, noise_px=0.3)
    s = pkg.InertialSolver()
    r = s.solve(pr)
    s.close()
    assert r["stats"]["chi2_final"] < 0.05 * r["stats"]["chi2_initial"]
    assert np.abs(r["twb"] - gt["twb"]).max() < 0.01
    assert np.array_equal(r["Rwb"][0], pr["Rwb"][0]) and np.array_equal(r["twb"][0], pr["twb"][0])
    for R in r["Rwb"]:
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12


def test_inertial_ba_bad_arguments(pkg, synth):
    pr, _ = synth.make_inertial_window(8, n_opt=3, n_points=30)
    s = pkg.InertialSolver()
    bad = dict(pr); bad["edge_kf"] = pr["edge_kf"].copy(); bad["edge_kf"][0] = 99
    with pytest.raises(pkg.OrbxError):
        s.solve(bad)
    bad = dict(pr); bad["lambda_init"] = 0.0
    with pytest.raises(pkg.OrbxError):
        s.solve(bad)
    s.close()


def test_inertial_ba_degenerate_windows(pkg, oracle, synth):
    """pure inertial chain (no visual edges), pure visual window (no links), a single optimised key frame"""
    s = pkg.InertialSolver()
    try:
        pr, _ = synth.make_inertial_window(20, n_opt=4, n_points=30)
        for k in ("edge_kf", "edge_point", "edge_inv_sigma2", "edge_stereo"):
            pr[k] = pr[k][:0]
        pr["edge_obs"] = pr["edge_obs"][:0]; pr["points"] = pr["points"][:0]
        pr["max_iters"] = 2                                    # an exactly solvable chain reaches chi2 ~ 1e-28 later: accept / reject would be rounding noise
        _compare(oracle_inertial_solve(oracle, pr), s.solve(pr), pr, "inertial only")
        pr, _ = synth.make_inertial_window(21, n_opt=4, n_points=120, obs_per_point=5)
        pr["links"] = []
        _compare(oracle_inertial_solve(oracle, pr), s.solve(pr), pr, "visual only")
        pr, _ = synth.make_inertial_window(22, n_opt=1, n_points=60, obs_per_point=2)
        _compare(oracle_inertial_solve(oracle, pr), s.solve(pr), pr, "one key frame")
    finally:
        s.close()


def test_golden_inertial_window_on_device(pkg, synth):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "inertial_5kf_120mp.npz"))
    pr = synth.make_inertial_window(45, n_opt=5, n_points=120, obs_per_point=4, stereo_frac=0.3, n_covisible_fixed=2)[0]
    s = pkg.InertialSolver()
    r = s.solve(pr)
    s.close()
    assert (r["stats"]["iterations"], r["stats"]["trials"]) == (int(g["iterations"]), int(g["trials"]))
    np.testing.assert_allclose(r["stats"]["chi2_final"], float(g["chi2_final"]), rtol=1e-6)
    for k in ("twb", "vel", "points"):
        d0, d1 = g[k] - np.asarray(pr[k]), r[k] - np.asarray(pr[k])
        assert np.abs(d0 - d1).max() <= 1e-4 * np.abs(d0).max(), k


def test_inertial_batch_equals_single(pkg, synth):
    """liba_solve_batch: W windows per launch give, window by window, the bits of liba_solve (same kernel bodies, grid.y = window)"""
    wins = []
    for seed, kw in ((0, dict(n_opt=6, n_points=200, obs_per_point=5)), (1, dict(n_opt=3, n_points=40)), (2, dict(n_opt=10, n_points=400, obs_per_point=4)),
                     (3, dict(n_opt=8, n_points=150, bias_error=0.002)), (4, dict(n_opt=25, n_points=600, obs_per_point=6)),
                     (5, dict(n_opt=4, n_points=90, stereo_frac=0.5, n_covisible_fixed=3))):
        pr, _ = synth.make_inertial_window(seed, **kw)
        if seed == 3:
            pr["lambda_init"] = 1e-2; pr["max_iters"] = 4
        wins.append(pr)
    s, b = pkg.InertialSolver(), pkg.LibaBatch()
    try:
        single = [s.solve(w) for w in wins]
        for _ in range(2):                  # the second call runs on the grown arenas
            batch = b.solve(wins)
            for i, (r0, r1) in enumerate(zip(single, batch)):
                assert r1["stats"] == r0["stats"], i
                for k in ("Rwb", "twb", "vel", "bg", "ba", "points", "chi2", "depth_positive"):
                    np.testing.assert_array_equal(r1[k], r0[k], err_msg="window %d %s" % (i, k))
        assert b.last_device_ms() > 0
    finally:
        s.close(); b.close()


def test_inertial_batch_32_windows_against_oracle(pkg, oracle, synth):
    """32 windows of different sizes in one call; every fourth is checked against the oracle"""
    rs = np.random.RandomState(77)
    wins = []
    for i in range(32):
        pr, _ = synth.make_inertial_window(300 + i, n_opt=int(rs.randint(2, 12)), n_points=int(rs.randint(30, 300)), obs_per_point=int(rs.randint(3, 7)),
                                           stereo_frac=float(rs.choice([0.0, 0.3])), n_covisible_fixed=int(rs.choice([0, 4])))
        wins.append(pr)
    b = pkg.LibaBatch()
    try:
        res = b.solve(wins)
    finally:
        b.close()
    for i in range(0, 32, 4):
        _compare(oracle_inertial_solve(oracle, wins[i]), res[i], wins[i], i)


def test_inertial_batch_edge_cases(pkg, synth):
    b = pkg.LibaBatch()
    try:
        assert b.solve([]) == []
        pr, _ = synth.make_inertial_window(1, n_opt=3, n_points=40)
        with pytest.raises(pkg.OrbxError):
            b.solve([pr] * 65)
        bad = dict(pr); bad["lambda_init"] = 0.0
        with pytest.raises(pkg.OrbxError):
            b.solve([pr, bad])
        r = b.solve([pr])               # the handle stays usable after a refused call
        assert r[0]["stats"]["iterations"] > 0
    finally:
        b.close()
