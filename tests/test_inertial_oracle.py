"""LocalInertialBA restatement (oracle/inertial_oracle.cpp; reference src/Optimizer.cc:2383-2958, src/G2oTypes.cc) -- GROUNDWORK:
there is no HIP counterpart yet.  PARITY UNPINNED; what can be checked without the reference is checked here: the restated
analytic Jacobians of EdgeInertial against central differences through the restated update rule, that consistent data is a
fixed point, and that a perturbed window converges back to the ground truth."""
import os

import numpy as np

from oracle_api import oracle_inertial_jacobian_check, oracle_inertial_solve


def _rot_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return float(np.arccos(np.clip(c, -1, 1)))


def test_inertial_jacobians_against_central_differences(oracle, synth):
    pr, _ = synth.make_inertial_window(0, n_opt=4, n_points=20)
    for l in range(len(pr["links"])):
        # the pre-integration getters work on a float bias, so the bias columns carry float-rounding noise (~1e-3 at h = 1e-4)
        assert oracle_inertial_jacobian_check(oracle, pr, l, 1e-4) < 5e-3


def test_consistent_window_is_a_fixed_point(oracle, synth):
    pr, gt = synth.make_inertial_window(1, n_opt=5, n_points=120, noise_px=0.0, perturb=False)
    for L in pr["links"]:                                  # exact pre-integration (no measurement noise)
        i, j, dt = L["kf1"], L["kf2"], float(L["dT"])
        g = np.array([0, 0, -9.81])
        L["dR"] = (gt["Rwb"][i].T @ gt["Rwb"][j]).astype(np.float32)
        L["dV"] = (gt["Rwb"][i].T @ (gt["vel"][j] - gt["vel"][i] - g * dt)).astype(np.float32)
        L["dP"] = (gt["Rwb"][i].T @ (gt["twb"][j] - gt["twb"][i] - gt["vel"][i] * dt - 0.5 * g * dt * dt)).astype(np.float32)
    r = oracle_inertial_solve(oracle, pr)
    assert r["stats"]["chi2_initial"] < 1.0                # only float rounding of the inputs is left
    assert np.abs(r["twb"] - pr["twb"]).max() < 2e-4 and np.abs(r["points"] - pr["points"]).max() < 5e-3


def test_perturbed_window_converges_to_ground_truth(oracle, synth):
    pr, gt = synth.make_inertial_window(2, n_opt=6, n_points=200, obs_per_point=5, noise_px=0.3)
    r = oracle_inertial_solve(oracle, pr)
    st = r["stats"]
    assert st["iterations"] >= 3 and st["chi2_final"] < 0.05 * st["chi2_initial"]
    err0 = np.abs(pr["twb"] - gt["twb"]).max(); err1 = np.abs(r["twb"] - gt["twb"]).max()
    assert err1 < 0.25 * err0 and err1 < 0.01
    ang0 = max(_rot_angle(pr["Rwb"][i], gt["Rwb"][i]) for i in range(1, pr["n_kf"]))
    ang1 = max(_rot_angle(r["Rwb"][i], gt["Rwb"][i]) for i in range(1, pr["n_kf"]))
    assert ang1 < 0.3 * ang0
    assert np.abs(r["vel"] - gt["vel"]).max() < np.abs(pr["vel"] - gt["vel"]).max()
    # the fixed key frame did not move, rotations stayed orthonormal, all points are in front of their cameras
    assert np.array_equal(r["Rwb"][0], pr["Rwb"][0]) and np.array_equal(r["twb"][0], pr["twb"][0]) and np.array_equal(r["bg"][0], pr["bg"][0])
    for R in r["Rwb"]:
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    assert r["depth_positive"].all()


def test_large_window_settings_and_bias_recovery(oracle, synth):
    """bLarge: 4 iterations from lambda 1e-2; a common bias error is pulled back by the inertial terms"""
    pr, gt = synth.make_inertial_window(3, n_opt=8, n_points=150, bias_error=0.002)
    pr["lambda_init"] = 1e-2; pr["max_iters"] = 4
    r = oracle_inertial_solve(oracle, pr)
    assert r["stats"]["iterations"] <= 4 and r["stats"]["chi2_final"] < r["stats"]["chi2_initial"]


def _golden_window(synth):
    return synth.make_inertial_window(45, n_opt=5, n_points=120, obs_per_point=4, stereo_frac=0.3, n_covisible_fixed=2)[0]


def test_golden_inertial_window(oracle, synth):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "inertial_5kf_120mp.npz"))
    r = oracle_inertial_solve(oracle, _golden_window(synth))
    assert (r["stats"]["iterations"], r["stats"]["trials"]) == (int(g["iterations"]), int(g["trials"]))
    np.testing.assert_allclose(r["stats"]["chi2_final"], float(g["chi2_final"]), rtol=1e-9)
    for k in ("Rwb", "twb", "vel", "bg", "ba", "points"):
        np.testing.assert_allclose(r[k], g[k], rtol=0, atol=1e-9)
