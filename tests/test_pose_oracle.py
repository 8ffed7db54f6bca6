"""Optimizer::PoseOptimization restatement (oracle/lba_oracle.cpp, reference src/Optimizer.cc:814-1115) -- PARITY UNPINNED:
the reference holds no fixture for it and cannot be built here (g2o/Eigen/OpenCV absent), so the restatement is checked
against the ground truth of synthetic frames and against its own frozen outputs (tests/golden/pose_*.npz)."""
import os

import numpy as np

from oracle_api import oracle_pose_optimize

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _quat_to_R(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _noise_free(w):
    """regenerate the inlier observations from the true pose (the gross outliers stay)"""
    Xc = w["Xw"] @ w["true_R"].T + w["true_t"]
    u = w["fx"] * Xc[:, 0] / Xc[:, 2] + w["cx"]
    v = w["fy"] * Xc[:, 1] / Xc[:, 2] + w["cy"]
    good = ~w["is_outlier"]
    w["obs"][good, 0] = u[good]; w["obs"][good, 1] = v[good]
    st = w["stereo"].astype(bool) & good
    w["obs"][st, 2] = (u - w["bf"] / Xc[:, 2])[st]
    return w


def test_pose_converges_to_truth_and_flags_gross_outliers(oracle, synth):
    for seed, stereo_frac in ((0, 0.0), (1, 0.5)):
        w = _noise_free(synth.make_pose_problem(seed, n=250, outlier_frac=0.12, stereo_frac=stereo_frac))
        r = oracle_pose_optimize(oracle, w)
        np.testing.assert_array_equal(r["outlier"].astype(bool), w["is_outlier"])
        assert r["n_bad"] == int(w["is_outlier"].sum()) and r["inliers"] == 250 - r["n_bad"]
        assert np.abs(_quat_to_R(r["q"]) - w["true_R"]).max() < 1e-5      # Xw is rounded to float: not exactly zero
        assert np.abs(r["t"] - w["true_t"]).max() < 1e-4


def test_pose_small_problems(oracle, synth):
    w = synth.make_pose_problem(4, n=2, outlier_frac=0.0)
    r = oracle_pose_optimize(oracle, w)
    assert r["inliers"] == 0 and r["n_bad"] == 0                        # nInitialCorrespondences < 3 (:998)
    np.testing.assert_allclose(r["q"], w["q"] / np.linalg.norm(w["q"]), rtol=0, atol=1e-15)
    np.testing.assert_array_equal(r["t"], w["t"])
    w = synth.make_pose_problem(5, n=9, outlier_frac=0.0)               # < 10 edges: a single round (:1099)
    r = oracle_pose_optimize(oracle, w)
    assert 0 < r["inliers"] <= 9
    w = synth.make_pose_problem(6, n=0, outlier_frac=0.0)
    assert oracle_pose_optimize(oracle, w)["inliers"] == 0


def test_pose_noisy_frame_is_close(oracle, synth):
    w = synth.make_pose_problem(7, n=400, outlier_frac=0.1, stereo_frac=0.3)
    r = oracle_pose_optimize(oracle, w)
    Rerr = _quat_to_R(r["q"]) @ w["true_R"].T
    ang = np.degrees(np.arccos(np.clip((np.trace(Rerr) - 1) / 2, -1, 1)))
    assert ang < 0.2 and np.abs(r["t"] - w["true_t"]).max() < 0.05     # started 2 deg / 5 cm off
    assert (r["outlier"].astype(bool) & w["is_outlier"]).sum() == w["is_outlier"].sum()   # every gross outlier is flagged
    assert r["n_bad"] < 0.25 * 400


def test_pose_golden(oracle, synth):
    for name, kw in (("pose_mono_300", dict(seed=2, n=300, outlier_frac=0.1, stereo_frac=0.0)),
                     ("pose_stereo_200", dict(seed=3, n=200, outlier_frac=0.15, stereo_frac=0.5))):
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        r = oracle_pose_optimize(oracle, synth.make_pose_problem(**kw))
        np.testing.assert_array_equal(r["outlier"], g["outlier"])
        assert (r["n_bad"], r["inliers"]) == (int(g["n_bad"]), int(g["inliers"]))
        np.testing.assert_allclose(r["q"], g["q"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["t"], g["t"], rtol=0, atol=1e-12)
