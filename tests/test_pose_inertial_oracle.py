"""PoseInertialOptimizationLastKeyFrame restatement (oracle/inertial_oracle.cpp; reference src/Optimizer.cc:4491-4873) -- GROUNDWORK,
PARITY UNPINNED: ground-truth recovery with gross outliers, the outlier set, and the Hessian of the new prior."""
import numpy as np

from oracle_api import oracle_pose_inertial_optimize


def _angle(Ra, Rb):
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


def test_recovers_ground_truth_and_outliers(oracle, synth):
    for seed, kw in ((0, dict(n=300, outlier_frac=0.1)), (1, dict(n=120, outlier_frac=0.2, stereo_frac=0.5)), (2, dict(n=40, outlier_frac=0.0))):
        pr, gt = synth.make_pose_inertial_problem(seed, **kw)
        r = oracle_pose_inertial_optimize(oracle, pr)
        assert np.abs(r["twb"] - gt["twb"]).max() < 0.2 * np.abs(pr["twb"][1] - gt["twb"]).max() + 2e-3
        assert _angle(r["Rwb"], gt["Rwb"]) < 0.3 * _angle(pr["Rwb"][1], gt["Rwb"]) + 1e-3
        # every gross outlier (>= 15 px) is flagged; a few noisy inliers may be flagged as well
        assert r["outlier"][gt["is_outlier"]].all()
        assert r["outlier"][~gt["is_outlier"]].mean() < 0.1
        assert r["inliers"] == len(pr["Xw"]) - r["n_bad"] and r["n_bad"] == int(r["outlier"].sum())
        # the biases are pulled to the key frame's by the random-walk edges; the prior Hessian is symmetric positive definite
        assert np.abs(r["bg"] - pr["bg"][0]).max() < 1e-9
        H = r["H"]
        assert np.abs(H - H.T).max() < 1e-6 * np.abs(H).max() and np.linalg.eigvalsh((H + H.T) / 2).min() > 0


def test_small_frames(oracle, synth):
    pr, _ = synth.make_pose_inertial_problem(3, n=5, outlier_frac=0.0)
    r = oracle_pose_inertial_optimize(oracle, pr)       # 5 + 3 edges < 10: one round only; < 30 inliers: the recovery pass runs
    assert r["inliers"] + r["n_bad"] == 5
    pr, _ = synth.make_pose_inertial_problem(4, n=0)
    r = oracle_pose_inertial_optimize(oracle, pr)       # pure inertial prediction
    assert r["inliers"] == 0 and np.isfinite(r["twb"]).all()


def test_last_frame_variant(oracle, synth):
    """PoseInertialOptimizationLastFrame: the previous frame is free and tied to its prior; 30 x 30 Hessian before Marginalize"""
    for seed, kw in ((5, dict(n=300, outlier_frac=0.1)), (6, dict(n=80, outlier_frac=0.2, stereo_frac=0.5))):
        pr, gt = synth.make_pose_inertial_problem(seed, last_frame=True, **kw)
        r = oracle_pose_inertial_optimize(oracle, pr)
        assert np.abs(r["twb"] - gt["twb"]).max() < 0.3 * np.abs(pr["twb"][1] - gt["twb"]).max() + 3e-3
        assert r["outlier"][gt["is_outlier"]].all() and r["outlier"][~gt["is_outlier"]].mean() < 0.15
        H = r["H"]
        assert H.shape == (30, 30) and np.abs(H - H.T).max() < 1e-6 * np.abs(H).max() and np.linalg.eigvalsh((H + H.T) / 2).min() > 0
