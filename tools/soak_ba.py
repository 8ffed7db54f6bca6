#!/usr/bin/env python3
"""Soak: many seeded LocalBA / LocalInertialBA windows, device vs oracle -- how often does the Levenberg control flow
(iterations, trials, stop reason) differ, and how large is the worst state difference?  (GPU box; not part of the test suite.)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from oracle_api import Oracle, oracle_inertial_solve  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
o = Oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0; worst = 0.0
isol = pkg.InertialSolver()
for seed in range(N):
    rs = np.random.RandomState(5000 + seed)
    kw = dict(n_opt=int(rs.randint(2, 26)), n_points=int(rs.randint(30, 600)), obs_per_point=int(rs.randint(3, 8)),
              stereo_frac=float(rs.choice([0.0, 0.4, 1.0])), n_covisible_fixed=int(rs.choice([0, 3, 10])), bias_error=float(rs.choice([0.0, 0.001])))
    pr, _ = synth.make_inertial_window(7000 + seed, **kw)
    if seed & 1:
        pr["lambda_init"] = 1e-2; pr["max_iters"] = 4
    r0, r1 = oracle_inertial_solve(o, pr), isol.solve(pr)
    same = (r0["stats"]["iterations"], r0["stats"]["trials"], r0["stats"]["stop_reason"]) == (r1["stats"]["iterations"], r1["stats"]["trials"], r1["stats"]["stop_reason"])
    d0, d1 = r0["twb"] - pr["twb"], r1["twb"] - pr["twb"]
    rel = np.abs(d0 - d1).max() / max(np.abs(d0).max(), 1e-12)
    worst = max(worst, rel if same else 0.0)
    if not same:
        bad += 1
        print("inertial seed %d: control flow differs" % seed, kw, r0["stats"], r1["stats"])
isol.close()
print("inertial: %d windows, %d with a different control flow, worst relative update difference %.2e" % (N, bad, worst))
bad = 0; worst = 0.0
s = pkg.LbaSolver()
for seed in range(N):
    rs = np.random.RandomState(6000 + seed)
    kw = dict(n_opt=int(rs.randint(1, 60)), n_fixed=int(rs.randint(1, 8)), n_points=int(rs.randint(20, 1500)), obs_per_point=int(rs.randint(2, 9)),
              stereo_frac=float(rs.choice([0.0, 0.3, 1.0])), outlier_frac=float(rs.choice([0.0, 0.03, 0.1])))
    w = synth.make_ba_window(8000 + seed, **kw)
    r0, r1 = o.lba_solve(w, 10), s.solve(w, 10)
    same = (r0["stats"]["iterations"], r0["stats"]["trials"], r0["stats"]["stop_reason"]) == (r1["stats"]["iterations"], r1["stats"]["trials"], r1["stats"]["stop_reason"])
    d0, d1 = r0["points"] - w["points"], r1["points"] - w["points"]
    rel = np.abs(d0 - d1).max() / max(np.abs(d0).max(), 1e-12)
    worst = max(worst, rel if same else 0.0)
    if not same:
        bad += 1
        print("lba seed %d: control flow differs" % seed, kw, r0["stats"], r1["stats"])
s.close()
print("local BA: %d windows, %d with a different control flow, worst relative update difference %.2e" % (N, bad, worst))
from oracle_api import oracle_pose_inertial_optimize  # noqa: E402

for lf in (False, True):
    probs = []
    for seed in range(N):
        rs = np.random.RandomState(9000 + seed)
        probs.append(synth.make_pose_inertial_problem(9500 + seed, n=int(rs.randint(0, 800)), outlier_frac=float(rs.choice([0.0, 0.1, 0.3])),
                                                      stereo_frac=float(rs.choice([0.0, 0.5, 1.0])), noise_px=float(rs.choice([0.3, 1.0])), last_frame=lf)[0])
    isol = pkg.InertialSolver()
    res = isol.pose_optimize_batch(probs)
    isol.close()
    bad = 0; worst = 0.0
    for pr, r1 in zip(probs, res):
        r0 = oracle_pose_inertial_optimize(o, pr)
        if not (np.array_equal(r0["outlier"], r1["outlier"]) and r0["n_bad"] == r1["n_bad"]):
            bad += 1
            continue
        d0, d1 = r0["twb"] - pr["twb"][1], r1["twb"] - pr["twb"][1]
        worst = max(worst, np.abs(d0 - d1).max() / max(np.abs(d0).max(), 1e-12))
    print("pose-inertial (%s): %d frames, %d with different outlier flags, worst relative update difference %.2e" % ("last frame" if lf else "last key frame", N, bad, worst))
