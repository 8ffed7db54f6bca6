#!/usr/bin/env python3
"""Single-frame latency of the per-frame inertial optimisation (liba_pose_optimize_batch on ONE frame) and of LocalInertialBA
(liba_solve): call time seen by the host; run under `rocprofv3 --kernel-trace --stats` for the kernel times."""
import importlib
import os
import sys
import time

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
sol = pkg.InertialSolver(device=0)
for lf in (False, True):
    w = [synth.make_pose_inertial_problem(100, n=300, outlier_frac=0.1, last_frame=lf)[0]]
    q = sol.pose_prepare(w)
    sol.pose_launch(q)
    t0 = time.perf_counter()
    for _ in range(20):
        sol.pose_launch(q)
    print("pose_inertial last_frame=%d: %.3f ms per single-frame C call" % (lf, 1e3 * (time.perf_counter() - t0) / 20))
iw, _ = synth.make_inertial_window(0, n_opt=10, n_points=800, obs_per_point=6, n_covisible_fixed=10)
sol.solve(iw)
t0 = time.perf_counter()
for _ in range(5):
    r = sol.solve(iw)
print("liba_solve: %.3f ms per call, %d iterations, %d trials" % (1e3 * (time.perf_counter() - t0) / 5, r["stats"]["iterations"], r["stats"]["trials"]))
sol.close()
