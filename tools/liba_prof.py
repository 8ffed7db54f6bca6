#!/usr/bin/env python3
"""LocalInertialBA alone (for ORBX_LBA_TIMING=1 and rocprofv3 --kernel-trace --stats): the bench window, N solves."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
w, _ = synth.make_inertial_window(0, n_opt=10, n_points=800, obs_per_point=6, n_covisible_fixed=10)
s = pkg.InertialSolver()
s.solve(w); s.solve(w)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter(); it = 0
for _ in range(N):
    it += s.solve(w)["stats"]["iterations"]
dt = time.perf_counter() - t0
print("%d solves, %d iterations, %.3f ms per call" % (N, it, 1e3 * dt / N))
s.close()
