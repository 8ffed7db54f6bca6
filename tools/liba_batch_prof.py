#!/usr/bin/env python3
"""liba_solve_batch alone, for rocprofv3 --kernel-trace --stats: W windows of the inertial_ba bench leg (10 temporal key frames, 800 points,
5200 edges, 10 links), N calls.
  python tools/liba_batch_prof.py [W] [N]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (one HIP runtime)

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ws = [synth.make_inertial_window(i, n_opt=10, n_points=800, obs_per_point=6, n_covisible_fixed=10)[0] for i in range(W)]
b = pkg.LibaBatch()
prep = b.prepare(ws)
b.run(prep); b.run(prep)
t0 = time.perf_counter()
it, dev = 0, 0.0
for _ in range(N):
    r = b.run(prep)
    it += sum(x["stats"]["iterations"] for x in r); dev += b.last_device_ms()
dt = time.perf_counter() - t0
print("%d windows x %d calls: %d iterations, %.3f ms per call, %.3f ms of it Levenberg rounds on the device -> %.0f iterations/s (device), %.0f (whole call)" % (
    W, N, it, 1e3 * dt / N, dev / N, it / (dev * 1e-3), it / dt))
b.close()
