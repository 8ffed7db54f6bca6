#!/usr/bin/env python3
"""LocalBA alone, for rocprofv3 --kernel-trace --stats (SURVEY config #4: 50 KF / 2000 MP / 20 k edges), N solves."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (one HIP runtime)

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
w = synth.make_ba_window(0)
s = pkg.LbaSolver()
s.solve(w, 10)
t0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
it = 0
for _ in range(N):
    r = s.solve(w, 10)
    it += r["stats"]["iterations"]
dt = time.perf_counter() - t0
print("%d solves, %d iterations, %.3f ms per solve call (upload + iterations + download)" % (N, it, 1e3 * dt / N))
s.close()
