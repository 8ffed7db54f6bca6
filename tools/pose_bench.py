#!/usr/bin/env python3
"""Times the HIP PoseOptimization for several batch sizes (GPU box): wall time per call and kernel time from HIP events."""
import importlib
import os
import sys
import time

import torch  # noqa: F401  (one HIP runtime per process: torch first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
n_edges = int(sys.argv[1]) if len(sys.argv) > 1 else 300
base = [synth.make_pose_problem(i, n=n_edges, outlier_frac=0.1, stereo_frac=0.0) for i in range(64)]
s = pkg.PoseSolver()
for B in (1, 16, 64, 256, 1024, 4096):
    ws = (base * ((B + 63) // 64))[:B]
    prep = s.prepare(ws)
    s.run(prep)
    reps = 20 if B <= 256 else 5
    t0 = time.perf_counter(); k = 0.0
    for _ in range(reps):
        s.launch(prep); k += s.last_kernel_ms()
    dt = (time.perf_counter() - t0) / reps
    print("B=%5d edges=%d call %.3f ms kernel %.3f ms -> %.0f frames/s (kernel-only %.0f)" % (B, n_edges, 1e3 * dt, k / reps, B / dt, B / (k / reps * 1e-3)))
s.close()
