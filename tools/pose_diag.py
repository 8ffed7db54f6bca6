#!/usr/bin/env python3
"""Prints HIP vs oracle per-round Levenberg statistics of PoseOptimization for a list of synthetic frames (GPU box)."""
import importlib
import os
import sys

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_api import Oracle, build_oracle, oracle_pose_optimize  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
build_oracle()
o = Oracle()
ws = [synth.make_pose_problem(20 + i, n=100 + 37 * i, outlier_frac=0.1, stereo_frac=0.25 * (i % 3)) for i in range(24)]
s = pkg.PoseSolver()
rs = s.optimize_batch(ws)
for i, (w, r) in enumerate(zip(ws, rs)):
    g = oracle_pose_optimize(o, w)
    dt = np.abs(g["t"] - w["t"]).max()
    print(i, "rel dt %.2e" % (np.abs(r["t"] - g["t"]).max() / dt), "outl diff", int((r["outlier"] != g["outlier"]).sum()),
          "it", r["iterations"], g["iterations"], "tr", r["trials"], g["trials"],
          "chi", ["%.3e" % (abs(a - b) / max(abs(b), 1e-300)) for a, b in zip(r["chi2"], g["chi2"])])
