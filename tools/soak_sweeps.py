#!/usr/bin/env python3
"""Soak of the matcher / BA / pose / vocabulary / stereo entry points (GPU box; not part of the test suite): the seeded sweep cases of
tests/test_sweeps_gpu.py -- each draws its sizes and parameters from its seed and compares the device result with the CPU oracle -- run
with MANY more seeds than the suite does.  usage: soak_sweeps.py [seeds_per_sweep] [first_seed]"""
import importlib
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (one HIP runtime: before the package)
from oracle_api import Oracle  # noqa: E402
import test_sweeps_gpu as T  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
sm = importlib.import_module("orb_slam3-1_amd.synth_match")
oracle = Oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
sweeps = [("SearchByBoW (KF,F) / (KF,KF)", T.test_bow_sweep, dict(pkg=pkg, oracle=oracle, synth=synth)),
          ("SearchByProjection (map points / last frame)", T.test_projection_sweep, dict(pkg=pkg, oracle=oracle, sm=sm)),
          ("LocalBA", T.test_lba_sweep, dict(pkg=pkg, oracle=oracle, synth=synth)),
          ("PoseOptimization + vocabulary transform", T.test_pose_and_vocab_sweep, dict(pkg=pkg, oracle=oracle, synth=synth)),
          ("matcher family (KF / Sim3 projection, Fuse, triangulation, initialisation)", T.test_matcher_family_sweep, dict(pkg=pkg, oracle=oracle, sm=sm)),
          ("ComputeStereoMatches", T.test_stereo_sweep, dict(pkg=pkg, oracle=oracle, synth=synth))]
total_bad = 0
for name, fn, fx in sweeps:
    bad = 0
    t0 = time.time()
    for seed in range(S0, S0 + N):
        try:
            fn(seed=seed, **fx)
        except Exception as e:  # noqa: BLE001
            bad += 1
            msg = traceback.format_exc().strip().splitlines()
            print("FAIL %s seed %d: %s | %s" % (name, seed, type(e).__name__, " / ".join(msg[-3:])[:400]), flush=True)
    total_bad += bad
    print("%s: %d seeds (%d..%d), %d failures, %.0f s" % (name, N, S0, S0 + N - 1, bad, time.time() - t0), flush=True)
print("sweep soak: %d cases, %d failures" % (N * len(sweeps), total_bad))
sys.exit(1 if total_bad else 0)
