#!/usr/bin/env python3
"""Soak of the two device-resident chains (GPU box; not part of the test suite): the runners of tests/test_chain_gpu.py -- extract ->
DBoW2 transform -> SearchByBoW on a device plan, and extract -> SearchByProjection(last frame) -> PoseOptimization, each against the
oracle chain on the same images -- with random image sizes, batch sizes, shifts, vocabularies and levelsup.  usage: soak_chain.py [n]"""
import importlib
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from oracle_api import Oracle  # noqa: E402
import test_chain_gpu as T  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
oracle = Oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(17)
bad = 0
for it in range(N):
    size = [(640, 480), (752, 480), (512, 384), (848, 480), (400, 300)][int(rs.randint(0, 5))]
    try:
        kw = dict(P=int(rs.choice([1, 2, 5, 17])), seed0=5000 + 40 * it, size=size, levelsup=int(rs.choice([0, 1, 2])), min_matches=20,
                  voc_kw=dict(k=int(rs.choice([6, 10])), L=int(rs.choice([3, 4])), ragged=bool(rs.randint(0, 2)), tie_frac=float(rs.choice([0.0, 0.05])),
                              stop_frac=float(rs.choice([0.0, 0.02]))))
        kw["levelsup"] = min(kw["levelsup"], kw["voc_kw"]["L"] - 1)
        T.run_front_end_chain(pkg, oracle, synth, **kw)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL front-end chain %d %r: %s | %s" % (it, kw, type(e).__name__, " / ".join(traceback.format_exc().strip().splitlines()[-3:])[:400]), flush=True)
    try:
        kw = dict(B=int(rs.choice([1, 3, 8, 33])), seed0=6000 + 40 * it, shift=int(rs.choice([1, 3, 6])), size=size, min_matches=50, min_inliers=30)
        T.run_tracking_chain(pkg, oracle, synth, **kw)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL tracking chain %d %r: %s | %s" % (it, kw, type(e).__name__, " / ".join(traceback.format_exc().strip().splitlines()[-3:])[:400]), flush=True)
print("chain soak: %d runs of each chain, %d failures" % (N, bad))
sys.exit(1 if bad else 0)
