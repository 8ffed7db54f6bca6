#!/usr/bin/env python3
"""Soak of the ORACLE itself (CPU only): the definitional / independent numpy restatements the suite pins the oracle with -- FAST-9/16
from the definition, IC_Angle + rBRIEF in float32 without FMA, the two tracking searches from a grid-free restatement -- with many more
seeds and thresholds than the suite runs.  The oracle is what every device result is compared with: this is its own parity check.
usage: soak_oracle.py [n_seeds]"""
import importlib
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_api import Oracle, build_oracle  # noqa: E402
import test_oracle_definitional as D  # noqa: E402
import test_projection_oracle as P  # noqa: E402

build_oracle()
oracle = Oracle()
synth = importlib.import_module("orb_slam3-1_amd.synth")
sm = importlib.import_module("orb_slam3-1_amd.synth_match")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0; runs = 0
t0 = time.time()


def guard(name, fn, **kw):
    global bad, runs
    runs += 1
    try:
        fn(**kw)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL %s %r: %s | %s" % (name, {k: v for k, v in kw.items() if k not in ("oracle", "synth", "sm")}, type(e).__name__,
                                         " / ".join(traceback.format_exc().strip().splitlines()[-3:])[:400]), flush=True)


for seed in range(100, 100 + N):
    guard("FAST vs the definition", D.test_fast_against_the_definition, oracle=oracle, synth=synth, seed=seed, t=[5, 7, 12, 20, 25, 30, 40][seed % 7])
    guard("IC_Angle / rBRIEF vs the definition", D.test_orientation_and_descriptors_against_the_definition, oracle=oracle, synth=synth, seed=seed)
for i, frac in enumerate([None, 0.0, 0.3, 0.5, 1.0]):
    for lw in (0, 1, 2):
        for th in (3.0, 7.0, 15.0):
            guard("last-frame search vs numpy", P.test_last_frame_search_vs_numpy, oracle=oracle, sm=sm, frac=frac, lw=lw if frac is not None else 0, th=th, ori=bool((i + lw) & 1))
    for th in (1.0, 3.0, 7.0):
        for far in (False, True):
            guard("map-point search vs numpy", P.test_map_point_search_vs_numpy, oracle=oracle, sm=sm, frac=frac, th=th, far=far)
print("oracle soak: %d runs (FAST / IC_Angle / rBRIEF definitional on %d seeds, both tracking searches vs the grid-free restatement), %d failures, %.0f s" %
      (runs, N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
