#!/usr/bin/env python3
"""Soak of the many-windows-per-launch entries (GPU box; not part of the test suite): random batches of 1 .. 64 LocalBA windows
(lba_solve_batch) and visual-inertial windows (liba_solve_batch) -- sizes from one free pose / a handful of points up to 80 free poses,
stereo shares, robust / non-robust, user lambda, iteration limits, pre-set stop flags -- every window must come out with the BITS the
single-window entry gives (same kernel bodies, grid.y = window).  usage: soak_ba_batch.py [n_batches]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(99)
bad = 0; wins = 0
s, b = pkg.LbaSolver(), pkg.LbaBatch()
for it in range(N):
    W = int(rs.choice([1, 2, 5, 16, 17, 33, 64]))
    ws, flags = [], []
    for i in range(W):
        big = rs.uniform() < 0.1
        kw = dict(n_opt=int(rs.randint(1, 80 if big else 25)), n_fixed=int(rs.randint(1, 6)), n_points=int(rs.randint(5, 1500 if big else 300)),
                  obs_per_point=int(rs.randint(2, 9)), stereo_frac=float(rs.choice([0.0, 0.0, 0.3, 1.0])), outlier_frac=float(rs.choice([0.0, 0.03, 0.1])))
        w = synth.make_ba_window(20000 + 100 * it + i, **kw)
        if rs.uniform() < 0.2:
            w["huber_mono"] = 0.0; w["huber_stereo"] = 0.0
        ws.append(w)
        flags.append(np.ones(1, np.uint8) if rs.uniform() < 0.05 else None)
    iters = int(rs.choice([1, 4, 10])); lam = float(rs.choice([0.0, 0.0, 100.0]))
    ref = [s.solve(w, iters, lambda_init=lam, stop_flag=f) for w, f in zip(ws, flags)]
    got = b.solve(ws, iters, lambda_init=lam, stop_flags=flags)
    for i, (r0, r1) in enumerate(zip(ref, got)):
        wins += 1
        ok = r1["stats"] == r0["stats"] and all(np.array_equal(r1[k], r0[k]) for k in ("pose_q", "pose_t", "points", "chi2", "depth_positive"))
        if not ok:
            bad += 1
            print("MISMATCH LocalBA batch %d (W %d iters %d lambda %g) window %d: %r vs %r" % (it, W, iters, lam, i, r1["stats"], r0["stats"]), flush=True)
s.close(); b.close()
print("lba_solve_batch: %d batches, %d windows, %d not bit-identical to lba_solve" % (N, wins, bad), flush=True)
bad_i = 0; wins_i = 0
si, bi = pkg.InertialSolver(), pkg.LibaBatch()
for it in range(N):
    W = int(rs.choice([1, 2, 5, 16, 33, 64]))
    ws = []
    for i in range(W):
        kw = dict(n_opt=int(rs.randint(2, 26)), n_points=int(rs.randint(30, 500)), obs_per_point=int(rs.randint(3, 8)),
                  stereo_frac=float(rs.choice([0.0, 0.4, 1.0])), n_covisible_fixed=int(rs.choice([0, 3, 10])), bias_error=float(rs.choice([0.0, 0.001])))
        pr, _ = synth.make_inertial_window(30000 + 100 * it + i, **kw)
        if rs.uniform() < 0.4:
            pr["lambda_init"] = 1e-2; pr["max_iters"] = 4
        ws.append(pr)
    ref = [si.solve(w) for w in ws]
    got = bi.solve(ws)
    for i, (r0, r1) in enumerate(zip(ref, got)):
        wins_i += 1
        ok = r1["stats"] == r0["stats"] and all(np.array_equal(r1[k], r0[k]) for k in ("Rwb", "twb", "vel", "bg", "ba", "points", "chi2", "depth_positive"))
        if not ok:
            bad_i += 1
            print("MISMATCH LocalInertialBA batch %d (W %d) window %d: %r vs %r" % (it, W, i, r1["stats"], r0["stats"]), flush=True)
si.close(); bi.close()
print("liba_solve_batch: %d batches, %d windows, %d not bit-identical to liba_solve" % (N, wins_i, bad_i))
sys.exit(1 if bad + bad_i else 0)
