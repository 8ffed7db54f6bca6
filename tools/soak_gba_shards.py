#!/usr/bin/env python3
"""Soak of the sharded global BA (GPU box; not part of the test suite): random maps split into 1 .. 5 landmark shards, one host THREAD per
shard on the one GPU, every shard running lba_shard_optimize (the C-ABI Levenberg driver) with an all-reduce callback that sums the
shards' device buffers through host memory behind a barrier -- what ncclAllReduce does between ranks.  Every shard must walk the
single-shard solver's Levenberg path (iterations, trials, stop reason), end at its chi2 (1e-7) and move its poses / own points like it
(1e-4 of the update).  usage: soak_gba_shards.py [n]"""
import ctypes as C
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
dmod = importlib.import_module("orb_slam3-1_amd.distributed")
hip = C.CDLL("libamdhip64.so")
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(1234)
bad = 0
for it in range(N):
    world = int(rs.choice([1, 2, 3, 5]))
    kw = dict(n_opt=int(rs.randint(2, 70)), n_fixed=int(rs.randint(1, 5)), n_points=int(rs.randint(40, 1500)), obs_per_point=int(rs.randint(2, 9)),
              stereo_frac=float(rs.choice([0.0, 0.3, 1.0])), outlier_frac=float(rs.choice([0.0, 0.05])))
    w = synth.make_ba_window(60000 + it, **kw)
    if rs.uniform() < 0.5:
        w["huber_mono"] = 0.0; w["huber_stereo"] = 0.0          # loop closing: bRobust = false
    iters = int(rs.choice([2, 5, 10])); lam = float(rs.choice([0.0, 0.0, 100.0]))
    s = pkg.LbaSolver()
    ref = s.solve(w, iters, lambda_init=lam)
    s.close()
    parts = [dmod.partition_landmarks(w, r, world) for r in range(world)]
    shards = [pkg.LbaShard(p[0]) for p in parts]
    bar = threading.Barrier(world)
    slots = [None] * world
    res = [None] * world; errs = []

    def make_cb(r):
        def cb(dev_ptr, count, op, stream):
            if hip.hipStreamSynchronize(stream):
                return 1
            host = np.empty(count, np.float64)
            if hip.hipMemcpy(host.ctypes.data, dev_ptr, 8 * count, 2):
                return 1
            slots[r] = host
            bar.wait()
            tot = np.maximum.reduce(slots) if op == 1 else np.sum(slots, axis=0)      # (the same order in every thread: identical sums)
            bar.wait()
            return 1 if hip.hipMemcpy(dev_ptr, tot.ctypes.data, 8 * count, 1) else 0
        return cb

    def run(r):
        try:
            res[r] = shards[r].optimize(make_cb(r) if world > 1 else None, world, max_iters=iters, lambda_init=lam)
        except Exception as e:  # noqa: BLE001
            errs.append(e); bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    ok = not errs
    msg = repr(errs[:1])
    if ok:
        rs0 = ref["stats"]
        for r in range(world):
            st = res[r]
            same = (st["iterations"], st["trials"], st["stop_reason"]) == (rs0["iterations"], rs0["trials"], rs0["stop_reason"])
            chi_ok = abs(st["chi2_final"] - rs0["chi2_final"]) <= 1e-7 * abs(rs0["chi2_final"]) + 1e-18 * max(rs0["chi2_initial"], 1.0)
            out = shards[r].download()
            d0 = ref["pose_t"] - w["pose_t"]; d1 = out["pose_t"] - w["pose_t"]
            pose_ok = np.abs(d0 - d1).max() <= 1e-4 * max(np.abs(d0).max(), 1e-12)
            lo, hi = parts[r][1]
            p0 = ref["points"][lo:hi] - w["points"][lo:hi]; p1 = out["points"] - parts[r][0]["points"]
            pts_ok = (hi == lo) or np.abs(p0 - p1).max() <= 1e-4 * max(np.abs(p0).max(), 1e-12)
            if not (same and chi_ok and pose_ok and pts_ok):
                ok = False
                msg = "shard %d: path %s vs %s, chi2 %.12g vs %.12g, poses %s points %s" % (r, (st["iterations"], st["trials"], st["stop_reason"]),
                                                                                          (rs0["iterations"], rs0["trials"], rs0["stop_reason"]), st["chi2_final"], rs0["chi2_final"], pose_ok, pts_ok)
    for sh in shards:
        sh.close()
    if not ok:
        bad += 1
        print("FAIL map %d (world %d, %r, iters %d lambda %g): %s" % (it, world, kw, iters, lam, msg), flush=True)
print("sharded global BA soak: %d maps over 1 .. 5 shards, %d failures" % (N, bad))
sys.exit(1 if bad else 0)
