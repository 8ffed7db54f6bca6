#!/usr/bin/env python3
"""Probe of size cliffs (GPU box; not part of the test suite): every entry point at sizes AROUND the internal limits of its kernels (LDS
staging budgets, fused / unfused solver paths, sort capacities), against the oracle.  A size must either work or be refused with a
clear error -- never fail with ORBX_ERR_HIP / ORBX_ERR_INTERNAL or differ.  usage: probe_limits.py"""
import importlib
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from oracle_api import Oracle, oracle_inertial_solve, oracle_pose_optimize, oracle_transform  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
sm = importlib.import_module("orb_slam3-1_amd.synth_match")
o = Oracle()
bad = 0


def case(name, fn):
    global bad
    try:
        r = fn()
        print("%-70s %s" % (name, r or "ok"), flush=True)
    except pkg.OrbxError as e:
        ok = e.code in (-2, -3)            # ORBX_ERR_CAPACITY / ORBX_ERR_ARG: an explicit refusal
        if not ok:
            bad += 1
        print("%-70s %s: %s" % (name, "refused" if ok else "FAIL", str(e)[:160]), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("%-70s FAIL %s | %s" % (name, type(e).__name__, " / ".join(traceback.format_exc().strip().splitlines()[-2:])[:300]), flush=True)


for n in (2500, 4000, 6000, 9000):
    def bow(n=n):
        ms = synth.make_match_set(7, n=n)
        n0, m0 = o.search_by_bow(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"], 0.7, True)
        m = pkg.Matcher(0.7, True)
        try:
            n1, m1 = m.SearchByBoW(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"])
        finally:
            m.close()
        assert n1 == n0 and np.array_equal(m1, m0)
    case("SearchByBoW %d x %d" % (n, n), bow)
for n in (1400, 1650, 1800, 2000, 3000, 5000, 8100, 9000, 17000):
    def proj(n=n):
        g, dF, aF, sc, last, a, oc = sm.make_last_frame_case(3, n=n, n_last=1200)
        a0, o0 = a.copy(), oc.copy()
        n0 = o.search_by_projection_last(g, dF, aF, sc, last, 15.0, True, a0, o0)
        m = pkg.Matcher(0.9, True)
        try:
            a1, o1 = a.copy(), oc.copy()
            n1 = m.SearchByProjection_last(g, dF, aF, sc, last, 15.0, a1, o1)
        finally:
            m.close()
        assert n1 == n0 and np.array_equal(a1, a0) and np.array_equal(o1, o0)
    case("SearchByProjection(F, lastF) %d features" % n, proj)
for npts in (3000, 9000, 30000):
    def fuse(npts=npts):
        g, dKF, sc, ur, isig, pts = sm.make_fuse_case(2, n=1500, n_pts=npts)
        bi0, bd0 = o.fuse_search(g, dKF, sc, ur, isig, pts, 3.0, True)
        m = pkg.Matcher(0.6, True)
        try:
            bi1, bd1 = m.FuseSearch(g, dKF, sc, ur, isig, pts, 3.0, True)
        finally:
            m.close()
        assert np.array_equal(bi1, bi0) and np.array_equal(bd1, bd0)
    case("Fuse search core, %d candidate points" % npts, fuse)
for n in (4096, 8192, 8193):
    def vocab(n=n):
        voc = synth.make_vocabulary(3, k=10, L=3)
        desc = np.ascontiguousarray(voc["desc"][np.arange(1, n + 1) % voc["n_nodes"]])
        (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(o, voc, desc, 2)
        v = pkg.Vocabulary(voc)
        try:
            (bi1, bv1), (fn1, fo1, ff1) = v.transform(desc, 2)
        finally:
            v.close()
        assert np.array_equal(bi1, bi0) and np.array_equal(bv1, bv0) and np.array_equal(ff1, ff0)
    case("vocabulary transform, %d descriptors" % n, vocab)
for n_opt in (78, 80, 81, 90, 130, 200):
    def lba(n_opt=n_opt):
        w = synth.make_ba_window(70 + n_opt, n_opt=n_opt, n_fixed=4, n_points=600, obs_per_point=8)
        r0 = o.lba_solve(w, 4)
        s = pkg.LbaSolver()
        try:
            r1 = s.solve(w, 4)
        finally:
            s.close()
        s0, s1 = r0["stats"], r1["stats"]
        assert (s1["iterations"], s1["trials"], s1["stop_reason"]) == (s0["iterations"], s0["trials"], s0["stop_reason"])
        d0, d1 = r0["points"] - w["points"], r1["points"] - w["points"]
        assert np.abs(d0 - d1).max() <= 1e-4 * max(np.abs(d0).max(), 1e-12)
    case("LocalBA, %d free poses (%d reduced unknowns)" % (n_opt, 6 * n_opt), lba)
for n_opt in (24, 25, 26, 32, 40):
    def liba(n_opt=n_opt):
        pr, _ = synth.make_inertial_window(9, n_opt=n_opt, n_points=300, obs_per_point=5)
        r0 = oracle_inertial_solve(o, pr)
        s = pkg.InertialSolver()
        try:
            r1 = s.solve(pr)
        finally:
            s.close()
        assert (r1["stats"]["iterations"], r1["stats"]["trials"]) == (r0["stats"]["iterations"], r0["stats"]["trials"])
        d0, d1 = r0["twb"] - pr["twb"], r1["twb"] - pr["twb"]
        assert np.abs(d0 - d1).max() <= 1e-4 * max(np.abs(d0).max(), 1e-12)
    case("LocalInertialBA, %d temporal key frames (%d unknowns)" % (n_opt, 15 * n_opt), liba)
for n in (511, 512, 513, 1024, 1025, 5000):
    def pose(n=n):
        w = synth.make_pose_problem(2, n=n, outlier_frac=0.1, stereo_frac=0.3)
        g = oracle_pose_optimize(o, w)
        s = pkg.PoseSolver()
        try:
            r = s.optimize(w)
        finally:
            s.close()
        assert np.array_equal(r["outlier"], g["outlier"]) and np.abs(r["t"] - g["t"]).max() <= 1e-4 * np.abs(np.asarray(g["t"]) - w["t"]).max() + 1e-12
    case("PoseOptimization, %d edges" % n, pose)
for size, nfeat in (((1920, 1080), 2000), ((1920, 1080), 8000), ((2560, 1440), 3000), ((4000, 3000), 1000), ((4200, 300), 1000)):
    def ext(size=size, nfeat=nfeat):
        img = synth.make_frame(4, size[0], size[1])
        r0, k0, d0 = o.extractor(nfeat, 1.2, 8, 20, 7).extract(img, (0, 1000))
        ex = pkg.Extractor(nfeat, 1.2, 8, 20, 7)
        try:
            r1, k1, d1 = ex(img, (0, 1000))
        finally:
            ex.close()
        assert r1 == r0 and len(k1) == len(k0) and all(np.array_equal(k1[f], k0[f]) for f in k0.dtype.names) and np.array_equal(d1, d0)
        return "ok (%d key points)" % len(k0)
    case("extractor %dx%d, %d features" % (size[0], size[1], nfeat), ext)
print("limits probe: %d failures" % bad)
sys.exit(1 if bad else 0)
