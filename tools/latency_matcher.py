#!/usr/bin/env python3
"""Single-call latencies of the matcher and pose entry points as a single-stream tracker makes them (host arrays in, host arrays out;
GPU box), beside the CPU oracle's time for the same call."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from oracle_api import Oracle, oracle_pose_optimize  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
sm = importlib.import_module("orb_slam3-1_amd.synth_match")
o = Oracle()


def med(fn, n=30):
    fn(); fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


ms = synth.make_match_set(1)
m = pkg.Matcher(0.7, True)
g = med(lambda: m.SearchByBoW(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"]))
c = med(lambda: o.search_by_bow(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"], 0.7, True), 10)
print("SearchByBoW(KF, F) 1000 x 1000:            device call %.3f ms, oracle %.3f ms" % (g, c))
m.close()
gq, dF, aF, sc, last, a, oc = sm.make_last_frame_case(3, n=1000, n_last=900)
m = pkg.Matcher(0.9, True)
g = med(lambda: m.SearchByProjection_last(gq, dF, aF, sc, last, 15.0, a.copy(), oc.copy()))
c = med(lambda: o.search_by_projection_last(gq, dF, aF, sc, last, 15.0, True, a.copy(), oc.copy()), 10)
print("SearchByProjection(F, lastF) 1000 x 900:    device call %.3f ms, oracle %.3f ms" % (g, c))
m.close()
g2, dF2, aF2, sc2, mp, a2, oc2 = sm.make_projection_case(3, n=1000, n_mp=2000)
m = pkg.Matcher(0.8, True)
g = med(lambda: m.SearchByProjection(g2, dF2, sc2, mp, 3.0, a2.copy(), oc2.copy()))
c = med(lambda: o.search_by_projection(g2, dF2, sc2, mp, 3.0, 0.8, a2.copy(), oc2.copy()), 10)
print("SearchByProjection(F, MapPoints) 1000 x 2000: device call %.3f ms, oracle %.3f ms" % (g, c))
m.close()
w = synth.make_pose_problem(1, n=300, outlier_frac=0.1)
ps = pkg.PoseSolver()
g = med(lambda: ps.optimize(w))
c = med(lambda: oracle_pose_optimize(o, w), 10)
print("PoseOptimization 300 edges:                  device call %.3f ms, oracle %.3f ms" % (g, c))
ps.close()
voc = synth.make_vocabulary(1, k=10, L=4)
vd = np.ascontiguousarray(voc["desc"][np.arange(1, 1001) % voc["n_nodes"]])
from oracle_api import oracle_transform  # noqa: E402
vv = pkg.Vocabulary(voc)
g = med(lambda: vv.transform(vd, 2))
c = med(lambda: oracle_transform(o, voc, vd, 2), 10)
print("DBoW2 transform 1000 descriptors (k 10, L 4):  device call %.3f ms, oracle %.3f ms" % (g, c))
vv.close()
