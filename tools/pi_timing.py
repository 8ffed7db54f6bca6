#!/usr/bin/env python3
"""Cycle split of the Gauss-Newton iterations of k_pose_inertial (one frame).  Needs lba_solver.hip compiled with
-DLIBA_PI_TIMING (hipcc ... -DLIBA_PI_TIMING -c lba_solver.hip, then make; rebuild with `make -B` afterwards)."""
import ctypes as C
import importlib
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
sol = pkg.InertialSolver(device=0)
names = ["visual edges + sums", "wait for the link wave", "-", "reduce + assemble", "solve", "update", "classification", "final Hessian"]
for lf in (False, True):
    w = [synth.make_pose_inertial_problem(100, n=300, outlier_frac=0.1, last_frame=lf)[0]]
    sol.pose_optimize_batch(w)
    out = (C.c_ulonglong * 16)()
    pkg.lib.liba_debug_pi_prof(out, 1)
    sol.pose_optimize_batch(w)
    pkg.lib.liba_debug_pi_prof(out, 1)
    v = list(out)
    print("last_frame=%d: total %d cycles" % (lf, sum(v)))
    print("  " + ", ".join("%s %d" % (n, c) for n, c in zip(names, v[:8])) + ", loop top %d" % v[15])
sol.close()
