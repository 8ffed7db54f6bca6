#!/usr/bin/env python3
"""Section-wise cycle counts of k_pose_opt (GPU box).  Needs a library whose pose_solver.hip was compiled with
-DPOSE_TIMING (the kernel then returns clock64() sums in PoseResult.chi2[] / t[] instead of the real values):
  cd orb_slam3-1_amd/csrc && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DPOSE_TIMING -c -o pose_solver.o \
     pose_solver.hip && hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbslam3_hip.so *.o
Rebuild with `make -B` afterwards."""
import os
import importlib, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd"); synth = importlib.import_module("orb_slam3-1_amd.synth")
s = pkg.PoseSolver()
w = synth.make_pose_problem(0, n=300, outlier_frac=0.1)
s.optimize(w); r = s.optimize(w)
print("kernel ms", s.last_kernel_ms(), "iters", r["iterations"], "trials", r["trials"])
print("cycles: build %d  reduceH %d  serial %d  trialpass %d  trialreduce+update %d  classify %d" % (r["chi2"][0], r["chi2"][1], r["chi2"][2], r["chi2"][3], r["t"][0], r["t"][1]))
