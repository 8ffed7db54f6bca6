#!/usr/bin/env python3
"""Phase split of k_chol_step's factoring workgroup (GPU box).  Needs lba_solver.hip compiled with -DLBA_STEP_TIMING:
  cd orb_slam3-1_amd/csrc && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DLBA_STEP_TIMING -c -o lba_solver.o lba_solver.hip \\
     && hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbslam3_hip.so *.o        (then `make -B` restores the product build)"""
import ctypes as C
import importlib
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
w = synth.make_ba_window(0)
s = pkg.LbaSolver()
s.solve(w, 10)
out = (C.c_ulonglong * 8)()
pkg.lib.lba_debug_step_prof(out)
pkg.lib.lba_debug_tile_prof(out)
for _ in range(5):
    s.solve(w, 10)
pkg.lib.lba_debug_step_prof(out)
v = list(out)
n = max(v[7], 1)
names = ["stage Linv / A into LDS", "X = A Linv^T", "store X to Lp", "load the tile", "T -= X X^T", "write back / pad", "factor + invert"]
for k, nm in enumerate(names):
    print("%-26s %7.2f us" % (nm, v[k] / n / 100.0))
print("%-26s %7.2f us over %d steps" % ("total", sum(v[:7]) / n / 100.0, v[7]))
pkg.lib.lba_debug_tile_prof(out)
t = list(out)
print("chol_tile_mfma, cycles over all 4-column groups of all workgroups' thread 0: MFMA drain + loop %d, publish %d, barrier %d, pivot block + M %d" % (t[0], t[1], t[2], t[3]))
s.close()
