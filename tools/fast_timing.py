#!/usr/bin/env python3
"""Section-wise cycle sums of k_fast_cells over a batch (GPU box).  Needs orbx_extractor.hip compiled with
-DORBX_FAST_TIMING:
  cd orb_slam3-1_amd/csrc && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DORBX_FAST_TIMING -c \\
     -o orbx_extractor.o orbx_extractor.hip && hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbslam3_hip.so *.o
Rebuild with `make -B` afterwards."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
imgs = np.concatenate([synth.make_frames(16, seed0=0)] * 4)
ex = pkg.Extractor()
ex.extract_batch(imgs)
out = (C.c_ulonglong * 8)()
pkg.lib.orbx_debug_fast_prof(out)
ex.extract_batch(imgs)
pkg.lib.orbx_debug_fast_prof(out)
v = list(out)
tot = sum(v[:6])
names = ["load+zero", "pass1 quick", "pass2 score", "pass3 nms", "pass4 prefix", "pass5 emit"]
print("workgroups %d, survivors/cell %.1f, cycles/cell %.0f" % (v[7], v[6] / v[7], tot / v[7]))
for n_, c in zip(names, v[:6]):
    print("  %-13s %6.0f cycles/cell  %5.1f %%" % (n_, c / v[7], 100.0 * c / tot))
