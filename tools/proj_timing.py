#!/usr/bin/env python3
"""Cycle split of k_proj for one frame (GPU box).  Needs orbm_matcher.hip compiled with -DORBM_PROJ_TIMING:
  cd orb_slam3-1_amd/csrc && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DORBM_PROJ_TIMING -c -o orbm_matcher.o \\
     orbm_matcher.hip && hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbslam3_hip.so *.o      (then `make -B`)"""
import ctypes as C
import importlib
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
sm = importlib.import_module("orb_slam3-1_amd.synth_match")
g, dF, angF, scale, last, assign, occ = sm.make_last_frame_case(0)
m = pkg.Matcher(0.9, True)
out = (C.c_ulonglong * 8)()
m.SearchByProjection_last(g, dF, angF, scale, last, 15.0, assign.copy(), occ.copy())
pkg.lib.orbm_debug_proj_prof(out)
n = m.SearchByProjection_last(g, dF, angF, scale, last, 15.0, assign.copy(), occ.copy())
pkg.lib.orbm_debug_proj_prof(out)
v = list(out)
print("matches %d, points %d, total cycles %d (%.0f per point): window walk %d, reduction %d" % (n, v[7], v[0], v[0] / max(v[7], 1), v[1], v[2]))
print("block-parallel kernel (k_proj_par, the slots mean something else there): passes %d, points searched again %d; cycles of the serial phase %d (wave 0 merge %d), "
      "of the overlapped phase %d (wave 0 resolution %d, wave 1 set-up %d, wave 2 item list %d)" % (v[3], v[4], v[5], v[1], v[6], v[2], v[0], v[7]))
