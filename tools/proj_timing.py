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
MODE = sys.argv[1] if len(sys.argv) > 1 else ""
if MODE in ("chain", "tlm"):       # the tracking chain's workload: features the extractor finds on a frame and on the same frame 3 px to the right
    import numpy as np
    synth = importlib.import_module("orb_slam3-1_amd.synth")
    img = synth.make_frame(0, 640, 480)
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    _, k_last, d_last = ex(img)
    _, k_cur, d_cur = ex(np.ascontiguousarray(np.roll(img, 3, axis=1)))
    scale = np.asarray(ex.GetScaleFactors(), np.float32)
    ex.close()
    g = dict(x=np.ascontiguousarray(k_cur["x"]), y=np.ascontiguousarray(k_cur["y"]), octave=np.ascontiguousarray(k_cur["octave"]).astype(np.int32),
             min_x=0.0, min_y=0.0, max_x=640.0, max_y=480.0, cols=64, rows=48)
    dF, angF = d_cur, np.ascontiguousarray(k_cur["angle"])
    n_l = len(k_last)
    last = dict(u=np.ascontiguousarray(k_last["x"] + 3.0), v=np.ascontiguousarray(k_last["y"]), octave=np.ascontiguousarray(k_last["octave"]).astype(np.int32),
                angle=np.ascontiguousarray(k_last["angle"]), valid=np.ones(n_l, np.uint8), desc=d_last, has_obs=np.ones(n_l, np.uint8))
    assign = np.full(len(k_cur), -1, np.int32); occ = np.zeros(len(k_cur), np.uint8)
else:
    g, dF, angF, scale, last, assign, occ = sm.make_last_frame_case(0)
out = (C.c_ulonglong * 8)()
if MODE == "tlm":       # TrackLocalMap's search: SearchByProjection(Frame, MapPoints, th = 1) on the last frame's points + as many again a few pixels off
    rs = np.random.RandomState(9)
    off = rs.uniform(-4, 4, n_l).astype(np.float32)
    mp = dict(in_view=np.ones(2 * n_l, np.uint8), u=np.concatenate([last["u"], last["u"] + off]), v=np.concatenate([last["v"], last["v"] + off[::-1]]),
              level=np.concatenate([last["octave"], last["octave"]]), view_cos=np.full(2 * n_l, 0.999, np.float32), depth=np.full(2 * n_l, 5.0, np.float32),
              desc=np.ascontiguousarray(np.concatenate([d_last, d_last])), has_obs=np.ones(2 * n_l, np.uint8), bad=np.zeros(2 * n_l, np.uint8))
    m = pkg.Matcher(0.8, True)
    run = lambda: m.SearchByProjection(g, dF, scale, mp, 1.0, assign.copy(), occ.copy())
else:
    m = pkg.Matcher(0.9, True)
    run = lambda: m.SearchByProjection_last(g, dF, angF, scale, last, 15.0, assign.copy(), occ.copy())
run()
pkg.lib.orbm_debug_proj_prof(out)
n = run()
pkg.lib.orbm_debug_proj_prof(out)
v = list(out)
print("matches %d, points %d, total cycles %d (%.0f per point): window walk %d, reduction %d" % (n, v[7], v[0], v[0] / max(v[7], 1), v[1], v[2]))
print("block-parallel kernel (k_proj_par, the slots mean something else there): passes %d, points resolved again %d (%d of them by a window walk); cycles of the serial phase %d (wave 0 merge %d), "
      "of the overlapped phase %d (wave 0 resolution %d, wave 1 set-up %d, slowest item-list wave %d)" % (v[3], v[4] & 0xFFFFFFFF, v[4] >> 32, v[5], v[1], v[6], v[2], v[0], v[7]))
