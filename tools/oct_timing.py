#!/usr/bin/env python3
"""Cycle split of the level-0 octree wave of one frame (GPU box).  Needs orbx_extractor.hip compiled with -DORBX_OCT_TIMING
(see tools/fast_timing.py for the recipe; rebuild with `make -B` afterwards)."""
import ctypes as C
import importlib
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
ex = pkg.Extractor()
img = synth.make_frame(0)
ex(img); ex(img)
out = (C.c_ulonglong * 8)()
pkg.lib.orbx_debug_oct_prof(out)
v = list(out)
print("level-0 wave: %d cycles, %d candidates, %d divides: partition %d, bookkeeping %d, sync %d cycles (per divide %.0f / %.0f / %.0f)" % (
    v[4], v[5], v[3], v[0], v[1], v[2], v[0] / max(v[3], 1), v[1] / max(v[3], 1), v[2] / max(v[3], 1)))
print("  gather + roots %d, introsort %d, final selection %d cycles" % (v[6], v[7] >> 32, v[7] & 0xFFFFFFFF))
