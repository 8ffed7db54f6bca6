#!/usr/bin/env python3
"""Cycle split of the level-0 octree wave of one frame (GPU box).  Needs orbx_extractor.hip compiled with -DORBX_OCT_TIMING
(hipcc ... -DORBX_OCT_TIMING -c orbx_extractor.hip, then make; rebuild with `make -B` afterwards)."""
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
out = (C.c_ulonglong * 10)()
pkg.lib.orbx_debug_oct_prof(out)
v = list(out)
names = ["gather", "roots", "count+scatter", "bookkeeping", "compact/other", "sort", "final selection"]
print("level-0 wave: %d cycles, %d candidates, %d divides" % (v[7], v[8], v[9]))
print("  " + ", ".join("%s %d" % (n, c) for n, c in zip(names, v[:7])))
