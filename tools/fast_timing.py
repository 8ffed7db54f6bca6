#!/usr/bin/env python3
"""Cycle split of k_fast_strips summed over thread 0 of every workgroup (GPU box).  Needs orbx_extractor.hip compiled with
-DORBX_FAST_TIMING (hipcc ... -DORBX_FAST_TIMING -c orbx_extractor.hip, then make; rebuild with `make -B` afterwards)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
ex = pkg.Extractor()
B = 64
imgs = np.stack([synth.make_frame(i) for i in range(B)])
ex.extract_batch(imgs) if hasattr(ex, "extract_batch") else [ex(im) for im in imgs]
out = (C.c_ulonglong * 16)()
pkg.lib.orbx_debug_fast_prof(out)
ex.extract_batch(imgs) if hasattr(ex, "extract_batch") else [ex(im) for im in imgs]
pkg.lib.orbx_debug_fast_prof(out)
v = list(out)
names = ["load + clears + tables", "stage 1 (compass test)", "expand + stage 2 (scores)", "barrier", "nms", "rank", "emit"]
tot = sum(v[:15])
print("iniThFAST pass:   " + ", ".join("%s %.1f%%" % (n, 100.0 * c / tot) for n, c in zip(names, v[:7])))
print("minThFAST passes: " + ", ".join("%s %.1f%%" % (n, 100.0 * c / tot) for n, c in zip(names[1:], v[9:15])) + " (%d cells fell back)" % v[15])
