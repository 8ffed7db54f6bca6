#!/usr/bin/env python3
"""How much of a 16x16x4 f64 MFMA tile would be useful work if the Schur products S_ij -= Z_a W_b^T (6x3 . 3x6 per shared
landmark) of the bench's LocalBA window were gathered into tiles covering 2x2 pose pairs?  CPU only."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
synth = importlib.import_module("orb_slam3-1_amd.synth")
w = synth.make_ba_window(0)
free = [p for p in range(len(w["pose_q"])) if not w["pose_fixed"][p]]
seen = {p: set() for p in range(len(w["pose_q"]))}
for e in range(len(w["edge_point"])):
    seen[int(w["edge_pose"][e])].add(int(w["edge_point"][e]))
pairs = [(free[k], free[k + 1]) for k in range(0, len(free) - 1, 2)]
live = tot = 0
for (i1, i2) in pairs:
    for (j1, j2) in pairs:
        if j1 > i1:
            continue
        union = (seen[i1] | seen[i2]) & (seen[j1] | seen[j2])       # K runs over every landmark any of the four pairs shares
        live += sum(len(seen[a] & seen[b]) for a in (i1, i2) for b in (j1, j2))
        tot += 4 * len(union)
print("live K slots per 2x2-pose tile: %.1f %%; with 12x12 of the 16x16 outputs used: %.1f %% of the MFMA work is useful" %
      (100.0 * live / tot, 100.0 * live / tot * 144 / 256))
