#!/usr/bin/env python3
"""Experiment (GPU box): does splitting a batch of B frames over S independent HIP streams (one Extractor + BoW plan per
stream, B/S frames each) raise frames/s?  The stages have different bottlenecks (FAST: VALU; descriptors / octree: latency),
so concurrent sub-batches can fill each other's bubbles."""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
host = synth.make_frames(8, seed0=0)
d_imgs = torch.from_numpy(np.concatenate([host] * (B // 8)).copy()).to(dev)
match_sets = [synth.make_match_set(50 + i) for i in range(4)]
matcher = pkg.Matcher(0.7, True)
for S in (1, 2, 4, 8):
    b = B // S
    exs = [pkg.Extractor(1000, 1.2, 8, 20, 7) for _ in range(S)]
    cap = exs[0].max_keypoints
    streams = [torch.cuda.Stream() for _ in range(S)]
    bufs = [dict(kps=torch.zeros(b * cap * 28, dtype=torch.uint8, device=dev), desc=torch.zeros(b * cap * 32, dtype=torch.uint8, device=dev),
                 n=torch.zeros(b, dtype=torch.int32, device=dev), mono=torch.zeros(b, dtype=torch.int32, device=dev),
                 st=torch.zeros(b, dtype=torch.int32, device=dev)) for _ in range(S)]
    plans = [matcher.bow_plan([match_sets[i % 4] for i in range(b)]) for _ in range(S)]

    def step():
        for s in range(S):
            q = bufs[s]
            exs[s].extract_batch_device(d_imgs.data_ptr() + s * b * 640 * 480, b, 640, 480, 640, 640 * 480, q["kps"].data_ptr(), q["desc"].data_ptr(), cap,
                                        q["n"].data_ptr(), q["mono"].data_ptr(), q["st"].data_ptr(), (0, 1000), streams[s].cuda_stream)
            plans[s].run(streams[s].cuda_stream)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("B=%d streams=%d (x%d frames): %.0f frames/s  %.3f ms/step" % (B, S, b, B * K / dt, 1e3 * dt / K))
    for e in exs:
        e.close()
    for p in plans:
        p.close()
