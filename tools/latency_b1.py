#!/usr/bin/env python3
"""Device time of one extract call for small batches (GPU box): frames resident in HBM, HIP-event stage times."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
dev = torch.device("cuda", 0)
host = synth.make_frames(8, seed0=0)
for B in (64, 1, 4, 16, 1):       # (the first configuration also brings the clocks up: a cold GPU runs the latency-bound stages at half speed)
    imgs = np.concatenate([host] * ((B + 7) // 8))[:B]
    d = torch.from_numpy(imgs.copy()).to(dev)
    ex = pkg.Extractor()
    cap = ex.max_keypoints
    kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev); desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev); mono = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), mono.data_ptr(),
                                st.data_ptr(), (0, 1000), stream)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ex.profile_enable(True); run(); torch.cuda.synchronize(); prof = ex.profile_read(); ex.profile_enable(False)
    print("B=%3d  %.3f ms per call (%.3f ms per frame)  stages %s" % (B, ms, ms / B, {k: round(v, 3) for k, v in prof.items()}))
    ex.close()

# the literal drop-in call: host image in, host key points / descriptors out (ORBextractor::operator())
import time
ex = pkg.Extractor()
img = host[0]
for _ in range(5):
    ex(img)
t0 = time.perf_counter()
for _ in range(50):
    mono, kps, desc = ex(img)
dt = (time.perf_counter() - t0) / 50
print("host-buffer operator(): %.3f ms per 640x480 frame (%d key points), upload + kernels + download + sync" % (1e3 * dt, len(kps)))
ex.close()
