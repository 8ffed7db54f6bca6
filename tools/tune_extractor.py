#!/usr/bin/env python3
"""Prints per-stage device times of the extractor for a few batch sizes / octree LDS key capacities (GPU box only)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the HIP library: one HIP runtime per process)

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")


def run(B, reps=5):
    dev = torch.device("cuda", 0)
    imgs = synth.make_frames(8)
    d_imgs = torch.from_numpy(np.concatenate([imgs] * ((B + 7) // 8))[:B].copy()).to(dev)
    ex = pkg.Extractor()
    cap = ex.max_keypoints
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n, d_m, d_s = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream

    def go():
        ex.extract_batch_device(d_imgs.data_ptr(), B, 640, 480, 640, 640 * 480, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                d_n.data_ptr(), d_m.data_ptr(), d_s.data_ptr(), (0, 1000), st)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ex.profile_enable(True)
    acc = {}
    for _ in range(reps):
        go(); torch.cuda.synchronize()
        for k, v in ex.profile_read().items():
            acc[k] = acc.get(k, 0) + v / reps
    ex.close()
    return wall, acc


if __name__ == "__main__":
    for cap in os.environ.get("CAPS", "6144").split(","):
        os.environ["ORBX_OCT_LDS_KEYS"] = cap
        for B in [int(b) for b in os.environ.get("BATCHES", "64,256,1024").split(",")]:
            wall, acc = run(B)
            print("lds_keys=%s B=%d wall %.3f ms (%.0f frames/s) | " % (cap, B, wall * 1e3, B / wall) +
                  " ".join("%s=%.3f" % (k, v) for k, v in acc.items()), flush=True)
