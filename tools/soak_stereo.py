#!/usr/bin/env python3
"""Soak of Frame::ComputeStereoMatches on the device (GPU box; not part of the test suite): random rectified pairs -- image sizes (odd
widths included), feature budgets, pyramid depths, scale factors, disparity ranges, baselines -- through the host entry and, for batches
of 1 .. 40 pairs, the device-resident entry, mvuRight / mvDepth bit for bit against the oracle.  usage: soak_stereo.py [n]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oracle_api import Oracle, oracle_stereo_matches  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
o = Oracle()
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(2024)
bad = 0; pairs_n = 0; matched = 0
for it in range(N):
    w = int(rs.randint(200, 900)); h = int(rs.randint(120, 600))
    nfeat = int(rs.choice([100, 500, 1000, 3000])); scale = float(rs.choice([1.1, 1.2, 1.5]))
    nlev = int(rs.randint(1, 9))
    while nlev > 1 and min(w, h) / scale ** (nlev - 1) < 60:
        nlev -= 1
    band = int(rs.choice([20, 60, 150])); dmax = int(rs.choice([5, 40, 120])); dmax = min(dmax, w // 3)
    mb = float(rs.choice([0.05, 0.11, 0.5])); mbf = mb * float(rs.choice([200.0, 435.0, 700.0]))
    B = int(rs.choice([1, 2, 9, 40]))
    n_img = min(B, 3)
    prs = [synth.make_stereo_pair(9000 + 10 * it + k, w, h, band=band, dmin=1, dmax=dmax) for k in range(n_img)]
    refs = []
    for (l_, r_) in prs:
        oL, oR = o.extractor(nfeat, scale, nlev, 20, 7), o.extractor(nfeat, scale, nlev, 20, 7)
        _, kL0, dL0 = oL.extract(l_, (0, 0)); _, kR0, dR0 = oR.extract(r_, (0, 0))
        _, ur0, dp0 = oracle_stereo_matches(oL, oR, kL0, dL0, kR0, dR0, mb, mbf)
        refs.append((len(kL0), ur0, dp0))
    exL, exR = pkg.Extractor(nfeat, scale, nlev, 20, 7), pkg.Extractor(nfeat, scale, nlev, 20, 7)
    try:
        # host entry on the first pair
        _, kL, dL = exL(prs[0][0], (0, 0)); _, kR, dR = exR(prs[0][1], (0, 0))
        ur1, dp1 = exL.stereo_matches(exR, kL, dL, kR, dR, mb, mbf)
        ok_host = np.array_equal(ur1, refs[0][1]) and np.array_equal(dp1, refs[0][2])
        # device entry on the batch
        cap = exL.max_keypoints
        L = np.stack([prs[b % n_img][0] for b in range(B)]); R = np.stack([prs[b % n_img][1] for b in range(B)])
        outs = []
        st = torch.cuda.current_stream().cuda_stream
        for ex, imgs in ((exL, L), (exR, R)):
            d_img = torch.from_numpy(imgs.copy()).to(dev)
            od = dict(kps=torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev), desc=torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev),
                      n=torch.zeros(B, dtype=torch.int32, device=dev), mono=torch.zeros(B, dtype=torch.int32, device=dev), st=torch.zeros(B, dtype=torch.int32, device=dev), img=d_img)
            ex.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, od["kps"].data_ptr(), od["desc"].data_ptr(), cap, od["n"].data_ptr(), od["mono"].data_ptr(), od["st"].data_ptr(), (0, 0), st)
            outs.append(od)
        d_ur = torch.zeros(B * cap, dtype=torch.float32, device=dev); d_dp = torch.zeros(B * cap, dtype=torch.float32, device=dev)
        exL.stereo_matches_device(exR, B, outs[0]["kps"].data_ptr(), outs[0]["desc"].data_ptr(), outs[0]["n"].data_ptr(),
                                  outs[1]["kps"].data_ptr(), outs[1]["desc"].data_ptr(), outs[1]["n"].data_ptr(), cap, mb, mbf, d_ur.data_ptr(), d_dp.data_ptr(), st)
        torch.cuda.synchronize()
        ur = d_ur.cpu().numpy().reshape(B, cap); dp = d_dp.cpu().numpy().reshape(B, cap); nl = outs[0]["n"].cpu().numpy()
    finally:
        exL.close(); exR.close()
    if not ok_host:
        bad += 1
        print("MISMATCH host entry, config %d (%dx%d nfeat %d scale %.2f levels %d band %d dmax %d mb %.2f mbf %.1f)" % (it, w, h, nfeat, scale, nlev, band, dmax, mb, mbf), flush=True)
    for b in range(B):
        n0, ur0, dp0 = refs[b % n_img]
        pairs_n += 1; matched += int((ur0 >= 0).sum())
        if not (nl[b] == n0 and np.array_equal(ur[b, :n0], ur0) and np.array_equal(dp[b, :n0], dp0)):
            bad += 1
            print("MISMATCH device entry, config %d (%dx%d nfeat %d scale %.2f levels %d band %d dmax %d mb %.2f mbf %.1f B %d) pair %d" %
                  (it, w, h, nfeat, scale, nlev, band, dmax, mb, mbf, B, b), flush=True)
print("stereo soak: %d configurations, %d pairs (%d stereo matches) against the oracle bit for bit, %d mismatches" % (N, pairs_n, matched, bad))
sys.exit(1 if bad else 0)
