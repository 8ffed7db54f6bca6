#!/usr/bin/env python3
"""Soak of pose_optimize_batch_device (GPU box; not part of the test suite): random batches -- frame counts, feature counts 0 .. cap,
match counts 0 .. features, map-point table sizes, stereo shares, out-of-range and negative assignment values -- against the host entry
pose_optimize_batch on the edges gathered on the host (must be bit-identical: same kernel, same edge order) and, for a sample, against
the CPU oracle.  usage: soak_pose_device.py [n_batches]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oracle_api import Oracle, oracle_pose_optimize  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
o = Oracle()
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(4242)
isig = (1.0 / (1.2 ** (2 * np.arange(8)))).astype(np.float32)
bad = 0; frames = 0; edges = 0; oracle_checked = 0
s = pkg.PoseSolver()
to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for it in range(N):
    B = int(rs.choice([1, 2, 7, 33, 64]))
    cap = int(rs.choice([16, 300, 511, 512, 513, 1100, 2100]))
    mp_cap = int(rs.choice([1, 50, cap, 3 * cap]))
    stereo = bool(rs.randint(0, 2))
    kps = np.zeros((B, cap), pkg.KP_DTYPE); ur = np.full((B, cap), -1.0, np.float32); assign = np.full((B, cap), -1, np.int32)
    mp = rs.normal(0, 5, (B, mp_cap, 3)).astype(np.float32); nk = np.zeros(B, np.int32); pose = np.zeros((B, 7))
    ws = []
    for b in range(B):
        n_feat = int(rs.choice([0, 1, cap // 2, cap, rs.randint(0, cap + 1)]))
        n = int(min(rs.choice([0, 2, 3, 9, 10, 100, n_feat]), n_feat, mp_cap))
        w = synth.make_pose_problem(10000 + 100 * it + b, n=n, outlier_frac=float(rs.choice([0.0, 0.1, 0.4])), stereo_frac=float(rs.choice([0.0, 0.5, 1.0])) if stereo else 0.0)
        nk[b] = n_feat
        kps[b]["x"] = rs.uniform(0, 640, cap); kps[b]["y"] = rs.uniform(0, 480, cap); kps[b]["octave"] = rs.randint(0, 8, cap)
        feat = np.sort(rs.choice(n_feat, n, replace=False)) if n else np.zeros(0, np.int64)
        rows = rs.choice(mp_cap, n, replace=False) if n else np.zeros(0, np.int64)
        if n:
            octv = np.round(np.log(1.0 / w["inv_sigma2"]) / (2 * np.log(1.2))).astype(np.int32)
            kps[b]["x"][feat] = w["obs"][:, 0]; kps[b]["y"][feat] = w["obs"][:, 1]; kps[b]["octave"][feat] = octv
            ur[b, feat] = np.where(w["stereo"] != 0, w["obs"][:, 2], -1.0)
            assign[b, feat] = rows
            mp[b, rows] = w["Xw"]
        # noise the entry must ignore: rows beyond the frame's count, out-of-range and other negative values on features without a point
        assign[b, n_feat:] = rs.randint(-3, mp_cap + 5, cap - n_feat)
        free = np.setdiff1d(np.arange(n_feat), feat)
        if len(free):
            pick = free[rs.uniform(size=len(free)) < 0.1]
            assign[b, pick] = rs.choice([-2, -100, mp_cap, mp_cap + 7], len(pick))
        pose[b, :4] = w["q"]; pose[b, 4:] = w["t"]
        w2 = dict(w)
        w2["obs"] = np.stack([kps[b]["x"][feat], kps[b]["y"][feat], ur[b, feat] if stereo else np.full(n, -1.0, np.float32)], 1).astype(np.float64).reshape(-1, 3)
        w2["Xw"] = mp[b, rows].astype(np.float64).reshape(-1, 3)
        w2["inv_sigma2"] = isig[kps[b]["octave"][feat]].astype(np.float64)
        w2["stereo"] = ((ur[b, feat] >= 0) & stereo).astype(np.uint8)
        w2["feat"] = feat
        ws.append(w2)
    d_kps = to(kps.view(np.uint8)); d_ur = to(ur); d_as = to(assign); d_mp = to(mp); d_nk = to(nk); d_pose = to(pose)
    d_out = torch.zeros(B, 7, dtype=torch.float64, device=dev); d_inl = torch.full((B,), -5, dtype=torch.int32, device=dev)
    d_outl = torch.full((B, cap), 9, dtype=torch.uint8, device=dev)
    s.optimize_batch_device(B, cap, d_kps.data_ptr(), d_nk.data_ptr(), d_as.data_ptr(), d_mp.data_ptr(), mp_cap, d_pose.data_ptr(), isig, ws[0],
                            d_out.data_ptr(), d_inl.data_ptr(), d_outl.data_ptr(), torch.cuda.current_stream().cuda_stream, d_u_right=d_ur.data_ptr() if stereo else None)
    torch.cuda.synchronize()
    host = s.optimize_batch(ws)
    out = d_out.cpu().numpy(); inl = d_inl.cpu().numpy(); outl = d_outl.cpu().numpy()
    for b, w in enumerate(ws):
        h = host[b]
        full = np.zeros(cap, np.uint8); full[w["feat"]] = h["outlier"]
        ok = np.array_equal(out[b, :4], h["q"]) and np.array_equal(out[b, 4:], h["t"]) and inl[b] == h["inliers"] and np.array_equal(outl[b], full)
        if ok and b == 0 and len(w["feat"]) >= 10:       # a sample against the oracle as well
            g = oracle_pose_optimize(o, w)
            dt = np.abs(np.asarray(g["t"]) - w["t"]).max()
            ok = np.array_equal(h["outlier"], g["outlier"]) and np.abs(out[b, 4:] - g["t"]).max() <= 1e-4 * dt + 1e-12
            oracle_checked += 1
        frames += 1; edges += len(w["feat"])
        if not ok:
            bad += 1
            print("MISMATCH batch %d (B %d cap %d mp_cap %d stereo %d) frame %d: %d features, %d edges" % (it, B, cap, mp_cap, stereo, b, nk[b], len(w["feat"])), flush=True)
s.close()
print("pose device-entry soak: %d batches, %d frames, %d edges: %d mismatches vs the host entry (bit for bit); %d frames also against the oracle" % (N, frames, edges, bad, oracle_checked))
sys.exit(1 if bad else 0)
