#!/usr/bin/env python3
"""Soak of the extractor (GPU box; not part of the test suite): random configurations (image size, feature budget, pyramid depth, scale
factor, FAST thresholds, lapping area) x random BATCH sizes (1 .. 80: single-stream path, k_octree_dyn with and without side streams,
the large-batch schedule with the resize chain beside FAST) through the host-batch and the device-resident entry points, every frame
against the CPU oracle bit for bit (key points, order, angles, responses, descriptors).  usage: soak_extractor.py [n_configs] [extreme] [images]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from oracle_api import Oracle  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
o = Oracle()
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
EXTREME = len(sys.argv) > 2 and "extreme" in sys.argv[2:]
IMAGES = len(sys.argv) > 2 and "images" in sys.argv[2:]       # adversarial image statistics (noise, flat, checkerboards, stripes, salt and pepper, blocks, ramps)
rs = np.random.RandomState((778 if EXTREME else 777) + (10 if IMAGES else 0))
bad = 0; frames = 0; scheds = {}
t0 = time.time()
for ci in range(N):
    if EXTREME:         # far corners of the parameter space: tiny and large images, extreme aspect ratios, 1 .. 12 000 features, scale 1.05 .. 3
        w = int(rs.choice([rs.randint(60, 200), rs.randint(200, 1400)])); h = int(rs.choice([rs.randint(60, 200), rs.randint(200, 1000)]))
        nfeat = int(rs.choice([1, 10, 50, 100, 1000, 5000, 12000]))
        scale = float(rs.choice([1.05, 1.1, 1.2, 1.5, 2.0, 2.5, 3.0]))
        nlev = int(rs.randint(1, 13))
        while nlev > 1 and min(w, h) / scale ** (nlev - 1) < 45:
            nlev -= 1
        ini = int(rs.choice([5, 20, 40, 80])); mn = min(int(rs.choice([2, 7, 20, 80])), ini)
    else:
        w = int(rs.randint(96, 900)) & ~3 if rs.uniform() < 0.5 else int(rs.randint(96, 900))
        h = int(rs.randint(80, 620))
        nfeat = int(rs.choice([100, 300, 700, 1000, 1000, 1500, 2500]))
        scale = float(rs.choice([1.1, 1.2, 1.2, 1.25, 1.33, 1.5, 2.0]))
        nlev = int(rs.randint(1, 11))
        while nlev > 1 and min(w, h) / scale ** (nlev - 1) < 60:
            nlev -= 1
        ini = int(rs.choice([20, 20, 12, 30])); mn = min(int(rs.choice([7, 7, 5, 10])), ini)
    lap = (0, 1000) if rs.uniform() < 0.5 else ((0, 0) if rs.uniform() < 0.5 else (int(w * 0.3), int(w * 0.6)))
    B = int(rs.choice([1, 1, 2, 5, 16, 31, 32, 40, 64, 65, 80])) if w * h < 700000 else int(rs.choice([1, 2, 33, 65]))
    n_img = min(B, 6)
    kind = IMAGES and ["synth", "noise", "zeros", "checker", "stripes", "salt", "blocks", "gradient"][int(rs.randint(0, 8))] or "synth"

    def make(k_):
        r2 = np.random.RandomState(91000 + 10 * ci + k_)
        if kind == "noise":         # a corner at almost every pixel: candidate slots, corner lists and the octree at their limits
            return r2.randint(0, 256, (h, w)).astype(np.uint8)
        if kind == "zeros":
            return np.full((h, w), int(r2.randint(0, 256)), np.uint8)
        if kind == "checker":
            c = int(r2.choice([1, 2, 3, 8, 31]))
            yy, xx = np.mgrid[0:h, 0:w]
            return (((yy // c + xx // c) & 1) * int(r2.randint(40, 256))).astype(np.uint8)
        if kind == "stripes":
            c = int(r2.choice([1, 2, 5, 16])); a = np.zeros((h, w), np.uint8); a[:, (np.arange(w) // c) % 2 == 0] = 200
            return a if r2.randint(0, 2) else np.ascontiguousarray(np.where((np.arange(h)[:, None] // c) % 2 == 0, 200, 0).astype(np.uint8) + np.zeros((1, w), np.uint8))
        if kind == "salt":
            a = np.full((h, w), 100, np.uint8); m_ = r2.uniform(size=(h, w)) < float(r2.choice([0.001, 0.02, 0.2])); a[m_] = r2.choice([0, 255], int(m_.sum()))
            return a
        if kind == "blocks":
            b_ = int(r2.choice([4, 8, 16])); blk = r2.randint(0, 256, (h // b_ + 2, w // b_ + 2)).astype(np.uint8)
            return np.ascontiguousarray(np.kron(blk, np.ones((b_, b_), np.uint8))[:h, :w])
        if kind == "gradient":
            return np.ascontiguousarray(((np.arange(w)[None, :] * 255 // max(w - 1, 1)) + np.zeros((h, 1), np.int64)).astype(np.uint8))
        return synth.make_frame(3000 + 10 * ci + k_, w, h)
    imgs = np.stack([make(k) for k in range(n_img)])
    oex = o.extractor(nfeat, scale, nlev, ini, mn)
    ref = [oex.extract(im, lap) for im in imgs]
    batch = np.ascontiguousarray(imgs[np.arange(B) % n_img])
    ex = pkg.Extractor(nfeat, scale, nlev, ini, mn)
    try:
        cap = max(ex.max_keypoints, ex.max_keypoints_for(w, h))       # (very wide images with tiny budgets: the bound for THIS image size)
        results = []
        mono, n, kps, desc = ex.extract_batch(batch, lap)
        results.append(("host", mono, n, kps, desc, ex.debug_last_schedule()))
        if True:                                        # device entry: read in place when base, row stride and frame stride are 16-byte multiples, copied otherwise
            # the caller's frames with a random row stride, frame stride and base offset (in place only when all three are 16-byte multiples)
            rstride = w + int(rs.choice([0, 0, 1, 3, 16, 64])); fstride = rstride * h + int(rs.choice([0, 0, 5, 16, 256])); boff = int(rs.choice([0, 0, 1, 4, 16]))
            host_buf = rs.randint(0, 256, boff + B * fstride + 64).astype(np.uint8)         # (noise in the padding: it must never be read as pixels)
            view = np.lib.stride_tricks.as_strided(host_buf[boff:], shape=(B, h, w), strides=(fstride, rstride, 1))
            view[...] = batch
            d_buf = torch.from_numpy(host_buf).to(dev)
            d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev); d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
            d_n = torch.zeros(B, dtype=torch.int32, device=dev); d_mono = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
            for _ in range(2):
                ex.extract_batch_device(d_buf.data_ptr() + boff, B, w, h, rstride, fstride, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(), d_mono.data_ptr(), d_st.data_ptr(),
                                        lap, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            if int(d_st.abs().sum().item()) != 0:
                raise pkg.OrbxError(int(d_st.min().item()), "device entry: per-frame status %r" % d_st.cpu().numpy()[:4])
            results.append(("device", d_mono.cpu().numpy(), d_n.cpu().numpy(), d_kps.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, cap),
                            d_desc.cpu().numpy().reshape(B, cap, 32), ex.debug_last_schedule()))
        for name, mono, n, kps, desc, sched in results:
            scheds[sched] = scheds.get(sched, 0) + 1
            for b in range(B):
                r0, k0, d0 = ref[b % n_img]
                frames += 1
                ok = mono[b] == r0 and n[b] == len(k0) and all(np.array_equal(kps[b, :n[b]][f], k0[f]) for f in k0.dtype.names) and np.array_equal(desc[b, :n[b]], d0)
                if not ok:
                    bad += 1
                    print("MISMATCH config %d (%s %dx%d nfeat %d scale %.2f levels %d th %d/%d lap %s B %d) entry %s frame %d schedule %d" %
                          (ci, kind, w, h, nfeat, scale, nlev, ini, mn, lap, B, name, b, sched), flush=True)
    except pkg.OrbxError as e_:
        bad += 1
        print("ERROR config %d (%s %dx%d nfeat %d scale %.2f levels %d th %d/%d lap %s B %d): %s" % (ci, kind, w, h, nfeat, scale, nlev, ini, mn, lap, B, e_), flush=True)
    finally:
        ex.close()
    if ci % 10 == 9:
        print("... %d configurations, %d frames, %d mismatches, %.0f s" % (ci + 1, frames, bad, time.time() - t0), flush=True)
print("extractor soak: %d configurations, %d frames compared with the oracle bit for bit, %d mismatches; schedules seen (bits of orbx_debug_last_schedule: count) %s" %
      (N, frames, bad, dict(sorted(scheds.items()))))
sys.exit(1 if bad else 0)
