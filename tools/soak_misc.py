#!/usr/bin/env python3
"""Soak of the smaller entry points (GPU box; not part of the test suite), each against the CPU oracle on random sizes:
edge-SLAM packets (pack / unpack, ragged counts, IMU blocks), map-point upkeep (ComputeDistinctiveDescriptors, UpdateNormalAndDepth),
the vocabulary transform's containers on random trees, and mixed batches of the per-frame inertial optimisation (both variants in ONE
call is not allowed: a batch per variant).  usage: soak_misc.py [n]"""
import importlib
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from oracle_api import (Oracle, oracle_distinctive, oracle_normal_and_depth, oracle_pack_packet, oracle_pose_inertial_optimize, oracle_transform,  # noqa: E402
                        oracle_unpack_packet)
import test_edge_packet_gpu as TP  # noqa: E402
import test_pose_inertial_gpu as TI  # noqa: E402
from test_map_point_oracle import make_geometry, make_points  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
o = Oracle()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(606)
bad = 0; cases = 0


def guard(name, fn):
    global bad, cases
    cases += 1
    try:
        fn()
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL %s: %s | %s" % (name, type(e).__name__, " / ".join(traceback.format_exc().strip().splitlines()[-3:])[:500]), flush=True)


codec = pkg.PacketCodec(); m = pkg.Matcher(); isol = pkg.InertialSolver()
last = float(np.float32(1.2) ** 7)
for it in range(N):
    def packets():
        B = int(rs.choice([1, 2, 7, 64, 130])); cap = int(rs.choice([1, 17, 300, 1200, 1790])); max_imu = int(rs.choice([0, 1, 20, 60]))
        kps, desc, n, fid, ts, imu, off = TP._batch(pkg, 40000 + it, B, cap, max_imu)
        pay, ln, head, st = codec.pack_batch(kps, desc, n, fid, ts, imu, off)
        size = 16 + 36 * n.astype(np.int64) + 32 * np.diff(off).astype(np.int64)
        big = size > 65536                      # getHead() cannot express such a packet: the frame reports ORBX_ERR_ARG (-3) and nothing is written
        assert ((st == 0) == ~big).all() and (st[big] == -3).all()
        for b in np.nonzero(~big)[0]:
            ref, rhead = oracle_pack_packet(o, fid[b], ts[b], kps[b, :n[b]], desc[b, :n[b]], imu[off[b]:off[b + 1]])
            assert ln[b] == len(ref) and np.array_equal(pay[b, :ln[b]], ref) and np.array_equal(head[b], rhead) and not pay[b, ln[b]:].any(), b
        if big.any():
            return
        out = codec.unpack_batch(pay, ln, cap=cap, imu_cap=max(max_imu, 1))
        assert (out["status"] == 0).all() and np.array_equal(out["n"], n) and np.array_equal(out["frame_id"], fid) and np.array_equal(out["timestamp"], ts)
        for b in range(0, B, max(B // 8, 1)):
            r, f2, t2, k2, d2, i2 = oracle_unpack_packet(o, pay[b, :ln[b]])
            assert r == 0 and out["kps"][b, :n[b]].tobytes() == k2.tobytes() and np.array_equal(out["desc"][b, :n[b]], d2)
            assert out["n_imu"][b] == len(i2) and out["imu"][b, :len(i2)].tobytes() == i2.tobytes()
    guard("packets %d" % it, packets)

    def distinctive():
        P = int(rs.choice([1, 5, 300, 2000])); mo = int(rs.choice([1, 3, 15, 70, 400]))
        desc, off = make_points(50000 + it, P, mo)
        bi, bm = m.DistinctiveDescriptors(desc, off)
        obi, obm = oracle_distinctive(o, desc, off)
        assert np.array_equal(bi, obi) and np.array_equal(bm, obm)
    guard("distinctive %d" % it, distinctive)

    def normal_depth():
        P = int(rs.choice([1, 4, 300, 3000])); mo = int(rs.choice([1, 2, 15, 120]))
        pos, centers, off, ref, ls = make_geometry(51000 + it, P, mo)
        nrm, mx, mn = m.UpdateNormalAndDepth(pos, centers, off, ref, ls, last)
        onrm, omx, omn = oracle_normal_and_depth(o, pos, centers, off, ref, ls, last)
        assert nrm.tobytes() == onrm.tobytes() and mx.tobytes() == omx.tobytes() and mn.tobytes() == omn.tobytes()
    guard("normal / depth %d" % it, normal_depth)

    def vocab():
        k = int(rs.choice([2, 5, 10, 13])); L = int(rs.choice([1, 2, 3, 5])); L = min(L, 4 if k > 6 else 5)
        voc = synth.make_vocabulary(52000 + it, k=k, L=L, ragged=bool(rs.randint(0, 2)), tie_frac=float(rs.choice([0.0, 0.05, 0.3])), stop_frac=float(rs.choice([0.0, 0.08])),
                                    shuffle_ids=bool(rs.randint(0, 2)))
        n = int(rs.choice([0, 1, 2, 333, 1000, 4097]))
        rs2 = np.random.RandomState(53000 + it)
        src = voc["desc"][rs2.randint(0, voc["n_nodes"], max(n, 1))][:n]
        flips = (rs2.uniform(size=(n, 256)) < 0.03).astype(np.uint8)
        desc = np.packbits(np.unpackbits(src, axis=1) ^ flips, axis=1) if n else np.zeros((0, 32), np.uint8)
        levelsup = int(rs.randint(0, L + 2))
        (bi0, bv0), (fn0, fo0, ff0) = oracle_transform(o, voc, desc, levelsup)
        v = pkg.Vocabulary(voc)
        try:
            (bi1, bv1), (fn1, fo1, ff1) = v.transform(desc, levelsup)
        finally:
            v.close()
        assert np.array_equal(bi1, bi0) and np.array_equal(bv1, bv0) and np.array_equal(fn1, fn0) and np.array_equal(fo1, fo0) and np.array_equal(ff1, ff0)
    guard("vocabulary %d" % it, vocab)

    def pose_inertial():
        lf = bool(rs.randint(0, 2))
        nb = int(rs.choice([1, 3, 9, 40]))
        probs = [synth.make_pose_inertial_problem(54000 + 50 * it + j, n=int(rs.choice([0, 3, 9, 10, 100, 511, 512, 513, 900])), outlier_frac=float(rs.choice([0.0, 0.1, 0.3])),
                                                  stereo_frac=float(rs.choice([0.0, 0.5, 1.0])), noise_px=float(rs.choice([0.3, 1.0])), last_frame=lf)[0] for j in range(nb)]
        res = isol.pose_optimize_batch(probs)
        for j in range(0, nb, max(nb // 6, 1)):
            TI._check(oracle_pose_inertial_optimize(o, probs[j]), res[j], probs[j], (it, j, lf))
    guard("pose-inertial batch %d" % it, pose_inertial)
codec.close(); m.close(); isol.close()
print("misc soak: %d cases (packets, distinctive descriptors, normal / depth, vocabulary containers, pose-inertial batches), %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
