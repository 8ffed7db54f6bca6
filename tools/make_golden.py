#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded end-to-end outputs of the CPU oracle.
The reference has no fixtures of its own and cannot be built here, so these freeze the oracle's behaviour; the HIP
path is compared against them on the GPU box (tests/test_golden_gpu.py) without needing the oracle to agree by luck."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_api import (IMU_DTYPE, KP_DTYPE, Oracle, build_oracle, oracle_inertial_solve, oracle_pack_packet, oracle_pose_optimize,  # noqa: E402
                        oracle_stereo_matches, oracle_transform, oracle_unpack_packet)

synth = importlib.import_module("orb_slam3-1_amd.synth")
sm = importlib.import_module("orb_slam3-1_amd.synth_match")
OUT = os.environ.get("ORB_GOLDEN_OUT", os.path.join(ROOT, "tests", "golden"))    # set it to regenerate into a scratch directory


def main():
    build_oracle()
    o = Oracle()
    os.makedirs(OUT, exist_ok=True)
    for name, args, img in (("extractor_160x120", (300, 1.2, 4, 20, 7), synth.make_frame(5, 160, 120)),
                            ("extractor_640x480", (1000, 1.2, 8, 20, 7), synth.make_frame(0))):
        r, kps, desc = o.extractor(*args).extract(img, (0, 1000))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), mono=r, desc=desc, **{"kp_" + f: kps[f] for f in kps.dtype.names})
    ms = synth.make_match_set(3, n=64)
    n, m = o.search_by_bow(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"], 0.7, True)
    np.savez_compressed(os.path.join(OUT, "bow_64.npz"), n=n, match=m)
    w = synth.make_ba_window(0, n_opt=5, n_fixed=2, n_points=60, obs_per_point=4)
    r = o.lba_solve(w, 10)
    np.savez_compressed(os.path.join(OUT, "lba_5kf_60mp.npz"), iterations=r["stats"]["iterations"], trials=r["stats"]["trials"],
                        points=r["points"], pose_t=r["pose_t"], pose_q=r["pose_q"], chi2=r["chi2"])
    gr, dF, angF, scale, mp, assign, occ = sm.make_projection_case(1, n=300, n_mp=250)
    n = o.search_by_projection(gr, dF, scale, mp, 3.0, 0.8, assign, occ)
    np.savez_compressed(os.path.join(OUT, "proj_300.npz"), n=n, assign=assign, occupied=occ)
    for name, kw in (("pose_mono_300", dict(seed=2, n=300, outlier_frac=0.1, stereo_frac=0.0)),
                     ("pose_stereo_200", dict(seed=3, n=200, outlier_frac=0.15, stereo_frac=0.5))):
        r = oracle_pose_optimize(o, synth.make_pose_problem(**kw))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), q=r["q"], t=r["t"], outlier=r["outlier"], n_bad=r["n_bad"], inliers=r["inliers"])
    voc = synth.make_vocabulary(40, k=6, L=3)
    rs = np.random.RandomState(40)
    vd = np.ascontiguousarray(voc["desc"][rs.randint(1, voc["n_nodes"], 300)] ^ (rs.uniform(size=(300, 32)) < 0.03).astype(np.uint8))
    (bi, bv), (fn, fo, ff) = oracle_transform(o, voc, vd, 2)
    np.savez_compressed(os.path.join(OUT, "vocab_k6_L3_300.npz"), bow_id=bi, bow_val=bv, fv_node=fn, fv_off=fo, fv_feat=ff)
    g, dKF, scale, u_right, inv_s2, pts = sm.make_fuse_case(41, n=600, n_pts=500)
    fbi, fbd = o.fuse_search(g, dKF, scale, u_right, inv_s2, pts, 3.0, True)
    np.savez_compressed(os.path.join(OUT, "fuse_500.npz"), best_idx=fbi, best_dist=fbd)
    k1, k2, ep, F12, sigma2, sc2 = sm.make_triangulation_case(42, n=600)
    tn, tm = o.search_for_triangulation(k1, k2, ep, F12, sigma2, sc2, False, False, True)
    np.savez_compressed(os.path.join(OUT, "triangulation_600.npz"), n=tn, match12=tm)
    left, right = synth.make_stereo_pair(43)
    eL, eR = o.extractor(), o.extractor()
    _, kL, dL = eL.extract(left, (0, 0)); _, kR, dR = eR.extract(right, (0, 0))
    sn, ur, dp = oracle_stereo_matches(eL, eR, kL, dL, kR, dR, 0.11, 47.9)
    np.savez_compressed(os.path.join(OUT, "stereo_pair_43.npz"), n=sn, u_right=ur, depth=dp)
    rs = np.random.RandomState(44)
    pk = np.zeros(400, KP_DTYPE); pk["x"] = rs.uniform(0, 752, 400).astype(np.float32); pk["y"] = rs.uniform(0, 480, 400).astype(np.float32)
    pd = rs.randint(0, 256, (400, 32)).astype(np.uint8)
    pi = np.zeros(10, IMU_DTYPE); pi["ts"] = 1403636579763555584 + np.arange(10) * 5000000
    pi["gyro"] = rs.normal(0, 0.2, (10, 3)); pi["acce"] = rs.normal(0, 9.8, (10, 3))
    pay, head = oracle_pack_packet(o, 4711, 1403636579813555456, pk, pd, pi)
    _, _, _, uk, _, _ = oracle_unpack_packet(o, pay)
    np.savez_compressed(os.path.join(OUT, "edge_packet_0.npz"), frame_id=4711, timestamp=1403636579813555456, kps_x=pk["x"], kps_y=pk["y"], desc=pd,
                        imu_ts=pi["ts"], imu_gyro=pi["gyro"], imu_acce=pi["acce"], payload=pay, head=head, unpacked_x=uk["x"], unpacked_y=uk["y"])
    iw, _ = synth.make_inertial_window(45, n_opt=5, n_points=120, obs_per_point=4, stereo_frac=0.3, n_covisible_fixed=2)
    ir = oracle_inertial_solve(o, iw)
    np.savez_compressed(os.path.join(OUT, "inertial_5kf_120mp.npz"), iterations=ir["stats"]["iterations"], trials=ir["stats"]["trials"],
                        chi2_final=ir["stats"]["chi2_final"], Rwb=ir["Rwb"], twb=ir["twb"], vel=ir["vel"], bg=ir["bg"], ba=ir["ba"], points=ir["points"])
    # the rectified-stereo branches of the two tracking searches (src/ORBmatcher.cc:92-98, :1692-1693, :1728-1733, :1751-1757)
    gr, dF, angF, scale, mp, assign, occ = sm.make_projection_case(46, n=300, n_mp=250, stereo_frac=0.5)
    n = o.search_by_projection(gr, dF, scale, mp, 3.0, 0.8, assign, occ)
    np.savez_compressed(os.path.join(OUT, "proj_stereo_300.npz"), n=n, assign=assign, occupied=occ)
    for lw, name in ((1, "forward"), (2, "backward")):
        gr, dF, angF, scale, last, assign, occ = sm.make_last_frame_case(47, n=300, n_last=250, stereo_frac=0.5, level_window=lw)
        n = o.search_by_projection_last(gr, dF, angF, scale, last, 15.0, True, assign, occ)
        np.savez_compressed(os.path.join(OUT, "proj_last_stereo_%s_300.npz" % name), n=n, assign=assign, occupied=occ)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
