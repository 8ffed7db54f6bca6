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
from oracle_api import Oracle, build_oracle, oracle_pose_optimize  # noqa: E402

synth = importlib.import_module("orb_slam3-1_amd.synth")
sm = importlib.import_module("orb_slam3-1_amd.synth_match")
OUT = os.path.join(ROOT, "tests", "golden")


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
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
