"""BASELINE config #1 restated: no dataset, vocabulary or OpenCV is available, so "EuRoC MH01 end to end"
becomes a synthetic plumbing run of the CPU restatement -- extract -> vocabulary-node assignment -> SearchByBoW against the
previous frame -> a LocalBA window every 25 frames -- on a 200-frame sequence (frame t = frame 0 translated by t pixels),
asserting determinism and non-empty, sane outputs.  The GPU twin of this pipeline is tests/test_pipeline_gpu.py."""
import numpy as np


def shifted(base, t):
    """frame t = the base image translated by t px in x (wrap-around keeps the statistics)."""
    return np.ascontiguousarray(np.roll(base, t, axis=1))


def run_sequence(extract, bow, lba, synth, n_frames, lba_every):
    tree = synth.make_tree(0)
    base = synth.make_frame(0)
    prev = None
    log = []
    for t in range(n_frames):
        mono, kps, desc = extract(shifted(base, t))
        fv = synth.feature_vector(synth.assign_nodes(desc, tree))
        ang = np.ascontiguousarray(kps["angle"])
        entry = dict(n=len(kps), mono=mono, sig=int(desc.astype(np.uint64).sum()) ^ int(kps["x"].astype(np.float64).sum() * 1000))
        if prev is not None:
            valid = np.ones(len(prev["desc"]), np.uint8)
            nm, match = bow(prev["desc"], valid, prev["ang"], prev["fv"], desc, ang, fv)
            entry["matches"] = nm
            # matched keypoints of a 1-px shift are displaced by about one pixel
            f = np.nonzero(match >= 0)[0]
            if len(f):
                dx = kps["x"][f] - prev["kps"]["x"][match[f]]
                entry["median_dx"] = float(np.median(dx))
        if lba is not None and t % lba_every == lba_every - 1:
            w = synth.make_ba_window(t, n_opt=6, n_fixed=2, n_points=80, obs_per_point=4)
            r = lba(w)
            entry["lba_iters"] = r["stats"]["iterations"]
            entry["lba_chi2"] = (r["stats"]["chi2_initial"], r["stats"]["chi2_final"])
        prev = dict(kps=kps, desc=desc, ang=ang, fv=fv)
        log.append(entry)
    return log


def test_cpu_plumbing_200_frames(oracle, synth):
    ex = oracle.extractor()

    def extract(img):
        return ex.extract(img, (0, 1000))

    def bow(dKF, valid, angKF, fvKF, dF, angF, fvF):
        return oracle.search_by_bow(dKF, valid, angKF, fvKF, dF, angF, fvF, 0.7, True)

    def lba(w):
        return oracle.lba_solve(w, 10)
    log = run_sequence(extract, bow, lba, synth, 200, 25)
    assert len(log) == 200
    assert all(900 <= e["n"] <= 1024 and e["mono"] == 0 for e in log)
    m = [e["matches"] for e in log[1:]]
    assert min(m) > 150                                   # consecutive frames of a 1-px shift match well
    dxs = [e["median_dx"] for e in log[1:] if "median_dx" in e]
    assert abs(np.median(dxs) - 1.0) < 0.6                # and the matches are geometrically right
    lbas = [e for e in log if "lba_iters" in e]
    assert len(lbas) == 8 and all(e["lba_iters"] >= 2 and e["lba_chi2"][1] < e["lba_chi2"][0] for e in lbas)
    # determinism: a second run gives the same signatures
    log2 = run_sequence(extract, bow, None, synth, 12, 25)
    assert [e["sig"] for e in log2] == [e["sig"] for e in log[:12]]
    assert [e.get("matches") for e in log2] == [e.get("matches") for e in log[:12]]
