"""HIP PoseOptimization (pose_solver.hip, one workgroup per frame) against the CPU restatement of Optimizer::PoseOptimization
(reference src/Optimizer.cc:814-1115) through the C ABI.  Tolerance: BASELINE's 1e-4 relative on the pose UPDATE; the
inlier / outlier classification (float chi2 against 5.991 / 7.815) must be identical."""
import os

import numpy as np
import pytest

from oracle_api import oracle_pose_optimize

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(w, r, g):
    np.testing.assert_array_equal(r["outlier"], g["outlier"])
    assert (r["n_bad"], r["inliers"]) == (int(g["n_bad"]), int(g["inliers"]))
    # (Levenberg trial counts are NOT compared: at convergence the sign of rho is rounding noise on both sides)
    q0 = np.asarray(w["q"]) / np.linalg.norm(w["q"])
    dq = np.abs(np.asarray(g["q"]) - q0).max(); dt = np.abs(np.asarray(g["t"]) - w["t"]).max()
    assert np.abs(r["q"] - g["q"]).max() <= 1e-4 * dq + 1e-12
    assert np.abs(r["t"] - g["t"]).max() <= 1e-4 * dt + 1e-12


@pytest.mark.parametrize("seed,n,of,sf", [(0, 300, 0.1, 0.0), (1, 300, 0.1, 0.4), (2, 1000, 0.2, 1.0), (3, 50, 0.0, 0.0),
                                           (4, 9, 0.0, 0.0), (5, 2, 0.0, 0.0), (6, 0, 0.0, 0.0), (7, 3000, 0.3, 0.5),
                                           (8, 257, 0.05, 0.2), (9, 12, 0.3, 0.0)])
def test_pose_equals_oracle(pkg, oracle, synth, seed, n, of, sf):
    w = synth.make_pose_problem(seed, n=n, outlier_frac=of, stereo_frac=sf)
    s = pkg.PoseSolver()
    try:
        r = s.optimize(w)
    finally:
        s.close()
    _check(w, r, oracle_pose_optimize(oracle, w))


def test_pose_batch_equals_single(pkg, oracle, synth):
    """a batch is one launch with one workgroup per frame: results equal the per-frame oracle runs, in order"""
    ws = [synth.make_pose_problem(20 + i, n=100 + 37 * i, outlier_frac=0.1, stereo_frac=0.25 * (i % 3)) for i in range(24)]
    s = pkg.PoseSolver()
    try:
        rs = s.optimize_batch(ws)
        rs2 = s.optimize_batch(ws[::-1])[::-1]          # reuse of the solver's device blob
    finally:
        s.close()
    for w, r, r2 in zip(ws, rs, rs2):
        _check(w, r, oracle_pose_optimize(oracle, w))
        np.testing.assert_array_equal(r["q"], r2["q"]); np.testing.assert_array_equal(r["outlier"], r2["outlier"])


def test_pose_golden(pkg, synth):
    s = pkg.PoseSolver()
    try:
        for name, kw in (("pose_mono_300", dict(seed=2, n=300, outlier_frac=0.1, stereo_frac=0.0)),
                         ("pose_stereo_200", dict(seed=3, n=200, outlier_frac=0.15, stereo_frac=0.5))):
            w = synth.make_pose_problem(**kw)
            _check(w, s.optimize(w), np.load(os.path.join(GOLDEN, name + ".npz")))
    finally:
        s.close()


def test_pose_is_invariant_to_edge_order(pkg, synth):
    """a property that needs no oracle: permuting the edges permutes the outlier flags and leaves the pose alone (the block
    reductions are ordered, but the sum is the same set of terms)"""
    w = synth.make_pose_problem(40, n=700, outlier_frac=0.15, stereo_frac=0.3)
    rs = np.random.RandomState(1)
    p = rs.permutation(700)
    w2 = dict(w)
    for k in ("Xw", "obs", "inv_sigma2", "stereo"):
        w2[k] = np.ascontiguousarray(w[k][p])
    s = pkg.PoseSolver()
    try:
        r1, r2 = s.optimize_batch([w, w2])
    finally:
        s.close()
    np.testing.assert_array_equal(r1["outlier"][p], r2["outlier"])
    assert r1["inliers"] == r2["inliers"]
    np.testing.assert_allclose(r2["q"], r1["q"], rtol=0, atol=1e-9); np.testing.assert_allclose(r2["t"], r1["t"], rtol=0, atol=1e-9)
    Rerr = np.abs(r1["t"] - w["true_t"]).max()
    assert Rerr < 0.05


def test_pose_device_entry_equals_host_entry_and_oracle(pkg, oracle, synth):
    """pose_optimize_batch_device gathers a frame's edges ON THE DEVICE from extractor-layout arrays (key-point records, mvuRight,
    the assignment the projection search wrote, the map points' float positions) in feature order; the result must be the host
    entry's on the same edges (bit for bit: same kernel, same edge order) and the oracle's within the usual tolerance.
    Frames: mono and stereo mixes, a frame without matches, one with 2 matches, one whose features all hold map points."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(77)
    cap, mp_cap = 1100, 1300
    specs = [(300, 0.1, 0.0), (450, 0.1, 0.4), (0, 0.0, 0.0), (2, 0.0, 0.0), (1000, 0.2, 1.0), (9, 0.0, 0.0), (700, 0.3, 0.5), (cap, 0.1, 0.0)]
    B = len(specs)
    scale2 = (1.2 ** (2 * np.arange(8)))
    isig = (1.0 / scale2).astype(np.float32)
    kps = np.zeros((B, cap), pkg.KP_DTYPE); ur = np.full((B, cap), -1.0, np.float32); assign = np.full((B, cap), -1, np.int32)
    mp = rs.normal(0, 5, (B, mp_cap, 3)).astype(np.float32); nk = np.zeros(B, np.int32); pose = np.zeros((B, 7))
    ws = []
    for b, (n, of, sf) in enumerate(specs):
        w = synth.make_pose_problem(300 + b, n=n, outlier_frac=of, stereo_frac=sf)
        n_feat = cap if n == cap else min(cap, n + int(rs.randint(50, 300)))
        nk[b] = n_feat
        kps[b]["x"] = rs.uniform(0, 640, cap); kps[b]["y"] = rs.uniform(0, 480, cap); kps[b]["octave"] = rs.randint(0, 8, cap)
        feat = np.sort(rs.choice(n_feat, n, replace=False))              # the features that hold a map point, in feature order
        rows = rs.choice(mp_cap, n, replace=False)                        # ... and where their map points sit
        octv = np.round(np.log(1.0 / w["inv_sigma2"]) / (2 * np.log(1.2))).astype(np.int32) if n else np.zeros(0, np.int32)
        kps[b]["x"][feat] = w["obs"][:, 0]; kps[b]["y"][feat] = w["obs"][:, 1]; kps[b]["octave"][feat] = octv
        ur[b, feat] = np.where(w["stereo"] != 0, w["obs"][:, 2], -1.0)
        assign[b, feat] = rows
        assign[b, n_feat:] = 5                                           # rows beyond the frame's count are never read
        mp[b, rows] = w["Xw"]
        pose[b, :4] = w["q"]; pose[b, 4:] = w["t"]
        # what the device gathers, gathered on the host (float -> double)
        w2 = dict(w)
        w2["obs"] = np.stack([kps[b]["x"][feat], kps[b]["y"][feat], ur[b, feat]], 1).astype(np.float64).reshape(-1, 3)
        w2["Xw"] = mp[b, rows].astype(np.float64).reshape(-1, 3)
        w2["inv_sigma2"] = isig[kps[b]["octave"][feat]].astype(np.float64)
        w2["stereo"] = (ur[b, feat] >= 0).astype(np.uint8)
        w2["feat"] = feat
        ws.append(w2)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_kps = to(kps.view(np.uint8)); d_ur = to(ur); d_as = to(assign); d_mp = to(mp); d_nk = to(nk); d_pose = to(pose)
    d_out = torch.zeros(B, 7, dtype=torch.float64, device=dev); d_inl = torch.full((B,), -5, dtype=torch.int32, device=dev)
    d_outl = torch.full((B, cap), 9, dtype=torch.uint8, device=dev)
    s = pkg.PoseSolver()
    try:
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):                      # the second call reuses the arena
            s.optimize_batch_device(B, cap, d_kps.data_ptr(), d_nk.data_ptr(), d_as.data_ptr(), d_mp.data_ptr(), mp_cap, d_pose.data_ptr(), isig, ws[0],
                                    d_out.data_ptr(), d_inl.data_ptr(), d_outl.data_ptr(), st, d_u_right=d_ur.data_ptr())
        torch.cuda.synchronize()
        host = s.optimize_batch(ws)
        # monocular instantiation (d_u_right = NULL) on the mono frames
        mono = [b for b, sp in enumerate(specs) if sp[2] == 0.0]
        d_out_m = torch.zeros(B, 7, dtype=torch.float64, device=dev); d_inl_m = torch.zeros(B, dtype=torch.int32, device=dev)
        s.optimize_batch_device(B, cap, d_kps.data_ptr(), d_nk.data_ptr(), d_as.data_ptr(), d_mp.data_ptr(), mp_cap, d_pose.data_ptr(), isig, ws[0],
                                d_out_m.data_ptr(), d_inl_m.data_ptr(), None, st)
        torch.cuda.synchronize()
    finally:
        s.close()
    out = d_out.cpu().numpy(); inl = d_inl.cpu().numpy(); outl = d_outl.cpu().numpy()
    out_m = d_out_m.cpu().numpy(); inl_m = d_inl_m.cpu().numpy()
    for b, w in enumerate(ws):
        h = host[b]
        np.testing.assert_array_equal(out[b, :4], h["q"]); np.testing.assert_array_equal(out[b, 4:], h["t"])
        assert inl[b] == h["inliers"]
        full = np.zeros(cap, np.uint8); full[w["feat"]] = h["outlier"]
        np.testing.assert_array_equal(outl[b], full)
        g = oracle_pose_optimize(oracle, w)
        _check(w, dict(h, q=out[b, :4], t=out[b, 4:]), g)
        if b in mono:
            np.testing.assert_array_equal(out_m[b], out[b]); assert inl_m[b] == inl[b]
