"""The tracking front end as ONE device-resident chain -- orbx_extract_batch_device -> orbv_transform_batch_device ->
SearchByBoW on a device-resident plan -- with no host round trip between the three, against the oracle chain
(ORBextractor::operator() -> TemplatedVocabulary::transform -> ORBmatcher::SearchByBoW) on the same images."""
import numpy as np
import pytest

from oracle_api import oracle_transform

pytestmark = pytest.mark.gpu


def run_front_end_chain(pkg, oracle, synth, P=3, seed0=60, size=(640, 480), voc_kw=None, levelsup=1, min_matches=100):
    """the chain on P pairs (frame, the same frame moved by a few pixels) of `size` images; also driven by tools/soak_chain.py"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    W_, H_ = size
    imgs = []
    for i in range(P):
        base = synth.make_frame(seed0 + i, W_, H_)
        imgs += [base, np.ascontiguousarray(np.roll(base, 3 + i % 5, axis=1))]
    imgs = np.stack(imgs)
    B = 2 * P
    voc = synth.make_vocabulary(5, **(voc_kw or dict(k=10, L=3, ragged=False, tie_frac=0.0, stop_frac=0.0)))
    ex, m, v = pkg.Extractor(), pkg.Matcher(0.7, True), pkg.Vocabulary(voc)
    cap = ex.max_keypoints
    try:
        d_img = torch.from_numpy(imgs.copy()).to(dev)
        d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev); d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev); d_mono = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        z = lambda dt, k: torch.zeros(B * k, dtype=dt, device=dev)
        d_bi, d_bv, d_nb = z(torch.int32, cap), z(torch.float64, cap), torch.zeros(B, dtype=torch.int32, device=dev)
        d_fn, d_fo, d_ff, d_nf = z(torch.int32, cap), z(torch.int32, cap + 1), z(torch.int32, cap), torch.zeros(B, dtype=torch.int32, device=dev)
        d_match = torch.full((P * cap,), -7, dtype=torch.int32, device=dev); d_nm = torch.zeros(P, dtype=torch.int32, device=dev)

        def side(b):
            return dict(desc=d_desc.data_ptr() + b * cap * 32, kps=d_kps.data_ptr() + b * cap * 28, n=d_n.data_ptr() + 4 * b, cap=cap,
                        fv_node=d_fn.data_ptr() + 4 * b * cap, fv_off=d_fo.data_ptr() + 4 * b * (cap + 1), fv_feat=d_ff.data_ptr() + 4 * b * cap,
                        n_fv_nodes=d_nf.data_ptr() + 4 * b)
        plan = pkg.DeviceBowPlan(m, [(side(2 * p), side(2 * p + 1), d_match.data_ptr() + 4 * p * cap, d_nm.data_ptr() + 4 * p) for p in range(P)])
        st = torch.cuda.current_stream().cuda_stream
        # the whole chain is enqueued before anything is read back
        ex.extract_batch_device(d_img.data_ptr(), B, W_, H_, W_, W_ * H_, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                d_n.data_ptr(), d_mono.data_ptr(), d_st.data_ptr(), (0, 1000), st)
        v.transform_batch_device(d_desc.data_ptr(), d_n.data_ptr(), B, cap, levelsup, d_bi.data_ptr(), d_bv.data_ptr(), d_nb.data_ptr(),
                                 d_fn.data_ptr(), d_fo.data_ptr(), d_ff.data_ptr(), d_nf.data_ptr(), st)
        plan.run(st)
        torch.cuda.synchronize()
        match = d_match.cpu().numpy().reshape(P, cap); nm = d_nm.cpu().numpy(); n = d_n.cpu().numpy()
        plan.close()
    finally:
        ex.close(); m.close(); v.close()
    oex = oracle.extractor()
    for p in range(P):
        _, kK, dK = oex.extract(imgs[2 * p], (0, 1000)); _, kF, dF = oex.extract(imgs[2 * p + 1], (0, 1000))
        assert n[2 * p] == len(kK) and n[2 * p + 1] == len(kF)
        (_, _), fvK = oracle_transform(oracle, voc, dK, levelsup)
        (_, _), fvF = oracle_transform(oracle, voc, dF, levelsup)
        n0, m0 = oracle.search_by_bow(dK, np.ones(len(kK), np.uint8), np.ascontiguousarray(kK["angle"]), fvK,
                                      dF, np.ascontiguousarray(kF["angle"]), fvF, 0.7, True)
        assert nm[p] == n0 and n0 > min_matches
        np.testing.assert_array_equal(match[p, :len(kF)], m0)


def test_extract_transform_bow_chain(pkg, oracle, synth):
    run_front_end_chain(pkg, oracle, synth)


def test_tracking_chain_extract_project_pose(pkg, oracle, synth):
    run_tracking_chain(pkg, oracle, synth)


def run_tracking_chain(pkg, oracle, synth, B=4, seed0=80, shift=3, size=(640, 480), min_matches=300, min_inliers=200):
    """TrackWithMotionModel's device work as ONE device-resident chain (src/Tracking.cc:2975-3053): orbx_extract_batch_device (current
    frames) -> orbm_search_by_projection_last_batch_device against resident last frames -> pose_optimize_batch_device (edges gathered
    on the device from the search's assignment), nothing read back in between -- against the oracle chain ORBextractor::operator() ->
    SearchByProjection(CurrentFrame, LastFrame) -> Optimizer::PoseOptimization on the same images.  The last frames are the same
    scenes 3 px to the left; their map points sit at random depths behind their features, consistent with a current pose = identity."""
    torch = pytest.importorskip("torch")
    from oracle_api import oracle_pose_optimize
    dev = torch.device("cuda", 0)
    W_, H_ = size
    last_imgs = np.stack([synth.make_frame(seed0 + i, W_, H_) for i in range(B)])
    cur_imgs = np.ascontiguousarray(np.roll(last_imgs, shift, axis=2))
    cam = dict(fx=float(np.float32(458.654)), fy=float(np.float32(457.296)), cx=float(np.float32(367.215)), cy=float(np.float32(248.375)), bf=0.0,
               huber_mono=float(np.float32(np.sqrt(5.991))), huber_stereo=float(np.float32(np.sqrt(7.815))))
    rs = np.random.RandomState(5)
    ex, m, ps = pkg.Extractor(), pkg.Matcher(0.9, True), pkg.PoseSolver()
    cap = ex.max_keypoints
    sfac = np.asarray(ex.GetScaleFactors(), np.float32)
    isig = (1.0 / sfac.astype(np.float64) ** 2).astype(np.float32)
    oex = oracle.extractor()
    try:
        st = torch.cuda.current_stream().cuda_stream
        z = lambda dt, k: torch.zeros(B * k, dtype=dt, device=dev)
        l_kps, l_desc, l_n = z(torch.uint8, cap * 28), z(torch.uint8, cap * 32), z(torch.int32, 1)
        c_kps, c_desc, c_n = z(torch.uint8, cap * 28), z(torch.uint8, cap * 32), z(torch.int32, 1)
        d_mono, d_st = z(torch.int32, 1), z(torch.int32, 1)
        d_last = torch.from_numpy(last_imgs.copy()).to(dev); d_cur = torch.from_numpy(cur_imgs.copy()).to(dev)
        ex.extract_batch_device(d_last.data_ptr(), B, W_, H_, W_, W_ * H_, l_kps.data_ptr(), l_desc.data_ptr(), cap, l_n.data_ptr(), d_mono.data_ptr(), d_st.data_ptr(), (0, 1000), st)
        torch.cuda.synchronize()
        # the last frames' per-feature arrays (resident in steady state): projection = the known shift, map point = back-projection at depth z
        lk = l_kps.view(torch.float32).view(B, cap, 7)
        l_oct = l_kps.view(torch.int32).view(B, cap, 7)[:, :, 5].contiguous()
        l_u = (lk[:, :, 0] + float(shift)).contiguous(); l_v = lk[:, :, 1].contiguous(); l_ang = lk[:, :, 3].contiguous()
        l_valid = (torch.arange(cap, device=dev)[None, :] < l_n[:, None]).to(torch.uint8).contiguous()
        zs = torch.from_numpy(rs.uniform(2.0, 14.0, (B, cap)).astype(np.float32)).to(dev)
        l_mp = torch.stack([(l_u - np.float32(cam["cx"])) / np.float32(cam["fx"]) * zs, (l_v - np.float32(cam["cy"])) / np.float32(cam["fy"]) * zs, zs], 2).contiguous()
        p0 = np.zeros((B, 7)); p0[:, 3] = 1.0
        p0[:, :3] = rs.normal(0, 0.01, (B, 3)); p0[:, 4:] = rs.normal(0, 0.03, (B, 3))
        p0 = p0.astype(np.float32).astype(np.float64)
        d_p0 = torch.from_numpy(p0).to(dev)
        t_assign = torch.full((B * cap,), -1, dtype=torch.int32, device=dev); t_occ = z(torch.uint8, cap); t_nm = z(torch.int32, 1)
        d_pose = torch.zeros(B, 7, dtype=torch.float64, device=dev); d_inl = z(torch.int32, 1); d_outl = z(torch.uint8, cap)
        # ---- the chain: three calls enqueued back to back, one synchronisation at the end ----
        ex.extract_batch_device(d_cur.data_ptr(), B, W_, H_, W_, W_ * H_, c_kps.data_ptr(), c_desc.data_ptr(), cap, c_n.data_ptr(), d_mono.data_ptr(), d_st.data_ptr(), (0, 1000), st)
        m.SearchByProjection_last_batch_device((c_kps.data_ptr(), c_desc.data_ptr(), c_n.data_ptr(), cap),
                                               (l_valid.data_ptr(), l_u.data_ptr(), l_v.data_ptr(), l_oct.data_ptr(), l_ang.data_ptr(), l_desc.data_ptr(), l_n.data_ptr(), cap),
                                               B, 15.0, t_assign.data_ptr(), t_occ.data_ptr(), t_nm.data_ptr(), st, bounds=(0.0, 0.0, float(W_), float(H_)), scale_factors=sfac)
        ps.optimize_batch_device(B, cap, c_kps.data_ptr(), c_n.data_ptr(), t_assign.data_ptr(), l_mp.data_ptr(), cap, d_p0.data_ptr(), isig, cam,
                                 d_pose.data_ptr(), d_inl.data_ptr(), d_outl.data_ptr(), st)
        torch.cuda.synchronize()
        assign = t_assign.cpu().numpy().reshape(B, cap); nm = t_nm.cpu().numpy(); pose = d_pose.cpu().numpy(); inl = d_inl.cpu().numpy()
        outl = d_outl.cpu().numpy().reshape(B, cap); mp_host = l_mp.cpu().numpy()
    finally:
        ex.close(); m.close(); ps.close()
    for b in range(B):
        _, kL, dL = oex.extract(last_imgs[b], (0, 1000)); _, kC, dC = oex.extract(cur_imgs[b], (0, 1000))
        nL, nC = len(kL), len(kC)
        g = dict(x=np.ascontiguousarray(kC["x"]), y=np.ascontiguousarray(kC["y"]), octave=np.ascontiguousarray(kC["octave"]),
                 min_x=0.0, min_y=0.0, max_x=float(W_), max_y=float(H_), cols=64, rows=48)
        last = dict(u=(kL["x"] + np.float32(shift)).astype(np.float32), v=np.ascontiguousarray(kL["y"]), octave=np.ascontiguousarray(kL["octave"]),
                    angle=np.ascontiguousarray(kL["angle"]), valid=np.ones(nL, np.uint8), desc=dL, has_obs=np.ones(nL, np.uint8))
        a0 = np.full(nC, -1, np.int32); o0 = np.zeros(nC, np.uint8)
        n0 = oracle.search_by_projection_last(g, dC, np.ascontiguousarray(kC["angle"]), sfac, last, 15.0, True, a0, o0)
        assert nm[b] == n0 > min_matches
        np.testing.assert_array_equal(assign[b, :nC], a0)
        feat = np.nonzero(a0 >= 0)[0]
        mpL = mp_host[b, :nL]              # (built on the device from the device's key points, which are the oracle's bit for bit)
        w = dict(cam, q=p0[b, :4], t=p0[b, 4:], Xw=mpL[a0[feat]].astype(np.float64), stereo=np.zeros(len(feat), np.uint8),
                 obs=np.stack([kC["x"][feat], kC["y"][feat], np.full(len(feat), -1.0, np.float32)], 1).astype(np.float64),
                 inv_sigma2=isig[kC["octave"][feat]].astype(np.float64))
        gref = oracle_pose_optimize(oracle, w)
        full = np.zeros(cap, np.uint8); full[feat] = gref["outlier"]
        np.testing.assert_array_equal(outl[b], full)
        assert inl[b] == int(gref["inliers"]) > min_inliers
        q0 = p0[b, :4] / np.linalg.norm(p0[b, :4])
        dq = np.abs(np.asarray(gref["q"]) - q0).max(); dt = np.abs(np.asarray(gref["t"]) - p0[b, 4:]).max()
        assert np.abs(pose[b, :4] - gref["q"]).max() <= 1e-4 * dq + 1e-12 and np.abs(pose[b, 4:] - gref["t"]).max() <= 1e-4 * dt + 1e-12
        assert np.abs(pose[b, 4:]).max() < 0.02          # and it found the true pose (identity)
