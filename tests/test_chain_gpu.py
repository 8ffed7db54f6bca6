"""The tracking front end as ONE device-resident chain -- orbx_extract_batch_device -> orbv_transform_batch_device ->
SearchByBoW on a device-resident plan -- with no host round trip between the three, against the oracle chain
(ORBextractor::operator() -> TemplatedVocabulary::transform -> ORBmatcher::SearchByBoW) on the same images."""
import numpy as np
import pytest

from oracle_api import oracle_transform

pytestmark = pytest.mark.gpu


def test_extract_transform_bow_chain(pkg, oracle, synth):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    P = 3                                                    # pairs: (frame, the same frame moved by a few pixels)
    imgs = []
    for i in range(P):
        base = synth.make_frame(60 + i)
        imgs += [base, np.ascontiguousarray(np.roll(base, 3 + i, axis=1))]
    imgs = np.stack(imgs)
    B = 2 * P
    voc = synth.make_vocabulary(5, k=10, L=3, ragged=False, tie_frac=0.0, stop_frac=0.0)
    levelsup = 1
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
        ex.extract_batch_device(d_img.data_ptr(), B, 640, 480, 640, 640 * 480, d_kps.data_ptr(), d_desc.data_ptr(), cap,
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
        assert nm[p] == n0 and n0 > 100
        np.testing.assert_array_equal(match[p, :len(kF)], m0)
