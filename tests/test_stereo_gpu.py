"""HIP Frame::ComputeStereoMatches (reference src/Frame.cc:931-1101) against the CPU restatement, on the pyramids and key
points the two extractors produced for a synthetic rectified pair: mvuRight / mvDepth must be bit-identical floats."""
import numpy as np
import pytest

from oracle_api import oracle_stereo_matches

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,nfeat,mb,mbf", [(0, 1000, 0.11, 47.9), (1, 1000, 0.11, 47.9), (2, 2000, 0.06, 25.0), (3, 300, 0.5, 200.0)])
def test_stereo_matches_equal_oracle(pkg, oracle, synth, seed, nfeat, mb, mbf):
    left, right = synth.make_stereo_pair(seed)
    oL, oR = oracle.extractor(nfeat, 1.2, 8, 20, 7), oracle.extractor(nfeat, 1.2, 8, 20, 7)
    _, kL0, dL0 = oL.extract(left, (0, 0)); _, kR0, dR0 = oR.extract(right, (0, 0))
    n0, ur0, dp0 = oracle_stereo_matches(oL, oR, kL0, dL0, kR0, dR0, mb, mbf)
    assert n0 > 0.3 * len(kL0)
    exL, exR = pkg.Extractor(nfeat, 1.2, 8, 20, 7), pkg.Extractor(nfeat, 1.2, 8, 20, 7)
    try:
        _, kL, dL = exL(left, (0, 0)); _, kR, dR = exR(right, (0, 0))
        for f in kL0.dtype.names:
            np.testing.assert_array_equal(kL[f], kL0[f]); np.testing.assert_array_equal(kR[f], kR0[f])
        ur1, dp1 = exL.stereo_matches(exR, kL, dL, kR, dR, mb, mbf)
    finally:
        exL.close(); exR.close()
    np.testing.assert_array_equal(ur1, ur0)
    np.testing.assert_array_equal(dp1, dp0)
    v = ur0 >= 0
    assert v.sum() > 0.2 * len(kL0) and (kL0["x"][v] - ur0[v] >= 0).all()


def test_stereo_matches_batch_device(pkg, oracle, synth):
    """device-resident batch: both extractors' outputs stay in HBM, one launch for all frames"""
    torch = pytest.importorskip("torch")
    B = 3
    pairs = [synth.make_stereo_pair(10 + b) for b in range(B)]
    L = np.stack([p[0] for p in pairs]); R = np.stack([p[1] for p in pairs])
    dev = torch.device("cuda", 0)
    exL, exR = pkg.Extractor(), pkg.Extractor()
    cap = exL.max_keypoints
    try:
        outs = []
        for ex, imgs in ((exL, L), (exR, R)):
            d_img = torch.from_numpy(imgs.copy()).to(dev)
            o = dict(kps=torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev), desc=torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev),
                     n=torch.zeros(B, dtype=torch.int32, device=dev), mono=torch.zeros(B, dtype=torch.int32, device=dev),
                     st=torch.zeros(B, dtype=torch.int32, device=dev), img=d_img)
            ex.extract_batch_device(d_img.data_ptr(), B, 640, 480, 640, 640 * 480, o["kps"].data_ptr(), o["desc"].data_ptr(), cap,
                                    o["n"].data_ptr(), o["mono"].data_ptr(), o["st"].data_ptr(), (0, 0), torch.cuda.current_stream().cuda_stream)
            outs.append(o)
        d_ur = torch.zeros(B * cap, dtype=torch.float32, device=dev); d_dp = torch.zeros(B * cap, dtype=torch.float32, device=dev)
        exL.stereo_matches_device(exR, B, outs[0]["kps"].data_ptr(), outs[0]["desc"].data_ptr(), outs[0]["n"].data_ptr(),
                                  outs[1]["kps"].data_ptr(), outs[1]["desc"].data_ptr(), outs[1]["n"].data_ptr(), cap, 0.11, 47.9,
                                  d_ur.data_ptr(), d_dp.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ur = d_ur.cpu().numpy().reshape(B, cap); dp = d_dp.cpu().numpy().reshape(B, cap); nl = outs[0]["n"].cpu().numpy()
    finally:
        exL.close(); exR.close()
    for b in range(B):
        oL, oR = oracle.extractor(), oracle.extractor()
        _, kL0, dL0 = oL.extract(pairs[b][0], (0, 0)); _, kR0, dR0 = oR.extract(pairs[b][1], (0, 0))
        _, ur0, dp0 = oracle_stereo_matches(oL, oR, kL0, dL0, kR0, dR0, 0.11, 47.9)
        assert nl[b] == len(kL0)
        np.testing.assert_array_equal(ur[b, :nl[b]], ur0); np.testing.assert_array_equal(dp[b, :nl[b]], dp0)
