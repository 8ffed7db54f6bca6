"""Edge-SLAM packets on the device (orbe_*; reference include/Socket/slampkt_vi.h:85-193) against the oracle -- bit-exact."""
import os

import numpy as np
import pytest

from oracle_api import IMU_DTYPE, oracle_pack_packet, oracle_undistort, oracle_unpack_packet

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _batch(pkg, seed, B, cap, max_imu):
    rs = np.random.RandomState(seed)
    n = rs.randint(0, cap + 1, B).astype(np.int32); n[0] = cap
    if B > 2:
        n[1] = 0
    kps = np.zeros((B, cap), pkg.KP_DTYPE)
    kps["x"] = rs.uniform(0, 752, (B, cap)).astype(np.float32); kps["y"] = rs.uniform(0, 480, (B, cap)).astype(np.float32)
    kps["size"] = 31; kps["angle"] = rs.uniform(0, 360, (B, cap)); kps["octave"] = rs.randint(0, 8, (B, cap))
    desc = rs.randint(0, 256, (B, cap, 32)).astype(np.uint8)
    m = rs.randint(0, max_imu + 1, B); off = np.concatenate([[0], np.cumsum(m)]).astype(np.int32)
    imu = np.zeros(int(off[-1]), IMU_DTYPE)
    imu["ts"] = rs.randint(0, 2 ** 62, len(imu)); imu["gyro"] = rs.normal(0, 1, (len(imu), 3)); imu["acce"] = rs.normal(0, 9, (len(imu), 3))
    fid = rs.randint(-5, 100000, B).astype(np.int32); ts = rs.randint(0, 2 ** 62, B).astype(np.int64)
    return kps, desc, n, fid, ts, imu, off


def test_pack_matches_oracle(pkg, oracle):
    codec = pkg.PacketCodec()
    for seed, B, cap, max_imu in ((0, 5, 1000, 20), (1, 64, 300, 3), (2, 1, 1, 0), (3, 3, 1700, 40)):
        kps, desc, n, fid, ts, imu, off = _batch(pkg, seed, B, cap, max_imu)
        pay, ln, head, st = codec.pack_batch(kps, desc, n, fid, ts, imu, off)
        assert (st == 0).all()
        for b in range(B):
            ref, rhead = oracle_pack_packet(oracle, fid[b], ts[b], kps[b, :n[b]], desc[b, :n[b]], imu[off[b]:off[b + 1]])
            assert ln[b] == len(ref) and np.array_equal(pay[b, :ln[b]], ref), (seed, b)
            assert np.array_equal(head[b], rhead)
            assert not pay[b, ln[b]:].any()                      # nothing written past the packet
    # no IMU block at all
    kps, desc, n, fid, ts, _, _ = _batch(pkg, 9, 4, 200, 0)
    pay, ln, head, st = codec.pack_batch(kps, desc, n, fid, ts)
    for b in range(4):
        ref, _ = oracle_pack_packet(oracle, fid[b], ts[b], kps[b, :n[b]], desc[b, :n[b]])
        assert np.array_equal(pay[b, :ln[b]], ref)
    codec.close()


def test_unpack_matches_oracle_and_roundtrip(pkg, oracle):
    codec = pkg.PacketCodec()
    kps, desc, n, fid, ts, imu, off = _batch(pkg, 4, 32, 1200, 25)
    pay, ln, _, st = codec.pack_batch(kps, desc, n, fid, ts, imu, off)
    out = codec.unpack_batch(pay, ln, cap=1200, imu_cap=25)
    assert (out["status"] == 0).all() and np.array_equal(out["n"], n) and np.array_equal(out["frame_id"], fid) and np.array_equal(out["timestamp"], ts)
    for b in range(32):
        r, f2, t2, k2, d2, i2 = oracle_unpack_packet(oracle, pay[b, :ln[b]])
        assert r == 0
        assert out["kps"][b, :n[b]].tobytes() == k2.tobytes() and np.array_equal(out["desc"][b, :n[b]], d2)
        assert out["n_imu"][b] == len(i2) and out["imu"][b, :len(i2)].tobytes() == i2.tobytes()
        # round trip: what the server sees is the client's key point truncated to u16, descriptors untouched
        assert np.array_equal(out["kps"]["x"][b, :n[b]], np.trunc(kps["x"][b, :n[b]])) and np.array_equal(out["desc"][b, :n[b]], desc[b, :n[b]])
        assert not out["desc"][b, n[b]:].any()
    codec.close()


def test_malformed_and_capacity(pkg, oracle):
    codec = pkg.PacketCodec()
    kps, desc, n, fid, ts, imu, off = _batch(pkg, 5, 4, 100, 4)
    n[:] = [100, 50, 10, 3]
    pay, ln, _, st = codec.pack_batch(kps, desc, n, fid, ts, imu, off)
    bad = ln.copy(); bad[1] -= 1; bad[2] = 15
    out = codec.unpack_batch(pay, bad, cap=60, imu_cap=4)
    assert list(out["status"]) == [-2, -3, -3, 0]              # capacity (100 > 60), shorter than its counts, shorter than the info block
    assert list(out["n"]) == [0, 0, 0, 3]
    assert oracle_unpack_packet(oracle, pay[1, :bad[1]])[0] == -1 and oracle_unpack_packet(oracle, pay[0, :ln[0]], cap_pts=60)[0] == -2
    # packing into a stride that is too small flags the frame and writes nothing
    pay2, ln2, _, st2 = codec.pack_batch(kps, desc, n, fid, ts, imu, off, stride=16 + 36 * 50 + 32 * 4)
    assert st2[0] == -2 and not pay2[0].any() and st2[1] == 0 and ln2[0] == ln[0]
    # a packet past 65536 bytes cannot be announced by getHead()
    kb = np.zeros((1, 1900), pkg.KP_DTYPE); db = np.zeros((1, 1900, 32), np.uint8)
    _, ln3, _, st3 = codec.pack_batch(kb, db, [1900], [0], [0])
    assert ln3[0] == 16 + 36 * 1900 and st3[0] == -3
    with pytest.raises(pkg.OrbxError):
        codec.pack_batch(kps, desc, n, fid, ts, imu, off, stride=18)
    codec.close()


def test_golden_packet(pkg):
    g = np.load(os.path.join(GOLDEN, "edge_packet_0.npz"))
    codec = pkg.PacketCodec()
    kps = np.zeros((1, len(g["kps_x"])), pkg.KP_DTYPE); kps["x"][0] = g["kps_x"]; kps["y"][0] = g["kps_y"]
    imu = np.zeros(len(g["imu_ts"]), IMU_DTYPE); imu["ts"] = g["imu_ts"]; imu["gyro"] = g["imu_gyro"]; imu["acce"] = g["imu_acce"]
    pay, ln, head, st = codec.pack_batch(kps, g["desc"][None], [kps.shape[1]], [int(g["frame_id"])], [int(g["timestamp"])], imu,
                                         np.array([0, len(imu)], np.int32))
    assert st[0] == 0 and np.array_equal(pay[0, :ln[0]], g["payload"]) and np.array_equal(head[0], g["head"])
    out = codec.unpack_batch(g["payload"][None], [len(g["payload"])], cap=kps.shape[1], imu_cap=len(imu))
    assert np.array_equal(out["kps"]["x"][0], g["unpacked_x"]) and np.array_equal(out["kps"]["y"][0], g["unpacked_y"])
    codec.close()


def test_extractor_to_packet_device_chain(pkg, oracle, synth):
    """client side of the fork: extract on the device, write the packets from the extractor's device arrays, unpack them
    again on the device (server side) -- the descriptors that arrive are the extractor's, bit for bit"""
    import torch
    B, cap = 6, 1200
    imgs = np.stack([synth.make_frame(40 + i) for i in range(B)])
    ex = pkg.Extractor()
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(imgs).to(dev)
    d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev); d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev); d_mono = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
    codec = pkg.PacketCodec()
    stride = (codec.packet_bytes(cap, 0) + 3) & ~3
    d_pay = torch.zeros(B * stride, dtype=torch.uint8, device=dev); d_len = torch.zeros(B, dtype=torch.int32, device=dev)
    d_pst = torch.zeros(B, dtype=torch.int32, device=dev)
    d_fid = torch.arange(B, dtype=torch.int32, device=dev); d_ts = torch.arange(B, dtype=torch.int64, device=dev) * 50000000
    s = torch.cuda.current_stream().cuda_stream
    # extract -> pack -> unpack are enqueued back to back on one stream; nothing is read in between
    ex.extract_batch_device(d_img.data_ptr(), B, 640, 480, 640, 640 * 480, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_n.data_ptr(),
                            d_mono.data_ptr(), d_st.data_ptr(), (0, 1000), s)
    codec.pack_batch_device(d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), B, cap, d_fid.data_ptr(), d_ts.data_ptr(), 0, 0,
                            d_pay.data_ptr(), stride, d_len.data_ptr(), 0, d_pst.data_ptr(), s)
    k2 = torch.zeros_like(d_kps); dd2 = torch.zeros_like(d_desc); n2 = torch.zeros_like(d_n); f2 = torch.zeros_like(d_fid); t2 = torch.zeros_like(d_ts)
    ni2 = torch.zeros_like(d_n); st2 = torch.zeros_like(d_n)
    codec.unpack_batch_device(d_pay.data_ptr(), stride, d_len.data_ptr(), B, cap, 0, k2.data_ptr(), dd2.data_ptr(), n2.data_ptr(), f2.data_ptr(),
                              t2.data_ptr(), 0, ni2.data_ptr(), st2.data_ptr(), s)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy()
    assert (d_pst.cpu().numpy() == 0).all() and (st2.cpu().numpy() == 0).all() and np.array_equal(n2.cpu().numpy(), n) and n.min() > 500
    kh = d_kps.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, cap); dh = d_desc.cpu().numpy().reshape(B, cap, 32)
    pay = d_pay.cpu().numpy().reshape(B, stride); ln = d_len.cpu().numpy()
    k2h = k2.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, cap); d2h = dd2.cpu().numpy().reshape(B, cap, 32)
    for b in range(B):
        ref, _ = oracle_pack_packet(oracle, b, b * 50000000, kh[b, :n[b]], dh[b, :n[b]])
        assert np.array_equal(pay[b, :ln[b]], ref)
        assert np.array_equal(d2h[b, :n[b]], dh[b, :n[b]]) and np.array_equal(k2h["x"][b, :n[b]], np.trunc(kh["x"][b, :n[b]]))
    codec.close(); ex.close()


def test_undistort_matches_oracle(pkg, oracle):
    """Frame::UndistortKeyPoints on the device arrays a server has just unpacked -- floats bit-identical to the oracle"""
    import torch
    K = (458.654, 457.296, 367.215, 248.375); dist = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)
    B, cap = 7, 1500
    kps, _, n, _, _, _, _ = _batch(pkg, 11, B, cap, 0)
    dev = torch.device("cuda:0")
    d_k = torch.from_numpy(kps.view(np.uint8).reshape(-1)).to(dev); d_n = torch.from_numpy(n).to(dev); d_o = torch.zeros_like(d_k)
    codec = pkg.PacketCodec()
    s = torch.cuda.current_stream().cuda_stream
    codec.undistort_batch_device(d_k.data_ptr(), d_n.data_ptr(), B, cap, K, dist, K, d_o.data_ptr(), s)
    torch.cuda.synchronize()
    out = d_o.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, cap)
    for b in range(B):
        ref = oracle_undistort(oracle, kps[b, :n[b]], K, dist, K)
        assert out[b, :n[b]].tobytes() == ref.tobytes(), b
        assert not out[b, n[b]:].view(np.uint8).any()           # rows past the count are not written
    # in place, and the no-distortion copy
    codec.undistort_batch_device(d_k.data_ptr(), d_n.data_ptr(), B, cap, K, (0, 0, 0, 0, 0), K, d_k.data_ptr(), s)
    torch.cuda.synchronize()
    assert d_k.cpu().numpy().tobytes() == kps.tobytes()
    codec.close()
