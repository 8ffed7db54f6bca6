"""Edge-SLAM wire format restatement (oracle/edge_packet_oracle.cpp; reference include/Socket/slampkt_vi.h:85-193).
PARITY UNPINNED: the reference holds no packet fixture.  The layout is pinned here to a packet written out by hand from the
header's own description (:19-21: 16-byte info block, 36 bytes per key point, 32 bytes per IMU sample) and to struct.pack."""
import os
import struct

import numpy as np

from oracle_api import IMU_DTYPE, KP_DTYPE, oracle_pack_packet, oracle_undistort, oracle_unpack_packet


def _python_pack(frame_id, ts, kps, desc, imu):
    out = struct.pack("<iq", frame_id, ts) + struct.pack(">HH", len(kps), len(imu))
    for k, d in zip(kps, desc):
        out += struct.pack(">HH", int(k["x"]), int(k["y"])) + bytes(d)
    for s in imu:
        out += struct.pack("<q3f3f", int(s["ts"]), *[float(v) for v in s["gyro"]], *[float(v) for v in s["acce"]])
    return np.frombuffer(out, np.uint8)


def _case(seed, n, m):
    rs = np.random.RandomState(seed)
    kps = np.zeros(n, KP_DTYPE)
    kps["x"] = rs.uniform(0, 752, n).astype(np.float32); kps["y"] = rs.uniform(0, 480, n).astype(np.float32)
    kps["size"] = 31; kps["angle"] = rs.uniform(0, 360, n); kps["response"] = rs.uniform(7, 200, n); kps["octave"] = rs.randint(0, 8, n)
    desc = rs.randint(0, 256, (n, 32)).astype(np.uint8)
    imu = np.zeros(m, IMU_DTYPE)
    imu["ts"] = 1403636579763555584 + np.arange(m) * 5000000
    imu["gyro"] = rs.normal(0, 0.2, (m, 3)); imu["acce"] = rs.normal(0, 9.8, (m, 3))
    return kps, desc, imu


def test_known_answer_packet(oracle):
    kps = np.zeros(2, KP_DTYPE)
    kps["x"] = [258.75, 3.0]; kps["y"] = [1.9, 479.0]
    desc = np.stack([np.arange(32), 255 - np.arange(32)]).astype(np.uint8)
    imu = np.zeros(1, IMU_DTYPE); imu["ts"] = 0x0102030405060708; imu["gyro"] = [[1.0, -2.0, 0.5]]; imu["acce"] = [[0.0, 9.81, -0.25]]
    pay, head = oracle_pack_packet(oracle, 0x11223344, 0x0A0B0C0D0E0F1011, kps, desc, imu)
    assert len(pay) == 16 + 2 * 36 + 32 and list(head) == [0, 120]
    assert list(pay[:16]) == [0x44, 0x33, 0x22, 0x11, 0x11, 0x10, 0x0F, 0x0E, 0x0D, 0x0C, 0x0B, 0x0A, 0, 2, 0, 1]
    assert list(pay[16:20]) == [1, 2, 0, 1]                     # (unsigned short)258.75 = 258 = 0x0102, (unsigned short)1.9 = 1
    assert list(pay[20:52]) == list(range(32))
    assert list(pay[52:56]) == [0, 3, 1, 0xDF]                  # 479 = 0x01DF
    assert list(pay[88:96]) == [8, 7, 6, 5, 4, 3, 2, 1]
    assert pay[96:].tobytes() == struct.pack("<6f", 1.0, -2.0, 0.5, 0.0, np.float32(9.81), -0.25)


def test_pack_against_struct(oracle):
    for seed, n, m in ((0, 1000, 20), (1, 0, 0), (2, 1, 0), (3, 0, 7), (4, 1500, 200)):
        kps, desc, imu = _case(seed, n, m)
        pay, head = oracle_pack_packet(oracle, 7 + seed, 1403636579763555584 + seed, kps, desc, imu)
        ref = _python_pack(7 + seed, 1403636579763555584 + seed, kps, desc, imu)
        assert np.array_equal(pay, ref)
        assert int(head[0]) * 256 + int(head[1]) == len(ref) & 0xffff


def test_unpack_roundtrip_and_defaults(oracle):
    kps, desc, imu = _case(5, 800, 12)
    pay, _ = oracle_pack_packet(oracle, -3, 123456789012345, kps, desc, imu)
    r, fid, ts, k2, d2, i2 = oracle_unpack_packet(oracle, pay)
    assert r == 0 and fid == -3 and ts == 123456789012345
    assert np.array_equal(k2["x"], np.trunc(kps["x"])) and np.array_equal(k2["y"], np.trunc(kps["y"]))
    assert (k2["size"] == 1).all() and (k2["angle"] == -1).all() and (k2["response"] == 0).all()       # cv::KeyPoint(x, y, 1)
    assert (k2["octave"] == 0).all() and (k2["class_id"] == -1).all()
    assert np.array_equal(d2, desc) and i2.tobytes() == imu.tobytes()


def test_unpack_rejects_short_packets(oracle):
    kps, desc, imu = _case(6, 10, 2)
    pay, _ = oracle_pack_packet(oracle, 1, 2, kps, desc, imu)
    assert oracle_unpack_packet(oracle, pay[:-1])[0] == -1
    assert oracle_unpack_packet(oracle, pay[:15])[0] == -1
    assert oracle_unpack_packet(oracle, pay, cap_pts=9)[0] == -2


def test_golden_packet(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "edge_packet_0.npz"))
    kps = np.zeros(len(g["kps_x"]), KP_DTYPE); kps["x"] = g["kps_x"]; kps["y"] = g["kps_y"]
    imu = np.zeros(len(g["imu_ts"]), IMU_DTYPE); imu["ts"] = g["imu_ts"]; imu["gyro"] = g["imu_gyro"]; imu["acce"] = g["imu_acce"]
    pay, head = oracle_pack_packet(oracle, int(g["frame_id"]), int(g["timestamp"]), kps, g["desc"], imu)
    assert np.array_equal(pay, g["payload"]) and np.array_equal(head, g["head"])
    assert np.array_equal(pay, _python_pack(int(g["frame_id"]), int(g["timestamp"]), kps, g["desc"], imu))


EUROC_K = (458.654, 457.296, 367.215, 248.375)
EUROC_DIST = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0)


def test_undistort_inverts_the_distortion_model(oracle):
    """the restated cv::undistortPoints against the forward radial / tangential model (Examples/Monocular/EuRoC.yaml:23-32)"""
    rs = np.random.RandomState(0)
    n = 500
    kps = np.zeros(n, KP_DTYPE)
    kps["x"] = rs.uniform(0, 752, n).astype(np.float32); kps["y"] = rs.uniform(0, 480, n).astype(np.float32)
    kps["size"] = 31; kps["octave"] = rs.randint(0, 8, n); kps["angle"] = rs.uniform(0, 360, n)
    un = oracle_undistort(oracle, kps, EUROC_K, EUROC_DIST, EUROC_K)
    fx, fy, cx, cy = EUROC_K; k1, k2, p1, p2, k3 = EUROC_DIST
    x = (un["x"].astype(np.float64) - cx) / fx; y = (un["y"].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x); yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    # five fixed iterations leave a few tenths of a pixel in the far corners of this strongly distorted camera, ~1e-5 px in the centre
    assert np.abs(xd * fx + cx - kps["x"]).max() < 0.5 and np.abs(yd * fy + cy - kps["y"]).max() < 0.5
    assert np.median(np.abs(xd * fx + cx - kps["x"])) < 2e-3
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(un[f], kps[f])
    # no distortion: a plain copy (src/Frame.cc:836-840)
    same = oracle_undistort(oracle, kps, EUROC_K, (0, 0.1, 0, 0, 0), EUROC_K)
    assert same.tobytes() == kps.tobytes()
