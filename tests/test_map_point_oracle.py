"""Map-point upkeep restatements (oracle/map_point_oracle.cpp; reference src/MapPoint.cc:329-402, :433-493) against plain
numpy -- PARITY UNPINNED (the reference holds no fixture for them)."""
import numpy as np

from oracle_api import oracle_distinctive, oracle_normal_and_depth

_POP = np.array([bin(i).count("1") for i in range(256)], np.int32)


def make_points(seed, P, max_obs):
    rs = np.random.RandomState(seed)
    cnt = rs.randint(1, max_obs + 1, P); cnt[0] = max_obs
    if P > 3:
        cnt[1] = 1; cnt[2] = 2
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    base = rs.randint(0, 256, (P, 32)).astype(np.uint8)
    desc = np.repeat(base, cnt, axis=0) ^ (np.packbits(rs.uniform(size=(off[-1], 256)) < 0.08, axis=1))
    if P > 4:                                               # ties: identical descriptors -> equal medians, first one wins
        desc[off[3]:off[4]] = desc[off[3]]
    return desc, off


def _numpy_distinctive(desc, off):
    out = []
    for p in range(len(off) - 1):
        d = desc[off[p]:off[p + 1]]
        N = len(d)
        D = _POP[d[:, None, :] ^ d[None, :, :]].sum(2)
        med = np.sort(D, axis=1)[:, int(0.5 * (N - 1))]
        out.append((int(np.argmin(med)), int(med.min())))
    return out


def test_distinctive_against_numpy(oracle):
    for seed, P, mo in ((0, 40, 12), (1, 10, 70), (2, 5, 200)):
        desc, off = make_points(seed, P, mo)
        bi, bm = oracle_distinctive(oracle, desc, off)
        assert list(zip(bi.tolist(), bm.tolist())) == _numpy_distinctive(desc, off)
    # a point without descriptors is left alone (the reference returns early)
    bi, _ = oracle_distinctive(oracle, np.zeros((3, 32), np.uint8), [0, 0, 3])
    assert list(bi) == [-1, 0]


def make_geometry(seed, P, max_obs):
    rs = np.random.RandomState(seed)
    cnt = rs.randint(1, max_obs + 1, P)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    pos = rs.uniform(-5, 5, (P, 3)).astype(np.float32)
    centers = (np.repeat(pos, cnt, axis=0) + rs.normal(0, 4, (off[-1], 3))).astype(np.float32)
    ref = centers[off[:-1]].copy()
    level_scale = (np.float32(1.2) ** rs.randint(0, 8, P)).astype(np.float32)
    return pos, centers, off, ref, level_scale


def test_normal_and_depth_against_numpy(oracle):
    pos, centers, off, ref, ls = make_geometry(3, 200, 15)
    last = np.float32(1.2) ** 7
    nrm, mx, mn = oracle_normal_and_depth(oracle, pos, centers, off, ref, ls, last)
    for p in range(200):
        d = pos[p].astype(np.float64) - centers[off[p]:off[p + 1]].astype(np.float64)
        n64 = (d / np.linalg.norm(d, axis=1, keepdims=True)).sum(0) / len(d)
        assert np.allclose(nrm[p], n64, rtol=0, atol=1e-5)
        dist = np.linalg.norm(pos[p].astype(np.float64) - ref[p])
        assert abs(mx[p] - dist * ls[p]) <= 1e-5 * dist * ls[p] and mn[p] == np.float32(mx[p] / last)
