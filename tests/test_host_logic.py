"""Host-side logic of the product, runnable without a GPU: the libstdc++-std::sort restatement used by the device
octree, the shared float helpers, and the synthetic generators."""
import ctypes as C

import numpy as np
import pytest


def _sort_both(pkg, oracle, count, ulx):
    n = len(count)
    tag = np.arange(n, dtype=np.int32)
    c1, u1, t1 = count.copy(), ulx.copy(), tag.copy()
    c2, u2, t2 = count.copy(), ulx.copy(), tag.copy()
    oracle.lib.orb_oracle_sort_nodes(c1.ctypes.data, u1.ctypes.data, t1.ctypes.data, n)          # std::sort
    assert pkg.lib.orbx_debug_introsort(c2.ctypes.data, u2.ctypes.data, t2.ctypes.data, n) == 0  # product restatement
    return (c1, u1, t1), (c2, u2, t2)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 100, 217, 500, 1085, 4000])
def test_introsort_matches_std_sort_with_ties(pkg, oracle, n):
    rs = np.random.RandomState(n)
    for trial in range(6):
        count = rs.randint(2, 2 + max(1, (trial + 1) * 3), n).astype(np.int32)      # heavy ties on count
        ulx = (rs.randint(0, 1 + trial * 2, n) * 19).astype(np.int32)               # and on UL.x
        a, b = _sort_both(pkg, oracle, count, ulx)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)     # including the order of tied elements (tag)


def test_introsort_adversarial(pkg, oracle):
    # sorted, reversed, all-equal, organ-pipe and a median-of-3 killer: exercise the depth limit / heapsort fallback
    n = 3000
    seqs = [np.arange(n), np.arange(n)[::-1], np.zeros(n), np.minimum(np.arange(n), n - np.arange(n))]
    k = n // 2
    killer = np.zeros(n, np.int64)
    for i in range(k):
        if i % 2 == 0:
            killer[i] = i + 1
        else:
            killer[i] = k + i + (1 if k % 2 == 0 else 0)
        killer[k + i] = (i + 1) * 2
    seqs.append(killer)
    for s in seqs:
        count = np.ascontiguousarray(s).astype(np.int32)
        ulx = np.zeros(n, np.int32)
        a, b = _sort_both(pkg, oracle, count, ulx)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


def test_shared_float_helpers_bit_identical(pkg, oracle):
    rs = np.random.RandomState(5)
    L, O = pkg.lib, oracle.lib
    for _ in range(5000):
        y, x = np.float32(rs.randint(-3_000_000, 3_000_000)), np.float32(rs.randint(-3_000_000, 3_000_000))
        assert L.orbx_debug_fast_atan2(float(y), float(x)) == O.orb_oracle_fast_atan2(float(y), float(x))
    c1, s1, c2, s2 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    for a in np.concatenate([rs.uniform(0, 6.2832, 5000), [0.0, np.pi / 2, np.pi, 2 * np.pi]]).astype(np.float32):
        L.orbx_debug_sincos(float(a), C.byref(c1), C.byref(s1))
        O.orb_oracle_sincos(float(a), C.byref(c2), C.byref(s2))
        assert c1.value == c2.value and s1.value == s2.value


def test_synth_generators_deterministic(synth):
    a, b = synth.make_frame(3), synth.make_frame(3)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (480, 640) and a.dtype == np.uint8 and 30 < a.std() < 90
    assert (a[:, :] == 97).all(axis=0).sum() >= 60          # the flat strip
    ms = synth.make_match_set(0)
    assert len(ms["fvKF"][0]) <= 100 and ms["fvF"][1][-1] == 1000
    w = synth.make_ba_window(0, n_opt=5, n_fixed=2, n_points=40, obs_per_point=4)
    assert w["pose_fixed"].sum() == 2 and len(w["edge_point"]) > 100
