"""Map-point upkeep on the device (orbm_distinctive_descriptors, orbm_update_normal_and_depth; reference
src/MapPoint.cc:329-402, :433-493) against the oracle: indices / medians identical, floats bit-identical."""
import numpy as np
import pytest

from oracle_api import oracle_distinctive, oracle_normal_and_depth
from test_map_point_oracle import make_geometry, make_points

pytestmark = pytest.mark.gpu


def test_distinctive_matches_oracle(pkg, oracle):
    m = pkg.Matcher()
    for seed, P, mo in ((0, 1000, 12), (1, 300, 70), (2, 20, 200), (3, 1, 1), (4, 3, 1000)):
        desc, off = make_points(seed, P, mo)
        bi, bm = m.DistinctiveDescriptors(desc, off)
        obi, obm = oracle_distinctive(oracle, desc, off)
        assert np.array_equal(bi, obi) and np.array_equal(bm, obm), seed
    bi, bm = m.DistinctiveDescriptors(np.zeros((3, 32), np.uint8), [0, 0, 3])
    assert list(bi) == [-1, 0] and list(bm) == [-1, 0]
    with pytest.raises(pkg.OrbxError):
        m.DistinctiveDescriptors(np.zeros((3, 32), np.uint8), [0, 2, 1])
    m.close()


def test_normal_and_depth_matches_oracle(pkg, oracle):
    m = pkg.Matcher()
    last = float(np.float32(1.2) ** 7)
    for seed, P, mo in ((5, 2000, 15), (6, 1, 1), (7, 300, 120)):
        pos, centers, off, ref, ls = make_geometry(seed, P, mo)
        nrm, mx, mn = m.UpdateNormalAndDepth(pos, centers, off, ref, ls, last)
        onrm, omx, omn = oracle_normal_and_depth(oracle, pos, centers, off, ref, ls, last)
        assert nrm.tobytes() == onrm.tobytes() and mx.tobytes() == omx.tobytes() and mn.tobytes() == omn.tobytes(), seed
    pos, centers, off, ref, ls = make_geometry(8, 4, 3)
    off2 = off.copy(); off2[2] = off2[1]                       # a point without observations: the reference returns early
    with pytest.raises(pkg.OrbxError):
        m.UpdateNormalAndDepth(pos, centers, off2, ref, ls, last)
    m.close()
