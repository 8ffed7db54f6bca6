"""CPU checks of the oracle's two tracking searches, rectified-stereo branches included, against a numpy restatement that
does not share the oracle's grid code (reference src/ORBmatcher.cc:43-138 and :1676-1887, Frame::GetFeaturesInArea
src/Frame.cc:744-810).  The reference ships no fixtures for this path: PARITY UNPINNED, this pins the restatement to a
second, independently written one."""
import numpy as np
import pytest

# (the synthetic-case module comes in through the `sm` fixture: importing the package at collection time would load the HIP
# library before torch and leave torch without a GPU in the same process -- see conftest.pkg)


def _visit_order(g):
    """Features in the order GetFeaturesInArea meets them when it walks every cell: ix outer, iy inner, insertion order
    (= ascending index) inside a cell; features PosInGrid rejects are in no cell (src/Frame.cc:472-503, :812-822)."""
    winv = np.float32(g["cols"]) / np.float32(g["max_x"] - g["min_x"])
    hinv = np.float32(g["rows"]) / np.float32(g["max_y"] - g["min_y"])
    fx = ((g["x"] - np.float32(g["min_x"])) * winv).astype(np.float32).astype(np.float64)
    fy = ((g["y"] - np.float32(g["min_y"])) * hinv).astype(np.float32).astype(np.float64)
    px = np.where(fx >= 0, np.floor(fx + 0.5), np.ceil(fx - 0.5)).astype(np.int64)      # C round(): half away from zero
    py = np.where(fy >= 0, np.floor(fy + 0.5), np.ceil(fy - 0.5)).astype(np.int64)
    ok = (px >= 0) & (px < g["cols"]) & (py >= 0) & (py < g["rows"])
    idx = np.nonzero(ok)[0]
    order = np.lexsort((idx, py[idx], px[idx]))
    return idx[order]


def _hamming(d, D):
    return np.unpackbits(d[None, :] ^ D, axis=1).sum(axis=1).astype(np.int64)


def _window(g, order, x, y, r, min_level, max_level):
    """box |dx| < r, |dy| < r (float), level band as GetFeaturesInArea's bCheckLevels logic"""
    c = order
    keep = (np.abs(g["x"][c] - np.float32(x)) < np.float32(r)) & (np.abs(g["y"][c] - np.float32(y)) < np.float32(r))
    if min_level > 0 or max_level >= 0:
        keep &= g["octave"][c] >= min_level
        if max_level >= 0:
            keep &= g["octave"][c] <= max_level
    return c[keep]


def numpy_search_last(g, dF, angF, scale, last, th, check_ori, assign, occupied):
    order = _visit_order(g)
    ur_f = g.get("u_right")
    lw = int(last.get("level_window", 0))
    hist = [[] for _ in range(30)]
    nm = 0
    for i in range(len(last["u"])):
        if not last["valid"][i]:
            continue
        u, v = last["u"][i], last["v"][i]
        if u < g["min_x"] or u > g["max_x"] or v < g["min_y"] or v > g["max_y"]:
            continue
        oc = int(last["octave"][i])
        r = np.float32(th) * scale[oc]
        lo, hi = {0: (oc - 1, oc + 1), 1: (oc, -1), 2: (0, oc)}[lw]
        c = _window(g, order, u, v, r, lo, hi)
        c = c[occupied[c] == 0]
        if ur_f is not None and len(c):
            er = np.abs(np.float32(last["ur"][i]) - ur_f[c])
            c = c[~((ur_f[c] > 0) & (er > r))]
        if not len(c):
            continue
        d = _hamming(last["desc"][i], dF[c])
        k = int(np.argmin(d))                       # first minimum in visiting order
        if d[k] > 100:
            continue
        j = int(c[k])
        assign[j] = i; occupied[j] = last["has_obs"][i]; nm += 1
        if check_ori:
            rot = np.float32(last["angle"][i]) - np.float32(angF[j])
            if rot < 0:
                rot = np.float32(rot + np.float32(360.0))
            b = float(np.float32(rot * np.float32(1.0 / 30)))
            b = int(np.floor(b + 0.5))
            hist[0 if b == 30 else b].append(j)
    if check_ori:
        cnt = [len(h) for h in hist]
        top = [b for b in sorted(range(30), key=lambda b: (-cnt[b], b))[:3] if cnt[b] > 0]      # strict '>' : first bin wins a tie
        keep = set(top[:1]) | {b for b in top[1:] if not (np.float32(cnt[b]) < np.float32(0.1) * np.float32(cnt[top[0]]))}
        for b in range(30):
            if b not in keep:
                for j in hist[b]:
                    assign[j] = -1; occupied[j] = 0; nm -= 1
    return nm


def numpy_search_mp(g, dF, scale, mp, th, nnratio, assign, occupied, b_far, th_far):
    order = _visit_order(g)
    ur_f = g.get("u_right")
    nm = 0
    for i in range(len(mp["u"])):
        if not mp["in_view"][i] or (b_far and mp["depth"][i] > th_far) or mp["bad"][i]:
            continue
        lvl = int(mp["level"][i])
        r = np.float32(2.5) if float(mp["view_cos"][i]) > 0.998 else np.float32(4.0)
        if th != 1.0:
            r = np.float32(r * np.float32(th))
        r = np.float32(r * scale[lvl])
        c = _window(g, order, mp["u"][i], mp["v"][i], r, lvl - 1, lvl)
        c = c[occupied[c] == 0]
        if ur_f is not None and len(c):
            er = np.abs(np.float32(mp["ur"][i]) - ur_f[c])
            c = c[~((ur_f[c] > 0) & (er > r))]
        if not len(c):
            continue
        d = _hamming(mp["desc"][i], dF[c])
        # sequential best / second-best bookkeeping of :104-120 (the second best is NOT simply the second smallest)
        best = best2 = 256; lev = lev2 = -1; bi = -1
        for dist, j in zip(d, c):
            if dist < best:
                best2, best, lev2, lev, bi = best, dist, lev, int(g["octave"][j]), int(j)
            elif dist < best2:
                lev2, best2 = int(g["octave"][j]), dist
        if best <= 100:
            if lev == lev2 and np.float32(best) > np.float32(nnratio) * np.float32(best2):
                continue
            assign[bi] = i; occupied[bi] = mp["has_obs"][i]; nm += 1
    return nm


@pytest.mark.parametrize("frac,lw,th,ori", [(None, 0, 15.0, True), (0.0, 0, 15.0, True), (0.5, 0, 15.0, True), (1.0, 0, 7.0, True),
                                            (0.5, 1, 15.0, True), (0.5, 2, 15.0, False), (1.0, 1, 7.0, True), (1.0, 2, 15.0, True)])
def test_last_frame_search_vs_numpy(oracle, sm, frac, lw, th, ori):
    g, dF, angF, scale, last, assign, occ = sm.make_last_frame_case(3, n=500, n_last=400, stereo_frac=frac, level_window=lw)
    a0, o0 = assign.copy(), occ.copy()
    n0 = oracle.search_by_projection_last(g, dF, angF, scale, last, th, ori, a0, o0)
    a1, o1 = assign.copy(), occ.copy()
    n1 = numpy_search_last(g, dF, angF, scale, last, th, ori, a1, o1)
    assert n0 == n1 and n0 > 30
    np.testing.assert_array_equal(a0, a1); np.testing.assert_array_equal(o0, o1)


@pytest.mark.parametrize("frac,th,far", [(None, 3.0, False), (0.0, 3.0, False), (0.5, 1.0, False), (0.5, 3.0, True), (1.0, 3.0, False)])
def test_map_point_search_vs_numpy(oracle, sm, frac, th, far):
    g, dF, angF, scale, mp, assign, occ = sm.make_projection_case(5, n=500, n_mp=400, stereo_frac=frac)
    a0, o0 = assign.copy(), occ.copy()
    n0 = oracle.search_by_projection(g, dF, scale, mp, th, 0.8, a0, o0, b_far=far, th_far=20.0)
    a1, o1 = assign.copy(), occ.copy()
    n1 = numpy_search_mp(g, dF, scale, mp, th, 0.8, a1, o1, far, 20.0)
    assert n0 == n1 and n0 > 30
    np.testing.assert_array_equal(a0, a1); np.testing.assert_array_equal(o0, o1)


def test_stereo_gate_properties(oracle, sm):
    """What the gates mean: u_right all < 0 is the monocular search; a full stereo frame loses exactly the candidates
    whose right columns disagree; the level windows bound the octave of every match."""
    base = sm.make_last_frame_case(7, n=600, n_last=500)
    g, dF, angF, scale, last, assign, occ = base
    a_m, o_m = assign.copy(), occ.copy()
    n_m = oracle.search_by_projection_last(g, dF, angF, scale, last, 15.0, True, a_m, o_m)
    g0, _, _, _, last0, _, _ = sm.make_last_frame_case(7, n=600, n_last=500, stereo_frac=0.0)
    assert (g0["u_right"] < 0).all()
    a0, o0 = assign.copy(), occ.copy()
    assert oracle.search_by_projection_last(g0, dF, angF, scale, last0, 15.0, True, a0, o0) == n_m
    np.testing.assert_array_equal(a0, a_m)
    for lw in (1, 2):
        g1, _, _, _, last1, _, _ = sm.make_last_frame_case(7, n=600, n_last=500, stereo_frac=1.0, level_window=lw)
        a1, o1 = assign.copy(), occ.copy()
        n1 = oracle.search_by_projection_last(g1, dF, angF, scale, last1, 15.0, False, a1, o1)
        j = np.nonzero(a1 >= 0)[0]
        assert n1 >= len(j) > 30 and n1 != n_m       # a feature whose point has no observations can be taken again (:1746-1748)
        i = a1[j]
        r = np.float32(15.0) * scale[last1["octave"][i]]
        assert (np.abs(last1["ur"][i] - g1["u_right"][j]) <= r).all()
        if lw == 1:
            assert (g1["octave"][j] >= last1["octave"][i]).all()
        else:
            assert (g1["octave"][j] <= last1["octave"][i]).all()
