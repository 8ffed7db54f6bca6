"""Synthetic inputs for the projection searches (ORBmatcher::SearchByProjection variants).

A frame with N keypoints (grid 64x48 over the image bounds, octaves 0..7) and a set of map points / last-frame
points that project close to some of them with descriptors a few bits away, plus distractors, occupied features
and ties, so that the ratio test, the level bookkeeping, the 'already holds a map point' skip and the rotation
histogram are all exercised."""
import numpy as np


def _flip(rs, d, p):
    bits = np.unpackbits(d, axis=-1)
    fl = (rs.uniform(size=bits.shape) < p).astype(np.uint8)
    return np.packbits(bits ^ fl, axis=-1)


def make_frame_features(seed, n=1000, w=752, h=480, nlevels=8, cluster=True):
    rs = np.random.RandomState(31337 + seed)
    x = rs.uniform(0, w - 1, n)
    y = rs.uniform(0, h - 1, n)
    if cluster:     # some crowded cells and exact duplicates of positions
        k = n // 5
        x[:k] = rs.normal(w * 0.4, 12.0, k).clip(0, w - 1)
        y[:k] = rs.normal(h * 0.5, 10.0, k).clip(0, h - 1)
    scale = (1.2 ** np.arange(nlevels)).astype(np.float32)
    g = dict(x=x.astype(np.float32), y=y.astype(np.float32), octave=rs.randint(0, nlevels, n).astype(np.int32),
             min_x=0.0, min_y=0.0, max_x=float(w), max_y=float(h), cols=64, rows=48)
    desc = rs.randint(0, 256, size=(n, 32)).astype(np.uint8)
    ang = rs.uniform(0, 360, n).astype(np.float32)
    return g, desc, ang, scale


def _add_stereo(seed, g, tgt, u, pts, stereo_frac):
    """Rectified-stereo side of a tracking search (reference src/ORBmatcher.cc:92-98, :1751-1757): a share of the frame's
    features gets a right-image column (mvuRight > 0, the rest -1) and every point a predicted one (mTrackProjXR /
    uv(0) - mbf*invzc) that lands on both sides of the window radius.  Drawn from its own stream: the monocular arrays of
    the case stay what they are without it."""
    rs = np.random.RandomState(86420 + seed)
    n = len(g["x"])
    g["u_right"] = np.where(rs.uniform(size=n) < stereo_frac, np.maximum(g["x"] - rs.uniform(2, 40, n), 0.5), -1.0).astype(np.float32)
    err = rs.normal(0, 1.0, len(tgt)) * np.where(rs.uniform(size=len(tgt)) < 0.6, 2.0, 15.0)
    ur = np.where(g["u_right"][tgt] > 0, g["u_right"][tgt] + err, u - rs.uniform(2, 40, len(tgt)))
    pts["ur"] = ur.astype(np.float32)


def make_projection_case(seed, n=1000, n_mp=900, nlevels=8, stereo_frac=None):
    """Inputs of SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) (reference src/ORBmatcher.cc:43-213);
    stereo_frac: share of the frame's features with a right-image coordinate (None = monocular frame, no mvuRight)."""
    rs = np.random.RandomState(777 + seed)
    g, dF, angF, scale = make_frame_features(seed, n, nlevels=nlevels)
    tgt = rs.randint(0, n, n_mp)
    kind = rs.uniform(size=n_mp)
    u = g["x"][tgt] + rs.normal(0, 1.5, n_mp)
    v = g["y"][tgt] + rs.normal(0, 1.5, n_mp)
    far = kind > 0.9                                       # project to nowhere in particular
    u[far] = rs.uniform(-20, g["max_x"] + 20, far.sum())
    v[far] = rs.uniform(-20, g["max_y"] + 20, far.sum())
    level = np.clip(g["octave"][tgt] + rs.randint(0, 2, n_mp), 0, nlevels - 1).astype(np.int32)
    desc = _flip(rs, dF[tgt], np.where(kind < 0.6, 0.04, 0.25)[:, None])
    dup = rs.uniform(size=n_mp) < 0.05                     # exact copies -> distance ties between candidates
    desc[dup] = dF[tgt[dup]]
    mp = dict(u=u.astype(np.float32), v=v.astype(np.float32), level=level,
              in_view=(rs.uniform(size=n_mp) < 0.92).astype(np.uint8),
              view_cos=np.where(rs.uniform(size=n_mp) < 0.5, 0.9995, 0.97).astype(np.float32),
              depth=rs.uniform(1, 40, n_mp).astype(np.float32), desc=np.ascontiguousarray(desc),
              has_obs=(rs.uniform(size=n_mp) < 0.9).astype(np.uint8), bad=(rs.uniform(size=n_mp) < 0.03).astype(np.uint8))
    occupied = (rs.uniform(size=n) < 0.1).astype(np.uint8)
    assign = np.where(occupied > 0, 100000 + np.arange(n), -1).astype(np.int32)
    if stereo_frac is not None:
        _add_stereo(seed, g, tgt, u, mp, stereo_frac)
    return g, dF, angF, scale, mp, assign, occupied


def make_last_frame_case(seed, n=1000, n_last=900, nlevels=8, stereo_frac=None, level_window=0):
    """Inputs of SearchByProjection(Frame& cur, const Frame& last, th, bMono) (reference src/ORBmatcher.cc:1676-1887);
    stereo_frac as above, level_window 1 = bForward / 2 = bBackward (:1692-1693; only a stereo caller, !bMono, sets them)."""
    rs = np.random.RandomState(999 + seed)
    g, dF, angF, scale = make_frame_features(seed + 50, n, nlevels=nlevels)
    tgt = rs.randint(0, n, n_last)
    u = g["x"][tgt] + rs.normal(0, 3.0, n_last)
    v = g["y"][tgt] + rs.normal(0, 3.0, n_last)
    out = rs.uniform(size=n_last) < 0.05
    u[out] = rs.choice([-5.0, g["max_x"] + 5.0], out.sum())        # outside the image bounds check
    octave = np.clip(g["octave"][tgt] + rs.randint(-1, 2, n_last), 0, nlevels - 1).astype(np.int32)
    desc = _flip(rs, dF[tgt], np.where(rs.uniform(size=n_last) < 0.7, 0.05, 0.3)[:, None])
    true_rot = rs.uniform(size=n_last) < 0.75
    ang = np.mod(angF[tgt].astype(np.float64) + np.where(true_rot, rs.normal(20, 4, n_last), rs.uniform(0, 360, n_last)), 360.0)
    last = dict(u=u.astype(np.float32), v=v.astype(np.float32), octave=octave, angle=ang.astype(np.float32),
                valid=(rs.uniform(size=n_last) < 0.85).astype(np.uint8), desc=np.ascontiguousarray(desc),
                has_obs=(rs.uniform(size=n_last) < 0.8).astype(np.uint8))
    occupied = np.zeros(n, np.uint8)
    assign = np.full(n, -1, np.int32)
    if stereo_frac is not None:
        _add_stereo(seed + 1000, g, tgt, u, last, stereo_frac)
    if level_window:
        last["level_window"] = int(level_window)
    return g, dF, angF, scale, last, assign, occupied


def make_kf_projection_case(seed, n=1000, n_pts=900, nlevels=8):
    """Inputs of SearchByProjection(Frame& cur, KeyFrame*, sAlreadyFound, th, ORBdist) (reference src/ORBmatcher.cc:1889-2010):
    the last-frame case with predicted levels, some features of the current frame already holding a map point."""
    g, dF, angF, scale, last, assign, occupied = make_last_frame_case(seed + 7, n, n_pts, nlevels)
    rs = np.random.RandomState(4242 + seed)
    occupied = (rs.uniform(size=n) < 0.15).astype(np.uint8)
    assign = np.where(occupied > 0, 100000 + np.arange(n), -1).astype(np.int32)
    pts = dict(u=last["u"], v=last["v"], level=last["octave"], angle=last["angle"], valid=last["valid"], desc=last["desc"])
    return g, dF, angF, scale, pts, assign, occupied


def make_fuse_case(seed, n=1000, n_pts=3000, nlevels=8, stereo_frac=0.4):
    """Inputs of the search core of ORBmatcher::Fuse (reference src/ORBmatcher.cc:1148-1338 / :1340-1455): a key frame
    (features, mvuRight, mvInvLevelSigma2) and candidate map points projected near some of its features."""
    rs = np.random.RandomState(5150 + seed)
    g, dKF, angKF, scale = make_frame_features(seed + 90, n, nlevels=nlevels)
    u_right = np.where(rs.uniform(size=n) < stereo_frac, g["x"] - rs.uniform(2, 40, n), -1.0).astype(np.float32)
    inv_sigma2 = (1.0 / (scale.astype(np.float64) ** 2)).astype(np.float32)
    tgt = rs.randint(0, n, n_pts)
    # mostly sub-pixel .. few-pixel reprojection errors so that the chi2 gate (5.99 / 7.8 sigma^2) cuts both ways
    err = rs.normal(0, 1.0, (n_pts, 3)) * np.where(rs.uniform(size=n_pts) < 0.7, 1.0, 4.0)[:, None]
    u = g["x"][tgt] + err[:, 0]; v = g["y"][tgt] + err[:, 1]
    ur = np.where(u_right[tgt] >= 0, u_right[tgt] + err[:, 2], u - rs.uniform(2, 40, n_pts))
    level = np.clip(g["octave"][tgt] + rs.randint(0, 2, n_pts), 0, nlevels - 1).astype(np.int32)
    desc = _flip(rs, dKF[tgt], np.where(rs.uniform(size=n_pts) < 0.6, 0.04, 0.25)[:, None])
    dup = rs.uniform(size=n_pts) < 0.05
    desc[dup] = dKF[tgt[dup]]
    pts = dict(u=u.astype(np.float32), v=v.astype(np.float32), ur=ur.astype(np.float32), level=level,
               valid=(rs.uniform(size=n_pts) < 0.9).astype(np.uint8), desc=np.ascontiguousarray(desc))
    return g, dKF, scale, u_right, inv_sigma2, pts


def make_triangulation_case(seed, n=1000, nlevels=8):
    """Inputs of ORBmatcher::SearchForTriangulation (reference src/ORBmatcher.cc:907-1146): two key frames whose unmatched
    features share vocabulary nodes; a sideways-translation fundamental matrix (epipolar lines = image rows) so that the
    3.84 sigma^2 gate cuts both ways; an epipole inside the image; some features already hold map points."""
    from . import synth
    rs = np.random.RandomState(6007 + seed)
    tree = synth.make_tree(seed)
    g1, d1, a1, scale = make_frame_features(seed + 200, n, nlevels=nlevels, cluster=False)
    perm = rs.permutation(n)
    d2 = _flip(rs, d1[perm], np.where(rs.uniform(size=n) < 0.6, 0.03, 0.3)[:, None])
    x2 = (g1["x"][perm] - rs.uniform(5, 60, n)).astype(np.float32)
    y2 = (g1["y"][perm] + rs.normal(0, 1.2, n) * scale[g1["octave"][perm]]).astype(np.float32)
    oct2 = np.clip(g1["octave"][perm] + rs.randint(-1, 2, n), 0, nlevels - 1).astype(np.int32)
    a2 = np.mod(a1[perm] + np.where(rs.uniform(size=n) < 0.8, rs.normal(10, 3, n), rs.uniform(0, 360, n)), 360).astype(np.float32)
    side = []
    for d, x, y, o, a in ((d1, g1["x"], g1["y"], g1["octave"], a1), (d2, x2, y2, oct2, a2)):
        nodes, off, feat = synth.feature_vector(synth.assign_nodes(d, tree))
        side.append(dict(desc=np.ascontiguousarray(d), x=np.ascontiguousarray(x), y=np.ascontiguousarray(y), octave=np.ascontiguousarray(o),
                         angle=np.ascontiguousarray(a), has_mp=(rs.uniform(size=n) < 0.3).astype(np.uint8),
                         stereo=(rs.uniform(size=n) < 0.4).astype(np.uint8), fv=(nodes, off, feat)))
    F12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32) * np.float32(0.7)
    sigma2 = (scale.astype(np.float64) ** 2).astype(np.float32)
    return side[0], side[1], (np.float32(300.0), np.float32(240.0)), F12, sigma2, scale


def make_initialization_case(seed, n=2000, nlevels=8):
    """Inputs of ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cc:648-763): F2 = F1 moved by a few pixels,
    clustered so that several F1 features compete for (and steal) the same F2 feature."""
    rs = np.random.RandomState(7331 + seed)
    g1, d1, a1, scale = make_frame_features(seed + 300, n, nlevels=nlevels, cluster=True)
    oct1 = np.where(rs.uniform(size=n) < 0.6, 0, g1["octave"]).astype(np.int32)
    perm = rs.permutation(n)
    g2 = dict(g1)
    g2["x"] = (g1["x"][perm] + rs.normal(6, 8, n)).clip(0, g1["max_x"] - 1).astype(np.float32)
    g2["y"] = (g1["y"][perm] + rs.normal(0, 8, n)).clip(0, g1["max_y"] - 1).astype(np.float32)
    g2["octave"] = np.where(rs.uniform(size=n) < 0.8, oct1[perm], 1).astype(np.int32)
    d2 = _flip(rs, d1[perm], np.where(rs.uniform(size=n) < 0.6, 0.03, 0.2)[:, None])
    dup = rs.uniform(size=n) < 0.1
    d2[dup] = d1[perm][dup]
    a2 = np.mod(a1[perm] + np.where(rs.uniform(size=n) < 0.8, rs.normal(5, 3, n), rs.uniform(0, 360, n)), 360).astype(np.float32)
    f1 = dict(desc=np.ascontiguousarray(d1), octave=oct1, angle=a1, prev_x=g1["x"].copy(), prev_y=g1["y"].copy())
    return f1, g2, np.ascontiguousarray(d2), a2, scale
