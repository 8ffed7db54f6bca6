"""Deterministic synthetic inputs for the hot path.

No dataset ships with the reference (EuRoC images / ORBvoc.txt are listed in
.MISSING_LARGE_BLOBS), so tests and bench.py use these seeded generators.
Pure numpy; identical output on every machine (legacy RandomState streams).
"""
import numpy as np


def _box3(a):
    """3x3 box filter with edge replication, float32."""
    p = np.pad(a, 1, mode="edge")
    acc = np.zeros_like(a, dtype=np.float32)
    for dy in range(3):
        for dx in range(3):
            acc += p[dy:dy + a.shape[0], dx:dx + a.shape[1]]
    return acc / 9.0


def make_frame(seed, w=640, h=480, n_rect=400, strip=64):
    """640x480 u8 mono frame: band-limited noise + random grey rectangles + a flat strip.

    The mix yields corners at both FAST thresholds (20 and 7) and some empty
    cells (the flat strip forces the minThFAST fallback, reference
    src/ORBextractor.cc:843-846)."""
    rs = np.random.RandomState(1000003 * (seed + 1) % (2 ** 31 - 1))
    noise = rs.randint(0, 256, size=(h, w)).astype(np.float32)
    for _ in range(3):
        noise = _box3(noise)
    img = 128.0 + (noise - 127.5) * 1.6
    n_rect = int(n_rect * (w * h) / (640.0 * 480.0)) if (w, h) != (640, 480) else n_rect
    for _ in range(max(n_rect, 4)):
        rw = int(rs.randint(6, max(8, w // 6)))
        rh = int(rs.randint(6, max(8, h // 6)))
        x0 = int(rs.randint(-rw // 2, w - rw // 2))
        y0 = int(rs.randint(-rh // 2, h - rh // 2))
        g = float(rs.randint(10, 246))
        a = float(rs.uniform(0.35, 1.0))
        xs, xe = max(x0, 0), min(x0 + rw, w)
        ys, ye = max(y0, 0), min(y0 + rh, h)
        img[ys:ye, xs:xe] = (1 - a) * img[ys:ye, xs:xe] + a * g
    # low-contrast texture so that some cells only fire at minThFAST
    img += rs.normal(0.0, 2.0, size=(h, w)).astype(np.float32)
    if strip > 0 and w > 3 * strip:
        sx = int(rs.randint(w // 4, w // 2))
        img[:, sx:sx + strip] = 97.0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_frames(batch, w=640, h=480, seed0=0):
    return np.stack([make_frame(seed0 + i, w, h) for i in range(batch)])


# ---------------------------------------------------------------- matcher inputs
def _popcount8(a):
    return np.unpackbits(a, axis=-1).sum(-1)


def make_tree(seed, k=10):
    """Synthetic 2-level k-ary vocabulary tree: (k level-1 centroids, k*k level-2 centroids).

    Stand-in for Vocabulary/ORBvoc.txt (not shipped).  Level-2 node j*k+i is
    child i of level-1 node j."""
    rs = np.random.RandomState(7919 + seed)
    l1 = rs.randint(0, 256, size=(k, 32)).astype(np.uint8)
    l2 = rs.randint(0, 256, size=(k * k, 32)).astype(np.uint8)
    return l1, l2


def assign_nodes(desc, tree):
    """Descend the tree with the DBoW2 rule (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1232-1254):
    at each level pick the child with the smallest Hamming distance, first minimum wins."""
    l1, l2 = tree
    k = l1.shape[0]
    d1 = _popcount8(desc[:, None, :] ^ l1[None, :, :])
    c1 = d1.argmin(1)
    ch = l2.reshape(k, k, 32)[c1]
    d2 = _popcount8(desc[:, None, :] ^ ch)
    c2 = d2.argmin(1)
    return (c1 * k + c2).astype(np.uint32)


def feature_vector(node_of_feature):
    """FeatureVector CSR (node ids ascending; inside a node feature indices ascending = insertion order,
    reference Thirdparty/DBoW2/DBoW2/FeatureVector.cpp:31-47)."""
    nodes = np.unique(node_of_feature)
    offs = [0]
    feats = []
    for nd in nodes:
        idx = np.nonzero(node_of_feature == nd)[0]
        feats.append(idx)
        offs.append(offs[-1] + len(idx))
    feat = np.concatenate(feats).astype(np.uint32) if feats else np.zeros(0, np.uint32)
    return nodes.astype(np.uint32), np.asarray(offs, np.int32), feat


def make_match_set(seed, n=1000, p_true=0.7, k=10):
    """Two descriptor sets (KF, F) of n x 32 B."""
    rs = np.random.RandomState(4241 + seed)
    dF = rs.randint(0, 256, size=(n, 32)).astype(np.uint8)
    perm = rs.permutation(n)
    is_true = rs.uniform(size=n) < p_true
    flip_p = np.where(is_true, 0.02, 0.5)
    bits = np.unpackbits(dF[perm], axis=1)
    flips = rs.uniform(size=bits.shape) < flip_p[:, None]
    dKF = np.packbits(bits ^ flips.astype(np.uint8), axis=1)
    angF = rs.uniform(0, 360, size=n).astype(np.float32)
    off = np.where(is_true, rs.normal(15.0, 3.0, size=n), rs.uniform(0, 360, size=n))
    angKF = np.mod(angF[perm].astype(np.float64) + off, 360.0).astype(np.float32)
    validKF = (rs.uniform(size=n) < 0.9).astype(np.uint8)
    tree = make_tree(seed, k)
    fvKF = feature_vector(assign_nodes(dKF, tree))
    fvF = feature_vector(assign_nodes(dF, tree))
    return dict(dKF=dKF, dF=dF, angKF=angKF, angF=angF, validKF=validKF, fvKF=fvKF, fvF=fvF, perm=perm, is_true=is_true)


# ---------------------------------------------------------------- BA window
def _quat_from_R(R):
    """Rotation matrix -> (x, y, z, w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        v = np.zeros(3)
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        q = np.array([v[0], v[1], v[2], (R[k, j] - R[j, k]) / s])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_ba_window(seed, n_opt=50, n_fixed=10, n_points=2000, obs_per_point=10, stereo_frac=0.0,
                   outlier_frac=0.03):
    """Synthetic LocalBA window: poses on a smooth arc looking at a point cloud in a
    10x10x4 m box at 3-15 m depth, EuRoC pinhole intrinsics (reference Examples/Monocular/EuRoC.yaml:23-26),
    pixel noise by octave, 3 % gross outliers, perturbed initial estimates.  All inputs are rounded through
    float32 as the reference does (src/Optimizer.cc:1217-1218,1286,1309,1316)."""
    rs = np.random.RandomState(9001 + seed)
    fx, fy, cx, cy = [float(np.float32(v)) for v in (458.654, 457.296, 367.215, 248.375)]
    bf = float(np.float32(47.90639384423901))
    n_poses = n_opt + n_fixed
    pts = np.stack([rs.uniform(-5, 5, n_points), rs.uniform(-2, 2, n_points), rs.uniform(3, 15, n_points)], 1)
    Rs, ts = [], []
    for i in range(n_poses):
        a = (i / max(n_poses - 1, 1) - 0.5) * 0.6          # yaw sweep +-0.3 rad
        c = np.array([3.0 * np.sin(a * 2), 0.2 * np.sin(i * 0.7), -1.0 + 0.5 * np.cos(a * 2)])   # camera centre
        Rwc = _rodrigues(np.array([0.0, a, 0.0])) @ _rodrigues(np.array([0.02 * np.sin(i), 0, 0.01 * np.cos(i)]))
        Rcw = Rwc.T
        Rs.append(Rcw)
        ts.append(-Rcw @ c)
    Rs = np.array(Rs)
    ts = np.array(ts)
    scale2 = 1.2 ** (2 * np.arange(8))
    e_pt, e_pose, e_obs, e_w, e_st = [], [], [], [], []
    for l in range(n_points):
        Xc = (Rs @ pts[l]) + ts                             # n_poses x 3
        u = fx * Xc[:, 0] / Xc[:, 2] + cx
        v = fy * Xc[:, 1] / Xc[:, 2] + cy
        vis = np.nonzero((Xc[:, 2] > 0.5) & (u > 0) & (u < 752) & (v > 0) & (v < 480))[0]
        if len(vis) == 0:
            vis = np.array([int(np.argmax(Xc[:, 2]))])
        sel = rs.permutation(vis)[:obs_per_point]
        sel.sort()
        for ip in sel:
            octv = int(rs.randint(0, 8))
            sig = np.sqrt(scale2[octv])
            uu = u[ip] + rs.normal(0, sig)
            vv = v[ip] + rs.normal(0, sig)
            if rs.uniform() < outlier_frac:
                uu += rs.choice([-30.0, 30.0])
                vv += rs.choice([-30.0, 30.0])
            st = rs.uniform() < stereo_frac
            ur = uu - bf / Xc[ip, 2] + (rs.normal(0, sig) if st else 0.0)
            e_pt.append(l); e_pose.append(int(ip))
            e_obs.append([float(np.float32(uu)), float(np.float32(vv)), float(np.float32(ur)) if st else -1.0])
            e_w.append(float(np.float32(1.0 / scale2[octv])))
            e_st.append(1 if st else 0)
    # perturbed initial estimates (1 deg / 2 cm poses, 3 cm points), through float32
    q0 = np.zeros((n_poses, 4)); t0 = np.zeros((n_poses, 3))
    fixed = np.zeros(n_poses, np.uint8)
    fixed[n_opt:] = 1
    for i in range(n_poses):
        if fixed[i]:
            R, t = Rs[i], ts[i]
        else:
            dR = _rodrigues(rs.normal(0, np.deg2rad(1.0) / np.sqrt(3), 3))
            R = dR @ Rs[i]
            t = dR @ ts[i] + rs.normal(0, 0.02 / np.sqrt(3), 3)
        q = _quat_from_R(R).astype(np.float32).astype(np.float64)
        q0[i] = q
        t0[i] = t.astype(np.float32).astype(np.float64)
    p0 = (pts + rs.normal(0, 0.03 / np.sqrt(3), pts.shape)).astype(np.float32).astype(np.float64)
    th_mono = float(np.float32(np.sqrt(5.991)))
    th_stereo = float(np.float32(np.sqrt(7.815)))
    return dict(pose_q=q0, pose_t=t0, pose_fixed=fixed, points=p0,
                edge_point=np.asarray(e_pt, np.int32), edge_pose=np.asarray(e_pose, np.int32),
                edge_obs=np.asarray(e_obs, np.float64), edge_inv_sigma2=np.asarray(e_w, np.float64),
                edge_stereo=np.asarray(e_st, np.uint8), fx=fx, fy=fy, cx=cx, cy=cy, bf=bf,
                huber_mono=th_mono, huber_stereo=th_stereo,
                true_R=Rs, true_t=ts, true_points=pts)


# ---------------------------------------------------------------- motion-only BA (PoseOptimization)
def make_pose_problem(seed, n=300, outlier_frac=0.1, stereo_frac=0.0):
    """One frame for Optimizer::PoseOptimization (reference src/Optimizer.cc:814-1115): n features holding map points,
    EuRoC pinhole intrinsics, pixel noise by octave, gross outliers, initial pose a few degrees / centimetres off."""
    rs = np.random.RandomState(5151 + seed)
    fx, fy, cx, cy = [float(np.float32(v)) for v in (458.654, 457.296, 367.215, 248.375)]
    bf = float(np.float32(47.90639384423901))
    R = _rodrigues(rs.normal(0, 0.2, 3))
    t = rs.normal(0, 0.5, 3)
    Xc = np.stack([rs.uniform(-4, 4, n), rs.uniform(-2.5, 2.5, n), rs.uniform(2, 14, n)], 1)
    Xw = (Xc - t) @ R            # R^T (Xc - t)
    scale2 = 1.2 ** (2 * np.arange(8))
    octv = rs.randint(0, 8, n)
    sig = np.sqrt(scale2[octv])
    u = fx * Xc[:, 0] / Xc[:, 2] + cx + rs.normal(0, 1, n) * sig
    v = fy * Xc[:, 1] / Xc[:, 2] + cy + rs.normal(0, 1, n) * sig
    out = rs.uniform(size=n) < outlier_frac
    u[out] += rs.choice([-40.0, 40.0], out.sum())
    v[out] += rs.choice([-25.0, 25.0], out.sum())
    st = (rs.uniform(size=n) < stereo_frac)
    ur = np.where(st, u - bf / Xc[:, 2] + rs.normal(0, 1, n) * sig, -1.0)
    dR = _rodrigues(rs.normal(0, np.deg2rad(2.0) / np.sqrt(3), 3))
    q0 = _quat_from_R(dR @ R).astype(np.float32).astype(np.float64)
    t0 = (dR @ t + rs.normal(0, 0.05 / np.sqrt(3), 3)).astype(np.float32).astype(np.float64)
    obs = np.stack([u, v, ur], 1).astype(np.float32).astype(np.float64)
    return dict(q=q0, t=t0, Xw=np.ascontiguousarray(Xw.astype(np.float32).astype(np.float64)), obs=np.ascontiguousarray(obs),
                inv_sigma2=(1.0 / scale2[octv]).astype(np.float32).astype(np.float64), stereo=st.astype(np.uint8),
                fx=fx, fy=fy, cx=cx, cy=cy, bf=bf, huber_mono=float(np.float32(np.sqrt(5.991))),
                huber_stereo=float(np.float32(np.sqrt(7.815))), true_R=R, true_t=t, is_outlier=out)


def make_vocabulary(seed, k=10, L=3, ragged=True, tie_frac=0.05, stop_frac=0.02, shuffle_ids=False):
    """Synthetic DBoW2 vocabulary tree in the flattened form of include/orbslam3_hip.h (stand-in for ORBvoc.txt: k=10, L=6,
    which the reference does not ship).  Children are noisy copies of their parent (what hierarchical k-medians produces);
    `ragged` removes some children and ends some branches early; `tie_frac` duplicates sibling centroids (first-minimum
    tie-break); `stop_frac` gives some words weight 0 (stopped words are skipped by transform)."""
    rs = np.random.RandomState(8191 + seed)
    parents = [0]
    desc = [np.zeros(32, np.uint8)]
    depth = [0]
    children = [[]]
    frontier = [0]
    for d in range(1, L + 1):
        nxt = []
        for p in frontier:
            if ragged and d >= 2 and rs.uniform() < 0.04:
                continue                                        # early leaf
            nc = k if not ragged else int(rs.randint(max(1, k - 3), k + 1))
            base = np.unpackbits(desc[p])
            for c in range(nc):
                fl = (rs.uniform(size=256) < (0.5 if d == 1 else 0.18)).astype(np.uint8)
                dd = np.packbits(base ^ fl)
                if c > 0 and rs.uniform() < tie_frac:
                    dd = desc[children[p][-1]].copy()           # identical to the previous sibling
                nid = len(desc)
                desc.append(dd); parents.append(p); depth.append(d); children.append([])
                children[p].append(nid)
                nxt.append(nid)
        frontier = nxt
    n = len(desc)
    perm = np.arange(n)
    if shuffle_ids:                                             # node ids need not be in creation order
        perm[1:] = 1 + rs.permutation(n - 1)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)       # new id -> old id
    child_off = np.zeros(n + 1, np.int32)
    child_id = []
    for new in range(n):
        ch = [perm[c] for c in children[inv[new]]]
        child_off[new + 1] = child_off[new] + len(ch)
        child_id += ch
    desc_a = np.stack(desc)[inv]
    is_leaf = np.diff(child_off) == 0
    word_id = np.full(n, -1, np.int32)
    word_id[is_leaf] = np.arange(int(is_leaf.sum()))
    weight = np.zeros(n, np.float64)
    weight[is_leaf] = rs.uniform(0.3, 9.0, int(is_leaf.sum()))
    stopped = is_leaf & (rs.uniform(size=n) < stop_frac)
    weight[stopped] = 0.0
    return dict(n_nodes=n, L=L, k=k, child_off=child_off, child_id=np.asarray(child_id, np.uint32),
                desc=np.ascontiguousarray(desc_a), weight=weight, word_id=word_id)


def make_vocabulary_fast(seed, k=10, L=6):
    """Complete k-ary tree with random centroids, built with array operations only (the k=10, L=6 ORBvoc shape has
    1 111 111 nodes / 35 MB of centroids): node ids in breadth-first order, children of node i are k*i+1 .. k*i+k."""
    rs = np.random.RandomState(1021 + seed)
    n = (k ** (L + 1) - 1) // (k - 1)
    n_inner = (k ** L - 1) // (k - 1)
    child_off = np.minimum(np.arange(n + 1, dtype=np.int64), n_inner) * k
    child_id = np.arange(1, n, dtype=np.uint32)
    desc = rs.randint(0, 256, size=(n, 32), dtype=np.uint8)
    weight = np.zeros(n, np.float64)
    weight[n_inner:] = rs.uniform(0.3, 9.0, n - n_inner)
    word_id = np.full(n, -1, np.int32)
    word_id[n_inner:] = np.arange(n - n_inner)
    return dict(n_nodes=n, L=L, k=k, child_off=child_off.astype(np.int32), child_id=child_id, desc=desc, weight=weight, word_id=word_id)


def make_stereo_pair(seed, w=640, h=480, band=60, dmin=3, dmax=40):
    """Rectified stereo pair for Frame::ComputeStereoMatches: the right image is the left one with a different integer
    disparity per horizontal band (fronto-parallel 'objects' at different depths)."""
    left = make_frame(seed, w, h)
    right = np.zeros_like(left)
    rs = np.random.RandomState(4409 + seed)
    for y0 in range(0, h, band):
        d = int(rs.randint(dmin, dmax + 1))
        right[y0:y0 + band, :w - d] = left[y0:y0 + band, d:]
        right[y0:y0 + band, w - d:] = left[y0:y0 + band, w - d:]
    return left, right


def _so3_exp(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        return np.eye(3) + W
    return np.eye(3) + W * (np.sin(th) / th) + W @ W * ((1 - np.cos(th)) / th ** 2)


def make_inertial_window(seed, n_opt=6, n_points=150, obs_per_point=4, dt=0.25, noise_px=0.5, perturb=True, bias_error=0.0, stereo_frac=0.0,
                         n_covisible_fixed=0):
    """A LocalInertialBA window (reference src/Optimizer.cc:2383-2958) with consistent synthetic data: key frame 0 is the fixed
    one in front of the temporal window (pose and IMU states fixed), key frames 1..n_opt are optimised; every consecutive pair
    is linked by a pre-integrated measurement computed from the ground-truth states (plus noise), so that the ground truth is a
    minimum of the inertial cost; mono observations of a point cloud.  Returns (problem dict, ground truth dict)."""
    rs = np.random.RandomState(seed)
    n = n_opt + 1
    g = np.array([0.0, 0.0, -9.81])
    # smooth body trajectory: constant-acceleration segments
    Rwb = [np.eye(3)]; pwb = [np.zeros(3)]; vel = [np.array([0.6, 0.1, 0.0])]
    acc_w = rs.normal(0, 0.4, (n, 3)); omg = rs.normal(0, 0.15, (n, 3))
    for i in range(1, n):
        Rwb.append(Rwb[-1] @ _so3_exp(omg[i] * dt))
        pwb.append(pwb[-1] + vel[-1] * dt + 0.5 * acc_w[i] * dt * dt)
        vel.append(vel[-1] + acc_w[i] * dt)
    bg_true = rs.normal(0, 0.01, 3); ba_true = rs.normal(0, 0.05, 3)
    Rcb = _so3_exp(np.array([0.01, -0.02, 0.015])) @ np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0.0]])   # camera looks along the body x axis
    tcb = np.array([0.02, -0.01, 0.03]); tbc = -Rcb.T @ tcb
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    links = []
    for i in range(1, n):
        R1, R2 = Rwb[i - 1], Rwb[i]
        dR = R1.T @ R2
        dV = R1.T @ (vel[i] - vel[i - 1] - g * dt)
        dP = R1.T @ (pwb[i] - pwb[i - 1] - vel[i - 1] * dt - 0.5 * g * dt * dt)
        sig_r, sig_v, sig_p = 2e-3, 1e-2, 5e-3
        dR = dR @ _so3_exp(rs.normal(0, sig_r * 0.2, 3)); dV = dV + rs.normal(0, sig_v * 0.2, 3); dP = dP + rs.normal(0, sig_p * 0.2, 3)
        info9 = np.diag([1 / sig_r ** 2] * 3 + [1 / sig_v ** 2] * 3 + [1 / sig_p ** 2] * 3)
        last = i == 1                      # the link to the fixed key frame (i == N-1 in the reference's reversed ordering)
        if last:
            info9 = info9 * 1e-2
        links.append(dict(kf1=i - 1, kf2=i, dR=dR.astype(np.float32), dV=dV.astype(np.float32), dP=dP.astype(np.float32),
                          JRg=(-dt * np.eye(3)).astype(np.float32), JVg=(rs.normal(0, 0.01, (3, 3))).astype(np.float32),
                          JVa=(-dt * dR).astype(np.float32), JPg=(rs.normal(0, 0.003, (3, 3))).astype(np.float32),
                          JPa=(-0.5 * dt * dt * dR).astype(np.float32), dT=np.float32(dt),
                          bias0=np.concatenate([ba_true, bg_true]).astype(np.float32),
                          info9=info9, info_gyro=np.eye(3) * 1e6 / dt, info_acc=np.eye(3) * 1e4 / dt, robust=np.uint8(last)))
    # points in front of the cameras
    centre = np.mean(pwb, axis=0)
    pts = centre + np.array([6.0, 0, 0]) + rs.uniform(-1, 1, (n_points, 3)) * np.array([2.0, 3.0, 2.0])
    e_kf, e_pt, e_obs, e_w, e_st = [], [], [], [], []
    bf = 47.9 if stereo_frac > 0 else 0.0
    for l in range(n_points):
        for i in sorted(rs.choice(n, min(obs_per_point, n), replace=False)):
            Rcw = Rcb @ Rwb[i].T; tcw = Rcb @ (-Rwb[i].T @ pwb[i]) + tcb
            Xc = Rcw @ pts[l] + tcw
            if Xc[2] < 0.5:
                continue
            octave = rs.randint(0, 4)
            sig = 1.2 ** octave
            uv = np.array([fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy]) + rs.normal(0, noise_px * sig, 2)
            st = rs.uniform() < stereo_frac
            ur = np.float32(uv[0] - bf / Xc[2] + rs.normal(0, noise_px * sig)) if st else -1.0
            e_kf.append(i); e_pt.append(l); e_obs.append([np.float32(uv[0]), np.float32(uv[1]), ur]); e_w.append(np.float32(1.0 / sig ** 2)); e_st.append(int(st))
    # extra fixed key frames that only observe points (lFixedKeyFrames from the covisibility graph: pose only, no IMU states)
    for c in range(n_covisible_fixed):
        Rc = Rwb[1 + c % n_opt] @ _so3_exp(rs.normal(0, 0.05, 3)); pc = pwb[1 + c % n_opt] + rs.normal(0, 0.3, 3)
        Rwb.append(Rc); pwb.append(pc); vel.append(np.zeros(3))
        for l in rs.choice(n_points, min(40, n_points), replace=False):
            Rcw = Rcb @ Rc.T; tcw = Rcb @ (-Rc.T @ pc) + tcb
            Xc = Rcw @ pts[l] + tcw
            if Xc[2] < 0.5:
                continue
            uv = np.array([fx * Xc[0] / Xc[2] + cx, fy * Xc[1] / Xc[2] + cy]) + rs.normal(0, noise_px, 2)
            e_kf.append(n + c); e_pt.append(int(l)); e_obs.append([np.float32(uv[0]), np.float32(uv[1]), -1.0]); e_w.append(np.float32(1.0)); e_st.append(0)
    # LocalInertialBA adds the edges per map point (Optimizer.cc:2720-2840): keep them grouped by point
    order = np.argsort(np.array(e_pt), kind="stable")
    e_kf = [e_kf[i] for i in order]; e_pt = [e_pt[i] for i in order]; e_obs = [e_obs[i] for i in order]; e_w = [e_w[i] for i in order]; e_st = [e_st[i] for i in order]
    nc = n_covisible_fixed
    gt = dict(Rwb=np.array(Rwb), twb=np.array(pwb), vel=np.array(vel), points=pts.copy())
    Rwb0 = np.array(Rwb); twb0 = np.array(pwb); vel0 = np.array(vel); pts0 = pts.copy()
    bg0 = np.tile(bg_true, (n + nc, 1)) + bias_error; ba0 = np.tile(ba_true, (n + nc, 1)) + bias_error
    if perturb:
        for i in range(1, n):
            Rwb0[i] = Rwb0[i] @ _so3_exp(rs.normal(0, 0.01, 3)); twb0[i] = twb0[i] + rs.normal(0, 0.02, 3); vel0[i] = vel0[i] + rs.normal(0, 0.03, 3)
        pts0 = pts0 + rs.normal(0, 0.03, pts0.shape)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)       # the reference loads float members into double vertices
    pr = dict(n_kf=n + nc, Rwb=f32(Rwb0), twb=f32(twb0), vel=f32(vel0), bg=f32(bg0), ba=f32(ba0),
              pose_fixed=np.array([1] + [0] * n_opt + [1] * nc, np.uint8), has_imu=np.array([1] * n + [0] * nc, np.uint8),
              imu_fixed=np.array([1] + [0] * n_opt + [1] * nc, np.uint8),
              Rcb=f32(Rcb), tcb=f32(tcb), tbc=f32(tbc), fx=float(np.float32(fx)), fy=float(np.float32(fy)), cx=float(np.float32(cx)), cy=float(np.float32(cy)), bf=float(np.float32(bf)),
              points=f32(pts0), edge_kf=np.array(e_kf, np.int32), edge_point=np.array(e_pt, np.int32), edge_obs=np.array(e_obs, np.float64),
              edge_inv_sigma2=np.array(e_w, np.float64), edge_stereo=np.array(e_st, np.uint8), links=links,
              huber_mono=float(np.float32(np.sqrt(5.991))), huber_stereo=float(np.float32(np.sqrt(7.815))), huber_inertial=float(np.sqrt(16.92)),
              lambda_init=1.0, max_iters=10)
    return pr, gt


def make_pose_inertial_problem(seed, n=300, outlier_frac=0.1, stereo_frac=0.0, noise_px=0.5, dt=0.05, perturb=True, last_frame=False):
    """The per-frame inertial optimisation (Optimizer::PoseInertialOptimizationLastKeyFrame, reference src/Optimizer.cc:4491-4873):
    the last key frame (fixed) and the current frame, linked by one pre-integrated measurement computed from the ground truth
    (plus noise); n map points seen by the frame with some gross outliers.  Returns (problem dict, ground truth dict)."""
    rs = np.random.RandomState(seed)
    g = np.array([0.0, 0.0, -9.81])
    R1 = _so3_exp(rs.normal(0, 0.2, 3)); p1 = rs.normal(0, 1.0, 3); v1 = np.array([0.8, 0.1, -0.05])
    acc = rs.normal(0, 0.5, 3); omg = rs.normal(0, 0.3, 3)
    R2 = R1 @ _so3_exp(omg * dt); p2 = p1 + v1 * dt + 0.5 * acc * dt * dt; v2 = v1 + acc * dt
    bgk = rs.normal(0, 0.01, 3); bak = rs.normal(0, 0.05, 3)
    Rcb = _so3_exp(np.array([0.01, -0.02, 0.015])) @ np.array([[0, -1, 0], [0, 0, -1], [1, 0, 0.0]])
    tcb = np.array([0.02, -0.01, 0.03]); tbc = -Rcb.T @ tcb
    fx, fy, cx, cy = 458.654, 457.296, 367.215, 248.375
    bf = 47.9                           # (only stereo observations use it; one camera per batch)
    sig_r, sig_v, sig_p = 1e-3, 5e-3, 2e-3
    dR = R1.T @ R2 @ _so3_exp(rs.normal(0, sig_r * 0.3, 3))
    dV = R1.T @ (v2 - v1 - g * dt) + rs.normal(0, sig_v * 0.3, 3)
    dP = R1.T @ (p2 - p1 - v1 * dt - 0.5 * g * dt * dt) + rs.normal(0, sig_p * 0.3, 3)
    link = dict(kf1=0, kf2=1, dR=dR.astype(np.float32), dV=dV.astype(np.float32), dP=dP.astype(np.float32),
                JRg=(-dt * np.eye(3)).astype(np.float32), JVg=rs.normal(0, 0.01, (3, 3)).astype(np.float32), JVa=(-dt * dR).astype(np.float32),
                JPg=rs.normal(0, 0.003, (3, 3)).astype(np.float32), JPa=(-0.5 * dt * dt * dR).astype(np.float32), dT=np.float32(dt),
                bias0=np.concatenate([bak, bgk]).astype(np.float32),
                info9=np.diag([1 / sig_r ** 2] * 3 + [1 / sig_v ** 2] * 3 + [1 / sig_p ** 2] * 3), info_gyro=np.eye(3) * 1e6 / dt,
                info_acc=np.eye(3) * 1e4 / dt, robust=np.uint8(0))
    Rcw = Rcb @ R2.T; tcw = Rcb @ (-R2.T @ p2) + tcb
    Xc = np.stack([rs.uniform(-3, 3, n), rs.uniform(-2, 2, n), rs.uniform(2, 12, n)], 1)
    Xw = (Rcw.T @ (Xc - tcw).T).T
    octave = rs.randint(0, 5, n); sig = 1.2 ** octave
    u = fx * Xc[:, 0] / Xc[:, 2] + cx + rs.normal(0, noise_px, n) * sig
    v = fy * Xc[:, 1] / Xc[:, 2] + cy + rs.normal(0, noise_px, n) * sig
    stereo = (rs.uniform(size=n) < stereo_frac).astype(np.uint8)
    ur = np.where(stereo > 0, u - bf / Xc[:, 2] + rs.normal(0, noise_px, n) * sig, -1.0)
    is_out = rs.uniform(size=n) < outlier_frac
    u = u + is_out * rs.choice([-1, 1], n) * rs.uniform(15, 40, n)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    R2i, p2i, v2i = R2, p2, v2
    if perturb:
        R2i = R2 @ _so3_exp(rs.normal(0, 0.01, 3)); p2i = p2 + rs.normal(0, 0.02, 3); v2i = v2 + rs.normal(0, 0.05, 3)
    pr = dict(Rwb=f32(np.stack([R1, R2i])), twb=f32(np.stack([p1, p2i])), vel=f32(np.stack([v1, v2i])), bg=f32(np.stack([bgk, bgk])), ba=f32(np.stack([bak, bak])),
              Rcb=f32(Rcb), tcb=f32(tcb), tbc=f32(tbc), fx=float(np.float32(fx)), fy=float(np.float32(fy)), cx=float(np.float32(cx)), cy=float(np.float32(cy)),
              bf=float(np.float32(bf)), Xw=f32(Xw), obs=np.stack([f32(u), f32(v), f32(ur)], 1), inv_sigma2=f32(1.0 / sig ** 2), stereo=stereo,
              close_point=(Xc[:, 2] < 10).astype(np.uint8), link=link, huber_mono=float(np.float32(np.sqrt(5.991))), huber_stereo=float(np.float32(np.sqrt(7.815))),
              rec_init=0)
    if last_frame:      # PoseInertialOptimizationLastFrame: [0] is the previous frame (free), tied to its prior pFp->mpcpi
        A = rs.normal(0, 1, (15, 15))
        Hp = np.diag([4e4] * 3 + [1e4] * 3 + [2e3] * 3 + [1e6] * 3 + [1e4] * 3) + 50.0 * (A @ A.T)
        pr.update(last_frame=1, prior_Rwb=f32(R1 @ _so3_exp(rs.normal(0, 0.002, 3))), prior_twb=f32(p1 + rs.normal(0, 0.004, 3)),
                  prior_vel=f32(v1 + rs.normal(0, 0.01, 3)), prior_bg=f32(bgk), prior_ba=f32(bak), prior_H=Hp)
        if perturb:
            pr["Rwb"][0] = f32(R1 @ _so3_exp(rs.normal(0, 0.004, 3))); pr["twb"][0] = f32(p1 + rs.normal(0, 0.01, 3)); pr["vel"][0] = f32(v1 + rs.normal(0, 0.02, 3))
    return pr, dict(Rwb=R2, twb=p2, vel=v2, is_outlier=is_out)
