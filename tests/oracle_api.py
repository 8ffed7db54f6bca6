"""ctypes bindings for the CPU ORACLE (oracle/liborb_oracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])
assert KP_DTYPE.itemsize == 28


def build_oracle(native=False):
    out = "liborb_oracle_native.so" if native else "liborb_oracle.so"
    args = ["make", "-C", ORACLE_DIR, "OUT=" + out]
    if native:
        args.append("ARCH=-march=native")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return os.path.join(ORACLE_DIR, out)


class FeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("offset", C.c_void_p), ("feat", C.c_void_p)]


class FrameGrid(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("cols", C.c_int32), ("rows", C.c_int32), ("u_right", C.c_void_p)]


class LbaProblem(C.Structure):
    _fields_ = [("n_poses", C.c_int32), ("pose_q", C.c_void_p), ("pose_t", C.c_void_p), ("pose_fixed", C.c_void_p),
                ("n_points", C.c_int32), ("points", C.c_void_p),
                ("n_edges", C.c_int32), ("edge_point", C.c_void_p), ("edge_pose", C.c_void_p), ("edge_obs", C.c_void_p),
                ("edge_inv_sigma2", C.c_void_p), ("edge_stereo", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double)]


class LbaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("stop_reason", C.c_int32),
                ("lambda_", C.c_double), ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
                ("chi2_trace", C.c_double * 16)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, path=None):
        if path is None:
            path = os.path.join(ORACLE_DIR, "liborb_oracle.so")
            if not os.path.exists(path):
                build_oracle()
        self.lib = L = C.CDLL(path)
        L.orb_oracle_create.restype = C.c_void_p
        L.orb_oracle_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orb_oracle_destroy.argtypes = [C.c_void_p]
        L.orb_oracle_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orb_oracle_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orb_oracle_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for f in (L.orb_oracle_level_image, L.orb_oracle_level_blurred):
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for f in (L.orb_oracle_level_candidates, L.orb_oracle_level_keypoints):
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orb_oracle_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orb_oracle_resize_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orb_oracle_gaussian7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orb_oracle_gauss_taps.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.orb_oracle_fast_atan2.restype = C.c_float
        L.orb_oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orb_oracle_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orb_oracle_cvround.argtypes = [C.c_double]
        L.orb_oracle_sort_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orbm_oracle_hamming.argtypes = [C.c_void_p, C.c_void_p]
        L.orbm_oracle_three_maxima.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 3
        L.orbm_oracle_search_by_bow.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(FeatVec),
                                                C.c_void_p, C.c_int, C.c_void_p, C.POINTER(FeatVec),
                                                C.c_float, C.c_int, C.c_void_p]
        L.orbm_oracle_search_by_bow_kfkf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(FeatVec),
                                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(FeatVec),
                                                     C.c_float, C.c_int, C.c_void_p]
        L.orbm_oracle_search_by_projection.argtypes = [C.POINTER(FrameGrid), C.c_void_p, C.c_void_p, C.c_int,
                                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orbm_oracle_search_by_projection_last.argtypes = [C.POINTER(FrameGrid), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                            C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.lba_oracle_solve.argtypes = [C.POINTER(LbaProblem), C.c_void_p, C.c_int, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(LbaStats)]

    # ---- extractor
    def extractor(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        return OracleExtractor(self, nfeatures, scale, nlevels, ini, mn)

    # ---- matcher
    def hamming(self, a, b):
        return self.lib.orbm_oracle_hamming(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))

    @staticmethod
    def _fv(fv):
        nodes, offs, feat = [np.ascontiguousarray(a) for a in fv]
        s = FeatVec(len(nodes), _p(nodes), _p(offs), _p(feat))
        s._keep = (nodes, offs, feat)
        return s

    def search_by_bow(self, dKF, validKF, angKF, fvKF, dF, angF, fvF, nnratio=0.7, check_ori=True):
        dKF, dF = np.ascontiguousarray(dKF), np.ascontiguousarray(dF)
        match = np.full(len(dF), -1, np.int32)
        a, b = self._fv(fvKF), self._fv(fvF)
        n = self.lib.orbm_oracle_search_by_bow(_p(dKF), len(dKF), _p(validKF), _p(angKF), C.byref(a),
                                               _p(dF), len(dF), _p(angF), C.byref(b),
                                               nnratio, int(check_ori), _p(match))
        return n, match

    def search_by_bow_kfkf(self, d1, valid1, ang1, fv1, d2, valid2, ang2, fv2, nnratio=0.7, check_ori=True):
        match = np.full(len(d1), -1, np.int32)
        a, b = self._fv(fv1), self._fv(fv2)
        n = self.lib.orbm_oracle_search_by_bow_kfkf(_p(d1), len(d1), _p(valid1), _p(ang1), C.byref(a),
                                                    _p(d2), len(d2), _p(valid2), _p(ang2), C.byref(b),
                                                    nnratio, int(check_ori), _p(match))
        return n, match

    @staticmethod
    def _grid(g):
        s = FrameGrid(len(g["x"]), _p(g["x"]), _p(g["y"]), _p(g["octave"]),
                      g["min_x"], g["min_y"], g["max_x"], g["max_y"], g.get("cols", 64), g.get("rows", 48), _p(g.get("u_right")))
        return s

    def search_by_projection(self, g, dF, scale_factors, mp, th, nnratio, assign, occupied, b_far=False, th_far=0.0):
        s = self._grid(g)
        return self.lib.orbm_oracle_search_by_projection(
            C.byref(s), _p(dF), _p(scale_factors), len(scale_factors), len(mp["u"]),
            _p(mp["in_view"]), _p(mp["u"]), _p(mp["v"]), _p(mp.get("ur")), _p(mp["level"]), _p(mp["view_cos"]), _p(mp["depth"]),
            _p(mp["desc"]), _p(mp["has_obs"]), _p(mp["bad"]), th, int(b_far), th_far, nnratio, _p(assign), _p(occupied))

    def search_by_projection_last(self, g, dF, angF, scale_factors, last, th, check_ori, assign, occupied):
        s = self._grid(g)
        return self.lib.orbm_oracle_search_by_projection_last(
            C.byref(s), _p(dF), _p(angF), _p(scale_factors), len(scale_factors), len(last["u"]),
            _p(last["valid"]), _p(last["u"]), _p(last["v"]), _p(last.get("ur")), _p(last["octave"]), _p(last["angle"]),
            _p(last["desc"]), _p(last["has_obs"]), th, int(last.get("level_window", 0)), int(check_ori), _p(assign), _p(occupied))

    def search_by_projection_kf(self, g, dF, angF, scale_factors, pts, th, orb_dist, check_ori, assign, occupied):
        s = self._grid(g)
        return self.lib.orbm_oracle_search_by_projection_kf(
            C.byref(s), _p(dF), _p(angF), _p(scale_factors), len(pts["u"]), _p(pts["valid"]), _p(pts["u"]), _p(pts["v"]),
            _p(pts["level"]), _p(pts["angle"]), _p(pts["desc"]), C.c_float(th), int(orb_dist), int(check_ori), _p(assign), _p(occupied))

    def search_by_projection_sim3(self, g, dKF, scale_factors, pts, th, ratio_hamming, assign, occupied):
        s = self._grid(g)
        return self.lib.orbm_oracle_search_by_projection_sim3(
            C.byref(s), _p(dKF), _p(scale_factors), len(pts["u"]), _p(pts["valid"]), _p(pts["u"]), _p(pts["v"]),
            _p(pts["level"]), _p(pts["desc"]), int(th), C.c_float(ratio_hamming), _p(assign), _p(occupied))

    def search_for_triangulation(self, k1, k2, ep, F12, sigma2_2, scale_2, only_stereo, coarse, check_ori):
        a, b = self._fv(k1["fv"]), self._fv(k2["fv"])
        m12 = np.full(max(len(k1["x"]), 1), -1, np.int32)
        n = self.lib.orbm_oracle_search_for_triangulation(
            _p(k1["desc"]), len(k1["x"]), _p(k1["has_mp"]), _p(k1["stereo"]), _p(k1["x"]), _p(k1["y"]), _p(k1["angle"]), C.byref(a),
            _p(k2["desc"]), len(k2["x"]), _p(k2["has_mp"]), _p(k2["stereo"]), _p(k2["x"]), _p(k2["y"]), _p(k2["octave"]), _p(k2["angle"]),
            C.byref(b), C.c_float(ep[0]), C.c_float(ep[1]), _p(F12), _p(sigma2_2), _p(scale_2), int(only_stereo), int(coarse), int(check_ori), _p(m12))
        return n, m12[:len(k1["x"])]

    def search_for_initialization(self, f1, g2, d2, ang2, window_size, nnratio, check_ori):
        s = self._grid(g2)
        n1 = len(f1["octave"])
        m12 = np.full(max(n1, 1), -1, np.int32)
        n = self.lib.orbm_oracle_search_for_initialization(
            _p(f1["desc"]), n1, _p(f1["octave"]), _p(f1["angle"]), _p(f1["prev_x"]), _p(f1["prev_y"]), C.byref(s), _p(d2), _p(ang2),
            int(window_size), C.c_float(nnratio), int(check_ori), _p(m12))
        return n, m12[:n1]

    def fuse_search(self, g, dKF, scale_factors, u_right, inv_sigma2, pts, th, chi2_check=True):
        s = self._grid(g)
        n = len(pts["u"])
        bi = np.full(max(n, 1), -1, np.int32); bd = np.full(max(n, 1), 256, np.int32)
        self.lib.orbm_oracle_fuse_search(C.byref(s), _p(dKF), _p(scale_factors), _p(u_right), _p(inv_sigma2), n, _p(pts["valid"]),
                                         _p(pts["u"]), _p(pts["v"]), _p(pts["ur"]), _p(pts["level"]), _p(pts["desc"]),
                                         C.c_float(th), int(chi2_check), _p(bi), _p(bd))
        return bi[:n], bd[:n]

    # ---- LBA
    def lba_solve(self, w, max_iters=10, lambda_init=0.0, stop_flag=None):
        keep = {k: np.ascontiguousarray(w[k]) for k in ("pose_q", "pose_t", "pose_fixed", "points", "edge_point",
                                                          "edge_pose", "edge_obs", "edge_inv_sigma2", "edge_stereo")}
        pr = LbaProblem(len(keep["pose_q"]), _p(keep["pose_q"]), _p(keep["pose_t"]), _p(keep["pose_fixed"]),
                        len(keep["points"]), _p(keep["points"]), len(keep["edge_point"]), _p(keep["edge_point"]),
                        _p(keep["edge_pose"]), _p(keep["edge_obs"]), _p(keep["edge_inv_sigma2"]), _p(keep["edge_stereo"]),
                        w["fx"], w["fy"], w["cx"], w["cy"], w["bf"], w["huber_mono"], w["huber_stereo"])
        q = np.zeros_like(keep["pose_q"]); t = np.zeros_like(keep["pose_t"]); pts = np.zeros_like(keep["points"])
        chi2 = np.zeros(len(keep["edge_point"])); dpos = np.zeros(len(keep["edge_point"]), np.uint8)
        st = LbaStats()
        sf = _p(stop_flag) if stop_flag is not None else None
        self.lib.lba_oracle_solve(C.byref(pr), sf, max_iters, lambda_init, _p(q), _p(t), _p(pts), _p(chi2), _p(dpos), C.byref(st))
        stats = dict(iterations=st.iterations, trials=st.trials, stop_reason=st.stop_reason, lambda_=st.lambda_,
                     chi2_initial=st.chi2_initial, chi2_final=st.chi2_final, chi2_trace=list(st.chi2_trace))
        return dict(pose_q=q, pose_t=t, points=pts, chi2=chi2, depth_positive=dpos, stats=stats)


class OracleShard:
    """CPU counterpart of capi.LbaShard + distributed.HipShard (same interface, numpy reduce buffer)."""

    def __init__(self, orc, w):
        L = self.L = orc.lib
        L.lba_oracle_shard_create.restype = C.c_void_p
        L.lba_oracle_shard_create.argtypes = [C.POINTER(LbaProblem)]
        L.lba_oracle_shard_destroy.argtypes = [C.c_void_p]
        L.lba_oracle_shard_reduce_len.restype = C.c_int64
        L.lba_oracle_shard_reduce_len.argtypes = [C.c_void_p]
        L.lba_oracle_shard_reduce_buffer.restype = C.POINTER(C.c_double)
        L.lba_oracle_shard_reduce_buffer.argtypes = [C.c_void_p]
        L.lba_oracle_shard_linearize.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 3
        L.lba_oracle_shard_reduce.argtypes = [C.c_void_p, C.c_double]
        L.lba_oracle_shard_finish.argtypes = [C.c_void_p, C.c_double] + [C.POINTER(C.c_double)] * 3
        L.lba_oracle_shard_accept.argtypes = [C.c_void_p, C.c_int]
        L.lba_oracle_shard_download.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        self.keep = {k: np.ascontiguousarray(w[k]) for k in ("pose_q", "pose_t", "pose_fixed", "points", "edge_point",
                                                               "edge_pose", "edge_obs", "edge_inv_sigma2", "edge_stereo")}
        k = self.keep
        pr = LbaProblem(len(k["pose_q"]), _p(k["pose_q"]), _p(k["pose_t"]), _p(k["pose_fixed"]),
                        len(k["points"]), _p(k["points"]), len(k["edge_point"]), _p(k["edge_point"]),
                        _p(k["edge_pose"]), _p(k["edge_obs"]), _p(k["edge_inv_sigma2"]), _p(k["edge_stereo"]),
                        w["fx"], w["fy"], w["cx"], w["cy"], w["bf"], w["huber_mono"], w["huber_stereo"])
        self.h = C.c_void_p(L.lba_oracle_shard_create(C.byref(pr)))
        self.n_red = L.lba_oracle_shard_reduce_len(self.h)
        buf = L.lba_oracle_shard_reduce_buffer(self.h)
        self.array = np.ctypeslib.as_array(buf, shape=(self.n_red,))        # aliases the C buffer
        self.n = int(round((-3 + np.sqrt(9 + 4 * self.n_red)) / 2))

    def __del__(self):
        try:
            self.L.lba_oracle_shard_destroy(self.h)
        except Exception:
            pass

    def linearize(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.L.lba_oracle_shard_linearize(self.h, C.byref(a), C.byref(b), C.byref(c))
        self._mdp = b.value
        return a.value, b.value, c.value

    def reduce(self, lam):
        self.L.lba_oracle_shard_reduce(self.h, lam)

    def finish(self, lam):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        r = self.L.lba_oracle_shard_finish(self.h, lam, C.byref(a), C.byref(b), C.byref(c))
        return r, a.value, b.value, c.value

    def accept(self, ok):
        self.L.lba_oracle_shard_accept(self.h, int(ok))

    def max_pose_diag(self):
        # summed diagonal section of the (all-reduced) buffer; before any reduce() it is zero and the local maximum applies
        sec = float(np.abs(self.array[self.n * self.n + 2 * self.n:]).max()) if self.n else 0.0
        return max(sec, getattr(self, "_mdp", 0.0))

    def download(self):
        k = self.keep
        q = np.zeros_like(k["pose_q"]); t = np.zeros_like(k["pose_t"]); pts = np.zeros_like(k["points"])
        chi2 = np.zeros(len(k["edge_point"])); dpos = np.zeros(len(k["edge_point"]), np.uint8)
        self.L.lba_oracle_shard_download(self.h, _p(q), _p(t), _p(pts), _p(chi2), _p(dpos))
        return dict(pose_q=q, pose_t=t, points=pts, chi2=chi2, depth_positive=dpos)


class OracleExtractor:
    def __init__(self, orc, nfeatures, scale, nlevels, ini, mn):
        self.o = orc
        self.L = orc.lib
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = C.c_void_p(self.L.orb_oracle_create(nfeatures, scale, nlevels, ini, mn))

    def __del__(self):
        try:
            self.L.orb_oracle_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        nf = np.zeros(n, np.int32); um = np.zeros(16, np.int32)
        self.L.orb_oracle_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(nf), _p(um))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, nfeat=nf, umax=um)

    def extract(self, img, lap=(0, 1000)):
        img = np.ascontiguousarray(img)
        h, w = img.shape
        cap = self.nfeatures + 4 * self.nlevels + 64
        for _ in range(2):      # (-2: more key points than `cap` -- tiny budgets on wide images return 4 nodes per root and level; n = the count needed)
            kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
            r = self.L.orb_oracle_extract(self.h, _p(img), w, h, w, lap[0], lap[1], _p(kps), _p(desc), cap, C.byref(n))
            if r != -2:
                break
            cap = n.value
        return r, kps[:n.value].copy(), desc[:n.value].copy()

    def level_size(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orb_oracle_level_size(self.h, l, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_image(self, l):
        w, h = self.level_size(l)
        out = np.zeros((h, w), np.uint8)
        self.L.orb_oracle_level_image(self.h, l, _p(out))
        return out

    def level_blurred(self, l):
        w, h = self.level_size(l)
        out = np.zeros((h, w), np.uint8)
        r = self.L.orb_oracle_level_blurred(self.h, l, _p(out))
        return out if r == 0 else None

    def level_candidates(self, l, cap=100000):
        out = np.zeros(cap, KP_DTYPE)
        n = self.L.orb_oracle_level_candidates(self.h, l, _p(out), cap)
        return out[:n].copy()

    def level_keypoints(self, l, cap=20000):
        out = np.zeros(cap, KP_DTYPE)
        n = self.L.orb_oracle_level_keypoints(self.h, l, _p(out), cap)
        return out[:n].copy()


class PoseProblem(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("stereo", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double)]


class PoseStats(C.Structure):
    _fields_ = [("iterations", C.c_int32 * 4), ("trials", C.c_int32 * 4), ("chi2", C.c_double * 4)]


def pose_problem_struct(w, cls=PoseProblem):
    keep = {k: np.ascontiguousarray(w[k]) for k in ("Xw", "obs", "inv_sigma2", "stereo")}
    pr = cls()
    for i in range(4):
        pr.q[i] = float(w["q"][i])
    for i in range(3):
        pr.t[i] = float(w["t"][i])
    pr.n = len(keep["Xw"])
    pr.Xw, pr.obs, pr.inv_sigma2, pr.stereo = (keep[k].ctypes.data for k in ("Xw", "obs", "inv_sigma2", "stereo"))
    for k in ("fx", "fy", "cx", "cy", "bf", "huber_mono", "huber_stereo"):
        setattr(pr, k, float(w[k]))
    pr._keep = keep
    return pr


def oracle_pose_optimize(orc, w):
    """Optimizer::PoseOptimization restatement: returns dict(q, t, outlier, n_bad, inliers)."""
    L = orc.lib
    L.pose_oracle_optimize.argtypes = [C.POINTER(PoseProblem), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
    pr = pose_problem_struct(w)
    q = np.zeros(4); t = np.zeros(3); outl = np.zeros(max(pr.n, 1), np.uint8); nb = C.c_int()
    st = PoseStats()
    r = L.pose_oracle_optimize(C.byref(pr), _p(q), _p(t), _p(outl), C.byref(nb), C.byref(st))
    return dict(q=q, t=t, outlier=outl[:pr.n], n_bad=nb.value, inliers=r,
                iterations=list(st.iterations), trials=list(st.trials), chi2=list(st.chi2))


class OracleVocabStruct(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("L", C.c_int32), ("child_off", C.c_void_p), ("child_id", C.c_void_p),
                ("desc", C.c_void_p), ("weight", C.c_void_p), ("word_id", C.c_void_p)]


def _vocab_struct(voc):
    keep = [np.ascontiguousarray(voc["child_off"], np.int32), np.ascontiguousarray(voc["child_id"], np.uint32),
            np.ascontiguousarray(voc["desc"], np.uint8), np.ascontiguousarray(voc["weight"], np.float64),
            np.ascontiguousarray(voc["word_id"], np.int32)]
    s = OracleVocabStruct(int(voc["n_nodes"]), int(voc["L"]), *[a.ctypes.data for a in keep])
    s._keep = keep
    return s


def oracle_transform_features(orc, voc, desc, levelsup=4):
    """TemplatedVocabulary::transform(feature, word_id, weight, &nid, levelsup) for every row of desc."""
    s = _vocab_struct(voc)
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    word = np.zeros(max(n, 1), np.uint32); weight = np.zeros(max(n, 1)); node = np.zeros(max(n, 1), np.uint32)
    orc.lib.dbow_oracle_transform_features(C.byref(s), _p(desc), n, int(levelsup), _p(word), _p(weight), _p(node))
    return word[:n], weight[:n], node[:n]


def oracle_transform(orc, voc, desc, levelsup=4):
    """transform(features, BowVector&, FeatureVector&, levelsup): (bow_id, bow_val), (fv_node, fv_off, fv_feat)"""
    s = _vocab_struct(voc)
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc); m = max(n, 1)
    bi = np.zeros(m, np.uint32); bv = np.zeros(m); fn = np.zeros(m, np.uint32); fo = np.zeros(m + 1, np.int32); ff = np.zeros(m, np.uint32)
    nb, nf = C.c_int32(), C.c_int32()
    used = orc.lib.dbow_oracle_transform(C.byref(s), _p(desc), n, int(levelsup), _p(bi), _p(bv), C.byref(nb), _p(fn), _p(fo), _p(ff), C.byref(nf))
    return (bi[:nb.value], bv[:nb.value]), (fn[:nf.value], fo[:nf.value + 1], ff[:used])


def oracle_stereo_matches(exL, exR, kpsL, descL, kpsR, descR, mb, mbf):
    """Frame::ComputeStereoMatches on the pyramids of exL / exR's last extract call: (n_before_cut, uRight, depth)"""
    kpsL = np.ascontiguousarray(kpsL); kpsR = np.ascontiguousarray(kpsR)
    descL = np.ascontiguousarray(descL); descR = np.ascontiguousarray(descR)
    n = len(kpsL)
    ur = np.full(max(n, 1), -1, np.float32); dp = np.full(max(n, 1), -1, np.float32)
    r = exL.L.orb_oracle_stereo_matches(exL.h, exR.h, _p(kpsL), _p(descL), n, _p(kpsR), _p(descR), len(kpsR),
                                        C.c_float(mb), C.c_float(mbf), _p(ur), _p(dp))
    return r, ur[:n], dp[:n]


IMU_DTYPE = np.dtype([("ts", "i8"), ("gyro", "f4", (3,)), ("acce", "f4", (3,))])
assert IMU_DTYPE.itemsize == 32


def oracle_pack_packet(orc, frame_id, timestamp, kps, desc, imu=None):
    """SlamPktVI(id, timestamp, kps, descriptors, imus): (payload bytes, head bytes)"""
    kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    imu = np.zeros(0, IMU_DTYPE) if imu is None else np.ascontiguousarray(imu, IMU_DTYPE)
    total = 16 + 36 * len(kps) + 32 * len(imu)
    out = np.zeros(total, np.uint8); head = np.zeros(2, np.uint8)
    r = orc.lib.edge_oracle_pack(C.c_int32(int(frame_id)), C.c_int64(int(timestamp)), _p(kps), _p(desc), len(kps), _p(imu), len(imu),
                                 _p(out), total, _p(head))
    assert r == total
    return out, head


def oracle_unpack_packet(orc, payload, cap_pts=4096, cap_imu=256):
    """SlamPktVI(buffer, packet_size): (status, frame_id, timestamp, kps, desc, imu)"""
    payload = np.ascontiguousarray(payload, np.uint8)
    kps = np.zeros(cap_pts, KP_DTYPE); desc = np.zeros((cap_pts, 32), np.uint8); imu = np.zeros(cap_imu, IMU_DTYPE)
    fid, ts, n, m = C.c_int32(), C.c_int64(), C.c_int(), C.c_int()
    r = orc.lib.edge_oracle_unpack(_p(payload), len(payload), C.byref(fid), C.byref(ts), _p(kps), _p(desc), cap_pts, C.byref(n),
                                   _p(imu), cap_imu, C.byref(m))
    return r, fid.value, ts.value, kps[:n.value], desc[:n.value], imu[:m.value]


def oracle_distinctive(orc, desc, off):
    """MapPoint::ComputeDistinctiveDescriptors for a batch: (BestIdx, BestMedian) per point"""
    desc = np.ascontiguousarray(desc, np.uint8); off = np.ascontiguousarray(off, np.int32)
    P = len(off) - 1
    bi = np.zeros(max(P, 1), np.int32); bm = np.zeros(max(P, 1), np.int32)
    orc.lib.map_oracle_distinctive(_p(desc), _p(off), P, _p(bi), _p(bm))
    return bi[:P], bm[:P]


def oracle_normal_and_depth(orc, pos, centers, off, ref_center, level_scale, last_level_scale):
    """MapPoint::UpdateNormalAndDepth for a batch: (normal [P][3], max_dist, min_dist)"""
    pos = np.ascontiguousarray(pos, np.float32); centers = np.ascontiguousarray(centers, np.float32); off = np.ascontiguousarray(off, np.int32)
    ref_center = np.ascontiguousarray(ref_center, np.float32); level_scale = np.ascontiguousarray(level_scale, np.float32)
    P = len(off) - 1
    nrm = np.zeros((max(P, 1), 3), np.float32); mx = np.zeros(max(P, 1), np.float32); mn = np.zeros(max(P, 1), np.float32)
    orc.lib.map_oracle_normal_and_depth(_p(pos), _p(centers), _p(off), _p(ref_center), _p(level_scale), C.c_float(last_level_scale), P,
                                        _p(nrm), _p(mx), _p(mn))
    return nrm[:P], mx[:P], mn[:P]


class InertialLink(C.Structure):
    _fields_ = [("kf1", C.c_int32), ("kf2", C.c_int32), ("dR", C.c_float * 9), ("dV", C.c_float * 3), ("dP", C.c_float * 3),
                ("JRg", C.c_float * 9), ("JVg", C.c_float * 9), ("JVa", C.c_float * 9), ("JPg", C.c_float * 9), ("JPa", C.c_float * 9),
                ("dT", C.c_float), ("bias0", C.c_float * 6), ("info9", C.c_double * 81), ("info_gyro", C.c_double * 9), ("info_acc", C.c_double * 9),
                ("robust", C.c_uint8)]


class InertialProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("Rwb", C.c_void_p), ("twb", C.c_void_p), ("vel", C.c_void_p), ("bg", C.c_void_p), ("ba", C.c_void_p),
                ("pose_fixed", C.c_void_p), ("has_imu", C.c_void_p), ("imu_fixed", C.c_void_p),
                ("Rcb", C.c_double * 9), ("tcb", C.c_double * 3), ("tbc", C.c_double * 3),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("n_points", C.c_int32), ("points", C.c_void_p), ("n_edges", C.c_int32), ("edge_kf", C.c_void_p), ("edge_point", C.c_void_p),
                ("edge_obs", C.c_void_p), ("edge_inv_sigma2", C.c_void_p), ("edge_stereo", C.c_void_p),
                ("n_links", C.c_int32), ("links", C.c_void_p),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("huber_inertial", C.c_double), ("lambda_init", C.c_double), ("max_iters", C.c_int32)]


def _inertial_struct(pr):
    keep = {k: np.ascontiguousarray(pr[k]) for k in ("Rwb", "twb", "vel", "bg", "ba", "pose_fixed", "has_imu", "imu_fixed", "points", "edge_kf",
                                                       "edge_point", "edge_obs", "edge_inv_sigma2", "edge_stereo")}
    links = (InertialLink * len(pr["links"]))()
    for L, d in zip(links, pr["links"]):
        L.kf1, L.kf2, L.dT, L.robust = int(d["kf1"]), int(d["kf2"]), float(d["dT"]), int(d["robust"])
        for name in ("dR", "dV", "dP", "JRg", "JVg", "JVa", "JPg", "JPa", "bias0"):
            getattr(L, name)[:] = np.asarray(d[name], np.float32).ravel().tolist()
        for name in ("info9", "info_gyro", "info_acc"):
            getattr(L, name)[:] = np.asarray(d[name], np.float64).ravel().tolist()
    s = InertialProblem()
    s.n_kf = int(pr["n_kf"])
    for k in ("Rwb", "twb", "vel", "bg", "ba", "pose_fixed", "has_imu", "imu_fixed", "points", "edge_kf", "edge_point", "edge_obs", "edge_inv_sigma2", "edge_stereo"):
        setattr(s, k, keep[k].ctypes.data)
    s.Rcb[:] = np.asarray(pr["Rcb"], np.float64).ravel().tolist(); s.tcb[:] = list(map(float, pr["tcb"])); s.tbc[:] = list(map(float, pr["tbc"]))
    s.fx, s.fy, s.cx, s.cy, s.bf = pr["fx"], pr["fy"], pr["cx"], pr["cy"], pr["bf"]
    s.n_points = len(keep["points"]); s.n_edges = len(keep["edge_kf"]); s.n_links = len(links); s.links = C.addressof(links)
    s.huber_mono, s.huber_stereo, s.huber_inertial = pr["huber_mono"], pr["huber_stereo"], pr["huber_inertial"]
    s.lambda_init, s.max_iters = float(pr["lambda_init"]), int(pr["max_iters"])
    s._keep = (keep, links)
    return s


def oracle_inertial_solve(orc, pr):
    """Optimizer::LocalInertialBA numerical core: dict(Rwb, twb, vel, bg, ba, points, chi2, depth_positive, stats)"""
    s = _inertial_struct(pr)
    n, m, ne = s.n_kf, s.n_points, s.n_edges
    Rwb = np.zeros((n, 3, 3)); twb = np.zeros((n, 3)); vel = np.zeros((n, 3)); bg = np.zeros((n, 3)); ba = np.zeros((n, 3))
    pts = np.zeros((max(m, 1), 3)); chi2 = np.zeros(max(ne, 1)); dpos = np.zeros(max(ne, 1), np.uint8)
    st = LbaStats()
    orc.lib.inertial_oracle_solve(C.byref(s), _p(Rwb), _p(twb), _p(vel), _p(bg), _p(ba), _p(pts), _p(chi2), _p(dpos), C.byref(st))
    stats = dict(iterations=st.iterations, trials=st.trials, stop_reason=st.stop_reason, lambda_=st.lambda_, chi2_initial=st.chi2_initial,
                 chi2_final=st.chi2_final)
    return dict(Rwb=Rwb, twb=twb, vel=vel, bg=bg, ba=ba, points=pts[:m], chi2=chi2[:ne], depth_positive=dpos[:ne], stats=stats)


def oracle_inertial_jacobian_check(orc, pr, link, h=1e-5):
    s = _inertial_struct(pr)
    orc.lib.inertial_oracle_jacobian_check.restype = C.c_double
    return orc.lib.inertial_oracle_jacobian_check(C.byref(s), int(link), C.c_double(h))


class PoseInertialProblem(C.Structure):
    _fields_ = [("Rwb", C.c_double * 18), ("twb", C.c_double * 6), ("vel", C.c_double * 6), ("bg", C.c_double * 6), ("ba", C.c_double * 6),
                ("Rcb", C.c_double * 9), ("tcb", C.c_double * 3), ("tbc", C.c_double * 3),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p), ("inv_sigma2", C.c_void_p), ("stereo", C.c_void_p), ("close_point", C.c_void_p),
                ("link", InertialLink), ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("rec_init", C.c_int32),
                ("last_frame", C.c_int32), ("prior_Rwb", C.c_double * 9), ("prior_twb", C.c_double * 3), ("prior_vel", C.c_double * 3),
                ("prior_bg", C.c_double * 3), ("prior_ba", C.c_double * 3), ("prior_H", C.c_double * 225)]


def fill_pose_inertial_struct(s, pr):
    """shared by the oracle binding and the product binding (same field layout)"""
    keep = {k: np.ascontiguousarray(pr[k], t) for k, t in (("Xw", np.float64), ("obs", np.float64), ("inv_sigma2", np.float64), ("stereo", np.uint8),
                                                           ("close_point", np.uint8))}
    for k, m in (("Rwb", 18), ("twb", 6), ("vel", 6), ("bg", 6), ("ba", 6), ("Rcb", 9), ("tcb", 3), ("tbc", 3)):
        getattr(s, k)[:] = np.asarray(pr[k], np.float64).ravel().tolist()
    s.fx, s.fy, s.cx, s.cy, s.bf = pr["fx"], pr["fy"], pr["cx"], pr["cy"], pr["bf"]
    s.n = len(keep["Xw"])
    for k in keep:
        setattr(s, k, keep[k].ctypes.data)
    d, L = pr["link"], s.link
    L.kf1, L.kf2, L.dT, L.robust = int(d["kf1"]), int(d["kf2"]), float(d["dT"]), int(d["robust"])
    for name in ("dR", "dV", "dP", "JRg", "JVg", "JVa", "JPg", "JPa", "bias0"):
        getattr(L, name)[:] = np.asarray(d[name], np.float32).ravel().tolist()
    for name in ("info9", "info_gyro", "info_acc"):
        getattr(L, name)[:] = np.asarray(d[name], np.float64).ravel().tolist()
    s.huber_mono, s.huber_stereo, s.rec_init = pr["huber_mono"], pr["huber_stereo"], int(pr["rec_init"])
    s.last_frame = int(pr.get("last_frame", 0))
    if s.last_frame:
        for k, m in (("prior_Rwb", 9), ("prior_twb", 3), ("prior_vel", 3), ("prior_bg", 3), ("prior_ba", 3), ("prior_H", 225)):
            getattr(s, k)[:] = np.asarray(pr[k], np.float64).ravel().tolist()
    s._keep = keep
    return s


def oracle_pose_inertial_optimize(orc, pr):
    """Optimizer::PoseInertialOptimizationLastKeyFrame: dict(Rwb, twb, vel, bg, ba, outlier, H, n_bad, inliers)"""
    s = fill_pose_inertial_struct(PoseInertialProblem(), pr)
    n = s.n
    Rwb = np.zeros((3, 3)); twb = np.zeros(3); vel = np.zeros(3); bg = np.zeros(3); ba = np.zeros(3)
    N = 30 if s.last_frame else 15
    out = np.zeros(max(n, 1), np.uint8); H = np.zeros((N, N)); nb = C.c_int()
    r = orc.lib.pose_inertial_oracle_optimize(C.byref(s), _p(Rwb), _p(twb), _p(vel), _p(bg), _p(ba), _p(out), _p(H), C.byref(nb))
    return dict(Rwb=Rwb, twb=twb, vel=vel, bg=bg, ba=ba, outlier=out[:n], H=H, n_bad=nb.value, inliers=r)


def oracle_undistort(orc, kps, K, dist, Knew):
    """Frame::UndistortKeyPoints: K / Knew = (fx, fy, cx, cy), dist = (k1, k2, p1, p2, k3)"""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    out = np.zeros_like(kps)
    K = np.ascontiguousarray(K, np.float32); dist = np.ascontiguousarray(dist, np.float32); Knew = np.ascontiguousarray(Knew, np.float32)
    orc.lib.edge_oracle_undistort(_p(kps), len(kps), _p(K), _p(dist), _p(Knew), _p(out))
    return out
