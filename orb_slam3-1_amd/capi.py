"""ctypes mirror of include/orbslam3_hip.h.  Names, argument meaning and error behaviour follow the C ABI,
which in turn follows the reference's ORBextractor / ORBmatcher / Optimizer signatures."""
import ctypes as C
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG_DIR, "liborbslam3_hip.so")

KP_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("size", "f4"), ("angle", "f4"), ("response", "f4"),
                     ("octave", "i4"), ("class_id", "i4")])
assert KP_DTYPE.itemsize == 28

ORBX_OK, ORBX_ERR_EMPTY, ORBX_ERR_CAPACITY, ORBX_ERR_ARG, ORBX_ERR_NO_DEVICE, ORBX_ERR_HIP, ORBX_ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbslam3_hip error %d: %s" % (code, msg))
        self.code = code


class FeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("offset", C.c_void_p), ("feat", C.c_void_p)]


class BowPair(C.Structure):
    _fields_ = [("desc_kf", C.c_void_p), ("n_kf", C.c_int32), ("valid_kf", C.c_void_p), ("angle_kf", C.c_void_p), ("fv_kf", FeatVec),
                ("desc_f", C.c_void_p), ("n_f", C.c_int32), ("angle_f", C.c_void_p), ("fv_f", FeatVec),
                ("match_f2kf", C.c_void_p), ("n_matches", C.c_int32)]


class Frame(C.Structure):
    _fields_ = [("n", C.c_int32), ("x", C.c_void_p), ("y", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p),
                ("desc", C.c_void_p), ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("scale_factors", C.c_void_p), ("n_levels", C.c_int32),
                ("u_right", C.c_void_p)]


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


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError("%s not found: build it with __graft_entry__.build() (hipcc, gfx950). "
                          "There is no CPU fallback." % _LIB_PATH)
    return C.CDLL(_LIB_PATH)


lib = _load()
lib.orbx_last_error.restype = C.c_char_p
lib.orbx_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
lib.orbx_destroy.argtypes = [C.c_void_p]
lib.orbx_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.orbx_extract_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
lib.orbx_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.orbx_max_keypoints.argtypes = [C.c_void_p]
lib.orbx_max_keypoints_for.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.orbx_levels.argtypes = [C.c_void_p]
lib.orbx_scale_factor.argtypes = [C.c_void_p]
lib.orbx_scale_factor.restype = C.c_float
lib.orbx_scale_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4
lib.orbx_features_per_level.argtypes = [C.c_void_p, C.c_void_p]
lib.orbx_pyramid_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.orbx_pyramid_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
lib.orbx_debug_blurred_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
lib.orbx_debug_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
lib.orbx_debug_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
lib.orbx_debug_set_tail_delay.argtypes = [C.c_void_p, C.c_int]
lib.orbx_debug_last_schedule.argtypes = [C.c_void_p]
lib.orbx_debug_introsort.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.orbx_debug_wave_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
lib.orbx_debug_fast_atan2.restype = C.c_float
lib.orbx_debug_fast_atan2.argtypes = [C.c_float, C.c_float]
lib.orbx_debug_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(code):
    if code < 0:
        raise OrbxError(code, (lib.orbx_last_error() or b"").decode())
    return code


def device_count():
    return lib.orbx_device_count()


class Extractor:
    """ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:44-109)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th_fast=20, min_th_fast=7, device=0):
        h = C.c_void_p()
        _check(lib.orbx_create(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast, device, C.byref(h)))
        self._h = h
        self.nlevels = nlevels
        self.max_keypoints = lib.orbx_max_keypoints(h)

    def close(self):
        if getattr(self, "_h", None):
            lib.orbx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # getters (include/ORBextractor.h:61-81)
    def GetLevels(self):
        return lib.orbx_levels(self._h)

    def GetScaleFactor(self):
        return lib.orbx_scale_factor(self._h)

    def _tables(self):
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)]
        _check(lib.orbx_scale_tables(self._h, *[_p(a) for a in t]))
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        _check(lib.orbx_features_per_level(self._h, _p(out)))
        return out

    def __call__(self, image, lapping_area=(0, 1000)):
        """operator(): returns (monoIndex, keypoints[KP_DTYPE], descriptors[n,32]); monoIndex == -1 for an empty image."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2       # assert(image.type() == CV_8UC1)
        img = image if image.strides[1] == 1 else np.ascontiguousarray(image)
        h, w = img.shape
        cap = self.max_keypoints
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n, mono = C.c_int(), C.c_int()
        _check(lib.orbx_extract(self._h, _p(img), w, h, img.strides[0], lapping_area[0], lapping_area[1],
                                _p(kps), _p(desc), cap, C.byref(n), C.byref(mono)))
        return mono.value, kps[:n.value].copy(), desc[:n.value].copy()

    def max_keypoints_for(self, width, height):
        """orbx_max_keypoints_for: the key-point bound for one image size (very wide images with tiny per-level budgets)"""
        return int(lib.orbx_max_keypoints_for(self._h, int(width), int(height)))

    def extract_batch(self, images, lapping_area=(0, 1000)):
        """images: uint8 array [B, H, W] (host).  Returns (mono[B], n[B], kps[B,cap], desc[B,cap,32])."""
        imgs = np.ascontiguousarray(images)
        B, h, w = imgs.shape
        cap = max(self.max_keypoints, self.max_keypoints_for(w, h))
        kps = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        mono = np.zeros(B, np.int32)
        ptrs = (C.c_void_p * B)(*[imgs[b].ctypes.data for b in range(B)])
        _check(lib.orbx_extract_batch(self._h, ptrs, B, w, h, imgs.strides[1], lapping_area[0], lapping_area[1],
                                      _p(kps), _p(desc), cap, _p(n), _p(mono)))
        return mono, n, kps, desc

    def extract_batch_device(self, d_imgs_ptr, batch, w, h, row_stride, frame_stride, d_kps_ptr, d_desc_ptr, cap,
                             d_n_ptr, d_mono_ptr, d_status_ptr, lapping_area=(0, 1000), stream=None):
        """Raw device-pointer variant (pointers as ints, e.g. torch.Tensor.data_ptr()); asynchronous."""
        _check(lib.orbx_extract_batch_device(self._h, d_imgs_ptr, batch, w, h, row_stride, frame_stride,
                                             lapping_area[0], lapping_area[1], d_kps_ptr, d_desc_ptr, cap,
                                             d_n_ptr, d_mono_ptr, d_status_ptr, stream))

    def stereo_matches(self, right, kps_l, desc_l, kps_r, desc_r, mb, mbf, frame=0):
        """Frame::ComputeStereoMatches with self = left extractor, `right` = right extractor (pyramids of their last calls)."""
        kps_l = np.ascontiguousarray(kps_l); kps_r = np.ascontiguousarray(kps_r)
        desc_l = np.ascontiguousarray(desc_l); desc_r = np.ascontiguousarray(desc_r)
        n = len(kps_l)
        ur = np.full(max(n, 1), -1, np.float32); dp = np.full(max(n, 1), -1, np.float32)
        _check(lib.orbx_stereo_matches(self._h, right._h, int(frame), _p(kps_l), _p(desc_l), n, _p(kps_r), _p(desc_r), len(kps_r),
                                       C.c_float(mb), C.c_float(mbf), _p(ur), _p(dp)))
        return ur[:n], dp[:n]

    def stereo_matches_device(self, right, batch, d_kps_l, d_desc_l, d_n_l, d_kps_r, d_desc_r, d_n_r, cap, mb, mbf, d_u_right, d_depth, stream=None):
        _check(lib.orbx_stereo_matches_device(self._h, right._h, int(batch), C.c_void_p(d_kps_l), C.c_void_p(d_desc_l), C.c_void_p(d_n_l),
                                              C.c_void_p(d_kps_r), C.c_void_p(d_desc_r), C.c_void_p(d_n_r), int(cap), C.c_float(mb), C.c_float(mbf),
                                              C.c_void_p(d_u_right), C.c_void_p(d_depth), C.c_void_p(stream or 0)))

    def profile_enable(self, on=True):
        lib.orbx_profile_enable.argtypes = [C.c_void_p, C.c_int]
        _check(lib.orbx_profile_enable(self._h, int(on)))

    STAGES = ("copy_level0", "resize", "fast_strips", "octree", "index", "blur", "orient_desc")

    def profile_read(self):
        """per-stage device milliseconds of the last call (HIP events on the launch stream)"""
        lib.orbx_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        ms = np.zeros(7, np.float32)
        _check(lib.orbx_profile_read(self._h, _p(ms), 7))
        return dict(zip(self.STAGES, ms.tolist()))

    # mvImagePyramid replacement + stage introspection
    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _check(lib.orbx_pyramid_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def pyramid_level(self, level, frame=0, border=0):
        w, h = self.level_size(level)
        out = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
        _check(lib.orbx_pyramid_level(self._h, frame, level, border, _p(out), out.strides[0]))
        return out

    def blurred_level(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(lib.orbx_debug_blurred_level(self._h, frame, level, _p(out), out.strides[0]))
        return out

    def debug_set_fast_corner_cap(self, cap):
        """test hook: shrink the FAST kernel's per-wave corner lists so that its overflow path runs"""
        lib.orbx_debug_set_fast_corner_cap.argtypes = [C.c_void_p, C.c_int]
        _check(lib.orbx_debug_set_fast_corner_cap(self._h, int(cap)))

    def debug_set_tail_delay(self, microseconds):
        _check(lib.orbx_debug_set_tail_delay(self._h, int(microseconds)))

    def debug_last_schedule(self):
        """bits: 0-1 octree instantiation (0 node pool in HBM, 1 keys + nodes in LDS, 2 keys in the scratch), 4 two octree launches,
        8 level-0 octree early, 16 level 0 read in place, 32 resize tail on the side stream"""
        return int(lib.orbx_debug_last_schedule(self._h))

    def candidates(self, level, frame=0, cap=200000):
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_int()
        _check(lib.orbx_debug_candidates(self._h, frame, level, _p(out), cap, C.byref(n)))
        return out[:min(n.value, cap)].copy()

    def level_keypoints(self, level, frame=0, cap=20000):
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_int()
        _check(lib.orbx_debug_level_keypoints(self._h, frame, level, _p(out), cap, C.byref(n)))
        return out[:min(n.value, cap)].copy()


def hamming(a, b):
    """ORBmatcher::DescriptorDistance."""
    lib.orbm_hamming.argtypes = [C.c_void_p, C.c_void_p]
    return lib.orbm_hamming(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))


def _fv(fv):
    nodes, offs, feat = [np.ascontiguousarray(a) for a in fv]
    s = FeatVec(len(nodes), nodes.ctypes.data, offs.ctypes.data, feat.ctypes.data)
    s._keep = (nodes, offs, feat)
    return s


class Matcher:
    """ORB_SLAM3::ORBmatcher (reference include/ORBmatcher.h:40-106): ORBmatcher(nnratio=0.6, checkOri=true)."""
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, check_orientation=True, device=0):
        self.nnratio = float(nnratio)
        self.check_ori = bool(check_orientation)
        lib.orbm_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.orbm_destroy.argtypes = [C.c_void_p]
        lib.orbm_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(FeatVec),
                                           C.c_void_p, C.c_int, C.c_void_p, C.POINTER(FeatVec), C.c_float, C.c_int, C.c_void_p]
        lib.orbm_search_by_bow_batch.argtypes = [C.c_void_p, C.POINTER(BowPair), C.c_int, C.c_float, C.c_int]
        lib.orbm_search_by_bow_kfkf.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(FeatVec),
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(FeatVec),
                                                C.c_float, C.c_int, C.c_void_p]
        lib.orbm_search_by_projection.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_int] + [C.c_void_p] * 10 + \
            [C.c_float, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        lib.orbm_search_by_projection_last.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_int] + [C.c_void_p] * 8 + \
            [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        h = C.c_void_p()
        _check(lib.orbm_create(device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.orbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def DescriptorDistance(a, b):
        return hamming(a, b)

    def SearchByBoW(self, dKF, validKF, angKF, fvKF, dF, angF, fvF):
        """SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&): returns (nmatches, match_f2kf[nF])."""
        dKF, dF = np.ascontiguousarray(dKF), np.ascontiguousarray(dF)
        match = np.full(len(dF), -1, np.int32)
        a, b = _fv(fvKF), _fv(fvF)
        n = _check(lib.orbm_search_by_bow(self._h, _p(dKF), len(dKF), _p(validKF), _p(angKF), C.byref(a),
                                          _p(dF), len(dF), _p(angF), C.byref(b), self.nnratio, int(self.check_ori), _p(match)))
        return n, match

    def SearchByBoW_batch(self, sets):
        """sets: list of dicts as synth.make_match_set.  Returns list of (nmatches, match)."""
        P = len(sets)
        arr = (BowPair * P)()
        keep = []
        outs = []
        for i, s in enumerate(sets):
            dKF, dF = np.ascontiguousarray(s["dKF"]), np.ascontiguousarray(s["dF"])
            a, b = _fv(s["fvKF"]), _fv(s["fvF"])
            m = np.full(len(dF), -1, np.int32)
            keep.append((dKF, dF, a, b, m))
            outs.append(m)
            arr[i] = BowPair(dKF.ctypes.data, len(dKF), s["validKF"].ctypes.data, s["angKF"].ctypes.data, a,
                             dF.ctypes.data, len(dF), s["angF"].ctypes.data, b, m.ctypes.data, 0)
        _check(lib.orbm_search_by_bow_batch(self._h, arr, P, self.nnratio, int(self.check_ori)))
        return [(arr[i].n_matches, outs[i]) for i in range(P)]

    def bow_plan(self, sets):
        """Uploads the pairs once; returns a BowPlan whose run() only launches the kernel."""
        return BowPlan(self, sets)

    def SearchByBoW_KFKF(self, d1, valid1, ang1, fv1, d2, valid2, ang2, fv2):
        match = np.full(len(d1), -1, np.int32)
        a, b = _fv(fv1), _fv(fv2)
        n = _check(lib.orbm_search_by_bow_kfkf(self._h, _p(d1), len(d1), _p(valid1), _p(ang1), C.byref(a),
                                               _p(d2), len(d2), _p(valid2), _p(ang2), C.byref(b),
                                               self.nnratio, int(self.check_ori), _p(match)))
        return n, match

    @staticmethod
    def _frame(g, desc, scale_factors, angle=None):
        f = Frame(len(g["x"]), g["x"].ctypes.data, g["y"].ctypes.data, g["octave"].ctypes.data,
                  angle.ctypes.data if angle is not None else None, desc.ctypes.data,
                  g["min_x"], g["min_y"], g["max_x"], g["max_y"], g.get("cols", 64), g.get("rows", 48),
                  scale_factors.ctypes.data, len(scale_factors),
                  g["u_right"].ctypes.data if g.get("u_right") is not None else None)      # mvuRight: rectified stereo / RGB-D frames
        return f

    def SearchByProjection(self, g, dF, scale_factors, mp, th, assign, occupied, far_points=False, th_far=0.0):
        """SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)."""
        f = self._frame(g, dF, scale_factors)
        return _check(lib.orbm_search_by_projection(
            self._h, C.byref(f), len(mp["u"]), _p(mp["in_view"]), _p(mp["u"]), _p(mp["v"]), _p(mp.get("ur")), _p(mp["level"]),
            _p(mp["view_cos"]), _p(mp["depth"]), _p(mp["desc"]), _p(mp["has_obs"]), _p(mp["bad"]),
            th, int(far_points), th_far, self.nnratio, _p(assign), _p(occupied)))

    def SearchByProjection_last(self, g, dF, angF, scale_factors, last, th, assign, occupied):
        """SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono): last["level_window"] carries
        bForward (1) / bBackward (2), last["ur"] the predicted right-image columns of a rectified-stereo frame."""
        f = self._frame(g, dF, scale_factors, angF)
        return _check(lib.orbm_search_by_projection_last(
            self._h, C.byref(f), len(last["u"]), _p(last["valid"]), _p(last["u"]), _p(last["v"]), _p(last.get("ur")), _p(last["octave"]),
            _p(last["angle"]), _p(last["desc"]), _p(last["has_obs"]), th, int(last.get("level_window", 0)), int(self.check_ori),
            _p(assign), _p(occupied)))

    def SearchByProjection_kf(self, g, dF, angF, scale_factors, pts, th, orb_dist, assign, occupied):
        """SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (relocalisation)."""
        f = self._frame(g, dF, scale_factors, angF)
        return _check(lib.orbm_search_by_projection_kf(
            self._h, C.byref(f), len(pts["u"]), _p(pts["valid"]), _p(pts["u"]), _p(pts["v"]), _p(pts["level"]),
            _p(pts["angle"]), _p(pts["desc"]), C.c_float(th), int(orb_dist), int(self.check_ori), _p(assign), _p(occupied)))

    def SearchByProjection_sim3(self, g, dKF, scale_factors, pts, th, ratio_hamming, assign, occupied):
        """SearchByProjection(KeyFrame* pKF, Sim3f& Scw, vpPoints, vpMatched, th, ratioHamming) (loop closing)."""
        f = self._frame(g, dKF, scale_factors)
        return _check(lib.orbm_search_by_projection_sim3(
            self._h, C.byref(f), len(pts["u"]), _p(pts["valid"]), _p(pts["u"]), _p(pts["v"]), _p(pts["level"]),
            _p(pts["desc"]), int(th), C.c_float(ratio_hamming), _p(assign), _p(occupied)))

    def FuseSearch(self, g, dKF, scale_factors, u_right, inv_sigma2, pts, th, chi2_check=True):
        """search core of both ORBmatcher::Fuse overloads: (best_idx, best_dist) per candidate map point"""
        f = self._frame(g, dKF, scale_factors)
        n = len(pts["u"])
        bi = np.full(max(n, 1), -1, np.int32); bd = np.full(max(n, 1), 256, np.int32)
        _check(lib.orbm_fuse_search(self._h, C.byref(f), _p(u_right), _p(inv_sigma2), n, _p(pts["valid"]), _p(pts["u"]), _p(pts["v"]),
                                    _p(pts["ur"]), _p(pts["level"]), _p(pts["desc"]), C.c_float(th), int(chi2_check), _p(bi), _p(bd)))
        return bi[:n], bd[:n]

    def DistinctiveDescriptors(self, desc, off):
        """MapPoint::ComputeDistinctiveDescriptors for a batch of map points: (BestIdx, BestMedian) per point"""
        desc = np.ascontiguousarray(desc, np.uint8); off = np.ascontiguousarray(off, np.int32)
        P = len(off) - 1
        bi = np.zeros(max(P, 1), np.int32); bm = np.zeros(max(P, 1), np.int32)
        _check(lib.orbm_distinctive_descriptors(self._h, _p(desc), _p(off), P, _p(bi), _p(bm)))
        return bi[:P], bm[:P]

    def UpdateNormalAndDepth(self, pos, centers, off, ref_center, level_scale, last_level_scale):
        """MapPoint::UpdateNormalAndDepth for a batch of map points: (normal [P][3], max_dist, min_dist)"""
        pos = np.ascontiguousarray(pos, np.float32); centers = np.ascontiguousarray(centers, np.float32); off = np.ascontiguousarray(off, np.int32)
        ref_center = np.ascontiguousarray(ref_center, np.float32); level_scale = np.ascontiguousarray(level_scale, np.float32)
        P = len(off) - 1
        nrm = np.zeros((max(P, 1), 3), np.float32); mx = np.zeros(max(P, 1), np.float32); mn = np.zeros(max(P, 1), np.float32)
        _check(lib.orbm_update_normal_and_depth(self._h, _p(pos), _p(centers), _p(off), _p(ref_center), _p(level_scale),
                                                C.c_float(last_level_scale), P, _p(nrm), _p(mx), _p(mn)))
        return nrm[:P], mx[:P], mn[:P]

    class _ProjQuery(C.Structure):
        _fields_ = [("frame", C.c_void_p), ("n_pts", C.c_int32), ("valid", C.c_void_p), ("proj_u", C.c_void_p), ("proj_v", C.c_void_p),
                    ("proj_ur", C.c_void_p), ("level", C.c_void_p), ("level_window", C.c_int32), ("view_cos", C.c_void_p), ("track_depth", C.c_void_p), ("mp_bad", C.c_void_p), ("angle", C.c_void_p),
                    ("desc_mp", C.c_void_p), ("mp_has_obs", C.c_void_p), ("assign", C.c_void_p), ("occupied", C.c_void_p), ("n_matches", C.c_int32)]

    def prepare_last_batch(self, cases):
        """cases: list of (g, dF, angF, scale_factors, last, assign, occupied) as for SearchByProjection_last"""
        n = len(cases)
        qs = (self._ProjQuery * n)()
        frames = []
        for i, (g, dF, angF, scale, last, assign, occ) in enumerate(cases):
            f = self._frame(g, dF, scale, angF)
            frames.append(f)
            q = qs[i]
            q.frame = C.addressof(f); q.n_pts = len(last["u"])
            q.valid, q.proj_u, q.proj_v, q.level = (last[k].ctypes.data for k in ("valid", "u", "v", "octave"))
            q.proj_ur = last["ur"].ctypes.data if last.get("ur") is not None else None
            q.level_window = int(last.get("level_window", 0))
            q.angle = last["angle"].ctypes.data; q.desc_mp = last["desc"].ctypes.data; q.mp_has_obs = last["has_obs"].ctypes.data
            q.assign = assign.ctypes.data; q.occupied = occ.ctypes.data
        return dict(qs=qs, frames=frames, cases=cases, n=n)

    def run_last_batch(self, prep, th):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono=true) for every case in one launch; returns n_matches per case"""
        _check(lib.orbm_search_by_projection_last_batch(self._h, prep["qs"], prep["n"], C.c_float(th), int(self.check_ori)))
        return [q.n_matches for q in prep["qs"]]

    class _DevFrames(C.Structure):
        _fields_ = [("d_kps", C.c_void_p), ("d_desc", C.c_void_p), ("d_n", C.c_void_p), ("cap", C.c_int32),
                    ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                    ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("scale_factors", C.c_void_p), ("n_levels", C.c_int32)]

    class _DevLastPoints(C.Structure):
        _fields_ = [("d_valid", C.c_void_p), ("d_u", C.c_void_p), ("d_v", C.c_void_p), ("d_octave", C.c_void_p), ("d_angle", C.c_void_p),
                    ("d_desc", C.c_void_p), ("d_n", C.c_void_p), ("cap", C.c_int32), ("d_has_obs", C.c_void_p)]

    def SearchByProjection_last_batch_device(self, cur, last, batch, th, d_assign, d_occupied, d_n_matches, stream=None, bounds=(0.0, 0.0, 640.0, 480.0),
                                             scale_factors=None, grid=(64, 48)):
        """cur = (d_kps, d_desc, d_n, cap) as orbx_extract_batch_device left them; last = (d_valid, d_u, d_v, d_octave, d_angle, d_desc, d_n, cap);
        every d_* a device pointer.  Only enqueues."""
        sf = np.ascontiguousarray(scale_factors, np.float32)
        self._dev_keep = sf
        cf = self._DevFrames(cur[0], cur[1], cur[2], int(cur[3]), bounds[0], bounds[1], bounds[2], bounds[3], grid[0], grid[1], sf.ctypes.data, len(sf))
        lp = self._DevLastPoints(last[0], last[1], last[2], last[3], last[4], last[5], last[6], int(last[7]), last[8] if len(last) > 8 else None)
        lib.orbm_search_by_projection_last_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib.orbm_search_by_projection_last_batch_device(self._h, C.byref(cf), C.byref(lp), int(batch), C.c_float(th), int(self.check_ori),
                                                               C.c_void_p(d_assign), C.c_void_p(d_occupied), C.c_void_p(d_n_matches), C.c_void_p(stream or 0)))

    class _DevMpExtras(C.Structure):
        _fields_ = [("d_view_cos", C.c_void_p), ("d_track_depth", C.c_void_p), ("d_bad", C.c_void_p), ("th_far", C.c_float), ("nnratio", C.c_float), ("far_points", C.c_int32)]

    def SearchByProjection_batch_device(self, cur, pts, extras, batch, th, d_assign, d_occupied, d_n_matches, stream=None, bounds=(0.0, 0.0, 640.0, 480.0),
                                        scale_factors=None, grid=(64, 48), far_points=False, th_far=0.0):
        """Frame x map points on device arrays: cur as above; pts = (d_valid, d_u, d_v, d_level, 0, d_desc, d_n, cap, d_has_obs);
        extras = (d_view_cos, d_track_depth, d_bad)"""
        sf = np.ascontiguousarray(scale_factors, np.float32)
        self._dev_keep = sf
        cf = self._DevFrames(cur[0], cur[1], cur[2], int(cur[3]), bounds[0], bounds[1], bounds[2], bounds[3], grid[0], grid[1], sf.ctypes.data, len(sf))
        lp = self._DevLastPoints(pts[0], pts[1], pts[2], pts[3], pts[4] or None, pts[5], pts[6], int(pts[7]), pts[8])
        ex = self._DevMpExtras(extras[0], extras[1], extras[2], float(th_far), self.nnratio, int(far_points))
        lib.orbm_search_by_projection_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _check(lib.orbm_search_by_projection_batch_device(self._h, C.byref(cf), C.byref(lp), C.byref(ex), int(batch), C.c_float(th),
                                                          C.c_void_p(d_assign), C.c_void_p(d_occupied), C.c_void_p(d_n_matches), C.c_void_p(stream or 0)))

    class _TriSide(C.Structure):
        _fields_ = [("n", C.c_int32), ("desc", C.c_void_p), ("has_mp", C.c_void_p), ("stereo", C.c_void_p), ("x", C.c_void_p),
                    ("y", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p), ("fv", FeatVec)]

    def SearchForTriangulation(self, k1, k2, ep, F12, sigma2_2, scale_2, only_stereo=False, coarse=False):
        """SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse): returns (nmatches, match12)"""
        sides = []
        for k in (k1, k2):
            fv = _fv(k["fv"])
            sides.append((self._TriSide(len(k["x"]), k["desc"].ctypes.data, k["has_mp"].ctypes.data, k["stereo"].ctypes.data,
                                        k["x"].ctypes.data, k["y"].ctypes.data, k["octave"].ctypes.data, k["angle"].ctypes.data, fv), fv))
        m12 = np.full(max(len(k1["x"]), 1), -1, np.int32)
        n = _check(lib.orbm_search_for_triangulation(self._h, C.byref(sides[0][0]), C.byref(sides[1][0]), C.c_float(ep[0]), C.c_float(ep[1]),
                                                     _p(F12), _p(sigma2_2), _p(scale_2), len(scale_2), int(only_stereo), int(coarse),
                                                     int(self.check_ori), _p(m12)))
        return n, m12[:len(k1["x"])]

    def SearchForInitialization(self, f1, g2, d2, ang2, scale_factors, window_size=100):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): returns (nmatches, vnMatches12)"""
        f = self._frame(g2, d2, scale_factors, ang2)
        n1 = len(f1["octave"])
        m12 = np.full(max(n1, 1), -1, np.int32)
        n = _check(lib.orbm_search_for_initialization(self._h, _p(f1["desc"]), n1, _p(f1["octave"]), _p(f1["angle"]), _p(f1["prev_x"]),
                                                      _p(f1["prev_y"]), C.byref(f), int(window_size), C.c_float(self.nnratio),
                                                      int(self.check_ori), _p(m12)))
        return n, m12[:n1]


class _BowSideDevice(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("kps", C.c_void_p), ("n", C.c_void_p), ("cap", C.c_int32), ("valid", C.c_void_p),
                ("fv_node", C.c_void_p), ("fv_off", C.c_void_p), ("fv_feat", C.c_void_p), ("n_fv_nodes", C.c_void_p)]


class _BowPairDevice(C.Structure):
    _fields_ = [("kf", _BowSideDevice), ("f", _BowSideDevice), ("match_f2kf", C.c_void_p), ("n_matches", C.c_void_p)]


class DeviceBowPlan:
    """SearchByBoW on device-resident pairs: sides = dicts of device addresses (desc, kps, n, cap, fv_node, fv_off, fv_feat,
    n_fv_nodes[, valid]) as produced by Extractor.extract_batch_device + Vocabulary.transform_batch_device."""

    def __init__(self, matcher, pairs):
        n = len(pairs)
        arr = (_BowPairDevice * n)()
        for i, (kf, f, match_ptr, nm_ptr) in enumerate(pairs):
            for side, d in ((arr[i].kf, kf), (arr[i].f, f)):
                side.desc, side.kps, side.n, side.cap = d["desc"], d["kps"], d["n"], int(d["cap"])
                side.valid = d.get("valid") or None
                side.fv_node, side.fv_off, side.fv_feat, side.n_fv_nodes = d["fv_node"], d["fv_off"], d["fv_feat"], d["n_fv_nodes"]
            arr[i].match_f2kf = match_ptr; arr[i].n_matches = nm_ptr
        h = C.c_void_p()
        _check(lib.orbm_bow_plan_create_device(matcher._h, arr, n, C.byref(h)))
        self._h, self._m = h, matcher

    def run(self, stream=None):
        _check(lib.orbm_bow_plan_run(self._h, C.c_float(self._m.nnratio), int(self._m.check_ori), C.c_void_p(stream or 0)))

    def close(self):
        if getattr(self, "_h", None):
            lib.orbm_bow_plan_destroy.argtypes = [C.c_void_p]
            lib.orbm_bow_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BowPlan:
    def __init__(self, matcher, sets):
        lib.orbm_bow_plan_create.argtypes = [C.c_void_p, C.POINTER(BowPair), C.c_int, C.POINTER(C.c_void_p)]
        lib.orbm_bow_plan_run.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        lib.orbm_bow_plan_fetch.argtypes = [C.c_void_p, C.POINTER(BowPair), C.c_void_p]
        lib.orbm_bow_plan_destroy.argtypes = [C.c_void_p]
        self.m = matcher
        P = self.P = len(sets)
        self.arr = (BowPair * P)()
        self.keep, self.outs = [], []
        for i, s in enumerate(sets):
            dKF, dF = np.ascontiguousarray(s["dKF"]), np.ascontiguousarray(s["dF"])
            a, b = _fv(s["fvKF"]), _fv(s["fvF"])
            m = np.full(len(dF), -1, np.int32)
            self.keep.append((dKF, dF, a, b, s["validKF"], s["angKF"], s["angF"]))
            self.outs.append(m)
            self.arr[i] = BowPair(dKF.ctypes.data, len(dKF), s["validKF"].ctypes.data, s["angKF"].ctypes.data, a,
                                  dF.ctypes.data, len(dF), s["angF"].ctypes.data, b, m.ctypes.data, 0)
        h = C.c_void_p()
        _check(lib.orbm_bow_plan_create(matcher._h, self.arr, P, C.byref(h)))
        self._h = h

    def run(self, stream=None):
        _check(lib.orbm_bow_plan_run(self._h, self.m.nnratio, int(self.m.check_ori), stream))

    def fetch(self, stream=None):
        _check(lib.orbm_bow_plan_fetch(self._h, self.arr, stream))
        return [(self.arr[i].n_matches, self.outs[i]) for i in range(self.P)]

    def close(self):
        if getattr(self, "_h", None):
            lib.orbm_bow_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _lba_problem(w):
    keep = {k: np.ascontiguousarray(w[k]) for k in ("pose_q", "pose_t", "pose_fixed", "points", "edge_point",
                                                      "edge_pose", "edge_obs", "edge_inv_sigma2", "edge_stereo")}
    pr = LbaProblem(len(keep["pose_q"]), keep["pose_q"].ctypes.data, keep["pose_t"].ctypes.data, keep["pose_fixed"].ctypes.data,
                    len(keep["points"]), keep["points"].ctypes.data, len(keep["edge_point"]), keep["edge_point"].ctypes.data,
                    keep["edge_pose"].ctypes.data, keep["edge_obs"].ctypes.data, keep["edge_inv_sigma2"].ctypes.data,
                    keep["edge_stereo"].ctypes.data, w["fx"], w["fy"], w["cx"], w["cy"], w["bf"], w["huber_mono"], w["huber_stereo"])
    pr._keep = keep
    return pr


def _stats_dict(st):
    return dict(iterations=st.iterations, trials=st.trials, stop_reason=st.stop_reason, lambda_=st.lambda_,
                chi2_initial=st.chi2_initial, chi2_final=st.chi2_final, chi2_trace=list(st.chi2_trace))


class LbaSolver:
    """Numerical core of Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:1116-1498)."""

    def __init__(self, device=0):
        lib.lba_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.lba_destroy.argtypes = [C.c_void_p]
        lib.lba_solve.argtypes = [C.c_void_p, C.POINTER(LbaProblem), C.c_void_p, C.c_int, C.c_double,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(LbaStats)]
        h = C.c_void_p()
        _check(lib.lba_create(device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.lba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, w, max_iters=10, lambda_init=0.0, stop_flag=None):
        pr = _lba_problem(w)
        k = pr._keep
        q = np.zeros_like(k["pose_q"]); t = np.zeros_like(k["pose_t"]); pts = np.zeros_like(k["points"])
        chi2 = np.zeros(len(k["edge_point"])); dpos = np.zeros(len(k["edge_point"]), np.uint8)
        st = LbaStats()
        _check(lib.lba_solve(self._h, C.byref(pr), _p(stop_flag), max_iters, lambda_init,
                             _p(q), _p(t), _p(pts), _p(chi2), _p(dpos), C.byref(st)))
        return dict(pose_q=q, pose_t=t, points=pts, chi2=chi2, depth_positive=dpos, stats=_stats_dict(st))


class LbaOutputs(C.Structure):
    _fields_ = [("pose_q", C.c_void_p), ("pose_t", C.c_void_p), ("points", C.c_void_p), ("chi2_per_edge", C.c_void_p), ("depth_positive", C.c_void_p)]


class LbaBatch:
    """lba_solve_batch: many LocalBundleAdjustment windows (one per map / client session) per launch."""

    def __init__(self, device=0):
        lib.lba_batch_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.lba_batch_destroy.argtypes = [C.c_void_p]
        lib.lba_solve_batch.argtypes = [C.c_void_p, C.POINTER(LbaProblem), C.POINTER(LbaOutputs), C.c_int, C.c_void_p, C.c_int, C.c_double, C.POINTER(LbaStats)]
        lib.lba_batch_last_device_ms.argtypes = [C.c_void_p]
        lib.lba_batch_last_device_ms.restype = C.c_double
        h = C.c_void_p()
        _check(lib.lba_batch_create(device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.lba_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, windows, want_outputs=True):
        """marshal once (a benchmark re-solves the same windows): problem structs, output arrays, stats"""
        n = len(windows)
        prs = [_lba_problem(w) for w in windows]
        arr = (LbaProblem * n)(*prs)
        outs, oarr = [], (LbaOutputs * n)()
        for i, pr in enumerate(prs):
            k = pr._keep
            o = dict(pose_q=np.zeros_like(k["pose_q"]), pose_t=np.zeros_like(k["pose_t"]), points=np.zeros_like(k["points"]),
                     chi2=np.zeros(len(k["edge_point"])), depth_positive=np.zeros(len(k["edge_point"]), np.uint8))
            outs.append(o)
            if want_outputs:
                oarr[i] = LbaOutputs(_p(o["pose_q"]), _p(o["pose_t"]), _p(o["points"]), _p(o["chi2"]), _p(o["depth_positive"]))
        return dict(n=n, prs=prs, arr=arr, outs=outs, oarr=oarr, stats=(LbaStats * n)(), want=want_outputs)

    def run(self, prep, max_iters=10, lambda_init=0.0, stop_flags=None):
        flags = None
        if stop_flags is not None:
            flags = (C.c_void_p * prep["n"])(*[(f.ctypes.data if f is not None else None) for f in stop_flags])
        _check(lib.lba_solve_batch(self._h, prep["arr"], prep["oarr"] if prep["want"] else None, prep["n"], flags, max_iters, lambda_init, prep["stats"]))
        return [dict(o, stats=_stats_dict(prep["stats"][i])) for i, o in enumerate(prep["outs"])]

    def solve(self, windows, max_iters=10, lambda_init=0.0, stop_flags=None):
        return self.run(self.prepare(windows), max_iters, lambda_init, stop_flags)

    def last_device_ms(self):
        return float(lib.lba_batch_last_device_ms(self._h))


class LbaShard:
    """One rank's share of a landmark-sharded global BA; see INTEGRATION.md."""

    def __init__(self, w, device=0):
        lib.lba_shard_create.argtypes = [C.c_int, C.POINTER(LbaProblem), C.POINTER(C.c_void_p)]
        lib.lba_shard_destroy.argtypes = [C.c_void_p]
        lib.lba_shard_reduce_len.argtypes = [C.c_void_p]
        lib.lba_shard_reduce_len.restype = C.c_int64
        lib.lba_shard_reduce_buffer.argtypes = [C.c_void_p]
        lib.lba_shard_reduce_buffer.restype = C.c_void_p
        lib.lba_shard_linearize.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 3
        lib.lba_shard_reduce.argtypes = [C.c_void_p, C.c_double]
        lib.lba_shard_finish.argtypes = [C.c_void_p, C.c_double] + [C.POINTER(C.c_double)] * 3
        lib.lba_shard_accept.argtypes = [C.c_void_p, C.c_int]
        lib.lba_shard_download.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        self._pr = _lba_problem(w)
        h = C.c_void_p()
        _check(lib.lba_shard_create(device, C.byref(self._pr), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.lba_shard_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reduce_len(self):
        return lib.lba_shard_reduce_len(self._h)

    def reduce_buffer_ptr(self):
        return lib.lba_shard_reduce_buffer(self._h)

    def set_local(self, local=True):
        lib.lba_shard_set_local.argtypes = [C.c_void_p, C.c_int]
        _check(lib.lba_shard_set_local(self._h, int(local)))

    def set_reduce_buffer(self, device_ptr):
        lib.lba_shard_set_reduce_buffer.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib.lba_shard_set_reduce_buffer(self._h, device_ptr))

    def hint_lambda(self, lam):
        """the lambda of the first trial after the next linearize(): lets it fold the landmark side of the Schur complement in"""
        _check(lib.lba_shard_hint_lambda(self._h, C.c_double(lam)))

    def linearize(self):
        """returns (chi2_local, max_diag_poses_local, max_diag_landmarks_local)"""
        chi, mp, ml = C.c_double(), C.c_double(), C.c_double()
        _check(lib.lba_shard_linearize(self._h, C.byref(chi), C.byref(mp), C.byref(ml)))
        return chi.value, mp.value, ml.value

    def reduce(self, lam):
        _check(lib.lba_shard_reduce(self._h, lam))

    def finish(self, lam):
        """returns (solved, chi2_local_new, scale_poses, scale_landmarks_local)"""
        chi, sp, sl = C.c_double(), C.c_double(), C.c_double()
        r = lib.lba_shard_finish(self._h, lam, C.byref(chi), C.byref(sp), C.byref(sl))
        _check(min(r, 0))
        return r, chi.value, sp.value, sl.value

    def accept(self, ok):
        _check(lib.lba_shard_accept(self._h, int(ok)))

    STAGES = ("linearize", "schur", "factor", "solve", "update", "reduce", "gaps")
    STAGE_KERNEL = {"linearize": "k_lin_all", "schur": "k_schur_blocks", "factor": "k_chol_flow", "solve": "k_chol_solve_update",
                    "update": "k_update_errors", "reduce": "k_reduce", "gaps": None}

    def profile_enable(self, on=True):
        lib.lba_shard_profile_enable.argtypes = [C.c_void_p, C.c_int]
        _check(lib.lba_shard_profile_enable(self._h, int(on)))

    def profile_read(self):
        """milliseconds per stage (HIP events on the shard's stream) accumulated since profile_enable(True)"""
        lib.lba_shard_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        ms = np.zeros(len(self.STAGES), np.float32)
        _check(lib.lba_shard_profile_read(self._h, _p(ms), len(ms)))
        return dict(zip(self.STAGES, [float(v) for v in ms]))

    def fence_out(self, other_stream):
        lib.lba_shard_fence_out.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib.lba_shard_fence_out(self._h, C.c_void_p(other_stream)))

    def fence_in(self, other_stream):
        lib.lba_shard_fence_in.argtypes = [C.c_void_p, C.c_void_p]
        _check(lib.lba_shard_fence_in(self._h, C.c_void_p(other_stream)))

    def set_async_reduce(self, on=True):
        lib.lba_shard_set_async_reduce.argtypes = [C.c_void_p, C.c_int]
        _check(lib.lba_shard_set_async_reduce(self._h, int(on)))

    def reset(self):
        lib.lba_shard_reset.argtypes = [C.c_void_p]
        _check(lib.lba_shard_reset(self._h))

    ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)

    def optimize(self, allreduce=None, world=1, max_iters=10, lambda_init=0.0, stop_flag=None):
        """lba_shard_optimize: the whole Levenberg loop in the library.  `allreduce(device_ptr, count, op, hip_stream) -> int` is
        called for every exchange (op 0 = sum, 1 = max; reduce `count` doubles at `device_ptr` in place over all ranks, ordered on
        `hip_stream`); None = world size 1.  Returns the stats dict."""
        lib.lba_shard_optimize.argtypes = [C.c_void_p, self.ALLREDUCE_FN, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.POINTER(LbaStats)]
        err = []

        def tramp(user, buf, count, op, stream):
            try:
                return int(allreduce(buf, int(count), int(op), stream) or 0)
            except Exception as e:  # noqa: BLE001  (an exception must not cross the C frame)
                err.append(e)
                return 1
        cb = self.ALLREDUCE_FN(tramp) if allreduce is not None else C.cast(None, self.ALLREDUCE_FN)
        st = LbaStats()
        rc = lib.lba_shard_optimize(self._h, cb, None, int(world), int(max_iters), float(lambda_init), _p(stop_flag), C.byref(st))
        if err:
            raise err[0]
        _check(rc)
        return _stats_dict(st)

    def download(self):
        k = self._pr._keep
        q = np.zeros_like(k["pose_q"]); t = np.zeros_like(k["pose_t"]); pts = np.zeros_like(k["points"])
        chi2 = np.zeros(len(k["edge_point"])); dpos = np.zeros(len(k["edge_point"]), np.uint8)
        _check(lib.lba_shard_download(self._h, _p(q), _p(t), _p(pts), _p(chi2), _p(dpos)))
        return dict(pose_q=q, pose_t=t, points=pts, chi2=chi2, depth_positive=dpos)


class PoseProblem(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("stereo", C.c_void_p),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double)]


class PoseResult(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("inliers", C.c_int32), ("n_bad", C.c_int32),
                ("iterations", C.c_int32 * 4), ("trials", C.c_int32 * 4), ("chi2", C.c_double * 4)]


class PoseDeviceFrames(C.Structure):
    _fields_ = [("d_kps", C.c_void_p), ("d_n", C.c_void_p), ("d_u_right", C.c_void_p), ("cap", C.c_int32),
                ("d_assign", C.c_void_p), ("d_mp_xyz", C.c_void_p), ("mp_cap", C.c_int32), ("d_pose", C.c_void_p),
                ("inv_level_sigma2", C.c_void_p), ("n_levels", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double)]


def _pose_problem(w, pr):
    keep = [np.ascontiguousarray(w["Xw"], np.float64), np.ascontiguousarray(w["obs"], np.float64),
            np.ascontiguousarray(w["inv_sigma2"], np.float64), np.ascontiguousarray(w["stereo"], np.uint8)]
    pr.q[:] = [float(v) for v in w["q"]]
    pr.t[:] = [float(v) for v in w["t"]]
    pr.n = len(keep[0])
    pr.Xw, pr.obs, pr.inv_sigma2, pr.stereo = (a.ctypes.data for a in keep)
    for k in ("fx", "fy", "cx", "cy", "bf", "huber_mono", "huber_stereo"):
        setattr(pr, k, float(w[k]))
    return keep


class PoseSolver:
    """Optimizer::PoseOptimization (reference src/Optimizer.cc:814-1115): motion-only BA, one workgroup per frame."""

    def __init__(self, device=0):
        lib.pose_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.pose_destroy.argtypes = [C.c_void_p]
        lib.pose_optimize_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        h = C.c_void_p()
        _check(lib.pose_create(device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.pose_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, problems):
        """flatten a list of frames once (benchmarks re-run the same batch)"""
        n = len(problems)
        prs = (PoseProblem * n)()
        keep = [_pose_problem(w, prs[i]) for i, w in enumerate(problems)]
        outl = [np.zeros(max(prs[i].n, 1), np.uint8) for i in range(n)]
        ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outl])
        return dict(n=n, prs=prs, keep=keep, outl=outl, ptrs=ptrs, res=(PoseResult * n)())

    def launch(self, prep):
        """the C call alone: upload, one kernel launch, download"""
        _check(lib.pose_optimize_batch(self._h, prep["prs"], prep["n"], prep["res"], prep["ptrs"]))

    def results(self, prep):
        return [dict(q=np.array(r.q[:]), t=np.array(r.t[:]), inliers=r.inliers, n_bad=r.n_bad,
                     iterations=list(r.iterations), trials=list(r.trials), chi2=list(r.chi2),
                     outlier=prep["outl"][i][:prep["prs"][i].n].copy()) for i, r in enumerate(prep["res"])]

    def run(self, prep):
        self.launch(prep)
        return self.results(prep)

    def optimize_batch_device(self, batch, cap, d_kps, d_n, d_assign, d_mp_xyz, mp_cap, d_pose, inv_level_sigma2, cam, d_pose_out, d_inliers,
                              d_outlier, stream, d_u_right=None, d_results=None):
        """pose_optimize_batch_device: every d_* argument is a device address (int); `cam` = dict(fx, fy, cx, cy, bf[, huber_mono,
        huber_stereo]); inv_level_sigma2 a host float array.  Only enqueues on `stream`."""
        lib.pose_optimize_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        isig = np.ascontiguousarray(inv_level_sigma2, np.float32)
        f = PoseDeviceFrames(d_kps, d_n, d_u_right, cap, d_assign, d_mp_xyz, mp_cap, d_pose, isig.ctypes.data, len(isig),
                             float(cam["fx"]), float(cam["fy"]), float(cam["cx"]), float(cam["cy"]), float(cam.get("bf", 0.0)),
                             float(cam.get("huber_mono", np.float32(np.sqrt(5.991)))), float(cam.get("huber_stereo", np.float32(np.sqrt(7.815)))))
        _check(lib.pose_optimize_batch_device(self._h, C.byref(f), batch, d_pose_out, d_inliers, d_outlier, d_results, stream))

    def last_kernel_ms(self):
        lib.pose_last_kernel_ms.restype = C.c_float
        lib.pose_last_kernel_ms.argtypes = [C.c_void_p]
        return float(lib.pose_last_kernel_ms(self._h))

    def optimize_batch(self, problems):
        return self.run(self.prepare(problems))

    def optimize(self, w):
        return self.optimize_batch([w])[0]


class OrbvVocabulary(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("L", C.c_int32), ("child_off", C.c_void_p), ("child_id", C.c_void_p),
                ("desc", C.c_void_p), ("weight", C.c_void_p), ("word_id", C.c_void_p)]


def vocabulary_struct(voc, cls=OrbvVocabulary):
    keep = [np.ascontiguousarray(voc["child_off"], np.int32), np.ascontiguousarray(voc["child_id"], np.uint32),
            np.ascontiguousarray(voc["desc"], np.uint8), np.ascontiguousarray(voc["weight"], np.float64),
            np.ascontiguousarray(voc["word_id"], np.int32)]
    s = cls(int(voc["n_nodes"]), int(voc["L"]), *[a.ctypes.data for a in keep])
    s._keep = keep
    return s


class Vocabulary:
    """DBoW2 TemplatedVocabulary::transform for ORB descriptors (TF_IDF, L1) on a device-resident tree."""

    def __init__(self, voc, device=0):
        s = vocabulary_struct(voc)
        h = C.c_void_p()
        _check(lib.orbv_create(device, C.byref(s), C.byref(h)))
        self._h = h
        lib.orbv_destroy.argtypes = [C.c_void_p]

    def close(self):
        if getattr(self, "_h", None):
            lib.orbv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform_features(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        word = np.zeros(max(n, 1), np.uint32); weight = np.zeros(max(n, 1)); node = np.zeros(max(n, 1), np.uint32)
        _check(lib.orbv_transform_features(self._h, _p(desc), n, int(levelsup), _p(word), _p(weight), _p(node)))
        return word[:n], weight[:n], node[:n]

    def transform(self, desc, levelsup=4):
        """returns (bow_id, bow_val), (fv_node, fv_off, fv_feat)"""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        m = max(n, 1)
        bi = np.zeros(m, np.uint32); bv = np.zeros(m); fn = np.zeros(m, np.uint32); fo = np.zeros(m + 1, np.int32); ff = np.zeros(m, np.uint32)
        nb, nf = C.c_int32(), C.c_int32()
        used = _check(lib.orbv_transform(self._h, _p(desc), n, int(levelsup), _p(bi), _p(bv), C.byref(nb), _p(fn), _p(fo), _p(ff), C.byref(nf)))
        return (bi[:nb.value], bv[:nb.value]), (fn[:nf.value], fo[:nf.value + 1], ff[:used])

    def transform_batch_device(self, d_desc, d_n, batch, cap, levelsup, d_bow_id, d_bow_val, d_n_bow, d_fv_node, d_fv_off, d_fv_feat,
                               d_n_fv, stream=None):
        _check(lib.orbv_transform_batch_device(self._h, C.c_void_p(d_desc), C.c_void_p(d_n), int(batch), int(cap), int(levelsup),
                                               C.c_void_p(d_bow_id), C.c_void_p(d_bow_val), C.c_void_p(d_n_bow), C.c_void_p(d_fv_node),
                                               C.c_void_p(d_fv_off), C.c_void_p(d_fv_feat), C.c_void_p(d_n_fv), C.c_void_p(stream or 0)))


IMU_DTYPE = np.dtype([("ts", "i8"), ("gyro", "f4", (3,)), ("acce", "f4", (3,))])      # OrbeImuSample, the 32-byte wire record


class PacketCodec:
    """The fork's edge-SLAM packets (class SlamPktVI, reference include/Socket/slampkt_vi.h) packed / unpacked on the device."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib.orbe_create(device, C.byref(h)))
        self._h = h
        lib.orbe_destroy.argtypes = [C.c_void_p]

    def close(self):
        if getattr(self, "_h", None):
            lib.orbe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def packet_bytes(n_pts, n_imu=0):
        return lib.orbe_packet_bytes(int(n_pts), int(n_imu))

    def pack_batch(self, kps, desc, n, frame_id, timestamp, imu=None, imu_off=None, stride=None):
        """kps [B][cap] KP_DTYPE, desc [B][cap][32], n [B]; returns (payload [B][stride] u8, len [B], head [B][2], status [B])"""
        kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
        B, cap = kps.shape
        n = np.ascontiguousarray(n, np.int32); frame_id = np.ascontiguousarray(frame_id, np.int32)
        timestamp = np.ascontiguousarray(timestamp, np.int64)
        assert desc.shape == (B, cap, 32) and n.shape == (B,) and frame_id.shape == (B,) and timestamp.shape == (B,)
        if imu is not None:
            imu = np.ascontiguousarray(imu, IMU_DTYPE); imu_off = np.ascontiguousarray(imu_off, np.int32)
            assert imu_off.shape == (B + 1,) and imu_off[-1] <= len(imu)
            max_imu = int(np.diff(imu_off).max()) if B else 0
        else:
            max_imu = 0
        if stride is None:
            stride = (self.packet_bytes(int(n.max()) if B else 0, max_imu) + 3) & ~3
        payload = np.zeros((B, stride), np.uint8); ln = np.zeros(B, np.int32); head = np.zeros((B, 2), np.uint8); st = np.zeros(B, np.int32)
        _check(lib.orbe_pack_batch(self._h, _p(kps), _p(desc), _p(n), B, cap, _p(frame_id), _p(timestamp),
                                   _p(imu) if imu is not None else None, _p(imu_off) if imu is not None else None,
                                   _p(payload), int(stride), _p(ln), _p(head), _p(st)))
        return payload, ln, head, st

    def unpack_batch(self, payload, ln, cap, imu_cap=0):
        """payload [B][stride] u8, ln [B]; returns dict(kps, desc, n, frame_id, timestamp, imu, n_imu, status)"""
        payload = np.ascontiguousarray(payload, np.uint8); ln = np.ascontiguousarray(ln, np.int32)
        B, stride = payload.shape
        kps = np.zeros((B, cap), KP_DTYPE); desc = np.zeros((B, cap, 32), np.uint8); n = np.zeros(B, np.int32)
        fid = np.zeros(B, np.int32); ts = np.zeros(B, np.int64); imu = np.zeros((B, max(imu_cap, 1)), IMU_DTYPE)
        ni = np.zeros(B, np.int32); st = np.zeros(B, np.int32)
        _check(lib.orbe_unpack_batch(self._h, _p(payload), int(stride), _p(ln), B, int(cap), int(imu_cap), _p(kps), _p(desc), _p(n),
                                     _p(fid), _p(ts), _p(imu) if imu_cap > 0 else None, _p(ni), _p(st)))
        return dict(kps=kps, desc=desc, n=n, frame_id=fid, timestamp=ts, imu=imu, n_imu=ni, status=st)

    def pack_batch_device(self, d_kps, d_desc, d_n, batch, cap, d_frame_id, d_timestamp, d_imu, d_imu_off, d_payload, stride, d_len, d_head,
                          d_status, stream=None):
        _check(lib.orbe_pack_batch_device(self._h, C.c_void_p(d_kps), C.c_void_p(d_desc), C.c_void_p(d_n), int(batch), int(cap),
                                          C.c_void_p(d_frame_id), C.c_void_p(d_timestamp), C.c_void_p(d_imu or 0), C.c_void_p(d_imu_off or 0),
                                          C.c_void_p(d_payload), int(stride), C.c_void_p(d_len), C.c_void_p(d_head or 0), C.c_void_p(d_status),
                                          C.c_void_p(stream or 0)))

    class _Camera(C.Structure):
        _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("k", C.c_float * 5),
                    ("fx_new", C.c_float), ("fy_new", C.c_float), ("cx_new", C.c_float), ("cy_new", C.c_float)]

    def undistort_batch_device(self, d_kps, d_n, batch, cap, K, dist, Knew, d_kps_un, stream=None):
        """Frame::UndistortKeyPoints on device arrays: K / Knew = (fx, fy, cx, cy), dist = (k1, k2, p1, p2, k3)"""
        cam = self._Camera(float(K[0]), float(K[1]), float(K[2]), float(K[3]), (C.c_float * 5)(*[float(v) for v in dist]),
                           float(Knew[0]), float(Knew[1]), float(Knew[2]), float(Knew[3]))
        _check(lib.orbe_undistort_batch_device(self._h, C.c_void_p(d_kps), C.c_void_p(d_n), int(batch), int(cap), C.byref(cam), C.c_void_p(d_kps_un),
                                               C.c_void_p(stream or 0)))

    def unpack_batch_device(self, d_payload, stride, d_len, batch, cap, imu_cap, d_kps, d_desc, d_n, d_frame_id, d_timestamp, d_imu, d_n_imu,
                            d_status, stream=None):
        _check(lib.orbe_unpack_batch_device(self._h, C.c_void_p(d_payload), int(stride), C.c_void_p(d_len), int(batch), int(cap), int(imu_cap),
                                            C.c_void_p(d_kps), C.c_void_p(d_desc), C.c_void_p(d_n), C.c_void_p(d_frame_id),
                                            C.c_void_p(d_timestamp), C.c_void_p(d_imu or 0), C.c_void_p(d_n_imu), C.c_void_p(d_status),
                                            C.c_void_p(stream or 0)))


class _LibaLink(C.Structure):
    _fields_ = [("kf1", C.c_int32), ("kf2", C.c_int32), ("dR", C.c_float * 9), ("dV", C.c_float * 3), ("dP", C.c_float * 3),
                ("JRg", C.c_float * 9), ("JVg", C.c_float * 9), ("JVa", C.c_float * 9), ("JPg", C.c_float * 9), ("JPa", C.c_float * 9),
                ("dT", C.c_float), ("bias0", C.c_float * 6), ("info9", C.c_double * 81), ("info_gyro", C.c_double * 9), ("info_acc", C.c_double * 9),
                ("robust", C.c_uint8)]


class _LibaProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("Rwb", C.c_void_p), ("twb", C.c_void_p), ("vel", C.c_void_p), ("bg", C.c_void_p), ("ba", C.c_void_p),
                ("pose_fixed", C.c_void_p), ("has_imu", C.c_void_p), ("imu_fixed", C.c_void_p),
                ("Rcb", C.c_double * 9), ("tcb", C.c_double * 3), ("tbc", C.c_double * 3),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("n_points", C.c_int32), ("points", C.c_void_p), ("n_edges", C.c_int32), ("edge_kf", C.c_void_p), ("edge_point", C.c_void_p),
                ("edge_obs", C.c_void_p), ("edge_inv_sigma2", C.c_void_p), ("edge_stereo", C.c_void_p),
                ("n_links", C.c_int32), ("links", C.c_void_p),
                ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("huber_inertial", C.c_double), ("lambda_init", C.c_double), ("max_iters", C.c_int32)]


class _LibaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("stop_reason", C.c_int32),
                ("lambda_", C.c_double), ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("chi2_trace", C.c_double * 16)]


def _liba_problem(pr):
    dt = dict(Rwb=np.float64, twb=np.float64, vel=np.float64, bg=np.float64, ba=np.float64, pose_fixed=np.uint8, has_imu=np.uint8, imu_fixed=np.uint8,
              points=np.float64, edge_kf=np.int32, edge_point=np.int32, edge_obs=np.float64, edge_inv_sigma2=np.float64, edge_stereo=np.uint8)
    keep = {k: np.ascontiguousarray(pr[k], t) for k, t in dt.items()}
    links = (_LibaLink * max(len(pr["links"]), 1))()
    for L, d in zip(links, pr["links"]):
        L.kf1, L.kf2, L.dT, L.robust = int(d["kf1"]), int(d["kf2"]), float(d["dT"]), int(d["robust"])
        for name in ("dR", "dV", "dP", "JRg", "JVg", "JVa", "JPg", "JPa", "bias0"):
            getattr(L, name)[:] = np.asarray(d[name], np.float32).ravel().tolist()
        for name in ("info9", "info_gyro", "info_acc"):
            getattr(L, name)[:] = np.asarray(d[name], np.float64).ravel().tolist()
    s = _LibaProblem()
    s.n_kf = int(pr["n_kf"])
    for k in dt:
        setattr(s, k, keep[k].ctypes.data)
    s.Rcb[:] = np.asarray(pr["Rcb"], np.float64).ravel().tolist(); s.tcb[:] = list(map(float, pr["tcb"])); s.tbc[:] = list(map(float, pr["tbc"]))
    s.fx, s.fy, s.cx, s.cy, s.bf = pr["fx"], pr["fy"], pr["cx"], pr["cy"], pr["bf"]
    s.n_points = len(keep["points"]); s.n_edges = len(keep["edge_kf"]); s.n_links = len(pr["links"]); s.links = C.addressof(links)
    s.huber_mono, s.huber_stereo, s.huber_inertial = pr["huber_mono"], pr["huber_stereo"], pr["huber_inertial"]
    s.lambda_init, s.max_iters = float(pr["lambda_init"]), int(pr["max_iters"])
    s._keep = (keep, links)
    return s


class InertialSolver:
    """The numerical core of Optimizer::LocalInertialBA (reference src/Optimizer.cc:2383-2958) on the device."""

    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib.liba_create(device, C.byref(h)))
        self._h = h
        lib.liba_destroy.argtypes = [C.c_void_p]

    def close(self):
        if getattr(self, "_h", None):
            lib.liba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve(self, pr):
        s = _liba_problem(pr)
        n, m, ne = s.n_kf, s.n_points, s.n_edges
        Rwb = np.zeros((n, 3, 3)); twb = np.zeros((n, 3)); vel = np.zeros((n, 3)); bg = np.zeros((n, 3)); ba = np.zeros((n, 3))
        pts = np.zeros((max(m, 1), 3)); chi2 = np.zeros(max(ne, 1)); dpos = np.zeros(max(ne, 1), np.uint8)
        st = _LibaStats()
        _check(lib.liba_solve(self._h, C.byref(s), _p(Rwb), _p(twb), _p(vel), _p(bg), _p(ba), _p(pts), _p(chi2), _p(dpos), C.byref(st)))
        stats = dict(iterations=st.iterations, trials=st.trials, stop_reason=st.stop_reason, lambda_=st.lambda_, chi2_initial=st.chi2_initial,
                     chi2_final=st.chi2_final)
        return dict(Rwb=Rwb, twb=twb, vel=vel, bg=bg, ba=ba, points=pts[:m], chi2=chi2[:ne], depth_positive=dpos[:ne], stats=stats)


class LibaOutputs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("Rwb", "twb", "vel", "bg", "ba", "points", "chi2_per_edge", "depth_positive")]


class LibaBatch:
    """liba_solve_batch: many LocalInertialBA windows (one per map / client session) per launch."""

    def __init__(self, device=0):
        lib.liba_batch_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.liba_batch_destroy.argtypes = [C.c_void_p]
        lib.liba_solve_batch.argtypes = [C.c_void_p, C.POINTER(_LibaProblem), C.POINTER(LibaOutputs), C.c_int, C.POINTER(_LibaStats)]
        lib.liba_batch_last_device_ms.argtypes = [C.c_void_p]
        lib.liba_batch_last_device_ms.restype = C.c_double
        h = C.c_void_p()
        _check(lib.liba_batch_create(device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.liba_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, windows, want_outputs=True):
        """marshal once (a benchmark re-solves the same windows)"""
        n = len(windows)
        prs = [_liba_problem(w) for w in windows]
        arr = (_LibaProblem * n)(*prs)
        outs, oarr = [], (LibaOutputs * n)()
        for i, s in enumerate(prs):
            nk, m, ne = s.n_kf, s.n_points, s.n_edges
            o = dict(Rwb=np.zeros((nk, 3, 3)), twb=np.zeros((nk, 3)), vel=np.zeros((nk, 3)), bg=np.zeros((nk, 3)), ba=np.zeros((nk, 3)),
                     points=np.zeros((max(m, 1), 3)), chi2=np.zeros(max(ne, 1)), depth_positive=np.zeros(max(ne, 1), np.uint8))
            outs.append(o)
            if want_outputs:
                oarr[i] = LibaOutputs(*[o[k].ctypes.data for k in ("Rwb", "twb", "vel", "bg", "ba", "points", "chi2", "depth_positive")])
        return dict(n=n, prs=prs, arr=arr, outs=outs, oarr=oarr, stats=(_LibaStats * n)(), want=want_outputs)

    def run(self, prep):
        _check(lib.liba_solve_batch(self._h, prep["arr"], prep["oarr"] if prep["want"] else None, prep["n"], prep["stats"]))
        res = []
        for i, o in enumerate(prep["outs"]):
            s, st = prep["prs"][i], prep["stats"][i]
            stats = dict(iterations=st.iterations, trials=st.trials, stop_reason=st.stop_reason, lambda_=st.lambda_, chi2_initial=st.chi2_initial,
                         chi2_final=st.chi2_final)
            res.append(dict(Rwb=o["Rwb"], twb=o["twb"], vel=o["vel"], bg=o["bg"], ba=o["ba"], points=o["points"][:s.n_points], chi2=o["chi2"][:s.n_edges],
                            depth_positive=o["depth_positive"][:s.n_edges], stats=stats))
        return res

    def solve(self, windows):
        return self.run(self.prepare(windows))

    def last_device_ms(self):
        return float(lib.liba_batch_last_device_ms(self._h))


class _LibaPoseProblem(C.Structure):
    _fields_ = [("Rwb", C.c_double * 18), ("twb", C.c_double * 6), ("vel", C.c_double * 6), ("bg", C.c_double * 6), ("ba", C.c_double * 6),
                ("Rcb", C.c_double * 9), ("tcb", C.c_double * 3), ("tbc", C.c_double * 3),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("bf", C.c_double),
                ("n", C.c_int32), ("Xw", C.c_void_p), ("obs", C.c_void_p), ("inv_sigma2", C.c_void_p), ("stereo", C.c_void_p), ("close_point", C.c_void_p),
                ("link", _LibaLink), ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("rec_init", C.c_int32),
                ("last_frame", C.c_int32), ("prior_Rwb", C.c_double * 9), ("prior_twb", C.c_double * 3), ("prior_vel", C.c_double * 3),
                ("prior_bg", C.c_double * 3), ("prior_ba", C.c_double * 3), ("prior_H", C.c_double * 225)]


def _fill_liba_pose(s, pr, keep):
    arrs = {k: np.ascontiguousarray(pr[k], t) for k, t in (("Xw", np.float64), ("obs", np.float64), ("inv_sigma2", np.float64), ("stereo", np.uint8),
                                                           ("close_point", np.uint8))}
    keep.append(arrs)
    for k, m in (("Rwb", 18), ("twb", 6), ("vel", 6), ("bg", 6), ("ba", 6), ("Rcb", 9), ("tcb", 3), ("tbc", 3)):
        getattr(s, k)[:] = np.asarray(pr[k], np.float64).ravel().tolist()
    s.fx, s.fy, s.cx, s.cy, s.bf = pr["fx"], pr["fy"], pr["cx"], pr["cy"], pr["bf"]
    s.n = len(arrs["Xw"])
    for k in arrs:
        setattr(s, k, arrs[k].ctypes.data)
    d, L = pr["link"], s.link
    L.kf1, L.kf2, L.dT, L.robust = int(d["kf1"]), int(d["kf2"]), float(d["dT"]), int(d["robust"])
    for name in ("dR", "dV", "dP", "JRg", "JVg", "JVa", "JPg", "JPa", "bias0"):
        getattr(L, name)[:] = np.asarray(d[name], np.float32).ravel().tolist()
    for name in ("info9", "info_gyro", "info_acc"):
        getattr(L, name)[:] = np.asarray(d[name], np.float64).ravel().tolist()
    s.huber_mono, s.huber_stereo, s.rec_init = pr["huber_mono"], pr["huber_stereo"], int(pr["rec_init"])
    s.last_frame = int(pr.get("last_frame", 0))
    if s.last_frame:
        for k in ("prior_Rwb", "prior_twb", "prior_vel", "prior_bg", "prior_ba", "prior_H"):
            getattr(s, k)[:] = np.asarray(pr[k], np.float64).ravel().tolist()


def _pose_inertial_prepare(self, problems):
    """flatten a list of frames once (what a C++ caller holds anyway; benchmarks re-run the same batch)"""
    B = len(problems)
    arr = (_LibaPoseProblem * B)()
    keep = []
    for s, pr in zip(arr, problems):
        _fill_liba_pose(s, pr, keep)
    tot = sum(int(s.n) for s in arr)
    N = 30 if int(problems[0].get("last_frame", 0)) else 15         # last-frame variant: the 30 x 30 Hessian before Optimizer::Marginalize
    return dict(B=B, arr=arr, keep=keep, tot=tot, Rwb=np.zeros((B, 3, 3)), twb=np.zeros((B, 3)), vel=np.zeros((B, 3)), bg=np.zeros((B, 3)),
                ba=np.zeros((B, 3)), out=np.zeros(max(tot, 1), np.uint8), H=np.zeros((B, N, N)), inl=np.zeros(B, np.int32), nb=np.zeros(B, np.int32))


def _pose_inertial_launch(self, q):
    """the C call alone: one copy in, one kernel launch, one copy out"""
    _check(lib.liba_pose_optimize_batch(self._h, q["arr"], q["B"], _p(q["Rwb"]), _p(q["twb"]), _p(q["vel"]), _p(q["bg"]), _p(q["ba"]),
                                        _p(q["out"]), _p(q["H"]), _p(q["inl"]), _p(q["nb"])))


def _pose_inertial_results(self, q):
    res, o = [], 0
    for b in range(q["B"]):
        n = int(q["arr"][b].n)
        res.append(dict(Rwb=q["Rwb"][b].copy(), twb=q["twb"][b].copy(), vel=q["vel"][b].copy(), bg=q["bg"][b].copy(), ba=q["ba"][b].copy(),
                        outlier=q["out"][o:o + n].copy(), H=q["H"][b].copy(), n_bad=int(q["nb"][b]), inliers=int(q["inl"][b])))
        o += n
    return res


def _pose_inertial_batch(self, problems):
    """Optimizer::PoseInertialOptimizationLastKeyFrame for a batch of frames: list of dict(Rwb, twb, vel, bg, ba, outlier, H, n_bad, inliers)"""
    q = _pose_inertial_prepare(self, problems)
    _pose_inertial_launch(self, q)
    return _pose_inertial_results(self, q)


InertialSolver.pose_prepare = _pose_inertial_prepare
InertialSolver.pose_launch = _pose_inertial_launch
InertialSolver.pose_results = _pose_inertial_results
InertialSolver.pose_optimize_batch = _pose_inertial_batch
