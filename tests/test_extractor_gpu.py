"""Parity of the HIP extractor (through the C ABI) against the CPU oracle: bit-exact keypoints
(x, y, octave, angle, response, size), descriptor bytes and output order, stage by stage and end to end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("x", "y", "size", "angle", "response", "octave", "class_id")


def _assert_kps_equal(a, b, what):
    assert len(a) == len(b), "%s: count %d vs %d" % (what, len(a), len(b))
    for f in FIELDS:
        if not np.array_equal(a[f], b[f]):
            bad = np.nonzero(a[f] != b[f])[0]
            raise AssertionError("%s: field %s differs at %d rows, first %d: %r vs %r" %
                                 (what, f, len(bad), bad[0], a[f][bad[0]], b[f][bad[0]]))


@pytest.fixture(scope="module")
def gpu_ex(pkg):
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    yield ex
    ex.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_stagewise_640x480(pkg, oracle, synth, gpu_ex, seed):
    img = synth.make_frame(seed)
    oex = oracle.extractor(1000, 1.2, 8, 20, 7)
    r0, okps, odesc = oex.extract(img, (0, 1000))
    mono, kps, desc = gpu_ex(img, (0, 1000))
    for l in range(8):
        assert gpu_ex.level_size(l) == oex.level_size(l)
        np.testing.assert_array_equal(gpu_ex.pyramid_level(l), oex.level_image(l), err_msg="pyramid level %d" % l)
    for l in range(8):
        oc = oex.level_candidates(l)
        gc = gpu_ex.candidates(l)
        assert len(gc) == len(oc), "level %d candidates %d vs %d" % (l, len(gc), len(oc))
        for f in ("x", "y", "response"):
            np.testing.assert_array_equal(gc[f], oc[f], err_msg="FAST candidates level %d field %s" % (l, f))
    for l in range(8):
        ob = oex.level_blurred(l)
        if ob is not None:
            np.testing.assert_array_equal(gpu_ex.blurred_level(l), ob, err_msg="blur level %d" % l)
    for l in range(8):
        _assert_kps_equal(gpu_ex.level_keypoints(l), oex.level_keypoints(l), "octree+angle level %d" % l)
    assert mono == r0
    _assert_kps_equal(kps, okps, "final keypoints")
    np.testing.assert_array_equal(desc, odesc)


@pytest.mark.parametrize("shape,params,lap", [
    ((120, 160), (300, 1.2, 4, 20, 7), (0, 1000)),
    ((240, 376), (500, 1.2, 8, 20, 7), (0, 0)),          # rectified-stereo style lapping: forward order
    ((480, 752), (1200, 1.2, 8, 20, 7), (100, 400)),     # fisheye style overlap bounds
    ((97, 131), (150, 1.5, 3, 15, 5), (0, 1000)),
    ((480, 640), (5000, 1.2, 8, 20, 7), (0, 1000)),      # mono initialisation extractor (5 x nFeatures)
    ((376, 1241), (2000, 1.2, 8, 20, 7), (0, 1000)),     # KITTI aspect: nIni = round(w/h) = 4 octree roots (:559)
    ((300, 1000), (800, 1.2, 6, 20, 7), (0, 0)),         # nIni = 3
    ((640, 480), (1000, 1.2, 8, 20, 7), (0, 1000)),      # portrait: nIni = round(0.73) = 1
    ((900, 400), (600, 1.2, 4, 20, 7), (0, 1000)),       # nIni = round(0.42) = 0: the reference would divide by zero; both sides yield no keypoints
    ((80, 90), (100, 1.2, 3, 20, 7), (0, 1000)),         # upper levels too small for a single 35-px cell
    ((480, 640), (1000, 2.0, 4, 20, 7), (0, 1000)),      # scale factor 2
    ((480, 640), (1000, 1.2, 8, 40, 12), (0, 1000)),     # other FAST thresholds
])
def test_end_to_end_shapes(pkg, oracle, synth, shape, params, lap):
    h, w = shape
    img = synth.make_frame(11, w, h)
    oex = oracle.extractor(*params)
    r0, okps, odesc = oex.extract(img, lap)
    ex = pkg.Extractor(*params)
    try:
        mono, kps, desc = ex(img, lap)
    finally:
        ex.close()
    assert mono == r0
    _assert_kps_equal(kps, okps, "final keypoints %r" % (shape,))
    np.testing.assert_array_equal(desc, odesc)


def test_flat_and_empty(pkg, oracle, gpu_ex):
    flat = np.full((480, 640), 90, np.uint8)
    mono, kps, desc = gpu_ex(flat)
    assert mono == 0 and len(kps) == 0 and desc.shape == (0, 32)
    mono, kps, desc = gpu_ex(np.zeros((0, 0), np.uint8))
    assert mono == -1                       # reference returns -1 for an empty image
    n = pkg.capi.C.c_int(); m = pkg.capi.C.c_int()
    r = pkg.lib.orbx_extract(gpu_ex._h, None, 0, 0, 0, 0, 1000, None, None, 0, pkg.capi.C.byref(n), pkg.capi.C.byref(m))
    assert r == -1 and m.value == -1


def test_batch_matches_single(pkg, oracle, synth):
    imgs = synth.make_frames(6, seed0=20)
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    try:
        mono, n, kps, desc = ex.extract_batch(imgs)
        oex = oracle.extractor(1000, 1.2, 8, 20, 7)
        for b in range(len(imgs)):
            r0, okps, odesc = oex.extract(imgs[b])
            assert mono[b] == r0 and n[b] == len(okps)
            _assert_kps_equal(kps[b, :n[b]], okps, "batch frame %d" % b)
            np.testing.assert_array_equal(desc[b, :n[b]], odesc)
    finally:
        ex.close()


def test_pyramid_border(pkg, oracle, synth, gpu_ex):
    img = synth.make_frame(3)
    gpu_ex(img)
    lvl = gpu_ex.pyramid_level(2, border=19)
    inner = gpu_ex.pyramid_level(2)
    np.testing.assert_array_equal(lvl, np.pad(inner, 19, mode="reflect"))     # numpy 'reflect' == BORDER_REFLECT_101


@pytest.mark.parametrize("args", [(1000, 1.2, 8, 20, 7), (5000, 1.2, 8, 20, 7), (150, 1.5, 3, 20, 7), (2000, 1.1, 12, 15, 5), (500, 2.0, 4, 20, 7)])
def test_constructor_tables_match_oracle(pkg, oracle, args):
    """E0: the tables Frame / KeyFrame read through the getters (include/ORBextractor.h:61-81, src/ORBextractor.cc:409-445) --
    mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2 and mnFeaturesPerLevel -- are the oracle's, bit for bit."""
    ex = pkg.Extractor(*args)
    try:
        t = oracle.extractor(*args).tables()
        assert ex.GetLevels() == args[2]
        assert np.float32(ex.GetScaleFactor()) == np.float32(args[1])
        for name, got in (("scale", ex.GetScaleFactors()), ("inv_scale", ex.GetInverseScaleFactors()),
                          ("sigma2", ex.GetScaleSigmaSquares()), ("inv_sigma2", ex.GetInverseScaleSigmaSquares())):
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got.view(np.uint32), t[name].view(np.uint32), err_msg=name)
        np.testing.assert_array_equal(ex.features_per_level(), t["nfeat"])
        assert ex.features_per_level().sum() == args[0]
    finally:
        ex.close()


@pytest.mark.parametrize("cap", [1, 7, 40])
def test_fast_corner_list_overflow_path(pkg, oracle, synth, cap):
    """the FAST kernel lists at most `corner_cap` corners per wave and scans all the pixels of the wave's rows beyond that:
    with a tiny capacity every strip takes that path, and candidates, keypoints and descriptors stay the oracle's"""
    img = synth.make_frame(4)
    oex = oracle.extractor(1000, 1.2, 8, 20, 7)
    r0, okps, odesc = oex.extract(img, (0, 1000))
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    try:
        ex.debug_set_fast_corner_cap(cap)
        mono, kps, desc = ex(img, (0, 1000))
        for l in range(8):
            oc, gc = oex.level_candidates(l), ex.candidates(l)
            assert len(oc) == len(gc)
            for f in ("x", "y", "response"):
                np.testing.assert_array_equal(gc[f], oc[f], err_msg="level %d %s" % (l, f))
        _assert_kps_equal(kps, okps, "overflow path")
        np.testing.assert_array_equal(desc, odesc)
    finally:
        ex.close()


def test_fast_threshold_fallback_cells(pkg, oracle):
    """cells without a keypoint at iniThFAST are searched again at minThFAST (:843-846): a faint texture (contrast between
    the two thresholds) next to strong corners, flat cells, and a frame that is faint everywhere"""
    rs = np.random.RandomState(12)
    img = np.full((240, 320), 100, np.uint8)
    img[:, :160] = (100 + 12 * (rs.uniform(size=(240, 160)) < 0.5)).astype(np.uint8)       # |step| 12: corners at 7, none at 20
    img[40:200:16, 200:300:16] = 255                                                        # isolated bright dots: corners at 20
    faint = (100 + 10 * (rs.uniform(size=(240, 320)) < 0.5)).astype(np.uint8)
    for im in (img, faint, np.full((240, 320), 7, np.uint8)):
        oex = oracle.extractor(500, 1.2, 6, 20, 7)
        r0, okps, odesc = oex.extract(im, (0, 1000))
        ex = pkg.Extractor(500, 1.2, 6, 20, 7)
        try:
            mono, kps, desc = ex(im, (0, 1000))
            for l in range(6):
                oc, gc = oex.level_candidates(l), ex.candidates(l)
                assert len(oc) == len(gc), "level %d: %d vs %d candidates" % (l, len(gc), len(oc))
                for f in ("x", "y", "response"):
                    np.testing.assert_array_equal(gc[f], oc[f], err_msg="level %d %s" % (l, f))
            _assert_kps_equal(kps, okps, "fallback")
            np.testing.assert_array_equal(desc, odesc)
        finally:
            ex.close()


def _adversarial_counts(n):
    # sorted, reversed, all-equal, organ-pipe and a median-of-3 killer: the depth limit / heapsort fallback
    seqs = [np.arange(n), np.arange(n)[::-1], np.zeros(n), np.minimum(np.arange(n), n - np.arange(n))]
    k = n // 2
    killer = np.zeros(n, np.int64)
    for i in range(k):
        killer[i] = i + 1 if i % 2 == 0 else k + i + (1 if k % 2 == 0 else 0)
        killer[k + i] = (i + 1) * 2
    seqs.append(killer)
    return seqs


@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 18, 33, 64, 65, 100, 129, 217, 500, 1085, 3000])
def test_wave_sort_is_std_sort(pkg, oracle, n):
    """the octree's wave-parallel sort (ranked Hoare partitions + stable leaf ranks) moves elements exactly like libstdc++'s
    std::sort with the reference's compareNodes, including the order of tied elements (ORBextractor.cc:538-553, :700)"""
    rs = np.random.RandomState(n)
    cases = []
    for trial in range(6):
        cases.append((rs.randint(2, 2 + max(1, (trial + 1) * 3), n), rs.randint(0, 1 + trial * 2, n) * 19))
    cases.append((rs.randint(2, 2000, n), rs.randint(0, 600, n)))
    if n >= 500:
        cases += [(s, np.zeros(n)) for s in _adversarial_counts(n)]
    for count, ulx in cases:
        count = np.ascontiguousarray(count).astype(np.int32); ulx = np.ascontiguousarray(ulx).astype(np.int32)
        tag = np.arange(n, dtype=np.int32)
        c1, u1, t1 = count.copy(), ulx.copy(), tag.copy()
        c2, u2, t2 = count.copy(), ulx.copy(), tag.copy()
        oracle.lib.orb_oracle_sort_nodes(c1.ctypes.data, u1.ctypes.data, t1.ctypes.data, n)
        assert pkg.lib.orbx_debug_wave_sort(c2.ctypes.data, u2.ctypes.data, t2.ctypes.data, n) == 0
        np.testing.assert_array_equal(c2, c1); np.testing.assert_array_equal(u2, u1); np.testing.assert_array_equal(t2, t1)


@pytest.mark.parametrize("B", [40, 72])
def test_large_batch_launch_paths(pkg, oracle, synth, B):
    """batches of >= 32 frames run the blur on the launch stream beside the octree -- whose upper levels are a launch of their
    own with a smaller node pool -- and from 64 frames on the upper pyramid levels come from the fused resize tail: the same key
    points and descriptors as the oracle, frame by frame"""
    distinct = [synth.make_frame(900 + i) for i in range(6)]
    oex = oracle.extractor(1000, 1.2, 8, 20, 7)
    ref = [oex.extract(im, (0, 1000)) for im in distinct]
    imgs = np.stack([distinct[i % 6] for i in range(B)])
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    try:
        for rep in range(2):            # twice: the second call reuses every buffer and stream
            mono, n, kps, desc = ex.extract_batch(imgs)
            for b in range(B):
                r0, k0, d0 = ref[b % 6]
                assert mono[b] == r0 and n[b] == len(k0)
                _assert_kps_equal(kps[b, :n[b]], k0, "frame %d" % b)
                np.testing.assert_array_equal(desc[b, :n[b]], d0)
    finally:
        ex.close()


@pytest.mark.parametrize("args,size", [((500, 1.5, 5, 20, 7), (376, 240)), ((1500, 1.2, 10, 12, 5), (752, 480)), ((300, 2.0, 3, 20, 7), (320, 240))])
def test_large_batch_other_geometries(pkg, oracle, synth, args, size):
    """the large-batch schedule (resize tail, split FAST / octree launches, side streams) on pyramids with other level counts"""
    B = 64
    distinct = [synth.make_frame(950 + i, *size) for i in range(4)]
    oex = oracle.extractor(*args)
    ref = [oex.extract(im, (0, 1000)) for im in distinct]
    imgs = np.stack([distinct[i % 4] for i in range(B)])
    ex = pkg.Extractor(*args)
    try:
        mono, n, kps, desc = ex.extract_batch(imgs)
        for b in range(B):
            r0, k0, d0 = ref[b % 4]
            assert mono[b] == r0 and n[b] == len(k0)
            _assert_kps_equal(kps[b, :n[b]], k0, "frame %d" % b)
            np.testing.assert_array_equal(desc[b, :n[b]], d0)
    finally:
        ex.close()


@pytest.mark.parametrize("B,tail_delay,mode", [(64, 0, 1), (96, 0, 1), (96, 400, 1), (96, 400, 0), (64, 0, 0)])
def test_large_batch_device_api_in_place(pkg, oracle, synth, monkeypatch, B, tail_delay, mode):
    """the bench's path: device-resident frames read in place (level 0 = the caller's buffer, copied into the pyramid by the
    blur), the large-batch schedule, results left on the device -- twice on the same handle, against the oracle frame by frame.
    Both batch sizes must take the large-batch schedule -- octree keys in the L2-resident scratch above 64 frames (the bench's 256), per
    level in LDS when they fit up to 64 (k_octree_dyn; the worst case of a 640x480 level never fits, see
    test_small_batches_of_large_images_pick_lds_keys_per_level), two octree launches, level 0's octree started early, resize tail on the side
    stream -- and the test asserts that it ran (orbx_debug_last_schedule).  With `tail_delay` a spin kernel holds the resize tail back by 0.4 ms -- far longer than FAST on the lower
    levels -- and the upper pyramid levels are poisoned first, so a blur that does not wait for the tail reads the poison and the
    upper levels' descriptors differ (the ordering bug of round 2, src/ORBextractor.cc:1132-1138 reads every level).
    `mode` picks the stream schedule (1, the default: the resize chain on the side stream beside FAST on level 0; 0: in front of it):
    both must give the same bits, also with the delayed tail."""
    import torch
    monkeypatch.setenv("ORBX_RESIZE_BESIDE", str(mode & 1))
    distinct = [synth.make_frame(970 + i) for i in range(8)]
    oex = oracle.extractor(1000, 1.2, 8, 20, 7)
    ref = [oex.extract(im, (0, 1000)) for im in distinct]
    dev = torch.device("cuda", 0)
    d_img = torch.from_numpy(np.stack([distinct[i % 8] for i in range(B)])).to(dev)
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    try:
        cap = ex.max_keypoints
        d_kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev); d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
        d_n = torch.zeros(B, dtype=torch.int32, device=dev); d_mono = torch.zeros(B, dtype=torch.int32, device=dev); d_st = torch.zeros(B, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(2):
            if tail_delay:
                ex.debug_set_tail_delay(tail_delay)
                if rep == 1:
                    # second call: the upper levels hold the previous batch's pixels -- shift the frames by one so that stale
                    # levels belong to ANOTHER frame
                    d_img = torch.roll(d_img, 1, 0).contiguous()
            ex.extract_batch_device(d_img.data_ptr(), B, 640, 480, 640, 640 * 480, d_kps.data_ptr(), d_desc.data_ptr(), cap,
                                    d_n.data_ptr(), d_mono.data_ptr(), d_st.data_ptr(), (0, 1000), st)
            torch.cuda.synchronize()
            sched = ex.debug_last_schedule()
            assert sched & 16, "level 0 was not read in place"
            # (octree: up to 64 frames k_octree_dyn -- keys in LDS per level when they fit --, above that the scratch instantiation)
            assert sched == ((3 if B <= 64 else 2) | 4 | 8 | 16 | 32 | (64 if mode & 1 else 0)), "not the expected schedule: %d" % sched
            assert int(d_st.abs().sum().item()) == 0
            n = d_n.cpu().numpy(); mono = d_mono.cpu().numpy()
            kps = d_kps.cpu().numpy().view(pkg.KP_DTYPE).reshape(B, cap); desc = d_desc.cpu().numpy().reshape(B, cap, 32)
            shift = 1 if (tail_delay and rep == 1) else 0
            for b in range(B):
                r0, k0, d0 = ref[(b - shift) % 8]
                assert mono[b] == r0 and n[b] == len(k0)
                _assert_kps_equal(kps[b, :n[b]], k0, "frame %d" % b)
                np.testing.assert_array_equal(desc[b, :n[b]], d0)
    finally:
        ex.close()


def test_small_images_keep_octree_keys_in_lds(pkg, oracle, synth):
    """the other octree instantiation (k_octree<true, true>: both key buffers of a level in LDS, one launch) serves small batches
    of small images; asserted via the schedule getter, results against the oracle"""
    imgs = np.stack([synth.make_frame(990 + i, 256, 192) for i in range(4)])
    oex = oracle.extractor(400, 1.2, 5, 20, 7)
    ex = pkg.Extractor(400, 1.2, 5, 20, 7)
    try:
        mono, n, kps, desc = ex.extract_batch(imgs)
        assert ex.debug_last_schedule() & 3 == 1 and not ex.debug_last_schedule() & 4
        for b in range(4):
            r0, k0, d0 = oex.extract(imgs[b], (0, 1000))
            assert mono[b] == r0 and n[b] == len(k0)
            _assert_kps_equal(kps[b, :n[b]], k0, "frame %d" % b)
            np.testing.assert_array_equal(desc[b, :n[b]], d0)
    finally:
        ex.close()


@pytest.mark.parametrize("env,expect", [({}, 3), ({"ORBX_OCT_DYN_KEYS": "1500"}, 3), ({"ORBX_OCT_DYN": "0"}, 2)])
def test_small_batches_of_large_images_pick_lds_keys_per_level(pkg, oracle, synth, monkeypatch, env, expect):
    """640 x 480: a level's worst-case key buffers do not fit LDS, its actual candidates do.  Batches of up to 64 frames run k_octree_dyn,
    which counts the level's candidates and takes the keys-in-LDS body when they fit, the scratch body otherwise -- with the LDS
    allotment capped at 1500 keys the lower levels (more candidates) take the scratch body and the upper ones the LDS body inside ONE
    launch; ORBX_OCT_DYN=0 keeps the scratch instantiation.  Same bits as the oracle in every case, B = 1 and B = 5."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    imgs = np.stack([synth.make_frame(940 + i) for i in range(5)])
    oex = oracle.extractor(1000, 1.2, 8, 20, 7)
    ref = [oex.extract(im, (0, 1000)) for im in imgs]
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    try:
        for B in (1, 5):
            mono, n, kps, desc = ex.extract_batch(imgs[:B])
            assert ex.debug_last_schedule() & 3 == expect
            for b in range(B):
                r0, k0, d0 = ref[b]
                assert mono[b] == r0 and n[b] == len(k0)
                _assert_kps_equal(kps[b, :n[b]], k0, "frame %d" % b)
                np.testing.assert_array_equal(desc[b, :n[b]], d0)
    finally:
        ex.close()
