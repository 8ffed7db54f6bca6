import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pkg = importlib.import_module("orb_slam3-1_amd")
synth = pkg.synth if hasattr(pkg, "synth") else importlib.import_module("orb_slam3-1_amd.synth")
from oracle_api import Oracle
print("devices", pkg.device_count())
ex = pkg.Extractor()
img = synth.make_frame(0)
mono, kps, desc = ex(img)
print("gpu n", len(kps), "mono", mono, kps[:2], desc[0, :8])
o = Oracle().extractor()
r, okps, odesc = o.extract(img)
print("cpu n", len(okps), "equal", np.array_equal(kps, okps), np.array_equal(desc, odesc))
for B in (1, 8, 64):
    imgs = synth.make_frames(min(B, 8))
    imgs = np.concatenate([imgs] * (B // len(imgs)))[:B]
    ex.extract_batch(imgs)
    t = time.time()
    for _ in range(3):
        m, n, k, d = ex.extract_batch(imgs)
    dt = (time.time() - t) / 3
    print("batch", B, "ms/batch %.2f" % (dt * 1e3), "frames/s %.0f" % (B / dt), "n[0]", n[0])
