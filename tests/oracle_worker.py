"""Child process of bench.py's all-cores CPU baseline: runs the CPU oracle (extract + SearchByBoW per frame) for a fixed
time on one core and prints "<frames> <seconds>".  Test infrastructure (oracle side), never imported by the product."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_api import Oracle  # noqa: E402


def main():
    budget = float(sys.argv[1])
    lib = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else None
    synth = importlib.import_module("orb_slam3-1_amd.synth")
    o = Oracle(lib)
    ex = o.extractor(1000, 1.2, 8, 20, 7)
    imgs = synth.make_frames(4, seed0=7)
    ms = synth.make_match_set(50)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        ex.extract(imgs[n % 4], (0, 1000))
        o.search_by_bow(ms["dKF"], ms["validKF"], ms["angKF"], ms["fvKF"], ms["dF"], ms["angF"], ms["fvF"], 0.7, True)
        n += 1
    print(n, time.perf_counter() - t0, flush=True)


if __name__ == "__main__":
    main()
