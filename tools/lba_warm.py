#!/usr/bin/env python3
"""Does LocalBA (a chain of tiny serial kernels) run at idle shader clocks?  Time solves alone and with a concurrent
background load on another stream that keeps the chip busy."""
import importlib
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
w = synth.make_ba_window(0)
s = pkg.LbaSolver()
s.solve(w, 10)


def run(n):
    t0 = time.perf_counter(); it = 0
    for _ in range(n):
        it += s.solve(w, 10)["stats"]["iterations"]
    dt = time.perf_counter() - t0
    return it / dt, 1e3 * dt / n


print("alone: %.0f it/s, %.3f ms per solve" % run(100))
stop = False
side = torch.cuda.Stream()
a = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 2048, device="cuda")


def spin():
    with torch.cuda.stream(side):
        while not stop:
            for _ in range(20):
                torch.mm(a, a.t())
            side.synchronize()


th = threading.Thread(target=spin); th.start()
time.sleep(0.5)
print("with a background matmul load: %.0f it/s, %.3f ms per solve" % run(100))
stop = True; th.join()
print("alone again: %.0f it/s, %.3f ms per solve" % run(100))
s.close()
