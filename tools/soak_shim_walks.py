#!/usr/bin/env python3
"""Soak of the reference-signature adapters' graph walks (CPU only: no device needed): include/orbslam3_shim.hpp compiled against the
stand-in ORB-SLAM3 types of tests/stubs/, its LocalBundleAdjustment walk (lists, vertex order, fixed flags, edge order, out-parameters,
src/Optimizer.cc:1118-1404) and its all-key-frame BundleAdjustment walk (:60-277) run on MANY seeded toy maps -- permuted ids, shuffled
covisibility lists, bad / foreign key frames and points, stereo observations -- against the independent Python restatements of
tests/test_shim_reference_typed.py.  usage: soak_shim_walks.py [n_seeds]"""
import os
import subprocess
import sys
import tempfile
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_shim_reference_typed as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tmp = tempfile.mkdtemp(prefix="shim_soak_")
exe = os.path.join(tmp, "shim_toy_map")
subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", T.STUBS, "-I", T.INC, os.path.join(T.STUBS, "shim_toy_map.cpp"), "-o", exe, "-L", T.LIBDIR, "-lorbslam3_hip",
                       "-Wl,-rpath," + T.LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"])
bad = 0; runs = 0
path = os.path.join(tmp, "map.txt")
for seed in range(100, 100 + N):
    for kind in ("graph", "gba_graph"):
        runs += 1
        try:
            m = T._toy_map(seed, bool(seed & 1)) if kind == "graph" else T._gba_map(seed)
            T._write_map(path, m)
            r = subprocess.run([exe, kind, path], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
            d = T._parse(r.stdout)
            if kind == "graph":
                T._check_graph(d, m)
            else:
                T._check_gba_graph(d, m)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL %s seed %d: %s | %s" % (kind, seed, type(e).__name__, " / ".join(traceback.format_exc().strip().splitlines()[-3:])[:500]), flush=True)
print("shim graph-walk soak: %d toy maps x 2 walks = %d runs, %d failures" % (N, runs, bad))
sys.exit(1 if bad else 0)
