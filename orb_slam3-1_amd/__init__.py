"""orb_slam3-1_amd -- MI355X (gfx950) implementation of ORB-SLAM3's per-frame hot path.

The product is the C-ABI shared library ``liborbslam3_hip.so`` (include/orbslam3_hip.h), built from the
hand-written HIP sources in ``csrc/``.  This package is the thin Python mirror of that ABI used by the
parity tests, bench.py and the multi-GPU harness.  It never computes anything itself and has no CPU
fallback: if the library is missing, loading fails loudly.

The package directory name is not a valid Python identifier; import it with
``importlib.import_module("orb_slam3-1_amd")``.
"""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "liborbslam3_hip.so")


def build(verbose=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU) into liborbslam3_hip.so."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", os.path.join(PKG_DIR, "csrc")], stdout=out)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build finished but %s is missing" % LIB_PATH)
    return LIB_PATH


from .capi import (Extractor, Matcher, LbaSolver, LbaShard, LbaBatch, PoseSolver, Vocabulary, DeviceBowPlan, PacketCodec, InertialSolver, LibaBatch, IMU_DTYPE, lib, KP_DTYPE, OrbxError, hamming,  # noqa: E402,F401
                   device_count)
