#!/bin/bash
# k_fast_strips cut after a prefix of its stages (libraries built with -DORBX_FAST_CUT=0/1/2 under build_variants/): the stage
# time of each variant tells what the stages cost in the full kernel.  Results of the cut variants are wrong by construction.
cp orb_slam3-1_amd/liborbslam3_hip.so /tmp/lib_full.so
for v in 0 1 2; do
  cp build_variants/lib_cut$v.so orb_slam3-1_amd/liborbslam3_hip.so
  timeout -k 10 120 python - <<PY
import importlib, numpy as np, torch, sys
sys.path.insert(0, ".")
pkg = importlib.import_module("orb_slam3-1_amd"); synth = importlib.import_module("orb_slam3-1_amd.synth")
ex = pkg.Extractor(); ex.profile_enable(True)
imgs = np.stack([synth.make_frame(i) for i in range(16)] * 16)
try:
    ex.extract_batch(imgs)
except Exception as e:
    pass
try:
    ex.extract_batch(imgs)
except Exception as e:
    pass
print("cut $v:", ex.profile_read())
PY
done
cp /tmp/lib_full.so orb_slam3-1_amd/liborbslam3_hip.so
