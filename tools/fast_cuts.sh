#!/bin/bash
# k_fast_strips cut after a prefix of its stages (-DORBX_FAST_CUT=0/1/2/3: after the tile load, after stage 1, after stage 2, after the whole iniThFAST pass): the
# stage time of each variant tells what the stages cost in the full kernel.  Results of the cut variants are wrong by construction.
# usage (GPU box, repo root, after `make -C orb_slam3-1_amd/csrc`):  bash tools/fast_cuts.sh
cp orb_slam3-1_amd/liborbslam3_hip.so /tmp/lib_full.so
# whatever happens (a failed build, a timeout, an interrupt): the full library comes back
trap 'cp /tmp/lib_full.so orb_slam3-1_amd/liborbslam3_hip.so' EXIT
for v in 0 1 2 3; do
  (cd orb_slam3-1_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DORBX_FAST_CUT=$v -c -o /tmp/ex_cut.o orbx_extractor.hip &&
   /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbslam3_hip.so dbow_vocab.o edge_packet.o lba_solver.o orbm_matcher.o /tmp/ex_cut.o pose_solver.o) || exit 1
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
