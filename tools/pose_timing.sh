cp orb_slam3-1_amd/liborbslam3_hip.so /tmp/lib_full.so
trap 'cp /tmp/lib_full.so orb_slam3-1_amd/liborbslam3_hip.so' EXIT
(cd orb_slam3-1_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -I../../include -DPOSE_TIMING -c -o /tmp/p_t.o pose_solver.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbslam3_hip.so dbow_vocab.o edge_packet.o lba_solver.o orbm_matcher.o orbx_extractor.o /tmp/p_t.o) || exit 1
python - <<'P'
import os, importlib, sys, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("orb_slam3-1_amd"); synth = importlib.import_module("orb_slam3-1_amd.synth")
s = pkg.PoseSolver()
for n in (300, 750):
    w = synth.make_pose_problem(0, n=n, outlier_frac=0.07)
    s.optimize(w); r = s.optimize(w)
    print("n", n, "kernel ms", s.last_kernel_ms(), "iters", r["iterations"], "trials", r["trials"])
    print("cycles: build %d  reduceH %d  serial %d  trialpass %d  trialreduce+update %d  classify %d" % (r["chi2"][0], r["chi2"][1], r["chi2"][2], r["chi2"][3], r["t"][0], r["t"][1]))
P
