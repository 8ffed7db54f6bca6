cp orb_slam3-1_amd/liborbslam3_hip.so /tmp/lib_full.so
trap 'cp /tmp/lib_full.so orb_slam3-1_amd/liborbslam3_hip.so' EXIT
(cd orb_slam3-1_amd/csrc && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -DORBM_PROJ_TIMING -c -o /tmp/m_t.o orbm_matcher.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../liborbslam3_hip.so dbow_vocab.o edge_packet.o lba_solver.o /tmp/m_t.o orbx_extractor.o pose_solver.o) || exit 1
timeout -k 10 120 python tools/proj_timing.py $1 < /dev/null
