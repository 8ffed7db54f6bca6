"""The POD-level part of include/orbslam3_shim.hpp compiles and links against the C-ABI library (the reference-typed
part needs OpenCV/Eigen/Sophus, absent from this image, and is guarded by ORBSLAM3_HIP_WITH_REFERENCE)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_compiles_and_links(tmp_path, pkg):
    src = tmp_path / "t.cpp"
    src.write_text('''
#include "orbslam3_shim.hpp"
#include <cstdio>
#include <map>
int main() {
    std::map<unsigned, std::vector<unsigned>> fv; fv[3] = {1, 2}; fv[9] = {0};
    orbslam3_hip::FlatFeatVec<std::map<unsigned, std::vector<unsigned>>> f(fv);
    if (f.view.n_nodes != 2 || f.off[2] != 3) return 2;
    unsigned char a[32] = {0}, b[32] = {0}; b[5] = 0x0F;
    if (orbm_hamming(a, b) != 4) return 3;
    try { orbslam3_hip::Extractor ex(1000, 1.2f, 8, 20, 7); std::printf("device ok\\n"); }
    catch (const orbslam3_hip::Error& e) { std::printf("no device: %d\\n", e.code); if (e.code != ORBX_ERR_NO_DEVICE) return 4; }
    try {
        orbslam3_hip::EdgePacketCodec codec;
        OrbxKeyPoint kp[2] = {}; kp[0].x = 258.75f; kp[0].y = 1.9f; kp[1].x = 3.f; kp[1].y = 479.f;
        unsigned char d[64]; for (int i = 0; i < 64; i++) d[i] = (unsigned char)i;
        unsigned char head[2];
        std::vector<uint8_t> pkt = codec.pack(7, 99, kp, d, 2, nullptr, 0, head);
        if (pkt.size() != 88 || pkt[13] != 2 || pkt[16] != 1 || pkt[17] != 2 || pkt[19] != 1 || head[1] != 88) return 5;
        int32_t id; int64_t ts; std::vector<OrbxKeyPoint> k; std::vector<uint8_t> dd; std::vector<OrbeImuSample> im;
        codec.unpack(pkt.data(), (int)pkt.size(), id, ts, k, dd, im);
        if (id != 7 || ts != 99 || k.size() != 2 || k[0].x != 258.f || k[1].y != 479.f || dd[63] != 63 || !im.empty()) return 6;
    } catch (const orbslam3_hip::Error& e) { if (e.code != ORBX_ERR_NO_DEVICE) return 7; }
    return 0;
}
''')
    exe = tmp_path / "t"
    libdir = os.path.join(ROOT, "orb_slam3-1_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lorbslam3_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
