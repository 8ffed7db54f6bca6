import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("orb_slam3-1_amd"); synth = importlib.import_module("orb_slam3-1_amd.synth")
dev = torch.device("cuda", 0)
W, H = 320, 200
host = np.stack([synth.make_frame(i, W, H) for i in range(4)])
for B in (16, 1, 4, 1):
    d = torch.from_numpy(np.concatenate([host] * 4)[:B].copy()).to(dev)
    ex = pkg.Extractor(1000, 1.2, 8, 20, 7)
    cap = ex.max_keypoints
    kps = torch.zeros(B * cap * 28, dtype=torch.uint8, device=dev); desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    n = torch.zeros(B, dtype=torch.int32, device=dev); mono = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: ex.extract_batch_device(d.data_ptr(), B, W, H, W, W * H, kps.data_ptr(), desc.data_ptr(), cap, n.data_ptr(), mono.data_ptr(), st.data_ptr(), (0, 1000), s)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ex.profile_enable(True); run(); torch.cuda.synchronize(); prof = ex.profile_read(); ex.profile_enable(False)
    print("B=%3d sched %d n=%d  %.3f ms per call  octree %.3f" % (B, ex.debug_last_schedule(), int(n[0]), ms, prof["octree"]))
    ex.close()
