#!/usr/bin/env python3
"""Two (or more) ranks on ONE GPU, gloo backend: the landmark-sharded global BA driver against the single-rank HIP solver.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/gba_rehearsal.py"""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("orb_slam3-1_amd")
synth = importlib.import_module("orb_slam3-1_amd.synth")
dmod = importlib.import_module("orb_slam3-1_amd.distributed")
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(os.environ.get("ORBX_BENCH_BACKEND", "gloo"))
sizes = ((20, 500), (190, 8000))
if os.environ.get("GBA_FULL_SIZE"):          # 500 poses / 20 k points / 200 k edges (69 MB reduce buffer)
    sizes = ((490, 20000),)
for n_opt, n_pts in sizes:
    w = synth.make_ba_window(3, n_opt=n_opt, n_fixed=10, n_points=n_pts, obs_per_point=10)
    loc, (lo, hi), _ = dmod.partition_landmarks(w, rank, world)
    # (a) the python LM driver with torch.distributed collectives
    sh = pkg.LbaShard(loc, device=0)
    ad = dmod.HipShard(sh, torch, dev)
    gs = dmod.sharded_bundle_adjustment(ad, ad.tensor, dmod.TorchDist(dist, dev), max_iters=5)
    sh.close()
    # (b) the C-ABI driver lba_shard_optimize with an all-reduce callback (what a C++ host calls with ncclAllReduce)
    sh2 = pkg.LbaShard(loc, device=0)
    cb = dmod.host_staged_allreduce(dist, torch)
    cs = sh2.optimize(cb, world, max_iters=5)
    out2 = sh2.download()
    sh2.close()
    if rank == 0:
        s = pkg.LbaSolver()
        r = s.solve(w, 5)
        s.close()
        print("n_opt %d sharded: it %d trials %d chi2 %.8g -> %.8g | single: it %d trials %d chi2 %.8g -> %.8g" % (
            n_opt, gs["iterations"], gs["trials"], gs["chi2_initial"], gs["chi2_final"],
            r["stats"]["iterations"], r["stats"]["trials"], r["stats"]["chi2_initial"], r["stats"]["chi2_final"]), flush=True)
        import numpy as np
        dq = float(np.abs(out2["pose_q"] - r["pose_q"]).max()); dt = float(np.abs(out2["pose_t"] - r["pose_t"]).max())
        dp = float(np.abs(out2["points"] - r["points"][lo:hi]).max())
        upd = float(np.abs(r["points"] - w["points"]).max())
        print("c_abi n_opt %d optimize: it %d trials %d chi2 %.8g -> %.8g | single: it %d trials %d chi2 %.8g -> %.8g | max diff q %.3g t %.3g points %.3g (update %.3g) | allreduce calls %d doubles %d reduce_len %d" % (
            n_opt, cs["iterations"], cs["trials"], cs["chi2_initial"], cs["chi2_final"],
            r["stats"]["iterations"], r["stats"]["trials"], r["stats"]["chi2_initial"], r["stats"]["chi2_final"], dq, dt, dp, upd,
            cb.calls["n"], cb.calls["doubles"], 36 * n_opt * n_opt + 18 * n_opt), flush=True)
    dist.barrier()
dist.destroy_process_group()
