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
for n_opt, n_pts in ((20, 500), (190, 8000)):
    w = synth.make_ba_window(3, n_opt=n_opt, n_fixed=10, n_points=n_pts, obs_per_point=10)
    loc, _, _ = dmod.partition_landmarks(w, rank, world)
    sh = pkg.LbaShard(loc, device=0)
    ad = dmod.HipShard(sh, torch, dev)
    gs = dmod.sharded_bundle_adjustment(ad, ad.tensor, dmod.TorchDist(dist, dev), max_iters=5)
    if rank == 0:
        s = pkg.LbaSolver()
        r = s.solve(w, 5)
        s.close()
        print("n_opt %d sharded: it %d trials %d chi2 %.8g -> %.8g | single: it %d trials %d chi2 %.8g -> %.8g" % (
            n_opt, gs["iterations"], gs["trials"], gs["chi2_initial"], gs["chi2_final"],
            r["stats"]["iterations"], r["stats"]["trials"], r["stats"]["chi2_initial"], r["stats"]["chi2_final"]), flush=True)
    sh.close()
    dist.barrier()
dist.destroy_process_group()
