"""N>1 path on CPU: world_size-2 gloo run of the landmark-sharded bundle adjustment driver
(orb_slam3-1_amd/distributed.py) over the oracle's shard; must reproduce the single-rank oracle solve.
Frame sharding has no collective; its partition helper is checked directly."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    d = importlib.import_module("orb_slam3-1_amd.distributed")
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [d.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_partition_landmarks(synth):
    d = importlib.import_module("orb_slam3-1_amd.distributed")
    w = synth.make_ba_window(2, n_opt=6, n_fixed=2, n_points=101, obs_per_point=4)
    seen = []
    for r in range(3):
        loc, (lo, hi), sel = d.partition_landmarks(w, r, 3)
        assert len(loc["points"]) == hi - lo and loc["edge_point"].max() < hi - lo
        np.testing.assert_array_equal(loc["points"], w["points"][lo:hi])
        seen.append(sel)
    allsel = np.sort(np.concatenate(seen))
    np.testing.assert_array_equal(allsel, np.arange(len(w["edge_point"])))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_api import Oracle, OracleShard
    synth = importlib.import_module("orb_slam3-1_amd.synth")
    d = importlib.import_module("orb_slam3-1_amd.distributed")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = synth.make_ba_window(21, n_opt=8, n_fixed=3, n_points=240, obs_per_point=5, stereo_frac=0.2)
        loc, (lo, hi), sel = d.partition_landmarks(w, rank, world)
        sh = OracleShard(Oracle(), loc)
        t = torch.from_numpy(sh.array)            # aliases the shard's reduce buffer: all-reduced in place
        stats = d.sharded_bundle_adjustment(sh, t, d.TorchDist(dist, "cpu"), max_iters=10)
        out = sh.download()
        q.put((rank, stats, lo, hi, out["points"], out["pose_q"], out["pose_t"], sel, out["chi2"]))
    finally:
        dist.destroy_process_group()


def test_sharded_ba_world2_gloo(oracle, synth):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    w = synth.make_ba_window(21, n_opt=8, n_fixed=3, n_points=240, obs_per_point=5, stereo_frac=0.2)
    ref = oracle.lba_solve(w, 10)
    s0, s1 = res[0][1], res[1][1]
    assert s0["iterations"] == s1["iterations"] == ref["stats"]["iterations"]
    assert s0["trials"] == s1["trials"] == ref["stats"]["trials"]
    assert s0["stop_reason"] == ref["stats"]["stop_reason"]
    np.testing.assert_allclose(s0["chi2_final"], ref["stats"]["chi2_final"], rtol=1e-9)
    np.testing.assert_allclose(res[0][5], res[1][5], rtol=0, atol=1e-13)        # replicated poses agree across ranks
    np.testing.assert_allclose(res[0][5], ref["pose_q"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(res[0][6], ref["pose_t"], rtol=0, atol=1e-9)
    chi2 = np.zeros(len(w["edge_point"]))
    for r in res:
        np.testing.assert_allclose(r[4], ref["points"][r[2]:r[3]], rtol=0, atol=1e-9)
        chi2[r[7]] = r[8]
    np.testing.assert_allclose(chi2, ref["chi2"], rtol=1e-7, atol=1e-9)


def test_sharded_driver_world1_equals_oracle(oracle, synth):
    from oracle_api import OracleShard
    d = importlib.import_module("orb_slam3-1_amd.distributed")
    w = synth.make_ba_window(22, n_opt=6, n_fixed=2, n_points=120, obs_per_point=5)
    sh = OracleShard(oracle, w)
    stats = d.sharded_bundle_adjustment(sh, None, None, max_iters=10)
    ref = oracle.lba_solve(w, 10)
    assert (stats["iterations"], stats["trials"], stats["stop_reason"]) == (ref["stats"]["iterations"], ref["stats"]["trials"], ref["stats"]["stop_reason"])
    np.testing.assert_allclose(sh.download()["points"], ref["points"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(stats["lambda_"], ref["stats"]["lambda_"], rtol=1e-9)
