"""Two ranks on the one GPU of the test box (gloo rendezvous, device-resident reduce buffers): the landmark-sharded global
BA -- one all-reduce of [S | b_schur | b_p | diag Hpp] per Levenberg trial -- must walk the same LM path as the
single-rank HIP solver.  Guards the hand-off between the shard's stream and the collective's stream."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, port):
    env = dict(os.environ, ORBX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), os.path.join(ROOT, "tools", "gba_rehearsal.py")],
                          capture_output=True, text=True, timeout=600, env=env)


def _check_c_abi_lines(r, n_expected):
    """lba_shard_optimize (the C-ABI LM driver with an all-reduce callback): same LM path as the single-rank solver, same
    estimates up to the summation order of the two partial systems, and the exchange volume the design states (one reduce
    buffer per trial + one for the lambda initialisation)"""
    lines = [l for l in r.stdout.splitlines() if l.startswith("c_abi")]
    assert len(lines) == n_expected, r.stdout
    for l in lines:
        m = re.search(r"optimize: it (\d+) trials (\d+) chi2 (\S+) -> (\S+) \| single: it (\d+) trials (\d+) chi2 (\S+) -> (\S+) \| max diff q (\S+) t (\S+) points (\S+) "
                      r"\(update (\S+)\) \| allreduce calls (\d+) doubles (\d+) reduce_len (\d+)", l)
        assert m, l
        assert (m.group(1), m.group(2)) == (m.group(5), m.group(6)), l
        c1, s1 = float(m.group(4)), float(m.group(8))
        assert abs(c1 - s1) <= 1e-6 * s1, l
        upd = float(m.group(12))
        assert max(float(m.group(9)), float(m.group(10)), float(m.group(11))) <= 1e-4 * upd, l
        trials, doubles, rl = int(m.group(2)), int(m.group(14)), int(m.group(15))
        assert (trials + 1) * rl <= doubles <= (trials + 1) * rl + 64 * (trials + 8), l


def test_sharded_gba_at_config5_size_two_ranks():
    """SURVEY.md 8(d) item 5 at its stated size: 500 poses / 20 000 points / 200 000 edges, landmarks over two ranks on the one
    GPU of the box, a 69 MB reduce buffer per Levenberg trial; both the python driver and lba_shard_optimize against the
    single-rank solver (which tests/test_lba_gpu.py::test_global_ba_at_config5_size pins to the oracle at this size)"""
    r = _run({"GBA_FULL_SIZE": "1"}, 29549)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("n_opt")]
    assert len(lines) == 1 and lines[0].startswith("n_opt 490"), r.stdout
    m = re.search(r"sharded: it (\d+) trials (\d+) chi2 (\S+) -> (\S+) \| single: it (\d+) trials (\d+) chi2 (\S+) -> (\S+)", lines[0])
    assert m and (m.group(1), m.group(2)) == (m.group(5), m.group(6)), lines[0]
    assert abs(float(m.group(4)) - float(m.group(8))) <= 1e-6 * float(m.group(8))
    _check_c_abi_lines(r, 1)


def test_sharded_gba_two_ranks_matches_single_solver():
    env = dict(os.environ, ORBX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29547", os.path.join(ROOT, "tools", "gba_rehearsal.py")],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("n_opt")]
    assert len(lines) == 2, r.stdout
    for l in lines:
        m = re.search(r"sharded: it (\d+) trials (\d+) chi2 (\S+) -> (\S+) \| single: it (\d+) trials (\d+) chi2 (\S+) -> (\S+)", l)
        assert m, l
        assert (m.group(1), m.group(2)) == (m.group(5), m.group(6)), l
        c0, c1, s0, s1 = float(m.group(3)), float(m.group(4)), float(m.group(7)), float(m.group(8))
        assert abs(c0 - s0) <= 1e-6 * s0 and abs(c1 - s1) <= 1e-6 * s1 and c1 < 0.5 * c0, l
    _check_c_abi_lines(r, 2)
