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
