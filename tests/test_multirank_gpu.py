"""N > 1 path with the HIP kernels: two and four ranks sharing the box's one GPU (gloo rendezvous,
host-staged all-to-all) must reproduce the single-domain forces, energy and virial."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_hip_forces_match_single_domain(world):
    env = dict(os.environ, MTP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29710 + world),
                        os.path.join(ROOT, "scripts", "check_multirank.py")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "multirank check" in r.stdout


@pytest.mark.gpu
def test_sharded_grades_match_single_domain():
    """compile_grades over ranks: all-reduce SUM of the candidate vector (configuration mode), all-reduce MAX of
    the per-rank maxima (neighbourhood mode)."""
    env = dict(os.environ, MTP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29719",
                        os.path.join(ROOT, "scripts", "check_multirank_grades.py")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("multirank grades") == 2
