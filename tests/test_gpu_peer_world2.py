"""The first REAL world-size > 1 runs of the HIP halo path (task 1): N processes that share the one GPU of the test box and
communicate through the peer-mapped backend (csrc/comm.hip) -- hipIpc-mapped windows, pack kernels that store into the neighbour's ghost
buffer, one-wave flag kernels, slot reductions.  RCCL cannot do this (it refuses two ranks on one device).  Checker: the CPU oracle on the
global lattice (tests/peer_world_worker.py).  PE grids of SURVEY.md 8(e), scaled down: (1,1,1,2), (1,1,2,1), (1,2,1,1); Wilson, staggered,
Wilson-clover; D, D^+ <= 1e-13; CG solution <= 1e-9, iteration count +-1; every halo schedule incl. the tuner."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_port = [29640]


def run_world(n, lattice, pe, kinds="Wilson,Staggered,WilsonClover", schedules="3,4,0,1,2,-1", timeout=420, extra_env=None):
    _port[0] += 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port[0]), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               PEER_TEST_LATTICE=",".join(map(str, lattice)), PEER_TEST_PE=",".join(map(str, pe)), PEER_TEST_KINDS=kinds, PEER_TEST_SCHEDULES=schedules)
    env.pop("LQCD_FORCE_PARTITION", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port[0]), os.path.join(ROOT, "tests", "peer_world_worker.py")],
                       capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    for k in range(n):
        assert f"PEER_WORLD_OK rank {k}" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.fixture(scope="module")
def gpu():
    import latticeqcd_jl_amd as lq
    if lq.lib.device_count() < 1:
        pytest.skip("no HIP device")
    return lq


@pytest.mark.parametrize("pe", [(1, 1, 1, 2), (1, 1, 2, 1), (1, 2, 1, 1)])
def test_two_processes_one_gpu_equal_the_oracle(gpu, orc, pe):
    run_world(2, (8, 8, 8, 16), pe)


def test_two_processes_x_partitioned(gpu, orc):
    """x (the contiguous axis) partitioned: the per-lane ghost selects of the folded twins."""
    run_world(2, (16, 4, 4, 8), (2, 1, 1, 1), kinds="Wilson,Staggered", schedules="3,4,0")


@pytest.mark.parametrize("pe", [(1, 1, 2, 2), (1, 1, 1, 4)])
def test_four_processes_one_gpu(gpu, orc, pe):
    """(1,1,2,2): two partitioned directions, four ranks.  (1,1,1,4): a ring of four in t -- the forward and the backward neighbour of a rank are DIFFERENT ranks
    (two windows per direction), the one-directional mailbox exchanges go round a ring (their acknowledgements matter), only ranks 0 and 3 own the global boundary."""
    run_world(4, (8, 8, 8, 16), pe, kinds="Wilson,Staggered,WilsonClover", schedules="3,4,0,-1", timeout=600)


def test_two_processes_coarse_grained_window(gpu, orc):
    """The window as plain (coarse-grained) device memory: a one-device experiment setting, same results."""
    run_world(2, (8, 8, 8, 16), (1, 1, 1, 2), kinds="Wilson", schedules="3,0", extra_env={"LQCD_PEER_FINEGRAINED": "0"})


def test_a_dead_rank_is_reported_not_waited_for(gpu, orc):
    """Rank 1 exits after the bootstrap.  Rank 0's next exchange waits for a flag that never rises: the one-wave wait gives up after peer_timeout_ms, the call returns
    LQCD_ERR_COMM (no hang, no result) and every later call fails at once."""
    _port[0] += 1
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port[0]), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               PEER_TEST_LATTICE="8,8,8,16", PEER_TEST_PE="1,1,1,2", PEER_TEST_DEAD_RANK="1")
    env.pop("LQCD_FORCE_PARTITION", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port[0]), os.path.join(ROOT, "tests", "peer_world_worker.py")],
                       capture_output=True, text=True, env=env, timeout=180, cwd=ROOT)
    assert "PEER_DEAD_OK rank 0" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
