"""The self-partitioned tests of the RCCL path (LQCD_FORCE_PARTITION + a one-rank communicator: pack, exchange, folded / exterior launches, fused CG tails,
mixed precision, clover, staple and fermion force, stout, HMC trajectories) once more on the PEER-MAPPED backend: LQCD_SELFCOMM_BACKEND=peer makes
Lattice.comm_init map the rank's window onto itself (csrc/comm.hip), everything else is the same test body checked against the same oracle values."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SELECTION = [
    "tests/test_gpu_halo_fuse.py",
    "tests/test_gpu_parity.py::test_rccl_self_partition_general_r_solvers",
    "tests/test_gpu_parity.py::test_rccl_self_partition_dslash_and_cg",
    "tests/test_gpu_clover.py::test_rccl_self_partition_clover",
    "tests/test_gpu_md_partitioned.py",
    "tests/test_gpu_mixed.py",
    "tests/test_gpu_pipe.py",
    "tests/test_gpu_stout.py",
    "tests/test_gpu_hmc_partitioned.py",
]


def test_self_partitioned_suite_on_the_peer_backend():
    import latticeqcd_jl_amd as lq
    if lq.lib.device_count() < 1:
        pytest.skip("no HIP device")
    env = dict(os.environ, LQCD_SELFCOMM_BACKEND="peer", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-k", "self or rccl", "-p", "no:cacheprovider"] + SELECTION,
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = r.stdout[-4000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
