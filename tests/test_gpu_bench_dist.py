"""bench.py's N > 1 control path on one GPU: LQCD_BENCH_FORCE_DIST takes the torch.distributed (gloo) branch at world size 1 and LQCD_FORCE_PARTITION makes the
lattice a partitioned one whose halos travel through a world-size-1 RCCL communicator -- every call of the multi-GPU bench (comm_init from a broadcast id,
partitioned Dslash and CG window, the halo-phase and all-reduce diagnostics, the reductions over ranks) runs exactly as under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mask", ["8", "12"])
def test_bench_distributed_control_path_on_a_self_partitioned_lattice(mask):
    env = dict(os.environ, LQCD_BENCH_FORCE_DIST="1", LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5", "--lattice", "16,16,16,32",
                        "--dslash-reps", "50", "--no-cpu-baseline", "--no-pmc"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["value"] > 0 and d["unit"] == "iter/s" and d["scaling"] == "strong"
    assert d["roofline"]["traffic"] is None and "N > 1" in d["roofline"]["traffic_source"]
    ph = d["halo_phases_ms_max_over_ranks"]
    assert all(ph[k] is not None and ph[k] >= 0 for k in ("pack", "interior", "exterior", "total_synchronised")), ph
    assert d["allreduce_latency_us"] > 0 and d["halo_stream_mode_rank0"]["chosen"] in (0, 1, 2, 3, 4)
    assert d["halo_selfcheck"]["ok"] is True


@pytest.mark.parametrize("nproc,lattice,comm", [(2, "16,16,16,32", "peer"), (4, "16,16,16,32", "auto")])
def test_bench_with_real_processes_on_one_gpu(nproc, lattice, comm):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` as the driver launches it, with N REAL processes -- on the one GPU of the test box
    (LQCD_BENCH_SHARE_DEVICE: every rank on device 0), which the peer-mapped backend allows and RCCL does not.  The whole N > 1 line: window descriptions gathered
    over gloo, PE grid, partitioned Dslash and CG window on the HIP halo path between processes, diagnostics, max over ranks, one JSON line from rank 0.
    (The figures of a shared device are not performance numbers; that the line comes out, and what it says it ran, is the test.)"""
    env = dict(os.environ, LQCD_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("LQCD_BENCH_FORCE_DIST", "LQCD_FORCE_PARTITION"):
        env.pop(k, None)
    port = str(29660 + nproc)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc, "--master-addr", "127.0.0.1", "--master-port", port,
                        os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "40", "--warmup", "5", "--lattice", lattice, "--dslash-reps", "50",
                        "--comm", comm], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == nproc and d["steps"] == 40 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["comm_backend"] == "peer" and d["config"]["ranks_share_device_0"] is True and d["config"]["comm_note"] is None
    assert d["halo_selfcheck"]["ok"] is True and d["halo_selfcheck"]["rel_diff"] < 1e-12      # the processes' halo path reproduces the one-GPU |D b|^2 of the same global problem
    pe = d["config"]["pe_grid"]
    assert pe[0] * pe[1] * pe[2] * pe[3] == nproc and pe[0] == 1
    ph = d["halo_phases_ms_max_over_ranks"]
    assert all(ph[k] is not None and ph[k] >= 0 for k in ("pack", "interior", "total_synchronised")), ph
    assert d["allreduce_latency_us"] > 0 and d["halo_stream_mode_rank0"]["chosen"] in (0, 1, 2, 3, 4)
