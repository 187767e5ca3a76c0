"""CPU-only checks of the C-ABI library (loads, exports every declared symbol, host-only index helpers are bit-exact)
and of the PE-grid host logic, including a world_size-2 gloo run of the halo decomposition."""
import ctypes as C
import itertools
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(lq):
    L = lq.lib.lib()
    syms = lq.lib.declared_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert L.lqcd_version() >= 100


def test_no_cpu_fallback(lq):
    """Without a GPU the product fails loudly instead of computing on the host."""
    if lq.lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(lq.LQCDError) as e:
        lq.Lattice((4, 4, 4, 4))
    assert e.value.code == lq.lib.ERR_HIP


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "latticeqcd.jl_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.lower(), f"{f} mentions the oracle"


@pytest.mark.parametrize("L", [(4, 4, 4, 4), (8, 4, 6, 2), (2, 2, 2, 2)])
def test_index_maps_bit_exact(lq, L):
    """Site/link indexing must be bit-exact (BASELINE.json north_star).  C helpers vs an independent numpy statement."""
    lib = lq.lib.lib()
    Lc = lq.lib.i4(L)
    XH = L[0] // 2
    seen = set()
    for t, z, y, x in itertools.product(range(L[3]), range(L[2]), range(L[1]), range(L[0])):
        assert lib.lqcd_index_lex(Lc, x, y, z, t) == x + L[0] * (y + L[1] * (z + L[2] * t))
        p, cb = C.c_int(-1), C.c_int64(-1)
        assert lib.lqcd_index_cb(Lc, x, y, z, t, C.byref(p), C.byref(cb)) == 0
        assert p.value == (x + y + z + t) % 2
        assert cb.value == x // 2 + XH * (y + L[1] * (z + L[2] * t))
        out = lq.lib.i4([0] * 4)
        assert lib.lqcd_coords_cb(Lc, p.value, cb.value, out) == 0
        assert tuple(out) == (x, y, z, t)
        seen.add((p.value, cb.value))
    assert len(seen) == L[0] * L[1] * L[2] * L[3]        # bijection onto [2] x [Vh]
    p, cb = C.c_int(), C.c_int64()
    assert lib.lqcd_index_cb(Lc, L[0], 0, 0, 0, C.byref(p), C.byref(cb)) == lq.lib.ERR_ARG
    assert b"out of range" in lib.lqcd_last_error()


@pytest.mark.parametrize("gL,pe", [((8, 8, 8, 16), (1, 2, 2, 2)), ((32, 32, 32, 64), (1, 1, 2, 4)), ((4, 4, 4, 4), (1, 1, 1, 2)),
                                   ((8, 8, 8, 8), (2, 1, 1, 2))])
def test_decompose_matches_python(lq, gL, pe):
    lib = lq.lib.lib()
    n = int(np.prod(pe))
    origins = set()
    for rank in range(n):
        lo, org, nf, nb = (lq.lib.i4([0] * 4) for _ in range(4))
        assert lib.lqcd_decompose(lq.lib.i4(gL), lq.lib.i4(pe), rank, lo, org, nf, nb) == 0
        pl, po, pf, pb = lq.pegrid.decompose(gL, pe, rank)
        assert (tuple(lo), tuple(org), tuple(nf), tuple(nb)) == (pl, po, pf, pb)
        origins.add(tuple(org))
        for mu in range(4):   # forward of backward is identity
            lo2, org2, nf2, nb2 = (lq.lib.i4([0] * 4) for _ in range(4))
            lib.lqcd_decompose(lq.lib.i4(gL), lq.lib.i4(pe), nf[mu], lo2, org2, nf2, nb2)
            assert nb2[mu] == rank
    assert len(origins) == n
    assert lib.lqcd_decompose(lq.lib.i4(gL), lq.lib.i4(pe), n, lo, org, nf, nb) == lq.lib.ERR_ARG
    assert lib.lqcd_decompose(lq.lib.i4((6, 4, 4, 4)), lq.lib.i4((1, 1, 1, 4)), 0, lo, org, nf, nb) == lq.lib.ERR_ARG  # odd local T


def test_choose_pe_grid(lq):
    assert lq.pegrid.choose_pe_grid((32, 32, 32, 64), 1) == (1, 1, 1, 1)
    assert lq.pegrid.choose_pe_grid((32, 32, 32, 64), 2) == (1, 1, 1, 2)
    assert lq.pegrid.choose_pe_grid((32, 32, 32, 64), 4) == (1, 1, 2, 2)
    assert lq.pegrid.choose_pe_grid((32, 32, 32, 64), 8) == (1, 2, 2, 2)   # 3 distinct peers, x unpartitioned
    with pytest.raises(ValueError):
        lq.pegrid.choose_pe_grid((4, 4, 4, 4), 16)


def test_gloo_world2_domain_decomposed_dslash():
    """world_size-2 gloo run: each rank owns half of the t extent, exchanges one-slice halos of the spinor with its
    neighbour over torch.distributed (gloo) following pegrid.decompose, applies the oracle stencil on its padded
    sub-lattice and must reproduce the single-domain oracle bit-for-bit-level (<= 1e-14), including the antiperiodic
    wrap which only the rank owning the global boundary applies."""
    script = os.path.join(ROOT, "tests", "gloo_dd_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29611", script], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DD_OK rank 0" in r.stdout and "DD_OK rank 1" in r.stdout


def _build_c_smoke(out):
    csrc = os.path.join(ROOT, "latticeqcd.jl_amd", "csrc")
    cmd = ["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", out,
           "-L", csrc, "-llqcd_hip", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm"]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_c_abi_program_compiles_and_fails_loudly_without_gpu(lq, tmp_path):
    """The boundary is usable from plain C; without a GPU the program reports it instead of computing on the host."""
    exe = str(tmp_path / "c_abi_smoke")
    r = _build_c_smoke(exe)
    assert r.returncode == 0, r.stderr
    if lq.lib.device_count() == 0:
        run = subprocess.run([exe], capture_output=True, text=True)
        assert run.returncode == 2 and "no HIP device" in run.stderr


def test_bench_control_path_two_processes(lq, tmp_path):
    """The N > 1 control path of bench.py executed by TWO real processes under torch.distributed.run on a machine without GPUs:
    gloo rendezvous on 127.0.0.1, broadcast of the communicator id from rank 0, PE-grid choice and decomposition (the real library's
    host code), barriers, max-over-ranks reductions and the single JSON line of rank 0.  The GPU entry points are a CPU stand-in
    (tests/stub/lqcd_stub.c, selected through LQCD_HIP_LIB; test infrastructure) whose timings are (rank + 1) x a base value, so the
    job-wide figures must be rank 1's."""
    import json
    stub_dir = os.path.join(ROOT, "tests", "stub")
    r = subprocess.run(["make", "-s", "-C", stub_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, LQCD_HIP_LIB=os.path.join(stub_dir, "liblqcd_stub.so"), LQCD_STUB_REAL_LIB=lq.lib.SO_PATH,
               LQCD_STUB_DIR=str(tmp_path))
    env.pop("LQCD_BENCH_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    out = json.loads(lines[-1])                                     # the JSON line is the LAST line of the job's stdout
    assert sum(1 for ln in lines if ln.startswith("{")) == 1        # and only rank 0 prints one
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "strong"
    assert out["config"]["pe_grid"] == [1, 1, 1, 2] and out["config"]["local_lattice"] == [32, 32, 32, 32]
    assert abs(out["dslash_ms"] - 0.1) < 1e-12                      # max over ranks: rank 1 reported 2 x 0.05 ms
    assert abs(out["allreduce_latency_us"] - 20.0) < 1e-12 and abs(out["halo_phases_ms_max_over_ranks"]["pack"] - 0.02) < 1e-12
    assert out["halo_bytes_per_peer_and_direction"] == [0, 0, 0, 96 * 32 * 32 * 32]
    assert out["roofline"]["traffic"] is None and "cpu_baseline" not in out      # N = 1 extras stay out of the N > 1 line
    ids = [open(os.path.join(str(tmp_path), "rank%d.id" % k), "rb").read() for k in (0, 1)]
    pattern = bytes((37 * i + 11) % 256 for i in range(256))
    assert ids[0][:256] == pattern and ids[1][:256] == pattern     # both ranks initialised their communicators with rank 0's id
    assert b"pe=1,1,1,2 nranks=2 device=0" in ids[0] and b"device=1" in ids[1]   # one device per local rank
