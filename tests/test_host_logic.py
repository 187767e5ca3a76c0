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
    assert lq.pegrid.choose_pe_grid((32, 32, 32, 64), 8) == (1, 1, 2, 4)   # 3 distinct peers (z, t-1, t+1), x unpartitioned; y too (round 6: whole-chunk faces)
    assert lq.pegrid.choose_pe_grid((48, 48, 48, 96), 8) == (1, 1, 2, 4)
    assert lq.pegrid.choose_pe_grid((32, 32, 32, 64), 16) == (1, 1, 4, 4) and lq.pegrid.choose_pe_grid((32, 32, 32, 64), 64) == (1, 2, 4, 8)   # y only once z, t are down to 8
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
    pattern = bytes((37 * i + 11) % 256 for i in range(256))
    # three bootstraps: the default (peer-mapped windows: the 256-byte descriptions gathered in rank order), --comm rccl (rank 0's id broadcast),
    # and the default on a machine where a rank cannot map its peers (every rank falls back to RCCL together, and the line says so)
    for k, (flags, extra, want) in enumerate(((["--comm", "rccl"], {}, "rccl"), ([], {}, "peer"), ([], {"LQCD_STUB_PEER_FAILS": "1"}, "rccl"),
                                              ([], {"LQCD_STUB_BAD_NORM": "1"}, "rccl"))):      # ... and where the peer path gives a wrong |D b|^2: fallback after the self-check
        d = tmp_path / ("run%d" % k)
        d.mkdir()
        env = dict(os.environ, LQCD_HIP_LIB=os.path.join(stub_dir, "liblqcd_stub.so"), LQCD_STUB_REAL_LIB=lq.lib.SO_PATH, LQCD_STUB_DIR=str(d), **extra)
        env.pop("LQCD_BENCH_FORCE_DIST", None)
        port = str(29617 + k)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"] + flags,
                           capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        out = json.loads(lines[-1])                                     # the JSON line is the LAST line of the job's stdout
        assert sum(1 for ln in lines if ln.startswith("{")) == 1        # and only rank 0 prints one
        assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == "strong"
        assert out["config"]["pe_grid"] == [1, 1, 1, 2] and out["config"]["local_lattice"] == [32, 32, 32, 32]
        assert out["config"]["comm_backend"] == want and out["config"]["comm_requested"] == (flags[1] if flags else "auto")
        assert (out["config"]["comm_note"] is not None) == bool(extra)  # the fallback is reported, nothing else is
        assert abs(out["dslash_ms"] - 0.1) < 1e-12                      # max over ranks: rank 1 reported 2 x 0.05 ms
        assert abs(out["allreduce_latency_us"] - 20.0) < 1e-12 and abs(out["halo_phases_ms_max_over_ranks"]["pack"] - 0.02) < 1e-12
        assert out["halo_bytes_per_peer_and_direction"] == [0, 0, 0, 96 * 32 * 32 * 32]
        assert out["roofline"]["traffic"] is None and "cpu_baseline" not in out      # N = 1 extras stay out of the N > 1 line
        assert out["halo_selfcheck"]["ok"] is True and out["halo_selfcheck"]["rel_diff"] == 0.0
        if want == "rccl":
            ids = [open(os.path.join(str(d), "rank%d.id" % q), "rb").read() for q in (0, 1)]
            assert ids[0][:256] == pattern and ids[1][:256] == pattern     # both ranks initialised their communicators with rank 0's id
            assert b"pe=1,1,1,2 nranks=2 device=0" in ids[0] and b"device=1" in ids[1]   # one device per local rank
        else:
            pr = [open(os.path.join(str(d), "rank%d.peer" % q), "rb").read() for q in (0, 1)]      # written once the gathered blobs were in rank order
            assert b"pe=1,1,1,2 nranks=2 device=0" in pr[0] and b"device=1" in pr[1]
            assert not os.path.exists(os.path.join(str(d), "rank0.id"))


# ------------------------------------------------------------------ the Julia binding against the header (no Julia in the image)
def _split_top(s):
    """split at top-level commas (parentheses, brackets and braces nest)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _c_prototypes():
    import re
    txt = open(os.path.join(ROOT, "include", "lqcd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|const char\s*\*|void|double)\s+(lqcd_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, params = m.group(1).replace(" ", ""), m.group(2), " ".join(m.group(3).split())
        protos[name] = (ret, [] if params in ("", "void") else _split_top(params))
    return protos


def _julia_ccalls():
    """every ccall((:sym, LIB), Ret, (types...), args...) of julia/LatticeQCDHIP.jl as (line, sym, ret, [types], [args])"""
    src = open(os.path.join(ROOT, "julia", "LatticeQCDHIP.jl")).read()
    calls, pos = [], 0
    while True:
        i = src.find("ccall(", pos)
        if i < 0:
            break
        j, depth = i + len("ccall("), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        body = src[i + len("ccall("):j - 1]
        pos = j
        line = src.count("\n", 0, i) + 1
        if src.rfind("\n", 0, i) < src.rfind("#", 0, i) and "\n" not in src[src.rfind("#", 0, i):i]:
            continue                                   # inside a comment
        parts = _split_top(body)
        sym = parts[0].strip("()").split(",")[0].strip().lstrip(":")
        types = _split_top(parts[2].strip()[1:-1]) if parts[2].strip() != "()" else []
        calls.append((line, sym, parts[1], [t for t in types if t], parts[3:]))
    return calls


def test_julia_binding_matches_the_c_header():
    """There is no Julia here, so the binding cannot run; what can be checked statically is that it cannot drift from the C ABI: every
    ccall of julia/LatticeQCDHIP.jl names a function declared in include/lqcd_hip.h, passes as many argument types as the C prototype
    has parameters and as many values as types, returns the declared type, and maps every parameter to a Julia type of the same width
    and kind (handles and arrays -> pointers, int -> Cint, double -> Float64/Cdouble, uint64_t -> UInt64)."""
    protos = _c_prototypes()
    calls = _julia_ccalls()
    assert len(calls) >= 50 and len(protos) >= 80
    ok_types = {
        "int": {"Cint"}, "double": {"Float64", "Cdouble"}, "uint64_t": {"UInt64"}, "int64_t": {"Int64"}, "size_t": {"Csize_t"},
    }
    used = set()
    for line, sym, ret, types, args in calls:
        assert sym in protos, f"julia/LatticeQCDHIP.jl:{line}: {sym} is not declared in include/lqcd_hip.h"
        cret, params = protos[sym]
        used.add(sym)
        assert len(types) == len(params), f"line {line}: {sym} takes {len(params)} parameters, the ccall passes {len(types)} types"
        assert len(args) == len(types), f"line {line}: {sym}: {len(types)} types but {len(args)} values"
        want_ret = {"int": "Cint", "constchar*": "Cstring", "int64_t": "Int64", "double": "Float64"}.get(cret)
        assert want_ret is None or ret.strip() == want_ret, f"line {line}: {sym} returns {cret}, the ccall says {ret}"
        for k, (ctype, jtype) in enumerate(zip(params, types)):
            ctype = ctype.replace("const ", "").strip()
            base = ctype.rsplit(" ", 1)[0].strip() if " " in ctype else ctype
            is_ptr = "*" in ctype or base.endswith("_t") and base.startswith("lqcd_") or "[" in ctype
            if is_ptr:
                assert jtype.startswith(("Ptr{", "Ref{")) or jtype in ("Cstring",), f"line {line}: {sym} argument {k} ({ctype}) -> {jtype}"
            else:
                assert jtype in ok_types.get(base, {jtype}), f"line {line}: {sym} argument {k} ({ctype}) -> {jtype}"
    # the per-direction entry points and the solvers the reference's callers need are all bound
    for need in ("lqcd_link_exp", "lqcd_link_mul", "lqcd_link_copy", "lqcd_link_add_ta", "lqcd_link_staple", "lqcd_solve_cg_DdagD",
                 "lqcd_solve_bicgstab", "lqcd_solve_bicg", "lqcd_solve_multishift_cg", "lqcd_solve_multishift_mixed_cg", "lqcd_action_create",
                 "lqcd_action_evaluate", "lqcd_action_force", "lqcd_action_sample_pseudofermions", "lqcd_rational_fit", "lqcd_op_apply",
                 "lqcd_op_create_domainwall", "lqcd_spinor_create_5d", "lqcd_spinor_slice", "lqcd_stout_smear", "lqcd_stout_backprop", "lqcd_link_mul_adj",
                 "lqcd_gauge_polyakov"):
        assert need in used, need
