"""Operator-level golden vectors produced by the REFERENCE ITSELF (scripts/ref_parity_dump.jl: LatticeDiracOperators.jl / Gaugefields.jl
on the reference's own 4^4 fixtures, closed-form source).  Nothing in this image can produce them (no Julia); the day a host with Julia
and the reference's packages runs the script and commits tests/golden/ref_*.bin, the consumers below turn "parity unpinned" into a
pinned comparison without a code change.  Shared by tests/test_gpu_reference_vectors.py, tests/test_oracle_reference_vectors.py, bench.py."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
L = (4, 4, 4, 4)
KAPPA, MASS, BC = 0.141139, 0.5, (1, 1, 1, -1)
FIXTURE = {"wilson": "wilson_4x4x4x4.ildg", "staggered": "staggered_4x4x4x4.ildg"}
FILES = ["ref_%s_%s.bin" % (k, w) for k in ("wilson", "staggered") for w in ("D", "Ddag", "cg_x")] + ["ref_parity_meta.json"]
ABSENT = ("reference vectors absent: run `julia scripts/ref_parity_dump.jl tests/golden` on a host with Julia and the reference's "
          "packages (Gaugefields.jl, LatticeDiracOperators.jl) and commit tests/golden/ref_*.bin")


def available(directory=GOLDEN):
    return all(os.path.exists(os.path.join(directory, f)) for f in FILES)


def closed_form_source(kind):
    """psi[c,x,y,z,t,s] of scripts/ref_parity_dump.jl (0-based indices) as the C ABI's reference-layout array: (4,T,Z,Y,X,3) / (T,Z,Y,X,3)."""
    ns = 4 if kind == "wilson" else 1
    s, t, z, y, x, c = np.meshgrid(np.arange(ns), np.arange(L[3]), np.arange(L[2]), np.arange(L[1]), np.arange(L[0]), np.arange(3), indexing="ij")
    re = np.sin(0.37 + 0.11 * c + 0.23 * x + 0.31 * y + 0.43 * z + 0.59 * t + 0.71 * s)
    im = np.cos(0.19 + 0.13 * c + 0.29 * x + 0.37 * y + 0.41 * z + 0.53 * t + 0.61 * s)
    psi = (re + 1j * im).astype(np.complex128)
    return psi if kind == "wilson" else psi[0]


def load(kind, which, directory=GOLDEN):
    """ref_<kind>_<which>.bin as an array of the C ABI's reference layout."""
    ns = 4 if kind == "wilson" else 1
    raw = np.fromfile(os.path.join(directory, "ref_%s_%s.bin" % (kind, which)), dtype="<f8")
    assert raw.size == 2 * 3 * 256 * ns, "ref_%s_%s.bin has %d doubles" % (kind, which, raw.size)
    a = raw.view(np.complex128).reshape((ns, L[3], L[2], L[1], L[0], 3))      # Julia column-major (c,x,y,z,t,s) == C order (s,t,z,y,x,c)
    return a if kind == "wilson" else a[0]


def meta(directory=GOLDEN):
    with open(os.path.join(directory, "ref_parity_meta.json")) as f:
        return json.load(f)
