"""CPU checks of the call-trace fixture (tests/golden/ref_call_trace.json) and of the interpreter that replays it (tests/ref_trace.py):
the committed JSON is what tests/golden/make_ref_call_trace.py derives from /root/reference (checked whenever that tree is present -- it is not on the
GPU box), every generic the trace names exists in both bindings, and a replay against a recording stand-in of the binding makes the calls the
integrators are known to make (counts per direction, per MD step, per scheme)."""
import json
import os
import sys

import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_trace import Fields, Raised, Replay  # noqa: E402

REF = "/root/reference"


def _trace():
    with open(os.path.join(GOLDEN, "ref_call_trace.json"), encoding="utf-8") as f:
        return json.load(f)


def _calls(steps, out):
    for s in steps:
        if "call" in s:
            out.append(s["call"])
        for key in ("do", "then", "else"):
            if key in s:
                _calls(s[key], out)
        for e in [s.get("expr"), s.get("if"), s.get("return"), s.get("from"), s.get("to")] + list(s.get("args", [])):
            _expr_calls(e, out)
    return out


def _expr_calls(e, out):
    if isinstance(e, dict):
        (k, v), = e.items()
        if k == "call":
            out.append(v[0])
            v = v[1:]
        for x in (v if isinstance(v, list) else [v]):
            _expr_calls(x, out)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")
def test_committed_trace_is_what_the_generator_derives_from_the_reference():
    sys.path.insert(0, GOLDEN)
    import make_ref_call_trace as gen
    assert gen.build(REF) == _trace(), "tests/golden/ref_call_trace.json is stale: run python tests/golden/make_ref_call_trace.py"


def test_trace_covers_the_callers_of_survey_8a_and_holds_no_source_text():
    tr = _trace()
    names = sorted((f["name"], f["file"]) for f in tr["functions"])
    assert names == sorted([("U_update!", "src/md/AbstractMD.jl"), ("P_update!", "src/md/AbstractMD.jl"), ("P_update_fermion!", "src/md/AbstractMD.jl"),
                            ("P_update_fermion!", "src/md/standardMD.jl"), ("initialize_MD!", "src/md/standardMD.jl"), ("runMD!", "src/md/standardMD.jl"),
                            ("runMD_QPQ!", "src/md/standardMD.jl"), ("runMD_QPQ_sw!", "src/md/standardMD.jl"), ("runMD_PQP!", "src/md/standardMD.jl"),
                            ("update!", "src/updates/standardHMC.jl")])
    text = json.dumps(tr, ensure_ascii=False)
    assert "#" not in text and "println" not in text and "::" not in text and "where" not in text      # no comments, printing, annotations or signatures
    assert [f["dispatch"] for f in tr["functions"] if f["name"] == "P_update_fermion!"] == [{}, {"TC": "CovNeuralnet"}]


def test_every_generic_of_the_trace_is_served(lq):
    """... by the Python mirror directly, and by the Julia binding through the caller inventory (tests/test_julia_binding_static.py resolves every call of
    ref_caller_inventory.json against julia/LatticeQCDHIP.jl: the trace may not name a generic the inventory does not)."""
    tr = _trace()
    traced = {f["name"] for f in tr["functions"]}
    builtins = set(Replay(lq).builtins)
    inv = json.load(open(os.path.join(GOLDEN, "ref_caller_inventory.json"), encoding="utf-8"))
    inv_names = {c["name"] for c in inv["calls"]}
    generics = set()
    for f in tr["functions"]:
        generics.update(_calls(f["steps"], []))
    for g in sorted(generics - traced):
        if g in ("real", "div", "exp", "rand"):                                # Base functions of Julia
            continue
        assert g in inv_names, g
        assert g in builtins or hasattr(lq, g.replace("!", "_").replace("μ", "mu")), "the Python mirror lacks " + g


class _Recorder:
    """A stand-in for the binding: every generic is recorded, get_temp hands out names, fields are plain objects indexed 1..4."""

    class Field:
        def __init__(self, name):
            self.name, self.NC = name, 3

        def __getitem__(self, mu):
            assert 1 <= mu <= 4
            return _Recorder.Link("%s[%d]" % (self.name, mu))

        def __mul__(self, other):
            return 2.0

        def similar(self):
            return _Recorder.Field(self.name + "'")

    class Link(str):
        NC = 3

        def adjoint(self):
            return _Recorder.Link(self + "'")

    def __init__(self):
        self.calls, self.ntemp, self.sf = [], 0, 1.0

    def __getattr__(self, name):
        def generic(*args):
            self.calls.append((name, args))
            if name == "get_temp":
                self.ntemp += 1
                n = args[1] if len(args) > 1 else None
                return (["t%d_%d" % (self.ntemp, k) for k in range(n)], list(range(n))) if n else ("t%d" % self.ntemp, self.ntemp)
            if name == "evaluate_GaugeAction":
                return 3.0
            if name == "evaluate_FermiAction":
                return self.sf
            if name == "dot":
                return 1.0 + 0j
            if name == "calc_smearedU":
                return args[0], None, None
            if name == "back_prop":
                return _Recorder.Field("dSdUbare")
            return None
        return generic


def _md(rec, **kw):
    base = {"gauge_action": "ga", "quench": False, "Δτ": 0.1, "MDsteps": 3, "p": rec.Field("p"), "QPQ": True, "fermi_action": "fa", "η": "eta", "ξ": "xi",
            "SextonWeingargten": False, "Nsw": 4, "cov_neural_net": None, "dSdU": None}
    base.update(kw)
    return Fields(base)


def test_replay_makes_the_calls_the_integrators_are_known_to_make():
    rec = _Recorder()
    rp = Replay(rec)
    U = rec.Field("U")
    rp.call("U_update!", U, rec.Field("p"), 0.5, _md(rec))
    names = [c[0] for c in rec.calls]
    assert names.count("exptU_") == names.count("mul_") == names.count("substitute_U_") == 4 and names.count("get_temp") == names.count("unused_") == 4
    exp_calls = [c for c in rec.calls if c[0] == "exptU_"]
    assert [c[1][2] for c in exp_calls] == ["p[%d]" % mu for mu in (1, 2, 3, 4)] and all(abs(c[1][1] - 0.05) < 1e-15 for c in exp_calls)      # eps * dtau
    rec.calls.clear()
    rp.call("P_update!", U, rec.Field("p"), 1.0, _md(rec))
    ta = [c for c in rec.calls if c[0] == "Traceless_antihermitian_add_"]
    assert len(ta) == 4 and all(abs(c[1][1] + 0.1 / 3) < 1e-15 for c in ta)                                                                  # factor = -eps dtau / NC
    for scheme, kw, n_u, n_pg, n_pf in (("QPQ", {}, 2, 1, 1), ("QPQ_sw", {"SextonWeingargten": True}, 8, 4, 1), ("PQP", {"QPQ": False}, 1, 2, 2)):
        rec.calls.clear()
        rp.log.clear()
        md = _md(rec, **kw)
        rp.call("runMD!", U, md)
        names = [c[0] for c in rec.calls]
        assert names.count("exptU_") == 4 * n_u * md["MDsteps"], scheme
        assert names.count("calc_dSdUmu_") == 4 * n_pg * md["MDsteps"], scheme
        assert names.count("calc_UdSfdU_") == n_pf * md["MDsteps"], scheme
    rec.calls.clear()
    rp.call("runMD!", U, _md(rec, quench=True))
    assert not any(c[0] == "calc_UdSfdU_" for c in rec.calls)
    with pytest.raises(Raised):
        rp.call("runMD!", U, _md(rec, QPQ=False, SextonWeingargten=True))
    # update!: the accept test draws one uniform deviate; a rejected trajectory copies the old links back
    rec.calls.clear()
    acc = rp.call("update!", Fields({"md": _md(rec), "Uold": rec.Field("Uold")}), U)
    subs = [c for c in rec.calls if c[0] == "substitute_U_" and not isinstance(c[1][0], str)]
    assert acc is True and len(subs) == 1                      # Snew == Sold in the stand-in: exp(0) >= rand() -- accepted, only the save of the old links
    rec.calls.clear()
    rec.sf = 60.0                                              # the action jumps: exp(-59) < rand() -- rejected, the old links come back
    acc = rp.call("update!", Fields({"md": _md(rec), "Uold": rec.Field("Uold")}), U)
    subs = [c[1] for c in rec.calls if c[0] == "substitute_U_" and not isinstance(c[1][0], str)]
    assert acc is False and len(subs) == 2 and subs[0][0].name == "Uold" and subs[1][0].name == "U" and subs[1][1].name == "Uold"
    rec.sf = 1.0
    # the smeared method of P_update_fermion! is chosen by its dispatch constraint
    rec.calls.clear()
    rp.call("P_update_fermion!", U, rec.Field("p"), 1.0, _md(rec, cov_neural_net="nn", dSdU=rec.Field("dSdU")))
    assert [c[0] for c in rec.calls].count("back_prop") == 1
    rec.calls.clear()
    rp.call("P_update_fermion!", U, rec.Field("p"), 1.0, _md(rec))
    assert [c[0] for c in rec.calls].count("back_prop") == 0
