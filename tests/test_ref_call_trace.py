"""CPU checks of the executed call traces (tests/golden/ref_exec_traces.json) and of the loop that replays them (tests/ref_trace.py): the committed file is what
tests/refgen/record_traces.py emits when it executes the reference's callers against its recording binding (checked whenever /root/reference is present -- it is
not on the GPU box, and neither are the parser and the recorder: .gpurunignore), the file holds runs, not programs (no loop, branch or expression node, none of the
callers' local names), every generic the traces name exists in both bindings, and a replay against a recording stand-in of the binding makes the calls the
integrators are known to make (counts per direction, per MD step, per scheme)."""
import json
import os
import re
import sys

import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_trace import Fields, Raised, Replay, cases  # noqa: E402

REF = "/root/reference"
REFGEN = os.path.join(ROOT, "tests", "refgen")


@pytest.mark.skipif(not (os.path.isdir(REF) and os.path.isdir(REFGEN)), reason="the reference tree and the recorder exist only in the build container")
def test_committed_traces_are_what_the_recorder_emits_from_the_reference():
    sys.path.insert(0, REFGEN)
    import record_traces as gen
    assert json.loads(json.dumps(gen.build(REF))) == json.load(open(os.path.join(GOLDEN, "ref_exec_traces.json"), encoding="utf-8")), \
        "tests/golden/ref_exec_traces.json is stale: run python tests/refgen/record_traces.py"


def test_golden_files_hold_runs_not_programs():
    """no file under tests/golden/ may carry the callers' program -- no control-flow or expression node, none of their local identifiers; and the
    parser / recorder that do read the reference stay in the build container."""
    locals_of_the_callers = ["Sp", "Sg", "Sold", "Sfold", "Snew", "Sfnew", "Uout", "Uout_multi", "accept", "temps", "temp1", "temp2", "it_temp1", "it_temp2", "expU",
                             "it_expU", "W", "it_W", "dSdUμ", "its_dSdUμ", "UdSfdUμ", "its_UdSfdUμ", "factor", "updatemethod", "itrj", "isw", "dSdUbare"]
    for fn in sorted(os.listdir(GOLDEN)):
        if not fn.endswith(".json"):
            continue
        text = open(os.path.join(GOLDEN, fn), encoding="utf-8").read()
        for key in ('"if"', '"for"', '"op"', '"then"', '"else"', '"do"', '"steps"', '"dispatch"'):
            assert key not in text, (fn, key)
        if fn == "ref_exec_traces.json":
            strings = set(re.findall(r'"((?:[^"\\]|\\.)*)"', text))
            assert not (strings & set(locals_of_the_callers)), strings & set(locals_of_the_callers)
            assert "#" not in text and "println" not in text and "::" not in text and "where" not in text
    assert not os.path.exists(os.path.join(GOLDEN, "ref_call_trace.json"))
    ignore = open(os.path.join(ROOT, ".gpurunignore")).read().split()
    assert "tests/refgen/" in ignore or "tests/refgen" in ignore


def test_traces_cover_the_callers_of_survey_8a():
    cs = cases()
    assert {c["entry"] for c in cs.values()} == {"U_update!", "P_update!", "P_update_fermion!", "initialize_MD!", "runMD!", "update!"}
    entered = set()
    for c in cs.values():
        entered.update(op[1] for op in c["ops"] if op[0] == "@enter")
    assert entered == {"U_update!", "P_update!", "P_update_fermion!", "initialize_MD!", "runMD!", "runMD_QPQ!", "runMD_QPQ_sw!", "runMD_PQP!", "update!"}
    assert cs["runMD!/PQP_sw2/quenched/steps2"]["raises"] and cs["runMD!/PQP_sw2/quenched/steps2"]["ops"][-1] == ["@raise"]
    # both methods of P_update_fermion! were reached: the smeared one goes through back_prop
    assert any(op[0] == "back_prop" for op in cs["P_update_fermion!/dynamical+smeared"]["ops"])
    assert not any(op[0] == "back_prop" for op in cs["P_update_fermion!/dynamical"]["ops"])
    # the only entries tagged with a run-time outcome: the restore of the old links of a rejected trajectory
    for name, c in cs.items():
        tagged = [op for op in c["ops"] if len(op) == 4]
        assert [op[0] for op in tagged] == (["substitute_U!"] if c["entry"] == "update!" else []), name
        assert all(op[3][1] is False for op in tagged)


def test_every_generic_of_the_traces_is_served(lq):
    """... by the Python mirror directly, and by the Julia binding through the caller inventory (tests/test_julia_binding_static.py resolves every call of
    ref_caller_inventory.json against julia/LatticeQCDHIP.jl: the traces may not name a generic the inventory does not)."""
    base = set(Replay(lq).base)
    inv = json.load(open(os.path.join(GOLDEN, "ref_caller_inventory.json"), encoding="utf-8"))
    inv_names = {c["name"] for c in inv["calls"]}
    generics = set()
    for c in cases().values():
        generics.update(op[0] for op in c["ops"] if op[0][0] != "@")
    for g in sorted(generics):
        if g in ("real", "exp", "rand", "getindex", "getproperty", "adjoint", "+", "-", "*", "/", ">="):                # Base functions / syntax of Julia
            continue
        assert g in inv_names, g
        assert g in base or hasattr(lq, g.replace("!", "_").replace("μ", "mu")), "the Python mirror lacks " + g


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
        rp.call("runMD!", U, _md(rec, QPQ=False, SextonWeingargten=True, Nsw=2, MDsteps=2, quench=True))
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
