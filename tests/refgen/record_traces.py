#!/usr/bin/env python3
"""Executed call traces of the reference's UNCHANGED callers -- BUILD CONTAINER ONLY (listed in .gpurunignore; needs /root/reference).

For every parameter set the tests use, the tree that tests/refgen/parse_callers.py derives from the reference's ten MD / HMC functions is EXECUTED here against a
recording binding: fields, actions and the two step sizes are opaque symbols; loop bounds, integrator flags and type parameters are the concrete values of the case.
Every generic the run reaches -- exptU!, mul!, substitute_U!, calc_dSdUμ!, Traceless_antihermitian_add!, calc_UdSfdU!, getindex, adjoint, the scalar arithmetic on
symbols -- is emitted as ONE flat entry (generic, argument slots / scalar values, result slots); loops arrive unrolled, branches on parameters resolved, locals
replaced by slot numbers.  The one branch that depends on a value only the device run knows (the accept test of update!) emits both arms, each entry tagged with
the outcome it belongs to.  What is committed is what the runs emitted: tests/golden/ref_exec_traces.json.  The GPU side (tests/ref_trace.py) is a loop over it.

Format (JSON):
  cases[name] = {"entry": generic executed, "args": [how the test's positional arguments are taken apart into input slots: {"slot", "path": [argument index, struct field...]}],
                 "shape": the structural parameters of the run (MDsteps, Nsw, QPQ, SextonWeingargten, quench, smeared),
                 "ops": [[generic, [arg...], [result slot...]] or [generic, [arg...], [result slot...], [slot, outcome]] or ["@enter" / "@leave", name] or ["@raise"]],
                 "result": arg | null, "watch": {"H_old": slot, "H_new": slot} (update! only)}
  arg = slot number | {"c": constant} | {"v": [arg...]} (a vector literal)

usage: python tests/refgen/record_traces.py [reference_root]      (rewrites tests/golden/ref_exec_traces.json)"""
import json
import math
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import parse_callers  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "golden", "ref_exec_traces.json")
NOTHING_TYPE = object()      # the TYPE Nothing (update! compares a value with it: always unequal, standardHMC.jl:67)
BASE = {"real", "div", "exp", "rand"}      # Base functions of Julia the callers use


class Sym:
    """An opaque value of the run: a field, an action, a scalar the device computes."""
    __slots__ = ("slot",)

    def __init__(self, slot):
        self.slot = slot


class Struct(dict):
    """A struct of the reference (StandardMD, StandardHMC): field name -> Sym or concrete value"""


class _Return(Exception):
    def __init__(self, value):
        self.value = value


class _Raise(Exception):
    pass


class Recorder:
    def __init__(self, functions, Dim=4):
        self.functions = {}
        for fn in functions:
            self.functions.setdefault(fn["name"], []).append(fn)
        self.Dim = Dim
        self.ops, self.nslots, self.inputs, self.when = [], 0, [], None
        self.watch = {}

    # ---- slots
    def new(self):
        self.nslots += 1
        return Sym(self.nslots - 1)

    def input(self, path):
        s = self.new()
        self.inputs.append({"slot": s.slot, "path": path})
        return s

    def enc(self, v):
        if isinstance(v, Sym):
            return v.slot
        if isinstance(v, (list, tuple)):
            return {"v": [self.enc(x) for x in v]}
        if v is None or isinstance(v, (bool, int, float, str)):
            return {"c": v}
        raise TypeError("cannot encode %r" % (v,))

    def emit(self, name, args, nout=1):
        outs = [self.new() for _ in range(nout)]
        op = [name, [self.enc(a) for a in args], [o.slot for o in outs]]
        if self.when is not None:
            op.append(list(self.when))
        self.ops.append(op)
        return outs

    # ---- calls
    def call(self, name, args, nout):
        if name in self.functions:
            best = None
            for fn in self.functions[name]:
                env = dict(zip(fn["params"], args))
                if all(self.constraint(env, k, v) for k, v in fn["dispatch"].items()) and (best is None or len(fn["dispatch"]) > len(best[0]["dispatch"])):
                    best = (fn, env)
            fn, env = best
            self.ops.append(["@enter", name])
            try:
                self.run(fn["steps"], env)
                out = None
            except _Return as r:
                out = r.value
            if name == "update!":      # what the tests read off a trajectory: the two Hamiltonians (by role, not by the callers' names)
                self.watch = {"H_old": env["Sold"].slot, "H_new": env["Snew"].slot}
            self.ops.append(["@leave", name])
            return out
        if name in BASE and not any(isinstance(a, Sym) for a in args) and name != "rand":
            return {"real": lambda z: z.real if isinstance(z, complex) else z, "div": lambda a, b: a // b, "exp": math.exp}[name](*args)
        outs = self.emit(name, args, max(nout, 1) if nout else 0)
        if nout == 0:
            return None
        return outs[0] if nout == 1 else outs

    def constraint(self, env, tparam, supertype):
        if tparam == "TC":
            return supertype != "CovNeuralnet" or env["md"]["cov_neural_net"] is not None
        return True

    # ---- expressions
    def value(self, e, env):
        if isinstance(e, (int, float)) or e is None:
            return e
        if isinstance(e, str):
            if e in env:
                return env[e]
            if e == "Dim":
                return self.Dim
            if e in ("false", "true", "nothing"):
                return {"false": False, "true": True, "nothing": None}[e]
            if e == "Nothing":
                return NOTHING_TYPE
            if e == "quench":                          # type parameter of StandardMD
                return env["md"]["quench"]
            if e == "TC":
                return NOTHING_TYPE if env["md"]["cov_neural_net"] is None else "CovNeuralnet"
            raise NameError(e)
        (k, v), = e.items()
        if k == "idx":
            base, i = self.value(v[0], env), self.value(v[1], env)
            if isinstance(base, (list, tuple)):
                return base[i - 1]
            return self.emit("getindex", [base, i])[0]
        if k == "dot":
            base = self.value(v[0], env)
            if isinstance(base, Struct):
                return base[v[1]]
            return self.emit("getproperty", [base, v[1]])[0]
        if k == "neg":
            x = self.value(v, env)
            return self.emit("-", [x])[0] if isinstance(x, Sym) else -x
        if k == "adj":
            return self.emit("adjoint", [self.value(v, env)])[0]
        if k == "vec":
            return [self.value(x, env) for x in v]
        if k == "call":
            return self.call(v[0], [self.value(x, env) for x in v[1:]], 1)
        if k == "op":
            a, b = self.value(v[1], env), self.value(v[2], env)
            if a is NOTHING_TYPE or b is NOTHING_TYPE:      # a value against the type Nothing: never equal
                return {"==": a is b, "!=": a is not b}[v[0]]
            if isinstance(a, Sym) or isinstance(b, Sym):
                return self.emit(v[0], [a, b])[0]
            return {"+": lambda: a + b, "-": lambda: a - b, "*": lambda: a * b, "/": lambda: a / b, ">=": lambda: a >= b, "<=": lambda: a <= b,
                    "<": lambda: a < b, ">": lambda: a > b, "==": lambda: a == b, "!=": lambda: a != b}[v[0]]()
        raise ValueError("unknown expression form %r" % (e,))

    # ---- statements
    def run(self, steps, env):
        for s in steps:
            if "call" in s:
                r = self.call(s["call"], [self.value(a, env) for a in s["args"]], len(s["out"]))
                if len(s["out"]) == 1:
                    env[s["out"][0]] = r
                elif s["out"]:
                    for n, x in zip(s["out"], r):
                        env[n] = x
            elif "set" in s:
                env[s["set"]] = self.value(s["expr"], env)
            elif "add" in s:
                x = self.value(s["expr"], env)
                cur = env[s["add"]]
                env[s["add"]] = self.emit("+", [cur, x])[0] if isinstance(cur, Sym) or isinstance(x, Sym) else cur + x
            elif "for" in s:
                for i in range(self.value(s["from"], env), self.value(s["to"], env) + 1):
                    env[s["for"]] = i
                    self.run(s["do"], env)
            elif "if" in s:
                c = self.value(s["if"], env)
                if isinstance(c, Sym):      # known only to the device run: both arms, each entry tagged with its outcome
                    assert self.when is None, "nested run-time branches do not occur in the callers"
                    for outcome, arm in ((True, s["then"]), (False, s["else"])):
                        self.when = (c.slot, outcome)
                        self.run(arm, env)
                    self.when = None
                else:
                    self.run(s["then"] if c else s["else"], env)
            elif "return" in s:
                raise _Return(self.value(s["return"], env))
            elif "raise" in s:
                raise _Raise()
            else:
                raise ValueError("unknown step %r" % (s,))


def make_md(rec, argi, shape):
    """A StandardMD as the traced functions read it (field names: tests/golden/ref_caller_inventory.json "struct_fields"): fields and actions are inputs of the
    run, the step size too; the integrator's flags and loop bounds are the concrete values of the case."""
    quench, smeared = shape["quench"], shape["smeared"]
    md = Struct({"QPQ": shape["QPQ"], "SextonWeingargten": shape["SextonWeingargten"], "Nsw": shape["Nsw"], "MDsteps": shape["MDsteps"], "quench": quench})
    for f in ("gauge_action", "p", "Δτ"):
        md[f] = rec.input(argi + [f])
    for f in ("fermi_action", "η", "ξ"):
        md[f] = None if quench else rec.input(argi + [f])
    md["cov_neural_net"] = rec.input(argi + ["cov_neural_net"]) if smeared else None
    md["dSdU"] = rec.input(argi + ["dSdU"]) if smeared else None
    return md


def shape_of(scheme="QPQ", quench=False, smeared=False, MDsteps=1, Nsw=2):
    return {"QPQ": scheme != "PQP" and scheme != "PQP_sw", "SextonWeingargten": scheme.endswith("_sw"), "Nsw": Nsw, "MDsteps": MDsteps, "quench": quench, "smeared": smeared}


def case_name(entry, shape):
    """The key tests/ref_trace.py looks a run up by: the generic + the structural parameters it was executed with."""
    if entry in ("U_update!", "P_update!"):
        return entry
    who = ("quenched" if shape["quench"] else "dynamical") + ("+smeared" if shape["smeared"] else "")
    if entry in ("P_update_fermion!", "initialize_MD!"):
        return "%s/%s" % (entry, who)
    scheme = ("QPQ" if shape["QPQ"] else "PQP") + ("_sw%d" % shape["Nsw"] if shape["SextonWeingargten"] else "")
    return "%s/%s/%s/steps%d" % (entry, scheme, who, shape["MDsteps"])


def record(functions, entry, shape):
    rec = Recorder(functions)
    if entry == "update!":                 # update!(updatemethod, U)
        hmc = Struct({"md": make_md(rec, [0, "md"], shape), "Uold": rec.input([0, "Uold"])})
        args = [hmc, rec.input([1])]
    elif entry in ("runMD!", "initialize_MD!"):      # (U, md)
        U = rec.input([0])
        args = [U, make_md(rec, [1], shape)]
    else:                                  # U_update! / P_update! / P_update_fermion! (U, p, eps, md)
        U, p, eps = rec.input([0]), rec.input([1]), rec.input([2])
        args = [U, p, eps, make_md(rec, [3], shape)]
    raised = False
    try:
        result = rec.call(entry, args, 1)
    except _Raise:
        rec.ops.append(["@raise"])
        result, raised = None, True
    return {"entry": entry, "shape": shape, "args": rec.inputs, "ops": rec.ops, "result": None if result is None else rec.enc(result), "watch": rec.watch,
            "raises": raised, "slots": rec.nslots}


# the runs the tests replay (tests/test_gpu_reference_callers.py, test_gpu_stout.py, test_gpu_domainwall.py, test_ref_call_trace.py)
CASES = [("U_update!", shape_of()), ("P_update!", shape_of()),
         ("P_update_fermion!", shape_of()), ("P_update_fermion!", shape_of(smeared=True)),
         ("initialize_MD!", shape_of()), ("initialize_MD!", shape_of(smeared=True)), ("initialize_MD!", shape_of(quench=True)),
         ("runMD!", shape_of("PQP_sw", quench=True, MDsteps=2)),                          # what the reference refuses (standardMD.jl:103-124)
         ("runMD!", shape_of("QPQ", MDsteps=3)), ("runMD!", shape_of("QPQ_sw", MDsteps=3, Nsw=4)), ("runMD!", shape_of("PQP", MDsteps=3)), ("runMD!", shape_of("QPQ", quench=True, MDsteps=3)),
         ("update!", shape_of("QPQ_sw", MDsteps=20, Nsw=10)), ("update!", shape_of("QPQ", MDsteps=20)), ("update!", shape_of("PQP", MDsteps=20)),
         ("update!", shape_of("QPQ", quench=True, MDsteps=20)), ("update!", shape_of("PQP", quench=True, MDsteps=20)),
         ("update!", shape_of("QPQ", smeared=True, MDsteps=10)), ("update!", shape_of("QPQ", smeared=True, MDsteps=20)),
         ("update!", shape_of("QPQ", MDsteps=3))]


def build(root):
    functions = parse_callers.build(root)["functions"]
    out = {"generated_by": "tests/refgen/record_traces.py", "sources": sorted({"%s:%d" % (f["file"], f["line"]) for f in functions}), "cases": {}}
    for entry, shape in CASES:
        out["cases"][case_name(entry, shape)] = record(functions, entry, shape)
    return out


def dump(tr, path):
    """one emitted entry per line: the file reads (and diffs) as the log it is"""
    with open(path, "w", encoding="utf-8") as f:
        f.write('{"generated_by": %s,\n "sources": %s,\n "cases": {\n' % (json.dumps(tr["generated_by"]), json.dumps(tr["sources"])))
        names = sorted(tr["cases"])
        for k, name in enumerate(names):
            c = tr["cases"][name]
            f.write('  %s: {"entry": %s, "shape": %s, "raises": %s, "slots": %d, "result": %s, "watch": %s,\n   "args": %s,\n   "ops": [\n' % (
                json.dumps(name, ensure_ascii=False), json.dumps(c["entry"], ensure_ascii=False), json.dumps(c["shape"], sort_keys=True), json.dumps(c["raises"]), c["slots"],
                json.dumps(c["result"]), json.dumps(c["watch"], sort_keys=True), json.dumps(c["args"], ensure_ascii=False)))
            f.write(",\n".join("    " + json.dumps(op, ensure_ascii=False, separators=(",", ":")) for op in c["ops"]))
            f.write("\n   ]}%s\n" % ("," if k + 1 < len(names) else ""))
        f.write(" }\n}\n")


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    tr = build(root)
    dump(tr, OUT)
    assert json.load(open(OUT, encoding="utf-8")) == json.loads(json.dumps(tr))
    print("%d cases, %d entries, %d bytes" % (len(tr["cases"]), sum(len(c["ops"]) for c in tr["cases"].values()), os.path.getsize(OUT)))
