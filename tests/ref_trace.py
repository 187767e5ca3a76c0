"""Replays the reference's callers from their EXECUTED traces (tests/golden/ref_exec_traces.json: what runs of the reference's MD / HMC functions against a recording
binding emitted in the build container, tests/refgen/record_traces.py) against a real binding.  A trace is a flat list of entries (generic, argument slots / scalar
values, result slots); this file is the loop over it: look the arguments up, call the binding function of that name (`!` -> `_`, `μ` -> `mu`), store the results.
No expression evaluator, no method dispatch, no loops or branches of the callers live here -- an entry tagged with an outcome of the accept test is skipped when the
run took the other one.  What the loop supplies: the handful of Base generics the traces name (getindex on 1-based vectors, adjoint, scalar arithmetic, exp, rand)
and the random numbers (the device generators are counter based and take a seed; the accept test draws from numpy) -- the reference's own RNG stream is not
reproducible outside Julia either."""
import json
import math
import os

import numpy as np

TRACES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_exec_traces.json")
_cases = None


def cases():
    global _cases
    if _cases is None:
        with open(TRACES, encoding="utf-8") as f:
            _cases = json.load(f)["cases"]
    return _cases


class Fields(dict):
    """A struct of the reference (StandardMD, StandardHMC) as the traces see it: fields by their reference names."""
    __getattr__ = dict.__getitem__


class Raised(RuntimeError):
    pass


def case_name(entry, md):
    """the run of `entry` that was recorded with this StandardMD's structural parameters"""
    if entry in ("U_update!", "P_update!"):
        return entry
    who = ("quenched" if md["quench"] else "dynamical") + ("+smeared" if md["cov_neural_net"] is not None else "")
    if entry in ("P_update_fermion!", "initialize_MD!"):
        return "%s/%s" % (entry, who)
    scheme = ("QPQ" if md["QPQ"] else "PQP") + ("_sw%d" % md["Nsw"] if md["SextonWeingargten"] else "")
    return "%s/%s/%s/steps%d" % (entry, scheme, who, md["MDsteps"])


class Replay:
    def __init__(self, lq, seed=0, hooks=None):
        self.lq, self.seed, self.rng = lq, seed, np.random.default_rng(seed)
        self.hooks = hooks or {}                      # {("before" | "after", name of a traced function): callable(replay)}
        self.log = []                                 # names of the generics called, in order
        self.slots = []
        self.base = {"getindex": lambda a, i: a[i - 1] if isinstance(a, (list, tuple)) else a[i],      # device fields index their directions 1..Dim themselves
                     "getproperty": getattr, "adjoint": lambda a: a.adjoint(),
                     "+": lambda a, b: a + b, "*": lambda a, b: a * b, "/": lambda a, b: a / b, ">=": lambda a, b: a >= b,
                     "-": lambda a, *b: a - b[0] if b else -a,
                     "real": lambda z: float(np.real(z)), "exp": math.exp, "rand": lambda: self.rng.random(),
                     "gauss_distribution!": self._seeded(lq.gauss_distribution_), "gauss_sampling_in_action!": self._seeded(lq.gauss_sampling_in_action_)}

    def _seeded(self, fn):
        def call(*args):
            self.seed += 1
            return fn(*args, self.seed)
        return call

    def arg(self, a):
        if isinstance(a, int):
            return self.slots[a]
        if "c" in a:
            return a["c"]
        return [self.arg(x) for x in a["v"]]

    def watch(self, role):
        return self.slots[self.case["watch"][role]]

    def call(self, entry, *args):
        md = args[0]["md"] if entry == "update!" else args[-1]
        self.case = case = cases()[case_name(entry, md)]
        self.slots = [None] * case["slots"]
        for inp in case["args"]:      # the test's arguments, taken apart into the input slots of the run
            v = args[inp["path"][0]]
            for field in inp["path"][1:]:
                v = v[field]
            self.slots[inp["slot"]] = v
        for op in case["ops"]:
            name = op[0]
            if name[0] == "@":
                if name == "@raise":
                    raise Raised("the reference raises here")
                hook = self.hooks.get(("before" if name == "@enter" else "after", op[1]))
                if hook:
                    hook(self)
                continue
            if len(op) == 4 and bool(self.slots[op[3][0]]) != op[3][1]:
                continue                                           # emitted under the other outcome of the accept test
            self.log.append(name)
            fn = self.base.get(name) or getattr(self.lq, name.replace("!", "_").replace("μ", "mu"))
            r = fn(*[self.arg(a) for a in op[1]])
            if len(op[2]) == 1:
                self.slots[op[2][0]] = r
            elif op[2]:
                for s, x in zip(op[2], r):
                    self.slots[s] = x
        return None if case["result"] is None else self.arg(case["result"])


def standard_md(lq, U, gauge_action, dtau, MDsteps, fermi_action=None, cov_neural_net=None, QPQ=True, SextonWeingargten=False, Nsw=2):
    """The fields of a StandardMD the traced functions read (names: tests/golden/ref_caller_inventory.json "struct_fields"), filled the way its
    constructor's call sites ask (momenta like U, eta / xi like the action's temporaries, dSdU like U when the links are smeared)."""
    quench = fermi_action is None
    eta = None if quench else fermi_action._temporary_fermionfields[0].similar()
    return Fields({"gauge_action": gauge_action, "quench": quench, "Δτ": dtau, "MDsteps": MDsteps, "p": lq.initialize_TA_Gaugefields(U), "QPQ": QPQ,
                   "fermi_action": fermi_action, "η": eta, "ξ": None if quench else eta.similar(), "SextonWeingargten": SextonWeingargten, "Nsw": Nsw,
                   "cov_neural_net": cov_neural_net, "dSdU": U.similar() if cov_neural_net is not None else None})


def standard_hmc(lq, U, md):
    return Fields({"md": md, "Uold": U.similar()})
