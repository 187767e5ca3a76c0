"""Replays the reference's callers from their call trace (tests/golden/ref_call_trace.json, generated from /root/reference by
tests/golden/make_ref_call_trace.py) against a binding: every "call" step names a generic -- exptU!, mul!, substitute_U!, calc_dSdUμ!,
Traceless_antihermitian_add!, calc_UdSfdU!, ... -- and is dispatched to the binding function of that name (`!` -> `_`, `μ` -> `mu`), or to another
traced function.  No function body of the reference lives in tests/: the order of the calls, the roles of their arguments, loop bounds and conditions
are data.  What this interpreter supplies: Julia's value semantics for the handful of expression forms the trace uses (1-based indexing of vectors,
inclusive ranges, div, real, adjoint), method dispatch on the trace's "dispatch" constraints, and the random numbers (the device generators are
counter based and take a seed; the accept test draws from numpy) -- the reference's own RNG stream is not reproducible outside Julia either."""
import json
import math
import os

import numpy as np

TRACE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_call_trace.json")
NOTHING_TYPE = object()      # the TYPE Nothing (update! compares a value with it: always unequal, standardHMC.jl:67)


class Fields(dict):
    """A struct of the reference (StandardMD, StandardHMC) as the trace sees it: fields by their reference names."""
    __getattr__ = dict.__getitem__


class Raised(RuntimeError):
    pass


class _Return(Exception):
    def __init__(self, value):
        self.value = value


class Replay:
    def __init__(self, lq, seed=0, Dim=4, hooks=None):
        self.lq, self.Dim, self.seed, self.rng = lq, Dim, seed, np.random.default_rng(seed)
        self.hooks = hooks or {}                      # {("before" | "after", traced function name): callable(env)}
        self.log = []                                 # names of the generics called, in order
        with open(TRACE, encoding="utf-8") as f:
            self.functions = {}
            for fn in json.load(f)["functions"]:
                self.functions.setdefault(fn["name"], []).append(fn)
        self.builtins = {"real": lambda z: float(np.real(z)), "div": lambda a, b: a // b, "exp": math.exp, "rand": lambda: self.rng.random(),
                         "gauss_distribution!": self._seeded(lq.gauss_distribution_), "gauss_sampling_in_action!": self._seeded(lq.gauss_sampling_in_action_)}

    def _seeded(self, fn):
        def call(*args):
            self.seed += 1
            return fn(*args, self.seed)
        return call

    # ---- calls: another traced function (method chosen by its dispatch constraints), a builtin, or the binding's generic of that name
    def call(self, name, *args):
        if name in self.functions:
            env0 = None
            best = None
            for fn in self.functions[name]:
                env = dict(zip(fn["params"], args))
                if all(self._constraint(env, k, v) for k, v in fn["dispatch"].items()) and (best is None or len(fn["dispatch"]) > len(best[0]["dispatch"])):
                    best, env0 = (fn, env), env
            fn, env = best
            if ("before", name) in self.hooks:
                self.hooks[("before", name)](env)
            try:
                self.run(fn["steps"], env)
                out = None
            except _Return as r:
                out = r.value
            if ("after", name) in self.hooks:
                self.hooks[("after", name)](env)
            return out
        self.log.append(name)
        if name in self.builtins:
            return self.builtins[name](*args)
        return getattr(self.lq, name.replace("!", "_").replace("μ", "mu"))(*args)

    def _constraint(self, env, tparam, supertype):
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
            return base[i - 1] if isinstance(base, (list, tuple)) else base[i]      # device fields index their directions 1..Dim themselves
        if k == "dot":
            base = self.value(v[0], env)
            return base[v[1]] if isinstance(base, dict) else getattr(base, v[1])
        if k == "neg":
            return -self.value(v, env)
        if k == "not":
            return not self.value(v, env)
        if k == "adj":
            return self.value(v, env).adjoint()
        if k == "vec":
            return [self.value(x, env) for x in v]
        if k == "call":
            return self.call(v[0], *[self.value(x, env) for x in v[1:]])
        if k == "op":
            a, b = self.value(v[1], env), self.value(v[2], env)
            if v[0] == "==":
                return a is b if (a is NOTHING_TYPE or b is NOTHING_TYPE) else a == b
            if v[0] == "!=":
                return a is not b if (a is NOTHING_TYPE or b is NOTHING_TYPE) else a != b
            return {"+": lambda: a + b, "-": lambda: a - b, "*": lambda: a * b, "/": lambda: a / b, ">=": lambda: a >= b, "<=": lambda: a <= b,
                    "<": lambda: a < b, ">": lambda: a > b, "&&": lambda: a and b, "||": lambda: a or b}[v[0]]()
        raise ValueError("unknown expression form %r" % (e,))

    # ---- steps
    def run(self, steps, env):
        for s in steps:
            if "call" in s:
                r = self.call(s["call"], *[self.value(a, env) for a in s["args"]])
                if len(s["out"]) == 1:
                    env[s["out"][0]] = r
                elif s["out"]:
                    for n, x in zip(s["out"], r):
                        env[n] = x
            elif "set" in s:
                env[s["set"]] = self.value(s["expr"], env)
            elif "add" in s:
                env[s["add"]] = env[s["add"]] + self.value(s["expr"], env)
            elif "for" in s:
                for i in range(self.value(s["from"], env), self.value(s["to"], env) + 1):
                    env[s["for"]] = i
                    self.run(s["do"], env)
            elif "if" in s:
                self.run(s["then"] if self.value(s["if"], env) else s["else"], env)
            elif "return" in s:
                raise _Return(self.value(s["return"], env))
            elif "raise" in s:
                raise Raised("the reference raises here")
            else:
                raise ValueError("unknown step %r" % (s,))


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
