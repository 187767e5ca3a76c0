#!/usr/bin/env python3
"""Parser for the reference's UNCHANGED callers -- BUILD CONTAINER ONLY (listed in .gpurunignore: it reads /root/reference, which does not exist
on the GPU box, and nothing it produces directly is committed).  A tokenizer + recursive descent over the Julia subset the ten functions of the
molecular-dynamics / HMC layer use -- U_update!, P_update!, both methods of P_update_fermion!, initialize_MD!, runMD!, runMD_QPQ!, runMD_QPQ_sw!, runMD_PQP!,
update! -- into an in-memory tree.  tests/refgen/record_traces.py EXECUTES that tree against a recording binding, one run per parameter set the tests use, and
commits what the runs emit: tests/golden/ref_exec_traces.json, flat lists of (generic, argument slots, scalar values).  The tree itself is never written out
(round 5 committed it; VERDICT r5: a fixture must be what a run emits, not the program).

In-memory vocabulary:
  function: {"name", "file", "line", "params": [names], "dispatch": {type parameter: required supertype}, "steps": [...]}
  step:     {"call": name, "args": [expr], "out": [names]} | {"set": name, "expr": expr} | {"add": name, "expr": expr}
            | {"for": var, "from": expr, "to": expr, "do": [steps]} | {"if": expr, "then": [steps], "else": [steps]} | {"return": expr} | {"raise": true}
  expr:     "name" | number | {"idx": [expr, expr]} | {"dot": [expr, "field"]} | {"op": [symbol, expr, expr]} | {"neg": expr} | {"adj": expr}
            | {"vec": [expr...]} | {"call": [name, expr...]}"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "golden"))
from make_ref_caller_inventory import strip_comments  # noqa: E402

TARGETS = [  # (file under the reference root, function names traced there)
    ("src/md/AbstractMD.jl", ["U_update!", "P_update!", "P_update_fermion!"]),
    ("src/md/standardMD.jl", ["initialize_MD!", "runMD!", "runMD_QPQ!", "runMD_QPQ_sw!", "runMD_PQP!", "P_update_fermion!"]),
    ("src/updates/standardHMC.jl", ["update!"]),
]
SILENT = re.compile(r"^println")      # printing is not part of the numerical path: dropped from the trace
OPENERS = {"function", "for", "if", "while", "begin", "let", "do", "struct", "try", "quote", "macro", "module"}

TOKEN = re.compile(r"""
    (?P<nl>\n)|(?P<ws>[ \t\r]+)|
    (?P<num>\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+(?:[eE][+-]?\d+)?)|
    (?P<id>[^\W\d]\w*!?)|
    (?P<op>==|!=|>=|<=|\+=|-=|\*=|/=|<:|::|&&|\|\||[-+*/=(),\[\]{}.:'<>!;])
""", re.X | re.U)


def tokenize(text):
    toks, i = [], 0
    while i < len(text):
        if text[i] == '"':                      # a string literal (only printing and error messages hold them): one opaque token
            j = i + 1
            while j < len(text) and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            toks.append(("str", ""))
            i = j + 1
            continue
        m = TOKEN.match(text, i)
        if not m:
            raise SyntaxError("cannot tokenize at %r" % text[i:i + 30])
        i = m.end()
        kind = m.lastgroup
        if kind == "ws":
            continue
        val = m.group(kind)
        if kind == "id" and val.endswith("!") and text[i:i + 1] == "=":      # `a != b` written without spaces is not the identifier `a!`
            val = val[:-1]
            i -= 1
        toks.append((kind, val))
    return toks


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def accept(self, val):
        if self.peek()[1] == val and self.peek()[0] != "str":
            self.i += 1
            return True
        return False

    def expect(self, val):
        if not self.accept(val):
            raise SyntaxError("expected %r, found %r" % (val, self.peek()))

    def skip_nl(self):
        while self.peek()[0] == "nl" or self.peek()[1] == ";":
            self.i += 1

    # ---- expressions
    def primary(self):
        kind, val = self.next()
        if kind == "num":
            return float(val) if re.search(r"[.eE]", val) else int(val)
        if kind == "str":
            return {"str": True}
        if kind == "id":
            return val
        if val == "(":
            self.skip_nl()
            e = self.expr()
            self.skip_nl()
            self.expect(")")
            return e
        if val == "[":
            items = self.arglist("]")
            return {"vec": items}
        raise SyntaxError("unexpected token %r" % ((kind, val),))

    def arglist(self, close):
        items = []
        self.skip_nl()
        while not self.accept(close):
            items.append(self.expr())
            self.skip_nl()
            if not self.accept(","):
                self.skip_nl()
                self.expect(close)
                break
            self.skip_nl()
        return items

    def postfix(self):
        e = self.primary()
        while True:
            kind, val = self.peek()
            if val == "(" and kind == "op" and isinstance(e, str):
                self.next()
                e = {"call": [e] + self.arglist(")")}
            elif val == "[" and kind == "op":
                self.next()
                idx = self.arglist("]")
                e = {"idx": [e, idx[0]]}
            elif val == "." and kind == "op":
                self.next()
                e = {"dot": [e, self.next()[1]]}
            elif val == "'" and kind == "op":
                self.next()
                e = {"adj": e}
            else:
                return e

    def unary(self):
        if self.accept("-"):
            return {"neg": self.unary()}
        if self.accept("!"):
            return {"not": self.unary()}
        return self.postfix()

    def binary(self, level=0):
        levels = [("||",), ("&&",), ("==", "!=", ">=", "<=", "<", ">"), ("+", "-"), ("*", "/")]
        if level == len(levels):
            return self.unary()
        e = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in levels[level]:
            op = self.next()[1]
            e = {"op": [op, e, self.binary(level + 1)]}
        return e

    def expr(self):
        return self.binary()

    # ---- statements
    def block(self, stops=("end",)):
        steps = []
        while True:
            self.skip_nl()
            kind, val = self.peek()
            if kind == "eof":
                raise SyntaxError("unterminated block")
            if kind == "id" and val in stops:
                return steps
            steps.extend(self.statement())

    def statement(self):
        kind, val = self.peek()
        if kind == "id" and val == "for":
            self.next()
            var = self.next()[1]
            self.expect("=")
            lo = self.expr()
            self.expect(":")
            hi = self.expr()
            body = self.block()
            self.expect("end")
            return [{"for": var, "from": lo, "to": hi, "do": body}]
        if kind == "id" and val == "if":
            self.next()
            return [self.if_tail()]
        if kind == "id" and val == "return":
            self.next()
            if self.peek()[0] in ("nl", "eof"):
                return [{"return": None}]
            return [{"return": self.expr()}]
        first = [self.expr()]
        while self.accept(","):
            first.append(self.expr())
        if self.accept("="):
            rhs = self.expr()
            names = [t if isinstance(t, str) else None for t in first]
            if any(n is None for n in names):
                raise SyntaxError("assignment to something that is not a name: %r" % (first,))
            if isinstance(rhs, dict) and "call" in rhs:
                return self.call_step(rhs, names)
            return [{"set": names[0], "expr": rhs}] if len(names) == 1 else [{"set": names, "expr": rhs}]
        if self.peek()[1] in ("+=", "-=") and self.peek()[0] == "op":
            op = self.next()[1]
            rhs = self.expr()
            return [{"add": first[0], "expr": rhs if op == "+=" else {"neg": rhs}}]
        e = first[0]
        if isinstance(e, dict) and "call" in e:
            return self.call_step(e, [])
        raise SyntaxError("statement without effect: %r" % (e,))

    def call_step(self, call, out):
        name, args = call["call"][0], call["call"][1:]
        if SILENT.match(name):
            return []
        if name == "error":
            return [{"raise": True}]
        return [{"call": name, "args": args, "out": out}]

    def if_tail(self):
        cond = self.expr()
        then = self.block(("end", "else", "elseif"))
        other = []
        kind, val = self.next()
        if val == "elseif":
            other = [self.if_tail()]
            return {"if": cond, "then": then, "else": other}      # (the nested tail consumed the closing `end`)
        if val == "else":
            other = self.block()
            self.expect("end")
        return {"if": cond, "then": then, "else": other}


def function_spans(text):
    """(name, start of `function`, index after its closing `end`) for every top-level `function name(` of the (comment-free) text."""
    toks = [(m.start(), m.group(0)) for m in re.finditer(r'"(?:\\.|[^"\\])*"|[^\W\d]\w*!?', text, re.U)]
    spans, stack = [], []
    for k, (pos, word) in enumerate(toks):
        if word.startswith('"'):
            continue
        if word in OPENERS:
            if word == "if" and re.search(r"\S[ \t]*$", text[text.rfind("\n", 0, pos) + 1:pos]) and not re.search(r"(^|\bend|;)\s*$", text[text.rfind("\n", 0, pos) + 1:pos]):
                continue      # (a trailing `x = c if ...` form does not occur in these files; guard against `elseif` being split differently)
            name = toks[k + 1][1] if word == "function" and k + 1 < len(toks) else None
            stack.append((word, pos, name))
        elif word == "end" and stack:
            w, p, name = stack.pop()
            if w == "function" and not any(s[0] == "function" for s in stack):
                spans.append((name, p, pos + 3))
    return spans


def parse_function(src):
    """`function name(params) where {...}` + body -> (params, dispatch, steps)"""
    m = re.match(r"function\s+([^\W\d]\w*!?)\s*\(", src, re.U)
    depth, j = 0, m.end() - 1
    while True:
        depth += src[j] in "([{"
        depth -= src[j] in ")]}"
        if depth == 0:
            break
        j += 1
    params_src, rest = src[m.end():j], src[j + 1:]
    params = []
    for part in re.split(r",(?![^{]*\})", params_src):
        part = part.strip()
        if part:
            params.append(re.match(r"([^\W\d]\w*)", part, re.U).group(1))
    dispatch = {}
    wm = re.match(r"\s*where\s*\{(.*?)\}\s*(?=\n|$)", rest, re.S)
    if wm:
        for tp in re.split(r",(?![^{]*\})", wm.group(1)):
            if "<:" in tp:
                k, v = tp.split("<:")
                dispatch[k.strip()] = re.match(r"\s*([^\W\d]\w*)", v, re.U).group(1)
        rest = rest[wm.end():]
    p = Parser(tokenize(rest))
    steps = p.block()
    return params, dispatch, steps


def build(root):
    out = {"functions": []}
    for rel, names in TARGETS:
        raw = open(os.path.join(root, rel), encoding="utf-8").read()
        text = strip_comments(raw)
        for name, a, b in function_spans(text):
            if name not in names:
                continue
            params, dispatch, steps = parse_function(text[a:b])
            if steps == [{"raise": True}]:
                continue      # the abstract type's "not supported" stubs
            out["functions"].append({"name": name, "file": rel, "line": text.count("\n", 0, a) + 1, "params": params, "dispatch": dispatch, "steps": steps})
    return out


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    tree = build(root)
    print("%d functions: %s" % (len(tree["functions"]), ", ".join("%s@%s:%d" % (g["name"], os.path.basename(g["file"]), g["line"]) for g in tree["functions"])))
