#!/usr/bin/env python3
"""Inventory of what the reference's UNCHANGED callers ask of the field / operator / action types they are handed
(SURVEY.md 8(a) a9, 8(b)): every function call inside the caller regions -- name, number of positional arguments, keyword names -- and the
declared field types of the three structs that hold those objects.  Output: tests/golden/ref_caller_inventory.json (derived data, no
source text).  tests/test_julia_binding_static.py checks julia/LatticeQCDHIP.jl against it, and re-derives it from /root/reference
whenever that tree is present (it is not on the GPU box).

usage: python tests/golden/make_ref_caller_inventory.py [reference_root]"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REGIONS = [  # (file under the reference root, first line, last line): the callers SURVEY.md 8(a) names
    ("src/md/AbstractMD.jl", 78, 135),
    ("src/md/standardMD.jl", 5, 166),
    ("src/md/standardMD.jl", 192, 227),      # P_update_fermion! with a CovNeuralnet (stout-smeared fermion action)
    ("src/updates/standardHMC.jl", 1, 91),
    ("src/system/universe.jl", 30, 191),     # ... including the construction of the stout net, :147-171
]
STRUCTS = {"src/md/standardMD.jl": ["StandardMD"], "src/updates/standardHMC.jl": ["StandardHMC"], "src/system/universe.jl": ["Univ"]}
KEYWORDS = {"if", "elseif", "for", "while", "function", "where", "return", "struct", "new", "begin", "let", "do", "in", "isa", "end"}


def strip_comments(text):
    """Remove #= ... =# blocks and # comments (not inside strings); line structure is kept."""
    out, i, n, in_str, depth = [], 0, len(text), False, 0
    while i < n:
        c = text[i]
        if depth:
            if text.startswith("=#", i):
                depth -= 1; i += 2; continue
            if text.startswith("#=", i):
                depth += 1; i += 2; continue
            out.append("\n" if c == "\n" else " "); i += 1; continue
        if in_str:
            out.append(c)
            if c == "\\":
                out.append(text[i + 1]); i += 2; continue
            if c == '"':
                in_str = False
            i += 1; continue
        if c == '"':
            in_str = True; out.append(c); i += 1; continue
        if text.startswith("#=", i):
            depth = 1; i += 2; continue
        if c == "#":
            while i < n and text[i] != "\n":
                i += 1
            continue
        out.append(c); i += 1
    return "".join(out)


def split_args(s):
    """Top-level comma split of an argument list (strings, nested brackets respected); returns (positional, keyword-names)."""
    parts, depth, cur, in_str, semi = [], 0, [], False, False
    pos, kw = [], []

    def flush():
        a = "".join(cur).strip()
        cur.clear()
        if not a:
            return
        m = re.match(r"^([A-Za-z_Ͱ-Ͽ][\wͰ-Ͽ!]*)\s*=(?!=)", a)
        if semi or m:
            kw.append(m.group(1) if m else a.rstrip(".").strip())
        else:
            pos.append(a)

    i = 0
    while i < len(s):
        c = s[i]
        if in_str:
            cur.append(c)
            if c == "\\":
                cur.append(s[i + 1]); i += 2; continue
            if c == '"':
                in_str = False
        elif c == '"':
            in_str = True; cur.append(c)
        elif c in "([{":
            depth += 1; cur.append(c)
        elif c in ")]}":
            depth -= 1; cur.append(c)
        elif c == "," and depth == 0:
            flush()
        elif c == ";" and depth == 0:
            flush(); semi = True
        else:
            cur.append(c)
        i += 1
    flush()
    return pos, kw


def find_calls(text, first, last):
    """(line, name, positional count, keyword names) of every call `name(...)` whose opening parenthesis lies in [first, last]."""
    calls = []
    line_of = [0] * (len(text) + 1)
    ln = 1
    for i, c in enumerate(text):
        line_of[i] = ln
        if c == "\n":
            ln += 1
    in_str = False
    i = 0
    ident = re.compile(r"[A-Za-z_Ͱ-Ͽ][\wͰ-Ͽ!]*$")
    while i < len(text):
        c = text[i]
        if in_str:
            if c == "\\":
                i += 2; continue
            if c == '"':
                in_str = False
            i += 1; continue
        if c == '"':
            in_str = True; i += 1; continue
        if c == "(" and first <= line_of[i] <= last:
            m = ident.search(text[max(0, i - 80):i])
            if m:
                name = m.group(0)
                start = i - len(name)
                prev = text[start - 1] if start > 0 else " "
                before = text[max(0, start - 12):start]
                is_def = re.search(r"function\s+$", before) is not None
                if name not in KEYWORDS and prev not in ".@:" and not is_def:
                    depth, j, s2 = 0, i, False
                    while j < len(text):
                        d = text[j]
                        if s2:
                            if d == "\\":
                                j += 2; continue
                            if d == '"':
                                s2 = False
                        elif d == '"':
                            s2 = True
                        elif d in "([{":
                            depth += 1
                        elif d in ")]}":
                            depth -= 1
                            if depth == 0:
                                break
                        j += 1
                    pos, kw = split_args(text[i + 1:j])
                    calls.append({"line": line_of[i], "name": name, "nargs": len(pos), "kwargs": sorted(kw)})
        i += 1
    return calls


def struct_fields(text, name):
    m = re.search(r"\bstruct\s+" + name + r"\b[^\n]*\n(.*?)\n\s*(?:function|end)\b", text, re.S)
    fields = {}
    if m:
        for line in m.group(1).splitlines():
            fm = re.match(r"^\s*([A-Za-z_Ͱ-Ͽ][\wͰ-Ͽ]*)::(.+?)\s*$", line)
            if fm:
                fields[fm.group(1)] = fm.group(2).replace(" ", "")
    return fields


def own_functions(root):
    """name -> list of (min, max) positional arities the reference defines itself anywhere under src/ WITHOUT a module prefix
    (`function Gaugefields.println_verbose_level1(univ, ...)` extends the package's generic and does not count): a call to one of these
    is the reference calling itself, not the packages."""
    own = {}
    for d, _, files in os.walk(os.path.join(root, "src")):
        for f in files:
            if not f.endswith(".jl"):
                continue
            t = strip_comments(open(os.path.join(d, f), encoding="utf-8").read())
            for m in re.finditer(r"\bfunction\s+([A-Za-z_Ͱ-Ͽ][\wͰ-Ͽ!]*)\s*\(", t):
                depth, j = 0, m.end() - 1
                while j < len(t):
                    if t[j] in "([{":
                        depth += 1
                    elif t[j] in ")]}":
                        depth -= 1
                        if depth == 0:
                            break
                    j += 1
                pos, _ = split_args(t[m.end():j])
                lo = sum(1 for a in pos if "..." not in a)
                hi = 99 if any("..." in a for a in pos) else len(pos)
                # arguments with defaults were classified as keywords by split_args (name = value): they widen the range
                _, kw = split_args(t[m.end():j].split(";")[0])
                own.setdefault(m.group(1), []).append((lo, hi + len(kw)))
            for m in re.finditer(r"\bstruct\s+([A-Za-z_]\w*)", t):
                own.setdefault(m.group(1), []).append((0, 99))
    return own


def build(root):
    inv = {"regions": [list(r) for r in REGIONS], "calls": [], "struct_fields": {}}
    own = own_functions(root)
    for rel, first, last in REGIONS:
        text = strip_comments(open(os.path.join(root, rel), encoding="utf-8").read())
        for c in find_calls(text, first, last):
            c["file"] = rel
            c["defined_by_reference"] = any(lo <= c["nargs"] <= hi for lo, hi in own.get(c["name"], []))
            inv["calls"].append(c)
        for s in STRUCTS.get(rel, []):
            inv["struct_fields"][s] = struct_fields(text, s)
    return inv


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    inv = build(root)
    with open(os.path.join(HERE, "ref_caller_inventory.json"), "w") as f:
        json.dump(inv, f, indent=1, ensure_ascii=False, sort_keys=True)
        f.write("\n")
    print("%d calls, structs %s" % (len(inv["calls"]), {k: len(v) for k, v in inv["struct_fields"].items()}))
