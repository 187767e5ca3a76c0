"""Static check of julia/LatticeQCDHIP.jl against the reference's UNCHANGED callers.

There is no Julia in the image, so the binding cannot run.  What can be established without running it:

  1. tests/golden/ref_caller_inventory.json lists every call the caller regions make (AbstractMD.jl:78-135, standardMD.jl:5-166,
     standardHMC.jl:1-91, universe.jl:30-143: name, positional arity, keyword names) and the declared field types of the structs that
     hold the fields (StandardMD, StandardHMC, Univ).  It is derived data; when /root/reference is present the test re-derives it and
     demands equality, so the fixture cannot go stale silently.
  2. Every call falls into exactly one class: Julia Base, the reference calling itself, a package generic that works on ANY
     AbstractGaugefields subtype through similar()/the type hierarchy (PACKAGE_GENERIC, each with the precondition it needs from
     the binding), or a generic the binding must specialise -- for those the binding must have a method of that name, importable
     into the package's function (listed in an `import Gaugefields: ...` / `import LatticeDiracOperators: ...` / LinearAlgebra / Base
     line), that accepts the positional arity and every keyword of the call, with the argument types EXPECTED below.
  3. The struct-field constraints are satisfiable: `gauge_action::GaugeAction{Dim,TG}` (the package's own type, so the binding may not
     bring a private action struct and must specialise on GaugeAction{4,HIPLink}), `p::Vector{TA}`, `Uold::Vector{TG}`,
     `U::Vector{TG}`, `dSdU::Union{Nothing,Vector{TG}}` -- the constructors involved return exactly those containers (declared return types)."""
import importlib.util
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINDING = os.path.join(ROOT, "julia", "LatticeQCDHIP.jl")
INVENTORY = os.path.join(ROOT, "tests", "golden", "ref_caller_inventory.json")
REFERENCE = "/root/reference"

spec = importlib.util.spec_from_file_location("make_ref_caller_inventory", os.path.join(ROOT, "tests", "golden", "make_ref_caller_inventory.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)

BASE = {"Dict", "Tuple", "close", "eltype", "error", "exp", "length", "pwd", "rand", "real", "typeof", "append!"}
# provided by the packages for any field type that satisfies the stated precondition (checked below)
PACKAGE_GENERIC = {
    "GaugeAction": "Gaugefields' own constructor GaugeAction(U::Vector{<:AbstractGaugefields{NC,Dim}}): needs HIPLink <: AbstractGaugefields{3,4} and similar(::HIPLink)",
    "push!": "push!(gauge_action, beta, loops) on the package's GaugeAction{4,HIPLink}; push!(cov_neural_net, layer) on the package's CovNeuralnet with the binding's HIPStoutLayer <: CovLayer{4}",
    "CovNeuralnet": "Gaugefields' own constructor CovNeuralnet(U): the container Univ's typed field holds (universe.jl:15); the layers in it are the binding's (STOUT_Layer dispatches on the links)",
    "make_loops_fromname": "Wilsonloop.jl, no field argument",
    "get_temporary_gaugefields": "accessor of the package's GaugeAction",
    "get_temp": "Temporalfields pool of the package's GaugeAction: allocates with similar(U[1])",
    "unused!": "Temporalfields pool",
    "Verbose_print": "Gaugefields' logger, no field argument",
    "ILDG": "Gaugefields' reader object, no field argument",
    "loadU": "JLD loader: replaces U by a host field -- not available on the HIP path (documented in INTEGRATION.md)",
    "Initialize_Gaugefields": "no argument to dispatch on: ONE edit in Univ (Initialize_HIPGaugefields) or none -- activate!() adds a more specific method of the package's function (checked below)",
}
# generics the binding specialises: name -> list of accepted signatures (positional argument types as written in the binding)
EXPECTED = {
    "exptU!": [["HIPLink", "Number", "HIPTALink", None]],
    "mul!": [["HIPLink", "HIPLink", "HIPLink"], ["HIPLink", "HIPLinkAdjoint", "HIPLink"]],
    "back_prop": [["Vector{HIPLink}", "CovNeuralnet{4}", None, "Vector{HIPLink}"]],
    "STOUT_Layer": [[None, None, "Vector{HIPLink}"]],
    "substitute_U!": [["HIPLink", "HIPLink"], ["Vector{HIPLink}", "Vector{HIPLink}"]],
    "calc_dSdUμ!": [["HIPLink", "GaugeAction{4,HIPLink}", "Integer", "Vector{HIPLink}"]],
    "Traceless_antihermitian_add!": [["HIPTALink", "Number", "HIPLink"]],
    "calc_UdSfdU!": [["Vector{HIPLink}", "HIPFermiAction", "Vector{HIPLink}", "HIPFermion"]],
    "initialize_TA_Gaugefields": [["Vector{HIPLink}"]],
    "similar": [["Vector{HIPLink}"], ["HIPLink"], ["HIPFermion"]],
    "gauss_distribution!": [["Vector{HIPTALink}"]],
    "calc_smearedU": [["Vector{HIPLink}", "Nothing"], ["Vector{HIPLink}", "CovNeuralnet{4}"]],
    "gauss_sampling_in_action!": [["HIPFermion", "Vector{HIPLink}", "HIPFermiAction"]],
    "sample_pseudofermions!": [["HIPFermion", "Vector{HIPLink}", "HIPFermiAction", "HIPFermion"]],
    "evaluate_GaugeAction": [["GaugeAction{4,HIPLink}", "Vector{HIPLink}"]],
    "evaluate_FermiAction": [["HIPFermiAction", "Vector{HIPLink}", "HIPFermion"]],
    "dot": [["HIPFermion", "HIPFermion"]],
    "println_verbose_level1": [["AnyLink"]], "println_verbose_level2": [["AnyLink"]], "println_verbose_level3": [["AnyLink"]],
    "get_myrank": [["AnyLink"]],
    "Initialize_pseudofermion_fields": [["HIPLink", "String"]],
    "Dirac_operator": [["Vector{HIPLink}", "HIPFermion", None]],
    "FermiAction": [["HIPDirac", None]],
    "load_BridgeText!": [["String", "Vector{HIPLink}", None, None]],
    "load_gaugefield!": [["Vector{HIPLink}", None, None, None, None]],
}
# calls that belong to paths the HIP binding does not serve (and says so) -- none since round 4: the Domainwall operator of universe.jl:116-128 is
# served (csrc/domainwall.hip; Initialize_pseudofermion_fields(U[1], "Domainwall", L5 = L5, nowing = true) resolves like the others)
NOT_SERVED = set()


def binding_text():
    return gen.strip_comments(open(BINDING, encoding="utf-8").read())


def binding_methods(text):
    """name -> list of dict(pos=[(name, type|None, has_default)], varargs, kwargs=[...]|'any', ret) for every method definition."""
    methods = {}
    pat = re.compile(r"(?m)^(?:function\s+)?((?:[A-Za-z_][\w]*\.)*(?::\*|[A-Za-z_Ͱ-Ͽ][\wͰ-Ͽ!]*))\(")
    for m in pat.finditer(text):
        line_start = text.rfind("\n", 0, m.start()) + 1
        if text[line_start:m.start()].strip() not in ("", "function"):
            continue
        depth, j = 0, m.end() - 1
        while j < len(text):
            if text[j] in "([{":
                depth += 1
            elif text[j] in ")]}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        rest = text[j + 1:j + 60]
        is_function = text[m.start():m.start() + 9].startswith("function")
        if not is_function and not re.match(r"\s*(::[^=\n]+?)?\s*(where\s*\{[^}]*\}\s*)?=(?!=)", rest):
            continue                                    # a call at the start of a line, not a definition
        args = text[m.end():j]
        head, _, tail = args.partition(";")
        pos = []
        varargs = False
        raw, _ = gen.split_args(head.replace("=", "\x00"))          # keep defaults inside the positional items
        for a in raw:
            a = a.replace("\x00", "=")
            default = "=" in a
            a = a.split("=")[0].strip()
            if a.endswith("..."):
                varargs = True
                continue
            nm, _, ty = a.partition("::")
            pos.append((nm.strip(), ty.strip() or None, default))
        kwargs = "any" if "..." in tail else [k.split("\x00")[0].split("::")[0].strip() for k in gen.split_args(tail.replace("=", "\x00"))[0] if k.strip()]
        ret = re.match(r"\s*::([^=\n]+?)\s*(?:where|=|\n)", rest)
        name = m.group(1).split(".")[-1].lstrip(":")
        methods.setdefault(name, []).append({"pos": pos, "varargs": varargs, "kwargs": kwargs, "ret": ret.group(1).strip() if ret else None})
    return methods


def imported_names(text):
    names = {}
    for m in re.finditer(r"(?m)^import\s+([\w.]+)\s*:\s*((?:[^\n]|\n\s{4,})+)", text):
        for n in re.split(r"[,\s]+", m.group(2).strip()):
            if n:
                names[n] = m.group(1)
    return names


def accepts(method, nargs, kwargs):
    required = sum(1 for _, _, d in method["pos"] if not d)
    if not (required <= nargs <= len(method["pos"]) or (method["varargs"] and nargs >= required)):
        return False
    return method["kwargs"] == "any" or all(k in method["kwargs"] for k in kwargs)


def test_inventory_is_current():
    inv = json.load(open(INVENTORY, encoding="utf-8"))
    assert len(inv["calls"]) > 100 and set(inv["struct_fields"]) == {"StandardMD", "StandardHMC", "Univ"}
    if not os.path.isdir(REFERENCE):
        pytest.skip("/root/reference is not present here: the committed inventory is used as it is")
    assert gen.build(REFERENCE) == inv, "tests/golden/ref_caller_inventory.json is stale: run tests/golden/make_ref_caller_inventory.py"


def test_every_generic_the_unchanged_callers_use_has_a_method_in_the_binding():
    inv = json.load(open(INVENTORY, encoding="utf-8"))
    text = binding_text()
    methods = binding_methods(text)
    imports = imported_names(text)
    checked = set()
    for c in inv["calls"]:
        name, nargs, kwargs = c["name"], c["nargs"], tuple(c["kwargs"])
        where = "%s:%d %s/%d" % (c["file"], c["line"], name, nargs)
        if c["defined_by_reference"] or name in BASE or (name, kwargs) in NOT_SERVED:
            continue
        if name in PACKAGE_GENERIC:
            continue
        assert name in EXPECTED, where + ": a package generic applied to the fields that this test does not know -- classify it"
        assert name in methods, where + ": the binding has no method of that name"
        assert name in imports or name in ("similar", "dot", "mul!"), where + ": the binding does not import the package's function, the callers would not reach its methods"
        ok = [m for m in methods[name] if accepts(m, nargs, kwargs)]
        assert ok, where + ": no method of the binding accepts %d positional arguments and keywords %s" % (nargs, kwargs)
        # one of the accepting methods has exactly the expected argument types
        sigs = [[t for _, t, _ in m["pos"]][:nargs] for m in ok]
        assert any(all(e is None or e == t for e, t in zip(exp, sig)) and len(exp) >= len(sig) for exp in EXPECTED[name] for sig in sigs), \
            where + ": argument types %s, expected one of %s" % (sigs, EXPECTED[name])
        checked.add(name)
    assert checked >= set(EXPECTED) - {"similar", "substitute_U!"} | {"similar", "substitute_U!"}, sorted(set(EXPECTED) - checked)


def test_struct_field_constraints_of_the_callers_are_satisfiable():
    inv = json.load(open(INVENTORY, encoding="utf-8"))
    f = inv["struct_fields"]
    text = binding_text()
    methods = binding_methods(text)
    # the callers' declarations this test was written against (a change upstream must be looked at, not silently accepted)
    assert f["StandardMD"]["gauge_action"] == "GaugeAction{Dim,TG}" and f["Univ"]["gauge_action"] == "GaugeAction{Dim,TG}"
    assert f["StandardMD"]["p"] == "Vector{TA}" and f["StandardMD"]["dSdU"] == "Union{Nothing,Vector{TG}}"
    assert f["StandardHMC"]["Uold"] == "Vector{TG}" and f["Univ"]["U"] == "Vector{TG}"
    # U::Vector{TG}, TG = eltype(U): the links are a real Vector of a subtype of the package's abstract field type
    assert re.search(r"struct\s+HIPLink\s*<:\s*AbstractGaugefields\{3,\s*4\}", text), "HIPLink must be an AbstractGaugefields{3,4}"
    init = methods["Initialize_HIPGaugefields"][0]
    assert init["ret"] == "Vector{HIPLink}" and init["varargs"] and "condition" in init["kwargs"]
    # Uold = similar(U); TG = eltype(Uold); StandardHMC{Tmd,TG}(md, Uold)  -- and dSdU = similar(U) when a smearing net is present
    sim = [m for m in methods["similar"] if m["pos"][0][1] == "Vector{HIPLink}"]
    assert sim and sim[0]["ret"] == "Vector{HIPLink}"
    # temporaries of the package's own GaugeAction / Temporalfields: similar(U[1]) is one link
    sim1 = [m for m in methods["similar"] if m["pos"][0][1] == "HIPLink"]
    assert sim1 and sim1[0]["ret"] == "HIPLink"
    # p::Vector{TA}, TA = eltype(p)
    ta = methods["initialize_TA_Gaugefields"][0]
    assert ta["ret"] == "Vector{HIPTALink}" and re.search(r"struct\s+HIPTALink\b", text)
    assert any([t for _, t, _ in m["pos"]] == ["Vector{HIPTALink}", "Vector{HIPTALink}"] for m in methods["*"]), "md.p * md.p (standardHMC.jl:49)"
    # gauge_action::GaugeAction{Dim,TG}: the package's type with TG = HIPLink -- no private action struct, arithmetic specialised on it
    assert not re.search(r"struct\s+HIPGaugeAction\b", text), "the callers hold a Gaugefields.GaugeAction{Dim,TG}; a private action type cannot be stored"
    imports = imported_names(text)
    assert imports.get("GaugeAction") == "Gaugefields" and imports.get("AbstractGaugefields") == "Gaugefields"
    for name in ("calc_dSdUμ!", "evaluate_GaugeAction"):
        assert any("GaugeAction{4,HIPLink}" in [t for _, t, _ in m["pos"]] for m in methods[name]), name
    # U[1].NC (AbstractMD.jl:100, standardHMC.jl:42)
    assert re.search(r"Base\.getproperty\(l::AnyLink, s::Symbol\)", text) and ":NC" in text
    # fermi_action._temporary_fermionfields[1] (standardMD.jl:50)
    assert re.search(r"_temporary_fermionfields::Vector\{HIPFermion\}", text)
    # md.cov_neural_net = nothing: update! still calls calc_smearedU(U, nothing) (standardHMC.jl:67 compares a value with the type Nothing)
    assert any([t for _, t, _ in m["pos"]] == ["Vector{HIPLink}", "Nothing"] for m in methods["calc_smearedU"])
    # the whole-field entry points insist on four views of one storage, in order
    assert "function whole(U::Vector{<:AnyLink})" in text


def test_binding_has_no_leftovers_of_the_container_type_the_callers_cannot_hold():
    text = binding_text()
    assert not re.search(r"\bHIPGaugefields\b", text), "round 2's mutable struct HIPGaugefields could not be stored in U::Vector{TG} / Uold::Vector{TG}"
    assert "HIPTemporalfields" not in text, "the package's own Temporalfields pool serves get_temp / unused! (it only needs similar(::HIPLink))"


def test_fermi_action_of_any_nf_goes_through_the_library_handle():
    """universe.jl:106-110 puts p.Nf into the Dict and :138 calls FermiAction(D, parameters_action); the reference's own test_Nf2.toml:8 /
    test_Nf3.toml:8 (runtests.jl:114-130) pass Nf = 2, 3 with the staggered operator.  The binding must not reject any Nf itself: the decision
    exact / rational and the coefficients belong to lqcd_action_create; the four generics of the action dispatch to the handle."""
    text = binding_text()
    methods = binding_methods(text)
    fa = [m for m in methods["FermiAction"] if [t for _, t, _ in m["pos"]][:1] == ["HIPDirac"]]
    assert len(fa) == 1
    start = text.index("function FermiAction(D::HIPDirac")
    body = text[start:text.index("\nend", start)]
    assert "lqcd_action_create" in body and "error(" not in body and '"Nf"' in body
    for generic, export in (("gauss_sampling_in_action!", "lqcd_action_gauss_sampling"), ("sample_pseudofermions!", "lqcd_action_sample_pseudofermions"),
                            ("evaluate_FermiAction", "lqcd_action_evaluate"), ("calc_UdSfdU!", "lqcd_action_force")):
        i = text.index(generic + "(")
        while "HIPFermiAction" not in text[i:text.index("\n", i)]:
            i = text.index(generic + "(", i + 1)
        assert export in text[i:i + 900], generic
    # no module-level rational-coefficient plumbing is left for the caller to do
    assert "needs the rational action" not in text


def test_domainwall_operator_of_universe_jl_resolves():
    """universe.jl:116-128: x = Initialize_pseudofermion_fields(U[1], "Domainwall", L5 = L5, nowing = true); params "Dirac_operator" => "Domainwall", "mass", "L5", "M";
    D = Dirac_operator(U, x, params); FermiAction(D, parameters_action) (test/test_domainwallhmc.toml, runtests.jl:132-137).  The binding makes a
    five-dimensional field, the operator through lqcd_op_create_domainwall, and the action through the same handle as every other operator."""
    text = binding_text()
    start = text.index("function Initialize_pseudofermion_fields(u::HIPLink")
    body = text[start:text.index("\nend", start)]
    assert '"domainwall"' in body and "DOMAINWALL" in body and "L5" in body
    start = text.index("function Dirac_operator(U::Vector{HIPLink}")
    body = text[start:text.index("\nend", start)]
    assert 'name == "Domainwall"' in body and "lqcd_op_create_domainwall" in body
    for key in ('"M"', '"mass"', '"L5"', '"boundarycondition"', '"eps_CG"', '"MaxCGstep"'):
        assert key in body[:body.index("return D")], key
    assert "lqcd_spinor_create_5d" in text and "lqcd_spinor_slice" in text
    # similar(eta) of a five-dimensional field keeps L5 (standardMD.jl:50-51: eta = similar(fermi_action._temporary_fermionfields[1]); xi = similar(eta))
    assert re.search(r"similar\(x::HIPFermion\)\s*=\s*HIPFermion\(x\.lat, x\.kind; L5 = x\.L5\)", text)


def test_binding_keeps_no_module_level_mutable_state():
    """"no global mutable state outside the context handle".  Module-level `const X = Ref(...)`, `Dict(...)`, `Any[]` ... would be
    shared by every lattice of the process (round 3's lazy-link record was); the lazy fusion now lives in the library's context (csrc/md.hip) and
    every link generic of the binding is one ccall."""
    text = binding_text()
    for m in re.finditer(r"(?m)^const\s+(\w+)\s*=\s*(.+)$", text):
        name, rhs = m.group(1), m.group(2)
        assert not re.search(r"\bRef\b|\bDict\b|\[\s*\]|Vector\{|Set\(", rhs), "module-level mutable state: const %s = %s" % (name, rhs)
    assert not re.search(r"(?m)^(global\s+)?\w+\s*=\s*(Ref|Dict|Any\[)", text)
    for generic, export in (("exptU!", "lqcd_link_exp"), ("mul!(C::HIPLink", "lqcd_link_mul"), ("substitute_U!(dst::HIPLink", "lqcd_link_copy"),
                            ("calc_dSdUμ!", "lqcd_link_staple"), ("Traceless_antihermitian_add!", "lqcd_link_add_ta")):
        i = text.index(generic)
        while "ccall" not in text[i:i + 400]:
            i = text.index(generic, i + 1)
        end = re.search(r"\n(?=\S)", text[i:])             # the definition ends where the next unindented line begins (comments are stripped)
        body = text[i:i + end.start()]
        if body.startswith("function") or text[text.rfind("\n", 0, i) + 1:i].startswith("function"):
            body = text[i:text.index("\nend", i)]
        assert body.count("ccall") == 1 and export + "," in body, generic


def test_univ_runs_unchanged_after_activate():
    """universe.jl:41-49 calls Initialize_Gaugefields(NC, Nwing, L...; condition = ...) -- nothing to dispatch on.  activate!() defines a method of
    the package's OWN function for (Int, Int, four Int extents; condition) that returns device links and hands every other call on with invoke; the
    package's method is never replaced and deactivate!() deletes the redirection.  Here: the four call sites of the inventory use exactly the one
    keyword the method names, the method is defined INTO Gaugefields, returns Initialize_HIPGaugefields' value for NC = 3, and falls through otherwise."""
    inv = json.load(open(INVENTORY, encoding="utf-8"))
    sites = [c for c in inv["calls"] if c["name"] == "Initialize_Gaugefields"]
    assert len(sites) == 4 and all(c["file"] == "src/system/universe.jl" and c["kwargs"] == ["condition"] and c["nargs"] == 3 for c in sites)   # NC, Nwing, L...
    text = binding_text()
    i = text.index("function activate!()")
    body = text[i:text.index("function deactivate!()")]
    assert re.search(r"@eval function Gaugefields\.Initialize_Gaugefields\(NC::Int, Nwing::Int, NX::Int, NY::Int, NZ::Int, NT::Int; condition = \"cold\", kwargs\.\.\.\)", body)
    assert "return Initialize_HIPGaugefields(NC, Nwing, NX, NY, NZ, NT; condition = condition)" in body
    assert "invoke(Gaugefields.Initialize_Gaugefields, Tuple{Any,Any,Vararg{Any}}" in body            # everything else: the package's own method
    assert "Base.delete_method" in text[text.index("function deactivate!()"):text.index("function deactivate!()") + 400]
    assert imported_names(text).get("Initialize_Gaugefields") == "Gaugefields"


def test_binding_blocks_are_balanced():
    """No Julia here to parse the file: at least every block opener (function, if, for, while, struct, let, begin, do, try, module) must meet its `end`.  Comments and
    strings are stripped; `for` / `if` inside open brackets or parentheses of the same line are comprehensions / generators; `a[end]` and `:end` are not block ends."""
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', binding_text())
    openers = re.compile(r"(?<![\w!.])(function|if|for|while|struct|let|begin|do|try|module|quote|macro)(?![\w!])")
    stack = []
    for ln, line in enumerate(src.split("\n"), 1):
        toks = [(m.start(), m.group(1)) for m in openers.finditer(line)] + [(m.start(), "end") for m in re.finditer(r"(?<![\w!.:\[])end(?![\w!])", line)]
        for pos, t in sorted(toks):
            if t == "end":
                assert stack, "line %d: `end` without an open block" % ln
                stack.pop()
            else:
                head = line[:pos]
                inside = head.count("[") > head.count("]") or head.count("(") > head.count(")")
                if t in ("for", "if") and inside:
                    continue
                stack.append((ln, t))
    assert not stack, "unclosed blocks: %s" % stack[:5]


def test_binding_calls_only_names_it_defines_imports_or_base_provides():
    """A typo in a function name would only show at run time in Julia: every `name(` of the binding must be a method it defines, a name it imports from the two
    packages / Base / LinearAlgebra, a struct of its own, or one of the Base functions listed here."""
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', binding_text())
    calls = set(re.findall(r"(?<![\w.:@])([A-Za-z_][\w!]*)\(", src))
    defs = set(re.findall(r"(?m)^\s*(?:function\s+)?(?:[A-Za-z_]\w*\.)*([A-Za-z_][\w!]*)\(", src))
    defs |= set(re.findall(r"(?m)^(?:mutable\s+)?struct\s+([A-Za-z_]\w*)", src))
    base = {"ccall", "error", "length", "push!", "Ref", "Cint", "Float64", "Int", "String", "isempty", "get", "haskey", "zeros", "complex", "real", "imag",
            "finalizer", "println", "lowercase", "rand", "UInt64", "sum", "all", "enumerate", "foreach", "invoke", "which", "getfield", "unsafe_string", "joinpath",
            "cat", "size", "Tuple", "T", "where", "prod"}
    unknown = sorted(c for c in calls if c not in defs and c not in imported_names(src) and c not in base)
    assert not unknown, unknown
