# ref_parity_dump.jl OUTDIR [GOLDEN_DIR] -- operator-level golden vectors FROM THE REFERENCE ITSELF (VERDICT r02 item 2, SURVEY.md 8(c)).
#
# Runs the reference's own CPU path (Gaugefields.jl + LatticeDiracOperators.jl, the packages LatticeQCD.jl calls at
# src/system/universe.jl:41-49,103-138) on the reference's own thermalised 4^4 fixtures (tests/golden/*.ildg are byte-identical copies of
# test/confs_*/conf_00000100.ildg) with a CLOSED-FORM source -- no random numbers, so the vectors do not depend on Julia's RNG:
#     psi[c,x,y,z,t,s] = sin(0.37 + 0.11c + 0.23x + 0.31y + 0.43z + 0.59t + 0.71s) + i cos(0.19 + 0.13c + 0.29x + 0.37y + 0.41z + 0.53t + 0.61s)
# with 0-based indices (tests/ref_vectors.py: closed_form_source is the same function).  Operator parameters are exactly those Univ passes
# (universe.jl:106-116,132-137; src/system/parameter_structs.jl:122-177): kappa = 0.141139, r = 1, mass = 0.5, eps_CG = 1e-19,
# MaxCGstep = 3000, boundarycondition = [1,1,1,-1], "faster version" = true.
#
# Output (raw little-endian Float64, re/im interleaved, Julia column-major order of (c,x,y,z,t,s) == the C ABI's reference layout):
#     ref_wilson_D.bin  ref_wilson_Ddag.bin  ref_wilson_cg_x.bin        mul!(y,D,x), mul!(y,D',x), solve_DinvX!(y, DdagD, x)
#     ref_staggered_D.bin  ref_staggered_Ddag.bin  ref_staggered_cg_x.bin
#     ref_parity_meta.json                                             true residuals |b - D'D y|^2, plaquettes, package versions
# Consumers: tests/test_gpu_reference_vectors.py (HIP path), tests/test_oracle_reference_vectors.py (CPU oracle), bench.py (when its julia
# probe succeeds).  NEVER EXECUTED in the build image (no Julia there): written from the interface the reference's own callers use; any
# API mismatch makes it fail loudly and the consumers report "reference vectors absent".
using LinearAlgebra
using Gaugefields, LatticeDiracOperators
import Pkg

outdir = ARGS[1]
golden = length(ARGS) >= 2 ? ARGS[2] : joinpath(@__DIR__, "..", "tests", "golden")
mkpath(outdir)
const L = (4, 4, 4, 4)
const NC = 3

src_re(c, x, y, z, t, s) = sin(0.37 + 0.11c + 0.23x + 0.31y + 0.43z + 0.59t + 0.71s)
src_im(c, x, y, z, t, s) = cos(0.19 + 0.13c + 0.29x + 0.37y + 0.41z + 0.53t + 0.61s)

function load(fixture)
    U = Initialize_Gaugefields(NC, 0, L..., condition = "cold")
    ildg = ILDG(joinpath(golden, fixture))
    load_gaugefield!(U, 1, ildg, L, NC)              # the reader Univ uses (universe.jl:62-64)
    return U
end
function fill_source!(b, nspin)
    for s = 1:nspin, t = 1:L[4], z = 1:L[3], y = 1:L[2], x = 1:L[1], c = 1:NC
        b[c, x, y, z, t, s] = complex(src_re(c - 1, x - 1, y - 1, z - 1, t - 1, s - 1), src_im(c - 1, x - 1, y - 1, z - 1, t - 1, s - 1))
    end
    set_wing_fermion!(b)
    return b
end
function dump(path, f, nspin)
    a = Array{ComplexF64,6}(undef, NC, L..., nspin)
    for s = 1:nspin, t = 1:L[4], z = 1:L[3], y = 1:L[2], x = 1:L[1], c = 1:NC
        a[c, x, y, z, t, s] = f[c, x, y, z, t, s]
    end
    open(io -> write(io, reinterpret(Float64, vec(a))), path, "w")
end

meta = Dict{String,Any}("lattice" => collect(L), "source" => "closed form, see header",
                        "packages" => Dict(string(p.name) => string(p.version) for p in values(Pkg.dependencies()) if p.name in ("Gaugefields", "LatticeDiracOperators", "Wilsonloop")))
for (name, fixture, nspin, opname) in (("wilson", "wilson_4x4x4x4.ildg", 4, "Wilson"), ("staggered", "staggered_4x4x4x4.ildg", 1, "staggered"))
    U = load(fixture)
    x = name == "wilson" ? Initialize_pseudofermion_fields(U[1], "Wilson", nowing = true) : Initialize_pseudofermion_fields(U[1], "staggered")
    params = Dict{String,Any}("Dirac_operator" => opname, "eps_CG" => 1e-19, "verbose_level" => 1, "MaxCGstep" => 3000,
                              "boundarycondition" => [1, 1, 1, -1])
    if name == "wilson"
        params["κ"] = 0.141139; params["r"] = 1.0; params["faster version"] = true
    else
        params["mass"] = 0.5
    end
    D = Dirac_operator(U, x, params)
    A = DdagD_operator(U, x, params)
    b = fill_source!(similar(x), nspin)
    y = similar(x)
    mul!(y, D, b);  dump(joinpath(outdir, "ref_$(name)_D.bin"), y, nspin)
    mul!(y, D', b); dump(joinpath(outdir, "ref_$(name)_Ddag.bin"), y, nspin)
    clear_fermion!(y)
    solve_DinvX!(y, A, b)                            # the force path's solve (AbstractMD.jl:129 through calc_UdSfdU!)
    dump(joinpath(outdir, "ref_$(name)_cg_x.bin"), y, nspin)
    r = similar(x)
    mul!(r, A, y); add_fermion!(r, -1, b)
    meta["$(name)_cg_true_residual"] = real(dot(r, r))
    meta["$(name)_source_norm2"] = real(dot(b, b))
    temps = [similar(U[1]) for _ = 1:3]
    meta["$(name)_plaquette"] = real(calculate_Plaquette(U, temps[1], temps[2])) / (6 * prod(L) * NC)
end
open(joinpath(outdir, "ref_parity_meta.json"), "w") do io
    print(io, "{")
    print(io, join(["\"$(k)\": " * (v isa AbstractString ? "\"$(v)\"" : v isa Dict ? "{" * join(["\"$(a)\": \"$(b)\"" for (a, b) in v], ", ") * "}" : v isa Vector ? "[" * join(v, ", ") * "]" : string(v)) for (k, v) in meta], ", "))
    print(io, "}\n")
end
println("reference vectors written to $(outdir)")
