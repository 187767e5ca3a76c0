# ref_cpu_baseline.jl NX NY NZ NT kappa -- times the REFERENCE's own CPU path (Gaugefields.jl + LatticeDiracOperators.jl, the packages
# LatticeQCD.jl calls at src/system/universe.jl:41-49,103-137) on a hot-start lattice: one mul!(y, D, x) and one CG iteration on D'D
# (mul!(q, DdagD, p) + the BLAS-1 of the iteration).  bench.py runs it only when `julia` is on PATH and falls back to the oracle port when
# it exits non-zero (packages missing -- there is no network).  NEVER EXECUTED in the build image (no Julia there): written from the
# packages' documented interface; any API mismatch makes it fail loudly and the fall-back is taken.
using LinearAlgebra, Random
using Gaugefields, LatticeDiracOperators
NX, NY, NZ, NT = parse.(Int, ARGS[1:4])
κ = parse(Float64, ARGS[5])
Random.seed!(111)
U = Initialize_Gaugefields(3, 0, NX, NY, NZ, NT, condition = "hot")
x = Initialize_pseudofermion_fields(U[1], "Wilson", nowing = true)
params = Dict{String,Any}("Dirac_operator" => "Wilson", "κ" => κ, "r" => 1.0, "faster version" => true, "eps_CG" => 1e-19,
                          "verbose_level" => 1, "MaxCGstep" => 3000, "boundarycondition" => [1, 1, 1, -1])
D = Dirac_operator(U, x, params)
A = DdagD_operator(U, x, params)
b = similar(x); gauss_distribution_fermion!(b)
y = similar(x); p = similar(x); q = similar(x); r = similar(x)
substitute_fermion!(p, b); substitute_fermion!(r, b)
mul!(y, D, b)                                   # warm (compilation)
mul!(q, A, p)
t_d = @elapsed mul!(y, D, b)
t_it = @elapsed begin                            # one CG iteration on D'D, zero initial guess
    mul!(q, A, p)
    α = real(dot(r, r)) / real(dot(p, q))
    add_fermion!(y, α, p)
    add_fermion!(r, -α, q)
    β = real(dot(r, r))
    add_fermion!(p, β - 1, p); add_fermion!(p, 1, r)
end
V = NX * NY * NZ * NT
println("{\"cg_iter_per_s\": $(1 / t_it), \"dslash_gflops\": $(1320 * V / t_d / 1e9), \"sample\": \"1 mul!(y,D,x) and 1 CG iteration on D'D, $(NX)x$(NY)x$(NZ)x$(NT), 1 thread\"}")
