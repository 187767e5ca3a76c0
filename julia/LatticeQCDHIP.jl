# LatticeQCDHIP.jl -- reference-side binding of liblqcd_hip.so (C ABI: include/lqcd_hip.h).
#
# STATUS: written against the documented LatticeDiracOperators.jl 0.6 / Gaugefields.jl 0.7 interface but NEVER EXECUTED:
# there is no Julia in the build image (SURVEY.md section 0.3).  Everything below the `ccall`s is exercised through the
# same C ABI by tests/ (Python ctypes).  The file is deliberately thin: every method is one ccall plus error translation.
#
# What it provides: device-backed field / operator / action types that dispatch EVERY generic the reference's unchanged callers use
#   src/system/universe.jl:88-138    GaugeAction(U), push!(gauge_action, beta/2, loops), Initialize_pseudofermion_fields(U[1], ...),
#                                    Dirac_operator(U, x, params), FermiAction(D, Dict)
#   src/md/AbstractMD.jl:78-135      get_temporary_gaugefields, get_temp, unused!, exptU!, mul!(W, expU, U[mu]), substitute_U!(U[mu], W),
#                                    calc_dSdUμ!, Traceless_antihermitian_add!(p[mu], ...), calc_UdSfdU!(UdSfdUμ::Vector, fa, U, η), U[1].NC
#   src/md/standardMD.jl:34-101      initialize_TA_Gaugefields(U), fermi_action._temporary_fermionfields[1], similar,
#                                    gauss_distribution!(md.p), gauss_sampling_in_action!, sample_pseudofermions!
#   src/updates/standardHMC.jl:41-91 similar(U), substitute_U!(Uold, U), md.p * md.p, evaluate_GaugeAction, dot(ξ, ξ), evaluate_FermiAction(fa, U, η)
# and, below them, mul!(y, D, x), mul!(y, D', x), solve_DinvX!, shiftedcg, dot, clear_fermion!, add_fermion!, ... (SURVEY.md 8(a), 8(b)).
# The per-direction objects U[mu], p[mu] and the temporaries are VIEWS (field, direction slot) into four-direction device fields;
# tests/test_gpu_reference_callers.py runs the same callers, transliterated line by line, through the same C entry points.
module LatticeQCDHIP

using LinearAlgebra
import LinearAlgebra: mul!, dot
import Base: similar, adjoint, getindex, length, push!

const LIB = get(ENV, "LQCD_HIP_LIB", joinpath(@__DIR__, "..", "latticeqcd.jl_amd", "csrc", "liblqcd_hip.so"))

const LQCD_OK = Cint(0)
const LQCD_ERR_NOT_CONVERGED = Cint(3)
const WILSON, STAGGERED = Cint(0), Cint(1)
const FULL, EVEN, ODD = Cint(0), Cint(1), Cint(2)
const LAYOUT_REFERENCE = Cint(0)

last_error() = unsafe_string(ccall((:lqcd_last_error, LIB), Cstring, ()))
function check(st::Cint)
    st == LQCD_OK && return nothing
    error(last_error())            # the reference raises error(...) on non-convergence / unsupported operators
end

# ------------------------------------------------------------------ context (one per process / GPU; PEs = PE grid of src/mpirun.jl:17-19)
mutable struct HIPLattice
    h::Ptr{Cvoid}
    L::NTuple{4,Int}
    PEs::NTuple{4,Int}
    rank::Int
end
function HIPLattice(L::NTuple{4,Int}; PEs = (1, 1, 1, 1), rank = 0, device = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_ctx_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cint}, Ptr{Cint}, Cint),
                h, device, Cint[L...], Cint[PEs...], rank))
    lat = HIPLattice(h[], L, PEs, rank)
    finalizer(l -> ccall((:lqcd_ctx_destroy, LIB), Cint, (Ptr{Cvoid},), l.h), lat)
    return lat
end
# RCCL bootstrap: rank 0 creates the id blob, MPI.Bcast distributes the 256 bytes (replaces src/mpi/mpimodule.jl:4-13)
function comm_unique_id()
    id = zeros(UInt8, 256)
    check(ccall((:lqcd_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    return id
end
comm_init!(lat::HIPLattice, id::Vector{UInt8}) =
    check(ccall((:lqcd_ctx_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), lat.h, id, prod(lat.PEs)))

# ------------------------------------------------------------------ gauge fields: the reference's U::Vector{TG} (U[1:4]) as ONE device object,
# U[mu] / p[mu] / temporaries as views (field, direction slot)
mutable struct HIPGaugefields   # stands where the reference holds Vector{<:Gaugefields.AbstractGaugefields{3,4}}
    h::Ptr{Cvoid}
    lat::HIPLattice
    NC::Int
end
function HIPGaugefields(lat::HIPLattice)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_gauge_create, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), lat.h, h))
    g = HIPGaugefields(h[], lat, 3)
    finalizer(x -> ccall((:lqcd_gauge_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), g)
    return g
end
struct HIPLink                  # in the reference tree: <: Gaugefields.AbstractGaugefields{3,4}; one direction of a HIPGaugefields
    parent::HIPGaugefields
    slot::Cint                  # 0..3
end
Base.getproperty(l::HIPLink, s::Symbol) = s === :NC ? getfield(l, :parent).NC : getfield(l, s)      # U[1].NC (AbstractMD.jl:101)
Base.getindex(U::HIPGaugefields, mu::Integer) = (1 <= mu <= 4 || throw(BoundsError(U, mu)); HIPLink(U, Cint(mu - 1)))
Base.length(::HIPGaugefields) = 4
Base.eltype(::Type{HIPGaugefields}) = HIPLink
similar(U::HIPGaugefields) = HIPGaugefields(U.lat)                                                   # Uold = similar(U) (standardHMC.jl:32)
Base.size(l::HIPLink) = (l.parent.NC, l.parent.NC, l.parent.lat.L...)

# upload: U is the reference's Vector of 4 Array{ComplexF64,6} (NC,NC,NX,NY,NZ,NT), Nwing = 0
function substitute_U!(g::HIPGaugefields, U::Vector{<:AbstractArray{ComplexF64,6}}; Nwing = 0)
    buf = cat(U...; dims = 7)      # [a,b,x,y,z,t,mu] column-major == lqcd LAYOUT_REFERENCE (extents L .+ 2Nwing when the fields carry wings)
    if Nwing == 0
        check(ccall((:lqcd_gauge_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), g.h, buf, LAYOUT_REFERENCE))
    else    # the reference's Initialize_Gaugefields(NC, Nwing, ...) arrays (test/test_wilson.toml: Nwing = 1): interior only
        check(ccall((:lqcd_gauge_upload_wing, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), g.h, buf, Nwing))
    end
    return g
end
function Initialize_Gaugefields(NC, Nwing, L...; condition = "cold", lattice = nothing, randomseed = 111)    # universe.jl:41-49
    NC == 3 || error("only NC = 3 is supported on the HIP path")
    U = HIPGaugefields(lattice === nothing ? HIPLattice(Tuple(L)) : lattice)
    if condition == "cold"
        check(ccall((:lqcd_gauge_unit, LIB), Cint, (Ptr{Cvoid},), U.h))
    elseif condition == "hot"
        check(ccall((:lqcd_gauge_hot_start, LIB), Cint, (Ptr{Cvoid}, UInt64), U.h, randomseed))
    else
        error("condition = $condition is not supported")
    end
    return U
end
function calculate_Plaquette(g::HIPGaugefields)
    p = Ref{Float64}(0)
    check(ccall((:lqcd_gauge_plaquette, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), g.h, p))
    return p[]
end

# substitute_U!(Uold, U) (standardHMC.jl:45) and substitute_U!(U[mu], W) (AbstractMD.jl:93)
substitute_U!(dst::HIPGaugefields, src::HIPGaugefields) =
    check(ccall((:lqcd_gauge_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), dst.h, src.h))
substitute_U!(dst::HIPLink, src::HIPLink) =
    check(ccall((:lqcd_link_copy, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), dst.parent.h, dst.slot, src.parent.h, src.slot))
# mul!(W, expU, U[mu]) / mul!(temp1, U[mu], dSdUμ) (AbstractMD.jl:92,109)
mul!(C::HIPLink, A::HIPLink, B::HIPLink) =
    (check(ccall((:lqcd_link_mul, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint),
                 C.parent.h, C.slot, A.parent.h, A.slot, B.parent.h, B.slot)); C)
# exptU!(expU, t, p[mu], [temp1, temp2]) (AbstractMD.jl:91)
exptU!(expU::HIPLink, t::Number, p::HIPLink, temps = nothing) =
    check(ccall((:lqcd_link_exp, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Cint), expU.parent.h, expU.slot, Float64(t), p.parent.h, p.slot))
# Traceless_antihermitian_add!(p[mu], factor, temp1) (AbstractMD.jl:110,131)
Traceless_antihermitian_add!(p::HIPLink, factor::Number, G::HIPLink) =
    check(ccall((:lqcd_link_add_ta, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Cint), p.parent.h, p.slot, Float64(factor), G.parent.h, G.slot))

# ---- momenta: initialize_TA_Gaugefields(U) (standardMD.jl:34), gauss_distribution!(md.p) (:86), md.p * md.p (standardHMC.jl:49)
initialize_TA_Gaugefields(U::HIPGaugefields) = HIPGaugefields(U.lat)
gauss_distribution!(p::HIPGaugefields; seed = rand(UInt64)) =
    check(ccall((:lqcd_momentum_gaussian, LIB), Cint, (Ptr{Cvoid}, UInt64), p.h, seed))
function Base.:*(p::HIPGaugefields, q::HIPGaugefields)
    p === q || error("only p * p (the kinetic term of standardHMC.jl:49) is defined")
    k = Ref{Float64}(0)
    check(ccall((:lqcd_momentum_action, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), p.h, k))
    return 2 * k[]            # lqcd_momentum_action returns p.p/2
end

# ---- temporaries: Gaugefields.Temporalfields_module (get_temp / unused!, AbstractMD.jl:80-97): a pool of link-field views
mutable struct HIPTemporalfields
    lat::HIPLattice
    fields::Vector{HIPGaugefields}
    free::Vector{Int}
end
HIPTemporalfields(lat::HIPLattice) = HIPTemporalfields(lat, HIPGaugefields[], Int[])
function _grow!(t::HIPTemporalfields)
    push!(t.fields, HIPGaugefields(t.lat))
    base = 4 * (length(t.fields) - 1)
    append!(t.free, base:base+3)
end
_view(t::HIPTemporalfields, it::Int) = HIPLink(t.fields[it ÷ 4 + 1], Cint(it % 4))
function get_temp(t::HIPTemporalfields)
    isempty(t.free) && _grow!(t)
    it = popfirst!(t.free)
    return _view(t, it), it
end
function get_temp(t::HIPTemporalfields, n::Integer)
    pairs = [get_temp(t) for _ = 1:n]
    return [p[1] for p in pairs], [p[2] for p in pairs]
end
unused!(t::HIPTemporalfields, it::Integer) = (push!(t.free, it); sort!(t.free); nothing)
unused!(t::HIPTemporalfields, its::AbstractVector) = (foreach(i -> unused!(t, i), its); nothing)

# ---- gauge action: GaugeAction(U); push!(gauge_action, beta/2, plaqloop ∪ plaqloop') (universe.jl:88-96)
mutable struct HIPGaugeAction
    beta_inp::Float64
    temps::HIPTemporalfields
end
GaugeAction(U::HIPGaugefields) = HIPGaugeAction(0.0, HIPTemporalfields(U.lat))
Base.push!(ga::HIPGaugeAction, beta_inp::Number, loops) = (ga.beta_inp += beta_inp; ga)   # plaquette + adjoint only: the loops of universe.jl:92-93
get_temporary_gaugefields(ga::HIPGaugeAction) = ga.temps
# calc_dSdUμ!(dSdUμ, gauge_action, μ, U) (AbstractMD.jl:108): beta_inp * (sum of the staples of U[μ])
calc_dSdUμ!(dSdUμ::HIPLink, ga::HIPGaugeAction, μ::Integer, U::HIPGaugefields) =
    check(ccall((:lqcd_link_staple, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Float64), dSdUμ.parent.h, dSdUμ.slot, U.h, μ - 1, 2 * ga.beta_inp))
# evaluate_GaugeAction(gauge_action, U) (standardHMC.jl:50; S_g = -that / NC): lqcd_gauge_action returns S_g itself
function evaluate_GaugeAction(ga::HIPGaugeAction, U::HIPGaugefields)
    s = Ref{Float64}(0)
    check(ccall((:lqcd_gauge_action, LIB), Cint, (Ptr{Cvoid}, Float64, Ref{Float64}), U.h, 2 * ga.beta_inp, s))
    return -U.NC * s[]
end

# ------------------------------------------------------------------ fermion fields
mutable struct HIPFermion       # in the reference tree: <: LatticeDiracOperators.AbstractFermionfields_4D{3}
    h::Ptr{Cvoid}
    lat::HIPLattice
    kind::Cint
end
function HIPFermion(lat::HIPLattice, kind::Cint)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_spinor_create, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint, Cint), lat.h, h, kind, FULL))
    f = HIPFermion(h[], lat, kind)
    finalizer(x -> ccall((:lqcd_spinor_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), f)
    return f
end
# Initialize_pseudofermion_fields(U[1], "Wilson"; nowing = true)   (universe.jl:107,112)
Initialize_pseudofermion_fields(U::HIPLink, name::String; kwargs...) =
    HIPFermion(U.parent.lat, lowercase(name) == "wilson" ? WILSON : lowercase(name) == "staggered" ? STAGGERED : error("$name is not supported"))
Initialize_pseudofermion_fields(U::HIPGaugefields, name::String; kwargs...) = Initialize_pseudofermion_fields(U[1], name; kwargs...)
similar(x::HIPFermion) = HIPFermion(x.lat, x.kind)
clear_fermion!(x::HIPFermion) = check(ccall((:lqcd_spinor_zero, LIB), Cint, (Ptr{Cvoid},), x.h))
substitute_fermion!(a::HIPFermion, b::HIPFermion) = check(ccall((:lqcd_spinor_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), a.h, b.h))
function dot(a::HIPFermion, b::HIPFermion)
    re, im_ = Ref{Float64}(0), Ref{Float64}(0)
    check(ccall((:lqcd_dot, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), a.h, b.h, re, im_))
    return complex(re[], im_[])
end
# add_fermion!(c, alpha, a [, beta, b]) : c += alpha*a (+ beta*b)
function add_fermion!(c::HIPFermion, α::Number, a::HIPFermion)
    check(ccall((:lqcd_axpy, LIB), Cint, (Float64, Float64, Ptr{Cvoid}, Ptr{Cvoid}), real(α), imag(α), a.h, c.h))
end
function add_fermion!(c::HIPFermion, α::Number, a::HIPFermion, β::Number, b::HIPFermion)
    add_fermion!(c, α, a); add_fermion!(c, β, b)
end
gauss_distribution_fermion!(x::HIPFermion; seed = rand(UInt64)) =
    check(ccall((:lqcd_spinor_gaussian, LIB), Cint, (Ptr{Cvoid}, UInt64), x.h, seed))
Z4_distribution_fermi!(x::HIPFermion; seed = rand(UInt64)) =
    check(ccall((:lqcd_spinor_z4, LIB), Cint, (Ptr{Cvoid}, UInt64), x.h, seed))
# host <-> device in the reference layout psi[ic,ix,iy,iz,it,is]
upload!(x::HIPFermion, a::Array{ComplexF64,6}) = check(ccall((:lqcd_spinor_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), x.h, a))
download!(a::Array{ComplexF64,6}, x::HIPFermion) = check(ccall((:lqcd_spinor_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), x.h, a))

# ------------------------------------------------------------------ Dirac operator
mutable struct HIPDirac         # in the reference tree: <: LatticeDiracOperators.Dirac_operator{4}
    h::Ptr{Cvoid}
    U::HIPGaugefields
    x::HIPFermion               # the field Dirac_operator(U, x, params) was built from (kind; similar(x) for the action's temporaries)
    dagger::Bool
    eps_CG::Float64
    MaxCGstep::Int
    method_CG::String
    owner::Bool
end
# Dirac_operator(U, x, params::Dict)  (universe.jl:137; keys universe.jl:103-135)
function Dirac_operator(U::HIPGaugefields, x::HIPFermion, params::Dict)
    name = params["Dirac_operator"]
    kind = name in ("Wilson", "WilsonClover") ? WILSON : name in ("Staggered", "staggered") ? STAGGERED : error("$name is not supported")
    km = kind == WILSON ? Float64(params["κ"]) : Float64(get(params, "mass", 0.5))
    r = Float64(get(params, "r", 1.0))
    bc = Cint[get(params, "boundarycondition", [1, 1, 1, -1])...]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_op_create, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Float64, Float64, Ptr{Cint}),
                U.lat.h, h, kind, U.h, km, r, bc))
    if name == "WilsonClover"      # Clover_coefficient (src/system/parameter_structs.jl:125)
        check(ccall((:lqcd_op_set_clover, LIB), Cint, (Ptr{Cvoid}, Float64), h[], Float64(get(params, "Clover_coefficient", 1.5612))))
    end
    D = HIPDirac(h[], U, x, false, Float64(get(params, "eps_CG", 1e-19)), Int(get(params, "MaxCGstep", 3000)),
                 String(get(params, "method_CG", "bicgstab")), true)
    finalizer(d -> d.owner && ccall((:lqcd_op_destroy, LIB), Cint, (Ptr{Cvoid},), d.h), D)
    return D
end
# D(U): rebind links (unusedfiles/measure_chiral_condensate.jl:173)
function (D::HIPDirac)(U::HIPGaugefields)
    check(ccall((:lqcd_op_set_gauge, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), D.h, U.h)); D.U = U; D
end
adjoint(D::HIPDirac) = HIPDirac(D.h, D.U, D.x, !D.dagger, D.eps_CG, D.MaxCGstep, D.method_CG, false)
struct HIPDdagD
    D::HIPDirac
end
DdagD_operator(D::HIPDirac) = HIPDdagD(D)

mul!(y::HIPFermion, D::HIPDirac, x::HIPFermion) =
    (check(ccall((:lqcd_op_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint), D.h, y.h, x.h, D.dagger)); y)
mul!(y::HIPFermion, A::HIPDdagD, x::HIPFermion) =
    (check(ccall((:lqcd_op_apply_DdagD, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), A.D.h, y.h, x.h)); y)

# solve_DinvX!(y, A, x): stopping rule real(r.r) < eps_CG, error after MaxCGstep (SURVEY.md 3.3)
function solve_DinvX!(y::HIPFermion, A::HIPDdagD, x::HIPFermion)
    it, rr = Ref{Cint}(0), Ref{Float64}(0)
    check(ccall((:lqcd_solve_cg_DdagD, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ref{Cint}, Ref{Float64}),
                A.D.h, y.h, x.h, A.D.eps_CG, A.D.MaxCGstep, it, rr))
end
function solve_DinvX!(y::HIPFermion, D::HIPDirac, x::HIPFermion)
    it, rr = Ref{Cint}(0), Ref{Float64}(0)
    if D.method_CG == "bicg"               # the reference's default method_CG
        check(ccall((:lqcd_solve_bicg, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Ref{Cint}, Ref{Float64}),
                    D.h, y.h, x.h, D.dagger, D.eps_CG, D.MaxCGstep, it, rr))
    elseif D.method_CG == "bicgstab_evenodd"
        check(ccall((:lqcd_solve_bicgstab_eo, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Ref{Cint}, Ref{Float64}),
                    D.h, y.h, x.h, D.dagger, D.eps_CG, D.MaxCGstep, it, rr))
    else
        check(ccall((:lqcd_solve_bicgstab, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Ref{Cint}, Ref{Float64}),
                    D.h, y.h, x.h, D.dagger, D.eps_CG, D.MaxCGstep, it, rr))
    end
end

# shiftedcg(vec_x, vec_β, x, A, b): the RHMC solver (README.md:132)
function shiftedcg(vec_x::Vector{HIPFermion}, vec_β::Vector{Float64}, x::HIPFermion, A::HIPDdagD, b::HIPFermion)
    it, rr = Ref{Cint}(0), Ref{Float64}(0)
    hs = [v.h for v in vec_x]
    check(ccall((:lqcd_solve_multishift_cg, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Float64}, Cint, Float64, Cint, Ref{Cint}, Ref{Float64}),
                A.D.h, x.h, hs, b.h, vec_β, length(vec_β), A.D.eps_CG, A.D.MaxCGstep, it, rr))
end

# mixed-precision shiftedcg (fp32 multi-shift pass + fp64 defect correction per shift; the stopping rule holds for the true fp64 residuals)
function shiftedcg_mixed(vec_x::Vector{HIPFermion}, vec_β::Vector{Float64}, x::HIPFermion, A::HIPDdagD, b::HIPFermion; inner_tol = 0.0)
    it, outer, rr = Ref{Cint}(0), Ref{Cint}(0), Ref{Float64}(0)
    hs = [v.h for v in vec_x]
    check(ccall((:lqcd_solve_multishift_mixed_cg, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Cvoid}, Ptr{Float64}, Cint, Float64, Cint, Float64, Ref{Cint}, Ref{Cint}, Ref{Float64}),
                A.D.h, x.h, hs, b.h, vec_β, length(vec_β), A.D.eps_CG, A.D.MaxCGstep, inner_tol, it, outer, rr))
end

# staggered: the parity block (D'D)_pp on half-lattice vectors (the 4-taste action of test/test_staggered.toml lives on the even sites)
function solve_parity_DinvX!(y::HIPFermion, A::HIPDdagD, x::HIPFermion, parity::Integer = 0)
    it, rr = Ref{Cint}(0), Ref{Float64}(0)
    check(ccall((:lqcd_solve_cg_DdagD_parity, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Float64, Cint, Ref{Cint}, Ref{Float64}),
                A.D.h, y.h, x.h, parity, A.D.eps_CG, A.D.MaxCGstep, it, rr))
    y
end
# mixed-precision variant of solve_DinvX!(y, DdagD, x): fp32 inner CG, stopping rule on the true fp64 residual
function solve_mixed_DinvX!(y::HIPFermion, A::HIPDdagD, x::HIPFermion; inner_tol = 1e-4)
    it, out, rr = Ref{Cint}(0), Ref{Cint}(0), Ref{Float64}(0)
    check(ccall((:lqcd_solve_mixed_cg_DdagD, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Float64, Ref{Cint}, Ref{Cint}, Ref{Float64}),
                A.D.h, y.h, x.h, A.D.eps_CG, A.D.MaxCGstep, inner_tol, it, out, rr))
end

# ---- pseudofermion action and force: FermiAction(D, Dict) (universe.jl:138) and the generics of standardMD.jl:95-96,
# standardHMC.jl:54,71 and AbstractMD.jl:129.  2-flavour Wilson(-clover) and the 4- / 8-taste staggered actions; for any other Nf the
# reference's package brings Remez tables -- here the partial fractions come from the caller (rational_apply! / rational_force! below;
# the Python mirror fits them with latticeqcd.jl_amd/rational.py).
mutable struct HIPFermiAction
    D::HIPDirac
    Nf::Int
    _temporary_fermionfields::Vector{HIPFermion}     # standardMD.jl:50: η = similar(fermi_action._temporary_fermionfields[1])
    force::HIPGaugefields                            # G of lqcd_calc_UdSfdU, handed out per direction
end
function FermiAction(D::HIPDirac, parameters_action::Dict = Dict())
    kind = D.x.kind
    Nf = get(parameters_action, "Nf", kind == WILSON ? 2 : 4)
    (kind == WILSON && Nf == 2) || (kind == STAGGERED && Nf in (4, 8)) ||
        error("FermiAction: Nf = $Nf needs the rational action (rational_apply! / rational_force! with partial fractions from the caller)")
    return HIPFermiAction(D, Nf, [similar(D.x), similar(D.x)], HIPGaugefields(D.U.lat))
end
# gauss_sampling_in_action!(ξ, U, fa) (standardMD.jl:95): ξ ~ exp(-ξ†ξ), i.e. re and im of variance 1/2
function gauss_sampling_in_action!(ξ::HIPFermion, U::HIPGaugefields, fa::HIPFermiAction; seed = rand(UInt64))
    check(ccall((:lqcd_spinor_gaussian, LIB), Cint, (Ptr{Cvoid}, UInt64), ξ.h, seed))
    check(ccall((:lqcd_scale, LIB), Cint, (Float64, Float64, Ptr{Cvoid}), sqrt(0.5), 0.0, ξ.h))
end
# sample_pseudofermions!(η, U, fa, ξ) (standardMD.jl:96): η = D†ξ (4 staggered tastes: restricted to the even sites)
function sample_pseudofermions!(η::HIPFermion, U::HIPGaugefields, fa::HIPFermiAction, ξ::HIPFermion)
    mul!(η, fa.D(U)', ξ)
    if η.kind == STAGGERED && fa.Nf == 4
        half = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:lqcd_spinor_create, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint, Cint), η.lat.h, half, η.kind, EVEN))
        check(ccall((:lqcd_spinor_extract, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), half[], η.h))
        clear_fermion!(η)
        check(ccall((:lqcd_spinor_insert, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), η.h, half[]))
        ccall((:lqcd_spinor_destroy, LIB), Cint, (Ptr{Cvoid},), half[])
    end
    return η
end
# evaluate_FermiAction(fa, U, η) (standardHMC.jl:71): S_f = η†(D†D)^-1 η; X = (D†D)^-1 η and Y = D X stay in the action's temporaries
function evaluate_FermiAction(fa::HIPFermiAction, U::HIPGaugefields, η::HIPFermion)
    S, it = Ref{Float64}(0), Ref{Cint}(0)
    D = fa.D(U)
    X, Y = fa._temporary_fermionfields
    check(ccall((:lqcd_fermi_action, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ref{Float64}, Ref{Cint}),
                D.h, η.h, X.h, Y.h, D.eps_CG, D.MaxCGstep, S, it))
    return S[]
end
# calc_UdSfdU!(UdSfdUμ, fa, U, η) (AbstractMD.jl:129) with UdSfdUμ = get_temp(temps, Dim): solve, Y = D X and the outer-product sweep run
# resident into fa.force (= G, dS_f/dε[U -> exp(iεT)U] = -2 Im tr(T G)); each direction is handed over as "U dS_f/dU" = -G, the sign
# the caller's factor = -ϵ Δτ expects (AbstractMD.jl:127-132)
function calc_UdSfdU!(UdSfdUμ::Vector{HIPLink}, fa::HIPFermiAction, U::HIPGaugefields, η::HIPFermion)
    D = fa.D(U)
    check(ccall((:lqcd_calc_UdSfdU, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ptr{Float64}, Ptr{Cint}),
                D.h, fa.force.h, η.h, D.eps_CG, D.MaxCGstep, C_NULL, C_NULL))
    for μ = 1:4
        check(ccall((:lqcd_link_scaled_copy, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Cint),
                    UdSfdUμ[μ].parent.h, UdSfdUμ[μ].slot, -1.0, fa.force.h, μ - 1))
    end
end
# general staggered Nf (test/test_Nf2.toml:8, test/test_Nf3.toml:8): rational action, coefficients (a0, res, poles) from the caller
rational_apply!(y::HIPFermion, D::HIPDirac, x::HIPFermion, a0, res::Vector{Float64}, poles::Vector{Float64}) =
    check(ccall((:lqcd_rational_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Float64, Cint, Ptr{Cint}),
                D.h, y.h, x.h, a0, length(res), res, poles, D.eps_CG, D.MaxCGstep, C_NULL))
rational_force!(G::HIPGaugefields, D::HIPDirac, φ::HIPFermion, res::Vector{Float64}, poles::Vector{Float64}) =
    check(ccall((:lqcd_rational_force, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Float64, Cint, Ptr{Cint}),
                D.h, G.h, φ.h, length(res), res, poles, D.eps_CG, D.MaxCGstep, C_NULL))

# ---- fused four-direction forms of the MD step (one kernel each; what a maintainer would call from a specialised
# P_update!(U::HIPGaugefields, p, ϵ, md) / U_update! method to skip the per-direction temporaries)
P_update_fused!(U::HIPGaugefields, p::HIPGaugefields, factor, β) =      # p += factor * TA(-(β/6) U * staples), the force field is never stored
    check(ccall((:lqcd_momentum_add_gauge_force, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64), p.h, factor, U.h, β))
U_update_fused!(U::HIPGaugefields, p::HIPGaugefields, dt) =             # U <- exp(dt p) U
    check(ccall((:lqcd_gauge_exp_update, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), U.h, dt, p.h))
momentum_add_ta_fused!(p::HIPGaugefields, factor, G::HIPGaugefields) =  # p += factor * TA(G), all four directions
    check(ccall((:lqcd_momentum_add_ta, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), p.h, factor, G.h))

end # module
