# LatticeQCDHIP.jl -- reference-side binding of liblqcd_hip.so (C ABI: include/lqcd_hip.h).
#
# STATUS: written against the documented LatticeDiracOperators.jl 0.6 / Gaugefields.jl 0.7 interface but NEVER EXECUTED:
# there is no Julia in the build image (SURVEY.md section 0.3).  Everything below the `ccall`s is exercised through the
# same C ABI by tests/ (Python ctypes).  The file is deliberately thin: every method is one ccall plus error translation.
#
# What it provides: device-backed field / operator types that dispatch the reference's hot-path generics
#   mul!(y, D, x), mul!(y, D', x), solve_DinvX!(y, A, x), dot, clear_fermion!, add_fermion!, substitute_fermion!, similar,
#   gauss_distribution_fermion!, Z4_distribution_fermi!   (SURVEY.md 8(a), 8(b))
# so that src/md/AbstractMD.jl:120-135, src/md/standardMD.jl:82-101 and src/updates/standardHMC.jl:41-91 run unchanged
# once `Univ` (src/system/universe.jl:100-143) constructs these types instead of the CPU ones.
module LatticeQCDHIP

using LinearAlgebra
import LinearAlgebra: mul!, dot
import Base: similar, adjoint

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

# ------------------------------------------------------------------ gauge field: the reference's U::Vector (U[1:4]) as one device object
mutable struct HIPGaugefields   # in the reference tree: <: Gaugefields.AbstractGaugefields{3,4}
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
# substitute_U!(Udev, U): U is the reference's Vector of 4 Array{ComplexF64,6} (NC,NC,NX,NY,NZ,NT), Nwing = 0
function substitute_U!(g::HIPGaugefields, U::Vector{<:AbstractArray{ComplexF64,6}}; Nwing = 0)
    buf = cat(U...; dims = 7)      # [a,b,x,y,z,t,mu] column-major == lqcd LAYOUT_REFERENCE (extents L .+ 2Nwing when the fields carry wings)
    if Nwing == 0
        check(ccall((:lqcd_gauge_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), g.h, buf, LAYOUT_REFERENCE))
    else    # the reference's Initialize_Gaugefields(NC, Nwing, ...) arrays (test/test_wilson.toml: Nwing = 1): interior only
        check(ccall((:lqcd_gauge_upload_wing, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), g.h, buf, Nwing))
    end
    return g
end
function calculate_Plaquette(g::HIPGaugefields)
    p = Ref{Float64}(0)
    check(ccall((:lqcd_gauge_plaquette, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), g.h, p))
    return p[]
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
Initialize_pseudofermion_fields(U::HIPGaugefields, name::String; kwargs...) =
    HIPFermion(U.lat, lowercase(name) == "wilson" ? WILSON : lowercase(name) == "staggered" ? STAGGERED : error("$name is not supported"))
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
    D = HIPDirac(h[], U, false, Float64(get(params, "eps_CG", 1e-19)), Int(get(params, "MaxCGstep", 3000)),
                 String(get(params, "method_CG", "bicgstab")), true)
    finalizer(d -> d.owner && ccall((:lqcd_op_destroy, LIB), Cint, (Ptr{Cvoid},), d.h), D)
    return D
end
# D(U): rebind links (unusedfiles/measure_chiral_condensate.jl:173)
function (D::HIPDirac)(U::HIPGaugefields)
    check(ccall((:lqcd_op_set_gauge, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), D.h, U.h)); D.U = U; D
end
adjoint(D::HIPDirac) = HIPDirac(D.h, D.U, !D.dagger, D.eps_CG, D.MaxCGstep, D.method_CG, false)
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
    f = D.method_CG == "bicgstab_evenodd" ? :lqcd_solve_bicgstab_eo : :lqcd_solve_bicgstab
    if f == :lqcd_solve_bicgstab_eo
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

# ---- pseudofermion action / force and the gauge side of the MD step: everything src/md/AbstractMD.jl:78-135 and
# src/updates/standardHMC.jl:41-91 call, with every field resident on the device
struct HIPFermiAction            # FermiAction(D, Dict("Nf" => 2)) (universe.jl:138)
    D::HIPDirac
end
function evaluate_FermiAction(fa::HIPFermiAction, U::HIPGaugefields, η::HIPFermion, X::HIPFermion, Y::HIPFermion)
    S, it = Ref{Float64}(0), Ref{Cint}(0)
    D = fa.D(U)
    check(ccall((:lqcd_fermi_action, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ref{Float64}, Ref{Cint}),
                D.h, η.h, X.h, Y.h, D.eps_CG, D.MaxCGstep, S, it))
    S[]
end
function calc_UdSfdU!(UdSfdU::HIPGaugefields, fa::HIPFermiAction, U::HIPGaugefields, η::HIPFermion)
    D = fa.D(U)
    check(ccall((:lqcd_calc_UdSfdU, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ptr{Float64}, Ptr{Cint}),
                D.h, UdSfdU.h, η.h, D.eps_CG, D.MaxCGstep, C_NULL, C_NULL))
end
# general staggered Nf (test/test_Nf2.toml:8, test/test_Nf3.toml:8): rational action, coefficients (a0, res, poles) from the host
rational_apply!(y::HIPFermion, D::HIPDirac, x::HIPFermion, a0, res::Vector{Float64}, poles::Vector{Float64}) =
    check(ccall((:lqcd_rational_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Float64, Cint, Ptr{Cint}),
                D.h, y.h, x.h, a0, length(res), res, poles, D.eps_CG, D.MaxCGstep, C_NULL))
rational_force!(UdSfdU::HIPGaugefields, D::HIPDirac, φ::HIPFermion, res::Vector{Float64}, poles::Vector{Float64}) =
    check(ccall((:lqcd_rational_force, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Float64, Cint, Ptr{Cint}),
                D.h, UdSfdU.h, φ.h, length(res), res, poles, D.eps_CG, D.MaxCGstep, C_NULL))
gauge_force!(G::HIPGaugefields, U::HIPGaugefields, β) =
    check(ccall((:lqcd_gauge_force, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64), G.h, U.h, β))
Traceless_antihermitian_add!(p::HIPGaugefields, factor, G::HIPGaugefields) =
    check(ccall((:lqcd_momentum_add_ta, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), p.h, factor, G.h))
U_update!(U::HIPGaugefields, p::HIPGaugefields, dt) =
    check(ccall((:lqcd_gauge_exp_update, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), U.h, dt, p.h))
gauss_distribution!(p::HIPGaugefields; seed = 114) =
    check(ccall((:lqcd_momentum_gaussian, LIB), Cint, (Ptr{Cvoid}, UInt64), p.h, seed))
substitute_U!(dst::HIPGaugefields, src::HIPGaugefields) =
    check(ccall((:lqcd_gauge_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), dst.h, src.h))
function momentum_action(p::HIPGaugefields)      # md.p * md.p / 2
    k = Ref{Float64}(0)
    check(ccall((:lqcd_momentum_action, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), p.h, k)); k[]
end
function evaluate_GaugeAction(U::HIPGaugefields, β)  # the -evaluate_GaugeAction/NC term of standardHMC.jl:50
    s = Ref{Float64}(0)
    check(ccall((:lqcd_gauge_action, LIB), Cint, (Ptr{Cvoid}, Float64, Ref{Float64}), U.h, β, s)); s[]
end

end # module
