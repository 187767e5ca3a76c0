# LatticeQCDHIP.jl -- reference-side binding of liblqcd_hip.so (C ABI: include/lqcd_hip.h).
#
# STATUS: written against the interface the reference's own callers use (every generic and every struct-field constraint of
# src/md/AbstractMD.jl:78-135, src/md/standardMD.jl:5-101, src/updates/standardHMC.jl:1-91, src/system/universe.jl:30-143 is listed in
# tests/golden/ref_caller_inventory.json and checked against this file by tests/test_julia_binding_static.py) but NEVER EXECUTED: there is no Julia
# in the build image (SURVEY.md section 0.3).  Everything below the `ccall`s is exercised through the same C ABI by tests/ (Python ctypes).
# The file is deliberately thin: every method is one ccall plus error translation; there is NO module-level mutable state (the lazy fusion of the
# callers' per-direction link-call triples lives below the C ABI, csrc/md.hip).
#
# How it plugs in.  The reference has no FFI; its seam is multiple dispatch on types of Gaugefields.jl / LatticeDiracOperators.jl.  This
# module therefore EXTENDS the packages' own generic functions (`import Gaugefields: substitute_U!, ...`) with methods on device-backed
# types, so that the unchanged callers -- which call Gaugefields.substitute_U!, LatticeDiracOperators.calc_UdSfdU!, ... -- land here:
#
#   U, Uold, dSdU :: Vector{HIPLink}     four views (one per direction) of ONE device field; HIPLink <: AbstractGaugefields{3,4}, so
#                                        `U::Vector{TG}`, `Uold::Vector{TG}` (universe.jl:12, standardHMC.jl:3) hold with TG = HIPLink and
#                                        the package's own GaugeAction(U) builds a GaugeAction{4,HIPLink} (standardMD.jl:6,23)
#   p             :: Vector{HIPTALink}   the same for the momenta (`p::Vector{TA}`, standardMD.jl:10)
#   temporaries   :: HIPLink             similar(U[1]) -- the package's Temporalfields pool (get_temp / unused!) works unchanged on them
#   x, η, ξ       :: HIPFermion          D :: HIPDirac, fermi_action :: HIPFermiAction
#
# `Univ` (src/system/universe.jl:41-49) creates the links with Initialize_Gaugefields(NC, Nwing, L...; condition = ...), a function with no argument
# to dispatch on.  Two ways in (INTEGRATION.md section 3): ONE edit -- write Initialize_HIPGaugefields there -- or NO edit: call
# LatticeQCDHIP.activate!() once before Univ(p); it adds a method of the package's own Initialize_Gaugefields for four Int extents that is more
# specific than the package's untyped one and returns device links (deactivate!() removes the redirection, the package's method was never replaced).
module LatticeQCDHIP

using LinearAlgebra
using Gaugefields
using LatticeDiracOperators
import LinearAlgebra: mul!, dot
import Base: similar, adjoint
import Gaugefields: AbstractGaugefields, GaugeAction, Initialize_Gaugefields, substitute_U!, exptU!, Traceless_antihermitian_add!, calc_dSdUμ!,
    evaluate_GaugeAction, initialize_TA_Gaugefields, gauss_distribution!, calc_smearedU, println_verbose_level1,
    println_verbose_level2, println_verbose_level3, get_myrank, calculate_Plaquette, calculate_Polyakov_loop, load_BridgeText!, load_gaugefield!,
    CovNeuralnet, CovLayer, STOUT_Layer, back_prop
import LatticeDiracOperators: Dirac_operator, DdagD_operator, FermiAction, Initialize_pseudofermion_fields,
    gauss_sampling_in_action!, sample_pseudofermions!, evaluate_FermiAction, calc_UdSfdU!, solve_DinvX!, shiftedcg,
    clear_fermion!, substitute_fermion!, add_fermion!, gauss_distribution_fermion!, Z4_distribution_fermi!,
    AbstractFermionfields_4D

export Initialize_HIPGaugefields, HIPLattice, HIPLink, HIPTALink, HIPFermion, HIPDirac, HIPFermiAction, reunitarize!, activate!, deactivate!

const LIB = get(ENV, "LQCD_HIP_LIB", joinpath(@__DIR__, "..", "latticeqcd.jl_amd", "csrc", "liblqcd_hip.so"))

const LQCD_OK = Cint(0)
const LQCD_ERR_NOT_CONVERGED = Cint(3)
const WILSON, STAGGERED, DOMAINWALL = Cint(0), Cint(1), Cint(2)
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
    verbose::Int        # println_verbose_level2/3(U[1], ...) print at or above this level (Univ: p.verboselevel)
    spare::Any          # the HIPGaugeStorage whose free slots the next temporaries (similar(U[1])) take, or nothing -- per context, not per module
    stout::Vector{Any}  # link fields of the stout layers (calc_smearedU: one per layer, then the two force fields of back_prop), made on first use
end
function HIPLattice(L::NTuple{4,Int}; PEs = (1, 1, 1, 1), rank = 0, device = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_ctx_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cint}, Ptr{Cint}, Cint),
                h, device, Cint[L...], Cint[PEs...], rank))
    lat = HIPLattice(h[], L, PEs, rank, 2, nothing, Any[])
    # U_update! / P_update! hand the temporaries of their per-direction call triples back to the pool unread (AbstractMD.jl:95-97,113-117): the binding
    # switches the library's fusion of those triples on (the plain C ABI is eager by default: a fused triple never writes its temporaries)
    check(ccall((:lqcd_ctx_set_param, LIB), Cint, (Ptr{Cvoid}, Cstring, Cint), lat.h, "lazy_links", 1))
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
# The second backend (csrc/comm.hip): peer-mapped windows.  Every rank exports the 256-byte description of its window, the host gathers them in rank order
# (peer_init!(lat, MPI.Allgather(peer_export(lat), comm)): nranks * 256 bytes) and every rank maps the others'.  One node, <= 8 ranks.
function peer_export(lat::HIPLattice)
    blob = zeros(UInt8, 256)
    check(ccall((:lqcd_ctx_peer_export, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}), lat.h, blob))
    return blob
end
peer_init!(lat::HIPLattice, blobs::Vector{UInt8}) =
    check(ccall((:lqcd_ctx_peer_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint), lat.h, blobs, prod(lat.PEs)))
function comm_backend(lat::HIPLattice)
    b = Ref{Cint}(0)
    check(ccall((:lqcd_ctx_comm_backend, LIB), Cint, (Ptr{Cvoid}, Ptr{Cint}), lat.h, b))
    return (:none, :rccl, :peer)[b[] + 1]
end
set_param!(lat::HIPLattice, key::String, value::Integer) =
    check(ccall((:lqcd_ctx_set_param, LIB), Cint, (Ptr{Cvoid}, Cstring, Cint), lat.h, key, value))

# ------------------------------------------------------------------ gauge-shaped device storage and its per-direction views
mutable struct HIPGaugeStorage      # one device allocation: four link-shaped slots [parity][chunk][slot][9][64] (lqcd_gauge_t)
    h::Ptr{Cvoid}
    lat::HIPLattice
    used::Int                       # slots handed out by similar(::HIPLink) (temporaries share storages four at a time)
end
function HIPGaugeStorage(lat::HIPLattice)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_gauge_create, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}), lat.h, h))
    g = HIPGaugeStorage(h[], lat, 0)
    finalizer(x -> ccall((:lqcd_gauge_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), g)      # the library runs recorded link operations that name the field before it frees it
    return g
end

# one direction of a gauge field: what the reference calls U[μ], a temporary, dSdUμ, expU, W ...
struct HIPLink <: AbstractGaugefields{3,4}
    parent::HIPGaugeStorage
    slot::Cint                      # 0..3
end
# one direction of the momentum field p[μ] (traceless anti-Hermitian 3x3 per site, stored link-shaped)
struct HIPTALink
    parent::HIPGaugeStorage
    slot::Cint
end
const AnyLink = Union{HIPLink,HIPTALink}
function Base.getproperty(l::AnyLink, s::Symbol)        # U[1].NC (AbstractMD.jl:100, standardHMC.jl:42) and the package's generic accessors
    s === :NC && return 3
    s === :Nwing && return 0
    s === :NX && return getfield(l, :parent).lat.L[1]
    s === :NY && return getfield(l, :parent).lat.L[2]
    s === :NZ && return getfield(l, :parent).lat.L[3]
    s === :NT && return getfield(l, :parent).lat.L[4]
    s === :NV && return prod(getfield(l, :parent).lat.L)
    return getfield(l, s)
end
Base.size(l::AnyLink) = (3, 3, getfield(l, :parent).lat.L...)
lattice(U::Vector{<:AnyLink}) = getfield(U[1], :parent).lat

# the four directions of one storage, in order?  Whole-field entry points (copy, plaquette, operator, action, force) need that;
# fields made by Initialize_HIPGaugefields / similar(U) / initialize_TA_Gaugefields(U) always are.
function whole(U::Vector{<:AnyLink})
    length(U) == 4 || error("the HIP path handles four-dimensional fields (length(U) == 4)")
    g = getfield(U[1], :parent)
    for μ = 1:4
        (getfield(U[μ], :parent) === g && getfield(U[μ], :slot) == μ - 1) ||
            error("U[1:4] must be the four directions of one device field (Initialize_HIPGaugefields, similar(U))")
    end
    return g
end
views(::Type{T}, g::HIPGaugeStorage) where {T<:AnyLink} = T[T(g, Cint(μ - 1)) for μ = 1:4]

# ---- the per-direction call triples of the reference's U_update! / P_update! (AbstractMD.jl:91-93, 108-110):
# exptU!(expU, t, p[μ]) -> mul!(W, expU, U[μ]) -> substitute_U!(U[μ], W) and calc_dSdUμ!(dSdUμ, ..) -> mul!(temp1, U[μ], dSdUμ) ->
# Traceless_antihermitian_add!(p[μ], factor, temp1).  Each generic below is ONE stateless ccall (lqcd_link_exp, lqcd_link_mul, lqcd_link_copy,
# lqcd_link_staple, lqcd_link_add_ta); the LIBRARY records the first two calls of a triple per context, launches one fused kernel at the third and
# turns four completed triples of one update into one four-direction launch (csrc/md.hip "lazy link triples", tunable lazy_links) -- every other
# entry point that touches a gauge-shaped field runs what is recorded first.  The temporaries of a completed triple (expU, W, dSdUμ, temp1) are not
# written; the callers return them to their pool unread.  set_param!(lattice, "lazy_links", 0) makes every call launch its own kernel.
slotof(l::AnyLink) = getfield(l, :slot)
hof(l::AnyLink) = getfield(l, :parent).h

# Initialize_Gaugefields(NC, Nwing, L...; condition) (universe.jl:41-49) -> U::Vector{HIPLink}, the value `Univ` stores as U::Vector{TG}
function Initialize_HIPGaugefields(NC, Nwing, L...; condition = "cold", lattice = nothing, randomseed = 111)::Vector{HIPLink}
    NC == 3 || error("only NC = 3 is supported on the HIP path")
    length(L) == 4 || error("only Dim = 4 is supported on the HIP path")
    g = HIPGaugeStorage(lattice === nothing ? HIPLattice(Tuple(Int.(L))) : lattice)
    g.used = 4
    if condition == "cold"
        check(ccall((:lqcd_gauge_unit, LIB), Cint, (Ptr{Cvoid},), g.h))
    elseif condition == "hot"
        check(ccall((:lqcd_gauge_hot_start, LIB), Cint, (Ptr{Cvoid}, UInt64), g.h, randomseed))
    else
        error("condition = $condition is not supported")
    end
    return views(HIPLink, g)
end
# `Univ` unchanged (universe.jl:41-49 calls Initialize_Gaugefields(NC, Nwing, L...; condition = p.initial | "cold") with Int arguments): after
# activate!() that call lands here for NC = 3 and four extents -- a method of the PACKAGE's function whose positional types (Int, Int, four Ints) are
# more specific than the package's own untyped signature, so dispatch prefers it; every other call (other NC, Dim = 2, MPI keywords) is handed on to
# the package's method through invoke.  The redirection is a method table entry, not module state: deactivate!() deletes it.
function activate!()
    @eval function Gaugefields.Initialize_Gaugefields(NC::Int, Nwing::Int, NX::Int, NY::Int, NZ::Int, NT::Int; condition = "cold", kwargs...)
        if NC == 3 && isempty(kwargs) && condition in ("cold", "hot")
            return Initialize_HIPGaugefields(NC, Nwing, NX, NY, NZ, NT; condition = condition)
        end
        return invoke(Gaugefields.Initialize_Gaugefields, Tuple{Any,Any,Vararg{Any}}, NC, Nwing, NX, NY, NZ, NT; condition = condition, kwargs...)
    end
    return nothing
end
function deactivate!()
    m = which(Gaugefields.Initialize_Gaugefields, Tuple{Int,Int,Int,Int,Int,Int})
    m.module === @__MODULE__() && Base.delete_method(m)
    return nothing
end
# Uold = similar(U) (standardHMC.jl:32), dSdU = similar(U) (standardMD.jl:58): a fresh four-direction field
function similar(U::Vector{HIPLink})::Vector{HIPLink}
    g = HIPGaugeStorage(lattice(U))
    g.used = 4
    return views(HIPLink, g)
end
# similar(U[1]): ONE link-shaped temporary -- what the package's Temporalfields pool (get_temp / unused!, AbstractMD.jl:80-97) and
# GaugeAction(U) allocate.  Four temporaries share one device storage.
function similar(l::HIPLink)::HIPLink
    lat = getfield(l, :parent).lat
    g = lat.spare
    if g === nothing || g.used >= 4
        g = HIPGaugeStorage(lat)
        lat.spare = g
    end
    g.used += 1
    return HIPLink(g, Cint(g.used - 1))
end
get_myrank(l::AnyLink) = getfield(l, :parent).lat.rank                                   # universe.jl:52
println_verbose_level1(l::AnyLink, val...) = (get_myrank(l) == 0 && println(val...); nothing)
println_verbose_level2(l::AnyLink, val...) = (get_myrank(l) == 0 && getfield(l, :parent).lat.verbose >= 2 && println(val...); nothing)   # standardHMC.jl:75-86
println_verbose_level3(l::AnyLink, val...) = (get_myrank(l) == 0 && getfield(l, :parent).lat.verbose >= 3 && println(val...); nothing)   # standardHMC.jl:51,55,63
# calc_smearedU(U, md.cov_neural_net) with cov_neural_net = nothing: always reached from update! (standardHMC.jl:67 compares the VALUE
# nothing with the TYPE Nothing, which is true) -- no smearing, the fermion action sees U itself
calc_smearedU(U::Vector{HIPLink}, ::Nothing) = (U, nothing, nothing)
# ---- stout smearing of the links the fermion action sees (universe.jl:147-171; standardMD.jl:91, 192-227; standardHMC.jl:67-68).
# Univ keeps the net in a field typed Union{Nothing,CovNeuralnet{Dim}} (universe.jl:15): the container is the package's own (CovNeuralnet(U) and push! are its
# generic methods); the LAYER is ours -- STOUT_Layer dispatches on the links it is given -- and so are the two methods that touch links.  Served: the
# plaquette loop with one rho per layer.  [EXT-RECALL: `CovLayer{Dim}` as the abstract layer type and `nn.layers` as the container's field are the
# package's names as this file's author knows them; there is no Julia here to run them against.]
struct HIPStoutLayer <: CovLayer{4}
    ρ::Float64
end
function STOUT_Layer(loops, ρ, U::Vector{HIPLink})
    (length(loops) == 1 && lowercase(String(loops[1])) == "plaquette" && length(ρ) == 1) ||
        error("STOUT_Layer on device links: loops = $loops with ρ = $ρ are not supported (the plaquette loop with one ρ is)")
    return HIPStoutLayer(Float64(ρ[1]))
end
function stout_field(lat::HIPLattice, k::Int, like::Vector{HIPLink})
    while length(lat.stout) < k
        push!(lat.stout, similar(like))
    end
    return lat.stout[k]::Vector{HIPLink}
end
function calc_smearedU(U::Vector{HIPLink}, nn::CovNeuralnet{4})
    lat = lattice(U)
    cur, multi = U, Vector{HIPLink}[]
    for (k, layer) in enumerate(nn.layers)
        out = stout_field(lat, k, U)
        check(ccall((:lqcd_stout_smear, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Float64), whole(out).h, whole(cur).h, (layer::HIPStoutLayer).ρ))
        push!(multi, out)
        cur = out
    end
    return cur, multi, nothing
end
# back_prop(md.dSdU, md.cov_neural_net, Uout_multi, U) (standardMD.jl:216): dSdU[μ] = Uout[μ]' (Uout dS/dUout)[μ] in, dSdUbare out with U[μ] dSdUbare[μ] = U dS/dU
function back_prop(dSdU::Vector{HIPLink}, nn::CovNeuralnet{4}, Uout_multi, U::Vector{HIPLink})
    n = length(nn.layers)
    n == 0 && return dSdU
    lat = lattice(U)
    F, bare = stout_field(lat, n + 1, U), stout_field(lat, n + 2, U)
    Uout = Uout_multi[n]
    for μ = 1:4
        mul!(F[μ], Uout[μ], dSdU[μ])
    end
    for k = n:-1:1
        thin = k == 1 ? U : Uout_multi[k-1]
        check(ccall((:lqcd_stout_backprop, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64), whole(F).h, whole(F).h, whole(thin).h, (nn.layers[k]::HIPStoutLayer).ρ))
    end
    for μ = 1:4
        mul!(bare[μ], U[μ]', F[μ])
    end
    return bare
end

# host <-> device.  U_host: the reference's Vector of 4 Array{ComplexF64,6} (NC,NC,NX,NY,NZ,NT) [+ wings]
function substitute_U!(U::Vector{HIPLink}, Uh::Vector{<:AbstractArray{ComplexF64,6}}; Nwing = 0)
    buf = cat(Uh...; dims = 7)     # [a,b,x,y,z,t,mu] column-major == LAYOUT_REFERENCE (extents L .+ 2Nwing when the fields carry wings)
    if Nwing == 0
        check(ccall((:lqcd_gauge_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), whole(U).h, buf, LAYOUT_REFERENCE))
    else    # the reference's Initialize_Gaugefields(NC, Nwing, ...) arrays (test/test_wilson.toml: Nwing = 1): interior only
        check(ccall((:lqcd_gauge_upload_wing, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), whole(U).h, buf, Nwing))
    end
    return U
end
# load_BridgeText!(filename, U, L, NC) / load_gaugefield!(U, i, ildg, L, NC) (universe.jl:60-66): the package's readers fill a host
# field, which is then uploaded
function load_BridgeText!(filename::String, U::Vector{HIPLink}, L, NC)
    Uh = Gaugefields.Initialize_Gaugefields(NC, 0, L..., condition = "cold")
    load_BridgeText!(filename, Uh, L, NC)
    substitute_U!(U, [Uh[μ].U for μ = 1:4])
end
function load_gaugefield!(U::Vector{HIPLink}, i, ildg, L, NC)
    Uh = Gaugefields.Initialize_Gaugefields(NC, 0, L..., condition = "cold")
    load_gaugefield!(Uh, i, ildg, L, NC)
    substitute_U!(U, [Uh[μ].U for μ = 1:4])
end
function calculate_Plaquette(U::Vector{HIPLink})
    p = Ref{Float64}(0)
    check(ccall((:lqcd_gauge_plaquette, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), whole(U).h, p))
    return p[]
end
# every link back onto SU(3): once per trajectory keeps the 12-real Dslash path alive when the links are updated through the per-direction
# entry points (the fused lqcd_gauge_exp_update does it in the same pass)
# calculate_Polyakov_loop(U, temp1, temp2): the second observable of every trajectory of the reference's runs (Polyakov_loop in test/*.toml)
function calculate_Polyakov_loop(U::Vector{HIPLink}, temps...)
    re, im_ = Ref{Float64}(0), Ref{Float64}(0)
    check(ccall((:lqcd_gauge_polyakov, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}, Ref{Float64}), whole(U).h, re, im_))
    return complex(re[], im_[])
end
reunitarize!(U::Vector{HIPLink}) = check(ccall((:lqcd_gauge_reunitarize, LIB), Cint, (Ptr{Cvoid},), whole(U).h))

# substitute_U!(Uold, U) / substitute_U!(U, Uold) (standardHMC.jl:45,84) and substitute_U!(U[mu], W) (AbstractMD.jl:93)
function substitute_U!(dst::Vector{HIPLink}, src::Vector{HIPLink})
    check(ccall((:lqcd_gauge_copy, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), whole(dst).h, whole(src).h))
end
substitute_U!(dst::HIPLink, src::HIPLink) =
    check(ccall((:lqcd_link_copy, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), hof(dst), slotof(dst), hof(src), slotof(src)))
# mul!(W, expU, U[mu]) / mul!(temp1, U[mu], dSdUμ) (AbstractMD.jl:92,109)
function mul!(C::HIPLink, A::HIPLink, B::HIPLink)
    check(ccall((:lqcd_link_mul, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), hof(C), slotof(C), hof(A), slotof(A), hof(B), slotof(B)))
    return C
end
# mul!(md.dSdU[μ], Uout[μ]', UdSfdUμ[μ]) (standardMD.jl:211): the adjoint of a link as the first factor
struct HIPLinkAdjoint
    parent::HIPLink
end
adjoint(l::HIPLink) = HIPLinkAdjoint(l)
function mul!(C::HIPLink, A::HIPLinkAdjoint, B::HIPLink)
    check(ccall((:lqcd_link_mul_adj, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), hof(C), slotof(C), hof(A.parent), slotof(A.parent), hof(B), slotof(B)))
    return C
end
# exptU!(expU, t, p[mu], [temp1, temp2]) (AbstractMD.jl:91)
exptU!(expU::HIPLink, t::Number, p::HIPTALink, temps) =
    check(ccall((:lqcd_link_exp, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Cint), hof(expU), slotof(expU), Float64(t), hof(p), slotof(p)))
# Traceless_antihermitian_add!(p[mu], factor, temp1) (AbstractMD.jl:110,131)
Traceless_antihermitian_add!(p::HIPTALink, factor::Number, G::HIPLink) =
    check(ccall((:lqcd_link_add_ta, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Cint), hof(p), slotof(p), Float64(factor), hof(G), slotof(G)))

# ---- momenta: initialize_TA_Gaugefields(U) (standardMD.jl:34), gauss_distribution!(md.p) (:86), md.p * md.p (standardHMC.jl:49,59)
function initialize_TA_Gaugefields(U::Vector{HIPLink})::Vector{HIPTALink}
    g = HIPGaugeStorage(lattice(U))
    g.used = 4
    return views(HIPTALink, g)
end
gauss_distribution!(p::Vector{HIPTALink}; seed = rand(UInt64)) =
    check(ccall((:lqcd_momentum_gaussian, LIB), Cint, (Ptr{Cvoid}, UInt64), whole(p).h, seed))
function Base.:*(p::Vector{HIPTALink}, q::Vector{HIPTALink})
    p === q || error("only p * p (the kinetic term of standardHMC.jl:49) is defined")
    k = Ref{Float64}(0)
    check(ccall((:lqcd_momentum_action, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), whole(p).h, k))
    return 2 * k[]            # lqcd_momentum_action returns p.p/2
end

# ---- gauge action.  GaugeAction(U), push!(gauge_action, β/2, plaqloop ∪ plaqloop') (universe.jl:88-96), get_temporary_gaugefields,
# get_temp and unused! are the PACKAGE's own: GaugeAction(U::Vector{<:AbstractGaugefields{NC,Dim}}) builds a GaugeAction{4,HIPLink}
# whose temporaries are similar(U[1]) (above).  Specialised here are the two generics that do arithmetic.  The coupling is read from the
# package's bookkeeping: dataset[i].β is the coefficient push! stored (β/2 for the plaquette and its adjoint together).
function beta_inp(ga::GaugeAction{4,HIPLink})
    all(d -> length(d.closedloops) == 12, ga.dataset) ||
        error("the HIP staple kernel implements the plaquette action (make_loops_fromname(\"plaquette\") and its adjoint, universe.jl:92-93)")
    return sum(d.β for d in ga.dataset)
end
# calc_dSdUμ!(dSdUμ, gauge_action, μ, U) (AbstractMD.jl:108): β_inp * (sum of the staples of U[μ])
calc_dSdUμ!(dSdUμ::HIPLink, ga::GaugeAction{4,HIPLink}, μ::Integer, U::Vector{HIPLink}) =
    check(ccall((:lqcd_link_staple, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Float64), hof(dSdUμ), slotof(dSdUμ), whole(U).h, μ - 1, 2 * beta_inp(ga)))
# evaluate_GaugeAction(gauge_action, U) (standardHMC.jl:50,60; S_g = -that / NC): lqcd_gauge_action returns S_g itself
function evaluate_GaugeAction(ga::GaugeAction{4,HIPLink}, U::Vector{HIPLink})
    s = Ref{Float64}(0)
    check(ccall((:lqcd_gauge_action, LIB), Cint, (Ptr{Cvoid}, Float64, Ref{Float64}), whole(U).h, 2 * beta_inp(ga), s))
    return -3 * s[]
end

# ------------------------------------------------------------------ fermion fields
mutable struct HIPFermion <: AbstractFermionfields_4D{3}
    h::Ptr{Cvoid}
    lat::HIPLattice
    kind::Cint
    L5::Int             # DOMAINWALL: extent of the fifth direction (L5 Wilson fields in one device allocation); 0 otherwise
    parent::Any         # a slice view (slice(x, i5)) keeps its five-dimensional field alive; nothing otherwise
end
function HIPFermion(lat::HIPLattice, kind::Cint; L5::Integer = 0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if kind == DOMAINWALL
        check(ccall((:lqcd_spinor_create_5d, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint), lat.h, h, L5))
    else
        check(ccall((:lqcd_spinor_create, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint, Cint), lat.h, h, kind, FULL))
    end
    f = HIPFermion(h[], lat, kind, Int(L5), nothing)
    finalizer(x -> ccall((:lqcd_spinor_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), f)
    return f
end
# x.w[i5] of the package's five-dimensional fields: a Wilson field that aliases slice i5 (1-based here) of x -- upload!, download! and fills go through it
function slice(x::HIPFermion, i5::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_spinor_slice, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}), x.h, i5 - 1, h))
    v = HIPFermion(h[], x.lat, WILSON, 0, x)
    finalizer(y -> ccall((:lqcd_spinor_destroy, LIB), Cint, (Ptr{Cvoid},), y.h), v)      # frees the handle only: the storage belongs to the parent
    return v
end
# Initialize_pseudofermion_fields(U[1], "Wilson"; nowing = true) / (U[1], "staggered") / (U[1], "Domainwall", L5 = L5, nowing = true)   (universe.jl:107,112,128)
function Initialize_pseudofermion_fields(u::HIPLink, name::String; L5 = 0, kwargs...)
    n = lowercase(name)
    kind = n == "wilson" ? WILSON : n == "staggered" ? STAGGERED : n == "domainwall" ? DOMAINWALL : error("$name is not supported")
    return HIPFermion(getfield(u, :parent).lat, kind; L5 = L5)
end
similar(x::HIPFermion) = HIPFermion(x.lat, x.kind; L5 = x.L5)
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
upload!(x::HIPFermion, a::Vector{Array{ComplexF64,6}}) = foreach(i5 -> upload!(slice(x, i5), a[i5]), 1:x.L5)        # five-dimensional: one array per slice
download!(a::Vector{Array{ComplexF64,6}}, x::HIPFermion) = foreach(i5 -> download!(a[i5], slice(x, i5)), 1:x.L5)
upload!(x::HIPFermion, a::Array{ComplexF64,6}) = check(ccall((:lqcd_spinor_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), x.h, a))
download!(a::Array{ComplexF64,6}, x::HIPFermion) = check(ccall((:lqcd_spinor_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), x.h, a))

# ------------------------------------------------------------------ Dirac operator
mutable struct HIPDirac <: Dirac_operator{4}
    h::Ptr{Cvoid}
    U::Vector{HIPLink}
    x::HIPFermion               # the field Dirac_operator(U, x, params) was built from (kind; similar(x) for the action's temporaries)
    dagger::Bool
    eps_CG::Float64
    MaxCGstep::Int
    method_CG::String
    owner::Bool
end
# Dirac_operator(U, x, params::Dict)  (universe.jl:137; keys universe.jl:103-135)
function Dirac_operator(U::Vector{HIPLink}, x::HIPFermion, params)
    name = params["Dirac_operator"]
    if name == "Domainwall"        # universe.jl:116-128: "mass" = Domainwall_m, "L5", "M" = Domainwall_M
        g = whole(U)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:lqcd_op_create_domainwall, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Float64, Float64, Cint, Ptr{Cint}),
                    g.lat.h, h, g.h, Float64(get(params, "M", -1.0)), Float64(params["mass"]), Cint(params["L5"]), Cint[get(params, "boundarycondition", [1, 1, 1, -1])...]))
        D = HIPDirac(h[], U, x, false, Float64(get(params, "eps_CG", 1e-19)), Int(get(params, "MaxCGstep", 3000)), "cg", true)
        finalizer(d -> d.owner && ccall((:lqcd_op_destroy, LIB), Cint, (Ptr{Cvoid},), d.h), D)
        return D
    end
    kind = name in ("Wilson", "WilsonClover") ? WILSON : name in ("Staggered", "staggered") ? STAGGERED : error("$name is not supported")
    km = kind == WILSON ? Float64(params["κ"]) : Float64(get(params, "mass", 0.5))
    r = Float64(get(params, "r", 1.0))
    bc = Cint[get(params, "boundarycondition", [1, 1, 1, -1])...]
    g = whole(U)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_op_create, LIB), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Cint, Ptr{Cvoid}, Float64, Float64, Ptr{Cint}),
                g.lat.h, h, kind, g.h, km, r, bc))
    if name == "WilsonClover"      # Clover_coefficient (src/system/parameter_structs.jl:125)
        check(ccall((:lqcd_op_set_clover, LIB), Cint, (Ptr{Cvoid}, Float64), h[], Float64(get(params, "Clover_coefficient", 1.5612))))
    end
    D = HIPDirac(h[], U, x, false, Float64(get(params, "eps_CG", 1e-19)), Int(get(params, "MaxCGstep", 3000)),
                 String(get(params, "method_CG", "bicgstab")), true)
    finalizer(d -> d.owner && ccall((:lqcd_op_destroy, LIB), Cint, (Ptr{Cvoid},), d.h), D)
    return D
end
# D(U): rebind links (unusedfiles/measure_chiral_condensate.jl:173)
function (D::HIPDirac)(U::Vector{HIPLink})
    check(ccall((:lqcd_op_set_gauge, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), D.h, whole(U).h)); D.U = U; D
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
function solve_mixed_DinvX!(y::HIPFermion, A::HIPDdagD, x::HIPFermion; inner_tol = 0.0)
    it, out, rr = Ref{Cint}(0), Ref{Cint}(0), Ref{Float64}(0)
    check(ccall((:lqcd_solve_mixed_cg_DdagD, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Float64, Ref{Cint}, Ref{Cint}, Ref{Float64}),
                A.D.h, y.h, x.h, A.D.eps_CG, A.D.MaxCGstep, inner_tol, it, out, rr))
end

# ---- pseudofermion action and force: FermiAction(D, Dict("Nf" => n)) (universe.jl:106-110,138) and the generics of standardMD.jl:95-96,
# standardHMC.jl:54,71 and AbstractMD.jl:129.  The action is a handle of the library (lqcd_action_*, csrc/rational.hip): which Nf is an exact action
# (Wilson 2; staggered 8, and 4 on the even sites) and which is the rational action S_f = η†(D†D)^(-Nf/n0)η (staggered Nf = 2, 3 of test/test_Nf2.toml:8,
# test/test_Nf3.toml:8, test/runtests.jl:114-130; Wilson Nf = 1), the spectral interval, the partial fractions (fitted by the library, where the
# reference's package brings Remez tables) and their refits are decided below the C ABI -- every method here is one ccall.
mutable struct HIPFermiAction <: FermiAction{4,HIPDirac,HIPFermion,HIPLink}
    h::Ptr{Cvoid}
    D::HIPDirac
    Nf::Float64
    _temporary_fermionfields::Vector{HIPFermion}     # standardMD.jl:50: η = similar(fermi_action._temporary_fermionfields[1])
    force::Vector{HIPLink}                           # G of lqcd_action_force, handed out per direction
end
const ACTION_KEYS = ("force_rational", "rhmc_lambda_min", "rhmc_lambda_max", "rhmc_tol_action", "rhmc_tol_MD", "rhmc_lanczos_steps")
function FermiAction(D::HIPDirac, parameters_action)
    keys = String[k for k in ACTION_KEYS if haskey(parameters_action, k)]
    vals = Float64[Float64(parameters_action[k]) for k in keys]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:lqcd_action_create, LIB), Cint, (Ptr{Cvoid}, Float64, Float64, Cint, Cint, Ptr{Cstring}, Ptr{Float64}, Ref{Ptr{Cvoid}}),
                D.h, Float64(get(parameters_action, "Nf", 0)), D.eps_CG, D.MaxCGstep, length(keys), keys, vals, h))
    fa = HIPFermiAction(h[], D, action_get(h[], "Nf"), [similar(D.x), similar(D.x)], similar(D.U))
    finalizer(a -> ccall((:lqcd_action_destroy, LIB), Cint, (Ptr{Cvoid},), a.h), fa)
    return fa
end
function action_get(h::Ptr{Cvoid}, key::String)
    v = Ref{Float64}(0)
    check(ccall((:lqcd_action_get, LIB), Cint, (Ptr{Cvoid}, Cstring, Ref{Float64}), h, key, v))
    return v[]
end
is_rational(fa::HIPFermiAction) = action_get(fa.h, "rational") != 0
# the operator's stopping rule may have been changed after the action was made (HIPDirac is mutable)
solver!(fa::HIPFermiAction) = check(ccall((:lqcd_action_set_solver, LIB), Cint, (Ptr{Cvoid}, Float64, Cint), fa.h, fa.D.eps_CG, fa.D.MaxCGstep))
# gauss_sampling_in_action!(ξ, U, fa) (standardMD.jl:95): ξ ~ exp(-ξ†ξ), i.e. re and im of variance 1/2
gauss_sampling_in_action!(ξ::HIPFermion, U::Vector{HIPLink}, fa::HIPFermiAction; seed = rand(UInt64)) =
    check(ccall((:lqcd_action_gauss_sampling, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64), fa.h, ξ.h, seed))
# sample_pseudofermions!(η, U, fa, ξ) (standardMD.jl:96): η = D†ξ (4 staggered tastes: restricted to the even sites); rational: η = (D†D)^(Nf/2n0) ξ
function sample_pseudofermions!(η::HIPFermion, U::Vector{HIPLink}, fa::HIPFermiAction, ξ::HIPFermion)
    solver!(fa)
    check(ccall((:lqcd_action_sample_pseudofermions, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), fa.h, whole(U).h, η.h, ξ.h))
    fa.D.U = U
    return η
end
# evaluate_FermiAction(fa, U, η) (standardHMC.jl:69,71): S_f = η†(D†D)^-1 η or its rational form; X = (D†D)^-1 η and Y = D X stay in the action's temporaries
function evaluate_FermiAction(fa::HIPFermiAction, U::Vector{HIPLink}, η::HIPFermion)
    S, it = Ref{Float64}(0), Ref{Cint}(0)
    X, Y = fa._temporary_fermionfields
    solver!(fa)
    check(ccall((:lqcd_action_evaluate, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Float64}, Ref{Cint}),
                fa.h, whole(U).h, η.h, X.h, Y.h, S, it))
    fa.D.U = U
    return S[]
end
# calc_UdSfdU!(UdSfdUμ, fa, U, η) (AbstractMD.jl:129) with UdSfdUμ = get_temp(temps, Dim): solve(s), Y = D X and the outer-product sweep(s) run
# resident into fa.force (= G, dS_f/dε[U -> exp(iεT)U] = -2 Im tr(T G)); each direction is handed over as "U dS_f/dU" = -G, the sign
# the caller's factor = -ϵ Δτ expects (AbstractMD.jl:127-132)
function calc_UdSfdU!(UdSfdUμ::Vector{HIPLink}, fa::HIPFermiAction, U::Vector{HIPLink}, η::HIPFermion)
    G = whole(fa.force)
    solver!(fa)
    check(ccall((:lqcd_action_force, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Cint}),
                fa.h, whole(U).h, G.h, η.h, C_NULL, C_NULL))
    fa.D.U = U
    for μ = 1:4
        check(ccall((:lqcd_link_scaled_copy, LIB), Cint, (Ptr{Cvoid}, Cint, Float64, Ptr{Cvoid}, Cint),
                    getfield(UdSfdUμ[μ], :parent).h, getfield(UdSfdUμ[μ], :slot), -1.0, G.h, μ - 1))
    end
end
# the coefficients: fitted by the library (rational_fit), or brought by the caller (e.g. Remez tables) for the building blocks below
function rational_fit(α, λmin, λmax; tol = 1e-10, max_poles = 40)
    a0, n, err = Ref{Float64}(0), Ref{Cint}(0), Ref{Float64}(0)
    res, poles = zeros(max_poles), zeros(max_poles)
    check(ccall((:lqcd_rational_fit, LIB), Cint, (Float64, Float64, Float64, Float64, Cint, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Cint}, Ref{Float64}),
                α, λmin, λmax, tol, max_poles, a0, res, poles, n, err))
    return a0[], res[1:n[]], poles[1:n[]], err[]
end
function estimate_spectrum(A::HIPDdagD; steps = 60, seed = 4711)
    lo, hi = Ref{Float64}(0), Ref{Float64}(0)
    check(ccall((:lqcd_estimate_spectrum, LIB), Cint, (Ptr{Cvoid}, Cint, UInt64, Ref{Float64}, Ref{Float64}), A.D.h, steps, seed, lo, hi))
    return lo[], hi[]
end
# Ritz value `index` (1-based, ascending) of a Lanczos tridiagonal and |last eigenvector component|: β_n times it bounds the distance to an eigenvalue
function tridiag_ritz(diag::Vector{Float64}, offdiag::Vector{Float64}, index::Integer)
    θ, s = Ref{Float64}(0), Ref{Float64}(0)
    check(ccall((:lqcd_tridiag_ritz, LIB), Cint, (Cint, Ptr{Float64}, Ptr{Float64}, Cint, Ref{Float64}, Ref{Float64}), length(diag), diag, offdiag, index - 1, θ, s))
    return θ[], s[]
end
rational_apply!(y::HIPFermion, D::HIPDirac, x::HIPFermion, a0, res::Vector{Float64}, poles::Vector{Float64}) =
    check(ccall((:lqcd_rational_apply, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Cint, Ptr{Float64}, Ptr{Float64}, Float64, Cint, Ptr{Cint}),
                D.h, y.h, x.h, a0, length(res), res, poles, D.eps_CG, D.MaxCGstep, C_NULL))
rational_force!(G::Vector{HIPLink}, D::HIPDirac, φ::HIPFermion, res::Vector{Float64}, poles::Vector{Float64}) =
    check(ccall((:lqcd_rational_force, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Float64, Cint, Ptr{Cint}),
                D.h, whole(G).h, φ.h, length(res), res, poles, D.eps_CG, D.MaxCGstep, C_NULL))

# ---- fused four-direction forms of the MD step (one kernel each; what a maintainer would call from a specialised
# P_update!(U::Vector{HIPLink}, p, ϵ, md) / U_update! method to skip the per-direction temporaries)
P_update_fused!(U::Vector{HIPLink}, p::Vector{HIPTALink}, factor, β) =      # p += factor * TA(-(β/6) U * staples), the force field is never stored
    check(ccall((:lqcd_momentum_add_gauge_force, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}, Float64), whole(p).h, factor, whole(U).h, β))
U_update_fused!(U::Vector{HIPLink}, p::Vector{HIPTALink}, dt) =             # U <- exp(dt p) U (reprojected onto SU(3): tunable md_reunitarize)
    check(ccall((:lqcd_gauge_exp_update, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), whole(U).h, dt, whole(p).h))
momentum_add_ta_fused!(p::Vector{HIPTALink}, factor, G::Vector{HIPLink}) =  # p += factor * TA(G), all four directions
    check(ccall((:lqcd_momentum_add_ta, LIB), Cint, (Ptr{Cvoid}, Float64, Ptr{Cvoid}), whole(p).h, factor, whole(G).h))

end # module
