# vSmartMOMROCmExt.jl -- reference-side binding for libvsmartmom_hip.so (NOT executed in this repo:
# Julia is not available in the build image).  It mirrors the method table of
# ext/vSmartMOMCUDAExt.jl + ext/gpu_batched_cuda.jl for an AMD array type `ROCArray`
# (AMDGPU.jl provides allocation / H2D / D2H only; no rocBLAS, no KernelAbstractions).
#
# Drop it under ext/, add to Project.toml:  [weakdeps] AMDGPU = "..."; [extensions] vSmartMOMROCmExt = "AMDGPU"
module vSmartMOMROCmExt

using AMDGPU
using vSmartMOM
import vSmartMOM.Architectures: devi, array_type, architecture, GPU, _has_cuda, _sync_gpu
import vSmartMOM.CoreRT: batched_mul, batch_inv!, batch_solve!, batched_pointer_cache,
                         elemental!, doubling!, interaction!, AddedLayer, CompositeLayer,
                         ScatteringInterface_00, ScatteringInterface_01, ScatteringInterface_10,
                         ScatteringInterface_11

const libvsm = get(ENV, "VSMARTMOM_HIP_LIB", "libvsmartmom_hip.so")

@inline function _chk(rc::Cint)
    rc == 0 || error("libvsmartmom_hip: status $rc: " *
                     unsafe_string(ccall((:vsm_last_error, libvsm), Cstring, ())))
    nothing
end
_stream() = Ptr{Cvoid}(AMDGPU.stream().stream)           # hipStream_t of the current task
_p(A::ROCArray) = Ptr{Cvoid}(pointer(A))

# ---- Architectures.jl:68-96 ----------------------------------------------------------------
array_type(::GPU) = ROCArray
architecture(::ROCArray) = GPU()
function __init__()
    n = Ref{Cint}(0)
    if ccall((:vsm_device_count, libvsm), Cint, (Ref{Cint},), n) == 0 && n[] > 0
        _has_cuda[] = true                                 # "a GPU backend is present"
        _sync_gpu[] = () -> _chk(ccall((:vsm_sync, libvsm), Cint, (Ptr{Cvoid},), _stream()))
    else
        @warn "vSmartMOMROCmExt: no MI355X visible; staying on CPU"   # vSmartMOMCUDAExt.jl:59-62
    end
end

# ---- gpu_batched_cuda.jl:208-233 ---------------------------------------------------------------
for (FT, sfx) in ((Float64, :f64), (Float32, :f32))
    mul = Symbol(:vsm_batched_mul_, sfx); inv = Symbol(:vsm_batch_inv_, sfx)
    ia  = Symbol(:vsm_interaction_, sfx); ed = Symbol(:vsm_elemental_doubling_, sfx)
    @eval begin
        function batched_mul(A::ROCArray{$FT,3}, B::ROCArray{$FT,3})
            M, K, S = size(A); Nc = size(B, 2)
            C = ROCArray{$FT}(undef, M, Nc, S)
            _chk(ccall(($(QuoteNode(mul)), libvsm), Cint,
                       (Cint, Cint, Cint, Cint, Ptr{Cvoid}, Clonglong, Ptr{Cvoid}, Clonglong, Ptr{Cvoid}, Ptr{Cvoid}),
                       M, Nc, K, S, _p(A), M * K, _p(B), size(B, 3) == 1 ? 0 : K * Nc, _p(C), _stream()))
            C
        end
        # gpu_batched_cuda.jl:97-182 (all three call forms land here; A is not clobbered)
        function batch_inv!(X::ROCArray{$FT,3}, A::ROCArray{$FT,3}, args...)
            N, _, S = size(A)
            _chk(ccall(($(QuoteNode(inv)), libvsm), Cint,
                       (Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}, Ptr{Cvoid}),
                       N, S, _p(A), _p(X), C_NULL, _stream()))
            X
        end
    end
end
batched_pointer_cache(::ROCArray) = nothing                # route to the 2-arg inverse

# ---- CoreKernel overrides (north_star: no KernelAbstractions on this backend) --------------------
struct VsmAdded;     r_mp::Ptr{Cvoid}; t_pp::Ptr{Cvoid}; r_pm::Ptr{Cvoid}; t_mm::Ptr{Cvoid}
                     j0_p::Ptr{Cvoid}; j0_m::Ptr{Cvoid}; mat_stride::Clonglong
                     d_symmetric::Cint; reserved::Cint end          # = vsm_added_f64 / _f32
struct VsmComposite; R_mp::Ptr{Cvoid}; R_pm::Ptr{Cvoid}; T_pp::Ptr{Cvoid}; T_mm::Ptr{Cvoid}
                     J0_p::Ptr{Cvoid}; J0_m::Ptr{Cvoid} end
_c(a::AddedLayer) = VsmAdded(_p(a.r⁻⁺), _p(a.t⁺⁺), _p(a.r⁺⁻), _p(a.t⁻⁻), _p(a.j₀⁺), _p(a.j₀⁻),
                             size(a.r⁻⁺, 3) == 1 ? 0 : size(a.r⁻⁺, 1)^2, 0, 0)
_c(c::CompositeLayer) = VsmComposite(_p(c.R⁻⁺), _p(c.R⁺⁻), _p(c.T⁺⁺), _p(c.T⁻⁻), _p(c.J₀⁺), _p(c.J₀⁻))
_tag(::ScatteringInterface_00) = 0; _tag(::ScatteringInterface_01) = 1
_tag(::ScatteringInterface_10) = 2; _tag(::ScatteringInterface_11) = 3

# interaction.jl:268-285
function interaction!(iface, SFI, c::CompositeLayer{FT}, a::AddedLayer{FT}, I_static) where {FT}
    N, _, S = size(c.R⁻⁺)
    f = FT === Float64 ? :vsm_interaction_f64 : :vsm_interaction_f32
    _chk(ccall((f, libvsm), Cint, (Cint, Cint, Cint, Ref{VsmComposite}, Ref{VsmAdded}, Ptr{Cvoid}, Ptr{Cvoid}),
               _tag(iface), N, S, _c(c), _c(a), C_NULL, _stream()))
end
# elemental! (elemental.jl:174-230) stores its inputs; doubling! (doubling.jl:112-131) then launches the fused
# vsm_elemental_doubling_* with them -- see INTEGRATION.md for the two-line patch in rt_kernel!.

# rt_kernel!(::noRS) scattering branch (rt_kernel.jl:204-249) as ONE call: elemental! + doubling! + (TOA copy | interaction!(_11))
struct VsmQuad{FT}; mu::Ptr{Cvoid}; wt::Ptr{Cvoid}; N::Cint; n_stokes::Cint; i_mu0::Cint; mu0::FT end
function layer_forward!(q::VsmQuad{Float64}, nSpec, m, ndoubl, dτ, ϖ, τ_sum, F₀, Z⁺⁺, Z⁻⁺, iz,
                        c::CompositeLayer{Float64}, a::AddedLayer{Float64})
    zs = size(Z⁺⁺, 3) == 1 ? 0 : size(Z⁺⁺, 1)^2
    _chk(ccall((:vsm_layer_forward_f64, libvsm), Cint,
               (Ref{VsmQuad{Float64}}, Cint, Cint, Cint, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid},
                Clonglong, Cint, Ref{VsmComposite}, Ref{VsmAdded}, Ptr{Cvoid}),
               q, nSpec, m, ndoubl, _p(dτ), _p(ϖ), _p(τ_sum), _p(F₀), _p(Z⁺⁺), _p(Z⁻⁺), zs, iz == 1 ? 1 : 0, _c(c), _c(a),
               _stream()))
end

# Rotational Raman (CoreKernel/*_inelastic.jl): the 4-D arrays keep the reference layout [N,N,nSpec,nRaman]
struct VsmAddedRS;     ier_mp::Ptr{Cvoid}; iet_pp::Ptr{Cvoid}; ier_pm::Ptr{Cvoid}; iet_mm::Ptr{Cvoid}
                       ieJ0_p::Ptr{Cvoid}; ieJ0_m::Ptr{Cvoid}; K::Cint; reserved::Cint end
struct VsmCompositeRS; ieR_mp::Ptr{Cvoid}; ieR_pm::Ptr{Cvoid}; ieT_pp::Ptr{Cvoid}; ieT_mm::Ptr{Cvoid}
                       ieJ0_p::Ptr{Cvoid}; ieJ0_m::Ptr{Cvoid}; K::Cint; reserved::Cint end
_c_rs(a) = VsmAddedRS(_p(a.ier⁻⁺), _p(a.iet⁺⁺), _p(a.ier⁺⁻), _p(a.iet⁻⁻), _p(a.ieJ₀⁺), _p(a.ieJ₀⁻), size(a.ier⁻⁺, 4), 0)
_C_rs(c) = VsmCompositeRS(_p(c.ieR⁻⁺), _p(c.ieR⁺⁻), _p(c.ieT⁺⁺), _p(c.ieT⁻⁻), _p(c.ieJ₀⁺), _p(c.ieJ₀⁻), size(c.ieR⁻⁺, 4), 0)
# interaction!(RS_type::RRS, ::ScatteringInterface_11, ...) (interaction_inelastic.jl:683-700); i_λ₁λ₀_dev = ROCArray{Cint}
function interaction_rrs!(i_λ₁λ₀_dev, c, a, work::ROCArray{Float64})
    N, _, S = size(c.R⁻⁺)
    _chk(ccall((:vsm_interaction_inelastic_rrs_f64, libvsm), Cint,
               (Cint, Cint, Cint, Ptr{Cvoid}, Ref{VsmComposite}, Ref{VsmCompositeRS}, Ref{VsmAdded}, Ref{VsmAddedRS},
                Ptr{Cvoid}, Ptr{Cvoid}),
               3, N, S, _p(i_λ₁λ₀_dev), _c(c), _C_rs(c), _c(a), _c_rs(a), _p(work), _stream()))
end
end # module
