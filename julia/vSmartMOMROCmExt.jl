# vSmartMOMROCmExt.jl -- reference-side binding of libvsmartmom_hip.so (C ABI: include/vsmartmom_hip.h).
# NOT executed in this repo (no Julia in the build image); it is the method table a maintainer drops under ext/ --
# the counterpart of ext/vSmartMOMCUDAExt.jl + ext/gpu_batched_cuda.jl for an AMD array type (`ROCArray`; AMDGPU.jl
# provides allocation / H2D / D2H only: no rocBLAS, no KernelAbstractions).  Project.toml: [weakdeps] AMDGPU,
# [extensions] vSmartMOMROCmExt = "AMDGPU".  Every method is a plain `ccall`; the kernels behind an entry point are
# chosen inside the library from N and the float type.  rt_run(model) and rt_run(model, lin_model, NAer, NGas, NSurf)
# reach these methods through the reference's own drivers (rt_run.jl:383-517, rt_run_lin.jl:200-322): see INTEGRATION.md.
module vSmartMOMROCmExt

using AMDGPU
using Libdl
using LinearAlgebra
using vSmartMOM
import vSmartMOM.Architectures: devi, array_type, architecture, GPU, _has_cuda, _sync_gpu
import vSmartMOM.CoreRT: batched_mul, batch_inv!, batch_solve!, batched_pointer_cache, elemental!, doubling!,
    doubling_allparams!, interaction!, rt_kernel!, create_surface_layer!, postprocessing_vza!, copy_added_to_composite!,
    AddedLayer, CompositeLayer, AddedLayerLin, CompositeLayerLin, AddedLayerRS, CompositeLayerRS, noRS, RRS,
    LambertianSurfaceScalar, CoxMunkSurface, QuadPoints, CoreScatteringOpticalProperties,
    CoreScatteringOpticalPropertiesLin, ScatteringInterface_00, ScatteringInterface_01, ScatteringInterface_10,
    ScatteringInterface_11, get_dtau_ndoubl, _get_n_water, expandOpticalProperties

const libvsm = get(ENV, "VSMARTMOM_HIP_LIB", "libvsmartmom_hip.so")
const FTs = Union{Float32,Float64}
const PV = Ptr{Cvoid}
_sfx(::Type{Float64}) = "f64"
_sfx(::Type{Float32}) = "f32"
_fn(base, FT) = Symbol(base, "_", _sfx(FT))            # e.g. :vsm_interaction_f64 (resolved by dlsym at call time)
@inline function _chk(rc::Cint)
    rc == 0 || error("libvsmartmom_hip: status $rc: " * unsafe_string(ccall((:vsm_last_error, libvsm), Cstring, ())))
    nothing
end
# ccall wants a LITERAL tuple of argument types (a tuple-valued variable does not lower: "ccall argument types must be a
# tuple"), while the function may be a run-time pointer.  `@vsm "base" FT (types...) args...` therefore splices the literal
# tuple into a ccall on the dlsym pointer of base_f64 / base_f32 (static parameters such as FT are allowed inside the tuple).
const _handle = Ref{Ptr{Cvoid}}(C_NULL)
const _symcache = Dict{Symbol,Ptr{Cvoid}}()
function _sym(f::Symbol)
    get!(_symcache, f) do
        _handle[] == C_NULL && (_handle[] = Libdl.dlopen(libvsm))
        Libdl.dlsym(_handle[], f)
    end
end
macro vsm(base, FT, types, args...)
    (types isa Expr && types.head === :tuple) || error("@vsm: the argument types must be a literal tuple")
    :(_chk(ccall(_sym(_fn($(esc(base)), $(esc(FT)))), Cint, $(esc(types)), $(map(esc, args)...))))
end
_stream() = PV(AMDGPU.stream().stream)                  # hipStream_t of the current task
# Streams and devices (include/vsmartmom_hip.h, Conventions): every call enqueues on the stream it is given and nowhere else; tasks
# on different streams -- or on different devices of one process (AMDGPU.device!(d) in the task, streams of that device) -- are
# independent: the library keeps its scratch per (device, stream) and its kernel attributes per device.
release_scratch() = _chk(ccall((:vsm_release_scratch, libvsm), Cint, ()))            # frees the current device's library scratch
build_id() = unsafe_string(ccall((:vsm_build_id, libvsm), Cstring, ()))               # source hash of the loaded library
# The in-kernel inverses of an asynchronous launch cannot return `info`; they raise device flags.  Call after the
# synchronisation that ends a run (rt_run does): a singular (I - R r) then throws like the CPU path's LU (cpu_batched.jl:32-47).
function check_device_status(what = "rt_run")
    flags = zeros(Cint, 4)
    _chk(ccall((:vsm_device_status, libvsm), Cint, (Ptr{Cint}, Cint, PV), flags, 1, _stream()))
    (flags[1] & 1) != 0 && throw(LinearAlgebra.SingularException(0))
    (flags[1] & 2) != 0 && error("$what: NaN / Inf operand in an in-kernel inverse")
    (flags[1] & 4) != 0 && error("$what: a phase matrix handed to vsm_run_layer couples Stokes components outside the declared mask (VSM_DEVSTAT_MASK)")
    flags
end
_p(A::ROCArray) = PV(pointer(A))
_p(::Nothing) = C_NULL

# ---- Architectures.jl:68-96 / vSmartMOMCUDAExt.jl:21-27,46-57 ------------------------------------------------------
array_type(::GPU) = ROCArray
architecture(::ROCArray) = GPU()
function __init__()
    n = Ref{Cint}(0)
    if ccall((:vsm_device_count, libvsm), Cint, (Ref{Cint},), n) == 0 && n[] > 0
        _has_cuda[] = true
        # the synchronisation hook that ends a run (Architectures.synchronize_if_gpu) also reads the device flags of the in-kernel
        # inverses: a singular (I - R r) throws here, like the LU of the reference's CPU path at the call (cpu_batched.jl:32-47)
        _sync_gpu[] = () -> (_chk(ccall((:vsm_sync, libvsm), Cint, (PV,), _stream())); check_device_status("synchronize_if_gpu"); nothing)
    else
        @warn "vSmartMOMROCmExt: no MI355X visible; staying on CPU"      # vSmartMOMCUDAExt.jl:59-62
    end
end

# ---- C structs (include/vsmartmom_hip.h) -------------------------------------------------------------------------------
struct VsmQuad{FT}; mu::PV; wt::PV; N::Cint; n_stokes::Cint; i_mu0::Cint; mu0::FT end
struct VsmAdded; r_mp::PV; t_pp::PV; r_pm::PV; t_mm::PV; j0_p::PV; j0_m::PV; mat_stride::Clonglong; d_symmetric::Cint; reserved::Cint end
struct VsmComposite; R_mp::PV; R_pm::PV; T_pp::PV; T_mm::PV; J0_p::PV; J0_m::PV end
struct VsmAddedLin; r_mp::PV; t_pp::PV; r_pm::PV; t_mm::PV; J0_p::PV; J0_m::PV; P::Cint; reserved::Cint; mat_stride::Clonglong end
struct VsmCompositeLin; R_mp::PV; R_pm::PV; T_pp::PV; T_mm::PV; J0_p::PV; J0_m::PV; P::Cint; reserved::Cint end
struct VsmAddedRS; ier_mp::PV; iet_pp::PV; ier_pm::PV; iet_mm::PV; ieJ0_p::PV; ieJ0_m::PV; K::Cint; reserved::Cint end
struct VsmCompositeRS; ieR_mp::PV; ieR_pm::PV; ieT_pp::PV; ieT_mm::PV; ieJ0_p::PV; ieJ0_m::PV; K::Cint; reserved::Cint end
struct VsmRRS; shift::PV; varpi_ie::PV; fscatt::PV; Zpp::PV; Zmp::PV end
struct VsmCoxMunk{FT}; wind_speed::FT; n_re::FT; n_im::FT; whitecap_albedo::FT; include_whitecaps::Cint; shadowing::Cint end
_ms(A) = size(A, 3) == 1 ? 0 : size(A, 1)^2             # slice stride; 0 = one block shared by all spectral points
_q(qp::QuadPoints, n, ::Type{FT}) where {FT} = VsmQuad{FT}(_p(qp.qp_μN), _p(qp.wt_μN), length(qp.qp_μN), n, qp.iμ₀ - 1, FT(qp.μ₀))
_c(a::AddedLayer) = VsmAdded(_p(a.r⁻⁺), _p(a.t⁺⁺), _p(a.r⁺⁻), _p(a.t⁻⁻), _p(a.j₀⁺), _p(a.j₀⁻), _ms(a.r⁻⁺), 0, 0)
_c(c::CompositeLayer) = VsmComposite(_p(c.R⁻⁺), _p(c.R⁺⁻), _p(c.T⁺⁺), _p(c.T⁻⁻), _p(c.J₀⁺), _p(c.J₀⁻))
_c(a::AddedLayerLin) = VsmAddedLin(_p(a.ap_ṙ⁻⁺), _p(a.ap_ṫ⁺⁺), _p(a.ap_ṙ⁺⁻), _p(a.ap_ṫ⁻⁻), _p(a.ap_J̇₀⁺), _p(a.ap_J̇₀⁻),
                                   size(a.ap_ṙ⁻⁺, 4), 0, _ms(a.ap_ṙ⁻⁺))
_c(c::CompositeLayerLin) = VsmCompositeLin(_p(c.Ṙ⁻⁺), _p(c.Ṙ⁺⁻), _p(c.Ṫ⁺⁺), _p(c.Ṫ⁻⁻), _p(c.J̇₀⁺), _p(c.J̇₀⁻), size(c.Ṙ⁻⁺, 4), 0)
_crs(a) = VsmAddedRS(_p(a.ier⁻⁺), _p(a.iet⁺⁺), _p(a.ier⁺⁻), _p(a.iet⁻⁻), _p(a.ieJ₀⁺), _p(a.ieJ₀⁻), size(a.ier⁻⁺, 4), 0)
_Crs(c) = VsmCompositeRS(_p(c.ieR⁻⁺), _p(c.ieR⁺⁻), _p(c.ieT⁺⁺), _p(c.ieT⁻⁻), _p(c.ieJ₀⁺), _p(c.ieJ₀⁻), size(c.ieR⁻⁺, 4), 0)
_tag(::ScatteringInterface_00) = 0; _tag(::ScatteringInterface_01) = 1
_tag(::ScatteringInterface_10) = 2; _tag(::ScatteringInterface_11) = 3
_work(FT, f::Symbol, a, b) = ROCArray{FT}(undef, max(1, Int(ccall(_sym(f), Csize_t, (Cint, Cint), a, b))))
_work(FT, f::Symbol, a, b, c) = ROCArray{FT}(undef, max(1, Int(ccall(_sym(f), Csize_t, (Cint, Cint, Cint), a, b, c))))

# ---- L1 operator API: gpu_batched_cuda.jl:65-233 -----------------------------------------------------------------------
function batched_mul(A::ROCArray{FT,3}, B::ROCArray{FT,3}) where {FT<:FTs}
    M, K, S = size(A); Nc = size(B, 2)
    C = ROCArray{FT}(undef, M, Nc, S)
    @vsm("vsm_batched_mul", FT, (Cint, Cint, Cint, Cint, PV, Clonglong, PV, Clonglong, PV, PV),
          M, Nc, K, S, _p(A), M * K, _p(B), size(B, 3) == 1 && S > 1 ? 0 : K * Nc, _p(C), _stream())
    C
end
function batch_inv!(X::ROCArray{FT,3}, A::ROCArray{FT,3}, args...) where {FT<:FTs}     # all three call forms; A intact
    @vsm("vsm_batch_inv", FT, (Cint, Cint, PV, PV, PV, PV), size(A, 1), size(A, 3), _p(A), _p(X), C_NULL, _stream()); X
end
function batch_solve!(X::ROCArray{FT,3}, A::ROCArray{FT,3}, B::ROCArray{FT,3}) where {FT<:FTs}
    @vsm("vsm_batch_solve", FT, (Cint, Cint, Cint, PV, PV, PV, PV, PV, PV), size(A, 1), size(B, 2), size(A, 3),
          _p(A), _p(B), _p(X), _p(similar(A)), C_NULL, _stream()); X
end
batched_pointer_cache(::ROCArray) = nothing              # route batch_inv! to the 2-argument form

# ---- L2 CoreKernel, forward (north_star: host-level functions are overridden, no KernelAbstractions kernels run) ------
# elemental! (elemental.jl:174-230)
function elemental!(pol_type, SFI::Bool, τ_sum::ROCArray, dτ::ROCArray, F₀::ROCArray, p::CoreScatteringOpticalProperties,
                    m::Int, ndoubl::Int, scatter::Bool, qp::QuadPoints, a::AddedLayer{FT}, arch) where {FT<:FTs}
    @vsm("vsm_elemental", FT, (Ref{VsmQuad{FT}}, Cint, Cint, Cint, PV, PV, PV, PV, PV, PV, Clonglong, Ref{VsmAdded}, PV),
          _q(qp, pol_type.n, FT), length(dτ), m, ndoubl, _p(dτ), _p(p.ϖ), _p(τ_sum), _p(F₀), _p(p.Z⁺⁺), _p(p.Z⁻⁺), _ms(p.Z⁺⁺), _c(a), _stream())
end
# doubling! (doubling.jl:121-131); expk is squared in place ndoubl times like the reference
function doubling!(pol_type, SFI, expk::ROCArray{FT}, ndoubl::Int, a::AddedLayer, I_static, arch) where {FT<:FTs}
    N, _, S = size(a.r⁻⁺)
    @vsm("vsm_doubling", FT, (Cint, Cint, Cint, Cint, PV, Ref{VsmAdded}, PV, PV), N, pol_type.n, S, ndoubl, _p(expk), _c(a),
          _p(_work(FT, :vsm_doubling_work_elems, N, S)), _stream())
end
# interaction! (interaction.jl:278-285); `work` is required off the fused "11" path, so it is always supplied
function interaction!(iface, SFI, c::CompositeLayer{FT}, a::AddedLayer{FT}, I_static) where {FT<:FTs}
    N, _, S = size(materialize!(c).R⁻⁺)                  # (the surface interaction of rt_run.jl:455-470 is the usual first consumer)
    @vsm("vsm_interaction", FT, (Cint, Cint, Cint, Ref{VsmComposite}, Ref{VsmAdded}, PV, PV), _tag(iface), N, S, _c(c), _c(a),
          _p(_work(FT, :vsm_interaction_work_elems, N, S)), _stream())
end
function copy_added_to_composite!(c::CompositeLayer{FT}, a::AddedLayer{FT}) where {FT<:FTs}
    _drop_native!(c)
    @vsm("vsm_copy_added_to_composite", FT, (Cint, Cint, Ref{VsmAdded}, Ref{VsmComposite}, PV), size(c.R⁻⁺, 1), size(c.R⁻⁺, 3), _c(a), _c(c), _stream())
end
# ---- the native-layout run behind the reference's OWN driver: a per-CompositeLayer registry ---------------------------------------
# rt_run.jl:383-453 loops `for m` outside `for iz` and hands rt_kernel! ONE CompositeLayer; nothing else reads that composite until
# the surface interaction (rt_run.jl:455-470).  rt_kernel!(::noRS) therefore keeps it in the layer kernels' strip layout
# (vsm_run_*: FP64 / FP32 storage, every block of coupled Stokes components <= 64 rows) from the TOA call -- where a one-moment run
# is created for it -- to the first method of this extension that needs the reference's [N,N,S] arrays (interaction!,
# copy_added_to_composite!, postprocessing_vza!: they call materialize! first).  No change to rt_run.jl.  The blocks of the run are
# the connected Stokes components of the phase matrices seen so far (vsm_stokes_coupling on the layer's Z: at m = 0 no phase matrix
# couples (I,Q) with (U,V), compute_Z_matrices.jl:26-110); a layer that couples two blocks re-opens the run under the wider mask
# (export -> create -> import); vsm_run_layer checks every Z against the masks on the device (VSM_DEVSTAT_MASK).  The Python host
# mirrors this registry line by line (vsmartmom.jl_amd/core_rt.py: _rt_kernel_native, CompositeLayer.materialize) and is what the
# -m gpu tests and bench.py's C2-dropin-native entry drive, since no Julia exists in the build image.
const NATIVE_DROPIN = Ref(true)
mutable struct NativeSlot
    handle::PV
    q::Any                       # VsmQuad{FT} (the run keeps its mu / wt pointers: qp outlives the run)
    m::Int
    coupling::Cint
    grp::Vector{Int}             # block of each Stokes component under `coupling`
    F₀::Any                      # device copy of RS.F₀ (the reference keeps it on the host, rt_run.jl:369-373)
end
const _native = IdDict{Any,NativeSlot}()                 # CompositeLayer => its native copy
const _native_ws = IdDict{Any,ROCArray{Float64,1}}()     # CompositeLayer => workspace (grow-only, reused from moment to moment)
_dev(x::ROCArray) = x
_dev(x::AbstractArray) = ROCArray(x)
function _groups(ns::Int, mask::Integer)                 # the rule of vsm_run_create: connected components of the symmetrised mask
    adj(a, b) = a == b || mask < 0 || (mask >> (4a + b)) & 1 == 1 || (mask >> (4b + a)) & 1 == 1
    grp = fill(-1, ns); ng = 0
    for a in 0:ns-1
        grp[a+1] >= 0 && continue
        grp[a+1] = ng; stack = [a]
        while !isempty(stack)
            x = pop!(stack)
            for b in 0:ns-1
                if adj(x, b) && grp[b+1] < 0
                    grp[b+1] = ng; push!(stack, b)
                end
            end
        end
        ng += 1
    end
    grp
end
function _layer_coupling(p::CoreScatteringOpticalProperties, N::Int, n::Int, ::Type{FT}) where {FT<:FTs}
    nb = size(p.Z⁺⁺, 3); mask = AMDGPU.zeros(Cint, nb); acc = Cint(0)
    for b0 in 1:65535:nb                                  # (vsm_stokes_coupling: at most 65535 matrices per call)
        b1 = min(nb, b0 + 65534); off = (b0 - 1) * N * N * sizeof(FT)
        @vsm("vsm_stokes_coupling", FT, (Cint, Cint, Cint, PV, PV, PV, PV), N, n, b1 - b0 + 1, _p(p.Z⁺⁺) + off, _p(p.Z⁻⁺) + off,
              PV(pointer(mask)) + (b0 - 1) * sizeof(Cint), _stream())
    end
    for v in Array(mask); acc |= v; end                   # (one small D2H; the reference's own rt_kernel! reads maximum(τ .* ϖ) per layer)
    acc
end
function _native_open!(c::CompositeLayer{FT}, qp::QuadPoints, n::Int, m::Int, mask::Cint, F₀, import_arrays::Bool) where {FT<:FTs}
    N, _, S = size(c.R⁻⁺)
    # (Float32 models: FP32 records and arithmetic, blocks of up to 128 rows -- vsm_run_supported_f32 / vsm_run_workspace_bytes_f32)
    f_sup = FT === Float32 ? :vsm_run_supported_f32 : :vsm_run_supported
    f_ws = FT === Float32 ? :vsm_run_workspace_bytes_f32 : :vsm_run_workspace_bytes
    ccall(_sym(f_sup), Cint, (Cint, Cint, Cint), N, n, mask) == 1 || return false
    cm = Cint[mask]; ms = Cint[m]
    nbytes = ccall(_sym(f_ws), Csize_t, (Cint, Cint, Cint, Cint, Ptr{Cint}), N, n, S, 1, cm)
    ws = get(_native_ws, c, nothing)
    if ws === nothing || sizeof(ws) < nbytes
        ws = _native_ws[c] = ROCArray{Float64}(undef, max(2, cld(Int(nbytes), 8)))
    end
    q = _q(qp, n, FT); h = Ref{PV}(C_NULL)
    @vsm("vsm_run_create", FT, (Ref{VsmQuad{FT}}, Cint, Cint, Ptr{Cint}, Ptr{Cint}, PV, Csize_t, Ref{PV}), q, S, 1, ms, cm, _p(ws), nbytes, h)
    import_arrays && @vsm("vsm_run_import", FT, (PV, Ref{VsmComposite}, PV), h[], _c(c), _stream())
    _native[c] = NativeSlot(h[], q, m, mask, _groups(n, mask), F₀)
    true
end
# the composite as the reference's arrays again: every method below that touches a CompositeLayer calls this first
function materialize!(c::CompositeLayer{FT}) where {FT<:FTs}
    slot = pop!(_native, c, nothing)
    slot === nothing && return c
    try
        @vsm("vsm_run_export", FT, (PV, Ref{VsmComposite}, PV), slot.handle, _c(c), _stream())
    finally
        ccall(_sym(:vsm_run_destroy), Cint, (PV,), slot.handle)
    end
    c
end
function _drop_native!(c)
    slot = pop!(_native, c, nothing)
    slot === nothing || ccall(_sym(:vsm_run_destroy), Cint, (PV,), slot.handle)
    nothing
end
# the scattering "11" / TOA branch on the native copy; false = not taken (the caller continues on the reference's arrays)
function _rt_kernel_native!(RS::noRS{FT}, pol_type, c::CompositeLayer{FT}, p::CoreScatteringOpticalProperties, τ_sum::ROCArray, m::Int,
                            qp::QuadPoints, iz::Int, dτ::ROCArray, ndoubl::Int) where {FT<:FTs}
    N = size(c.R⁻⁺, 1); n = pol_type.n
    lc = _layer_coupling(p, N, n, FT)
    if iz == 1
        _drop_native!(c)                                  # copy_added_to_composite! overwrites whatever the composite held
        _native_open!(c, qp, n, m, lc, _dev(RS.F₀), false) || return false
    end
    slot = get(_native, c, nothing)
    (slot === nothing || slot.m != m) && return false
    g = slot.grp
    if any((lc >> (4a + b)) & 1 == 1 && g[a+1] != g[b+1] for a in 0:n-1, b in 0:n-1)
        wider = slot.coupling | lc; F₀ = slot.F₀
        materialize!(c)
        _native_open!(c, qp, n, m, wider, F₀, true) || return false
        slot = _native[c]
    end
    zpp = PV[_p(p.Z⁺⁺)]; zmp = PV[_p(p.Z⁻⁺)]; lcv = Cint[lc]        # (bound to names: ccall roots its array arguments for the call)
    @vsm("vsm_run_layer", FT, (PV, Cint, PV, PV, PV, PV, Cint, Ptr{PV}, Ptr{PV}, Clonglong, PV, Cint, Ptr{Cint}, PV),
          slot.handle, ndoubl, _p(dτ), _p(p.ϖ), _p(τ_sum), _p(slot.F₀), 0, zpp, zmp, _ms(p.Z⁺⁺), C_NULL, iz == 1 ? 1 : 0, lcv, _stream())
    true
end
# expandOpticalProperties (compEffectiveLayerProperties.jl:106-117) for this array type: a phase matrix that is the same for all
# spectral points stays ONE N x N block on the device (every entry point takes a slice stride, 0 = shared: `_ms`), instead of nSpec
# host-side copies and their H2D per layer and moment (2 N^2 nSpec sizeof(FT): 576 MB at N = 60, 10^4 points)
const SHARE_Z = Ref(true)
function expandOpticalProperties(in::CoreScatteringOpticalProperties, arr_type::Type{<:ROCArray})
    (; τ, ϖ, Z⁺⁺, Z⁻⁺) = in
    @assert length(τ) == length(ϖ) "τ and ϖ sizes need to match"
    if size(Z⁺⁺, 3) == 1 && !SHARE_Z[]
        Z⁺⁺ = repeat(Z⁺⁺, 1, 1, length(τ)); Z⁻⁺ = repeat(Z⁻⁺, 1, 1, length(τ))
    end
    @assert size(Z⁺⁺, 3) in (1, length(τ)) "Z and τ dimensions need to match"
    CoreScatteringOpticalProperties(arr_type(τ), arr_type(ϖ), arr_type(Z⁺⁺), arr_type(Z⁻⁺))
end

# rt_kernel!(::noRS) (rt_kernel.jl:175-250): the scattering branch of a "11" / TOA layer is ONE call (elemental! + doubling! +
# copy | interaction!) -- on the composite's native copy where the blocks fit (registry above), else on the reference's arrays
# (vsm_layer_forward); the other branches keep the reference's sequence, whose pieces are the methods above.
function rt_kernel!(RS::noRS{FT}, pol_type, SFI, a::AddedLayer{FT}, c::CompositeLayer{FT}, p::CoreScatteringOpticalProperties,
                    iface, τ_sum::ROCArray, m, qp, I_static, arch, qp_μN, iz; workspace=nothing, prepared_sources=CoreRT.NoSource(),
                    dτ_max_threshold=nothing, dτ_min_floor=nothing) where {FT<:FTs}
    scatter = maximum(Array(p.τ .* p.ϖ)) > 2eps(FT)
    # (per-source slots, e.g. :thermal: the reference's sequence -- elemental!, contribute!, doubling!, interaction! are methods here)
    if scatter && (iz == 1 || iface isa ScatteringInterface_11) && isempty(a.j₀_by_src)
        dτ, ndoubl = get_dtau_ndoubl(p, qp; dτ_max_threshold, dτ_min_floor)
        NATIVE_DROPIN[] && _rt_kernel_native!(RS, pol_type, c, p, τ_sum, Int(m), qp, Int(iz), dτ, Int(ndoubl)) && return nothing
        materialize!(c)
        return @vsm("vsm_layer_forward", FT, (Ref{VsmQuad{FT}}, Cint, Cint, Cint, PV, PV, PV, PV, PV, PV, Clonglong, Cint, Ref{VsmComposite}, Ref{VsmAdded}, PV),
                     _q(qp, pol_type.n, FT), length(τ_sum), m, ndoubl, _p(dτ), _p(p.ϖ), _p(τ_sum), _p(_dev(RS.F₀)), _p(p.Z⁺⁺), _p(p.Z⁻⁺),
                     _ms(p.Z⁺⁺), iz == 1 ? 1 : 0, _c(c), _c(a), _stream())
    end
    materialize!(c)
    invoke(rt_kernel!, Tuple{noRS, Any, Any, Any, Any, Any, Any, Any, Any, Any, Any, Any, Any, Any}, RS, pol_type, SFI, a, c, p, iface,
           τ_sum, m, qp, I_static, arch, qp_μN, iz; workspace, prepared_sources, dτ_max_threshold, dτ_min_floor)
end

# All Fourier moments of a scattering "11" / TOA layer in ONE launch (vsm_layer_forward_multi): for a patched rt_run that walks
# `for iz` outside `for m` with one CompositeLayer per moment (the moments are independent until postprocessing_vza!);
# ps[i] = the layer's CoreScatteringOpticalProperties of moment ms[i] (they differ in Z only), cs[i] its composite.
function rt_kernel_moments!(RS::noRS{FT}, pol_type, a::AddedLayer{FT}, cs::Vector{<:CompositeLayer{FT}}, ps::Vector, τ_sum::ROCArray,
                            ms::Vector{<:Integer}, qp, iz; dτ_max_threshold=nothing, dτ_min_floor=nothing) where {FT<:FTs}
    dτ, ndoubl = get_dtau_ndoubl(ps[1], qp; dτ_max_threshold, dτ_min_floor)
    comps = [_c(c) for c in cs]
    zpp, zmp = [_p(p.Z⁺⁺) for p in ps], [_p(p.Z⁻⁺) for p in ps]
    @vsm("vsm_layer_forward_multi", FT, (Ref{VsmQuad{FT}}, Cint, Cint, Ptr{Cint}, Cint, PV, PV, PV, PV, Cint, Ptr{PV}, Ptr{PV}, Clonglong, PV, PV, Cint,
           Ptr{VsmComposite}, Ref{VsmAdded}, PV),
          _q(qp, pol_type.n, FT), length(τ_sum), length(ms), Cint.(ms), ndoubl, _p(dτ), _p(ps[1].ϖ), _p(τ_sum), _p(RS.F₀), 0, zpp, zmp,
          _ms(ps[1].Z⁺⁺), C_NULL, C_NULL, iz == 1 ? 1 : 0, comps, _c(a), _stream())
end

# Optional, for a driver that is willing to interchange its loops (NOT needed for the native layout: rt_kernel!(::noRS) above reaches
# it through the per-composite registry in the reference's own call order).  The layer loop of rt_run (rt_run.jl:383-453) with the
# CompositeLayers of SEVERAL moments in one run (vsm_run_*: one launch per layer and class of blocks for all of them): a rt_run
# that walks `for iz` outside `for m` (every layer scattering, interface 11) replaces its per-layer rt_kernel_moments! calls by
#     run = NativeRun(qp, pol_type, nSpec, ms, Zstacks)            # make_composite_layer of the moments ms
#     for iz = 1:Nz;  rt_kernel!(run, ps_of_layer(iz), τ_sum[iz], RS.F₀, qp, iz);  end
#     export!(run, cs)                                              # cs[i]: the reference's CompositeLayer of moment ms[i]
# and continues with create_surface_layer! / interaction! / postprocessing_vza! on cs as before.  Zstacks[i] = (Z⁺⁺, Z⁻⁺) stacks
# [N,N,nScatterers] of moment ms[i] (every phase matrix a layer of the run can hand in): their Stokes coupling decides which
# components run as independent blocks (m = 0: (I,Q) | (U,V); compute_Z_matrices.jl:26-110).
mutable struct NativeRun
    handle::PV
    ws::ROCArray{Float64,1}
    q::VsmQuad{Float64}
    coupling::Vector{Cint}
end
function stokes_coupling(N::Int, n::Int, Zpp::ROCArray{Float64,3}, Zmp::ROCArray{Float64,3})
    nb = size(Zpp, 3)
    mask = AMDGPU.zeros(Cint, nb)
    _chk(ccall(_sym(:vsm_stokes_coupling_f64), Cint, (Cint, Cint, Cint, PV, PV, PV, PV), N, n, nb, _p(Zpp), _p(Zmp), PV(pointer(mask)), _stream()))
    Array(mask)                                          # one mask per scatterer (synchronises)
end
function NativeRun(qp::QuadPoints, pol_type, nSpec::Int, ms::Vector{<:Integer}, Zstacks::Vector)
    N, n = length(qp.qp_μN), pol_type.n
    coupling = Cint[reduce(|, stokes_coupling(N, n, Z[1], Z[2])) for Z in Zstacks]
    nbytes = ccall(_sym(:vsm_run_workspace_bytes), Csize_t, (Cint, Cint, Cint, Cint, Ptr{Cint}), N, n, nSpec, length(ms), coupling)
    nbytes > 0 || error("vsm_run: a block of coupled Stokes components exceeds the native kernels (96 rows; Float32: 128)")
    ws = ROCArray{Float64}(undef, cld(Int(nbytes), 8))
    q = _q(qp, n, Float64)
    h = Ref{PV}(C_NULL)
    _chk(ccall(_sym(:vsm_run_create_f64), Cint, (Ref{VsmQuad{Float64}}, Cint, Cint, Ptr{Cint}, Ptr{Cint}, PV, Csize_t, Ref{PV}),
               q, nSpec, length(ms), Cint.(ms), coupling, _p(ws), nbytes, h))
    run = NativeRun(h[], ws, q, coupling)
    finalizer(r -> ccall(_sym(:vsm_run_destroy), Cint, (PV,), r.handle), run)
end
# rt_kernel!(::noRS) for one scattering layer and all moments of the run; layer_coupling[i]: the coupling mask of THIS layer's
# phase matrices at moment i (OR over the scatterers present; `nothing` = the run's): a block they leave exactly zero is a
# diagonal step
function rt_kernel!(run::NativeRun, ps::Vector, τ_sum::ROCArray{Float64}, F₀::ROCArray{Float64}, qp, iz::Integer;
                    layer_coupling=nothing, dτ_max_threshold=nothing, dτ_min_floor=nothing)
    dτ, ndoubl = get_dtau_ndoubl(ps[1], qp; dτ_max_threshold, dτ_min_floor)
    zpp, zmp = [_p(p.Z⁺⁺) for p in ps], [_p(p.Z⁻⁺) for p in ps]
    # (the converted mask is bound to a name and passed as an array: ccall roots its arguments for the call -- a bare
    #  pointer(Cint.(layer_coupling)) would leave the temporary to the GC while the library reads it)
    lc = layer_coupling === nothing ? Cint[] : Cint.(layer_coupling)
    _chk(ccall(_sym(:vsm_run_layer_f64), Cint, (PV, Cint, PV, PV, PV, PV, Cint, Ptr{PV}, Ptr{PV}, Clonglong, PV, Cint, Ptr{Cint}, PV),
               run.handle, ndoubl, _p(dτ), _p(ps[1].ϖ), _p(τ_sum), _p(F₀), 0, zpp, zmp, _ms(ps[1].Z⁺⁺), C_NULL, iz == 1 ? 1 : 0,
               isempty(lc) ? Ptr{Cint}(C_NULL) : lc, _stream()))
end
export!(run::NativeRun, cs::Vector{<:CompositeLayer{Float64}}) =
    _chk(ccall(_sym(:vsm_run_export_f64), Cint, (PV, Ptr{VsmComposite}, PV), run.handle, [_c(c) for c in cs], _stream()))
import!(run::NativeRun, cs::Vector{<:CompositeLayer{Float64}}) =
    _chk(ccall(_sym(:vsm_run_import_f64), Cint, (PV, Ptr{VsmComposite}, PV), run.handle, [_c(c) for c in cs], _stream()))

# contribute!(::PreparedThermalEmission, ...) (Sources/thermal_emission.jl:241-301): the :thermal slot of the elemental layer
function CoreRT.contribute!(prep::CoreRT.PreparedThermalEmission, a::AddedLayer{FT}, ϖ::ROCArray, dτ::ROCArray, iz::Integer, m::Integer,
                            pol_type, qp::QuadPoints, arch) where {FT<:FTs}
    (m == 0 && iz <= size(prep.B_layer, 1) && haskey(a.j₀_by_src, :thermal)) || return nothing
    slot = a.j₀_by_src[:thermal]
    th = VsmAdded(_p(a.r⁻⁺), _p(a.t⁺⁺), _p(a.r⁺⁻), _p(a.t⁻⁻), _p(slot.j₀⁺), _p(slot.j₀⁻), _ms(a.r⁻⁺), 0, 0)   # the slot's vectors
    @vsm("vsm_thermal_source", FT, (Ref{VsmQuad{FT}}, Cint, PV, PV, PV, Ref{VsmAdded}, PV), _q(qp, pol_type.n, FT), length(dτ),
          _p(dτ), _p(ϖ), _p(ROCArray(prep.B_layer[iz, :])), th, _stream())
end

# the :thermal slot of a whole scattering "11" / TOA layer on the slot's own composite `cth` in ONE launch where the shape is fused
# (vsm_layer_thermal_fused; rt_kernel.jl:205-232, doubling.jl:62-81): call it from the patched rt_kernel! next to vsm_layer_forward
function thermal_layer_forward!(prep::CoreRT.PreparedThermalEmission, cth::CompositeLayer{FT}, p::CoreScatteringOpticalProperties, dτ::ROCArray,
                                ndoubl::Integer, pol_type, qp::QuadPoints, iz::Integer) where {FT<:FTs}
    N = size(cth.R⁻⁺, 1)
    ccall((:vsm_layer_thermal_fused, libvsm), Cint, (Cint, Cint), N, FT == Float64 ? 1 : 0) == 1 || return false
    @vsm("vsm_layer_forward_thermal", FT, (Ref{VsmQuad{FT}}, Cint, Cint, PV, PV, PV, Cint, PV, PV, Clonglong, PV, Cint, Ref{VsmComposite}, PV),
          _q(qp, pol_type.n, FT), length(dτ), ndoubl, _p(dτ), _p(p.ϖ), _p(ROCArray(prep.B_layer[iz, :])), 0, _p(p.Z⁺⁺), _p(p.Z⁻⁺),
          _ms(p.Z⁺⁺), C_NULL, iz == 1 ? 1 : 0, _c(cth), _stream())
    return true
end

# ---- surfaces + post-processing ------------------------------------------------------------------------------------------
# any BRDF surface (rpv_surface.jl:51-97): its Fourier block comes from the reference's own reflectance(brdf, pol_type, μ, m)
function create_surface_layer!(brdf::CoreRT.AbstractSurfaceType, a::AddedLayer{FT}, SFI, m::Int, pol_type, qp, τ_sum::ROCArray, arch) where {FT<:FTs}
    ρ = ROCArray(FT.(CoreRT.reflectance(brdf, pol_type, collect(qp.qp_μ), m)))
    @vsm("vsm_brdf_surface", FT, (Ref{VsmQuad{FT}}, Cint, Cint, PV, PV, Ref{VsmAdded}, PV), _q(qp, pol_type.n, FT), length(τ_sum), m,
          _p(ρ), _p(τ_sum), _c(a), _stream())
end
function create_surface_layer!(s::LambertianSurfaceScalar{FT}, a::AddedLayer, SFI, m::Int, pol_type, qp, τ_sum::ROCArray, arch) where {FT<:FTs}
    @vsm("vsm_lambertian_surface", FT, (Ref{VsmQuad{FT}}, Cint, Cint, FT, PV, Ref{VsmAdded}, PV), _q(qp, pol_type.n, FT),
          length(τ_sum), m, s.albedo, _p(τ_sum), _c(a), _stream())
end
_cm(s::CoxMunkSurface{FT}) where {FT} = (n = _get_n_water(s, FT(550)); VsmCoxMunk{FT}(s.wind_speed, real(n), imag(n), s.whitecap_albedo, s.include_whitecaps, s.shadowing))
const _ϕw = Dict{DataType,Any}()                        # 100-point Gauss-Legendre on [0, π] (coxmunk_surface.jl:394), device copy
_phi(FT) = get!(() -> map(x -> ROCArray(FT.(x)), vSmartMOM.CoreRT.CanopyOptics.gauleg(100, 0.0, Float64(π))), _ϕw, FT)
function _reflectance(s::CoxMunkSurface{FT}, q, m, N, deriv) where {FT}
    ρ = ROCArray{FT}(undef, N, N); ρ̇ = deriv ? similar(ρ) : nothing; ϕ, w = _phi(FT)
    @vsm("vsm_coxmunk_reflectance", FT, (Ref{VsmCoxMunk{FT}}, Ref{VsmQuad{FT}}, Cint, Cint, PV, PV, PV, PV, PV), _cm(s), q, m,
          length(ϕ), _p(ϕ), _p(w), _p(ρ), _p(ρ̇), _stream())
    ρ, ρ̇
end
function create_surface_layer!(s::CoxMunkSurface{FT}, a::AddedLayer, SFI, m::Int, pol_type, qp, τ_sum::ROCArray, arch) where {FT<:FTs}
    q = _q(qp, pol_type.n, FT); ρ, _ = _reflectance(s, q, m, size(a.r⁻⁺, 1), false)
    @vsm("vsm_brdf_surface", FT, (Ref{VsmQuad{FT}}, Cint, Cint, PV, PV, Ref{VsmAdded}, PV), q, length(τ_sum), m, _p(ρ), _p(τ_sum), _c(a), _stream())
end
# apply_ss_correction! (coxmunk_surface.jl:481-545); R_SFI lives on the device until the end of rt_run
function apply_ss_correction!(R_SFI::ROCArray{FT,3}, s::CoxMunkSurface{FT}, pol_type, vza, vaz, μ₀, τ_total::ROCArray, m_max, nSpec) where {FT<:FTs}
    ϕ, w = _phi(FT); nV = length(vza); coef = ROCArray{FT}(undef, nV, pol_type.n)
    @vsm("vsm_coxmunk_ss_correction", FT, (Ref{VsmCoxMunk{FT}}, Cint, Cint, Cint, Ptr{FT}, Ptr{FT}, FT, Cint, Cint, PV, PV, PV, PV, PV, PV),
          _cm(s), pol_type.n, nSpec, nV, FT.(cosd.(vza)), FT.(deg2rad.(vaz)), FT(μ₀), m_max, length(ϕ), _p(ϕ), _p(w), _p(τ_total), _p(coef), _p(R_SFI), _stream())
end
# postprocessing_vza! noRS/SFI (postprocessing_vza.jl:23-94) on device-resident R_SFI / T_SFI: no per-moment D2H of J₀∓
function postprocessing_vza!(::noRS, iμ₀, pol_type, c::CompositeLayer{FT}, vza, qp_μ, m, vaz, μ₀, weight, nSpec, SFI, R, R_SFI::ROCArray, T, T_SFI::ROCArray, ie...) where {FT<:FTs}
    n = pol_type.n; nV = length(vza); materialize!(c)
    row0 = Cint[n * (vSmartMOM.CoreRT.nearest_point(qp_μ, cosd(v)) - 1) for v in vza]
    w = FT[weight * (k <= 2 ? cosd(m * vaz[v]) : sind(m * vaz[v])) for v in 1:nV, k in 1:n]
    @vsm("vsm_postprocess_vza", FT, (Cint, Cint, Cint, Cint, Ptr{Cint}, Ptr{FT}, PV, PV, PV, PV, PV), size(c.J₀⁻, 1), n, nSpec, nV,
          row0, w, _p(c.J₀⁻), _p(c.J₀⁺), _p(R_SFI), _p(T_SFI), _stream())
end

# ---- linearized pass: rt_run(model, lin_model, NAer, NGas, NSurf) (rt_run_lin.jl; CoreKernel/*_lin.jl) --------------------
function elemental!(pol_type, SFI::Bool, τ_sum::ROCArray, τ̇_sum::ROCArray, dτ::ROCArray, F₀::ROCArray, p, ṗ::CoreScatteringOpticalPropertiesLin,
                    m::Int, ndoubl::Int, scatter::Bool, qp::QuadPoints, a::AddedLayer{FT}, ȧ::AddedLayerLin{FT}, arch; kw...) where {FT<:FTs}
    pl = size(ṗ.τ̇, 2); dτ̇ = ṗ.τ̇ ./ FT(2)^ndoubl; Ż = ṗ.Ż⁺⁺
    @vsm("vsm_elemental_lin", FT, (Ref{VsmQuad{FT}}, Cint, Cint, Cint, PV, PV, PV, PV, PV, PV, Clonglong, Cint, PV, PV, PV, PV, PV, Clonglong,
                                         Clonglong, Ref{VsmAdded}, Ref{VsmAddedLin}, PV),
          _q(qp, pol_type.n, FT), length(dτ), m, ndoubl, _p(dτ), _p(p.ϖ), _p(τ_sum), _p(F₀), _p(p.Z⁺⁺), _p(p.Z⁻⁺), _ms(p.Z⁺⁺), pl, _p(dτ̇), _p(ṗ.ϖ̇),
          _p(τ̇_sum), _p(Ż), _p(ṗ.Ż⁻⁺), _ms(Ż), size(Ż, 1)^2 * size(Ż, 3), _c(a), _c(ȧ), _stream())
end
# constructCoreOpticalProperties with lin_model on the device (compEffectiveLayerProperties_lin.jl:43-197, types_lin.jl:196-380):
# one call per band fills τ̇/2^ndoubl, ϖ̇, τ̇_sum for every layer and -- with aerosol slots -- the per-point weights fz and the
# coefficients zdcoef of Ż over the component blocks [Z_Rayleigh, Z_aer.., ∂Z_aer/∂(nᵣ, nᵢ, rₘ, σᵣ)..]; Ż[N,N,nSpec,P] is not built.
# `d` = NamedTuple of the raw device inputs (FP64): τ_rayl, τ_abs [nSpec,Nz]; τ_aer [nAer,Nz]; ϖ̃, fᵗ [nAer]; τ̇_abs [nSpec,Nz,nGas];
# τ̇_aer [7,nAer,Nz]; ϖ̃̇, ḟᵗ [4,nAer]; ndoubl::ROCArray{Cint} [Nz].  lo = 0-based start of this rank's block of S points.
function layer_optics_lin!(::Type{FT}, d, S_full::Int, lo::Int, S::Int, Nz::Int, nAer::Int, nGas::Int, P::Int, ϖ_Cabannes,
                           dτ̇::ROCArray, ϖ̇::ROCArray, τ̇_sum::ROCArray, fz, zdcoef) where {FT<:FTs}
    @vsm("vsm_layer_optics_lin", FT, (Cint, Cint, Cint, Cint, Cint, Cint, Cint, PV, PV, Cdouble, PV, PV, PV, PV, PV, PV, PV, PV, PV, PV, PV, PV, PV, PV),
          S_full, lo, S, Nz, nAer, nGas, P, _p(d.τ_rayl), _p(d.τ_abs), Float64(ϖ_Cabannes), _p(d.τ_aer), _p(d.ϖ̃), _p(d.fᵗ), _p(d.τ̇_abs),
          _p(d.τ̇_aer), _p(d.ϖ̃̇), _p(d.ḟᵗ), _p(d.ndoubl), _p(dτ̇), _p(ϖ̇), _p(τ̇_sum), _p(fz), _p(zdcoef), _stream())
end
# elemental! (lin) of a layer whose Z / Ż are per-point mixes of the component blocks Zc (one Fourier moment): zsel = -1 mixes
# Z = Σ fz[c,s] Zc[c], zsel ≥ 0 takes Zc[zsel + 1]; Ż[:,:,s,p] = Σ zdcoef[c,p,s] Zc[c] is formed inside the kernel.
function elemental_mix!(pol_type, τ_sum::ROCArray, τ̇_sum::ROCArray, dτ::ROCArray, dτ̇::ROCArray, F₀::ROCArray, ϖ::ROCArray, ϖ̇::ROCArray,
                        Zc⁺⁺::ROCArray{FT,3}, Zc⁻⁺::ROCArray{FT,3}, ncomp::Int, zsel::Int, fz, zdcoef::ROCArray, m::Int, ndoubl::Int,
                        qp::QuadPoints, a::AddedLayer{FT}, ȧ::AddedLayerLin{FT}) where {FT<:FTs}
    @vsm("vsm_elemental_lin_mix", FT, (Ref{VsmQuad{FT}}, Cint, Cint, Cint, PV, PV, PV, PV, Cint, Cint, PV, PV, Cint, PV, Cint, PV, PV, PV, PV,
                                             Ref{VsmAdded}, Ref{VsmAddedLin}, PV),
          _q(qp, pol_type.n, FT), length(dτ), m, ndoubl, _p(dτ), _p(ϖ), _p(τ_sum), _p(F₀), ncomp, size(Zc⁺⁺, 3), _p(Zc⁺⁺), _p(Zc⁻⁺), zsel, _p(fz),
          size(ϖ̇, 2), _p(dτ̇), _p(ϖ̇), _p(τ̇_sum), _p(zdcoef), _c(a), _c(ȧ), _stream())
end
# expk = exp.(-dτ ./ μ₀) (init_layer, rt_kernel.jl:339-349) without a host round trip
layer_expk!(expk::ROCArray{FT}, dτ::ROCArray{FT}, μ₀) where {FT<:FTs} =
    @vsm("vsm_layer_expk", FT, (Cint, PV, FT, PV, PV), length(dτ), _p(dτ), FT(μ₀), _p(expk), _stream())
function doubling_allparams!(pol_type, SFI, expk::ROCArray{FT}, ndoubl::Int, a::AddedLayer, ȧ::AddedLayerLin, I_static, arch, dτ̇::ROCArray, μ₀; N_active::Int=0) where {FT<:FTs}
    N, _, S = size(a.r⁻⁺); P = size(ȧ.ap_ṙ⁻⁺, 4)
    @vsm("vsm_doubling_lin", FT, (Cint, Cint, Cint, Cint, PV, PV, FT, Cint, Ref{VsmAdded}, Ref{VsmAddedLin}, PV, PV), N, pol_type.n, S, ndoubl,
          _p(expk), _p(dτ̇), FT(μ₀), N_active, _c(a), _c(ȧ), _p(_work(FT, :vsm_doubling_lin_work_elems, N, S, P)), _stream())
end
function interaction!(iface, SFI, c::CompositeLayer{FT}, ċ::CompositeLayerLin{FT}, a::AddedLayer{FT}, ȧ::AddedLayerLin{FT}, I_static) where {FT<:FTs}
    N, _, S = size(c.R⁻⁺); P = size(ċ.Ṙ⁻⁺, 4)
    @vsm("vsm_interaction_lin", FT, (Cint, Cint, Cint, Ref{VsmComposite}, Ref{VsmCompositeLin}, Ref{VsmAdded}, Ref{VsmAddedLin}, PV, PV),
          _tag(iface), N, S, _c(c), _c(ċ), _c(a), _c(ȧ), _p(_work(FT, :vsm_interaction_lin_work_elems, N, S, P)), _stream())
end
# The same for the parameter slots lo:hi only (1-based, inclusive).  rt_run_lin.jl's layer loop may pass 1:n_layer_params(layout) for the
# atmospheric layers: the composite above the surface does not depend on a surface parameter, so the loop `for iparam = 1:Nparams`
# of interaction_lin.jl:242,291 computes exact zeros for the surface slots there (identical results, the surface interaction runs all).
function interaction!(iface, SFI, c::CompositeLayer{FT}, ċ::CompositeLayerLin{FT}, a::AddedLayer{FT}, ȧ::AddedLayerLin{FT}, I_static,
                      slots::UnitRange{Int}) where {FT<:FTs}
    N, _, S = size(c.R⁻⁺); P = size(ċ.Ṙ⁻⁺, 4)
    @vsm("vsm_interaction_lin_range", FT, (Cint, Cint, Cint, Ref{VsmComposite}, Ref{VsmCompositeLin}, Ref{VsmAdded}, Ref{VsmAddedLin}, Cint, Cint, PV, PV),
          _tag(iface), N, S, _c(c), _c(ċ), _c(a), _c(ȧ), first(slots) - 1, last(slots), _p(_work(FT, :vsm_interaction_lin_work_elems, N, S, P)), _stream())
end
function create_surface_layer!(::noRS, s::LambertianSurfaceScalar{FT}, a::AddedLayer, ȧ::AddedLayerLin, iparam::Int, SFI, m::Int, pol_type, qp, τ_sum, τ̇_sum, F₀, arch) where {FT<:FTs}
    @vsm("vsm_lambertian_surface_lin", FT, (Ref{VsmQuad{FT}}, Cint, Cint, FT, Cint, PV, PV, Cint, PV, Ref{VsmAdded}, Ref{VsmAddedLin}, PV),
          _q(qp, pol_type.n, FT), length(τ_sum), m, s.albedo, iparam - 1, _p(τ_sum), _p(τ̇_sum), size(τ̇_sum, 2), _p(F₀), _c(a), _c(ȧ), _stream())
end
function create_surface_layer!(::noRS, s::CoxMunkSurface{FT}, a::AddedLayer, ȧ::AddedLayerLin, iparam::Int, SFI, m::Int, pol_type, qp, τ_sum, τ̇_sum, F₀, arch) where {FT<:FTs}
    q = _q(qp, pol_type.n, FT); ρ, ρ̇ = _reflectance(s, q, m, size(a.r⁻⁺, 1), true)
    @vsm("vsm_brdf_surface_lin", FT, (Ref{VsmQuad{FT}}, Cint, Cint, PV, PV, Cint, PV, PV, Cint, PV, Ref{VsmAdded}, Ref{VsmAddedLin}, PV),
          q, length(τ_sum), m, _p(ρ), _p(ρ̇), iparam - 1, _p(τ_sum), _p(τ̇_sum), size(τ̇_sum, 2), _p(F₀), _c(a), _c(ȧ), _stream())
end

# ---- rotational Raman (CoreKernel/*_inelastic.jl): 4-D arrays keep the reference layout [N,N,nSpec,nRaman] ------------------
_rrs(RS, fscatt) = VsmRRS(_p(RS.i_λ₁λ₀), _p(RS.ϖ_λ₁λ₀), _p(fscatt), _p(RS.Z⁺⁺_λ₁λ₀), _p(RS.Z⁻⁺_λ₁λ₀))   # i_λ₁λ₀ as ROCArray{Cint}
function interaction!(RS::RRS{FT}, iface::ScatteringInterface_11, SFI, c, a, I_static; workspace=nothing) where {FT<:FTs}
    N, _, S = size(c.R⁻⁺); K = size(c.ieR⁻⁺, 4)
    @vsm("vsm_interaction_inelastic_rrs", FT, (Cint, Cint, Cint, PV, Ref{VsmComposite}, Ref{VsmCompositeRS}, Ref{VsmAdded}, Ref{VsmAddedRS}, PV, PV),
          3, N, S, _p(RS.i_λ₁λ₀), _c(c), _Crs(c), _c(a), _crs(a), _p(_work(FT, :vsm_interaction_inelastic_work_elems, N, S, K)), _stream())
end
# elemental_inelastic! -> vsm_elemental_inelastic_rrs_*, doubling_inelastic! -> vsm_doubling_inelastic_rrs_*, copy_added_to_composite_ie!
# -> vsm_copy_added_to_composite_ie_*, postprocessing_vza!(::RRS) -> vsm_postprocess_vza_ie_*: same pattern (signatures in the header).
end # module
