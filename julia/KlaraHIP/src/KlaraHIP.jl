# KlaraHIP.jl — Julia host side of libklara_hip.so: Klara's own sampler / tuner / range structs in, Klara's own
# BasicContMuvParameterNState out, the transition loop on an MI355X in between (include/klara_hip.h, INTEGRATION.md).
# Package layout: julia/KlaraHIP/{REQUIRE, src/KlaraHIP.jl, test/runtests.jl}; put julia/ on LOAD_PATH (or Pkg.clone the directory).
#
#     using Klara, KlaraHIP
#     plogtarget = GaussDiagTarget(2)                                     # README.md:23 `plogtarget(z) = -dot(z, z)` as a device target family
#     p      = HIPParameter(:p, logtarget=plogtarget)                     # README.md:30  BasicContMuvParameter(:p, logtarget=plogtarget)
#     model  = likelihood_model(p, false)                                 # README.md:35  unchanged: HIPParameter <: Klara.Parameter{Continuous, Multivariate}
#     job    = HIPMCJob(model, MH(ones(2)), BasicMCRange(nsteps=10000, burnin=1000), Dict(:p => [5.1, -0.9]))     # README.md:39-51 BasicMCJob(model, sampler, mcrange, v0)
#     run(job); chain = output(job); mean(chain); acceptance(chain)      # README.md:55-59 unchanged
#     reset(job); run(job)                                                # an independent replicate (BasicMCJob.jl:187-201)
#
# Many chains of one model — what the device is for — are the same call with `nchains` (v0's vector is every chain's start) or a D x N matrix of starts:
#     job    = HIPMCJob(model, MALA(0.9), BasicMCRange(nsteps=10000, burnin=1000), Dict(:p => randn(100, 65536));
#                       tuner=VanillaMCTuner(), outopts=Dict(:monitor => [:value], :diagnostics => [:accept]))
#     run(job); chain = output(job, 1); mean(chain); acceptance(chain)
#
# `run`, `reset` (Base generics Klara extends: src/Klara.jl:26-27) and `output` (Klara's own, exported at src/Klara.jl:232) are
# IMPORTED before the methods below are defined, so `run(job)`, `reset(job[, x])`, `output(job[, c])` on an HIPMCJob are methods
# of the same functions a Klara user already calls on a BasicMCJob (src/jobs/BasicMCJob.jl:187-201,212,279); HIPMCJob <: Klara.MCJob,
# so `run(jobs::Vector)` (src/jobs/jobs.jl:212) maps over HIP jobs too.  Everything else the example names is exported below.
#
# UNTESTED in the build image (no Julia there); tests/test_host_api.py checks mechanically what can be checked without it: the
# struct layout against the C header, the descriptor builder's argument order, the Klara field names each mapping reads
# (cited below), that only declared symbols are bound, that blocks / brackets balance, that every unqualified name of the example
# above and of INTEGRATION.md is exported here or by Klara, that every generic of Klara / Base this file adds methods to is imported
# first, and that nothing reaches through the caller's top-level module.  One file for Klara's own Julia 0.6
# (REQUIRE:1 — `Void`, `Array{T}(dims)`, `finalizer(obj, f)`) and for Julia >= 0.7 (`Cvoid`, `Array{T}(undef, dims)`,
# `finalizer(f, obj)`): the three differences are confined to the compatibility block below, everything else is common syntax.
module KlaraHIP
import Klara
import Klara: output, MCJob, MH, MALA, HMC, SliceSampler, VanillaMCTuner, AcceptanceRateMCTuner, DualAveragingMCTuner,
              BasicContMuvParameterState, BasicContMuvParameterNState, erf_rate_score, logistic_rate_score,
              Parameter, GenericModel, VariableState, VariableStateVector
import Distributions
import Distributions: Continuous, Multivariate
import Base: run, reset, show
export HIPMCJob, HIPParameter, HIPTarget, GaussDiagTarget, GaussDenseTarget, LogisticTarget, HierNormalTarget, CustomTarget,
       chainvalue, chainmeans, chainacceptance, chainmcvar_bm, streamkey, launchmodes, shaderclock, check_custom_target,
       HIPComm, comm_unique_id, comm_info, gather_summaries, gather_moments, pooledmoments, KlaraDesc, klara_desc
const lib = "libklara_hip"            # klara.jl_amd/lib/libklara_hip.so on LD_LIBRARY_PATH

# ---------------------------------------------------------------- Julia 0.6 / >= 0.7 compatibility (the only version-dependent code)
@static if VERSION < v"0.7.0-"
    const Cvoid = Void
    newarray(::Type{T}, dims::Integer...) where {T} = Array{T}(dims...)
    on_finalize(obj, f) = finalizer(obj, f)
else
    newarray(::Type{T}, dims::Integer...) where {T} = Array{T}(undef, dims...)
    on_finalize(obj, f) = finalizer(f, obj)
end

# ---------------------------------------------------------------- constants of include/klara_hip.h
const KLARA_ABI_VERSION = UInt32(6)
const SAMPLER_MH, SAMPLER_MALA, SAMPLER_HMC, SAMPLER_SLICE = Int32(0), Int32(1), Int32(2), Int32(3)
const TARGET_GAUSS_DIAG, TARGET_GAUSS_DENSE, TARGET_LOGISTIC, TARGET_HIER_NORMAL, TARGET_CUSTOM = Int32(0), Int32(1), Int32(2), Int32(3), Int32(4)
const TUNER_VANILLA, TUNER_ACCEPT_RATE, TUNER_DUAL_AVERAGING = Int32(0), Int32(1), Int32(2)
const TUNE_PER_CHAIN, TUNE_POOLED = Int32(0), Int32(1)
const MON_ACCEPT, MON_HISTORY, MON_SUMMARIES, MON_HIST_LT, MON_HIST_GRAD, MON_HIST_LLLP = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20

# struct klara_desc — field order/types exactly as include/klara_hip.h
struct KlaraDesc
    struct_size::UInt32; abi_version::UInt32
    sampler::Int32; target::Int32; tuner::Int32; tuner_mode::Int32
    nchains::Int64; chain_offset::Int64; ndims::Int32; device::Int32
    mh_sigma::Ptr{Float64}; driftstep::Float64; leapstep::Float64
    nleaps::Int32; slice_stepout::Int32; slice_widths::Ptr{Float64}
    targetrate::Float64; score_k::Float64; period::Int32; verbose::Int32
    da_nadapt::Int64; da_eps0bar::Float64; da_h0bar::Float64; da_gamma::Float64; da_kappa::Float64
    da_t0::Int32; tuner_score::Int32
    nsteps::Int64; burnin::Int64; thinning::Int64
    gauss_w::Ptr{Float64}; gauss_mu::Ptr{Float64}; gauss_const::Float64; gauss_prec::Ptr{Float64}
    logit_X::Ptr{Float64}; logit_y::Ptr{Float64}; logit_ndata::Int32; nstreams::Int32; logit_lambda::Float64
    hier_Y::Ptr{Float64}; hier_xc::Ptr{Float64}; hier_nunits::Int32; hier_ntimes::Int32
    hier_prior_prec::Float64; hier_gamma_a::Float64; hier_gamma_b::Float64
    custom_src::Cstring; custom_data::Ptr{Float64}; custom_ndata::Int64; bm_batchlen::Int64
    hist_ring_cols::Int64; acov_maxlag::Int32; sparse_moves::Int32
    seed::UInt64; monitor::UInt32; steps_per_launch::Int32; stream::Ptr{Cvoid}
end

# every field by keyword, defaults = "not used"; the positional call below must list the fields in struct order (checked by
# tests/test_host_api.py)
function klara_desc(; sampler=SAMPLER_MH, target=TARGET_GAUSS_DIAG, tuner=TUNER_VANILLA, tuner_mode=TUNE_PER_CHAIN,
                    nchains=1, chain_offset=0, ndims=1, device=0,
                    mh_sigma=Ptr{Float64}(C_NULL), driftstep=1.0, leapstep=0.1, nleaps=10, slice_stepout=1, slice_widths=Ptr{Float64}(C_NULL),
                    targetrate=0.0, score_k=7.0, period=100, verbose=0,
                    da_nadapt=0, da_eps0bar=1.0, da_h0bar=0.0, da_gamma=0.05, da_kappa=0.75, da_t0=10, tuner_score=0,
                    nsteps=100, burnin=0, thinning=1,
                    gauss_w=Ptr{Float64}(C_NULL), gauss_mu=Ptr{Float64}(C_NULL), gauss_const=0.0, gauss_prec=Ptr{Float64}(C_NULL),
                    logit_X=Ptr{Float64}(C_NULL), logit_y=Ptr{Float64}(C_NULL), logit_ndata=0, nstreams=0, logit_lambda=100.0,
                    hier_Y=Ptr{Float64}(C_NULL), hier_xc=Ptr{Float64}(C_NULL), hier_nunits=0, hier_ntimes=0,
                    hier_prior_prec=1e-4, hier_gamma_a=1e-3, hier_gamma_b=1e-3,
                    custom_src=Cstring(C_NULL), custom_data=Ptr{Float64}(C_NULL), custom_ndata=0, bm_batchlen=0,
                    hist_ring_cols=0, acov_maxlag=0, sparse_moves=0,
                    seed=UInt64(0), monitor=UInt32(0), steps_per_launch=0, stream=C_NULL)
    KlaraDesc(UInt32(sizeof(KlaraDesc)), KLARA_ABI_VERSION,
              sampler, target, tuner, tuner_mode,
              nchains, chain_offset, ndims, device,
              mh_sigma, driftstep, leapstep,
              nleaps, slice_stepout, slice_widths,
              targetrate, score_k, period, verbose,
              da_nadapt, da_eps0bar, da_h0bar, da_gamma, da_kappa,
              da_t0, tuner_score,
              nsteps, burnin, thinning,
              gauss_w, gauss_mu, gauss_const, gauss_prec,
              logit_X, logit_y, logit_ndata, nstreams, logit_lambda,
              hier_Y, hier_xc, hier_nunits, hier_ntimes,
              hier_prior_prec, hier_gamma_a, hier_gamma_b,
              custom_src, custom_data, custom_ndata, bm_batchlen,
              hist_ring_cols, acov_maxlag, sparse_moves,
              seed, monitor, steps_per_launch, stream)
end

check(st::Cint, what) = st == 0 || error(what, ": ", unsafe_string(ccall((:klara_strerror, lib), Cstring, (Cint,), st)))

# ---------------------------------------------------------------- device target families (klara_target)
# Julia closures cannot run on the GPU: the parameter's target is one of the enumerated families, or C text (CustomTarget) —
# the device form of BasicContMuvParameter(:p, logtarget=f, gradlogtarget=g) / (loglikelihood=..., logprior=...).
abstract type HIPTarget end
struct GaussDiagTarget <: HIPTarget          # lt = c - sum w_i (x_i - mu_i)^2 ; README.md:23 is GaussDiagTarget(D)
    ndims::Int; w::Vector{Float64}; mu::Vector{Float64}; c::Float64
end
GaussDiagTarget(D::Integer) = GaussDiagTarget(D, Float64[], Float64[], 0.0)
struct GaussDenseTarget <: HIPTarget         # lt = c - 1/2 (x-mu)' P (x-mu), P row-major D x D; mu empty = 0
    P::Matrix{Float64}; c::Float64; mu::Vector{Float64}
end
GaussDenseTarget(P::Matrix{Float64}, c::Float64=0.0) = GaussDenseTarget(P, c, Float64[])
struct LogisticTarget <: HIPTarget           # doc/examples/swiss/MALA/analytical.jl:11-18; X is ndata x D
    X::Matrix{Float64}; y::Vector{Float64}; lambda::Float64
end
struct HierNormalTarget <: HIPTarget         # BUGS "Rats" model on data/rats/*.csv (include/klara_hip.h KLARA_TARGET_HIER_NORMAL)
    Y::Matrix{Float64}; xc::Vector{Float64}; prior_prec::Float64; gamma_a::Float64; gamma_b::Float64
end
struct CustomTarget <: HIPTarget             # C text of klara_user_logtarget / klara_user_gradlogtarget (or the likelihood + prior form)
    ndims::Int; src::String; data::Vector{Float64}
end
ndims_of(t::GaussDiagTarget) = t.ndims
ndims_of(t::GaussDenseTarget) = size(t.P, 1)
ndims_of(t::LogisticTarget) = size(t.X, 2)
ndims_of(t::HierNormalTarget) = 2 * size(t.Y, 1) + 5
ndims_of(t::CustomTarget) = t.ndims

# The device form of BasicContMuvParameter (variables/parameters/BasicContMuvParameter.jl:3-28): a Klara.Parameter{Continuous, Multivariate}
# with the fields a GenericModel reads and writes — `key` (GenericModel.jl:44 `m.ofkey[v.key] = n`), `index` (assigned by
# likelihood_model(p, false): GenericModel.jl:110-113 `m.vertices[i].index = i`, hence mutable) and `states` (BasicContMuvParameter.jl:27) — and, in place
# of the closure fields, the device target family.  `likelihood_model(p, false)` (models/generators.jl:20) therefore works on it unchanged.
mutable struct HIPParameter <: Parameter{Continuous, Multivariate}
    key::Symbol
    index::Integer
    target::HIPTarget
    states::VariableStateVector
end
HIPParameter(key::Symbol, target::HIPTarget; index::Integer=0, states::VariableStateVector=VariableState[]) = HIPParameter(key, index, target, states)
# keyword form, mirroring BasicContMuvParameter(key; logtarget=...) (BasicContMuvParameter.jl:383-411): the "closure" is a device target
HIPParameter(key::Symbol; logtarget::HIPTarget=error("HIPParameter: logtarget=<a device target family, e.g. GaussDiagTarget(D)> is required"),
             index::Integer=0, states::VariableStateVector=VariableState[]) = HIPParameter(key, index, logtarget, states)

# ---------------------------------------------------------------- the job
mutable struct HIPMCJob <: MCJob   # stands for N BasicMCJobs of one model (run(jobs::Vector) = map(run, jobs), jobs.jl:212)
    handle::Ptr{Cvoid}; nchains::Int; ndims::Int
    parameter::HIPParameter; sampler; tuner; range        # Klara's own MCSampler / MCTuner / BasicMCRange
    outopts::Dict{Symbol, Any}; monitor::UInt32
    keep::Vector{Any}                                      # host arrays the descriptor pointed at
end

rowmajor(A::Matrix{Float64}) = collect(transpose(A))       # Julia is column-major, the C ABI row-major

# Klara's structs -> klara_desc.  Field names read from Klara (file:line in /root/reference/src):
#   MALA.driftstep                         samplers/MALA.jl:61-70
#   HMC.leapstep, HMC.nleaps               samplers/HMC.jl:89-100
#   SliceSampler.widths, .stepout          samplers/SliceSampler.jl:22-34
#   MH.setproposal (sigma is inside the closure: MH(sigma) = MH(x -> MvNormal(x, sigma)))   samplers/MH.jl:46-66
#   VanillaMCTuner.period, .verbose        tuners/VanillaMCTuner.jl:6-17
#   AcceptanceRateMCTuner.targetrate, .score, .period, .verbose          tuners/AcceptanceRateMCTuner.jl:25-44
#   DualAveragingMCTuner.targetrate, .nadapt, .ε0bar, .h0bar, .γ, .t0, .κ, .period, .verbose   tuners/DualAveragingMCTuner.jl:54-93
#   BasicMCRange.burnin, .thinning, .nsteps, .postrange, .npoststeps     ranges/BasicMCRange.jl:7-36
# v0[key]: a D x N matrix of starts (column = chain), or one vector — BasicMCJob's form (README.md:47 `Dict(:p=>[5.1, -0.9])`) — that every one of
# `nchains` chains starts from
startmatrix(x::Matrix{Float64}, nchains::Integer) = x
function startmatrix(x::AbstractVector, nchains::Integer)
    X = newarray(Float64, length(x), Int(nchains))
    for j in 1:Int(nchains), i in 1:length(x)
        X[i, j] = x[i]
    end
    X
end
startmatrix(x, nchains::Integer) = convert(Matrix{Float64}, x)

# BasicMCJob(model, sampler, range, v0; tuner, outopts) (jobs/BasicMCJob.jl:140-185): the parameter is the model's first Parameter vertex
# (`pindex`, :144) and v0 is keyed by the variables' keys (:156-158)
function firstparameter(vs)               # BasicMCJob.jl:144 `findfirst(v -> isa(v, Parameter), model.vertices)` (findfirst's "none" differs between 0.6 and 0.7)
    for i in 1:length(vs)
        isa(vs[i], Parameter) && return i
    end
    error("the model has no Parameter vertex")
end
function HIPMCJob(model::GenericModel, sampler, mcrange, v0::Dict;
                  pindex::Integer=firstparameter(model.vertices), kwargs...)
    parameter = model.vertices[pindex]
    isa(parameter, HIPParameter) || error("the model's parameter must be a HIPParameter (a device target family), got $(typeof(parameter))")
    HIPMCJob(parameter, sampler, mcrange, v0; kwargs...)
end

function HIPMCJob(parameter::HIPParameter, sampler, mcrange, v0::Dict;
                  tuner=nothing, outopts::Dict=Dict{Symbol, Any}(), pooled::Bool=false, nchains::Integer=1,
                  seed::Integer=rand(UInt64), chain_offset::Integer=0, device::Integer=0, steps_per_launch::Integer=0,
                  summaries::Bool=true, bm_batchlen::Integer=0)
    X0 = startmatrix(v0[parameter.key], nchains)                          # D x N: column = chain == N x D row-major
    D, N = size(X0)
    t = parameter.target
    D == ndims_of(t) || error("v0 has the wrong number of dimensions for the target")
    keep = Any[X0]
    kw = Dict{Symbol, Any}(:nchains => N, :ndims => D, :chain_offset => chain_offset, :device => device, :seed => UInt64(seed),
                           :steps_per_launch => steps_per_launch, :bm_batchlen => bm_batchlen,
                           :nsteps => mcrange.nsteps, :burnin => mcrange.burnin, :thinning => mcrange.thinning)
    # --- sampler
    if isa(sampler, MALA)
        kw[:sampler] = SAMPLER_MALA; kw[:driftstep] = Float64(sampler.driftstep)
    elseif isa(sampler, HMC)
        kw[:sampler] = SAMPLER_HMC; kw[:leapstep] = Float64(sampler.leapstep); kw[:nleaps] = Int32(sampler.nleaps)
    elseif isa(sampler, SliceSampler)
        w = convert(Vector{Float64}, sampler.widths); push!(keep, w)
        length(w) == D || error("SliceSampler widths must have one entry per dimension")
        kw[:sampler] = SAMPLER_SLICE; kw[:slice_widths] = pointer(w); kw[:slice_stepout] = Int32(sampler.stepout)
    elseif isa(sampler, MH)
        (sampler.symmetric && sampler.normalised) || error("only the symmetric normalised random-walk MH(sigma) runs on the device (iterate/MH.jl:72-124)")
        prop = sampler.setproposal(BasicContMuvParameterState(zeros(D)))    # MvNormal(x, sigma): its variances give sigma
        sig = sqrt.(Distributions.var(prop)); push!(keep, sig)
        kw[:sampler] = SAMPLER_MH; kw[:mh_sigma] = pointer(sig)
    else
        error("sampler $(typeof(sampler)) is not on the device path (MH, MALA, HMC, SliceSampler are)")
    end
    # --- tuner
    tn = tuner === nothing ? VanillaMCTuner() : tuner
    kw[:period] = Int32(tn.period); kw[:verbose] = Int32(tn.verbose)
    if isa(tn, VanillaMCTuner)
        kw[:tuner] = TUNER_VANILLA
    elseif isa(tn, AcceptanceRateMCTuner)
        kw[:tuner] = TUNER_ACCEPT_RATE; kw[:targetrate] = Float64(tn.targetrate)
        kw[:tuner_mode] = pooled ? TUNE_POOLED : TUNE_PER_CHAIN
        # tuner.score is a function: logistic_rate_score (k = 7) and erf_rate_score (k = 3) are the two Klara ships
        if tn.score === erf_rate_score
            kw[:tuner_score] = Int32(1); kw[:score_k] = 3.0
        elseif tn.score === logistic_rate_score
            kw[:tuner_score] = Int32(0); kw[:score_k] = 7.0
        else
            error("AcceptanceRateMCTuner score must be logistic_rate_score or erf_rate_score on the device")
        end
    elseif isa(tn, DualAveragingMCTuner)
        isa(sampler, HMC) || error("DualAveragingMCTuner is wired into HMC only (HMC.jl:124-133)")
        kw[:tuner] = TUNER_DUAL_AVERAGING; kw[:targetrate] = Float64(tn.targetrate); kw[:da_nadapt] = Int64(tn.nadapt)
        kw[:da_eps0bar] = Float64(tn.ε0bar); kw[:da_h0bar] = Float64(tn.h0bar); kw[:da_gamma] = Float64(tn.γ)
        kw[:da_t0] = Int32(tn.t0); kw[:da_kappa] = Float64(tn.κ)
    else
        error("tuner $(typeof(tn)) is not on the device path")
    end
    # --- target
    if isa(t, GaussDiagTarget)
        kw[:target] = TARGET_GAUSS_DIAG; kw[:gauss_const] = t.c
        if !isempty(t.w); push!(keep, t.w); kw[:gauss_w] = pointer(t.w); end
        if !isempty(t.mu); push!(keep, t.mu); kw[:gauss_mu] = pointer(t.mu); end
    elseif isa(t, GaussDenseTarget)
        P = rowmajor(t.P); push!(keep, P)
        kw[:target] = TARGET_GAUSS_DENSE; kw[:gauss_prec] = pointer(P); kw[:gauss_const] = t.c
        if !isempty(t.mu); push!(keep, t.mu); kw[:gauss_mu] = pointer(t.mu); end
    elseif isa(t, LogisticTarget)
        X = rowmajor(t.X); push!(keep, X); push!(keep, t.y)
        kw[:target] = TARGET_LOGISTIC; kw[:logit_X] = pointer(X); kw[:logit_y] = pointer(t.y)
        kw[:logit_ndata] = Int32(size(t.X, 1)); kw[:logit_lambda] = t.lambda
    elseif isa(t, HierNormalTarget)
        Y = rowmajor(t.Y); push!(keep, Y); push!(keep, t.xc)
        kw[:target] = TARGET_HIER_NORMAL; kw[:hier_Y] = pointer(Y); kw[:hier_xc] = pointer(t.xc)
        kw[:hier_nunits] = Int32(size(t.Y, 1)); kw[:hier_ntimes] = Int32(size(t.Y, 2))
        kw[:hier_prior_prec] = t.prior_prec; kw[:hier_gamma_a] = t.gamma_a; kw[:hier_gamma_b] = t.gamma_b
    else
        push!(keep, t.src)
        kw[:target] = TARGET_CUSTOM; kw[:custom_src] = Base.unsafe_convert(Cstring, t.src)
        if !isempty(t.data); push!(keep, t.data); kw[:custom_data] = pointer(t.data); kw[:custom_ndata] = length(t.data); end
    end
    # --- outopts (jobs.jl:9-43): destination, monitor, diagnostics
    oo = Dict{Symbol, Any}(outopts)
    get!(oo, :destination, :nstate)
    if oo[:destination] != :none
        get!(oo, :monitor, [:value]); get!(oo, :diagnostics, Symbol[])
    end
    mon = UInt32(summaries ? MON_SUMMARIES : 0x00)
    if :accept in get(oo, :diagnostics, Symbol[]); mon |= MON_ACCEPT; end
    for m in get(oo, :monitor, Symbol[])
        m == :value && (mon |= MON_HISTORY)
        m == :logtarget && (mon |= MON_HIST_LT)
        m == :gradlogtarget && (mon |= MON_HIST_GRAD)
        (m == :loglikelihood || m == :logprior) && (mon |= MON_HIST_LLLP)
    end
    kw[:monitor] = mon
    desc = klara_desc(; kw...)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:klara_create, lib), Cint, (Ref{KlaraDesc}, Ref{Ptr{Cvoid}}), desc, h), "klara_create")
    job = HIPMCJob(h[], N, D, parameter, sampler, tn, mcrange, oo, mon, keep)
    on_finalize(job, j -> ccall((:klara_destroy, lib), Cint, (Ptr{Cvoid},), j.handle))
    # initialize!(pstate, parameter, sampler, outopts): BasicMCJob.jl:73 (finiteness asserts on device)
    check(ccall((:klara_set_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, X0), "klara_set_state")
    job
end

# raw form for callers that fill the descriptor themselves
function HIPMCJob(desc::KlaraDesc, X0::Matrix{Float64}, range)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:klara_create, lib), Cint, (Ref{KlaraDesc}, Ref{Ptr{Cvoid}}), desc, h), "klara_create")
    job = HIPMCJob(h[], size(X0, 2), size(X0, 1), HIPParameter(:p, GaussDiagTarget(size(X0, 1))), nothing, nothing, range,
                   Dict{Symbol, Any}(), desc.monitor, Any[X0])
    on_finalize(job, j -> ccall((:klara_destroy, lib), Cint, (Ptr{Cvoid},), j.handle))
    check(ccall((:klara_set_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, X0), "klara_set_state")
    job
end

# run / reset: methods of Base.run / Base.reset — the generics Klara itself extends (src/Klara.jl:26-27; imported above)
run(job::HIPMCJob) =                                     # BasicMCJob.jl:212-244 for all chains
    check(ccall((:klara_run, lib), Cint, (Ptr{Cvoid}, Clonglong), job.handle, job.range.nsteps), "klara_run")

reset(job::HIPMCJob) =                                   # BasicMCJob.jl:187-195; the job moves to its next Philox key (klara_hip.h)
    check(ccall((:klara_reset, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, C_NULL), "klara_reset")
reset(job::HIPMCJob, X::Matrix{Float64}) =               # BasicMCJob.jl:197-201 reset(job, x): new initial values (D x N, one column per chain)
    check(ccall((:klara_reset, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, X), "klara_reset")
reset(job::HIPMCJob, x::AbstractVector) =                # README.md:68 `reset(job, [3.2, 9.4])`: every chain restarts from x
    reset(job, startmatrix(x, job.nchains))

function show(io::IO, job::HIPMCJob)                     # BasicMCJob.jl:281-295 prints parameter, model, sampler, tuner, range
    println(io, "HIPMCJob: ", job.nchains, " chains of ", job.ndims, " dimensions on libklara_hip")
    print(io, "  "); show(io, job.parameter.target)
    print(io, "\n  "); show(io, job.sampler)
    print(io, "\n  "); show(io, job.tuner)
    print(io, "\n  "); show(io, job.range)
end

# value matrix of chain c exactly as BasicContMuvParameterNState.value (D x npoststeps)
function chainvalue(job::HIPMCJob, c::Integer)
    n = Ref{Clonglong}(0)
    ccall((:klara_get_chain, lib), Cint, (Ptr{Cvoid}, Clonglong, Ptr{Float64}, Clonglong, Ref{Clonglong}),
          job.handle, c - 1, C_NULL, 0, n)
    v = newarray(Float64, job.ndims, n[])
    check(ccall((:klara_get_chain, lib), Cint, (Ptr{Cvoid}, Clonglong, Ptr{Float64}, Clonglong, Ref{Clonglong}),
                job.handle, c - 1, v, n[], n), "klara_get_chain")
    v
end

# output(job, c): a method of Klara.output (imported above) — the BasicContMuvParameterNState of chain c (BasicMCJob.jl:279; nstates/ParameterNStates/
# BasicContMuvParameterNState.jl:23-61: fields value, loglikelihood, logprior, logtarget, gradloglikelihood, gradlogprior,
# gradlogtarget, ..., diagnosticvalues, size, monitor, n, diagnostickeys) filled from the device history
function output(job::HIPMCJob, c::Integer=1)
    n = job.range.npoststeps
    monitor = fill(false, 13)
    monitor[1] = (job.monitor & MON_HISTORY) != 0           # value
    monitor[2] = monitor[3] = (job.monitor & MON_HIST_LLLP) != 0   # loglikelihood, logprior
    monitor[4] = (job.monitor & MON_HIST_LT) != 0           # logtarget
    monitor[7] = (job.monitor & MON_HIST_GRAD) != 0         # gradlogtarget
    dkeys = (job.monitor & MON_ACCEPT) != 0 ? [:accept] : Symbol[]
    ns = BasicContMuvParameterNState(job.ndims, n, monitor, dkeys)
    if monitor[1]; ns.value = chainvalue(job, c); end
    nc = Ref{Clonglong}(0)
    if monitor[4] || monitor[7]
        check(ccall((:klara_get_chain_fields, lib), Cint, (Ptr{Cvoid}, Clonglong, Ptr{Float64}, Ptr{Float64}, Clonglong, Ref{Clonglong}),
                    job.handle, c - 1, monitor[4] ? ns.logtarget : C_NULL, monitor[7] ? ns.gradlogtarget : C_NULL, n, nc), "klara_get_chain_fields")
    end
    if monitor[2]
        check(ccall((:klara_get_chain_likelihood_prior, lib), Cint, (Ptr{Cvoid}, Clonglong, Ptr{Float64}, Ptr{Float64}, Clonglong, Ref{Clonglong}),
                    job.handle, c - 1, ns.loglikelihood, ns.logprior, n, nc), "klara_get_chain_likelihood_prior")
    end
    if !isempty(dkeys)                                      # diagnosticvalues[1, i] = accept flag of saved step i (iterate/MALA.jl:112-117)
        nst = Ref{Clonglong}(0)
        ccall((:klara_get_accept_mask, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Clonglong, Ref{Clonglong}), job.handle, C_NULL, 0, nst)
        mask = newarray(UInt8, job.nchains, nst[])     # step-major rows of nchains bytes == column = step
        check(ccall((:klara_get_accept_mask, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Clonglong, Ref{Clonglong}), job.handle, mask, nst[], nst), "klara_get_accept_mask")
        for (i, step) in enumerate(job.range.postrange)
            step <= nst[] && (ns.diagnosticvalues[1, i] = mask[c, step] != 0)
        end
    end
    ns
end

# mean(chain) for every chain from the on-device running sums (stats/mean.jl:7-11): D x N
function chainmeans(job::HIPMCJob)
    s = newarray(Float64, job.ndims, job.nchains); q = similar(s); n = Ref{Clonglong}(0)
    check(ccall((:klara_get_chain_sums, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Clonglong}),
                job.handle, s, q, n), "klara_get_chain_sums")
    s ./ n[]
end
# acceptance rate of every chain over all transitions (stats/acceptance.jl:28-34 counts the saved steps' diagnostics)
function chainacceptance(job::HIPMCJob)
    a = newarray(UInt64, job.nchains); nst = Ref{UInt64}(0)
    check(ccall((:klara_get_accept_counts, lib), Cint, (Ptr{Cvoid}, Ptr{UInt64}, Ref{UInt64}), job.handle, a, nst), "klara_get_accept_counts")
    a ./ Float64(nst[])
end
# mcvar(chain, Val{:bm}) for every chain and dimension from the streaming batch means (bm_batchlen > 0): D x N, nbatches
function chainmcvar_bm(job::HIPMCJob)
    v = newarray(Float64, job.ndims, job.nchains); nb = Ref{Clonglong}(0)
    check(ccall((:klara_get_chain_bm, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Clonglong}), job.handle, v, nb), "klara_get_chain_bm")
    (v, nb[])
end
# the Philox key the job draws from and the number of resets so far (klara_hip.h klara_reset)
function streamkey(job::HIPMCJob)
    k = Ref{UInt64}(0); e = Ref{UInt64}(0)
    check(ccall((:klara_stream_key, lib), Cint, (Ptr{Cvoid}, Ref{UInt64}, Ref{UInt64}), job.handle, k, e), "klara_stream_key")
    (k[], e[])
end

# how the launches were issued so far (klara_get_launch_modes): (4-lane kernel alone, 8-lane kernel alone, device-decided pair), and per
# chain partition the last decision (0: 4 lanes, 1: 8 lanes) and the accepted proposals of the launch that took it
function launchmodes(job::HIPMCJob)
    counts = newarray(Int64, 3); lastmode = newarray(Int32, 4); lastaccepted = newarray(Int64, 4)
    check(ccall((:klara_get_launch_modes, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{Int64}), job.handle, counts, lastmode, lastaccepted),
          "klara_get_launch_modes")
    (counts, lastmode, lastaccepted)
end

# shader clock (MHz) during the job's last launch of a pair-transposed kernel (an in-kernel probe, klara_get_shader_clock); 0.0 when unknown
function shaderclock(job::HIPMCJob)
    mhz = newarray(Float64, 1)
    check(ccall((:klara_get_shader_clock, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, mhz), "klara_get_shader_clock")
    mhz[1]
end

# user-defined target: compile the closures' C text without a GPU; the compiler's message on failure
function check_custom_target(src::String, sampler::Integer, ndims::Integer)
    st = ccall((:klara_check_custom_target, lib), Cint, (Cstring, Cint, Cint), src, sampler, ndims)
    st == 0 || error(unsafe_string(ccall((:klara_compile_log, lib), Cstring, ())))
    true
end

# ---- multi-GPU: one process per GPU; the only exchange is the all-reduce of the pooled summaries (RCCL over xGMI)
mutable struct HIPComm; handle::Ptr{Cvoid}; end
function comm_unique_id()
    id = newarray(UInt8, 128)
    check(ccall((:klara_comm_unique_id, lib), Cint, (Ptr{UInt8},), id), "klara_comm_unique_id")
    id
end
function HIPComm(nranks::Integer, rank::Integer, id::Vector{UInt8}, device::Integer)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:klara_comm_init, lib), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Ptr{UInt8}, Cint), c, nranks, rank, id, device), "klara_comm_init")
    comm = HIPComm(c[])
    on_finalize(comm, x -> ccall((:klara_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.handle))
    comm
end
# (ranks, this rank, device) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)
function comm_info(comm::HIPComm)
    n = Ref{Cint}(0); r = Ref{Cint}(0); d = Ref{Cint}(0)
    check(ccall((:klara_comm_info, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}), comm.handle, n, r, d), "klara_comm_info")
    (Int(n[]), Int(r[]), Int(d[]))
end
# (sum x, sum x^2 per dimension over every chain of every GPU, accepted, transitions, saved samples, chains)
function gather_summaries(job::HIPMCJob, comm::HIPComm)
    s = newarray(Float64, job.ndims); q = similar(s)
    na = Ref{UInt64}(0); nt = Ref{UInt64}(0); ns = Ref{UInt64}(0); nc = Ref{UInt64}(0)
    check(ccall((:klara_gather_summaries, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}),
                job.handle, comm.handle, s, q, na, nt, ns, nc), "klara_gather_summaries")
    (s, q, na[], nt[], ns[], nc[])
end
# The pooled posterior moments over every chain of every GPU without the cancellation of sumsq/n - mean^2 (klara_gather_moments:
# per-chain (n, mean, M2) on the device, Chan's merge over chains, blocks and ranks): (mean, var, nsamples, accepted, transitions, chains)
function gather_moments(job::HIPMCJob, comm::HIPComm)
    m = newarray(Float64, job.ndims); m2 = similar(m)
    ns = Ref{UInt64}(0); na = Ref{UInt64}(0); nt = Ref{UInt64}(0); nc = Ref{UInt64}(0)
    check(ccall((:klara_gather_moments, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}),
                job.handle, comm.handle, m, m2, ns, na, nt, nc), "klara_gather_moments")
    (m, m2 ./ Float64(ns[]), ns[], na[], nt[], nc[])
end
# ... and of this job's chains alone (no communicator, no RCCL)
function pooledmoments(job::HIPMCJob)
    m = newarray(Float64, job.ndims); m2 = similar(m)
    ns = Ref{UInt64}(0); na = Ref{UInt64}(0); nt = Ref{UInt64}(0); nc = Ref{UInt64}(0)
    check(ccall((:klara_gather_moments, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}),
                job.handle, C_NULL, m, m2, ns, na, nt, nc), "klara_gather_moments")
    (m, m2 ./ Float64(ns[]), ns[], na[], nt[], nc[])
end
end # module
