# KlaraHIP.jl — ccall binding of libklara_hip.so (see INTEGRATION.md). UNTESTED in the build image (no Julia).
module KlaraHIP
const lib = "libklara_hip"            # klara.jl_amd/lib/libklara_hip.so on LD_LIBRARY_PATH

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
    seed::UInt64; monitor::UInt32; steps_per_launch::Int32; stream::Ptr{Cvoid}
end

check(st::Cint, what) = st == 0 || error(what, ": ", unsafe_string(ccall((:klara_strerror, lib), Cstring, (Cint,), st)))

mutable struct HIPMCJob            # stands for N BasicMCJobs of one model
    handle::Ptr{Cvoid}; nchains::Int; ndims::Int; range   # range::BasicMCRange
end

# samplers/tuners are Klara's own structs: MH(σ), MALA(h), HMC(ε, L), SliceSampler(w, stepout),
# VanillaMCTuner(), AcceptanceRateMCTuner(rate)
function HIPMCJob(desc::KlaraDesc, X0::Matrix{Float64}, range)   # X0 is D × N (column = chain) == N×D row-major
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:klara_create, lib), Cint, (Ref{KlaraDesc}, Ref{Ptr{Cvoid}}), desc, h), "klara_create")
    job = HIPMCJob(h[], size(X0, 2), size(X0, 1), range)
    finalizer(j -> ccall((:klara_destroy, lib), Cint, (Ptr{Cvoid},), j.handle), job)
    check(ccall((:klara_set_state, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, X0), "klara_set_state")
    job
end

run(job::HIPMCJob) =                                     # BasicMCJob.jl:212-244 for all chains
    check(ccall((:klara_run, lib), Cint, (Ptr{Cvoid}, Clonglong), job.handle, job.range.nsteps), "klara_run")

reset(job::HIPMCJob) = check(ccall((:klara_reset, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, C_NULL), "klara_reset")
reset(job::HIPMCJob, X::Matrix{Float64}) =
    check(ccall((:klara_reset, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}), job.handle, X), "klara_reset")

# output(job)[c]: value matrix of chain c exactly as BasicContMuvParameterNState.value (D × npoststeps)
function chainvalue(job::HIPMCJob, c::Integer)
    n = Ref{Clonglong}(0)
    ccall((:klara_get_chain, lib), Cint, (Ptr{Cvoid}, Clonglong, Ptr{Float64}, Clonglong, Ref{Clonglong}),
          job.handle, c - 1, C_NULL, 0, n)
    v = Matrix{Float64}(undef, job.ndims, n[])
    check(ccall((:klara_get_chain, lib), Cint, (Ptr{Cvoid}, Clonglong, Ptr{Float64}, Clonglong, Ref{Clonglong}),
                job.handle, c - 1, v, n[], n), "klara_get_chain")
    v
end

# mean(chain) for every chain from the on-device running sums (stats/mean.jl:7-11): D × N
function chainmeans(job::HIPMCJob)
    s = Matrix{Float64}(undef, job.ndims, job.nchains); q = similar(s); n = Ref{Clonglong}(0)
    check(ccall((:klara_get_chain_sums, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Clonglong}),
                job.handle, s, q, n), "klara_get_chain_sums")
    s ./ n[]
end
# mcvar(chain, Val{:bm}) for every chain and dimension from the streaming batch means (desc.bm_batchlen > 0): D × N, nbatches
function chainmcvar_bm(job::HIPMCJob)
    v = Matrix{Float64}(undef, job.ndims, job.nchains); nb = Ref{Clonglong}(0)
    check(ccall((:klara_get_chain_bm, lib), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Clonglong}), job.handle, v, nb), "klara_get_chain_bm")
    (v, nb[])
end

# user-defined target (desc.target = 4): compile the closures' C text without a GPU; the compiler's message on failure
function check_custom_target(src::String, sampler::Integer, ndims::Integer)
    st = ccall((:klara_check_custom_target, lib), Cint, (Cstring, Cint, Cint), src, sampler, ndims)
    st == 0 || error(unsafe_string(ccall((:klara_compile_log, lib), Cstring, ())))
    true
end

# ---- multi-GPU: one process per GPU; the only exchange is the all-reduce of the pooled summaries (RCCL over xGMI)
mutable struct HIPComm; handle::Ptr{Cvoid}; end
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    check(ccall((:klara_comm_unique_id, lib), Cint, (Ptr{UInt8},), id), "klara_comm_unique_id")
    id
end
function HIPComm(nranks::Integer, rank::Integer, id::Vector{UInt8}, device::Integer)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:klara_comm_init, lib), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Ptr{UInt8}, Cint), c, nranks, rank, id, device), "klara_comm_init")
    comm = HIPComm(c[])
    finalizer(x -> ccall((:klara_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.handle), comm)
    comm
end
# (sum x, sum x^2 per dimension over every chain of every GPU, accepted, transitions, saved samples, chains)
function gather_summaries(job::HIPMCJob, comm::HIPComm)
    s = Vector{Float64}(undef, job.ndims); q = similar(s)
    na = Ref{UInt64}(0); nt = Ref{UInt64}(0); ns = Ref{UInt64}(0); nc = Ref{UInt64}(0)
    check(ccall((:klara_gather_summaries, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}, Ref{UInt64}),
                job.handle, comm.handle, s, q, na, nt, ns, nc), "klara_gather_summaries")
    (s, q, na[], nt[], ns[], nc[])
end
end # module
