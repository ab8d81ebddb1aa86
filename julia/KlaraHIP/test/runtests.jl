# For a maintainer with Julia 0.6, Klara and an MI355X (none of the three exists in the build image, so this file has never run):
#     LD_LIBRARY_PATH=klara.jl_amd/lib julia -e 'push!(LOAD_PATH, "julia"); include("julia/KlaraHIP/test/runtests.jl")'
# It is the module header's example, then the checks the Python mirror's tests make on the same job (tests/test_gpu_parity.py
# test_readme_flow_basic_mc_job): posterior moments of the README target (-dot(z, z): mean 0, variance 1/2) and Klara's own generics
# (run / reset / output / mean / acceptance) dispatching on the HIP job.
using Klara, KlaraHIP
using Base.Test

# --- the reference's README script (README.md:17-59) with its constructors swapped: BasicContMuvParameter -> HIPParameter, BasicMCJob -> HIPMCJob;
# the log-target closure becomes the device target family that states the same function
plogtarget = GaussDiagTarget(2)                       # README.md:23  plogtarget(z::Vector{Float64}) = -dot(z, z)
p = HIPParameter(:p, logtarget=plogtarget)            # README.md:30
@test isa(p, Klara.Parameter) && isa(p, Klara.ContinuousParameter) && isa(p, Klara.MultivariateParameter)
model = likelihood_model(p, false)                    # README.md:35 (Klara's own generator: assigns p.index)
@test p.index == 1 && model[:p] === p
sampler = MH(ones(2))                                 # README.md:39
mcrange = BasicMCRange(nsteps=10000, burnin=1000)     # README.md:43
v0 = Dict(:p=>[5.1, -0.9])                            # README.md:47
job = HIPMCJob(model, sampler, mcrange, v0)           # README.md:51
run(job)                                              # README.md:55
chain = output(job)                                   # README.md:59
@test isa(chain, BasicContMuvParameterNState) && size(chain.value) == (2, 9000)
@test maximum(abs.(mean(chain))) < 0.15
reset(job, [3.2, 9.4])                                # README.md:71
run(job)
@test output(job).value != chain.value
many = HIPMCJob(model, sampler, mcrange, v0; nchains=4096, seed=7)      # the same job as 4,096 replicas from the same start
run(many)
m, v, ns, na, nt, nc = pooledmoments(many)
@test nc == 4096 && maximum(abs.(m)) < 5e-3 && maximum(abs.(v .- 0.5)) < 5e-3

D, N = 100, 4096
p   = HIPParameter(:p, GaussDiagTarget(D))
job = HIPMCJob(p, MALA(0.1), BasicMCRange(nsteps=2000, burnin=1000), Dict(:p => randn(D, N));
               tuner=VanillaMCTuner(), outopts=Dict(:monitor => [:value], :diagnostics => [:accept]), seed=20260927)
@test isa(job, Klara.MCJob)
@test first(methods(output, (HIPMCJob,))).module == KlaraHIP       # a method of Klara.output, not a new function
@test output === Klara.output && run === Base.run && reset === Base.reset
run(job)
chain = output(job, 1)
@test isa(chain, BasicContMuvParameterNState) && size(chain.value) == (D, 1000)
@test 0.5 < acceptance(chain) <= 1.0
@test maximum(abs.(mean(chain))) < 0.5
m, v, ns, na, nt, nc = pooledmoments(job)
@test ns == 1000 * N && nc == N && nt == 2000 * N
@test maximum(abs.(m)) < 5e-3 && maximum(abs.(v .- 0.5)) < 5e-3
k0, e0 = streamkey(job)
reset(job); run(job)                                               # an independent replicate on the job's next key
k1, e1 = streamkey(job)
@test e1 == e0 + 1 && k1 != k0
@test output(job, 1).value != chain.value
jobs = [HIPMCJob(p, HMC(0.1, 10), BasicMCRange(nsteps=200, burnin=100), Dict(:p => randn(D, 64)); seed=s) for s in 1:2]
run(jobs)                                                          # Klara's run(::Vector{<:MCJob}) = map(run, jobs), jobs.jl:212
@test all(j -> size(output(j, 64).value) == (D, 100), jobs)
println("KlaraHIP: all checks passed")
