// klara_api.hip — host side of libklara_hip.so: handle management, validation, launches, read-back.
// The ABI and the reference lines each entry point replaces are documented in include/klara_hip.h.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <rccl/rccl.h>      // types only: librccl.so is dlopen'ed on first use (klara_comm_*)
#include <dlfcn.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_map>
#include <vector>
#include "klara_launch.h"

// user-defined targets keep the whole vector in one lane (klara_custom.h): pow2ceil(D) elements per lane
#ifndef KLARA_CUSTOM_MAXD
#define KLARA_CUSTOM_MAXD 1024           // (round 6: 64 lanes x 16 elements of the staged form; 256 before)
#endif

// device-decided kernel choice (KAuto): launches shorter than this are issued as one kernel chosen on the host
#ifndef KLARA_AUTO_PAIR_MIN_STEPS
#define KLARA_AUTO_PAIR_MIN_STEPS 4
#endif

thread_local hipFuncAttributes* klara_attr_query = nullptr;     // see klara_launch.h klara_go

#define HIPCHK(expr)                                                                   \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) return (e__ == hipErrorOutOfMemory) ? KLARA_ERR_NOMEM : KLARA_ERR_HIP; \
    } while (0)

struct klara_handle {
    klara_desc d;
    // layout
    int kind, G, E;            // kind 0: group layout (G lanes x E elems); kind 1: MFMA (E = NE); kind 2: logistic row split
    int RS = 1;                // kind 2: lanes sharing one chain's data rows (G = 1 there)
    // device buffers
    double *X = nullptr, *GR = nullptr, *LT = nullptr;
    double* tune_step = nullptr;
    long long *tune_acc = nullptr, *tune_prop = nullptr, *tune_tot = nullptr;
    double *da_epsbar = nullptr, *da_hbar = nullptr;
    unsigned long long* pooled_acc = nullptr;
    uint8_t* accept = nullptr; long long accept_cap = 0;
    unsigned long long* naccept = nullptr;
    double *sum = nullptr, *sumsq = nullptr;
    long long* held = nullptr;      // running sums in sojourn form: saved steps at the current state not yet in sum / sumsq (KParams::held)
    double* hist = nullptr; long long hist_cols = 0;     // hist_cols: columns of the history buffers (= ring when > 0 and ring is set)
    bool ring = false;                                   // the history buffers hold the last hist_cols saved steps only
    // streaming autocovariances (acov_maxlag > 0): W = maxlag + 1 lags; [k][series] layouts
    int acov_W = 0; double *acov_S = nullptr, *acov_head = nullptr, *acov_tail = nullptr, *acov_total = nullptr, *acov_near = nullptr; long long acov_n = 0;
    double *hist_lt = nullptr, *hist_g = nullptr, *hist_ll = nullptr, *hist_lp = nullptr;
    unsigned long long* clock_probe = nullptr;        // pair-transposed kernels: (s_memtime, s_memrealtime) at the end / start of one workgroup of the last launch
    bool pair_enqueued = false;                       // a launch of this handle has enqueued both kernel families (their one-time scratch set-up is behind us)
    int custom_wpb = 4;                               // staged closures: wavefronts per workgroup (2 where four wavefronts' rows do not fit the LDS)
    int custom_rows = 2;                              // staged closures: vectors per chain in LDS (3 for the likelihood + prior form)
    int* err = nullptr; int* flag_host = nullptr;     // error flag as the kernels address it; the same word as the host reads it (null: err is device memory)
    double *vecparam = nullptr, *gw = nullptr, *gmu = nullptr, *lX = nullptr, *ly = nullptr, *Pfrag = nullptr,
           *hY = nullptr, *hxc = nullptr;
    double* pooled_out = nullptr;   // 2*D doubles + 1 u64 scratch for pooled summaries
    double* pool_partial = nullptr; // KLARA_POOL_BLOCKS x (2 D doubles + 1 u64): stage-1 partials of the pooled summaries
    double* cdata = nullptr; KlaraJit* jit = nullptr;   // user-defined target: data block, run-time compiled kernels
    bool jit_pair = false;          // ... given as a pair closure: run-time compiled k_diagt instantiations (layout kind 3)
    // streaming batch means (bm_batchlen > 0): running sum at the last batch boundary, Welford mean / M2 of the batch means
    double *bm_prev = nullptr, *bm_mean = nullptr, *bm_m2 = nullptr; long long bm_count = 0;
    KParams* d_params = nullptr;    // device copy of the handle's static kernel parameters
    double lpconst = 0.0;
    bool dense_mu = false;          // dense target with a mean: Pfrag carries mu behind the matrix fragments
    int logit_nblocks = 0;          // layout kind 5 (klara_logit_mfma.h): row blocks of the fragment stream in Pfrag; ly holds the zero-padded responses
    // run state
    bool have_state = false;
    unsigned long long epoch = 0;   // klara_reset calls so far: the Philox key of the job is seed + epoch * KLARA_EPOCH_KEY_STRIDE
    long long steps_done = 0;       // transitions since set_state/reset (= global transition index)
    long long nsaved = 0;           // postrange steps passed so far
    // host mirror of the pooled tuner counters (decides where launches must end)
    long long m_prop = 0, m_tot = 0;
    hipStream_t stream = nullptr; bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; long long last_launches = 0; bool timed = false;
    // layout kind 3: the chain groups are cut into `nparts` contiguous partitions, partition j > 0 runs on its own
    // internal stream.  Chains are independent, so partition j's transition t+1 only follows its own transition t; the
    // streams drift apart and one partition's kernel fills the SIMDs while the other's drains / ramps up.
    int nparts = 1; hipStream_t side[3] = { nullptr, nullptr, nullptr }; hipEvent_t fork_ev = nullptr, join_ev[3] = { nullptr, nullptr, nullptr };
    // layout kind 3, untuned MH / MALA with 17 <= D <= 104: the 4-lanes-per-chain kernels are available as well (np4 pairs per lane).
    // They sum in the 8-lane order, so which of the two kernel families runs a launch changes no bit; with running sums on, the
    // choice is taken on the device launch by launch (KAuto, klara_diagt.h): auto_cells = [partition][launch parity] decision,
    // auto_ctr = [partition] launch counter, auto_mirror = host-visible {mode, accepted} per partition (may be null).
    bool q4_ok = false; int np4 = 0;
    int* auto_cells = nullptr; unsigned long long* auto_ctr = nullptr; int* auto_mirror = nullptr; int* auto_mirror_dev = nullptr;
    long long launch_idx = 0;
    double auto_threshold = 0.06;   // acceptance above which a launch keeps resident sums (8 lanes) — the measured crossover: the 4-lane kernels' folds are atomic adds
                                    // since round 4 (flat to ~4.5 % acceptance, saturating the atomic units beyond: profiles/r4_acceptance_cost_atomic.txt; round 3's
                                    // read-modify-write folds crossed at 0.12, profiles/r3_acceptance_cost_probe.txt)
    int query_lanes = 4;            // klara_get_kernel_attributes: which of the two kernel families to report
    long long n_launch_mode[3] = { 0, 0, 0 };   // launches issued as: forced / single 4-lane, forced / single 8-lane, device-decided pair
};

static int cnt_predicate(const klara_desc& d)
{
    if (d.sampler == KLARA_SAMPLER_MH || d.sampler == KLARA_SAMPLER_SLICE) return d.verbose != 0;
    return (d.tuner == KLARA_TUNER_VANILLA && d.verbose) || d.tuner == KLARA_TUNER_ACCEPT_RATE ||
           (d.tuner == KLARA_TUNER_DUAL_AVERAGING && d.verbose);
}

static int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// layout kind 3 serves the jobs whose transition is pure elementwise work plus three sums (see klara_diagt.h):
// diagonal Gaussian, every sampler, every tuner, any monitor
static bool diagt_eligible(const klara_desc& d)
{
    if (d.target != KLARA_TARGET_GAUSS_DIAG) return false;
    // (D <= 16: the group layout already puts a chain on <= 4 lanes, 16..64 chains per wavefront)
    if (d.ndims < 17 || d.ndims > 2 * 64 * KLARA_DIAGT_NP_MAX) return false;       // (Q = 8, 16, 32 or 64 lanes per chain)
    if (const char* s = getenv("KLARA_LAYOUT_KIND")) { if (atoi(s) == 0) return false; }
    if (getenv("KLARA_LAYOUT_E")) return false;
    return true;
}

// the slice sampler on the pair-transposed layout with nothing counting: every lane takes its elements through a whole launch on
// its own (klara_diagt_slice.h); a wavefront only waits for its slowest lane once per element slot and launch, so longer launches waste less
static bool slice_free_eligible(const klara_desc& d)
{
    static const bool lockstep = getenv("KLARA_SLICE_LOCKSTEP") != nullptr;
    const uint32_t lane_local = KLARA_MON_ACCEPT | KLARA_MON_SUMMARIES | KLARA_MON_HISTORY | KLARA_MON_HIST_LT;
    const bool values_kept = (d.monitor & KLARA_MON_HISTORY) != 0 || d.acov_maxlag > 0;       // (the log-target history is formed from the saved values)
    // (nothing counts AND nothing tunes — a pooled or dual-averaging tuner setting, verbose or not, runs k_diagt<SLICE> at its 32 transitions per
    // launch: the launch length and the kernel choice come from this one predicate)
    const bool plain = !cnt_predicate(d) && d.tuner_mode == KLARA_TUNE_PER_CHAIN && d.tuner != KLARA_TUNER_DUAL_AVERAGING;
    return diagt_eligible(d) && d.sampler == KLARA_SAMPLER_SLICE && plain && !lockstep &&
           (d.monitor & ~lane_local) == 0 && (!(d.monitor & KLARA_MON_HIST_LT) || values_kept);
}
// transitions per launch when klara_desc.steps_per_launch = 0
static long long default_steps_per_launch(const klara_desc& d) { return slice_free_eligible(d) ? KLARA_DEFAULT_STEPS_PER_LAUNCH_SLICE : KLARA_DEFAULT_STEPS_PER_LAUNCH; }

// untuned MH / MALA on the pair-transposed layout up to D = 104: 4 lanes per chain, 16 chains per wavefront, NP = ceil(D/8) in 3..13
// (klara_launch.h) — kernels that keep no resident running sums (a moving chain's sums are folded into memory by atomic adds)
static bool q4_eligible(const klara_desc& d)
{
    const bool plain = !cnt_predicate(d) && d.tuner_mode == KLARA_TUNE_PER_CHAIN && d.tuner != KLARA_TUNER_DUAL_AVERAGING;
    // (round 5: HMC too while no saved-sample monitor is on — 12.5 of 13 pair slots real at D = 100 instead of 6.25 of 7, the reductions in the 8-lane order as
    // for MH / MALA; with running sums or a history the 8-lane kernels keep it: at HMC's acceptance every transition would fold)
    // (a non-unit diagonal beyond 9 pairs per lane spills: those stay on 8 lanes)
    const bool hmc_ok = d.sampler == KLARA_SAMPLER_HMC && (d.monitor & ~(uint32_t)KLARA_MON_ACCEPT) == 0 && d.acov_maxlag == 0 &&
                        ((d.gauss_w == nullptr && d.gauss_mu == nullptr) || d.ndims <= 72) && !getenv("KLARA_DIAGT_NO_Q4_HMC");
    return diagt_eligible(d) && plain && (d.sampler == KLARA_SAMPLER_MH || d.sampler == KLARA_SAMPLER_MALA || hmc_ok) && d.ndims <= 104 &&
           !getenv("KLARA_DIAGT_NO_Q4");
}

// layout kind 4 (klara_hiert.h): MH / MALA / HMC on the hierarchical target, 8 lanes per chain, 4 units per lane
static bool hiert_eligible(const klara_desc& d)
{
    if (d.target != KLARA_TARGET_HIER_NORMAL || d.sampler == KLARA_SAMPLER_SLICE) return false;
    if (d.hier_nunits < 9 || d.hier_nunits > 32) return false;
    if (const char* s = getenv("KLARA_LAYOUT_KIND")) { if (atoi(s) == 0) return false; }
    if (getenv("KLARA_LAYOUT_E")) return false;
    return true;
}

// a user target given as a pair closure (`#define KLARA_USER_PAIR_TARGET 1` + klara_user_pair, include/klara_hip.h)
static bool pair_source(const char* src) { return src != nullptr && strstr(src, "KLARA_USER_PAIR_TARGET") != nullptr && strstr(src, "KLARA_PAIR_AS_WHOLE") == nullptr; }
static bool pair_form(const klara_desc& d) { return d.target == KLARA_TARGET_CUSTOM && pair_source(d.custom_src); }
// ... that the pair-transposed kernels do not serve (fewer than 9 pairs, the slice sampler): it runs as a whole-vector closure, the sum over its pairs
// formed by klara_custom_compose.h (round 5; these jobs used to be refused)
// (round 6: the slice sampler takes pair closures on the few-lanes kernels too — k_diagt<SLICE, .., USERPAIR> —; KLARA_PAIR_SLICE_AS_WHOLE=1 keeps the whole-vector form of round 5)
static bool pair_as_whole(const char* src, int sampler, int ndims)
{
    return pair_source(src) && (ndims < 17 || (sampler == KLARA_SAMPLER_SLICE && getenv("KLARA_PAIR_SLICE_AS_WHOLE") != nullptr));
}
static std::string pair_as_whole_source(const char* src) { return std::string("#define KLARA_PAIR_AS_WHOLE 1\n") + src; }

// Whole-vector closures (klara_custom.h).  Up to 32 dimensions a lane keeps the whole vector in registers (one chain per lane); beyond,
// the chain is spread over G lanes with E = 2 ceil(D / 2G) <= 16 elements each and evaluations read the vector from the chain's row of
// LDS (STAGED).  G: enough lanes for 16 elements per lane, and a workgroup's rows (4 wavefronts x 64 / G chains) within the 56 KB of
// dynamic LDS a launch gets without asking.  `lanes` = 1 keeps one chain per lane at any D (the library's own heavy closures — the
// logistic regression beyond 8 parameters, the dense Gaussian beyond 128 dimensions — cost O(n D) / O(D^2) per evaluation, and the
// G identical evaluations of the staged form would multiply exactly that part; KLARA_CUSTOM_LANES=1 in the environment for a user's).
static size_t custom_stage_bytes(int D, int G, int nrows, int wpb)        // (klara_custom.h klara_custom_stage_stride)
{
    int s = nrows * ((D + 1) & ~1) + 2;
    if ((2 * s) % 64 == 0) s += 2;
    return (size_t)wpb * (size_t)(64 / G) * (size_t)s * sizeof(double);
}
// wpb: wavefronts per workgroup of the staged kernels — 4, or 2 where four wavefronts' rows would not fit and force more lanes per chain
static void custom_layout(int D, int lanes, bool lik_prior, int* G, int* E, int* wpb)
{
    *wpb = 4;
    if (const char* s = getenv("KLARA_CUSTOM_LANES")) { const int v = atoi(s); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) lanes = v; }
    if (lanes == 1 || (lanes == 0 && D <= 32)) { *G = 1; *E = pow2ceil(D < 2 ? 2 : D); return; }
    int g = lanes > 1 ? lanes : 4;
    while (g < 64 && (D + 2 * g - 1) / (2 * g) > 8) g *= 2;                      // at most 16 elements per lane (round 6: up to 64 lanes — one chain per wavefront —: D <= 1024)
    const int nrows = lik_prior ? 3 : 2;
    int w = 4;
    if (const char* s = getenv("KLARA_CUSTOM_WPB")) { const int v = atoi(s); if (v == 2 || v == 4) w = v; }
    else if (g >= 8 && custom_stage_bytes(D, g, nrows, 4) > KLARA_LDS_DEFAULT_DYNAMIC) w = 2;     // (measured: D = 128 +20 %, D = 256 +21 %; at 4 lanes x 16 elements the kernels spill and 8 x 8 on four wavefronts is faster)
    while (g < 64 && custom_stage_bytes(D, g, nrows, w) > KLARA_LDS_DEFAULT_DYNAMIC) g *= 2;
    *G = g; *E = 2 * ((D + 2 * g - 1) / (2 * g)); *wpb = w;
}
static bool custom_lik_prior(const char* src) { return src != nullptr && strstr(src, "KLARA_USER_LIKELIHOOD_PRIOR") != nullptr; }

// the dense Gaussian beyond the LDS-resident layouts (D = 129 .. 256) stays on the matrix cores for every sampler — HMC (every tuner), MALA, MH
// and (round 5) the slice sampler (klara_dense_big.h); beyond D = 256 the closure form (klara_create)
static bool dense_streamed(const klara_desc& d)
{
    return d.target == KLARA_TARGET_GAUSS_DENSE && d.ndims > 128 && d.ndims <= 256 && getenv("KLARA_DENSE_NO_STREAM") == nullptr &&
           (d.sampler != KLARA_SAMPLER_SLICE || getenv("KLARA_DENSE_SLICE_NO_STREAM") == nullptr);
}

// the dense Gaussian beyond D = 256 (round 6): the tile of 16 chains on a WORKGROUP of W = 4, 8, 12 or 16 wavefronts that deal the ceil(D / 16) row tiles
// of P evenly, 2 .. 4 each (klara_dense_split.h, layout kind 6; MH, MALA, HMC with every tuner; 257 <= D <= 1024).  KLARA_DENSE_SPLIT=1 in the environment puts the smaller
// dense targets on it as well (measurements, tests).
static bool dense_split(const klara_desc& d)
{
    if (d.target != KLARA_TARGET_GAUSS_DENSE || d.ndims > 64 * 16) return false;
    if (getenv("KLARA_DENSE_NO_SPLIT") != nullptr) return false;
    if (d.ndims > 256) return true;
    const char* s = getenv("KLARA_DENSE_SPLIT");
    return s != nullptr && atoi(s) != 0;
}

// the logistic regression beyond 16 parameters on the matrix cores (klara_logit_mfma.h, layout kind 5): X p and X' (y - 1/(1+exp(-Xp))) of 16 chains per
// wavefront as two MFMA passes over streamed fragments of X; every sampler, every tuner, the monitors of the dense layouts.  The likelihood / prior
// history keeps the closure form (klara_create).
static bool logit_mfma_eligible(const klara_desc& d)
{
    // (also 9 .. 16 parameters whose rows, padded to 16 columns, do not fit the LDS of the row-split kernels: the stream has no such limit)
    const bool beyond_rowsplit = d.ndims > 16 || (d.ndims > 8 && (size_t)d.logit_ndata * 17 > KLARA_LOGIT_MAX_LDS_DOUBLES);
    // (the kernels address the fragment stream with 32-bit byte offsets: 2 x 16 ceil(n / 16) x 4 NE doubles stay below 2 GB — 2 million rows at 128 parameters)
    const size_t stream_bytes = 2 * 16 * (((size_t)d.logit_ndata + 31) / 32 * 2) * 4 * (8 * (((size_t)d.ndims + 31) / 32)) * sizeof(double);
    if (stream_bytes >= ((size_t)1 << 31)) return false;
    return d.target == KLARA_TARGET_LOGISTIC && beyond_rowsplit && d.ndims <= 256 && d.logit_ndata >= 1 &&
           !(d.monitor & KLARA_MON_HIST_LLLP) && getenv("KLARA_LOGIT_NO_MFMA") == nullptr;
}

static klara_status select_layout(const klara_desc& d, int* kind, int* G, int* E, int custom_lanes = 0, int* custom_wpb = nullptr)
{
    if (logit_mfma_eligible(d)) { *kind = 5; *G = 4; *E = 8 * ((d.ndims + 31) / 32); return KLARA_OK; }
    if (hiert_eligible(d)) { *kind = 4; *G = 8; *E = 8; return KLARA_OK; }
    const int D = d.ndims;
    if (dense_split(d)) { *kind = 6; *G = klara_split_waves(D); *E = klara_split_new(D); return KLARA_OK; }      // G: wavefronts per tile of 16 chains (klara_dense_split.h klara_split_waves)
    if (d.target == KLARA_TARGET_GAUSS_DENSE) {
        *kind = 1; *G = 4;
        if (D <= 32) *E = 8; else if (D <= 64) *E = 16; else if (D <= 100) *E = 25; else if (D <= 128) *E = 32;
        else if (dense_streamed(d)) *E = 8 * ((D + 31) / 32);       // HMC to D = 256 (NE = 40, 48, 56, 64): P streamed from memory, momentum in LDS (klara_dense_big.h)
        else return KLARA_ERR_UNSUPPORTED;
        return KLARA_OK;
    }
    *kind = 0;
    if (d.target == KLARA_TARGET_CUSTOM && pair_form(d)) {
        // pair closure (klara_diagt.h USERPAIR): the pair-transposed layout, Q = 8 / 16 / 32 lanes per chain
        if (D < 17 || D > 2 * 64 * KLARA_DIAGT_NP_MAX) return KLARA_ERR_UNSUPPORTED;       // (fewer than 9 pairs: use the whole-vector form)
        const int Q = D <= 128 ? 8 : (D <= 256 ? 16 : (D <= 512 ? 32 : 64));
        *kind = 3; *G = Q; *E = 2 * ((D + 2 * Q - 1) / (2 * Q));
        return KLARA_OK;
    }
    if (d.target == KLARA_TARGET_CUSTOM) {       // whole-vector closure: one chain per lane, or staged through LDS on G lanes (klara_custom.h)
        if (D > KLARA_CUSTOM_MAXD) return KLARA_ERR_UNSUPPORTED;
        int wpb = 4;
        custom_layout(D, custom_lanes, custom_lik_prior(d.custom_src), G, E, &wpb);
        if (*G > 1 && (*E > 16 || custom_stage_bytes(D, *G, custom_lik_prior(d.custom_src) ? 3 : 2, wpb) > KLARA_LDS_DEFAULT_DYNAMIC)) { *G = 1; *E = pow2ceil(D); wpb = 4; }
        if (custom_wpb) *custom_wpb = wpb;
        return KLARA_OK;
    }
    if (d.target == KLARA_TARGET_LOGISTIC) {
        *G = 1;   // every lane holds the whole parameter vector; klara_create may turn on the row split (kind 2)
        if (D <= 2) *E = 2; else if (D <= 4) *E = 4; else if (D <= 8) *E = 8; else if (D <= 16) *E = 16; else return KLARA_ERR_UNSUPPORTED;
        return KLARA_OK;
    }
    // diagonal Gaussian and nothing tunes: the pair-transposed layout
    // (klara_diagt.h), Q = 8 lanes per chain, NP element pairs per lane
    if (diagt_eligible(d)) {
        // lanes per chain of the layout, i.e. of the summation order klara_get_layout reports.  (Untuned MH / MALA up to D = 104 also
        // run on 4-lane kernels that reproduce the 8-lane order: q4_eligible.)
        const int Q = D <= 128 ? 8 : (D <= 256 ? 16 : (D <= 512 ? 32 : 64));
        const int np = (D + 2 * Q - 1) / (2 * Q);                                 // NP = ceil(D/2 / Q) exactly (see klara_diagt.h)
        if (np >= 2 && np <= KLARA_DIAGT_NP_MAX) { *kind = 3; *G = Q; *E = 2 * np; return KLARA_OK; }
    }
    // diagonal Gaussian: E elements per lane, G lanes; optional override for layout experiments
    // D <= 128: E = 2 or 4, whichever wastes fewer lanes; on a tie E = 4 (twice the chains per wavefront
    // amortise the per-wave fixed work — measured 80 us vs 90 us per launch at D = 100, 65,536 chains)
    int e = (D <= 256) ? 4 : (D <= 512) ? 8 : 0;
    if (D <= 128 && pow2ceil((D + 1) / 2) * 2 < pow2ceil((D + 3) / 4) * 4) e = 2;
    if (const char* s = getenv("KLARA_LAYOUT_E")) {
        const int v = atoi(s);
        if ((v == 2 || v == 4 || v == 8) && (D + v - 1) / v <= 64) e = v;
    }
    if (e == 0 || (d.target == KLARA_TARGET_HIER_NORMAL && e > 4)) return KLARA_ERR_UNSUPPORTED;
    *E = e; *G = pow2ceil((D + e - 1) / e);
    return KLARA_OK;
}

static klara_status validate(const klara_desc* d)
{
    if (!d) return KLARA_ERR_INVALID_ARG;
    if (d->struct_size != sizeof(klara_desc) || d->abi_version != KLARA_ABI_VERSION) return KLARA_ERR_INVALID_ARG;
    if (d->nchains <= 0 || d->ndims <= 0 || d->chain_offset < 0) return KLARA_ERR_INVALID_ARG;
    if (d->sampler < KLARA_SAMPLER_MH || d->sampler > KLARA_SAMPLER_SLICE) return KLARA_ERR_INVALID_ARG;
    if (d->target < KLARA_TARGET_GAUSS_DIAG || d->target > KLARA_TARGET_CUSTOM) return KLARA_ERR_INVALID_ARG;
    if (d->tuner < KLARA_TUNER_VANILLA || d->tuner > KLARA_TUNER_DUAL_AVERAGING) return KLARA_ERR_INVALID_ARG;
    if (d->tuner == KLARA_TUNER_DUAL_AVERAGING) {                // DualAveragingMCTuner.jl:65-70
        if (!(d->targetrate > 0.0 && d->targetrate < 1.0) || d->da_nadapt <= 0 || !(d->da_eps0bar > 0.0) || d->da_t0 <= 0 ||
            !(d->da_gamma > 0.0))
            return KLARA_ERR_INVALID_ARG;
        if (d->sampler != KLARA_SAMPLER_HMC || d->tuner_mode != KLARA_TUNE_PER_CHAIN) return KLARA_ERR_UNSUPPORTED;
    }
    if (d->tuner_mode != KLARA_TUNE_PER_CHAIN && d->tuner_mode != KLARA_TUNE_POOLED) return KLARA_ERR_INVALID_ARG;
    // BasicMCRange.jl:22-24
    if (d->burnin < 0 || d->thinning < 1 || d->thinning > 0x7fffffff || d->nsteps <= d->burnin) return KLARA_ERR_INVALID_ARG;
    // VanillaMCTuner / AcceptanceRateMCTuner.jl:32-33
    if (d->period <= 0) return KLARA_ERR_INVALID_ARG;
    if (d->nstreams < 0 || d->nstreams > 4) return KLARA_ERR_INVALID_ARG;
    if (d->tuner == KLARA_TUNER_ACCEPT_RATE && !(d->targetrate > 0.0 && d->targetrate < 1.0)) return KLARA_ERR_INVALID_ARG;
    switch (d->sampler) {
    case KLARA_SAMPLER_MH:
        if (!d->mh_sigma) return KLARA_ERR_INVALID_ARG;
        for (int i = 0; i < d->ndims; ++i) if (!(d->mh_sigma[i] > 0.0)) return KLARA_ERR_INVALID_ARG;
        break;
    case KLARA_SAMPLER_MALA:                                     // MALA.jl:65
        if (!(d->driftstep > 0.0)) return KLARA_ERR_INVALID_ARG;
        break;
    case KLARA_SAMPLER_HMC:                                      // HMC.jl:94-95
        if (!(d->leapstep > 0.0) || d->nleaps <= 0) return KLARA_ERR_INVALID_ARG;
        break;
    default:                                                     // SliceSampler.jl:27
        if (!d->slice_widths) return KLARA_ERR_INVALID_ARG;
        for (int i = 0; i < d->ndims; ++i) if (!(d->slice_widths[i] > 0.0)) return KLARA_ERR_INVALID_ARG;
        break;
    }
    if (d->target == KLARA_TARGET_GAUSS_DENSE && !d->gauss_prec) return KLARA_ERR_INVALID_ARG;
    if (d->target == KLARA_TARGET_HIER_NORMAL &&
        (!d->hier_Y || !d->hier_xc || d->hier_nunits <= 0 || d->hier_ntimes <= 0 || d->hier_ntimes > 16 ||
         d->ndims != 2 * d->hier_nunits + 5 || !(d->hier_prior_prec >= 0.0) || !(d->hier_gamma_a >= 0.0) ||
         !(d->hier_gamma_b >= 0.0)))
        return KLARA_ERR_INVALID_ARG;
    if (d->target == KLARA_TARGET_LOGISTIC &&
        (!d->logit_X || !d->logit_y || d->logit_ndata <= 0 || !(d->logit_lambda > 0.0)))
        return KLARA_ERR_INVALID_ARG;
    if (d->target == KLARA_TARGET_CUSTOM && (!d->custom_src || d->custom_ndata < 0 || (d->custom_ndata > 0 && !d->custom_data)))
        return KLARA_ERR_INVALID_ARG;
    if (d->hist_ring_cols < 0 || d->acov_maxlag < 0 || d->acov_maxlag > 127 || d->sparse_moves < 0 || d->sparse_moves > 2) return KLARA_ERR_INVALID_ARG;
    if (d->bm_batchlen < 0 || (d->bm_batchlen > 0 && !(d->monitor & KLARA_MON_SUMMARIES))) return KLARA_ERR_INVALID_ARG;
    if (d->steps_per_launch < 0 || d->tuner_score < 0 || d->tuner_score > 1) return KLARA_ERR_INVALID_ARG;   // (int32: a launch length always fits KLaunch::nsteps)
    return KLARA_OK;
}

// ---- device allocations, optionally between canaries.  Every kernel addresses the chains' vectors through buffer
// resources whose num_records clamps ragged tails and whose KLARA_BUF_OOB offsets mask padding lanes: a window sized one element too
// large, or an offset that is not out of range when it should be, reads or writes a neighbour's memory silently.  KLARA_DEBUG_CANARY=1 in the
// environment puts 4 KiB of a signalling-NaN pattern (0x7FF4DEAD7FF4DEAD at every 8-byte aligned address) directly before and after
// every device array: a stray WRITE changes the pattern (checked when the handle is destroyed: klara_destroy returns KLARA_ERR_STATE; and
// by klara_selftest_canary for everything alive), a stray READ brings a NaN into a result that the parity tests compare bit for bit.
static const size_t KCANARY = 4096;
static const unsigned long long KCANARY_WORD = 0x7FF4DEAD7FF4DEADull;
static bool canary_on()
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("KLARA_DEBUG_CANARY"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
struct CanaryRec { char* base; size_t bytes; };
static std::mutex canary_mu;
static std::unordered_map<void*, CanaryRec> canary_map;
static void canary_pattern(std::vector<unsigned char>& buf, size_t addr0)          // the pattern as it lies at device addresses addr0, addr0 + 1, ...
{
    for (size_t i = 0; i < buf.size(); ++i) buf[i] = (unsigned char)(KCANARY_WORD >> (8 * ((addr0 + i) & 7)));
}
static hipError_t canary_fill(const CanaryRec& r)
{
    std::vector<unsigned char> pat(KCANARY);
    canary_pattern(pat, (size_t)r.base);
    hipError_t e = hipMemcpy(r.base, pat.data(), KCANARY, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    canary_pattern(pat, (size_t)r.base + KCANARY + r.bytes);
    return hipMemcpy(r.base + KCANARY + r.bytes, pat.data(), KCANARY, hipMemcpyHostToDevice);
}
static bool canary_intact(const CanaryRec& r)
{
    std::vector<unsigned char> got(KCANARY), pat(KCANARY);
    for (int side = 0; side < 2; ++side) {
        const char* at = side == 0 ? r.base : r.base + KCANARY + r.bytes;
        if (hipMemcpy(got.data(), at, KCANARY, hipMemcpyDeviceToHost) != hipSuccess) return false;
        canary_pattern(pat, (size_t)at);
        if (memcmp(got.data(), pat.data(), KCANARY) != 0) return false;
    }
    return true;
}
static hipError_t dalloc_bytes(void** p, size_t bytes)
{
    if (!canary_on()) return hipMalloc(p, bytes);
    char* base = nullptr;
    hipError_t e = hipMalloc((void**)&base, bytes + 2 * KCANARY);
    if (e != hipSuccess) return e;
    const CanaryRec r = { base, bytes };
    if ((e = canary_fill(r)) != hipSuccess) { hipFree(base); return e; }
    std::lock_guard<std::mutex> g(canary_mu);
    canary_map[base + KCANARY] = r;
    *p = base + KCANARY;
    return hipSuccess;
}
// frees a device array; false when its canaries were found damaged (always true without KLARA_DEBUG_CANARY)
static bool dfree(void* p)
{
    if (!p) return true;
    if (!canary_on()) { hipFree(p); return true; }
    CanaryRec r;
    {
        std::lock_guard<std::mutex> g(canary_mu);
        auto it = canary_map.find(p);
        if (it == canary_map.end()) { hipFree(p); return true; }
        r = it->second;
        canary_map.erase(it);
    }
    hipDeviceSynchronize();
    const bool ok = canary_intact(r);
    hipFree(r.base);
    return ok;
}
template <class T>
static hipError_t dalloc(T** p, size_t n) { return dalloc_bytes((void**)p, n * sizeof(T)); }

static klara_status upload(double** dst, const double* src, size_t n)
{
    HIPCHK(dalloc(dst, n));
    HIPCHK(hipMemcpy(*dst, src, n * sizeof(double), hipMemcpyHostToDevice));
    return KLARA_OK;
}

static bool free_all(klara_handle* h)
{
    bool ok = true;
    ok &= dfree(h->X); ok &= dfree(h->GR); ok &= dfree(h->LT); ok &= dfree(h->tune_step); ok &= dfree(h->tune_acc);
    ok &= dfree(h->tune_prop); ok &= dfree(h->tune_tot); ok &= dfree(h->da_epsbar); ok &= dfree(h->da_hbar); ok &= dfree(h->pooled_acc); ok &= dfree(h->accept);
    ok &= dfree(h->naccept); ok &= dfree(h->sum); ok &= dfree(h->sumsq); ok &= dfree(h->held); ok &= dfree(h->hist); ok &= dfree(h->acov_S); ok &= dfree(h->acov_head); ok &= dfree(h->acov_tail); ok &= dfree(h->acov_total); ok &= dfree(h->acov_near); ok &= dfree(h->hist_lt); ok &= dfree(h->hist_g); ok &= dfree(h->hist_ll); ok &= dfree(h->hist_lp); if (!h->flag_host) ok &= dfree(h->err);
    ok &= dfree(h->vecparam); ok &= dfree(h->gw); ok &= dfree(h->gmu); ok &= dfree(h->lX); ok &= dfree(h->ly); ok &= dfree(h->hY); ok &= dfree(h->hxc);
    ok &= dfree(h->Pfrag); ok &= dfree(h->pooled_out); ok &= dfree(h->pool_partial); ok &= dfree(h->d_params); ok &= dfree(h->cdata);
    ok &= dfree(h->bm_prev); ok &= dfree(h->bm_mean); ok &= dfree(h->bm_m2); ok &= dfree(h->auto_cells); ok &= dfree(h->auto_ctr); ok &= dfree(h->clock_probe);
    if (h->auto_mirror) hipHostFree(h->auto_mirror);
    if (h->flag_host) hipHostFree(h->flag_host);
    klara_jit_destroy(h->jit);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    for (int j = 0; j < 3; ++j) { if (h->side[j]) hipStreamDestroy(h->side[j]); if (h->join_ev[j]) hipEventDestroy(h->join_ev[j]); }
    if (h->fork_ev) hipEventDestroy(h->fork_ev);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    return ok;
}

static KParams make_params(klara_handle* h);

// the k_transitions MODE values a job can launch (see launch_steps): nothing counts/tunes and nothing is monitored -> 3 and
// its one-transition-per-launch form 7; nothing counts/tunes -> 1; general -> 0
static int kernel_modes(const klara_desc& d, int (&modes)[2])
{
    const bool plain = !cnt_predicate(d) && d.tuner_mode == KLARA_TUNE_PER_CHAIN && d.tuner != KLARA_TUNER_DUAL_AVERAGING;
    if (plain && d.monitor == 0) { modes[0] = 3; modes[1] = 7; return 2; }
    modes[0] = plain ? 1 : 0;
    return 1;
}

// The logistic-regression target beyond D = 8 (the row-split kernels hold the whole parameter vector of a chain in every lane's
// registers and the data rows in LDS): the same closures — doc/examples/swiss/MALA/analytical.jl:11-18, operation for operation what
// LogisticTarget::eval and the oracle's ko_logistic_eval compute with all rows on one lane — as source text for the run-time compiled
// path (klara_custom.h): one chain per lane, E = pow2ceil(D) <= 256 elements, the data block [lambda, D log(2 pi lambda), X, y] read
// from global memory, any number of rows.
static const char* const KLARA_LOGIT_WIDE_SRC = R"SRC(
KLARA_USER_FN double klara_user_logtarget(const double* p, int D, const double* data, long long ndata)
{
    const long long n = (ndata - 2) / (KLARA_D + 1);
    const double lambda = data[0], lpconst = data[1];
    const double* X = data + 2; const double* y = X + n * KLARA_D;
    double dotxy = 0.0, slog = 0.0;
    for (long long r = 0; r < n; ++r) {
        double xp = 0.0;
        for (int e = 0; e < KLARA_D; ++e) xp = kd_fma(X[r * KLARA_D + e], p[e], xp);
        double sp, lg;
        kd_softplus_logistic_rows(xp, &sp, &lg);
        dotxy = dotxy + xp * y[r];
        slog = slog + sp;
    }
    double dotpp = 0.0;
    for (int e = 0; e < KLARA_D; ++e) dotpp = dotpp + p[e] * p[e];
    const double ll = dotxy - slog;
    const double lp = -0.5 * (dotpp / lambda + lpconst);
    return ll + lp;
}
KLARA_USER_FN void klara_user_gradlogtarget(const double* p, int D, const double* data, long long ndata, double* g)
{
    const long long n = (ndata - 2) / (KLARA_D + 1);
    const double lambda = data[0];
    const double* X = data + 2; const double* y = X + n * KLARA_D;
    for (int e = 0; e < KLARA_D; ++e) g[e] = 0.0;
    for (long long r = 0; r < n; ++r) {
        double xp = 0.0;
        for (int e = 0; e < KLARA_D; ++e) xp = kd_fma(X[r * KLARA_D + e], p[e], xp);
        double sp, lg;
        kd_softplus_logistic_rows(xp, &sp, &lg);
        const double res = y[r] - lg;
        for (int e = 0; e < KLARA_D; ++e) g[e] = kd_fma(X[r * KLARA_D + e], res, g[e]);
    }
    for (int e = 0; e < KLARA_D; ++e) g[e] = g[e] - p[e] / lambda;
}
)SRC";

// The dense Gaussian beyond D = 128 (the matrix-core layouts end there): the same closure form — g = -(P d) as a k-ascending fma chain
// per row, lt = c + 1/2 sum_i d_i g_i, d = x - mu (the oracle's ko_dense_grad / ko_dense_lt_from_grad with all elements on one lane);
// data block [c, P (D x D row-major), mu (D)].
static const char* const KLARA_DENSE_WIDE_SRC = R"SRC(
KLARA_USER_FN void klara_user_gradlogtarget(const double* x, int D, const double* data, long long ndata, double* g)
{
    const double* P = data + 1; const double* mu = P + (long long)KLARA_D * KLARA_D;
    _Pragma("nounroll")
    for (int i = 0; i < KLARA_D; ++i) {
        double acc = 0.0;
        _Pragma("unroll 4")
        for (int k = 0; k < KLARA_D; ++k) acc = kd_fma(P[(long long)i * KLARA_D + k], x[k] - mu[k], acc);
        g[i] = -acc;
    }
}
KLARA_USER_FN double klara_user_logtarget(const double* x, int D, const double* data, long long ndata)
{
    const double* P = data + 1; const double* mu = P + (long long)KLARA_D * KLARA_D;
    double s = 0.0;
    _Pragma("nounroll")
    for (int i = 0; i < KLARA_D; ++i) {
        double acc = 0.0;
        _Pragma("unroll 4")
        for (int k = 0; k < KLARA_D; ++k) acc = kd_fma(P[(long long)i * KLARA_D + k], x[k] - mu[k], acc);
        s = s + (x[i] - mu[i]) * (-acc);
    }
    return data[0] + 0.5 * s;
}
)SRC";

static klara_status create_impl(const klara_desc* desc, klara_handle** out, int custom_lanes = 0);

extern "C" klara_status klara_create(const klara_desc* desc, klara_handle** out)
{
    if (!out) return KLARA_ERR_INVALID_ARG;
    *out = nullptr;
    klara_status st = validate(desc);
    if (st != KLARA_OK) return st;
    // the logistic regression beyond 16 parameters — or beyond 8 when its rows, padded to 16 columns, do not fit the LDS — runs as a closure
    if (desc->target == KLARA_TARGET_LOGISTIC && !logit_mfma_eligible(*desc) &&
        (desc->ndims > 16 || (desc->ndims > 8 && (size_t)desc->logit_ndata * 17 > KLARA_LOGIT_MAX_LDS_DOUBLES))) {
        if (desc->ndims > KLARA_CUSTOM_MAXD) return KLARA_ERR_UNSUPPORTED;
        if (desc->monitor & KLARA_MON_HIST_LLLP) return KLARA_ERR_UNSUPPORTED;
        const size_t n = (size_t)desc->logit_ndata, D = (size_t)desc->ndims;
        std::vector<double> blk(2 + n * (D + 1));
        blk[0] = desc->logit_lambda;
        blk[1] = (double)desc->ndims * kd_log(2.0 * 3.141592653589793 * desc->logit_lambda);
        memcpy(blk.data() + 2, desc->logit_X, n * D * sizeof(double));
        memcpy(blk.data() + 2 + n * D, desc->logit_y, n * sizeof(double));
        klara_desc dd = *desc;
        dd.target = KLARA_TARGET_CUSTOM; dd.custom_src = KLARA_LOGIT_WIDE_SRC; dd.custom_data = blk.data(); dd.custom_ndata = (int64_t)blk.size();
        dd.logit_X = nullptr; dd.logit_y = nullptr; dd.logit_ndata = 0;
        return create_impl(&dd, out, 1);
    }
    if (desc->target == KLARA_TARGET_CUSTOM && pair_as_whole(desc->custom_src, desc->sampler, desc->ndims)) {
        const std::string src2 = pair_as_whole_source(desc->custom_src);
        klara_desc dd = *desc;
        dd.custom_src = src2.c_str();
        return create_impl(&dd, out);
    }
    if (desc->target == KLARA_TARGET_GAUSS_DENSE && desc->ndims > 128 && !dense_streamed(*desc) && !dense_split(*desc)) {
        if (desc->ndims > KLARA_CUSTOM_MAXD) return KLARA_ERR_UNSUPPORTED;
        if (desc->monitor & KLARA_MON_HIST_LLLP) return KLARA_ERR_UNSUPPORTED;
        const size_t D = (size_t)desc->ndims;
        std::vector<double> blk(1 + D * D + D, 0.0);
        blk[0] = desc->gauss_const;
        memcpy(blk.data() + 1, desc->gauss_prec, D * D * sizeof(double));
        if (desc->gauss_mu) memcpy(blk.data() + 1 + D * D, desc->gauss_mu, D * sizeof(double));
        klara_desc dd = *desc;
        dd.target = KLARA_TARGET_CUSTOM; dd.custom_src = KLARA_DENSE_WIDE_SRC; dd.custom_data = blk.data(); dd.custom_ndata = (int64_t)blk.size();
        dd.gauss_prec = nullptr; dd.gauss_mu = nullptr;
        return create_impl(&dd, out, 1);
    }
    return create_impl(desc, out);
}

static klara_status create_impl(const klara_desc* desc, klara_handle** out, int custom_lanes)
{
    klara_status st = validate(desc);
    if (st != KLARA_OK) return st;
    int kind, G, E;
    int custom_wpb = 4;
    st = select_layout(*desc, &kind, &G, &E, custom_lanes, &custom_wpb);
    if (st != KLARA_OK) return st;
    if (desc->tuner_mode == KLARA_TUNE_POOLED && desc->sampler == KLARA_SAMPLER_SLICE) return KLARA_ERR_UNSUPPORTED;
    // the logistic kernels keep the data rows (padded to E columns, + the responses) in LDS next to the 8 KB of math tables: up to
    // 144 KB of the CU's 160 (swiss: 200 x 5 doubles = 8 KB); beyond the 56 KB a launch gets by default the launchers raise the
    // kernel's limit, and fewer workgroups share a CU
    if (desc->target == KLARA_TARGET_LOGISTIC && kind != 5 && (size_t)desc->logit_ndata * (size_t)(E + 1) > KLARA_LOGIT_MAX_LDS_DOUBLES)
        return KLARA_ERR_UNSUPPORTED;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || desc->device < 0 || desc->device >= ndev)
        return KLARA_ERR_HIP;
    HIPCHK(hipSetDevice(desc->device));

    klara_handle* h = new (std::nothrow) klara_handle();
    if (!h) return KLARA_ERR_NOMEM;
    h->d = *desc; h->kind = kind; h->G = G; h->E = E;
    h->custom_rows = (desc->target == KLARA_TARGET_CUSTOM && custom_lik_prior(desc->custom_src)) ? 3 : 2;
    h->custom_wpb = custom_wpb;
    if (desc->target == KLARA_TARGET_LOGISTIC && kind != 5) {
        // D <= 8 parameters cannot fill a wavefront's lanes usefully, the ndata-row likelihood can: RS lanes share a chain
        // and each takes every RS-th row (fixed by ndata alone, so results do not depend on how chains are sharded)
        // 4 lanes from 64 rows on (round 4; rounds 1-3: 8 from 128 rows).  With the rows of an evaluation going through their stages in batches
        // a wavefront no longer needs partners to cover its LDS round trips, so the kernels run at 2 wavefronts per SIMD with the registers
        // of a 4-row batch, and 16 chains per wavefront share the sampler's per-wavefront work instead of 8: swiss (200 rows), 32,768 chains,
        // same box, running sums on: 2.10e9 transitions/s at 4 lanes against 1.88e9 at 8 (gpurun_out/r4_gpu3/ab_logit.txt; profiles/README.md)
        int rs = desc->logit_ndata >= 64 ? 4 : 1;
        if (const char* s = getenv("KLARA_LOGIT_ROWSPLIT")) { const int v = atoi(s); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) rs = v; }
        h->RS = rs;
        if (rs > 1) h->kind = 2;
    }
    const size_t N = (size_t)desc->nchains, D = (size_t)desc->ndims;
    const bool pooled = desc->tuner_mode == KLARA_TUNE_POOLED;
    const size_t NT = pooled ? 1 : N;

#define CK(call) do { klara_status s__ = (call); if (s__ != KLARA_OK) { free_all(h); delete h; return s__; } } while (0)
#define CKH(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { free_all(h); delete h; \
        return e__ == hipErrorOutOfMemory ? KLARA_ERR_NOMEM : KLARA_ERR_HIP; } } while (0)

    if (desc->stream) { h->stream = (hipStream_t)desc->stream; h->own_stream = false; }
    else { CKH(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    CKH(hipEventCreate(&h->ev0)); CKH(hipEventCreate(&h->ev1));
    if (h->kind == 3) {
        h->q4_ok = q4_eligible(*desc);
        h->np4 = (desc->ndims + 7) / 8;
        const long long cpw = h->q4_ok ? 16 : 64 / G, groups = (desc->nchains + cpw - 1) / cpw;
        int np = groups >= 4096 ? 2 : 1;                       // >= one full round of wavefronts (4 per SIMD) per partition
        if (desc->nstreams >= 1 && desc->nstreams <= 4) np = desc->nstreams;
        if (const char* s = getenv("KLARA_STREAMS")) { const int v = atoi(s); if (v >= 1 && v <= 4) np = v; }
        if (desc->tuner_mode == KLARA_TUNE_POOLED) np = 1;     // the pooled tuner update sits between launches, on one stream
        if (desc->acov_maxlag > 0) np = 1;                     // ... and so does the consumer of the launch's saved samples
        if (np > groups) np = (int)groups;
        h->nparts = np;
        if (np > 1) CKH(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
        for (int j = 0; j + 1 < np; ++j) {
            CKH(hipStreamCreateWithFlags(&h->side[j], hipStreamNonBlocking));
            CKH(hipEventCreateWithFlags(&h->join_ev[j], hipEventDisableTiming));
        }
        CKH(dalloc(&h->clock_probe, 4)); CKH(hipMemset(h->clock_probe, 0, 4 * sizeof(unsigned long long)));
        if (h->q4_ok && (desc->monitor & KLARA_MON_SUMMARIES)) {
            CKH(dalloc(&h->auto_cells, 8)); CKH(dalloc(&h->auto_ctr, 4));
            if (hipHostMalloc((void**)&h->auto_mirror, 16 * sizeof(int), hipHostMallocMapped) == hipSuccess) {
                for (int i = 0; i < 16; ++i) h->auto_mirror[i] = (i & 3) == 0 ? 1 : ((i & 3) == 2 ? -1 : 0);
                if (hipHostGetDevicePointer((void**)&h->auto_mirror_dev, h->auto_mirror, 0) != hipSuccess) { hipHostFree(h->auto_mirror); h->auto_mirror = nullptr; h->auto_mirror_dev = nullptr; }
            } else { h->auto_mirror = nullptr; (void)hipGetLastError(); }
            if (const char* s = getenv("KLARA_AUTO_THRESHOLD")) { const double v = atof(s); if (v >= 0.0 && v <= 1.0) h->auto_threshold = v; }
        }
    }

    CKH(dalloc(&h->X, N * D)); CKH(dalloc(&h->GR, N * D)); CKH(dalloc(&h->LT, N));
    CKH(dalloc(&h->tune_step, NT)); CKH(dalloc(&h->tune_acc, NT)); CKH(dalloc(&h->tune_prop, NT));
    CKH(dalloc(&h->tune_tot, NT));
    if (desc->tuner == KLARA_TUNER_DUAL_AVERAGING) { CKH(dalloc(&h->da_epsbar, NT)); CKH(dalloc(&h->da_hbar, NT)); } CKH(dalloc(&h->pooled_acc, 1)); CKH(dalloc(&h->naccept, N));
    // error flag: a mapped word of host memory the kernels store to directly — klara_synchronize then needs no copy command behind the
    // kernels (a 20-transition run of the headline job is ~350 us: a 4-byte device-to-host copy and its completion signal are ~2 % of that)
    if (hipHostMalloc((void**)&h->flag_host, sizeof(int), hipHostMallocMapped) == hipSuccess
        && hipHostGetDevicePointer((void**)&h->err, h->flag_host, 0) == hipSuccess) *h->flag_host = 0;
    else { if (h->flag_host) hipHostFree(h->flag_host); h->flag_host = nullptr; h->err = nullptr; (void)hipGetLastError(); CKH(dalloc(&h->err, 1)); CKH(hipMemset(h->err, 0, sizeof(int))); }
    CKH(dalloc(&h->pooled_out, 2 * D + 2)); CKH(dalloc(&h->pool_partial, (size_t)1024 * (2 * D + 1)));
    if (desc->monitor & KLARA_MON_SUMMARIES) { CKH(dalloc(&h->sum, N * D)); CKH(dalloc(&h->sumsq, N * D)); CKH(dalloc(&h->held, N)); }
    if (desc->bm_batchlen > 0) { CKH(dalloc(&h->bm_prev, N * D)); CKH(dalloc(&h->bm_mean, N * D)); CKH(dalloc(&h->bm_m2, N * D)); }
    if (desc->monitor & KLARA_MON_ACCEPT) {
        h->accept_cap = desc->nsteps;
        CKH(dalloc(&h->accept, (size_t)desc->nsteps * N));
    }
    if ((desc->monitor & KLARA_MON_HIST_LLLP) && (desc->target != KLARA_TARGET_CUSTOM || !strstr(desc->custom_src, "KLARA_USER_LIKELIHOOD_PRIOR"))) {
        free_all(h); delete h; return KLARA_ERR_INVALID_ARG;           // only a likelihood + prior user target has the two parts
    }
    const bool acov = desc->acov_maxlag > 0;
    if ((desc->monitor & (KLARA_MON_HISTORY | KLARA_MON_HIST_LT | KLARA_MON_HIST_GRAD | KLARA_MON_HIST_LLLP)) || acov) {
        // npoststeps = length((burnin+1):thinning:nsteps)  (BasicMCRange.jl:26)
        h->hist_cols = (desc->nsteps - desc->burnin - 1) / desc->thinning + 1;
        long long ringc = desc->hist_ring_cols;
        if (acov && !(desc->monitor & KLARA_MON_HISTORY) && ringc == 0) ringc = 32;     // the estimator's own value ring
        if (acov) { h->d.monitor |= KLARA_MON_HISTORY; h->d.hist_ring_cols = ringc; }   // (the kernels save values; ring_cols(h->d) == ringc)
        if (ringc > 0 && ringc < h->hist_cols) { h->hist_cols = ringc; h->ring = true; }
        if ((desc->monitor & KLARA_MON_HISTORY) || acov) CKH(dalloc(&h->hist, (size_t)h->hist_cols * N * D));
        if (acov) {
            h->acov_W = desc->acov_maxlag + 1;
            const size_t ws = (size_t)h->acov_W * N * D;
            CKH(dalloc(&h->acov_S, ws)); CKH(dalloc(&h->acov_head, ws)); CKH(dalloc(&h->acov_tail, ws)); CKH(dalloc(&h->acov_total, N * D));
            if (h->acov_W > 32) CKH(dalloc(&h->acov_near, (size_t)32 * N * D));      // (scratch of the far-tail update, launch_acov_update)
        }
        if (desc->monitor & KLARA_MON_HIST_LT) CKH(dalloc(&h->hist_lt, (size_t)h->hist_cols * N));
        if (desc->monitor & KLARA_MON_HIST_LLLP) { CKH(dalloc(&h->hist_ll, (size_t)h->hist_cols * N)); CKH(dalloc(&h->hist_lp, (size_t)h->hist_cols * N)); }
        if (desc->monitor & KLARA_MON_HIST_GRAD) {
            if (desc->sampler != KLARA_SAMPLER_MALA && desc->sampler != KLARA_SAMPLER_HMC) {
                free_all(h); delete h; return KLARA_ERR_INVALID_ARG;   // no gradient is carried by MH / slice
            }
            CKH(dalloc(&h->hist_g, (size_t)h->hist_cols * N * D));
        }
    }
    if (desc->sampler == KLARA_SAMPLER_MH) CK(upload(&h->vecparam, desc->mh_sigma, D));
    if (desc->sampler == KLARA_SAMPLER_SLICE) CK(upload(&h->vecparam, desc->slice_widths, D));
    if (desc->target == KLARA_TARGET_GAUSS_DIAG) {
        if (desc->gauss_w) CK(upload(&h->gw, desc->gauss_w, D));
        if (desc->gauss_mu) CK(upload(&h->gmu, desc->gauss_mu, D));
    } else if (desc->target == KLARA_TARGET_HIER_NORMAL) {
        CK(upload(&h->hY, desc->hier_Y, (size_t)desc->hier_nunits * (size_t)desc->hier_ntimes));
        CK(upload(&h->hxc, desc->hier_xc, (size_t)desc->hier_ntimes));
    } else if (desc->target == KLARA_TARGET_CUSTOM) {
        if (desc->custom_ndata > 0) CK(upload(&h->cdata, desc->custom_data, (size_t)desc->custom_ndata));
        if (pair_form(*desc)) {
            // k_diagt instantiations for this job: fused launches always; one transition per launch where that kernel exists
            const bool plain_ = !cnt_predicate(h->d) && desc->tuner_mode == KLARA_TUNE_PER_CHAIN && desc->tuner != KLARA_TUNER_DUAL_AVERAGING;
            const bool mon_ = (h->d.monitor & ~(uint32_t)KLARA_MON_ACCEPT) != 0, da_ = desc->tuner == KLARA_TUNER_DUAL_AVERAGING;
            const bool tune_ = !plain_ || da_;
            if (desc->monitor & KLARA_MON_HIST_LLLP) { free_all(h); delete h; return KLARA_ERR_INVALID_ARG; }
            const int modes[2] = { 0, 1 };
            h->jit_pair = true;
            CK(klara_jit_create_pair(desc->custom_src, desc->sampler, desc->ndims, E / 2, G, mon_, tune_, da_, modes, (!mon_ && !tune_ && desc->sampler != KLARA_SAMPLER_SLICE) ? 2 : 1, true, &h->jit));
        } else {
        int modes[2];
        const int nmodes = kernel_modes(h->d, modes);           // (h->d: the monitor word with what the library turned on itself)
        CK(klara_jit_create(desc->custom_src, desc->sampler, desc->ndims, E, G, modes, nmodes, true, &h->jit));
        }
    } else if (desc->target == KLARA_TARGET_LOGISTIC && kind == 5) {
        // the A fragments of both MFMA passes in the order of consumption (klara_logit_mfma.h logitm_eval), zero beyond the n rows / D columns:
        // block b = RBT tiles of 16 rows; pass 1, step kk RBT + tt: lane l holds X[16 (b RBT + tt) + (l & 15)][4 kk + (l >> 4)];
        // pass 2, step (4 tt + j) MT + t: lane l holds X[16 (b RBT + tt) + 4 j + (l >> 4)][16 t + (l & 15)].  The responses are zero-padded to the blocks' rows.
        const int NE = E, MT = NE / 4, RBT = klara_logit_mfma_rbt(), S1 = RBT * NE;
        const size_t n = (size_t)desc->logit_ndata;
        const int NT = (int)((n + 15) / 16), nb = (NT + RBT - 1) / RBT;
        std::vector<double> frag((size_t)nb * 2 * S1 * 64, 0.0), ypad((size_t)nb * RBT * 16, 0.0);
        for (int b = 0; b < nb; ++b) {
            double* const f1 = frag.data() + (size_t)b * 2 * S1 * 64;
            double* const f2 = f1 + (size_t)S1 * 64;
            for (int sidx = 0; sidx < S1; ++sidx) {
                const int kk = sidx / RBT, tt = sidx % RBT;
                const int tt2 = sidx / (4 * MT), j = (sidx / MT) & 3, t = sidx % MT;
                for (int l = 0; l < 64; ++l) {
                    const size_t r1 = 16 * (size_t)(b * RBT + tt) + (l & 15), c1 = 4 * (size_t)kk + (l >> 4);
                    if (r1 < n && c1 < D) f1[(size_t)sidx * 64 + l] = desc->logit_X[r1 * D + c1];
                    const size_t r2 = 16 * (size_t)(b * RBT + tt2) + 4 * (size_t)j + (l >> 4), c2 = 16 * (size_t)t + (l & 15);
                    if (r2 < n && c2 < D) f2[(size_t)sidx * 64 + l] = desc->logit_X[r2 * D + c2];
                }
            }
        }
        for (size_t r = 0; r < n; ++r) ypad[r] = desc->logit_y[r];
        h->logit_nblocks = nb;
        CK(upload(&h->Pfrag, frag.data(), frag.size()));
        CK(upload(&h->ly, ypad.data(), ypad.size()));
        h->lpconst = (double)desc->ndims * kd_log(2.0 * 3.141592653589793 * desc->logit_lambda);
    } else if (desc->target == KLARA_TARGET_LOGISTIC) {
        CK(upload(&h->lX, desc->logit_X, (size_t)desc->logit_ndata * D));
        CK(upload(&h->ly, desc->logit_y, (size_t)desc->logit_ndata));
        // length(p)*log(2*pi*v[1])  (doc/examples/swiss/MALA/analytical.jl:16)
        h->lpconst = (double)desc->ndims * kd_log(2.0 * 3.141592653589793 * desc->logit_lambda);
    } else {
        // fragment-ordered, zero-padded P for the MFMA A operand (klara_dense.h)
        // (layout kind 6: ceil(D / 4) k-steps of MT = ceil(D / 16) tiles, k-major, and KLARA_SPLIT_PAD = 8 k-steps of zeros behind them — the ring's last
        // prefetch, klara_dense_split.h)
        const int NE = kind == 6 ? (D + 3) / 4 : E, MT = kind == 6 ? (D + 15) / 16 : (NE + 3) / 4;
        std::vector<double> frag((size_t)MT * (NE + (kind == 6 ? 8 : 0)) * 64, 0.0);
        // (NE % 4 == 1: the last tile is the 4-row tail for v_mfma_f64_4x4x4_4b, A_b[i][k] on lane 16k + 4b + i)
        const bool tail = kind != 6 && (NE % 4) == 1;
        // tile-major (t, kk) for the LDS-resident layouts; k-major (kk, t) — the order of consumption — for the streamed ones (NE > 32)
        const bool kmajor = NE > 32 || kind == 6;
        for (int t = 0; t < MT; ++t)
            for (int kk = 0; kk < NE; ++kk)
                for (int l = 0; l < 64; ++l) {
                    const size_t row = 16 * (size_t)t + ((tail && t == MT - 1) ? (l & 3) : (l & 15)), col = 4 * (size_t)kk + (l >> 4);
                    const size_t f = kmajor ? (size_t)kk * MT + t : (size_t)t * NE + kk;
                    if (row < D && col < D) frag[f * 64 + l] = desc->gauss_prec[row * D + col];
                }
        h->dense_mu = desc->gauss_mu != nullptr;
        if (h->dense_mu) {                                       // the mean, [4 e + q] = mu[4 e + q], zero beyond D
            const size_t at = frag.size();
            frag.resize(at + 4 * (size_t)(kind == 6 ? 4 * MT : NE) + (kind == 6 ? 32 : 0), 0.0);         // (kind 6: 4 MT rows + KLARA_SPLIT_PAD the pass's last prefetch touches)
            for (int i = 0; i < D; ++i) frag[at + i] = desc->gauss_mu[i];
        }
        CK(upload(&h->Pfrag, frag.data(), frag.size()));
    }
    // the descriptor's host pointers are not retained
    h->d.mh_sigma = nullptr; h->d.slice_widths = nullptr; h->d.gauss_w = nullptr; h->d.gauss_mu = nullptr;
    h->d.gauss_prec = nullptr; h->d.logit_X = nullptr; h->d.logit_y = nullptr; h->d.hier_Y = nullptr; h->d.hier_xc = nullptr; h->d.stream = nullptr;
    h->d.custom_src = nullptr; h->d.custom_data = nullptr;
    {   // static kernel parameters live in device memory (read with scalar loads at the point of use)
        const KParams hp = make_params(h);
        CKH(dalloc(&h->d_params, 1));
        CKH(hipMemcpy(h->d_params, &hp, sizeof(KParams), hipMemcpyHostToDevice));
    }
#undef CK
#undef CKH
    *out = h;
    return KLARA_OK;
}

extern "C" klara_status klara_destroy(klara_handle* h)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    hipSetDevice(h->d.device);
    hipStreamSynchronize(h->stream);
    for (int j = 0; j < 3; ++j) if (h->side[j]) hipStreamSynchronize(h->side[j]);
    const bool intact = free_all(h);
    delete h;
    return intact ? KLARA_OK : KLARA_ERR_STATE;        // (KLARA_DEBUG_CANARY=1: a kernel of this job wrote outside one of its arrays)
}

static KParams make_params(klara_handle* h)
{
    KParams p;
    memset(&p, 0, sizeof(p));
    const klara_desc& d = h->d;
    p.X = (decltype(p.X))h->X; p.GR = (decltype(p.GR))h->GR; p.LT = (decltype(p.LT))h->LT;
    p.tune_step = (decltype(p.tune_step))h->tune_step; p.tune_accepted = (decltype(p.tune_accepted))h->tune_acc; p.tune_proposed = (decltype(p.tune_proposed))h->tune_prop;
    p.tune_totproposed = (decltype(p.tune_totproposed))h->tune_tot; p.pooled_accepted = (decltype(p.pooled_accepted))h->pooled_acc;
    p.accept = (decltype(p.accept))h->accept; p.naccept = (decltype(p.naccept))h->naccept; p.sum = (decltype(p.sum))h->sum; p.sumsq = (decltype(p.sumsq))h->sumsq; p.held = (decltype(p.held))h->held;
    p.hist = (decltype(p.hist))h->hist; p.hist_cols = h->hist_cols; p.error_flag = (decltype(p.error_flag))h->err;
    p.hist_lt = (decltype(p.hist_lt))h->hist_lt; p.hist_g = (decltype(p.hist_g))h->hist_g;
    p.hist_ll = (decltype(p.hist_ll))h->hist_ll; p.hist_lp = (decltype(p.hist_lp))h->hist_lp;
    p.nchains = d.nchains; p.chain_offset = d.chain_offset; p.D = d.ndims; p.G = h->G; p.rs = h->RS;
    p.pooled = d.tuner_mode == KLARA_TUNE_POOLED;
    p.seed = d.seed + h->epoch * KLARA_EPOCH_KEY_STRIDE;      // (mod 2^64)
    p.vecparam = (decltype(p.vecparam))h->vecparam; p.nleaps = d.nleaps; p.stepout = d.slice_stepout;
    p.tuner = d.tuner; p.cnt = cnt_predicate(d); p.targetrate = d.targetrate;
    p.tuner_score = d.tuner_score; p.score_k = d.score_k; p.period = d.period; p.is_mh = d.sampler == KLARA_SAMPLER_MH;
    p.da_epsbar = (decltype(p.da_epsbar))h->da_epsbar; p.da_hbar = (decltype(p.da_hbar))h->da_hbar; p.da_nadapt = d.da_nadapt; p.da_gamma = d.da_gamma;
    p.da_kappa = d.da_kappa; p.da_t0 = d.da_t0;
    // sampler_state(..., tuner::DualAveragingMCTuner): lambda = nleaps*leapstep, mu = log(10*step) (HMC.jl:124-133,192-213)
    p.da_lambda = (double)d.nleaps * d.leapstep; p.da_mu = kd_log(10.0 * d.leapstep);
    p.step0 = d.sampler == KLARA_SAMPLER_MH ? 1.0 : d.sampler == KLARA_SAMPLER_MALA ? d.driftstep
            : d.sampler == KLARA_SAMPLER_HMC ? d.leapstep : (double)NAN;
    p.sqrt_step0 = std::sqrt(p.step0); p.inv_step0 = 1.0 / p.step0;
    p.burnin = d.burnin; p.thinning = d.thinning; p.nsteps_total = d.nsteps;
    p.gw = (decltype(p.gw))h->gw; p.gmu = (decltype(p.gmu))h->gmu; p.gconst = d.gauss_const;
    p.lX = (decltype(p.lX))h->lX; p.ly = (decltype(p.ly))h->ly; p.ndata = d.logit_ndata; p.lambda = d.logit_lambda; p.lpconst = h->lpconst;
    p.hY = (decltype(p.hY))h->hY; p.hxc = (decltype(p.hxc))h->hxc; p.hR = d.hier_nunits; p.hT = d.hier_ntimes; p.hp0 = d.hier_prior_prec;
    p.ha0 = d.hier_gamma_a; p.hb0 = d.hier_gamma_b;
    p.cdata = (decltype(p.cdata))h->cdata; p.cndata = d.custom_ndata;
    p.clock_probe = (decltype(p.clock_probe))h->clock_probe;
    return p;
}

// one wave per chain group for the init kernels and the MFMA kernels
static dim3 grid_for(const klara_handle* h)
{
    if (h->kind == 6) return dim3((unsigned)((h->d.nchains + 15) / 16));          // one workgroup (G wavefronts) per tile of 16 chains
    const long long cpw = (h->kind == 1 || h->kind == 5) ? 16 : 64 / (h->G * h->RS);
    const long long waves = (h->d.nchains + cpw - 1) / cpw;
    const long long wpb = h->kind == 5 ? 4 : h->kind == 1 ? (h->E > 32 ? 4 : 8) : h->custom_wpb;      // (streamed layouts: one wavefront per SIMD, workgroups of 4)
    return dim3((unsigned)((waves + wpb - 1) / wpb));
}

// group-layout transition kernels are persistent over `groups_per_wave` chain groups (prefetch of the
// next group's state overlaps the current group's compute); keep >= ~8 waves per SIMD of work for balance
static dim3 grid_for_transitions(const klara_handle* h)
{
    const long long cpw = 64 / (h->G * h->RS);
    const long long groups = (h->d.nchains + cpw - 1) / cpw;
    long long gpw = 4;
    while (gpw > 1 && groups / gpw < 8192) gpw >>= 1;
    if (const char* s = getenv("KLARA_GROUPS_PER_WAVE")) { const long long v = atoll(s); if (v >= 1 && v <= 1024) gpw = v; }
    const long long waves = (groups + gpw - 1) / gpw;
    return dim3((unsigned)((waves + h->custom_wpb - 1) / h->custom_wpb));
}

static size_t lds_for(const klara_handle* h)
{
    if (h->kind != 1 && h->kind != 5 && h->d.target == KLARA_TARGET_LOGISTIC)        // data rows + responses + the kernel's copy of kd_log12's table (LogisticTarget::lds_bytes)
        return sizeof(double) * (((size_t)h->d.logit_ndata * (size_t)(h->E + 1) + 1) / 2 * 2 + 256);
    if (h->kind == 0 && h->d.target == KLARA_TARGET_CUSTOM && h->G > 1)              // staged closure: the rows of a workgroup's chains
        return custom_stage_bytes(h->d.ndims, h->G, h->custom_rows, h->custom_wpb);
    return 0;
}

// ---- init kernels (group layout) instantiated here
template <int TARGET>
static hipError_t launch_init_t(const KParams& p, int E, int G, int needgrad, dim3 grid, size_t lds, hipStream_t st)
{
    const dim3 blk(256);
#define KLARA_INIT_LAUNCH(E_, G_)                                                                                                  \
    do {                                                                                                                          \
        if (lds > KLARA_LDS_DEFAULT_DYNAMIC) {                                                                                    \
            hipError_t e_ = hipFuncSetAttribute((const void*)k_init<TARGET, E_, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e_ != hipSuccess) return e_;                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((k_init<TARGET, E_, G_>), grid, blk, lds, st, p, needgrad);                                             \
    } while (0)
    if (E == 2 && G == 64 && TARGET == KLARA_TARGET_GAUSS_DIAG) KLARA_INIT_LAUNCH(2, 64);
    else if (E == 2) KLARA_INIT_LAUNCH(2, 0);
    else if (E == 4) KLARA_INIT_LAUNCH(4, 0);
    else if (E == 8 && TARGET != KLARA_TARGET_HIER_NORMAL) KLARA_INIT_LAUNCH((TARGET == KLARA_TARGET_HIER_NORMAL ? 4 : 8), 0);
    else if (E == 16 && TARGET == KLARA_TARGET_LOGISTIC) KLARA_INIT_LAUNCH((TARGET == KLARA_TARGET_LOGISTIC ? 16 : 2), 0);
    else return hipErrorInvalidValue;
#undef KLARA_INIT_LAUNCH
    return hipGetLastError();
}

__global__ void k_fill_tune(double* step, long long* acc, long long* prop, long long* tot, long long n,
                            double step0, long long period)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { step[i] = step0; acc[i] = 0; prop[i] = 0; tot[i] = period; }
}

__global__ void k_fill2(double* a, double* b, long long n, double va, double vb)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = va; b[i] = vb; }
}

// pooled tuner update after a launch of `k` transitions: tuners.jl:27-32, AcceptanceRateMCTuner.jl:46
// with the rate pooled over the GPU's chains (KLARA_TUNE_POOLED; SURVEY §7 hard part 5).
__global__ void k_pooled_tune(KParams p, int k)
{
    kd_tables_to_lds();            // rate_score -> kd_exp reads its table from LDS
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (!p.cnt) return;
    long long prop = p.tune_proposed[0] + k;
    long long acc = p.tune_accepted[0] + (long long)(*p.pooled_accepted);
    long long tot = p.tune_totproposed[0];
    double step = p.tune_step[0];
    *p.pooled_accepted = 0ull;
    if (tot <= p.burnin && (prop % p.period) == 0) {
        const double rate = (double)acc / (double)(prop * p.nchains);
        if (p.tuner == KLARA_TUNER_ACCEPT_RATE && !p.is_mh) {
            const double xr = rate - p.targetrate;
            step *= rate_score(p, xr);
        }
        tot += prop; acc = 0; prop = 0;
    }
    p.tune_step[0] = step; p.tune_accepted[0] = acc; p.tune_proposed[0] = prop; p.tune_totproposed[0] = tot;
}

static klara_status init_common(klara_handle* h)
{
    const klara_desc& d = h->d;
    const size_t N = (size_t)d.nchains, D = (size_t)d.ndims;
    const bool pooled = d.tuner_mode == KLARA_TUNE_POOLED;
    const long long NT = pooled ? 1 : (long long)N;
    hipStream_t st = h->stream;
    HIPCHK(hipMemsetAsync(h->err, 0, sizeof(int), st));
    HIPCHK(hipMemsetAsync(h->naccept, 0, N * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(h->pooled_acc, 0, sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(h->GR, 0, N * D * sizeof(double), st));
    if (h->sum) {
        HIPCHK(hipMemsetAsync(h->sum, 0, N * D * sizeof(double), st)); HIPCHK(hipMemsetAsync(h->sumsq, 0, N * D * sizeof(double), st));
        HIPCHK(hipMemsetAsync(h->held, 0, N * sizeof(long long), st));
    }
    if (h->bm_prev) {
        HIPCHK(hipMemsetAsync(h->bm_prev, 0, N * D * sizeof(double), st)); HIPCHK(hipMemsetAsync(h->bm_mean, 0, N * D * sizeof(double), st));
        HIPCHK(hipMemsetAsync(h->bm_m2, 0, N * D * sizeof(double), st));
    }
    h->bm_count = 0;
    if (h->auto_cells) {             // first launch of a job: resident sums (flat cost whatever the acceptance turns out to be)
        const int ones[8] = { 1, 1, 1, 1, 1, 1, 1, 1 };
        HIPCHK(hipMemcpyAsync(h->auto_cells, ones, sizeof(ones), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(h->auto_ctr, 0, 4 * sizeof(unsigned long long), st));
    }
    if (h->acov_S) {
        const size_t ws = (size_t)h->acov_W * N * D * sizeof(double);
        HIPCHK(hipMemsetAsync(h->acov_S, 0, ws, st)); HIPCHK(hipMemsetAsync(h->acov_head, 0, ws, st)); HIPCHK(hipMemsetAsync(h->acov_tail, 0, ws, st));
        HIPCHK(hipMemsetAsync(h->acov_total, 0, N * D * sizeof(double), st));
    }
    h->acov_n = 0;
    // tuner_state: samplers.jl:29-45 — step per sampler, accepted = proposed = 0, totproposed = period
    const double step0 = d.sampler == KLARA_SAMPLER_MH ? 1.0
                       : d.sampler == KLARA_SAMPLER_MALA ? d.driftstep
                       : d.sampler == KLARA_SAMPLER_HMC ? d.leapstep : (double)NAN;
    hipLaunchKernelGGL(k_fill_tune, dim3((unsigned)((NT + 255) / 256)), dim3(256), 0, st, h->tune_step,
                       h->tune_acc, h->tune_prop, h->tune_tot, NT, step0, (long long)d.period);
    HIPCHK(hipGetLastError());
    if (h->da_epsbar) {
        hipLaunchKernelGGL(k_fill2, dim3((unsigned)((NT + 255) / 256)), dim3(256), 0, st, h->da_epsbar, h->da_hbar, NT,
                           d.da_eps0bar, d.da_h0bar);
        HIPCHK(hipGetLastError());
    }
    const int needgrad = d.sampler == KLARA_SAMPLER_MALA || d.sampler == KLARA_SAMPLER_HMC;
    KParams p = make_params(h);
    hipError_t e;
    if (h->kind == 1) e = klara_launch_dense_init(p, h->E, h->Pfrag, h->dense_mu, needgrad, grid_for(h), st);
    else if (h->kind == 5) e = klara_launch_logit_mfma_init(p, h->E, h->Pfrag, h->ly, h->logit_nblocks, needgrad, grid_for(h), st);
    else if (h->kind == 6) e = klara_launch_dense_split_init(p, h->G, h->E, h->Pfrag, h->dense_mu, needgrad, grid_for(h), st);
    else if (h->kind == 3 && h->jit_pair) e = klara_jit_launch_init(h->jit, p, needgrad, grid_for(h), 0, st);
    else if (h->kind == 3)
        e = h->G == 8 ? klara_launch_diagt_init(p, h->E / 2, needgrad, grid_for(h), st)
          : h->G == 16 ? klara_launch_diagt_init_q16(p, h->E / 2, needgrad, grid_for(h), st)
          : h->G == 32 ? klara_launch_diagt_init_q32(p, h->E / 2, needgrad, grid_for(h), st)
                       : klara_launch_diagt_init_q64(p, h->E / 2, needgrad, grid_for(h), st);
    else if (h->kind == 4) e = klara_launch_hiert_init(p, h->E / 2, d.hier_ntimes, needgrad, grid_for(h), st);
    else if (d.target == KLARA_TARGET_CUSTOM) e = klara_jit_launch_init(h->jit, p, needgrad, grid_for(h), lds_for(h), st, 64 * h->custom_wpb);
    else if (d.target == KLARA_TARGET_GAUSS_DIAG)
        e = launch_init_t<KLARA_TARGET_GAUSS_DIAG>(p, h->E, h->G, needgrad, grid_for(h), lds_for(h), st);
    else if (d.target == KLARA_TARGET_HIER_NORMAL)
        e = launch_init_t<KLARA_TARGET_HIER_NORMAL>(p, h->E, h->G, needgrad, grid_for(h), lds_for(h), st);
    else e = launch_init_t<KLARA_TARGET_LOGISTIC>(p, h->E, h->G, needgrad, grid_for(h), lds_for(h), st);
    HIPCHK(e);
    int flag = 0;
    if (h->flag_host) { HIPCHK(hipStreamSynchronize(st)); flag = __atomic_load_n(h->flag_host, __ATOMIC_ACQUIRE); }
    else { HIPCHK(hipMemcpyAsync(&flag, h->err, sizeof(int), hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); }
    // (host mirror of the decision cells, written once nothing is in flight)
    if (h->auto_mirror) for (int i = 0; i < 16; ++i) h->auto_mirror[i] = (i & 3) == 0 ? 1 : ((i & 3) == 2 ? (int)(h->launch_idx - 1) : 0);
    h->steps_done = 0; h->nsaved = 0; h->m_prop = 0; h->m_tot = d.period; h->timed = false;
    if (flag != 0) { h->have_state = false; return (klara_status)flag; }
    h->have_state = true;
    return KLARA_OK;
}

extern "C" klara_status klara_set_state(klara_handle* h, const double* x_host)
{
    if (!h || !x_host) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->d.device));
    const size_t n = (size_t)h->d.nchains * (size_t)h->d.ndims;
    HIPCHK(hipMemcpyAsync(h->X, x_host, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return init_common(h);
}

extern "C" klara_status klara_init_state_normal(klara_handle* h)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->d.device));
    // the stream is layout-free (element i <- slot i>>1), so any group layout draws the same x0:
    // use the handle's own (kind 0) or a group layout that fits the wavefront for the other kinds
    KParams p = make_params(h);
    const int D = h->d.ndims;
    int E = D <= 128 ? 2 : (D <= 256 ? 4 : (D <= 512 ? 8 : 16)), G = pow2ceil((D + E - 1) / E);       // (at most 64 lanes per chain: D <= 1024)
    if ((h->kind == 0 || h->kind == 2) && h->d.target != KLARA_TARGET_CUSTOM) { E = h->E; G = h->G; }
    p.G = G; p.rs = 1;     // the init stream is drawn without the row split (same values, any layout)
    const long long cpw = 64 / G, waves = (h->d.nchains + cpw - 1) / cpw;
    const dim3 grid((unsigned)((waves + 3) / 4)), blk(256);
    if (E == 2) hipLaunchKernelGGL((k_init_normal<2, 0>), grid, blk, 0, h->stream, p);
    else if (E == 4) hipLaunchKernelGGL((k_init_normal<4, 0>), grid, blk, 0, h->stream, p);
    else if (E == 8) hipLaunchKernelGGL((k_init_normal<8, 0>), grid, blk, 0, h->stream, p);
    else hipLaunchKernelGGL((k_init_normal<16, 0>), grid, blk, 0, h->stream, p);
    HIPCHK(hipGetLastError());
    return init_common(h);
}

// reset(job[, x]) rewinds the sampler / tuner state and the counters; the reference's random generator keeps advancing, so the
// next run is an independent replicate (BasicMCJob.jl:187-201).  The counter-based stream restarts its transition index at 0, so
// the job moves on to a fresh Philox key instead: seed + epoch * KLARA_EPOCH_KEY_STRIDE (klara_stream_key reports it).
// klara_set_state alone keeps the key: it is the way to replay a job from chosen values.
extern "C" klara_status klara_reset(klara_handle* h, const double* x_host)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!x_host && !h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    // the job moves to its next key only if the reset succeeds: a failed one — non-finite initial values, a device
    // error — leaves the epoch, the key and the device copy of the parameters as they were
    const auto put_params = [&]() -> klara_status {
        const KParams hp = make_params(h);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->d_params, &hp, sizeof(KParams), hipMemcpyHostToDevice));
        return KLARA_OK;
    };
    h->epoch += 1;
    klara_status st = put_params();
    if (st == KLARA_OK) st = x_host ? klara_set_state(h, x_host) : init_common(h);
    if (st != KLARA_OK) { h->epoch -= 1; (void)put_params(); }
    return st;
}

extern "C" klara_status klara_stream_key(klara_handle* h, uint64_t* key, uint64_t* epoch)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (key) *key = h->d.seed + h->epoch * KLARA_EPOCH_KEY_STRIDE;
    if (epoch) *epoch = h->epoch;
    return KLARA_OK;
}

// layout kind 3: the chains [c0, c1) of partition j of nparts — cut in blocks of 16 chains (64 / G for the wider layouts), i.e. whole
// chain groups of every kernel that may run the job
static void part_range(const klara_handle* h, int nparts, int j, long long* c0, long long* c1)
{
    const long long N = h->d.nchains, blk = 16;
    const long long blocks = (N + blk - 1) / blk, per = (blocks + nparts - 1) / nparts;
    *c0 = j * per * blk; *c1 = (j + 1) * per * blk;
    if (*c0 > N) *c0 = N;
    if (*c1 > N) *c1 = N;
}

static long long saved_upto(const klara_desc& d, long long steps);       // (launch planning, below)
static hipError_t launch_steps(klara_handle* h, const KLaunch& kl, int nparts)
{
    const klara_desc& d = h->d;
    const KParams* p = h->d_params;
    // PLAIN kernels: nothing counts proposals, nothing tunes (VanillaMCTuner, not verbose)
    const bool plain = !cnt_predicate(d) && d.tuner_mode == KLARA_TUNE_PER_CHAIN && d.tuner != KLARA_TUNER_DUAL_AVERAGING;
    int mode = (plain ? 1 : 0) | ((plain && d.monitor == 0) ? 2 : 0);         // 3: no monitors either
    if (mode == 3 && kl.nsteps == 1) mode = 7;                                 // one iterate! per launch
    if (h->kind == 1) return klara_launch_dense(p, kl, d.sampler, d.tuner, plain, h->E, h->Pfrag, h->dense_mu, grid_for(h), h->stream);
    if (h->kind == 5) return klara_launch_logit_mfma(p, kl, d.sampler, d.tuner == KLARA_TUNER_DUAL_AVERAGING, h->E, h->Pfrag, h->ly, h->logit_nblocks, grid_for(h), h->stream);
    if (h->kind == 6) return klara_launch_dense_split(p, kl, d.sampler, d.tuner == KLARA_TUNER_DUAL_AVERAGING, h->G, h->E, d.ndims, h->Pfrag, h->dense_mu, grid_for(h), h->stream);
    if (h->kind == 3) {
        const bool unitw = h->gw == nullptr && h->gmu == nullptr, onestep = kl.nsteps == 1;   // (device copies; the host pointers are dropped at create)
        const bool mon = (d.monitor & ~(uint32_t)KLARA_MON_ACCEPT) != 0;                      // a saved-sample monitor is on
        const bool da = d.tuner == KLARA_TUNER_DUAL_AVERAGING;                                 // (HMC only: validate())
        const bool tune = !plain || da;                                                        // something counts proposals / tunes
        // which kernel family runs the launch (q4_ok jobs: same bits either way)
        const bool sums = (d.monitor & KLARA_MON_SUMMARIES) != 0;
        // slice sampler: nothing counts and no history is kept -> the lanes run out of lockstep (klara_diagt_slice.h); same draws, same bits
        const bool slice_free = slice_free_eligible(d);
        int force = -1;                                      // 0: 4 lanes per chain; 1: 8 lanes; -1: decided on the device
        if (!h->q4_ok) force = 1;
        else if (d.sampler == KLARA_SAMPLER_HMC && onestep) force = 1;   // (one transition per launch: the 8-lane single-transition kernel measured 5 % faster)
        else if (!sums) force = 0;                           // nothing to fold: the 4-lane kernels
        else if (d.sparse_moves == 1) force = 0;
        else if (d.sparse_moves == 2) force = 1;
        if (h->q4_ok && sums) if (const char* sm = getenv("KLARA_SUM_MODE")) { const int v = atoi(sm); if (v == 0 || v == 1) force = v; }
        const bool query = klara_attr_query != nullptr;       // klara_get_kernel_attributes: which kernel, not a launch
        if (query && h->q4_ok) force = h->query_lanes == 8 ? 1 : 0;
        const long long idx = query ? h->launch_idx : h->launch_idx++;
        for (int j = 0; j < nparts; ++j) {
            long long c0, c1;
            part_range(h, nparts, j, &c0, &c1);
            if (c0 >= c1) break;
            hipStream_t st = j == 0 ? h->stream : h->side[j - 1];
            hipError_t e = hipSuccess;
            const auto go = [&](int lanes, const KAuto& ka) -> hipError_t {
                KLaunch kp = kl;
                const long long cpw = 64 / lanes;
                kp.group0 = c0 / cpw; kp.group_end = (c1 + cpw - 1) / cpw;
                const long long nw = kp.group_end - kp.group0;               // one wavefront per group of cpw chains
                const int np = lanes == 4 ? h->np4 : h->E / 2;
                if (h->jit_pair) return klara_jit_launch_pair(h->jit, (onestep && !tune && !mon && d.sampler != KLARA_SAMPLER_SLICE) ? 1 : 0, p, kp, nw, st);
                if (lanes == 4)
                    return d.sampler == KLARA_SAMPLER_MH ? klara_launch_diagt_mh_q4(p, kp, np, onestep && !tune, unitw, mon, tune, da, ka, nw, st)
                         : d.sampler == KLARA_SAMPLER_HMC ? klara_launch_diagt_hmc_q4(p, kp, np, false, unitw, mon, tune, da, ka, nw, st)
                                                          : klara_launch_diagt_mala_q4(p, kp, np, onestep && !tune, unitw, mon, tune, da, ka, nw, st);
#define KLARA_DIAGT_LAUNCH(SUFFIX)                                                                                                          \
                switch (d.sampler) {                                                                                                          \
                case KLARA_SAMPLER_MH: return klara_launch_diagt_mh##SUFFIX(p, kp, np, onestep && !tune, unitw, mon, tune, da, ka, nw, st);        \
                case KLARA_SAMPLER_SLICE:                                                                                                     \
                    if (slice_free) {                                                                                                         \
                        const hipError_t e_ = klara_launch_diagt_slice_free##SUFFIX(p, kp, np, unitw, mon, ka, nw, st);                        \
                        if (e_ != hipSuccess || h->hist_lt == nullptr || query) return e_;                                                    \
                        /* the saved states' log-targets, from their saved values (klara_diagt_slice.h) */                                   \
                        const long long nsaved_ = saved_upto(d, (long long)kl.t0 + kl.nsteps) - saved_upto(d, (long long)kl.t0);              \
                        return klara_launch_diagt_hist_lt##SUFFIX(p, kp, np, unitw, kl.save_col0, (int)nsaved_, nw, st);                       \
                    }                                                                                                                         \
                    return klara_launch_diagt_slice##SUFFIX(p, kp, np, unitw, mon, tune, ka, nw, st);                                          \
                case KLARA_SAMPLER_MALA: return klara_launch_diagt_mala##SUFFIX(p, kp, np, onestep && !tune, unitw, mon, tune, da, ka, nw, st);    \
                default: return klara_launch_diagt_hmc##SUFFIX(p, kp, np, onestep && !tune, unitw, mon, tune, da, ka, nw, st);                     \
                }
                if (lanes == 8) { KLARA_DIAGT_LAUNCH() } else if (lanes == 16) { KLARA_DIAGT_LAUNCH(_q16) } else if (lanes == 32) { KLARA_DIAGT_LAUNCH(_q32) } else { KLARA_DIAGT_LAUNCH(_q64) }
#undef KLARA_DIAGT_LAUNCH
            };
            if (force >= 0 && !(h->q4_ok && sums)) {
                e = go(force == 0 ? 4 : h->G, KLARA_AUTO_NONE);
                if (h->q4_ok) h->n_launch_mode[force == 0 ? 0 : 1] += (j == 0 && !query);      // (klara_get_launch_modes: which family ran)
            } else {
                // running sums on a job both kernel families can run: every launch counts its accepted proposals and leaves the
                // decision for the next one in the partition's cell of the other parity (KAuto)
                KAuto ka;
                ka.cell_in = nullptr;
                ka.cell_out = h->auto_cells + 2 * j + (int)((idx + 1) & 1);
                ka.acc_ctr = h->auto_ctr + j;
                ka.mirror = h->auto_mirror_dev ? h->auto_mirror_dev + 4 * j : nullptr;
                ka.thr_work = (unsigned long long)(h->auto_threshold * 65536.0 * (double)(c1 - c0) * (double)kl.nsteps);
                ka.fanout = nparts == 1 ? 4 : 1;             // a whole-job launch decides for every partition
                ka.launch_idx = (int)idx;
                if (force >= 0) {
                    ka.my_mode = force;
                    e = go(force == 0 ? 4 : 8, ka);
                    h->n_launch_mode[force] += (j == 0 && !query);
                } else if (kl.nsteps < KLARA_AUTO_PAIR_MIN_STEPS) {
                    // a short launch is not worth an idle sibling: the host picks from the last decision it has seen (stale at worst),
                    // and only every 16th such launch counts its accepted proposals and renews the decision (the workgroups of a
                    // one-transition launch all finish together: their counter updates on one address are ~20 % of such a launch)
                    ka.my_mode = h->auto_mirror ? (__atomic_load_n(h->auto_mirror + 4 * j, __ATOMIC_RELAXED) != 0) : 1;
                    e = go(ka.my_mode == 0 ? 4 : 8, (idx & 15) == 0 ? ka : KLARA_AUTO_NONE);
                    h->n_launch_mode[ka.my_mode] += (j == 0 && !query);
                } else if (nparts == 1 && h->pair_enqueued && h->auto_mirror && !query
                           && __atomic_load_n(h->auto_mirror + 4 * j + 2, __ATOMIC_ACQUIRE) == (int)(idx - 1)) {
                    // a synchronous caller (one launch per run, the previous one complete): the decision for this launch is already on the
                    // host — one kernel, unconditional, no idle sibling (~5 us of a 20-transition launch).  Only after a launch has enqueued both
                    // families: a kernel family's first dispatch on a queue sets up its scratch (~120 us, once).
                    ka.my_mode = __atomic_load_n(h->auto_mirror + 4 * j, __ATOMIC_RELAXED) != 0;
                    e = go(ka.my_mode == 0 ? 4 : 8, ka);
                    h->n_launch_mode[2] += (j == 0);
                } else {
                    // both kernels, each subject to the decision the previous launch left (an idle sibling costs ~3 us of a launch that
                    // takes hundreds: 17.9 against 17.7 us per transition for 20-transition launches, nothing measurable on two streams)
                    h->pair_enqueued = true;
                    ka.cell_in = h->auto_cells + 2 * j + (int)(idx & 1);
                    ka.my_mode = 0; e = go(4, ka);
                    if (e == hipSuccess) { ka.my_mode = 1; e = go(8, ka); }
                    h->n_launch_mode[2] += (j == 0 && !query);
                }
            }
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    if (h->kind == 4) {
        const bool mon = (d.monitor & ~(uint32_t)KLARA_MON_ACCEPT) != 0;
        const bool da = d.tuner == KLARA_TUNER_DUAL_AVERAGING;
        return klara_launch_hiert(p, kl, d.sampler, h->E / 2, d.hier_ntimes, mon, !plain || da, da, grid_for(h), h->stream);
    }
    if (d.target == KLARA_TARGET_CUSTOM) return klara_jit_launch(h->jit, mode, p, kl, grid_for_transitions(h), lds_for(h), h->stream, 64 * h->custom_wpb);
    switch (d.sampler) {
    case KLARA_SAMPLER_MH: return klara_launch_mh(p, kl, mode, d.target, h->E, h->G, grid_for_transitions(h), lds_for(h), h->stream);
    case KLARA_SAMPLER_MALA: return klara_launch_mala(p, kl, mode, d.target, h->E, h->G, grid_for_transitions(h), lds_for(h), h->stream);
    case KLARA_SAMPLER_HMC: return klara_launch_hmc(p, kl, mode, d.target, h->E, h->G, grid_for_transitions(h), lds_for(h), h->stream);
    default: return klara_launch_slice(p, kl, mode, d.target, h->E, h->G, grid_for_transitions(h), lds_for(h), h->stream);
    }
}

// Closes one batch of every (chain, dimension) series in [i0, i1): batch mean from the running sums at the two batch
// boundaries, then Welford's update of the mean and the sum of squared deviations of the batch means (count = batches closed
// before this one).  mcvar.jl:35-41 takes var(batch means); the history-free form never revisits a sample.
__global__ __launch_bounds__(256) void k_bm_close(const double* __restrict__ sum, const double* __restrict__ X, const long long* __restrict__ held,
                                                  int D, double* __restrict__ prev, double* __restrict__ mean,
                                                  double* __restrict__ m2, long long i0, long long i1, long long count, double batchlen)
{
    const long long i = i0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i1) return;
    const long long hd = held[i / D];
    const double s = hd > 0 ? sum[i] + (double)hd * X[i] : sum[i];       // the running sum over the saved steps (sojourn form)
    const double b = (s - prev[i]) / batchlen;
    prev[i] = s;
    const double delta = b - mean[i];
    const double mn = mean[i] + delta / (double)(count + 1);
    mean[i] = mn;
    m2[i] = m2[i] + delta * (b - mn);
}

// every partition closes its own chains on its own stream (layout kind 3 runs chain partitions on internal streams)
static hipError_t launch_bm_close(klara_handle* h, int nparts)
{
    const long long N = h->d.nchains, D = h->d.ndims;
    const int np = h->kind == 3 ? nparts : 1;
    for (int j = 0; j < np; ++j) {
        long long c0 = 0, c1 = N;
        if (np > 1) part_range(h, np, j, &c0, &c1);          // (the chains the partition's transition kernels cover)
        if (c0 >= c1) break;
        const long long n = (c1 - c0) * D;
        hipLaunchKernelGGL(k_bm_close, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, j == 0 ? h->stream : h->side[j - 1], h->sum, h->X, h->held, (int)D,
                           h->bm_prev, h->bm_mean, h->bm_m2, c0 * D, c1 * D, h->bm_count, (double)h->d.bm_batchlen);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- launch planning: pure host logic (no device), shared by klara_run_async and klara_selftest_plan ------------------------
// Where the next launch of a run ends: after steps_per_launch transitions (library default KLARA_DEFAULT_STEPS_PER_LAUNCH), at the next event of the pooled
// tuner (tuners.jl:27-32 — the rate is pooled over the GPU's chains between launches) or where the next batch of saved samples
// closes (streaming batch means), whichever comes first; plus the save-rule bookkeeping the kernels take from the host
// (BasicMCRange.jl:36 postrange = (burnin+1):thinning:nsteps), so that they carry no 64-bit division.
struct RunCursor { long long steps_done, m_prop, m_tot, bm_count; };
struct PlannedLaunch { long long k; int save_phase0; long long save_col0; bool tune_after, bm_close_after; long long saved; };

// columns of the history ring of a job (0: every saved step is kept) — klara_desc.hist_ring_cols, or the 32-column value ring the
// streaming autocovariances keep for themselves when no value history was asked for
static long long ring_cols(const klara_desc& d)
{
    long long r = d.hist_ring_cols;
    if (d.acov_maxlag > 0 && !(d.monitor & KLARA_MON_HISTORY) && r == 0) r = 32;
    const long long npost = (d.nsteps - d.burnin - 1) / d.thinning + 1;
    return (r > 0 && r < npost) ? r : 0;
}
// saved (post-burn-in, thinned) steps among the first `steps` transitions
static long long saved_upto(const klara_desc& d, long long steps)
{
    const long long sd = steps < d.nsteps ? steps : d.nsteps;
    return sd > d.burnin ? (sd - d.burnin - 1) / d.thinning + 1 : 0;
}

static PlannedLaunch plan_launch(const klara_desc& d, const RunCursor& c, long long remaining)
{
    PlannedLaunch pl;
    const bool pooled_cnt = d.tuner_mode == KLARA_TUNE_POOLED && cnt_predicate(d);
    const long long spl = d.steps_per_launch > 0 ? d.steps_per_launch : default_steps_per_launch(d);
    long long k = remaining < spl ? remaining : spl;
    if (pooled_cnt && c.m_tot <= d.burnin) {
        const long long to_boundary = d.period - (c.m_prop % d.period);
        if (k > to_boundary) k = to_boundary;
    }
    // saved sample number S is transition burnin + (S - 1) * thinning + 1
    long long bm_close_at = -1;
    if (d.bm_batchlen > 0) {
        bm_close_at = d.burnin + ((c.bm_count + 1) * d.bm_batchlen - 1) * d.thinning + 1;
        if (bm_close_at > d.nsteps) bm_close_at = -1;
        else if (bm_close_at > c.steps_done && k > bm_close_at - c.steps_done) k = bm_close_at - c.steps_done;
    }
    // phase of the first post-burn-in step of this launch and the number of columns already saved
    pl.save_phase0 = c.steps_done >= d.burnin ? (int)((c.steps_done - d.burnin) % d.thinning) : 0;
    pl.save_col0 = c.steps_done > d.burnin ? (c.steps_done - d.burnin - 1) / d.thinning + 1 : 0;
    // history ring: a launch never wraps — it ends with the last column of the ring at the latest, the kernels get the ring
    // column of their first saved step (saved sample number S is transition burnin + (S - 1) * thinning + 1)
    const long long ring = ring_cols(d);
    if (ring > 0) {
        const long long room = ring - pl.save_col0 % ring;
        const long long last_ok = d.burnin + (pl.save_col0 + room - 1) * d.thinning + 1;       // transition of the last sample that fits
        const long long next = d.burnin + (pl.save_col0 + room) * d.thinning + 1;              // (the one after it would wrap)
        (void)last_ok;
        if (c.steps_done + k >= next) k = next - 1 - c.steps_done;
        if (k < 1) k = 1;
    }
    pl.k = k;
    pl.saved = saved_upto(d, c.steps_done + k) - saved_upto(d, c.steps_done);
    if (ring > 0) pl.save_col0 %= ring;
    pl.tune_after = pooled_cnt;
    pl.bm_close_after = bm_close_at >= 0 && c.steps_done + k == bm_close_at;
    return pl;
}

static void advance_cursor(const klara_desc& d, RunCursor& c, const PlannedLaunch& pl)
{
    if (pl.tune_after) {
        c.m_prop += pl.k;
        if (c.m_tot <= d.burnin && (c.m_prop % d.period) == 0) { c.m_tot += c.m_prop; c.m_prop = 0; }
    }
    c.steps_done += pl.k;
    if (pl.bm_close_after) ++c.bm_count;
}

// ---- streaming autocovariances (klara_desc.acov_maxlag): after every launch the samples it saved — columns [col0, col0 + m) of
// the value ring — update, for every (chain, dimension) series, the lagged cross-products S_k = sum_t x_t x_(t-k), k < W, the total,
// the first W samples and the last W samples (most recent first).  One thread per series, S and the window in registers.
template <int WMAX>
__global__ __launch_bounds__(256) void k_acov_update(const double* __restrict__ hist, long long col0, int m, long long n_before, int W,
                                                     long long nd, double* __restrict__ S, double* __restrict__ head,
                                                     double* __restrict__ tail, double* __restrict__ total)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd) return;
    double s[WMAX], win[WMAX];
#pragma unroll
    for (int k = 0; k < WMAX; ++k) { s[k] = k < W ? S[(long long)k * nd + i] : 0.0; win[k] = k < W ? tail[(long long)k * nd + i] : 0.0; }
    double tot = total[i];
    for (int j = 0; j < m; ++j) {
        const double x = hist[(col0 + j) * nd + i];
        s[0] = s[0] + x * x;
#pragma unroll
        for (int k = 1; k < WMAX; ++k) s[k] = s[k] + x * win[k - 1];        // (win holds zeros where no sample exists yet)
        if (n_before + j < W) head[(n_before + j) * nd + i] = x;
#pragma unroll
        for (int k = WMAX - 1; k > 0; --k) win[k] = win[k - 1];
        win[0] = x;
        tot = tot + x;
    }
#pragma unroll
    for (int k = 0; k < WMAX; ++k) if (k < W) { S[(long long)k * nd + i] = s[k]; tail[(long long)k * nd + i] = win[k]; }
    total[i] = tot;
}

// Lags 32 b .. 32 b + 31 (b >= 1; windows beyond 32 lags, klara_desc.acov_maxlag up to 127): the cross-products of x with its own history delayed by
// 32 b samples, y_j = x_(j - 32 b) — the same 32-lag update on the pair (x, y): S_(32b + r) += x_j y_(j - r).  y comes from the launch's own
// columns where j >= 32 b and from the tail kept by the earlier launches (tail[k] = the (k + 1)-th most recent sample before this launch)
// otherwise; the passes of the higher blocks run BEFORE the lag-0 pass rewrites the tail.
__global__ __launch_bounds__(256) void k_acov_update_block(const double* __restrict__ hist, long long col0, int m, int b, int W, long long nd,
                                                           double* __restrict__ S, const double* __restrict__ tail)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd) return;
    const int k0 = 32 * b;
    double s[32], win[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        s[r] = k0 + r < W ? S[(long long)(k0 + r) * nd + i] : 0.0;
        win[r] = k0 + r < W ? tail[(long long)(k0 + r) * nd + i] : 0.0;            // y_(-1-r) = x_(-1-r-k0)   (zeros where no sample exists yet)
    }
    for (int j = 0; j < m; ++j) {
        const double x = hist[(col0 + j) * nd + i];
        const double y = j >= k0 ? hist[(col0 + j - k0) * nd + i] : tail[(long long)(k0 - j - 1) * nd + i];
        s[0] = s[0] + x * y;
#pragma unroll
        for (int r = 1; r < 32; ++r) s[r] = s[r] + x * win[r - 1];
#pragma unroll
        for (int r = 31; r > 0; --r) win[r] = win[r - 1];
        win[0] = y;
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) if (k0 + r < W) S[(long long)(k0 + r) * nd + i] = s[r];
}
// ... and the tail of a window beyond 32 lags, after the lag-0 pass (which has moved entries 0..31): entries 32 .. W-1 take the samples that are now
// 33 .. W back — from this launch's columns or from the old tail, moved from the far end so that nothing is overwritten before it is read
__global__ __launch_bounds__(256) void k_acov_tail_far(const double* __restrict__ hist, long long col0, int m, int W, long long nd,
                                                       double* __restrict__ tail, const double* __restrict__ old_near)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd) return;
    for (int k = W - 1; k >= 32; --k) {                     // new tail[k] = x_(m-1-k): a column of this launch, or the old tail[k - m]
        double v;
        if (k < m) v = hist[(col0 + m - 1 - k) * nd + i];
        else if (k - m >= 32) v = tail[(long long)(k - m) * nd + i];            // (an entry further down: not yet overwritten — k descends)
        else v = old_near[(long long)(k - m) * nd + i];                          // (one of the first 32 entries as they were BEFORE the lag-0 pass)
        tail[(long long)k * nd + i] = v;
    }
}

static hipError_t launch_acov_update(klara_handle* h, long long col0, long long m)
{
    const long long nd = (long long)h->d.nchains * h->d.ndims;
    const dim3 grid((unsigned)((nd + 255) / 256)), blk(256);
    const int W = h->acov_W;
    if (W > 32) {       // windows beyond 32 lags: the higher lag blocks first (they read the tail as the earlier launches left it)
        for (int b = (W - 1) / 32; b >= 1; --b)
            hipLaunchKernelGGL(k_acov_update_block, grid, blk, 0, h->stream, h->hist, col0, (int)m, b, W, nd, h->acov_S, h->acov_tail);
        // (the lag-0 pass below rewrites tail[0..31]; the far tail needs their old values when fewer than 32 samples arrive: keep a copy)
        hipError_t e = hipMemcpyAsync(h->acov_near, h->acov_tail, (size_t)32 * nd * sizeof(double), hipMemcpyDeviceToDevice, h->stream);
        if (e != hipSuccess) return e;
    }
    if (W <= 8) hipLaunchKernelGGL((k_acov_update<8>), grid, blk, 0, h->stream, h->hist, col0, (int)m, h->acov_n, W, nd, h->acov_S, h->acov_head, h->acov_tail, h->acov_total);
    else if (W <= 16) hipLaunchKernelGGL((k_acov_update<16>), grid, blk, 0, h->stream, h->hist, col0, (int)m, h->acov_n, W, nd, h->acov_S, h->acov_head, h->acov_tail, h->acov_total);
    else hipLaunchKernelGGL((k_acov_update<32>), grid, blk, 0, h->stream, h->hist, col0, (int)m, h->acov_n, W, nd, h->acov_S, h->acov_head, h->acov_tail, h->acov_total);
    if (W > 32) hipLaunchKernelGGL(k_acov_tail_far, grid, blk, 0, h->stream, h->hist, col0, (int)m, W, nd, h->acov_tail, h->acov_near);
    return hipGetLastError();
}

// autocov(v, 0:maxlag) of StatsBase (demean = true) from the streamed sums: with m = total / n,
//   n acv_k = sum_(t>k) (x_t - m)(x_(t-k) - m) = S_k - m [(total - first k) + (total - last k)] + (n - k) m^2,
// then Geyer's truncation (mcvar.jl:75-105 imse, :137-158 ipse).
__global__ __launch_bounds__(256) void k_acov_finalize(const double* __restrict__ S, const double* __restrict__ head, const double* __restrict__ tail,
                                                       const double* __restrict__ total, long long n, int W, long long nd,
                                                       double* __restrict__ imse, double* __restrict__ ipse)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd) return;
    const double tot = total[i], mean = tot / (double)n;
    const long long maxlag = (W - 1) < (n - 1) ? (W - 1) : (n - 1);
    const long long kk = (maxlag - 1) / 2;                       // floor((maxlag-1)/2), mcvar.jl:76
    double hs = 0.0, ts = 0.0, acv0 = 0.0, gsum_m = 0.0, gsum_p = 0.0, gprev = 0.0;
    bool stop = false;
    for (long long j = 0; j <= kk && !stop; ++j) {
        double pair = 0.0;
        for (int h2 = 0; h2 < 2; ++h2) {
            const long long k = 2 * j + h2;
            // hs / ts = sum of the first / last k samples
            const double a = (S[k * nd + i] - mean * ((tot - ts) + (tot - hs)) + (double)(n - k) * mean * mean) / (double)n;
            if (k == 0) acv0 = a;
            pair += a;
            hs += head[k * nd + i]; ts += tail[k * nd + i];
        }
        if (pair <= 0.0) { stop = true; break; }                 // m = j (mcvar.jl:87-90)
        gsum_p += pair;
        double gm = pair;
        if (j > 0 && gm > gprev) gm = gprev;                     // monotone sequence (mcvar.jl:94-100)
        gsum_m += gm; gprev = gm;
    }
    if (imse) imse[i] = (-acv0 + 2.0 * gsum_m) / (double)n;
    if (ipse) ipse[i] = (-acv0 + 2.0 * gsum_p) / (double)n;
}

extern "C" klara_status klara_get_chain_acov_mcvar(klara_handle* h, double* mcvar_imse, double* mcvar_ipse, int64_t* nsamples_out)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->acov_S || !h->have_state || h->acov_n < 2) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const long long nd = (long long)h->d.nchains * h->d.ndims;
    double* buf = nullptr;
    HIPCHK(dalloc(&buf, (size_t)2 * nd));
    hipLaunchKernelGGL(k_acov_finalize, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, h->stream, h->acov_S, h->acov_head, h->acov_tail,
                       h->acov_total, h->acov_n, h->acov_W, nd, mcvar_imse ? buf : nullptr, mcvar_ipse ? buf + nd : nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && mcvar_imse) e = hipMemcpy(mcvar_imse, buf, nd * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && mcvar_ipse) e = hipMemcpy(mcvar_ipse, buf + nd, nd * sizeof(double), hipMemcpyDeviceToHost);
    (void)dfree(buf);
    if (nsamples_out) *nsamples_out = h->acov_n;
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

extern "C" klara_status klara_saved_steps(klara_handle* h, int64_t* nsaved_out)
{
    if (!h || !nsaved_out) return KLARA_ERR_INVALID_ARG;
    *nsaved_out = h->nsaved;
    return KLARA_OK;
}

extern "C" klara_status klara_run_async(klara_handle* h, int64_t nsteps)
{
    if (!h || nsteps < 0) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const klara_desc& d = h->d;
    if ((d.monitor & KLARA_MON_ACCEPT) && h->steps_done + nsteps > h->accept_cap) return KLARA_ERR_STATE;
    KParams p = make_params(h);
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    // Chain partitions on internal streams pay off when a run issues several launches per partition (the streams drift apart and
    // one partition's kernel fills the SIMDs while the other's ramps up or drains: 12.9 vs 14.1 us per transition at 65,536 x 100).
    // A run that is a single launch is issued whole on the caller's stream: two half-size kernels that start together only split
    // the machine unevenly (16.7 vs 17.8 us per transition for 20-transition runs).
    const long long spl_run = d.steps_per_launch > 0 ? d.steps_per_launch : default_steps_per_launch(d);
    const int nparts = nsteps > spl_run ? h->nparts : 1;
    if (nparts > 1) {                                      // fork: the partition streams start after everything queued so far
        HIPCHK(hipEventRecord(h->fork_ev, h->stream));
        for (int j = 0; j + 1 < nparts; ++j) HIPCHK(hipStreamWaitEvent(h->side[j], h->fork_ev, 0));
    }
    long long remaining = nsteps, launches = 0;
    RunCursor cur = { h->steps_done, h->m_prop, h->m_tot, h->bm_count };
    hipError_t err = hipSuccess;
    while (remaining > 0 && err == hipSuccess) {
        const PlannedLaunch pl = plan_launch(d, cur, remaining);
        KLaunch kl;
        kl.group0 = 0; kl.group_end = 0x7fffffffffffffffll;
        kl.t0 = (unsigned long long)cur.steps_done;
        kl.nsteps = (int)pl.k;                                             // (plan_launch keeps k <= steps_per_launch <= INT_MAX)
        kl.save_phase0 = pl.save_phase0;
        kl.save_col0 = pl.save_col0;
        err = launch_steps(h, kl, nparts);
        if (err == hipSuccess && pl.tune_after) {
            hipLaunchKernelGGL(k_pooled_tune, dim3(1), dim3(64), 0, h->stream, p, (int)pl.k);
            err = hipGetLastError();
        }
        if (err == hipSuccess && pl.bm_close_after) err = launch_bm_close(h, nparts);   // (reads h->bm_count: batches closed before this one)
        if (err == hipSuccess && h->acov_S && pl.saved > 0) {                   // streaming autocovariances consume the launch's samples
            err = launch_acov_update(h, pl.save_col0, pl.saved);
            h->acov_n += pl.saved;
        }
        if (err != hipSuccess) break;
        advance_cursor(d, cur, pl);
        h->steps_done = cur.steps_done; h->m_prop = cur.m_prop; h->m_tot = cur.m_tot; h->bm_count = cur.bm_count;
        remaining -= pl.k; ++launches;
    }
    // join — also when a launch failed: the partition streams may still have work in flight on X / GR / LT, and every later
    // call (get_state, reset, destroy) synchronises the caller's stream only
    for (int j = 0; j + 1 < nparts; ++j) {
        const hipError_t e1 = hipEventRecord(h->join_ev[j], h->side[j]);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(h->stream, h->join_ev[j], 0) : e1;
        if (e2 != hipSuccess) { hipStreamSynchronize(h->side[j]); if (err == hipSuccess) err = e2; }
    }
    HIPCHK(err);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->last_launches = launches; h->timed = true;
    // saved-sample bookkeeping: count of i in postrange with i <= steps_done
    const long long sd = h->steps_done < d.nsteps ? h->steps_done : d.nsteps;
    h->nsaved = sd > d.burnin ? (sd - d.burnin - 1) / d.thinning + 1 : 0;
    return KLARA_OK;
}

extern "C" klara_status klara_synchronize(klara_handle* h)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->d.device));
    int flag = 0;
    if (h->flag_host) {                     // the kernels store into this word of host memory themselves
        HIPCHK(hipStreamSynchronize(h->stream));
        flag = __atomic_load_n(h->flag_host, __ATOMIC_ACQUIRE);
    } else {
        HIPCHK(hipMemcpyAsync(&flag, h->err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (flag != 0) return (klara_status)flag;
    return KLARA_OK;
}

extern "C" klara_status klara_run(klara_handle* h, int64_t nsteps)
{
    klara_status s = klara_run_async(h, nsteps);
    if (s != KLARA_OK) return s;
    return klara_synchronize(h);
}

extern "C" klara_status klara_last_run_ms(klara_handle* h, double* kernel_ms, int64_t* nlaunches)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->timed) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (kernel_ms) *kernel_ms = (double)ms;
    if (nlaunches) *nlaunches = h->last_launches;
    return KLARA_OK;
}

extern "C" klara_status klara_get_state(klara_handle* h, double* x, double* logtarget, double* gradlogtarget)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const size_t N = (size_t)h->d.nchains, D = (size_t)h->d.ndims;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (x) HIPCHK(hipMemcpy(x, h->X, N * D * sizeof(double), hipMemcpyDeviceToHost));
    if (logtarget) HIPCHK(hipMemcpy(logtarget, h->LT, N * sizeof(double), hipMemcpyDeviceToHost));
    if (gradlogtarget) HIPCHK(hipMemcpy(gradlogtarget, h->GR, N * D * sizeof(double), hipMemcpyDeviceToHost));
    return KLARA_OK;
}

extern "C" klara_status klara_get_accept_mask(klara_handle* h, uint8_t* mask, int64_t capacity_steps,
                                              int64_t* nsteps_out)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->accept) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const long long n = h->steps_done < capacity_steps ? h->steps_done : capacity_steps;
    if (mask && n > 0) HIPCHK(hipMemcpy(mask, h->accept, (size_t)n * (size_t)h->d.nchains, hipMemcpyDeviceToHost));
    if (nsteps_out) *nsteps_out = h->steps_done;
    return KLARA_OK;
}

extern "C" klara_status klara_get_accept_rows(klara_handle* h, int64_t first_step, int64_t nsteps, uint8_t* mask)
{
    if (!h || first_step < 0 || nsteps < 0 || (nsteps > 0 && !mask)) return KLARA_ERR_INVALID_ARG;
    if (!h->accept || !h->have_state) return KLARA_ERR_STATE;
    if (first_step + nsteps > h->steps_done) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (nsteps > 0)
        HIPCHK(hipMemcpy(mask, h->accept + (size_t)first_step * (size_t)h->d.nchains, (size_t)nsteps * (size_t)h->d.nchains, hipMemcpyDeviceToHost));
    return KLARA_OK;
}

extern "C" klara_status klara_get_accept_counts(klara_handle* h, uint64_t* naccept, uint64_t* nsteps_out)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (naccept) HIPCHK(hipMemcpy(naccept, h->naccept, (size_t)h->d.nchains * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (nsteps_out) *nsteps_out = (uint64_t)h->steps_done;
    return KLARA_OK;
}

// sums over the saved steps from their sojourn form: out = part + held * x  (sq: x -> x * x); one product, one sum
__global__ __launch_bounds__(256) void k_sum_view(const double* __restrict__ part, const double* __restrict__ X, const long long* __restrict__ held,
                                                  long long n, int D, int sq, double* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long hd = held[i / D];
    const double x = X[i], v = sq ? x * x : x;
    out[i] = hd > 0 ? part[i] + (double)hd * v : part[i];
}

extern "C" klara_status klara_get_chain_sums(klara_handle* h, double* sum, double* sumsq, int64_t* nsaved_out)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->sum || !h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const size_t n = (size_t)h->d.nchains * (size_t)h->d.ndims;
    if (sum || sumsq) {
        double* view = nullptr;
        HIPCHK(dalloc(&view, n));
        hipError_t e = hipSuccess;
        for (int sq = 0; sq < 2 && e == hipSuccess; ++sq) {
            double* dst = sq ? sumsq : sum;
            if (!dst) continue;
            hipLaunchKernelGGL(k_sum_view, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, sq ? h->sumsq : h->sum, h->X, h->held,
                               (long long)n, h->d.ndims, sq, view);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(dst, view, n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        }
        (void)dfree(view);
        if (e != hipSuccess) return KLARA_ERR_HIP;
    } else {
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (nsaved_out) *nsaved_out = h->nsaved;
    return KLARA_OK;
}

// Pooled (over chains) per-dimension sums and the accepted-transition total, two stages of fixed shape => deterministic for a
// given chain count: block b of KLARA_POOL_BLOCKS adds the chains c = b, b + NB, ... column by column (a chain's D values are
// contiguous: coalesced) into partial[b][0..2D) and their accept counters into partial_acc[b]; one small block then adds the
// partials in ascending b.  (100 blocks striding over 105 MB took 167 us at 65,536 x 100; this takes ~25.)
#define KLARA_POOL_BLOCKS 1024
__global__ __launch_bounds__(256) void k_pool_stage1(const double* __restrict__ sum, const double* __restrict__ sumsq,
                                                     const double* __restrict__ X, const long long* __restrict__ held,
                                                     const unsigned long long* __restrict__ nacc, long long N, int D, int nb,
                                                     double* __restrict__ partial, unsigned long long* __restrict__ partial_acc)
{
    const int b = blockIdx.x;
    if (sum != nullptr) {
        for (int j = threadIdx.x; j < 2 * D; j += 256) {
            const bool sq = j >= D;
            const double* src = sq ? sumsq + (j - D) : sum + j;
            const double* xs = X + (sq ? j - D : j);
            double a = 0.0;
            for (long long c = b; c < N; c += nb) {                         // (a chain's sum over its saved steps: part + held * x)
                const long long hd = held[c];
                const double x = xs[c * D], v = sq ? x * x : x;
                a += hd > 0 ? src[c * D] + (double)hd * v : src[c * D];
            }
            partial[(long long)b * 2 * D + j] = a;
        }
    }
    __shared__ unsigned long long s[256];
    unsigned long long a = 0;
    for (long long c = (long long)b + (long long)threadIdx.x * nb; c < N; c += 256ll * nb) a += nacc[c];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) { if ((int)threadIdx.x < m) s[threadIdx.x] += s[threadIdx.x + m]; __syncthreads(); }
    if (threadIdx.x == 0) partial_acc[b] = s[0];
}
__global__ __launch_bounds__(256) void k_pool_stage2(const double* __restrict__ partial, const unsigned long long* __restrict__ partial_acc,
                                                     int D, int nb, bool with_sums, double* __restrict__ out)
{
    // block j < 2 D: column j of the partials; block 2 D: the accept counters.  Thread t adds the partials b = t, t + 256, ...
    // (ascending), then a fixed-shape tree over the 256 threads.
    __shared__ double sd[256];
    __shared__ unsigned long long su[256];
    const int j = blockIdx.x, t = threadIdx.x;
    if (j < 2 * D) {
        if (!with_sums) return;
        double a = 0.0;
        for (int b = t; b < nb; b += 256) a += partial[(long long)b * 2 * D + j];
        sd[t] = a;
        __syncthreads();
        for (int m = 128; m > 0; m >>= 1) { if (t < m) sd[t] += sd[t + m]; __syncthreads(); }
        if (t == 0) out[j] = sd[0];
    } else {
        unsigned long long a = 0;
        for (int b = t; b < nb; b += 256) a += partial_acc[b];
        su[t] = a;
        __syncthreads();
        for (int m = 128; m > 0; m >>= 1) { if (t < m) su[t] += su[t + m]; __syncthreads(); }
        if (t == 0) *reinterpret_cast<unsigned long long*>(out + 2 * D) = su[0];
    }
}
// out: 2 D doubles (sum, sumsq; untouched unless with_sums) followed by the u64 accept total; on the handle's stream
static hipError_t pool_summaries_async(klara_handle* h, bool with_sums, double* out)
{
    const int D = h->d.ndims;
    const long long N = h->d.nchains;
    const int nb = (int)(N < KLARA_POOL_BLOCKS ? N : KLARA_POOL_BLOCKS);
    hipLaunchKernelGGL(k_pool_stage1, dim3(nb), dim3(256), 0, h->stream, with_sums ? h->sum : nullptr, h->sumsq, h->X, h->held, h->naccept, N, D, nb,
                       h->pool_partial, reinterpret_cast<unsigned long long*>(h->pool_partial + (size_t)KLARA_POOL_BLOCKS * 2 * D));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_pool_stage2, dim3(2 * D + 1), dim3(256), 0, h->stream, h->pool_partial,
                       reinterpret_cast<const unsigned long long*>(h->pool_partial + (size_t)KLARA_POOL_BLOCKS * 2 * D), D, nb, with_sums, out);
    return hipGetLastError();
}

extern "C" klara_status klara_get_pooled_summaries(klara_handle* h, double* sum, double* sumsq,
                                                   uint64_t* naccept, uint64_t* ntransitions,
                                                   int64_t* nsaved_out)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const int D = h->d.ndims;
    if ((sum || sumsq) && !h->sum) return KLARA_ERR_STATE;
    HIPCHK(pool_summaries_async(h, sum || sumsq, h->pooled_out));
    std::vector<double> host(2 * (size_t)D + 1);
    HIPCHK(hipMemcpyAsync(host.data(), h->pooled_out, (2 * (size_t)D + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));   // one copy
    HIPCHK(hipStreamSynchronize(h->stream));
    if (sum) memcpy(sum, host.data(), D * sizeof(double));
    if (sumsq) memcpy(sumsq, host.data() + D, D * sizeof(double));
    if (naccept) memcpy(naccept, host.data() + 2 * D, sizeof(uint64_t));
    if (ntransitions) *ntransitions = (uint64_t)h->steps_done * (uint64_t)h->d.nchains;
    if (nsaved_out) *nsaved_out = h->nsaved;
    return KLARA_OK;
}

// ---- RCCL summary all-reduce behind the C ABI (librccl.so loaded lazily; see include/klara_hip.h)
struct klara_comm {
    void* dl = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    double* buf = nullptr; size_t cap = 0;                      // device staging: 2D doubles + 4 u64
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
static void* rccl_dl()
{
    static void* dl = nullptr;
    if (!dl) dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) dl = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return dl;
}
extern "C" klara_status klara_comm_unique_id(uint8_t id[KLARA_COMM_ID_BYTES])
{
    static_assert(KLARA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id) return KLARA_ERR_INVALID_ARG;
    void* dl = rccl_dl();
    if (!dl) return KLARA_ERR_UNSUPPORTED;
    auto get = reinterpret_cast<ncclResult_t (*)(ncclUniqueId*)>(dlsym(dl, "ncclGetUniqueId"));
    ncclUniqueId u;
    if (!get || get(&u) != ncclSuccess) return KLARA_ERR_HIP;
    memcpy(id, u.internal, KLARA_COMM_ID_BYTES);
    return KLARA_OK;
}
extern "C" klara_status klara_comm_init(klara_comm** out, int32_t nranks, int32_t rank, const uint8_t id[KLARA_COMM_ID_BYTES],
                                        int32_t device)
{
    if (!out || !id || nranks <= 0 || rank < 0 || rank >= nranks) return KLARA_ERR_INVALID_ARG;
    void* dl = rccl_dl();
    if (!dl) return KLARA_ERR_UNSUPPORTED;
    HIPCHK(hipSetDevice(device));
    auto init = reinterpret_cast<ncclResult_t (*)(ncclComm_t*, int, ncclUniqueId, int)>(dlsym(dl, "ncclCommInitRank"));
    klara_comm* c = new (std::nothrow) klara_comm();
    if (!c) return KLARA_ERR_NOMEM;
    c->dl = dl; c->nranks = nranks; c->rank = rank; c->device = device;
    c->AllReduce = reinterpret_cast<decltype(c->AllReduce)>(dlsym(dl, "ncclAllReduce"));
    c->CommDestroy = reinterpret_cast<decltype(c->CommDestroy)>(dlsym(dl, "ncclCommDestroy"));
    c->CommCount = reinterpret_cast<decltype(c->CommCount)>(dlsym(dl, "ncclCommCount"));
    c->CommUserRank = reinterpret_cast<decltype(c->CommUserRank)>(dlsym(dl, "ncclCommUserRank"));
    ncclUniqueId u;
    memcpy(u.internal, id, KLARA_COMM_ID_BYTES);
    if (!init || !c->AllReduce || !c->CommDestroy || init(&c->comm, nranks, u, rank) != ncclSuccess) { delete c; return KLARA_ERR_HIP; }
    *out = c;
    return KLARA_OK;
}
// what the communicator itself says it is (ncclCommCount / ncclCommUserRank), not what the caller passed to klara_comm_init
extern "C" klara_status klara_comm_info(klara_comm* c, int32_t* nranks, int32_t* rank, int32_t* device)
{
    if (!c || !c->comm) return KLARA_ERR_INVALID_ARG;
    if (!c->CommCount || !c->CommUserRank) return KLARA_ERR_UNSUPPORTED;
    int n = 0, r = 0;
    if (c->CommCount(c->comm, &n) != ncclSuccess || c->CommUserRank(c->comm, &r) != ncclSuccess) return KLARA_ERR_HIP;
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = c->device;
    return KLARA_OK;
}
extern "C" klara_status klara_comm_destroy(klara_comm* c)
{
    if (!c) return KLARA_ERR_INVALID_ARG;
    hipSetDevice(c->device);
    if (c->comm) c->CommDestroy(c->comm);
    if (c->buf) (void)dfree(c->buf);
    delete c;
    return KLARA_OK;
}
extern "C" klara_status klara_gather_summaries(klara_handle* h, klara_comm* c, double* sum, double* sumsq, uint64_t* naccept,
                                               uint64_t* ntransitions, uint64_t* nsamples, uint64_t* nchains)
{
    if (!h || !c) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state) return KLARA_ERR_STATE;
    if ((sum || sumsq) && !h->sum) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const size_t D = (size_t)h->d.ndims;
    if (c->cap < 2 * D + 4) {
        if (c->buf) (void)dfree(c->buf);
        c->buf = nullptr; c->cap = 0;
        HIPCHK(dalloc(&c->buf, 2 * D + 4));
        c->cap = 2 * D + 4;
    }
    // From here on every rank is inside a fixed sequence of collectives: a rank that returned at its first local failure would leave the others
    // waiting in theirs.  Local errors are remembered, every collective is still entered, the status is reported after the last one.
    bool bad = false;
    const auto H = [&](hipError_t e) { if (e != hipSuccess) bad = true; };
    H(hipMemsetAsync(c->buf, 0, (2 * D + 4) * sizeof(double), h->stream));
    H(pool_summaries_async(h, h->sum != nullptr, c->buf));
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(c->buf + 2 * D);
    const unsigned long long local[3] = { (unsigned long long)h->steps_done * (unsigned long long)h->d.nchains,
                                          (unsigned long long)h->nsaved * (unsigned long long)h->d.nchains,
                                          (unsigned long long)h->d.nchains };
    H(hipMemcpyAsync(cnt + 1, local, sizeof(local), hipMemcpyHostToDevice, h->stream));
    // two in-place all-reduces on the job's stream: 2D doubles, 4 counters — (2D + 4) x 8 B per rank, latency-bound
    if (c->AllReduce(c->buf, c->buf, 2 * D, ncclDouble, ncclSum, c->comm, h->stream) != ncclSuccess) bad = true;
    if (c->AllReduce(cnt, cnt, 4, ncclUint64, ncclSum, c->comm, h->stream) != ncclSuccess) bad = true;
    std::vector<double> host(2 * D + 4);
    H(hipMemcpyAsync(host.data(), c->buf, (2 * D + 4) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    H(hipStreamSynchronize(h->stream));
    if (bad) return KLARA_ERR_HIP;
    if (sum) memcpy(sum, host.data(), D * sizeof(double));
    if (sumsq) memcpy(sumsq, host.data() + D, D * sizeof(double));
    unsigned long long out[4];
    memcpy(out, host.data() + 2 * D, sizeof(out));
    if (naccept) *naccept = out[0];
    if (ntransitions) *ntransitions = out[1];
    if (nsamples) *nsamples = out[2];
    if (nchains) *nchains = out[3];
    return KLARA_OK;
}

// ---- pooled moments without cancellation (the consumer is mean(chain) / var over every chain,
// src/stats/mean.jl:7-11).  sumsq/n - mean^2 from pooled raw sums loses mean^2/var digits (rats alpha_c: mean 242, sd 2.7 -> 4
// digits).  Here every chain's raw sums become (n_c, mean_c, M2_c) with the one cancelling subtraction, q - s^2/n, carried in
// double-double (s^2 = p + pe exactly by fma, the quotient's remainder by fma), and chains, blocks and ranks are merged by
// Chan's update — sums of non-negative terms only.  Fixed shapes => deterministic for a given chain count.
__global__ void k_scale(double* __restrict__ dst, const double* __restrict__ src, double f, int n)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = f * src[j];
}
__device__ inline void chan_merge(double& n, double& mean, double& m2, double nb, double meanb, double m2b)
{
    const double nt = n + nb;
    if (!(nt > 0.0)) return;
    const double delta = meanb - mean, w = nb / nt;
    mean = mean + delta * w;
    m2 = (m2 + m2b) + (delta * delta) * (n * w);
    n = nt;
}
__global__ __launch_bounds__(256) void k_moments_stage1(const double* __restrict__ sum, const double* __restrict__ sumsq,
                                                        const double* __restrict__ X, const long long* __restrict__ held,
                                                        long long N, int D, int nb, double nsaved, double* __restrict__ partial)
{
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < D; j += 256) {
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (long long c = b; c < N; c += nb) {                             // (a chain's sums over its saved steps: part + held * x)
            const long long hd = held[c];
            const double x = X[c * D + j];
            const double s = hd > 0 ? sum[c * D + j] + (double)hd * x : sum[c * D + j];
            const double q = hd > 0 ? sumsq[c * D + j] + (double)hd * (x * x) : sumsq[c * D + j];
            const double p = s * s, pe = fma(s, s, -p);                    // s^2 = p + pe
            const double qh = p / nsaved, r = fma(-qh, nsaved, p);         // p = qh * nsaved + r
            const double ql = (r + pe) / nsaved;                           // s^2 / nsaved = qh + ql (+ O(2^-105))
            double m2c = (q - qh) - ql;
            if (m2c < 0.0) m2c = 0.0;
            chan_merge(n, mean, m2, nsaved, s / nsaved, m2c);
        }
        partial[(long long)b * 2 * D + j] = mean;
        partial[(long long)b * 2 * D + D + j] = m2;
    }
}
// block j: dimension j.  Thread t merges the blocks' partials b = t, t + 256, ... (ascending), then a fixed tree over the threads.
// out[j] = mean, out[D + j] = M2 over the handle's chains; the count is nsaved * N.
__global__ __launch_bounds__(256) void k_moments_stage2(const double* __restrict__ partial, long long N, int D, int nb, double nsaved,
                                                        double* __restrict__ out)
{
    __shared__ double sn[256], sm[256], sq[256];
    const int j = blockIdx.x, t = threadIdx.x;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int b = t; b < nb; b += 256) {
        const long long chains = (N - b + nb - 1) / nb;                     // chains c = b, b + nb, ... < N
        chan_merge(n, mean, m2, nsaved * (double)chains, partial[(long long)b * 2 * D + j], partial[(long long)b * 2 * D + D + j]);
    }
    sn[t] = n; sm[t] = mean; sq[t] = m2;
    __syncthreads();
    for (int m = 128; m > 0; m >>= 1) {
        if (t < m) chan_merge(sn[t], sm[t], sq[t], sn[t + m], sm[t + m], sq[t + m]);
        __syncthreads();
    }
    if (t == 0) { out[j] = sm[0]; out[D + j] = sq[0]; }
}
// rank-local part of the between-rank merge: buf[j] = mean_r -> M2_r + n_r (mean_r - mean)^2 with mean = wsum[j] / ntot
__global__ void k_moments_between(double* __restrict__ m2, const double* mean_r, const double* __restrict__ wsum,
                                  const unsigned long long* __restrict__ ntot, double n_r, int D, double* mean_out)   // (mean_out may be mean_r)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= D) return;
    const double nt = (double)*ntot;
    const double mean = nt > 0.0 ? wsum[j] / nt : 0.0, d = mean_r[j] - mean;       // (no saved sample on any rank: zeros, as chan_merge leaves them without a communicator)
    mean_out[j] = mean;
    m2[j] = m2[j] + n_r * (d * d);
}
// out: mean[D], M2[D], then the u64 accept total; on the handle's stream
static hipError_t pool_moments_async(klara_handle* h, double* out)
{
    const int D = h->d.ndims;
    const long long N = h->d.nchains;
    const int nb = (int)(N < KLARA_POOL_BLOCKS ? N : KLARA_POOL_BLOCKS);
    hipError_t e = pool_summaries_async(h, false, out);                     // the accept total -> out[2 D]
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_moments_stage1, dim3(nb), dim3(256), 0, h->stream, h->sum, h->sumsq, h->X, h->held, N, D, nb, (double)h->nsaved, h->pool_partial);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k_moments_stage2, dim3(D), dim3(256), 0, h->stream, h->pool_partial, N, D, nb, (double)h->nsaved, out);
    return hipGetLastError();
}

extern "C" klara_status klara_gather_moments(klara_handle* h, klara_comm* c, double* mean, double* m2, uint64_t* nsamples,
                                             uint64_t* naccept, uint64_t* ntransitions, uint64_t* nchains)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state || !h->sum) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const size_t D = (size_t)h->d.ndims;
    unsigned long long cnt[4] = { 0, (unsigned long long)h->steps_done * (unsigned long long)h->d.nchains,
                                  (unsigned long long)h->nsaved * (unsigned long long)h->d.nchains, (unsigned long long)h->d.nchains };
    std::vector<double> host(2 * D + 1);
    if (!c) {                                                               // this handle's chains only
        HIPCHK(pool_moments_async(h, h->pooled_out));
        HIPCHK(hipMemcpyAsync(host.data(), h->pooled_out, (2 * D + 1) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        memcpy(&cnt[0], host.data() + 2 * D, sizeof(cnt[0]));
    } else {
        // device staging: [0, D) mean_r -> mean, [D, 2D) M2_r -> M2, [2D] accept total, then [2D+1, 2D+4) counters, [2D+4, 3D+4) n_r mean_r
        if (c->cap < 3 * D + 4) {
            if (c->buf) (void)dfree(c->buf);
            c->buf = nullptr; c->cap = 0;
            HIPCHK(dalloc(&c->buf, 3 * D + 4));
            c->cap = 3 * D + 4;
        }
        // (as in klara_gather_summaries: once the sequence of collectives starts, every rank goes through all of it; errors are reported after the last)
        bool bad = false;
        const auto H = [&](hipError_t e) { if (e != hipSuccess) bad = true; };
        H(pool_moments_async(h, c->buf));
        unsigned long long* dcnt = reinterpret_cast<unsigned long long*>(c->buf + 2 * D);
        H(hipMemcpyAsync(dcnt + 1, cnt + 1, 3 * sizeof(cnt[0]), hipMemcpyHostToDevice, h->stream));
        double* wsum = c->buf + 2 * D + 4;
        const double n_r = (double)cnt[2];
        hipLaunchKernelGGL(k_scale, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, h->stream, wsum, c->buf, n_r, (int)D);
        H(hipGetLastError());
        // three in-place all-reduces on the job's stream: 4 counters, D weighted means, D sums of squares — latency-bound
        if (c->AllReduce(dcnt, dcnt, 4, ncclUint64, ncclSum, c->comm, h->stream) != ncclSuccess) bad = true;
        if (c->AllReduce(wsum, wsum, D, ncclDouble, ncclSum, c->comm, h->stream) != ncclSuccess) bad = true;
        hipLaunchKernelGGL(k_moments_between, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, h->stream, c->buf + D, c->buf, wsum, dcnt + 2, n_r, (int)D, c->buf);
        H(hipGetLastError());
        if (c->AllReduce(c->buf + D, c->buf + D, D, ncclDouble, ncclSum, c->comm, h->stream) != ncclSuccess) bad = true;
        std::vector<double> hostc(2 * D + 4);
        H(hipMemcpyAsync(hostc.data(), c->buf, (2 * D + 4) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        H(hipStreamSynchronize(h->stream));
        if (bad) return KLARA_ERR_HIP;
        memcpy(host.data(), hostc.data(), 2 * D * sizeof(double));
        memcpy(cnt, hostc.data() + 2 * D, sizeof(cnt));
    }
    if (mean) memcpy(mean, host.data(), D * sizeof(double));
    if (m2) memcpy(m2, host.data() + D, D * sizeof(double));
    if (naccept) *naccept = cnt[0];
    if (ntransitions) *ntransitions = cnt[1];
    if (nsamples) *nsamples = cnt[2];
    if (nchains) *nchains = cnt[3];
    return KLARA_OK;
}

// Columns [first, first + count) of the saved steps from a history buffer whose column c lives at slot c % hist_cols (ring) or c:
// `width` bytes per column copied to consecutive rows of dst, `stride` bytes between the buffer's columns.
static hipError_t copy_saved_columns(const klara_handle* h, void* dst, const char* src, size_t width, size_t stride, long long first, long long count)
{
    long long done = 0;
    while (done < count) {
        const long long slot = h->ring ? (first + done) % h->hist_cols : first + done;
        long long run = count - done;
        if (h->ring && slot + run > h->hist_cols) run = h->hist_cols - slot;
        const hipError_t e = hipMemcpy2D((char*)dst + (size_t)done * width, width, src + (size_t)slot * stride, stride, width, (size_t)run, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        done += run;
    }
    return hipSuccess;
}
// the saved steps a history read returns: all of them, or the last hist_cols of a ring
static void saved_window(const klara_handle* h, long long capacity, long long* first, long long* count)
{
    const long long avail = h->ring && h->nsaved > h->hist_cols ? h->hist_cols : h->nsaved;
    *count = avail < capacity ? avail : capacity;
    *first = h->nsaved - avail;
}

extern "C" klara_status klara_get_chain(klara_handle* h, int64_t local_chain, double* value,
                                        int64_t capacity_cols, int64_t* ncols_out)
{
    if (!h || local_chain < 0 || local_chain >= h->d.nchains) return KLARA_ERR_INVALID_ARG;
    if (!h->hist || !(h->d.monitor & KLARA_MON_HISTORY)) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->d.nchains, D = (size_t)h->d.ndims;
    long long first, n;
    saved_window(h, capacity_cols, &first, &n);
    // device layout [col][chain][D] -> Klara NState.value (D x n) column-major = n contiguous D-vectors
    if (value && n > 0)
        HIPCHK(copy_saved_columns(h, value, (const char*)(h->hist + (size_t)local_chain * D), D * sizeof(double), N * D * sizeof(double), first, n));
    if (ncols_out) { long long f2, all; saved_window(h, 0x7fffffffffffffffll, &f2, &all); *ncols_out = all; }
    return KLARA_OK;
}

extern "C" klara_status klara_get_chain_fields(klara_handle* h, int64_t local_chain, double* logtarget,
                                               double* gradlogtarget, int64_t capacity_cols, int64_t* ncols_out)
{
    if (!h || local_chain < 0 || local_chain >= h->d.nchains) return KLARA_ERR_INVALID_ARG;
    if ((logtarget && !h->hist_lt) || (gradlogtarget && !h->hist_g) || (!h->hist_lt && !h->hist_g)) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->d.nchains, D = (size_t)h->d.ndims;
    long long first, n;
    saved_window(h, capacity_cols, &first, &n);
    if (logtarget && n > 0)
        HIPCHK(copy_saved_columns(h, logtarget, (const char*)(h->hist_lt + (size_t)local_chain), sizeof(double), N * sizeof(double), first, n));
    if (gradlogtarget && n > 0)
        HIPCHK(copy_saved_columns(h, gradlogtarget, (const char*)(h->hist_g + (size_t)local_chain * D), D * sizeof(double), N * D * sizeof(double), first, n));
    if (ncols_out) { long long f2, all; saved_window(h, 0x7fffffffffffffffll, &f2, &all); *ncols_out = all; }
    return KLARA_OK;
}

extern "C" klara_status klara_get_chain_likelihood_prior(klara_handle* h, int64_t local_chain, double* loglikelihood, double* logprior,
                                                         int64_t capacity_cols, int64_t* ncols_out)
{
    if (!h || local_chain < 0 || local_chain >= h->d.nchains) return KLARA_ERR_INVALID_ARG;
    if (!h->hist_ll) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->d.nchains;
    long long first, n;
    saved_window(h, capacity_cols, &first, &n);
    if (loglikelihood && n > 0)
        HIPCHK(copy_saved_columns(h, loglikelihood, (const char*)(h->hist_ll + (size_t)local_chain), sizeof(double), N * sizeof(double), first, n));
    if (logprior && n > 0)
        HIPCHK(copy_saved_columns(h, logprior, (const char*)(h->hist_lp + (size_t)local_chain), sizeof(double), N * sizeof(double), first, n));
    if (ncols_out) { long long f2, all; saved_window(h, 0x7fffffffffffffffll, &f2, &all); *ncols_out = all; }
    return KLARA_OK;
}

// Post-hoc Monte Carlo variance estimators of src/stats/variance/mcvar.jl over the stored history, one thread per
// (chain, dimension) series; consecutive threads read consecutive dimensions of one chain (coalesced).
//   iid  : var(v)/n                                   (mcvar.jl:5)
//   bm   : batchlen * var(batch means) / nbsamples    (mcvar.jl:35-41, Flegal & Jones)
//   imse : Geyer's initial monotone sequence estimator on the empirical autocovariance (mcvar.jl:75-105);
//          autocov(v, k) = sum_t (v_t - m)(v_{t+k} - m) / n  (StatsBase, demean = true), evaluated lag pair by lag
//          pair until the first non-positive pair sum, so the cost is O(n * stopping lag) instead of O(n^2).
__global__ __launch_bounds__(256) void k_chain_stats(const double* __restrict__ hist, long long ncols, long long N, int D,
                                                     long long batchlen, long long maxlag, double* iid, double* bm,
                                                     double* imse, double* ipse)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const double* v = hist + i;                 // v[t * stride]
    const long long stride = N * D, n = ncols;
    double m = 0.0;
    for (long long t = 0; t < n; ++t) m += v[t * stride];
    m /= (double)n;
    double acv0 = 0.0;
    for (long long t = 0; t < n; ++t) { const double z = v[t * stride] - m; acv0 += z * z; }
    if (iid) iid[i] = (n > 1) ? acv0 / (double)(n - 1) / (double)n : NAN;
    if (bm) {
        const long long nb = batchlen > 0 ? n / batchlen : 0;
        if (nb > 1) {
            double sb = 0.0, sb2 = 0.0;                       // two-pass over batch means for stability
            for (long long b = 0; b < nb; ++b) {
                double a = 0.0;
                for (long long t = 0; t < batchlen; ++t) a += v[(b * batchlen + t) * stride];
                sb += a / (double)batchlen;
            }
            const double mb = sb / (double)nb;
            for (long long b = 0; b < nb; ++b) {
                double a = 0.0;
                for (long long t = 0; t < batchlen; ++t) a += v[(b * batchlen + t) * stride];
                const double dm = a / (double)batchlen - mb;
                sb2 += dm * dm;
            }
            bm[i] = (double)batchlen * (sb2 / (double)(nb - 1)) / (double)(nb * batchlen);
        } else bm[i] = NAN;                                   // "Choose batch size such that the number of batches is > 1"
    }
    if (imse || ipse) {
        const long long ml = maxlag < n - 1 ? maxlag : n - 1;
        const long long k = (ml - 1) / 2;                     // floor((maxlag-1)/2), mcvar.jl:76
        double gsum = 0.0, gprev = 0.0, gsum_p = 0.0;
        for (long long j = 0; j <= k; ++j) {
            double a0 = 0.0, a1 = 0.0;
            const long long l0 = 2 * j, l1 = 2 * j + 1;
            if (j == 0) a0 = acv0;
            else for (long long t = 0; t + l0 < n; ++t) a0 += (v[t * stride] - m) * (v[(t + l0) * stride] - m);
            for (long long t = 0; t + l1 < n; ++t) a1 += (v[t * stride] - m) * (v[(t + l1) * stride] - m);
            double gj = (a0 + a1) / (double)n;
            if (gj <= 0.0) break;                             // m = j (mcvar.jl:87-90)
            gsum_p += gj;                                     // initial positive sequence (mcvar.jl:137-158): no monotone step
            if (j > 0 && gj > gprev) gj = gprev;              // monotone sequence (mcvar.jl:94-100)
            gsum += gj; gprev = gj;
        }
        if (imse) imse[i] = (-acv0 / (double)n + 2.0 * gsum) / (double)n;
        if (ipse) ipse[i] = (-acv0 / (double)n + 2.0 * gsum_p) / (double)n;
    }
}

extern "C" klara_status klara_get_chain_mcvar(klara_handle* h, int64_t batchlen, int64_t maxlag, double* mcvar_iid,
                                              double* mcvar_bm, double* mcvar_imse)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->hist || h->ring || !(h->d.monitor & KLARA_MON_HISTORY) || !h->have_state || h->nsaved < 2) return KLARA_ERR_STATE;   // (needs every saved step)
    HIPCHK(hipSetDevice(h->d.device));
    const long long N = h->d.nchains, D = h->d.ndims, total = N * D;
    double* buf = nullptr;
    HIPCHK(dalloc(&buf, (size_t)3 * total));
    hipLaunchKernelGGL(k_chain_stats, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->hist,
                       (long long)h->nsaved, N, (int)D, (long long)batchlen, (long long)(maxlag > 0 ? maxlag : h->nsaved - 1),
                       mcvar_iid ? buf : nullptr, mcvar_bm ? buf + total : nullptr, mcvar_imse ? buf + 2 * total : nullptr, (double*)nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && mcvar_iid) e = hipMemcpy(mcvar_iid, buf, total * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && mcvar_bm) e = hipMemcpy(mcvar_bm, buf + total, total * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && mcvar_imse) e = hipMemcpy(mcvar_imse, buf + 2 * total, total * sizeof(double), hipMemcpyDeviceToHost);
    (void)dfree(buf);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

extern "C" klara_status klara_get_chain_mcvar_ipse(klara_handle* h, int64_t maxlag, double* mcvar_ipse)
{
    if (!h || !mcvar_ipse) return KLARA_ERR_INVALID_ARG;
    if (!h->hist || h->ring || !(h->d.monitor & KLARA_MON_HISTORY) || !h->have_state || h->nsaved < 2) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    const long long N = h->d.nchains, D = h->d.ndims, total = N * D;
    double* buf = nullptr;
    HIPCHK(dalloc(&buf, (size_t)total));
    hipLaunchKernelGGL(k_chain_stats, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->hist, (long long)h->nsaved, N, (int)D, 0ll,
                       (long long)(maxlag > 0 ? maxlag : h->nsaved - 1), (double*)nullptr, (double*)nullptr, (double*)nullptr, buf);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) e = hipMemcpy(mcvar_ipse, buf, total * sizeof(double), hipMemcpyDeviceToHost);
    (void)dfree(buf);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

extern "C" klara_status klara_get_chain_bm(klara_handle* h, double* mcvar_bm, int64_t* nbatches_out)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->bm_prev || !h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->d.nchains * (size_t)h->d.ndims;
    const long long nb = h->bm_count;
    if (mcvar_bm) {
        HIPCHK(hipMemcpy(mcvar_bm, h->bm_m2, n * sizeof(double), hipMemcpyDeviceToHost));
        // batchlen * var(batch means) / (nbatches * batchlen)  (mcvar.jl:39-41), var = M2 / (nbatches - 1)
        for (size_t i = 0; i < n; ++i)
            mcvar_bm[i] = nb > 1 ? (double)h->d.bm_batchlen * (mcvar_bm[i] / (double)(nb - 1)) / (double)(nb * h->d.bm_batchlen) : (double)NAN;
    }
    if (nbatches_out) *nbatches_out = nb;
    return KLARA_OK;
}

extern "C" klara_status klara_get_tune(klara_handle* h, double* step, int64_t* accepted, int64_t* proposed,
                                       int64_t* totproposed)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->d.nchains;
    if (h->d.tuner_mode == KLARA_TUNE_POOLED) {
        double s; long long a, pr, t;
        HIPCHK(hipMemcpy(&s, h->tune_step, sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&a, h->tune_acc, sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&pr, h->tune_prop, sizeof(long long), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&t, h->tune_tot, sizeof(long long), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < N; ++i) {
            if (step) step[i] = s;
            if (accepted) accepted[i] = a;
            if (proposed) proposed[i] = pr;
            if (totproposed) totproposed[i] = t;
        }
        return KLARA_OK;
    }
    if (step) HIPCHK(hipMemcpy(step, h->tune_step, N * sizeof(double), hipMemcpyDeviceToHost));
    if (accepted) HIPCHK(hipMemcpy(accepted, h->tune_acc, N * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (proposed) HIPCHK(hipMemcpy(proposed, h->tune_prop, N * sizeof(int64_t), hipMemcpyDeviceToHost));
    if (totproposed) HIPCHK(hipMemcpy(totproposed, h->tune_tot, N * sizeof(int64_t), hipMemcpyDeviceToHost));
    return KLARA_OK;
}

extern "C" klara_status klara_get_dual_averaging(klara_handle* h, double* epsbar, double* hbar)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (!h->have_state || !h->da_epsbar) return KLARA_ERR_STATE;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const size_t N = (size_t)h->d.nchains;
    if (epsbar) HIPCHK(hipMemcpy(epsbar, h->da_epsbar, N * sizeof(double), hipMemcpyDeviceToHost));
    if (hbar) HIPCHK(hipMemcpy(hbar, h->da_hbar, N * sizeof(double), hipMemcpyDeviceToHost));
    return KLARA_OK;
}

extern "C" klara_status klara_device_ptrs(klara_handle* h, void** x, void** logtarget, void** gradlogtarget)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (x) *x = h->X;
    if (logtarget) *logtarget = h->LT;
    if (gradlogtarget) *gradlogtarget = h->GR;
    return KLARA_OK;
}

extern "C" klara_status klara_get_layout(klara_handle* h, int32_t* kind, int32_t* lanes_per_chain,
                                         int32_t* elems_per_lane)
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (kind) *kind = h->kind;
    if (lanes_per_chain) *lanes_per_chain = h->kind == 2 ? h->RS : h->G;
    if (elems_per_lane) *elems_per_lane = h->E;
    return KLARA_OK;
}

extern "C" klara_status klara_get_kernel_attributes(klara_handle* h, int32_t which, int32_t nsteps, int32_t* vgprs, int32_t* scratch_bytes,
                                                    int32_t* static_lds_bytes)
{
    if (!h || nsteps < 1 || which < 0 || which > 1) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(h->d.device));
    hipFuncAttributes a;
    memset(&a, 0, sizeof(a));
    KLaunch kl;
    kl.group0 = 0; kl.group_end = 0x7fffffffffffffffll; kl.t0 = 0; kl.nsteps = nsteps; kl.save_phase0 = 0; kl.save_col0 = 0;
    h->query_lanes = which == 1 ? 8 : 4;
    klara_attr_query = &a;
    const hipError_t e = launch_steps(h, kl, 1);
    klara_attr_query = nullptr;
    HIPCHK(e);
    if (vgprs) *vgprs = a.numRegs;
    if (scratch_bytes) *scratch_bytes = (int32_t)a.localSizeBytes;
    if (static_lds_bytes) *static_lds_bytes = (int32_t)a.sharedSizeBytes;
    return KLARA_OK;
}

extern "C" klara_status klara_get_shader_clock(klara_handle* h, double* mhz)
{
    if (!h || !mhz) return KLARA_ERR_INVALID_ARG;
    *mhz = 0.0;
    if (!h->clock_probe) return KLARA_OK;
    HIPCHK(hipSetDevice(h->d.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int j = 0; j < 3; ++j) if (h->side[j]) HIPCHK(hipStreamSynchronize(h->side[j]));
    unsigned long long v[4] = { 0, 0, 0, 0 };
    HIPCHK(hipMemcpy(v, h->clock_probe, sizeof(v), hipMemcpyDeviceToHost));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->d.device) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
    if (v[1] > v[3] && v[0] > v[2]) *mhz = (double)(v[0] - v[2]) / (double)(v[1] - v[3]) * (double)khz * 1e-3;
    return KLARA_OK;
}

extern "C" klara_status klara_get_launch_modes(klara_handle* h, int64_t counts[3], int32_t last_mode[4], int64_t last_accepted[4])
{
    if (!h) return KLARA_ERR_INVALID_ARG;
    if (counts) for (int i = 0; i < 3; ++i) counts[i] = h->n_launch_mode[i];
    for (int j = 0; j < 4; ++j) {
        const bool have = h->auto_mirror != nullptr;
        if (last_mode) last_mode[j] = have ? __atomic_load_n(h->auto_mirror + 4 * j, __ATOMIC_ACQUIRE) : -1;
        if (last_accepted) last_accepted[j] = have ? __atomic_load_n(h->auto_mirror + 4 * j + 1, __ATOMIC_RELAXED) : -1;
    }
    return KLARA_OK;
}

// ---- self tests --------------------------------------------------------------------------------
__global__ void k_rocrand_blocks(unsigned long long seed, unsigned long long subseq,
                                 unsigned long long first_block, int nblocks, unsigned int* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblocks) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, subseq, 4ull * (first_block + (unsigned long long)i), &st);
    const uint4 r = rocrand4(&st);
    out[4 * i + 0] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

extern "C" klara_status klara_selftest_rocrand_blocks(int32_t device, uint64_t seed, uint64_t subsequence,
                                                      uint64_t first_block, int32_t nblocks, uint32_t* out)
{
    if (!out || nblocks <= 0) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    unsigned int* d = nullptr;
    HIPCHK(dalloc(&d, (size_t)4 * nblocks));
    hipLaunchKernelGGL(k_rocrand_blocks, dim3((nblocks + 63) / 64), dim3(64), 0, 0, (unsigned long long)seed,
                       (unsigned long long)subsequence, (unsigned long long)first_block, nblocks, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d, sizeof(uint32_t) * 4 * (size_t)nblocks, hipMemcpyDeviceToHost);
    (void)dfree(d);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

__global__ void k_math(int op, long long n, const double* in, const double* in2, double* out)
{
    kd_tables_to_lds();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s, c;
    switch (op) {
    case 0: out[i] = kd_log(in[i]); break;
    case 1: out[i] = kd_exp(in[i]); break;
    case 2: kd_sincos2pi(in[i], &s, &c); out[i] = s; break;
    case 3: kd_sincos2pi(in[i], &s, &c); out[i] = c; break;
    case 4: out[i] = __builtin_sqrt(in[i]); break;
    case 6: out[i] = kd_erf(in[i]); break;
    case 7: out[i] = kd_log_u01(in[i]); break;
    case 8: out[i] = kd_sqrt_radicand(in[i]); break;
    case 9: out[i] = kd_exp_neg(in[i]); break;
    case 10: kd_softplus_logistic(in[i], &s, &c); out[i] = s; break;
    case 11: kd_softplus_logistic(in[i], &s, &c); out[i] = c; break;
    case 12: out[i] = kd_log12(in[i]); break;
    default: out[i] = in[i] / in2[i]; break;
    }
}

extern "C" klara_status klara_selftest_math(int32_t device, int32_t op, int64_t n, const double* in,
                                            const double* in2, double* out)
{
    if (!in || !out || n <= 0) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    double *di = nullptr, *di2 = nullptr, *dout = nullptr;
    hipError_t e = dalloc(&di, (size_t)n);
    if (e == hipSuccess) e = dalloc(&di2, (size_t)n);
    if (e == hipSuccess) e = dalloc(&dout, (size_t)n);
    if (e == hipSuccess) e = hipMemcpy(di, in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(di2, in2 ? in2 : in, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, (long long)n, di, di2, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
    (void)dfree(di); (void)dfree(di2); (void)dfree(dout);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

// |z| exceedance counts and raw power sums of the proposal normals, drawn exactly as the transition kernels draw them
// (kd_normal_pair_w on both halves of kd_stream_block(seed, chain, transition, slot 0)); one thread per chain, one atomic per thread
// and threshold.  counts[k] = #{|z| > thr[k]} over 4 * nchains * ntransitions normals.
__global__ __launch_bounds__(256) void k_normal_tail(unsigned long long seed, unsigned long long first_chain, long long nchains,
                                                     long long ntransitions, int nthr, const double* __restrict__ thr,
                                                     unsigned long long* __restrict__ counts, double* __restrict__ moments)
{
    kd_tables_to_lds();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < nchains;
    unsigned long long c[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    double s1 = 0.0, s2 = 0.0, s4 = 0.0, mx = 0.0;
    for (long long t = 0; t < ntransitions; ++t) {
        const kd_u32x4 b = kd_stream_block(seed, first_chain + (unsigned long long)(ok ? i : 0), (unsigned long long)t, 0u);
        for (int h = 0; h < 2; ++h) {                           // the block's two pairs: pair indices 0 and 8 of a transition
            double z0, z1, u1, lg;
            kd_normal_pair_w(h ? b.z : b.x, h ? b.w : b.y, &z0, &z1, &u1, &lg);
            if (!ok) continue;
            const double a0 = z0 < 0.0 ? -z0 : z0, a1 = z1 < 0.0 ? -z1 : z1;
            for (int k = 0; k < 8; ++k) if (k < nthr) c[k] += (a0 > thr[k] ? 1ull : 0ull) + (a1 > thr[k] ? 1ull : 0ull);
            s1 += z0 + z1; s2 += z0 * z0 + z1 * z1; s4 += (z0 * z0) * (z0 * z0) + (z1 * z1) * (z1 * z1);
            mx = a0 > mx ? a0 : mx; mx = a1 > mx ? a1 : mx;
        }
    }
    for (int k = 0; k < 8; ++k) if (k < nthr && c[k] != 0) atomicAdd(&counts[k], c[k]);
    if (ok) {
        atomicAdd(&moments[0], s1); atomicAdd(&moments[1], s2); atomicAdd(&moments[2], s4);
        atomicMax((unsigned long long*)&moments[3], (unsigned long long)__double_as_longlong(mx));   // (non-negative doubles order like integers)
    }
}

extern "C" klara_status klara_selftest_normal_tail(int32_t device, uint64_t seed, uint64_t first_chain, int64_t nchains,
                                                   int64_t ntransitions, int32_t nthr, const double* thr, uint64_t* counts,
                                                   double* moments)
{
    if (!thr || !counts || nthr <= 0 || nthr > 8 || nchains <= 0 || ntransitions <= 0) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    double* dthr = nullptr; unsigned long long* dc = nullptr; double* dm = nullptr;
    hipError_t e = dalloc(&dthr, 8);
    if (e == hipSuccess) e = dalloc(&dc, 8);
    if (e == hipSuccess) e = dalloc(&dm, 4);
    if (e == hipSuccess) e = hipMemcpy(dthr, thr, sizeof(double) * (size_t)nthr, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dc, 0, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(dm, 0, 4 * sizeof(double));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_normal_tail, dim3((unsigned)((nchains + 255) / 256)), dim3(256), 0, 0, (unsigned long long)seed,
                           (unsigned long long)first_chain, (long long)nchains, (long long)ntransitions, (int)nthr, dthr, dc, dm);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(counts, dc, sizeof(uint64_t) * (size_t)nthr, hipMemcpyDeviceToHost);
    if (e == hipSuccess && moments) e = hipMemcpy(moments, dm, 4 * sizeof(double), hipMemcpyDeviceToHost);
    (void)dfree(dthr); (void)dfree(dc); (void)dfree(dm);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

// the normals of one transition as the samplers draw them: one thread per (chain, element pair)
__global__ __launch_bounds__(256) void k_transition_normals(unsigned long long seed, unsigned long long first_chain, long long nchains,
                                                            unsigned long long t, int D, double* __restrict__ z, double* __restrict__ accept_u)
{
    kd_tables_to_lds();
    const int npairs = (D + 1) / 2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = i < nchains * (long long)(npairs + 1);
    const long long c = ok ? i / (npairs + 1) : 0;
    const int p = ok ? (int)(i % (npairs + 1)) : 0;
    const unsigned long long chain = first_chain + (unsigned long long)c;
    if (p < npairs) {
        double z0, z1, u1, lg;
        kd_normal_pair_at(seed, chain, t, (uint32_t)p, (uint32_t)npairs, &z0, &z1, &u1, &lg);
        if (ok) {
            z[c * D + 2 * p] = z0;
            if (2 * p + 1 < D) z[c * D + 2 * p + 1] = z1;
        }
    } else if (ok && accept_u != nullptr) {
        accept_u[c] = kd_accept_uniform(kd_stream_block(seed, chain, t, (uint32_t)npairs));
    }
}

extern "C" klara_status klara_selftest_transition_normals(int32_t device, uint64_t seed, uint64_t first_chain, int64_t nchains,
                                                          uint64_t transition, int32_t ndims, double* z, double* accept_u)
{
    if (!z || nchains <= 0 || ndims <= 0 || ndims > 4096) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    double* dz = nullptr; double* du = nullptr;
    hipError_t e = dalloc(&dz, (size_t)nchains * (size_t)ndims);
    if (e == hipSuccess) e = dalloc(&du, (size_t)nchains);
    if (e == hipSuccess) {
        const long long nthreads = (long long)nchains * ((ndims + 1) / 2 + 1);
        hipLaunchKernelGGL(k_transition_normals, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, 0, (unsigned long long)seed,
                           (unsigned long long)first_chain, (long long)nchains, (unsigned long long)transition, (int)ndims, dz, du);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(z, dz, sizeof(double) * (size_t)nchains * (size_t)ndims, hipMemcpyDeviceToHost);
    if (e == hipSuccess && accept_u) e = hipMemcpy(accept_u, du, sizeof(double) * (size_t)nchains, hipMemcpyDeviceToHost);
    (void)dfree(dz); (void)dfree(du);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

extern "C" klara_status klara_selftest_mfma_f64(int32_t device, const double* A, const double* B,
                                                const double* C, double* D)
{
    if (!A || !B || !C || !D) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    double* buf = nullptr;
    HIPCHK(dalloc(&buf, 64 + 64 + 256 + 256));
    hipError_t e = hipMemcpy(buf, A, 64 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf + 64, B, 64 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf + 128, C, 256 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = klara_launch_mfma_probe(buf, buf + 64, buf + 128, buf + 384, 0);
    if (e == hipSuccess) e = hipMemcpy(D, buf + 384, 256 * sizeof(double), hipMemcpyDeviceToHost);
    (void)dfree(buf);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

extern "C" klara_status klara_selftest_mfma_f64_4x4x4(int32_t device, const double* A, const double* B,
                                                      const double* C, double* D)
{
    if (!A || !B || !C || !D) return KLARA_ERR_INVALID_ARG;
    HIPCHK(hipSetDevice(device));
    double* buf = nullptr;
    HIPCHK(dalloc(&buf, 256));
    hipError_t e = hipMemcpy(buf, A, 64 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf + 64, B, 64 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf + 128, C, 64 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = klara_launch_mfma4_probe(buf, buf + 64, buf + 128, buf + 192, 0);
    if (e == hipSuccess) e = hipMemcpy(D, buf + 192, 64 * sizeof(double), hipMemcpyDeviceToHost);
    (void)dfree(buf);
    return e == hipSuccess ? KLARA_OK : KLARA_ERR_HIP;
}

extern "C" klara_status klara_check_custom_target(const char* src, int32_t sampler, int32_t ndims)
{
    if (!src || sampler < KLARA_SAMPLER_MH || sampler > KLARA_SAMPLER_SLICE || ndims <= 0) return KLARA_ERR_INVALID_ARG;
    const int modes[1] = { 0 };
    std::string src2;
    if (pair_as_whole(src, sampler, ndims)) { src2 = pair_as_whole_source(src); src = src2.c_str(); }
    if (pair_source(src)) {                                 // pair closure: the plain fused instantiation of its layout
        if (ndims > 2 * 64 * KLARA_DIAGT_NP_MAX) return KLARA_ERR_UNSUPPORTED;
        const int Q = ndims <= 128 ? 8 : (ndims <= 256 ? 16 : (ndims <= 512 ? 32 : 64));
        return klara_jit_create_pair(src, sampler, ndims, (ndims + 2 * Q - 1) / (2 * Q), Q, false, false, false, modes, 1, false, nullptr);
    }
    if (ndims > KLARA_CUSTOM_MAXD) return KLARA_ERR_UNSUPPORTED;
    int G = 1, E = 2, wpb = 4;
    custom_layout(ndims, 0, custom_lik_prior(src), &G, &E, &wpb);
    return klara_jit_create(src, sampler, ndims, E, G, modes, 1, false, nullptr);
}

extern "C" const char* klara_compile_log(void) { return klara_jit_log(); }

extern "C" klara_status klara_selftest_plan(const klara_desc* desc, int32_t nruns, const int64_t* run_lengths, int64_t capacity,
                                            int64_t* k, int64_t* save_col0, int32_t* save_phase0, int32_t* flags, int64_t* nlaunches)
{
    if (!run_lengths || nruns <= 0 || !nlaunches) return KLARA_ERR_INVALID_ARG;
    const klara_status st = validate(desc);
    if (st != KLARA_OK) return st;
    RunCursor cur = { 0, 0, desc->period, 0 };           // tuner_state: totproposed starts at period (samplers.jl:39-45)
    long long n = 0;
    for (int r = 0; r < nruns; ++r) {
        long long remaining = run_lengths[r];
        if (remaining < 0) return KLARA_ERR_INVALID_ARG;
        while (remaining > 0) {
            const PlannedLaunch pl = plan_launch(*desc, cur, remaining);
            if (n < capacity) {
                if (k) k[n] = pl.k;
                if (save_col0) save_col0[n] = pl.save_col0;
                if (save_phase0) save_phase0[n] = pl.save_phase0;
                if (flags) flags[n] = (pl.tune_after ? 1 : 0) | (pl.bm_close_after ? 2 : 0) | (remaining == pl.k ? 4 : 0);
            }
            advance_cursor(*desc, cur, pl);
            remaining -= pl.k; ++n;
        }
    }
    *nlaunches = n;
    return KLARA_OK;
}

extern "C" const char* klara_strerror(klara_status s)
{
    switch (s) {
    case KLARA_OK: return "ok";
    case KLARA_ERR_INVALID_ARG: return "invalid argument";
    case KLARA_ERR_NONFINITE_INIT: return "log-target (or its gradient) not finite at the initial values";
    case KLARA_ERR_HIP: return "HIP runtime error or no device";
    case KLARA_ERR_NOMEM: return "out of device memory";
    case KLARA_ERR_UNSUPPORTED: return "option not supported by this build";
    case KLARA_ERR_STATE: return "call order / missing state";
    case KLARA_ERR_SLICE_STUCK: return "slice sampler shrunk to current position and still not acceptable";
    case KLARA_ERR_COMPILE: return "user-defined target did not compile (see klara_compile_log)";
    default: return "unknown status";
    }
}

extern "C" klara_status klara_selftest_canary(int32_t poke, int64_t* nalloc, int64_t* ncorrupt)
{
    if (!canary_on()) return KLARA_ERR_STATE;
    hipDeviceSynchronize();
    std::lock_guard<std::mutex> g(canary_mu);
    if (poke && !canary_map.empty()) {                  // an off-by-one store right behind the largest live array: the check below must see it
        const CanaryRec* big = nullptr;
        for (auto& kv : canary_map) if (!big || kv.second.bytes > big->bytes) big = &kv.second;
        const double zero = 0.0;
        if (hipMemcpy(big->base + KCANARY + big->bytes, &zero, sizeof(zero), hipMemcpyHostToDevice) != hipSuccess) return KLARA_ERR_HIP;
    }
    long long bad = 0;
    for (auto& kv : canary_map) {
        if (!canary_intact(kv.second)) { ++bad; if (canary_fill(kv.second) != hipSuccess) return KLARA_ERR_HIP; }    // (repaired: reported once)
    }
    if (nalloc) *nalloc = (int64_t)canary_map.size();
    if (ncorrupt) *ncorrupt = bad;
    return KLARA_OK;
}

extern "C" int32_t klara_abi_version(void) { return KLARA_ABI_VERSION; }
