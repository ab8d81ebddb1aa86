// klara_jit.hip — run-time compilation of user-defined targets (KLARA_TARGET_CUSTOM) for gfx950.
//
// The reference's targets are arbitrary Julia closures (BasicContMuvParameter.jl:174-201,264-279).  Their device form
// here is source text: klara_create hands it to hiprtc together with the transition kernels of klara_kernels.h (embedded
// in this library at build time), instantiates k_init / k_transitions for the job's sampler and element count only, and
// loads the code object as a HIP module.  Compilation needs no GPU (klara_check_custom_target); code objects are cached
// per process by (source, sampler, D, kernel modes) and on disk ($KLARA_JIT_CACHE_DIR, else $XDG_CACHE_HOME/klara_hip, else
// ~/.cache/klara_hip; KLARA_JIT_CACHE=0 turns the disk cache off), keyed additionally by the embedded kernel headers so that a
// rebuilt library never picks up a stale code object.  libhiprtc.so is loaded on first use.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include "klara_launch.h"
#include "klara_jit_headers.inc"

namespace {

struct Rtc {
    void* dl = nullptr;
    hiprtcResult (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    hiprtcResult (*AddNameExpression)(hiprtcProgram, const char*) = nullptr;
    hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char* const*) = nullptr;
    hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*GetLoweredName)(hiprtcProgram, const char*, const char**) = nullptr;
    hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
    hiprtcResult (*GetCode)(hiprtcProgram, char*) = nullptr;
    hiprtcResult (*DestroyProgram)(hiprtcProgram*) = nullptr;
    bool ok = false;
};

Rtc* rtc()
{
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = { "libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so" };
        for (const char* n : names) { r.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.dl) break; }
        if (!r.dl) return;
#define SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.dl, "hiprtc" #f))
        SYM(CreateProgram); SYM(AddNameExpression); SYM(CompileProgram); SYM(GetProgramLogSize); SYM(GetProgramLog);
        SYM(GetLoweredName); SYM(GetCodeSize); SYM(GetCode); SYM(DestroyProgram);
#undef SYM
        r.ok = r.CreateProgram && r.AddNameExpression && r.CompileProgram && r.GetProgramLogSize && r.GetProgramLog &&
               r.GetLoweredName && r.GetCodeSize && r.GetCode && r.DestroyProgram;
    });
    return &r;
}

struct CodeObject {
    std::vector<char> code;
    std::string init_name;
    std::map<int, std::string> trans_names;      // kernel mode -> lowered name
};

std::mutex g_cache_mutex;
std::map<std::string, CodeObject> g_cache;
thread_local std::string g_log;

// ---- disk cache ------------------------------------------------------------------------------------------------------
uint64_t fnv1a(const void* data, size_t n, uint64_t h)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

std::string cache_dir()
{
    if (const char* off = getenv("KLARA_JIT_CACHE")) { if (off[0] == '0') return ""; }
    std::string d;
    if (const char* e = getenv("KLARA_JIT_CACHE_DIR")) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/klara_hip";
    else if (const char* h = getenv("HOME")) { d = std::string(h) + "/.cache"; mkdir(d.c_str(), 0755); d += "/klara_hip"; }
    if (d.empty()) return d;
    mkdir(d.c_str(), 0700);                          // (EEXIST is fine; an unusable directory just disables the cache)
    // Cached files are code objects that get loaded and executed: only a directory that belongs to this user and that nobody
    // else can write to is trusted (a shared or world-writable KLARA_JIT_CACHE_DIR such as /tmp would let another user plant
    // a .kjit file under a predictable name).  Anything else disables the disk cache, never the job.
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) return "";
    return d;
}

// file name of a (key, options) combination: two 64-bit FNV-1a hashes over the key, the compile options and every embedded header
std::string cache_file(const std::string& key, const std::string& opts)
{
    const std::string dir = cache_dir();
    if (dir.empty()) return "";
    uint64_t h1 = 0xcbf29ce484222325ull, h2 = 0x84222325cbf29ce4ull;
    const auto mix = [&](const void* p, size_t n) { h1 = fnv1a(p, n, h1); h2 = fnv1a(p, n, h2 ^ 0x9e3779b97f4a7c15ull); };
    mix(key.data(), key.size()); mix(opts.data(), opts.size());
    for (int i = 0; i < klara_jit_nheaders; ++i) mix(klara_jit_header_sources[i], strlen(klara_jit_header_sources[i]));
    char name[64];
    snprintf(name, sizeof name, "/%016llx%016llx.kjit", (unsigned long long)h1, (unsigned long long)h2);
    return dir + name;
}

// layout: "KJIT1\n", u32 count, count x (i32 mode (-1: init kernel), u32 length, name bytes), u64 code size, code, u64 FNV-1a of all before
bool cache_read(const std::string& path, CodeObject& co)
{
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);       // (no symlinks; regular files of this user only)
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH)) != 0) { close(fd); return false; }
    FILE* f = fdopen(fd, "rb");
    if (!f) { close(fd); return false; }
    std::vector<char> buf;
    char chunk[65536];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    fclose(f);
    if (buf.size() < 6 + 4 + 8 + 8 || memcmp(buf.data(), "KJIT1\n", 6) != 0) return false;
    uint64_t sum;
    memcpy(&sum, buf.data() + buf.size() - 8, 8);
    if (sum != fnv1a(buf.data(), buf.size() - 8, 0xcbf29ce484222325ull)) return false;
    size_t o = 6;
    const auto take = [&](void* dst, size_t k) { if (o + k > buf.size() - 8) return false; memcpy(dst, buf.data() + o, k); o += k; return true; };
    uint32_t cnt;
    if (!take(&cnt, 4) || cnt > 16) return false;
    for (uint32_t i = 0; i < cnt; ++i) {
        int32_t mode; uint32_t len;
        if (!take(&mode, 4) || !take(&len, 4) || len > 4096 || o + len > buf.size() - 8) return false;
        std::string nm(buf.data() + o, len); o += len;
        if (mode < 0) co.init_name = nm; else co.trans_names[mode] = nm;
    }
    uint64_t cs;
    if (!take(&cs, 8) || o + cs != buf.size() - 8) return false;
    co.code.assign(buf.begin() + (long)o, buf.begin() + (long)(o + cs));
    return !co.init_name.empty() && !co.code.empty();
}

void cache_write(const std::string& path, const CodeObject& co)
{
    std::vector<char> buf;
    const auto put = [&](const void* p, size_t k) { const char* c = static_cast<const char*>(p); buf.insert(buf.end(), c, c + k); };
    put("KJIT1\n", 6);
    const uint32_t cnt = 1 + (uint32_t)co.trans_names.size();
    put(&cnt, 4);
    const auto put_name = [&](int32_t mode, const std::string& nm) { const uint32_t len = (uint32_t)nm.size(); put(&mode, 4); put(&len, 4); put(nm.data(), len); };
    put_name(-1, co.init_name);
    for (const auto& kv : co.trans_names) put_name(kv.first, kv.second);
    const uint64_t cs = co.code.size();
    put(&cs, 8); put(co.code.data(), co.code.size());
    const uint64_t sum = fnv1a(buf.data(), buf.size(), 0xcbf29ce484222325ull);
    put(&sum, 8);
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    FILE* f = fdopen(fd, "wb");
    if (!f) { close(fd); remove(tmp.c_str()); return; }
    const bool ok = fwrite(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());     // (atomic publish; a concurrent writer produces the same bytes)
}

#ifndef KLARA_JIT_UNROLL_MAX_E
#define KLARA_JIT_UNROLL_MAX_E 128   /* measured, MALA on 65,536 chains: E = 128 unrolled 2.3e8 transitions/s (11 s to compile) against 1.7e8 as loops (0.4 s); E = 256 unrolled 9.1e7 (32 s) against 8.1e7 (0.3 s); KLARA_JIT_UNROLL_MAX_E in the environment overrides */
#endif
// pair closures (klara_diagt.h USERPAIR) on the pair-transposed layout: NP pairs per lane, Q lanes per chain; `mode` here is
// bit 0 = one transition per launch (ONESTEP), the job's monitor / tuner flags are fixed per handle
struct PairForm { int NP = 0, Q = 0; bool mon = false, tune = false, da = false; };
std::string pair_trans_expr(int sampler, const PairForm& f, int mode)
{
    char b[160];
    const bool onestep = (mode & 1) != 0;
    snprintf(b, sizeof b, "k_diagt<%d, %d, %d, %s, true, %s, %s, %s, true>", sampler, f.NP, f.Q, onestep ? "true" : "false",
             f.mon ? "true" : "false", f.tune ? "true" : "false", f.da ? "true" : "false");
    return b;
}
std::string pair_init_expr(const PairForm& f)
{
    char b[96];
    snprintf(b, sizeof b, "k_diagt_init<%d, %d, true>", f.NP, f.Q);
    return b;
}
std::string trans_expr(int sampler, int E, int G, int mode)
{
    char b[128];
    snprintf(b, sizeof b, "k_transitions<%d, KLARA_TARGET_CUSTOM, %d, %d, %d>", sampler, E, G, mode);
    return b;
}
std::string init_expr(int E, int G)
{
    char b[96];
    snprintf(b, sizeof b, "k_init<KLARA_TARGET_CUSTOM, %d, %d>", E, G);
    return b;
}

// compile (or fetch) the code object of one (source, sampler, D, modes) combination
klara_status compile(const char* src, int sampler, int D, int E, int G, const int* modes, int nmodes, const CodeObject** out, const PairForm* pf = nullptr)
{
    g_log.clear();
    Rtc* r = rtc();
    if (!r->ok) { g_log = "libhiprtc.so could not be loaded"; return KLARA_ERR_UNSUPPORTED; }
    std::string key = std::to_string(sampler) + "/" + std::to_string(D) + "/" + std::to_string(E) + "x" + std::to_string(G) + "/";
    if (pf) key += "pair/" + std::to_string(pf->NP) + "/" + std::to_string(pf->Q) + "/" + std::to_string((int)pf->mon) + std::to_string((int)pf->tune) + std::to_string((int)pf->da) + "/";
    for (int i = 0; i < nmodes; ++i) key += std::to_string(modes[i]) + ",";
    key += (getenv("KLARA_JIT_UNROLL_MAX_E") ? getenv("KLARA_JIT_UNROLL_MAX_E") : ""); key += "\n"; key += src;
    {
        std::lock_guard<std::mutex> lk(g_cache_mutex);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) { *out = &it->second; return KLARA_OK; }
    }
    // same arithmetic contract as the ahead-of-time kernels: no contraction of a*b+c
    const char* opts[] = { "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value",
                           "-Wno-unused-variable", "-Wno-unused-function" };
    std::string optstr;
    for (const char* o : opts) { optstr += o; optstr += ' '; }
    const std::string disk = cache_file(key, optstr);
    if (!disk.empty()) {
        CodeObject cached;
        if (cache_read(disk, cached)) {
            bool complete = true;
            for (int i = 0; i < nmodes; ++i) complete = complete && cached.trans_names.count(modes[i]) == 1;
            if (complete) {
                std::lock_guard<std::mutex> lk(g_cache_mutex);
                auto ins = g_cache.emplace(key, std::move(cached));
                *out = &ins.first->second;
                return KLARA_OK;
            }
        }
    }
    const bool needgrad = sampler == KLARA_SAMPLER_MALA || sampler == KLARA_SAMPLER_HMC;
    std::string tu;
    tu += "#define KLARA_D " + std::to_string(D) + "\n";
    if (!needgrad) tu += "#define KLARA_CUSTOM_NOGRAD 1\n";
    int unroll_max = KLARA_JIT_UNROLL_MAX_E;
    if (const char* s = getenv("KLARA_JIT_UNROLL_MAX_E")) unroll_max = atoi(s);
    const bool loops = E > unroll_max;                    // element loops over scratch-resident arrays (klara_kernels.h)
    if (loops) tu += "#define KLARA_PRAGMA_UNROLL_E _Pragma(\"nounroll\")\n";
    // (the run-time compiler has no <stdint.h>; its own fixed-width types live in a private namespace)
    tu += "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
          "typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;\n";
    if (pf) tu += "#ifndef KLARA_USER_PAIR_TARGET\n#define KLARA_USER_PAIR_TARGET 1\n#endif\n";
    tu += "#include \"klara_kernels.h\"\n"
          "#define KLARA_USER_FN static __device__ __forceinline__\n"
          "#line 1 \"klara_user_target\"\n";
    tu += src;
    // the glue: whole-vector closures instantiate the group-layout kernels on CustomTarget (klara_custom.h); a pair closure is
    // declared by now, so klara_diagt.h's USERPAIR branches can call it
    tu += pf ? "\n#line 1 \"klara_custom_pair_glue\"\n#include \"klara_diagt.h\"\n" : "\n#line 1 \"klara_custom_glue\"\n#include \"klara_custom.h\"\n";

    hiprtcProgram prog;
    if (r->CreateProgram(&prog, tu.c_str(), "klara_custom_target.hip", klara_jit_nheaders, klara_jit_header_sources,
                         klara_jit_header_names) != HIPRTC_SUCCESS) {
        g_log = "hiprtcCreateProgram failed";
        return KLARA_ERR_COMPILE;
    }
    const std::string ie = pf ? pair_init_expr(*pf) : init_expr(E, G);
    r->AddNameExpression(prog, ie.c_str());
    std::vector<std::string> te;
    for (int i = 0; i < nmodes; ++i) { te.push_back(pf ? pair_trans_expr(sampler, *pf, modes[i]) : trans_expr(sampler, E, G, modes[i])); r->AddNameExpression(prog, te.back().c_str()); }
    const hiprtcResult cr = r->CompileProgram(prog, (int)(sizeof opts / sizeof *opts), opts);
    size_t ls = 0;
    if (r->GetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) { g_log.resize(ls); r->GetProgramLog(prog, &g_log[0]); }
    if (cr != HIPRTC_SUCCESS) { r->DestroyProgram(&prog); if (g_log.empty()) g_log = "compilation failed"; return KLARA_ERR_COMPILE; }
    CodeObject co;
    const char* low = nullptr;
    bool ok = r->GetLoweredName(prog, ie.c_str(), &low) == HIPRTC_SUCCESS && low;
    if (ok) co.init_name = low;
    for (int i = 0; ok && i < nmodes; ++i) {
        ok = r->GetLoweredName(prog, te[i].c_str(), &low) == HIPRTC_SUCCESS && low;
        if (ok) co.trans_names[modes[i]] = low;
    }
    size_t cs = 0;
    ok = ok && r->GetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs > 0;
    if (ok) { co.code.resize(cs); ok = r->GetCode(prog, co.code.data()) == HIPRTC_SUCCESS; }
    r->DestroyProgram(&prog);
    if (!ok) { g_log += "\n(could not retrieve the code object)"; return KLARA_ERR_COMPILE; }
    if (!disk.empty()) cache_write(disk, co);
    std::lock_guard<std::mutex> lk(g_cache_mutex);
    auto ins = g_cache.emplace(key, std::move(co));
    *out = &ins.first->second;
    return KLARA_OK;
}

}  // namespace

struct KlaraJit {
    hipModule_t mod = nullptr;
    hipFunction_t init = nullptr;
    std::map<int, hipFunction_t> trans;
};

klara_status klara_jit_create_pair(const char* src, int sampler, int D, int NP, int Q, bool mon, bool tune, bool da, const int* modes, int nmodes,
                                   bool load, KlaraJit** out)
{
    PairForm pf; pf.NP = NP; pf.Q = Q; pf.mon = mon; pf.tune = tune; pf.da = da;
    const CodeObject* co = nullptr;
    klara_status st = compile(src, sampler, D, 2 * NP, Q, modes, nmodes, &co, &pf);
    if (st != KLARA_OK || !load) return st;
    KlaraJit* j = new (std::nothrow) KlaraJit();
    if (!j) return KLARA_ERR_NOMEM;
    bool ok = hipModuleLoadData(&j->mod, co->code.data()) == hipSuccess;
    ok = ok && hipModuleGetFunction(&j->init, j->mod, co->init_name.c_str()) == hipSuccess;
    for (auto it = co->trans_names.begin(); ok && it != co->trans_names.end(); ++it) {
        hipFunction_t f = nullptr;
        ok = hipModuleGetFunction(&f, j->mod, it->second.c_str()) == hipSuccess;
        j->trans[it->first] = f;
    }
    if (!ok) { klara_jit_destroy(j); return KLARA_ERR_HIP; }
    *out = j;
    return KLARA_OK;
}

klara_status klara_jit_create(const char* src, int sampler, int D, int E, int G, const int* modes, int nmodes, bool load, KlaraJit** out)
{
    const CodeObject* co = nullptr;
    klara_status st = compile(src, sampler, D, E, G, modes, nmodes, &co);
    if (st != KLARA_OK || !load) return st;
    KlaraJit* j = new (std::nothrow) KlaraJit();
    if (!j) return KLARA_ERR_NOMEM;
    bool ok = hipModuleLoadData(&j->mod, co->code.data()) == hipSuccess;
    ok = ok && hipModuleGetFunction(&j->init, j->mod, co->init_name.c_str()) == hipSuccess;
    for (auto it = co->trans_names.begin(); ok && it != co->trans_names.end(); ++it) {
        hipFunction_t f = nullptr;
        ok = hipModuleGetFunction(&f, j->mod, it->second.c_str()) == hipSuccess;
        j->trans[it->first] = f;
    }
    if (!ok) { klara_jit_destroy(j); return KLARA_ERR_HIP; }
    *out = j;
    return KLARA_OK;
}

void klara_jit_destroy(KlaraJit* j)
{
    if (!j) return;
    if (j->mod) hipModuleUnload(j->mod);
    delete j;
}

hipError_t klara_jit_launch_init(KlaraJit* j, const KParams& p, int needgrad, dim3 grid, size_t lds, hipStream_t st, int block)
{
    KParams pv = p;
    void* args[] = { &pv, &needgrad };
    return hipModuleLaunchKernel(j->init, grid.x, 1, 1, (unsigned)block, 1, 1, (unsigned)lds, st, args, nullptr);
}

hipError_t klara_jit_launch(KlaraJit* j, int mode, const KParams* p, const KLaunch& kl, dim3 grid, size_t lds, hipStream_t st, int block)
{
    auto it = j->trans.find(mode == 7 ? 7 : (mode & 3) == 3 ? 3 : (mode & 1) ? 1 : 0);
    if (it == j->trans.end()) return hipErrorInvalidValue;
    if (klara_attr_query != nullptr) {               // klara_get_kernel_attributes: report instead of launching
        int regs = 0, scratch = 0, lds = 0;
        hipError_t e = hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, it->second);
        if (e == hipSuccess) e = hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, it->second);
        if (e == hipSuccess) e = hipFuncGetAttribute(&lds, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, it->second);
        klara_attr_query->numRegs = regs; klara_attr_query->localSizeBytes = (size_t)scratch; klara_attr_query->sharedSizeBytes = (size_t)lds;
        return e;
    }
    KLaunch klv = kl;
    void* args[] = { &p, &klv };
    return hipModuleLaunchKernel(it->second, grid.x, 1, 1, (unsigned)block, 1, 1, (unsigned)lds, st, args, nullptr);
}

// the pair-closure kernels take (KParams*, KLaunch, KAuto) like every k_diagt instantiation; one wavefront per chain group
hipError_t klara_jit_launch_pair(KlaraJit* j, int mode, const KParams* p, const KLaunch& kl, long long nwaves, hipStream_t st)
{
    auto it = j->trans.find(mode);
    if (it == j->trans.end()) return hipErrorInvalidValue;
    if (klara_attr_query != nullptr) {
        int regs = 0, scratch = 0, lds = 0;
        hipError_t e = hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, it->second);
        if (e == hipSuccess) e = hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, it->second);
        if (e == hipSuccess) e = hipFuncGetAttribute(&lds, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, it->second);
        klara_attr_query->numRegs = regs; klara_attr_query->localSizeBytes = (size_t)scratch; klara_attr_query->sharedSizeBytes = (size_t)lds;
        return e;
    }
    KLaunch klv = kl;
    KAuto ka = KLARA_AUTO_NONE;
    void* args[] = { &p, &klv, &ka };
    return hipModuleLaunchKernel(it->second, (unsigned)((nwaves + 3) / 4), 1, 1, 256, 1, 1, 0, st, args, nullptr);
}

const char* klara_jit_log() { return g_log.c_str(); }
