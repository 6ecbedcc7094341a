// jit.cpp — run-time compilation of specialised island kernels (island_spec.inc) with hiprtc.
//
// One kernel per island SHAPE: the text codegen.cpp writes for an island is appended to the node library
// (device.h + island_ops.inc + island_spec.inc, embedded in this library at build time) and compiled for gfx950 with
// the same float rules as the ahead-of-time kernels (-ffp-contract=off). Compilation runs on worker threads; the
// engine keeps rendering through the interpreter kernel until a shape's code object is ready. Code objects are cached
// in memory (per process) and on disk (kcache/ next to the library, or $ELEMHIP_KCACHE), keyed by a hash of the whole
// program text and the compiler version, so a shape is compiled once per machine.
#include "jit.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <thread>
#include <unordered_set>

#include "build/spec_text.inc"   // kSpecDeviceH, kSpecOpsInc, kSpecInc: the sources as raw string literals

namespace elemhip {

static uint64_t fnv1a(const std::string& s, uint64_t h) {
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ull; }
    return h;
}

static std::string libraryDir() {
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&libraryDir), &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t slash = p.rfind('/');
        return slash == std::string::npos ? "." : p.substr(0, slash);
    }
    return ".";
}

struct Jit::Impl {
    std::mutex mu;
    std::condition_variable cv;
    std::unordered_map<std::string, std::shared_ptr<SpecEntry>> entries;   // key -> entry
    std::deque<std::shared_ptr<SpecEntry>> queue;
    std::unordered_map<uint32_t, std::pair<uint64_t, uint64_t>> prefixHash;   // LDS words -> key hash state behind the fixed part of the source
    std::string prefixDefs;                                                 // ... under this ELEMHIP_JIT_DEFINES
    std::unordered_map<std::string, uint32_t> sightings;                    // one-island shapes a background-mode plan left to the interpreter
    std::unordered_set<std::string> notOnDisk;                             // keys the disk cache was asked about in vain
    std::vector<std::thread> workers;
    bool stop = false;
    static Impl* exitHookTarget;
    bool joined = false;
    std::string cacheDir;
    std::string versionTag;

    Impl() {
        exitHookTarget = this;
        const char* env = std::getenv("ELEMHIP_KCACHE");
        cacheDir = env && env[0] ? env : libraryDir() + "/kcache";
        int maj = 0, min = 0;
        (void)hiprtcVersion(&maj, &min);
        versionTag = "hiprtc" + std::to_string(maj) + "." + std::to_string(min) + ";gfx950;-O3;-ffp-contract=off;v1";
        unsigned n = std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 4u));   // a plan of a new graph brings several shapes at once
        if (const char* t = std::getenv("ELEMHIP_JIT_THREADS")) n = (unsigned)std::max(1, std::atoi(t));
        // The compiler library (comgr) is loaded lazily by the first hiprtc compile and registers its static destructors
        // then. Do that first compile here, on the calling thread (~45 ms, once per process), and register the exit hook
        // right after it: the hook then runs BEFORE those destructors whenever the process exits, however early.
        warmUp();
        for (unsigned i = 0; i < n; ++i) workers.emplace_back([this] { run(); });
    }
    ~Impl() { shutdown(); }

    // Stop taking work and wait for the compile in flight. Runs from an atexit hook registered AFTER the compiler library
    // has registered its own static destructors (see the constructor), hence before them: a worker still inside the
    // compiler while its globals are torn down crashes the exiting process.
    void shutdown() {
        {
            std::lock_guard<std::mutex> l(mu);
            if (joined) return;
            joined = true; stop = true; queue.clear();
        }
        cv.notify_all();
        for (auto& t : workers) if (t.joinable()) t.join();
    }

    void warmUp() {
        hiprtcProgram prog = nullptr;
        if (hiprtcCreateProgram(&prog, "extern \"C\" __global__ void elemhip_jit_probe() {}\n", "probe.hip", 0, nullptr, nullptr) == HIPRTC_SUCCESS) {
            const char* opts[] = {"--offload-arch=gfx950"};
            (void)hiprtcCompileProgram(prog, 1, opts);
            (void)hiprtcDestroyProgram(&prog);
        }
        std::atexit([] { if (Impl* i = exitHookTarget) i->shutdown(); });
    }

    void run() {
        for (;;) {
            std::shared_ptr<SpecEntry> e;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return stop || !queue.empty(); });
                if (stop) return;
                e = queue.front(); queue.pop_front();
            }
            compile(*e);
            { std::lock_guard<std::mutex> l(mu); }
            cv.notify_all();
        }
    }

    void compile(SpecEntry& e) {
        const std::string path = cacheDir + "/" + e.key + ".hsaco";
        {   // disk cache
            std::ifstream f(path, std::ios::binary);
            if (f) {
                std::vector<char> code((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
                if (code.size() > 64) { e.code.swap(code); e.fromDisk = true; e.state.store(1, std::memory_order_release); return; }
            }
        }
        hiprtcProgram prog = nullptr;
        if (hiprtcCreateProgram(&prog, e.source.c_str(), "elemhip_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
            e.log = "hiprtcCreateProgram failed"; e.state.store(-1, std::memory_order_release); return;
        }
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-pragma-once-outside-header"};
        const hiprtcResult rc = hiprtcCompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
        size_t logSize = 0;
        (void)hiprtcGetProgramLogSize(prog, &logSize);
        if (logSize > 1) { e.log.resize(logSize); (void)hiprtcGetProgramLog(prog, &e.log[0]); }
        if (rc != HIPRTC_SUCCESS) {
            std::fprintf(stderr, "[elemhip] jit: compilation of shape %s failed:\n%.3000s\n", e.key.c_str(), e.log.c_str());
            (void)hiprtcDestroyProgram(&prog);
            e.state.store(-1, std::memory_order_release);
            return;
        }
        size_t sz = 0;
        (void)hiprtcGetCodeSize(prog, &sz);
        e.code.resize(sz);
        (void)hiprtcGetCode(prog, e.code.data());
        (void)hiprtcDestroyProgram(&prog);
        // disk cache: best effort, atomic rename
        (void)mkdir(cacheDir.c_str(), 0755);
        const std::string tmp = path + "." + std::to_string((long)getpid()) + ".tmp";
        {
            std::ofstream f(tmp, std::ios::binary);
            if (f) { f.write(e.code.data(), (std::streamsize)e.code.size()); f.close(); if (std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str()); }
        }
        e.state.store(1, std::memory_order_release);
    }
};

Jit::Impl* Jit::Impl::exitHookTarget = nullptr;

Jit& Jit::get() { static Jit* j = new Jit; return *j; }   // never destroyed: the exit hook (warmUp) stops the workers
Jit::Jit() : impl(new Impl) {}
Jit::~Jit() { delete impl; }
void Jit::shutdownAtExit() { impl->shutdown(); }

// everything in front of the generated text (a function of the LDS size and of ELEMHIP_JIT_DEFINES)
static std::string sourcePrefix(uint32_t ldsWords, size_t reserveExtra) {
    std::string s;
    s.reserve(sizeof(kSpecDeviceH) + sizeof(kSpecOpsInc) + sizeof(kSpecInc) + reserveExtra + 256);
    s += "#define ELEMHIP_SPEC 1\n#define ELEMHIP_SPEC_LDS_WORDS " + std::to_string(ldsWords) + "\n";
    if (const char* d = std::getenv("ELEMHIP_JIT_DEFINES")) {   // tuning experiments: "NAME=VALUE NAME2=VALUE2" -> #define lines (part of the cache key)
        std::string t = d, tok;
        for (size_t i = 0; i <= t.size(); ++i) {
            if (i == t.size() || t[i] == ' ') {
                if (!tok.empty()) { const size_t eq = tok.find('='); s += "#define " + (eq == std::string::npos ? tok + " 1" : tok.substr(0, eq) + " " + tok.substr(eq + 1)) + "\n"; }
                tok.clear();
            } else tok += t[i];
        }
    }
    // hiprtc has no host headers: the fixed-width names the sources use (same underlying types as <stdint.h> on this target)
    s += "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long uint64_t;\n"
         "typedef signed char int8_t; typedef short int16_t; typedef int int32_t; typedef long int64_t; typedef unsigned long uintptr_t;\n";
    s += kSpecDeviceH; s += "\n"; s += kSpecOpsInc; s += "\n"; s += kSpecInc; s += "\n";
    return s;
}

std::string Jit::fullSource(const std::string& generated, uint32_t ldsWords) {
    std::string s = sourcePrefix(ldsWords, generated.size());
    s += generated;
    return s;
}

// The key is a hash of the whole translation unit; its first ~300 KB are the same for every shape of one LDS size, so the hash
// state behind them is kept (FNV-1a runs front to back: same keys as hashing the full source, a tenth of the time).
std::string Jit::keyFor(const std::string& generated, uint32_t ldsWords) {
    const char* d = std::getenv("ELEMHIP_JIT_DEFINES");
    const std::string defs = d ? d : "";
    uint64_t h1 = 0, h2 = 0;
    bool have = false;
    {
        std::lock_guard<std::mutex> l(impl->mu);
        if (impl->prefixDefs == defs) { auto it = impl->prefixHash.find(ldsWords); if (it != impl->prefixHash.end()) { h1 = it->second.first; h2 = it->second.second; have = true; } }
    }
    if (!have) {
        const std::string pre = sourcePrefix(ldsWords, 0);
        h1 = fnv1a(pre, fnv1a(impl->versionTag, 1469598103934665603ull));
        h2 = fnv1a(pre, fnv1a(impl->versionTag, 0x9E3779B97F4A7C15ull) ^ 0xA5A5A5A5ull);
        std::lock_guard<std::mutex> l(impl->mu);
        if (impl->prefixDefs != defs) { impl->prefixHash.clear(); impl->prefixDefs = defs; }
        impl->prefixHash[ldsWords] = {h1, h2};
    }
    h1 = fnv1a(generated, h1); h2 = fnv1a(generated, h2);
    char key[40];
    std::snprintf(key, sizeof key, "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    return key;
}

bool Jit::knownKey(const std::string& key) {
    {
        std::lock_guard<std::mutex> l(impl->mu);
        if (impl->entries.count(key)) return true;
        if (impl->notOnDisk.count(key)) return false;
    }
    // (asked on every re-plan of a live graph for the one-island shapes of a voice that is fading out: the answer from the
    //  file system — 100 us and more on a network mount — is remembered; a key that gets compiled later is found in `entries`)
    struct stat st;
    const bool there = ::stat((impl->cacheDir + "/" + key + ".hsaco").c_str(), &st) == 0;
    if (!there) { std::lock_guard<std::mutex> l(impl->mu); if (impl->notOnDisk.size() > 4096) impl->notOnDisk.clear(); impl->notOnDisk.insert(key); }
    return there;
}
uint32_t Jit::sighting(const std::string& key) {
    std::lock_guard<std::mutex> l(impl->mu);
    if (impl->sightings.size() > 4096) impl->sightings.clear();
    return ++impl->sightings[key];
}
bool Jit::known(const std::string& generated, uint32_t ldsWords) { return knownKey(keyFor(generated, ldsWords)); }

std::shared_ptr<SpecEntry> Jit::requestKey(const std::string& key, const std::string& generated, uint32_t ldsWords) {
    {
        std::lock_guard<std::mutex> l(impl->mu);
        auto it = impl->entries.find(key);
        if (it != impl->entries.end()) return it->second;
    }
    std::string src = fullSource(generated, ldsWords);
    std::lock_guard<std::mutex> l(impl->mu);
    auto it = impl->entries.find(key);
    if (it != impl->entries.end()) return it->second;
    auto e = std::make_shared<SpecEntry>();
    e->key = key; e->source.swap(src); e->ldsBytes = ldsWords * 4u;
    impl->entries.emplace(e->key, e);
    impl->queue.push_back(e);
    impl->cv.notify_one();
    return e;
}
std::shared_ptr<SpecEntry> Jit::request(const std::string& generated, uint32_t ldsWords) { return requestKey(keyFor(generated, ldsWords), generated, ldsWords); }

int Jit::wait(const std::shared_ptr<SpecEntry>& e) {
    std::unique_lock<std::mutex> l(impl->mu);
    impl->cv.wait(l, [&] { return impl->stop || e->state.load(std::memory_order_acquire) != 0; });
    return e->state.load(std::memory_order_acquire);
}

// engine thread, device current: load the code object on this device (once)
hipFunction_t SpecEntry::function(int device) {
    if (state.load(std::memory_order_acquire) != 1) return nullptr;
    std::lock_guard<std::mutex> l(mu);
    auto it = perDevice.find(device);
    if (it != perDevice.end()) return it->second.second;
    hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess || hipModuleGetFunction(&fn, mod, "elemhip_spec_island") != hipSuccess) {
        std::fprintf(stderr, "[elemhip] jit: loading shape %s failed: %s\n", key.c_str(), hipGetErrorString(hipGetLastError()));
        perDevice.emplace(device, std::make_pair(mod, (hipFunction_t) nullptr));
        return nullptr;
    }
    perDevice.emplace(device, std::make_pair(mod, fn));
    return fn;
}

} // namespace elemhip
