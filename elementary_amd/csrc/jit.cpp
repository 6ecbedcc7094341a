// jit.cpp — run-time compilation of specialised island kernels (island_spec.inc) with hiprtc.
//
// One kernel per island SHAPE: the text codegen.cpp writes for an island is appended to the node library
// (device.h + island_ops.inc + island_spec.inc, embedded in this library at build time) and compiled for gfx950 with
// the same float rules as the ahead-of-time kernels (-ffp-contract=off). Compilation runs on worker threads; the
// engine keeps rendering through the interpreter kernel until a shape's code object is ready. Code objects are cached
// in memory (per process) and on disk (kcache/ next to the library, or $ELEMHIP_KCACHE), keyed by a hash of the whole
// program text and the compiler version, so a shape is compiled once per machine.
//
// Everything here is BOUNDED (r05; a live-coding session keeps producing shapes it never meets again):
//   * an entry keeps its code object and the few KB of generated text, not the 300 KB translation unit (dropped after the compile);
//   * the in-memory table is capped (`jit_cache_entries`, default 256): entries no plan references any more are evicted oldest
//     first and their modules unloaded (hipModuleUnload) — plans hold their shapes' entries, and a plan dies only after the
//     synchronise that follows its last launch, so an unreferenced entry has no kernel in flight;
//   * the on-disk cache is capped ($ELEMHIP_KCACHE_MAX_MB, default 512): oldest files (mtime; a disk hit refreshes it) go first;
//   * the compile queue drops requests nobody waits for any more (the voice was replaced before a worker got to its shape) and
//     serves the newest request first when it is backed up; one-off shapes of background-mode plans enter it only once their plan
//     has rendered for a while (Jit::promote), behind everything else.
#include "jit.h"

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/wait.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cerrno>
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

// ---- the on-disk cache holds code this process will RUN: only a directory that is ours is read from or written to ----
// The directory must be a real directory (no symlink) owned by this user (or root) that nobody else may write to; it is created 0700.
static bool trustedCacheDir(const std::string& dir) {
    (void)mkdir(dir.c_str(), 0700);
    struct stat sb;
    if (::lstat(dir.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) return false;
    if (sb.st_uid != geteuid() && sb.st_uid != 0) return false;
    return (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
// a gfx code object as hiprtc hands it out: a 64-bit little-endian ELF for EM_AMDGPU (224), or a clang offload bundle of such
static bool looksLikeCodeObject(const std::vector<char>& c) {
    static const char kBundle[] = "__CLANG_OFFLOAD_BUNDLE__";
    if (c.size() >= sizeof(kBundle) - 1 + 8 && std::memcmp(c.data(), kBundle, sizeof(kBundle) - 1) == 0) return true;
    if (c.size() < 64) return false;
    const unsigned char* u = reinterpret_cast<const unsigned char*>(c.data());
    return u[0] == 0x7f && u[1] == 'E' && u[2] == 'L' && u[3] == 'F' && u[4] == 2 /* ELFCLASS64 */ && u[5] == 1 /* little endian */
        && (unsigned)(u[18] | (u[19] << 8)) == 224u;
}
// a cached code object: a regular file (never through a symlink) owned by this user or root
static bool readTrustedFile(const std::string& path, std::vector<char>& out) {
    const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat sb;
    bool ok = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && (sb.st_uid == geteuid() || sb.st_uid == 0) && sb.st_size > 64 && sb.st_size < (64 << 20);
    if (ok) {
        out.resize((size_t)sb.st_size);
        size_t got = 0;
        while (got < out.size()) { const ssize_t r = ::read(fd, out.data() + got, out.size() - got); if (r <= 0) break; got += (size_t)r; }
        ok = got == out.size();
    }
    ::close(fd);
    return ok && looksLikeCodeObject(out);
}
static bool writeNewFile(const std::string& path, const char* data, size_t n) {      // O_EXCL | O_NOFOLLOW: never onto something pre-placed
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return false;
    size_t put = 0;
    while (put < n) { const ssize_t r = ::write(fd, data + put, n - put); if (r <= 0) break; put += (size_t)r; }
    return ::close(fd) == 0 && put == n;
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

// tuning experiments ("NAME=VALUE NAME2=VALUE2" -> #define lines in front of the node library, part of the cache key) exist only
// in a `make EXPERIMENTAL=1` build: several of the hooks they reach render wrong samples by design (measurement only)
// whitelisted tunings (Jit::setTuning): "#define NAME VALUE" lines in front of the node library
static std::mutex gTuningMu;
static std::string gTuning;
static std::string tuningDefines() { std::lock_guard<std::mutex> l(gTuningMu); return gTuning; }

static std::string experimentalDefines() {
#ifdef ELEMHIP_EXPERIMENTAL
    const char* d = std::getenv("ELEMHIP_JIT_DEFINES");
    return d ? d : "";
#else
    return "";
#endif
}

struct Jit::Impl {
    std::mutex mu;
    std::condition_variable cv;
    std::unordered_map<std::string, std::shared_ptr<SpecEntry>> entries;   // key -> entry
    std::deque<std::shared_ptr<SpecEntry>> queue;                          // wanted by a plan now (newest served first when backed up)
    std::deque<std::shared_ptr<SpecEntry>> lowQueue;                       // promoted one-off shapes
    std::unordered_map<uint64_t, std::pair<uint64_t, uint64_t>> prefixHash; // (LDS words, block) -> key hash state behind the fixed part of the source
    std::string prefixDefs;                                                 // ... under these experimental defines
    std::unordered_map<std::string, uint32_t> sightings;                    // one-island shapes a background-mode plan left to the interpreter
    std::unordered_set<std::string> notOnDisk;                             // keys the disk cache was asked about in vain
    std::vector<std::thread> workers;
    bool stop = false;
    static Impl* exitHookTarget;
    bool joined = false;
    std::string cacheDir;
    std::string versionTag;
    bool keepSource = false;
    std::string helper;                     // elemhip_jitc next to the library: one compiler process per worker ("" = compile in-process)
    bool diskOk = true;                     // the on-disk cache is ours to trust (trustedCacheDir); false: compile, never read or publish files
    uint32_t entryCap = 256;
    uint64_t diskCapBytes = 512ull << 20;
    int64_t diskBytes = -1;                 // -1: not scanned yet
    std::atomic<uint64_t> tick{1};
    JitStats st;
    std::atomic<int64_t> modulesLoaded{0};

    Impl() {
        exitHookTarget = this;
        const char* env = std::getenv("ELEMHIP_KCACHE");
        cacheDir = env && env[0] ? env : libraryDir() + "/kcache";
        diskOk = trustedCacheDir(cacheDir);
        if (!diskOk) std::fprintf(stderr, "[elemhip] jit: %s is not a directory of this user that only they can write to: the on-disk kernel cache is off\n", cacheDir.c_str());
        if (const char* k = std::getenv("ELEMHIP_JIT_KEEP_SOURCE")) keepSource = std::atoi(k) != 0;
        if (const char* k = std::getenv("ELEMHIP_JIT_CACHE_ENTRIES")) entryCap = (uint32_t)std::max(4, std::atoi(k));
        if (const char* k = std::getenv("ELEMHIP_KCACHE_MAX_MB")) diskCapBytes = (uint64_t)std::max(1, std::atoi(k)) << 20;
        int maj = 0, min = 0;
        (void)hiprtcVersion(&maj, &min);
        versionTag = "hiprtc" + std::to_string(maj) + "." + std::to_string(min) + ";gfx950;-O3;-ffp-contract=off;v2";
#ifdef ELEMHIP_EXPERIMENTAL
        versionTag += ";experimental";
#endif
        unsigned n = std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 4u));   // a plan of a new graph brings several shapes at once
        if (const char* t = std::getenv("ELEMHIP_JIT_THREADS")) n = (unsigned)std::max(1, std::atoi(t));
        // hiprtc serialises compilations inside a process (jitc_main.cpp): the workers hand their shapes to a helper process each
        // (elemhip_jitc next to the library; ELEMHIP_JIT_INPROCESS=1 or a missing helper: the in-process compiler, one at a time)
        {
            const char* inproc = std::getenv("ELEMHIP_JIT_INPROCESS");
            const std::string h = libraryDir() + "/elemhip_jitc";
            if (!(inproc && std::atoi(inproc) != 0) && ::access(h.c_str(), X_OK) == 0) helper = h;
        }
        // The compiler library (comgr) is loaded lazily by the first hiprtc compile and registers its static destructors
        // then. Do that first compile here, on the calling thread (~45 ms, once per process), and register the exit hook
        // right after it: the hook then runs BEFORE those destructors whenever the process exits, however early.
        // (With the helper the compiler never runs in this process: only the exit hook that stops the workers is registered.)
        if (helper.empty()) warmUp();
        else std::atexit([] { if (Impl* i = exitHookTarget) i->shutdown(); });
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
            joined = true; stop = true; queue.clear(); lowQueue.clear();
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

    // (mu held) the next piece of work. The plan queue goes first, newest request first once it is backed up (under a stream of
    // ever-new shapes the newest voice is the one that will live longest after its compile); requests whose plans are all gone are
    // dropped instead of compiled while others wait.
    std::shared_ptr<SpecEntry> take() {
        for (;;) {
            std::shared_ptr<SpecEntry> e;
            bool others;
            if (!queue.empty()) {
                if (queue.size() > workers.size()) { e = std::move(queue.back()); queue.pop_back(); }
                else { e = std::move(queue.front()); queue.pop_front(); }
                others = !queue.empty();
            } else if (!lowQueue.empty()) {
                e = std::move(lowQueue.front()); lowQueue.pop_front();
                others = !lowQueue.empty();
            } else return nullptr;
            // references: the table's and this one; anything above that is a plan (or a waiting commit)
            if (others && e.use_count() <= 2) {
                e->state.store(-2, std::memory_order_release);
                auto it = entries.find(e->key);
                if (it != entries.end() && it->second == e) entries.erase(it);
                st.abandoned++;
                continue;
            }
            return e;
        }
    }

    void run() {
        for (;;) {
            std::shared_ptr<SpecEntry> e;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return stop || !queue.empty() || !lowQueue.empty(); });
                if (stop) return;
                e = take();
                if (!e) continue;
            }
            compile(*e);
            e->preload();
            {
                std::lock_guard<std::mutex> l(mu);
                const int s = e->state.load(std::memory_order_acquire);
                if (s == 1 && e->fromDisk) st.diskHits++;
                else if (s == 1) { st.compiles++; st.compileMsTotal += e->compileMs; st.compileMsLast = e->compileMs; st.compileMsMax = std::max(st.compileMsMax, e->compileMs); }
                else st.failed++;
            }
            cv.notify_all();
        }
    }

    void compile(SpecEntry& e) {
        const std::string path = cacheDir + "/" + e.key + ".hsaco";
        if (diskOk) {   // disk cache
            std::vector<char> code;
            if (readTrustedFile(path, code)) {
                e.code.swap(code); e.fromDisk = true;
                (void)utimensat(AT_FDCWD, path.c_str(), nullptr, AT_SYMLINK_NOFOLLOW);      // a hit keeps the file young (the cap removes oldest first)
                e.state.store(1, std::memory_order_release);
                return;
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        // the 300 KB translation unit lives for the duration of the compile only (ELEMHIP_JIT_KEEP_SOURCE=1: it stays in the entry)
        const std::string src = Jit::fullSource(e.generated, e.ldsWords, e.block);
        if (keepSource) { std::lock_guard<std::mutex> l(e.mu); e.source = src; }
        if (!helper.empty()) {
            // out of process, in a scratch directory of this compile alone (mkdtemp: mode 0700, a name nobody can pre-place): the source
            // goes in, the helper writes its code object and its log beside it; the code object is checked (an ELF for gfx) and renamed
            // into the shared cache only then
            std::string scratch = (diskOk ? cacheDir : std::string("/tmp")) + "/.jit.XXXXXX";
            const bool haveScratch = mkdtemp(&scratch[0]) != nullptr;
            const std::string srcPath = scratch + "/k.src", outPath = scratch + "/k.hsaco", logPath = scratch + "/k.log";
            bool ok = haveScratch && writeNewFile(srcPath, src.data(), src.size());
            int status = -1;
            if (ok) {
                posix_spawn_file_actions_t fa;
                posix_spawn_file_actions_init(&fa);
                posix_spawn_file_actions_addopen(&fa, 2, logPath.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
                posix_spawn_file_actions_addopen(&fa, 1, "/dev/null", O_WRONLY, 0644);
                char* const argv[] = {const_cast<char*>(helper.c_str()), const_cast<char*>(srcPath.c_str()), const_cast<char*>(outPath.c_str()), nullptr};
                pid_t pid = 0;
                if (posix_spawn(&pid, helper.c_str(), &fa, nullptr, argv, ::environ) == 0) { while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {} }
                posix_spawn_file_actions_destroy(&fa);
            }
            { std::ifstream lf(logPath, std::ios::binary); if (lf) e.log.assign((std::istreambuf_iterator<char>(lf)), std::istreambuf_iterator<char>()); }
            bool done = false;
            if (ok && WIFEXITED(status) && WEXITSTATUS(status) == 0) {
                std::vector<char> code;
                if (readTrustedFile(outPath, code)) {
                    e.code.swap(code);
                    e.compileMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    const bool written = diskOk && std::rename(outPath.c_str(), path.c_str()) == 0;
                    e.state.store(1, std::memory_order_release);
                    if (written) trimDisk((int64_t)e.code.size());
                    done = true;
                } else e.log += "\n(the helper's output is not a gfx code object)";
            }
            if (haveScratch) {
                (void)std::remove(srcPath.c_str()); (void)std::remove(logPath.c_str()); (void)std::remove(outPath.c_str());
                (void)rmdir(scratch.c_str());
            }
            if (done) return;
            std::fprintf(stderr, "[elemhip] jit: compilation of shape %s failed (helper status %d):\n%.3000s\n", e.key.c_str(), status, e.log.c_str());
            e.state.store(-1, std::memory_order_release);
            return;
        }
        hiprtcProgram prog = nullptr;
        if (hiprtcCreateProgram(&prog, src.c_str(), "elemhip_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
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
        e.compileMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        // disk cache: best effort, a new file of our own (O_EXCL) renamed into place
        bool written = false;
        if (diskOk) {
            const std::string tmp = path + "." + std::to_string((long)getpid()) + "." + std::to_string((unsigned long long)tick.fetch_add(1)) + ".tmp";
            if (writeNewFile(tmp, e.code.data(), e.code.size())) {
                if (std::rename(tmp.c_str(), path.c_str()) != 0) (void)std::remove(tmp.c_str()); else written = true;
            }
        }
        e.state.store(1, std::memory_order_release);
        if (written) trimDisk((int64_t)e.code.size());
    }

    // the on-disk cache stays under its cap: when a write takes it over, the oldest code objects (mtime) go until 3/4 are left
    void trimDisk(int64_t added) {
        struct F { std::string name; int64_t size; int64_t mtime; };
        std::vector<F> files;
        {
            std::lock_guard<std::mutex> l(mu);
            if (diskBytes >= 0) { diskBytes += added; st.diskBytes = (uint64_t)diskBytes; if ((uint64_t)diskBytes <= diskCapBytes) return; }
        }
        DIR* d = opendir(cacheDir.c_str());
        if (!d) return;
        int64_t total = 0;
        while (struct dirent* en = readdir(d)) {
            const std::string n = en->d_name;
            if (n.size() < 7 || n.compare(n.size() - 6, 6, ".hsaco") != 0) continue;
            struct stat sb;
            if (::stat((cacheDir + "/" + n).c_str(), &sb) != 0) continue;
            files.push_back({n, (int64_t)sb.st_size, (int64_t)sb.st_mtime});
            total += (int64_t)sb.st_size;
        }
        closedir(d);
        uint64_t removed = 0;
        if ((uint64_t)total > diskCapBytes) {
            std::sort(files.begin(), files.end(), [](const F& a, const F& b) { return a.mtime < b.mtime; });
            for (const F& f : files) {
                if ((uint64_t)total <= diskCapBytes / 4 * 3) break;
                if (std::remove((cacheDir + "/" + f.name).c_str()) == 0) { total -= f.size; ++removed; }
            }
        }
        std::lock_guard<std::mutex> l(mu);
        diskBytes = total; st.diskBytes = (uint64_t)total; st.diskFilesRemoved += removed;
        if (removed) notOnDisk.clear();
    }

    // (mu held) a backed-up plan queue is swept for requests nobody waits for any more — the voice was replaced before a worker got to
    // its shape: they would only be dropped when their turn came, and until then they sit in the table above its cap
    void sweep(const std::shared_ptr<SpecEntry>& keep) {
        if (queue.size() <= std::max<size_t>(16, 2 * workers.size())) return;
        for (auto q = queue.begin(); q != queue.end();) {
            if (q->use_count() <= 2 && *q != keep) {
                (*q)->state.store(-2, std::memory_order_release);
                auto m = entries.find((*q)->key);
                if (m != entries.end() && m->second == *q) entries.erase(m);
                st.abandoned++;
                q = queue.erase(q);
            } else ++q;
        }
    }

    // (mu held) the table stays under its cap: entries nobody references (no plan, no queue) go, least recently used first
    void evict() {
        if (entries.size() <= entryCap) return;
        std::vector<std::pair<uint64_t, std::string>> idle;
        for (auto& kv : entries) {
            const int s = kv.second->state.load(std::memory_order_acquire);
            if (kv.second.use_count() == 1 && (s == 1 || s == -1 || s == 2)) idle.emplace_back(kv.second->lastUse.load(std::memory_order_relaxed), kv.first);
        }
        std::sort(idle.begin(), idle.end());
        const size_t target = entryCap - entryCap / 4;
        for (auto& p : idle) {
            if (entries.size() <= target) break;
            entries.erase(p.second);          // ~SpecEntry unloads the modules
            st.evictions++;
        }
    }
};

Jit::Impl* Jit::Impl::exitHookTarget = nullptr;

Jit& Jit::get() { static Jit* j = new Jit; return *j; }   // never destroyed: the exit hook (warmUp) stops the workers
Jit::Jit() : impl(new Impl) {}
Jit::~Jit() { delete impl; }
void Jit::shutdownAtExit() { impl->shutdown(); }
void Jit::noteModuleLoaded(int delta) { impl->modulesLoaded.fetch_add(delta, std::memory_order_relaxed); }
void Jit::setEntryCap(uint32_t cap) {
    std::lock_guard<std::mutex> l(impl->mu);
    impl->entryCap = cap ? std::max(4u, cap) : 256u;
    impl->evict();
}

// everything in front of the generated text (a function of the LDS size, the engine's block size and the experimental defines)
static std::string sourcePrefix(uint32_t ldsWords, uint32_t block, size_t reserveExtra) {
    std::string s;
    s.reserve(sizeof(kSpecDeviceH) + sizeof(kSpecOpsInc) + sizeof(kSpecInc) + reserveExtra + 256);
    s += "#define ELEMHIP_SPEC 1\n#define ELEMHIP_SPEC_LDS_WORDS " + std::to_string(ldsWords) + "\n#define ELEMHIP_SPEC_BLOCK " + std::to_string(block) + "\n";
#ifdef ELEMHIP_EXPERIMENTAL
    s += "#define ELEMHIP_EXPERIMENTAL 1\n";
#endif
    s += tuningDefines();
    const std::string defs = experimentalDefines();
    if (!defs.empty()) {
        std::string tok;
        for (size_t i = 0; i <= defs.size(); ++i) {
            if (i == defs.size() || defs[i] == ' ') {
                if (!tok.empty()) { const size_t eq = tok.find('='); s += "#define " + (eq == std::string::npos ? tok + " 1" : tok.substr(0, eq) + " " + tok.substr(eq + 1)) + "\n"; }
                tok.clear();
            } else tok += defs[i];
        }
    }
    // hiprtc has no host headers: the fixed-width names the sources use (same underlying types as <stdint.h> on this target)
    s += "typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long uint64_t;\n"
         "typedef signed char int8_t; typedef short int16_t; typedef int int32_t; typedef long int64_t; typedef unsigned long uintptr_t;\n";
    s += kSpecDeviceH; s += "\n"; s += kSpecOpsInc; s += "\n"; s += kSpecInc; s += "\n";
    return s;
}

std::string Jit::fullSource(const std::string& generated, uint32_t ldsWords, uint32_t block) {
    std::string s = sourcePrefix(ldsWords, block, generated.size());
    s += generated;
    return s;
}

// The key is a hash of the whole translation unit; its first ~300 KB are the same for every shape of one LDS size, so the hash
// state behind them is kept (FNV-1a runs front to back: same keys as hashing the full source, a tenth of the time).
bool Jit::setTuning(const std::string& name, int value) {
    const bool ok = (name == "ELEMHIP_BIQUAD_FORM" && value >= 0 && value <= 5) ||
                    (name == "ELEMHIP_WIDE_CHAIN_DEPTH" && (value == 2 || value == 4 || value == 8));
    if (!ok) return false;
    std::lock_guard<std::mutex> l(gTuningMu);
    // replace an earlier line of the same name
    const std::string head = "#define " + name + " ";
    size_t at = gTuning.find(head);
    if (at != std::string::npos) gTuning.erase(at, gTuning.find('\n', at) - at + 1);
    gTuning += head + std::to_string(value) + "\n";
    return true;
}

std::string Jit::keyFor(const std::string& generated, uint32_t ldsWords, uint32_t block) {
    const std::string defs = experimentalDefines() + "|" + tuningDefines();
    const uint64_t pk = ((uint64_t)block << 32) | ldsWords;
    uint64_t h1 = 0, h2 = 0;
    bool have = false;
    {
        std::lock_guard<std::mutex> l(impl->mu);
        if (impl->prefixDefs == defs) { auto it = impl->prefixHash.find(pk); if (it != impl->prefixHash.end()) { h1 = it->second.first; h2 = it->second.second; have = true; } }
    }
    if (!have) {
        const std::string pre = sourcePrefix(ldsWords, block, 0);
        h1 = fnv1a(pre, fnv1a(impl->versionTag, 1469598103934665603ull));
        h2 = fnv1a(pre, fnv1a(impl->versionTag, 0x9E3779B97F4A7C15ull) ^ 0xA5A5A5A5ull);
        std::lock_guard<std::mutex> l(impl->mu);
        if (impl->prefixDefs != defs) { impl->prefixHash.clear(); impl->prefixDefs = defs; }
        if (impl->prefixHash.size() > 256) impl->prefixHash.clear();
        impl->prefixHash[pk] = {h1, h2};
    }
    h1 = fnv1a(generated, h1); h2 = fnv1a(generated, h2);
    char key[40];
    std::snprintf(key, sizeof key, "%016llx%016llx", (unsigned long long)h1, (unsigned long long)h2);
    return key;
}

bool Jit::knownKey(const std::string& key) {
    {
        std::lock_guard<std::mutex> l(impl->mu);
        auto it = impl->entries.find(key);
        if (it != impl->entries.end() && it->second->state.load(std::memory_order_acquire) != 2) return true;   // (a deferred entry has not been asked for yet)
        if (impl->notOnDisk.count(key)) return false;
    }
    // (asked on every re-plan of a live graph for the one-island shapes of a voice that is fading out: the answer from the
    //  file system — 100 us and more on a network mount — is remembered; a key that gets compiled later is found in `entries`)
    struct stat st;
    const bool there = impl->diskOk && ::lstat((impl->cacheDir + "/" + key + ".hsaco").c_str(), &st) == 0 && S_ISREG(st.st_mode);
    if (!there) { std::lock_guard<std::mutex> l(impl->mu); if (impl->notOnDisk.size() > 4096) impl->notOnDisk.clear(); impl->notOnDisk.insert(key); }
    return there;
}
uint32_t Jit::sighting(const std::string& key) {
    std::lock_guard<std::mutex> l(impl->mu);
    if (impl->sightings.size() > 4096) impl->sightings.clear();
    return ++impl->sightings[key];
}

std::shared_ptr<SpecEntry> Jit::requestKey(const std::string& key, const std::string& generated, uint32_t ldsWords, uint32_t block, bool deferred) {
    std::shared_ptr<SpecEntry> found;
    {
        std::lock_guard<std::mutex> l(impl->mu);
        auto it = impl->entries.find(key);
        if (it != impl->entries.end()) found = it->second;
    }
    if (!found) {
        auto e = std::make_shared<SpecEntry>();
        e->key = key; e->generated = generated; e->ldsWords = ldsWords; e->block = block; e->ldsBytes = ldsWords * 4u;
        e->state.store(deferred ? 2 : 0, std::memory_order_release);
        std::lock_guard<std::mutex> l(impl->mu);
        auto it = impl->entries.find(key);
        if (it != impl->entries.end()) found = it->second;
        else {
            e->lastUse.store(impl->tick.fetch_add(1), std::memory_order_relaxed);
            impl->entries.emplace(key, e);
            if (deferred) impl->st.deferred++;
            else {
                impl->queue.push_back(e); impl->st.queued++;
                impl->sweep(e);
                impl->cv.notify_one();
            }
            impl->evict();
            return e;
        }
    }
    found->lastUse.store(impl->tick.fetch_add(1), std::memory_order_relaxed);
    if (!deferred && found->state.load(std::memory_order_acquire) == 2) promote(found, true);   // wanted in earnest now (second sighting, a waiting commit)
    return found;
}

void Jit::promote(const std::shared_ptr<SpecEntry>& e, bool urgent) {
    int expect = 2;
    if (!e->state.compare_exchange_strong(expect, 0, std::memory_order_acq_rel)) return;
    std::lock_guard<std::mutex> l(impl->mu);
    if (urgent) { impl->queue.push_back(e); impl->st.queued++; impl->sweep(e); impl->evict(); }
    else {
        impl->lowQueue.push_back(e); impl->st.promoted++;
        // (promoted one-off shapes whose plans are gone by now: same sweep, the low queue is only served when the plan queue is empty)
        if (impl->lowQueue.size() > 32)
            for (auto q = impl->lowQueue.begin(); q != impl->lowQueue.end();) {
                if (q->use_count() <= 2 && *q != e) {
                    (*q)->state.store(-2, std::memory_order_release);
                    auto m = impl->entries.find((*q)->key);
                    if (m != impl->entries.end() && m->second == *q) impl->entries.erase(m);
                    impl->st.abandoned++;
                    q = impl->lowQueue.erase(q);
                } else ++q;
            }
    }
    impl->cv.notify_one();
}

int Jit::wait(const std::shared_ptr<SpecEntry>& e) {
    if (e->state.load(std::memory_order_acquire) == 2) promote(e, true);
    std::unique_lock<std::mutex> l(impl->mu);
    impl->cv.wait(l, [&] { const int s = e->state.load(std::memory_order_acquire); return impl->stop || (s != 0 && s != 2); });
    return e->state.load(std::memory_order_acquire);
}

JitStats Jit::stats() {
    std::lock_guard<std::mutex> l(impl->mu);
    JitStats s = impl->st;
    s.entries = impl->entries.size();
    s.modulesLoaded = (uint64_t)std::max<int64_t>(0, impl->modulesLoaded.load(std::memory_order_relaxed));
    s.entryCap = impl->entryCap; s.workers = (uint32_t)impl->workers.size(); s.diskCapBytes = impl->diskCapBytes;
    s.sourceBytesHeld = 0; s.codeBytesHeld = 0;
    for (auto& kv : impl->entries) { s.sourceBytesHeld += kv.second->source.capacity() + kv.second->generated.capacity(); s.codeBytesHeld += kv.second->code.capacity(); }
    return s;
}

std::string SpecEntry::fullText() {
    { std::lock_guard<std::mutex> l(mu); if (!source.empty()) return source; }
    return Jit::fullSource(generated, ldsWords, block);
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
        if (mod) { (void)hipModuleUnload(mod); mod = nullptr; }
        perDevice.emplace(device, std::make_pair(mod, (hipFunction_t) nullptr));
        return nullptr;
    }
    perDevice.emplace(device, std::make_pair(mod, fn));
    Jit::get().noteModuleLoaded(+1);
    return fn;
}

void SpecEntry::wantOn(int device) {
    if (device < 0) return;
    bool ready;
    {
        std::lock_guard<std::mutex> l(mu);
        if (std::find(wantDevices.begin(), wantDevices.end(), device) == wantDevices.end()) wantDevices.push_back(device);
        ready = state.load(std::memory_order_acquire) == 1 && perDevice.find(device) == perDevice.end();
    }
    if (ready && hipSetDevice(device) == hipSuccess) (void)function(device);       // (the caller is a commit on the control thread)
}

void SpecEntry::preload() {
    if (state.load(std::memory_order_acquire) != 1) return;
    std::vector<int> devs;
    { std::lock_guard<std::mutex> l(mu); devs = wantDevices; }
    for (int d : devs) {
        if (hipSetDevice(d) != hipSuccess) continue;
        (void)function(d);
    }
}

// The last reference is gone (evicted from the table and no plan left that names it — a plan is destroyed only after the
// synchronise that follows its last launch): the modules can be unloaded. Each on the device it was loaded on.
SpecEntry::~SpecEntry() {
    if (perDevice.empty()) return;
    int cur = -1;
    const bool haveCur = hipGetDevice(&cur) == hipSuccess;
    for (auto& kv : perDevice) {
        if (!kv.second.first) continue;
        if (hipSetDevice(kv.first) == hipSuccess && hipModuleUnload(kv.second.first) == hipSuccess) Jit::get().noteModuleLoaded(-1);
    }
    if (haveCur && cur >= 0) (void)hipSetDevice(cur);
}

} // namespace elemhip
