// jit.h — run-time compiled, per-island-shape kernels (jit.cpp, codegen.cpp, island_spec.inc).
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace elemhip {

struct SpecEntry {
    std::string key;                 // hex hash of the whole program text + compiler tag
    std::string generated;           // the shape's own part of the text (codegen.cpp): kept, a few KB — the node library in front of
                                     // it is the same for every shape and is re-attached on demand (fullText)
    std::string source;              // the whole translation unit: only between request and compile (ELEMHIP_JIT_KEEP_SOURCE=1 keeps it)
    std::string log;
    std::vector<char> code;          // gfx950 code object
    uint32_t ldsBytes = 0, ldsWords = 0, block = 0;
    bool fromDisk = false;
    double compileMs = 0.0;          // hiprtc time of this shape (0: disk hit)
    // 2 deferred (known to the cache, not queued: a one-off shape of a background-mode plan, Jit::promote queues it);
    // 0 queued / compiling; 1 ready; -1 failed; -2 abandoned (nobody wanted it any more when a worker got to it)
    std::atomic<int> state{0};
    std::atomic<uint64_t> lastUse{0};   // Jit tick of the newest request / launch look-up (eviction order)
    std::mutex mu;
    std::unordered_map<int, std::pair<hipModule_t, hipFunction_t>> perDevice;
    std::vector<int> wantDevices;         // devices whose engines asked for this shape: the compile worker loads the code object there, so that
                                          // the first launch does not pay the ~1 ms of hipModuleLoadData inside a render call
    void wantOn(int device);              // (control thread) remember the device; a ready entry is loaded at once, off the render path
    void preload();                       // (compile worker) load on every wanted device
    hipFunction_t function(int device);   // nullptr until ready (or if loading failed)
    std::string fullText();               // the translation unit hiprtc saw (debug / tests: elemhip_spec_info)
    ~SpecEntry();                         // unloads its modules (each on its own device)
};

struct JitStats {
    uint64_t entries = 0, modulesLoaded = 0, compiles = 0, diskHits = 0, failed = 0, evictions = 0, abandoned = 0, queued = 0,
             deferred = 0, promoted = 0, diskBytes = 0, diskFilesRemoved = 0, sourceBytesHeld = 0, codeBytesHeld = 0;
    double compileMsTotal = 0.0, compileMsMax = 0.0, compileMsLast = 0.0;
    uint32_t entryCap = 0, workers = 0;
    uint64_t diskCapBytes = 0;
};

class Jit {
public:
    static Jit& get();
    // the cache key of a shape (hashing the program text costs 0.3 ms: callers that see the same text object again keep the key)
    std::string keyFor(const std::string& generated, uint32_t ldsWords, uint32_t block);
    bool knownKey(const std::string& key);          // requested in this process (and not abandoned), or on disk
    uint32_t sighting(const std::string& key);     // how many plans (this one included) have wanted this not-yet-compiled shape
    // the entry of `key`, created and queued for compilation if new. `deferred`: create it WITHOUT queueing (a shape only one island
    // of a background-mode plan has); promote() queues it at low priority once its plan has proved to stay
    std::shared_ptr<SpecEntry> requestKey(const std::string& key, const std::string& generated, uint32_t ldsWords, uint32_t block, bool deferred = false);
    void promote(const std::shared_ptr<SpecEntry>& e, bool urgent = false);
    int wait(const std::shared_ptr<SpecEntry>& e);   // blocks until compiled: 1 ready, -1 failed
    static std::string fullSource(const std::string& generated, uint32_t ldsWords, uint32_t block);
    JitStats stats();
    // a whitelisted tuning define of the run-time compiler (PROCESS-wide, part of every later kernel's cache key). Only knobs that
    // leave every sample bit-identical are accepted: "ELEMHIP_BIQUAD_FORM" 0..2, "ELEMHIP_WIDE_CHAIN_DEPTH" 2 | 4 | 8. false = refused.
    bool setTuning(const std::string& name, int value);
    void setEntryCap(uint32_t cap);                  // option "jit_cache_entries": in-memory entries (code objects + loaded modules) kept; 0 = default
    void noteModuleLoaded(int delta);
    void shutdownAtExit();
private:
    Jit();
    ~Jit();
    struct Impl;
    Impl* impl;
};

} // namespace elemhip
