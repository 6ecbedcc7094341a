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
    std::string source;
    std::string log;
    std::vector<char> code;          // gfx950 code object
    uint32_t ldsBytes = 0;
    bool fromDisk = false;
    std::atomic<int> state{0};       // 0 compiling, 1 ready, -1 failed
    std::mutex mu;
    std::unordered_map<int, std::pair<hipModule_t, hipFunction_t>> perDevice;
    hipFunction_t function(int device);   // nullptr until ready (or if loading failed)
};

class Jit {
public:
    static Jit& get();
    // queue `generated` (codegen.cpp text of one island shape) for compilation; identical text -> the same entry
    std::shared_ptr<SpecEntry> request(const std::string& generated, uint32_t ldsWords);
    int wait(const std::shared_ptr<SpecEntry>& e);   // blocks until compiled: 1 ready, -1 failed
    bool known(const std::string& generated, uint32_t ldsWords);   // already requested in this process, or on disk
    // the cache key of a shape (hashing the ~170 KB program text costs 0.3 ms: callers that see the same text object again
    // keep the key) and the two calls above by key; `request` builds the program text only for a key it has not seen
    std::string keyFor(const std::string& generated, uint32_t ldsWords);
    bool knownKey(const std::string& key);
    uint32_t sighting(const std::string& key);     // how many plans (this one included) have wanted this not-yet-compiled shape
    std::shared_ptr<SpecEntry> requestKey(const std::string& key, const std::string& generated, uint32_t ldsWords);
    static std::string fullSource(const std::string& generated, uint32_t ldsWords);
    void shutdownAtExit();
private:
    Jit();
    ~Jit();
    struct Impl;
    Impl* impl;
};

} // namespace elemhip
