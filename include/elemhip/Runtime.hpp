// elemhip/Runtime.hpp — header-only C++ façade over the C-ABI (include/elemhip.h) with the public
// surface of `elem::Runtime<float>` (runtime/elem/Runtime.h:39-153), so a host written like
// cli/Benchmark.cpp switches engines by changing the class it instantiates.
#pragma once
#include <cstdint>
#include <functional>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../elemhip.h"

namespace elemhip {

template <typename FloatType>
class Runtime;

template <>
class Runtime<float> {
public:
    // Runtime(double sampleRate, int blockSize)                       Runtime.h:44
    Runtime(double sampleRate, int blockSize, int deviceOrdinal = 0)
        : h(elemhip_create(sampleRate, blockSize, deviceOrdinal)) {
        if (!h) throw std::runtime_error(std::string("elemhip_create: ") + elemhip_describe(elemhip_last_create_error()));
    }
    ~Runtime() { elemhip_destroy(h); }
    Runtime(Runtime const&) = delete;
    Runtime& operator=(Runtime const&) = delete;

    // int applyInstructions(js::Array const& batch)                   Runtime.h:48 — as the JSON text the
    // hosts already hold (cli/Benchmark.cpp:40-43 parses it only to hand it over)
    int applyInstructionsJSON(std::string const& batch) { return elemhip_apply_instructions_json(h, batch.data(), batch.size()); }

    // void process(const F** in, size_t nIn, F** out, size_t nOut, size_t n, void* userData)   Runtime.h:51-57
    // userData is what the reference's hosts pass: a pointer to the int64_t sample time (wasm/Main.cpp:212) or null.
    void process(const float** in, size_t nIn, float** out, size_t nOut, size_t numSamples, void* userData = nullptr) {
        const int64_t t = userData ? *static_cast<int64_t*>(userData) : implicitTime;
        (void)elemhip_process(h, in, nIn, out, nOut, numSamples, t);
        if (!userData) implicitTime += (int64_t)numSamples;
    }

    // bool addSharedResource(name, unique_ptr<SharedResource>)         Runtime.h:83 — planar float channels
    bool addSharedResource(std::string const& name, const float* const* channels, size_t nCh, size_t nSamples) {
        return elemhip_add_shared_resource(h, name.c_str(), channels, nCh, nSamples) != 0;
    }
    void pruneSharedResources() { elemhip_prune_shared_resources(h); }                    // Runtime.h:89
    // processQueuedEvents: the payload arrives as JSON text (parse with elem::js::parseJSON to get the js::Value back)
    void processQueuedEvents(std::function<void(std::string const&, std::string const&)>&& cb) {   // Runtime.h:64
        auto tramp = [](const char* type, const char* json, void* user) {
            (*static_cast<std::function<void(std::string const&, std::string const&)>*>(user))(type, json);
        };
        elemhip_process_queued_events(h, tramp, &cb);
    }
    void reset() { elemhip_reset(h); }                                                    // Runtime.h:70
    std::set<int32_t> gc() {                                                              // Runtime.h:76
        std::vector<int32_t> buf(1 << 16);
        size_t n = elemhip_gc(h, buf.data(), buf.size());
        if (n > buf.size()) n = buf.size();
        return std::set<int32_t>(buf.begin(), buf.begin() + (long)n);
    }
    elemhip_t* handle() { return h; }

private:
    elemhip_t* h;
    int64_t implicitTime = 0;
};

} // namespace elemhip
