// oracle/ref_driver.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin C-ABI around the UNMODIFIED reference engine, compiled in place from
// /root/reference/runtime (header-only `elem::Runtime<F>`, runtime/elem/Runtime.h:39-153)
// plus the two wasm-host nodes that compile natively (wasm/SampleTime.h, wasm/Metro.h,
// registered the way wasm/Main.cpp:55-61 does).  Output goes to oracle/_ref/ only.
//
// The entry points mirror include/elemhip.h one-for-one (`elemref_*` vs `elemhip_*`) so
// the parity tests can drive both engines with the same instruction batches.
//
// Not available natively: `convolve`/`fft` (wasm/Convolve.h:3 needs the un-vendored
// FFTConvolver submodule) — see oracle/wasm_convolve.js for that oracle.
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include <elem/Runtime.h>
#include <elem/AudioBufferResource.h>
#include <SampleTime.h>
#include <Metro.h>

namespace {

struct RefBase {
    virtual ~RefBase() = default;
    virtual int apply(const char* json, size_t len) = 0;
    virtual int process(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime) = 0;
    virtual int addResource(const char* name, const float* const* ch, size_t nCh, size_t nSamples) = 0;
    virtual void prune() = 0;
    virtual size_t gc(int32_t* out, size_t cap) = 0;
    virtual void reset() = 0;
    virtual void events(void (*cb)(const char*, const char*, void*), void* user) = 0;
};

template <typename F>
struct Ref final : RefBase {
    elem::Runtime<F> rt;
    int blockSize;
    std::vector<std::vector<F>> inScratch, outScratch;
    std::vector<const F*> inPtrs;
    std::vector<F*> outPtrs;

    Ref(double sr, int bs) : rt(sr, bs), blockSize(bs) {
        rt.registerNodeType("time", [](elem::NodeId id, double fs, int b) {
            return std::make_shared<elem::SampleTimeNode<F>>(id, fs, b);
        });
        rt.registerNodeType("metro", [](elem::NodeId id, double fs, int b) {
            return std::make_shared<elem::MetronomeNode<F>>(id, fs, b);
        });
    }

    int apply(const char* json, size_t len) override {
        try {
            auto v = elem::js::parseJSON(std::string(json, len));
            if (!v.isArray()) return elem::ReturnCode::InvalidInstructionFormat();
            return rt.applyInstructions(v.getArray());
        } catch (std::exception const&) {
            // The reference lets exceptions (bad_variant_access on wrong value types,
            // parse errors) propagate to its host; the wasm host reports failure.
            return -1;
        }
    }

    int process(const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime) override {
        int64_t st = sampleTime;
        if constexpr (std::is_same_v<F, float>) {
            rt.process(const_cast<const float**>(in), nIn, const_cast<float**>(out), nOut, n, &st);
        } else {
            inScratch.resize(nIn); outScratch.resize(nOut);
            inPtrs.resize(nIn); outPtrs.resize(nOut);
            for (size_t c = 0; c < nIn; ++c) {
                inScratch[c].assign(in[c], in[c] + n);
                inPtrs[c] = inScratch[c].data();
            }
            for (size_t c = 0; c < nOut; ++c) {
                outScratch[c].assign(n, F(0));
                outPtrs[c] = outScratch[c].data();
            }
            rt.process(inPtrs.data(), nIn, outPtrs.data(), nOut, n, &st);
            for (size_t c = 0; c < nOut; ++c)
                for (size_t i = 0; i < n; ++i)
                    out[c][i] = static_cast<float>(outScratch[c][i]);
        }
        return 0;
    }

    int addResource(const char* name, const float* const* ch, size_t nCh, size_t nSamples) override {
        std::vector<float*> ptrs(nCh);
        for (size_t c = 0; c < nCh; ++c) ptrs[c] = const_cast<float*>(ch[c]);
        auto res = std::make_unique<elem::AudioBufferResource>(ptrs.data(), nCh, nSamples);
        return rt.addSharedResource(name, std::move(res)) ? 1 : 0;
    }

    void prune() override { rt.pruneSharedResources(); }

    size_t gc(int32_t* out, size_t cap) override {
        auto pruned = rt.gc();
        size_t k = 0;
        for (auto id : pruned) { if (k < cap && out) out[k] = id; ++k; }
        return k;
    }

    void reset() override { rt.reset(); }

    void events(void (*cb)(const char*, const char*, void*), void* user) override {   // Runtime.h:437-446
        rt.processQueuedEvents([&](std::string const& type, elem::js::Value evt) {
            const std::string json = elem::js::serialize(evt);
            cb(type.c_str(), json.c_str(), user);
        });
    }
};

} // namespace

extern "C" {

void* elemref_create(double sampleRate, int blockSize, int useDouble) {
    if (useDouble) return new Ref<double>(sampleRate, blockSize);
    return new Ref<float>(sampleRate, blockSize);
}
void elemref_destroy(void* h) { delete static_cast<RefBase*>(h); }
int elemref_apply_instructions_json(void* h, const char* json, size_t len) {
    return static_cast<RefBase*>(h)->apply(json, len);
}
int elemref_process(void* h, const float* const* in, size_t nIn, float* const* out, size_t nOut, size_t n, int64_t sampleTime) {
    return static_cast<RefBase*>(h)->process(in, nIn, out, nOut, n, sampleTime);
}
int elemref_add_shared_resource(void* h, const char* name, const float* const* ch, size_t nCh, size_t nSamples) {
    return static_cast<RefBase*>(h)->addResource(name, ch, nCh, nSamples);
}
void elemref_prune_shared_resources(void* h) { static_cast<RefBase*>(h)->prune(); }
size_t elemref_gc(void* h, int32_t* out, size_t cap) { return static_cast<RefBase*>(h)->gc(out, cap); }
void elemref_reset(void* h) { static_cast<RefBase*>(h)->reset(); }
int elemref_process_queued_events(void* h, void (*cb)(const char*, const char*, void*), void* user) {
    static_cast<RefBase*>(h)->events(cb, user); return 0;
}

} // extern "C"
