// elemhip/Runtime.hpp — header-only C++ façade over the C-ABI (include/elemhip.h) with the public
// surface of `elem::Runtime<float>` (runtime/elem/Runtime.h:39-153), so a host written like
// cli/Benchmark.cpp switches engines by changing the class it instantiates:
//
//     #include <elem/Runtime.h>        ->   #include <elemhip/Runtime.hpp>
//     elem::Runtime<float> rt(sr, bs); ->   elemhip::Runtime<float> rt(sr, bs);
//
// When the reference's own headers are on the include path (<elem/Value.h>, <elem/JSON.h>, <elem/GraphNode.h>) the
// façade also offers the members that speak `elem::js::Value`: applyInstructions(js::Array const&), snapshot(),
// processQueuedEvents with a js::Value payload, and registerNodeType with an unmodified `elem::GraphNode<float>`
// subclass (rendered as a CPU call-out node, see elemhip_register_node_type). Define ELEMHIP_NO_ELEM_HEADERS to
// opt out. Without them the JSON-text members below are the whole surface.
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../elemhip.h"

#if !defined(ELEMHIP_NO_ELEM_HEADERS) && defined(__has_include)
#if __has_include(<elem/Value.h>) && __has_include(<elem/JSON.h>) && __has_include(<elem/GraphNode.h>)
#include <array>
#include <atomic>
#include <cmath>
#include <list>
#include <map>
#include <numeric>
#include <elem/Value.h>
#include <elem/JSON.h>
#include <elem/GraphNode.h>
#define ELEMHIP_HAVE_ELEM_HEADERS 1
#endif
#endif

namespace elemhip {

#ifdef ELEMHIP_HAVE_ELEM_HEADERS
namespace detail {
// elem::js::Value -> JSON text (Value::toString is a debug print, not JSON)
inline void toJSON(elem::js::Value const& v, std::string& out) {
    if (v.isBool()) out += ((elem::js::Boolean)v) ? "true" : "false";
    else if (v.isNumber()) { char b[40]; std::snprintf(b, sizeof b, "%.17g", (double)(elem::js::Number)v); out += b; }
    else if (v.isString()) {
        out += '"';
        for (unsigned char ch : (elem::js::String)v) {
            if (ch == '"' || ch == '\\') { out += '\\'; out += (char)ch; }
            else if (ch < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", ch); out += b; }
            else out += (char)ch;
        }
        out += '"';
    } else if (v.isArray()) {
        out += '[';
        auto const& a = v.getArray();
        for (size_t i = 0; i < a.size(); ++i) { if (i) out += ','; toJSON(a[i], out); }
        out += ']';
    } else if (v.isFloat32Array()) {
        out += '[';
        auto const& a = v.getFloat32Array();
        for (size_t i = 0; i < a.size(); ++i) { char b[40]; std::snprintf(b, sizeof b, "%s%.9g", i ? "," : "", (double)a[i]); out += b; }
        out += ']';
    } else if (v.isObject()) {
        out += '{';
        bool first = true;
        for (auto const& kv : v.getObject()) {
            if (!first) out += ',';
            first = false;
            toJSON(elem::js::Value(kv.first), out); out += ':'; toJSON(kv.second, out);
        }
        out += '}';
    } else out += "null";
}
} // namespace detail
#endif

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

    // The offline caller's block loop (js/packages/offline-renderer/index.ts:87-133) in one call: planar HOST arrays of
    // numFrames frames per channel, ceil(numFrames / blockSize) full blocks, outputs overwritten. Same userData rule as
    // process(): a pointer to the int64_t sample time of the first frame, or null for the implicit running clock.
    int processBlocks(const float** in, size_t nIn, float** out, size_t nOut, size_t numFrames, void* userData = nullptr) {
        const int64_t t = userData ? *static_cast<int64_t*>(userData) : implicitTime;
        const int rc = elemhip_process_blocks_host(h, in, nIn, out, nOut, numFrames, t);
        if (!userData) implicitTime += (int64_t)numFrames;
        return rc;
    }

    // bool addSharedResource(name, unique_ptr<SharedResource>)         Runtime.h:83 — planar float channels
    bool addSharedResource(std::string const& name, const float* const* channels, size_t nCh, size_t nSamples) {
        return elemhip_add_shared_resource(h, name.c_str(), channels, nCh, nSamples) != 0;
    }
    void pruneSharedResources() { elemhip_prune_shared_resources(h); }                    // Runtime.h:89
    std::vector<std::string> getSharedResourceMapKeys() {                                 // Runtime.h:94
        std::string j = text([&](char* b, size_t c) { return elemhip_shared_resource_keys_json(h, b, c); });
        std::vector<std::string> keys;   // a flat JSON array of strings
        size_t i = 0;
        while ((i = j.find('"', i)) != std::string::npos) {
            std::string k;
            for (++i; i < j.size() && j[i] != '"'; ++i) { if (j[i] == '\\' && i + 1 < j.size()) ++i; k += j[i]; }
            keys.push_back(k);
            ++i;
        }
        return keys;
    }
    std::string snapshotJSON() { return text([&](char* b, size_t c) { return elemhip_snapshot_json(h, b, c); }); }   // Runtime.h:110
    // processQueuedEvents with the payload as JSON text                                  Runtime.h:64
    void processQueuedEventsJSON(std::function<void(std::string const&, std::string const&)>&& cb) {
        auto tramp = [](const char* type, const char* json, void* user) {
            (*static_cast<std::function<void(std::string const&, std::string const&)>*>(user))(type, json);
        };
        elemhip_process_queued_events(h, tramp, &cb);
    }
#ifndef ELEMHIP_HAVE_ELEM_HEADERS
    void processQueuedEvents(std::function<void(std::string const&, std::string const&)>&& cb) { processQueuedEventsJSON(std::move(cb)); }
#endif
    void reset() { elemhip_reset(h); }                                                    // Runtime.h:70
    std::set<int32_t> gc() {                                                              // Runtime.h:76
        std::vector<int32_t> buf(1 << 12);
        size_t n = elemhip_gc(h, buf.data(), buf.size());
        if (n > buf.size()) { buf.resize(n); n = elemhip_last_gc(h, buf.data(), buf.size()); }
        return std::set<int32_t>(buf.begin(), buf.begin() + (long)n);
    }
    // int registerNodeType(type, NodeFactoryFn&&)                      Runtime.h:105 — C form (include/elemhip.h)
    int registerNodeType(std::string const& type, elemhip_node_type const& vt) { return elemhip_register_node_type(h, type.c_str(), &vt); }

#ifdef ELEMHIP_HAVE_ELEM_HEADERS
    // ---- the members that speak elem::js::Value (reference headers on the include path) ----
    // bool addSharedResource(std::string const& name, std::unique_ptr<SharedResource> resource)   Runtime.h:83,461-465
    // The engine copies the channels (the resource may die when this returns); insert-only like the reference.
    bool addSharedResource(std::string const& name, std::unique_ptr<elem::SharedResource> resource) {
        if (!resource) return false;
        const size_t nCh = resource->numChannels(), n = resource->numSamples();
        std::vector<const float*> chans(nCh ? nCh : 1, nullptr);
        for (size_t c = 0; c < nCh; ++c) chans[c] = resource->getChannelData(c).data();
        return addSharedResource(name, chans.data(), nCh, n);
    }
    int applyInstructions(elem::js::Array const& batch) {                                 // Runtime.h:48,170-218
        std::string j;
        detail::toJSON(elem::js::Value(batch), j);
        return applyInstructionsJSON(j);
    }
    elem::js::Object snapshot() {                                                         // Runtime.h:110
        auto v = elem::js::parseJSON(snapshotJSON());
        return v.isObject() ? v.getObject() : elem::js::Object();
    }
    void processQueuedEvents(std::function<void(std::string const&, elem::js::Value)>&& cb) {   // Runtime.h:64
        processQueuedEventsJSON([&](std::string const& type, std::string const& json) { cb(type, elem::js::parseJSON(json)); });
    }
    // registerNodeType with the reference's own factory type (Runtime.h:100-106): the node is an unmodified
    // elem::GraphNode<float> subclass, rendered as a CPU call-out between two GPU launch levels.
    using NodeFactoryFn = std::function<std::shared_ptr<elem::GraphNode<float>>(elem::NodeId const id, double sampleRate, int const blockSize)>;
    int registerNodeType(std::string const& type, NodeFactoryFn&& fn) {
        auto* holder = new Factory{std::move(fn)};   // lives as long as the engine may create nodes of the type
        factories.emplace_back(holder);
        elemhip_node_type vt{};
        vt.user = holder;
        vt.create = [](int32_t id, double sr, int bs, void* user) -> void* {
            auto node = static_cast<Factory*>(user)->fn(id, sr, bs);
            return node ? new std::shared_ptr<elem::GraphNode<float>>(std::move(node)) : nullptr;
        };
        vt.destroy = [](void* node, void*) { delete static_cast<std::shared_ptr<elem::GraphNode<float>>*>(node); };
        vt.set_property = [](void* node, const char* key, const char* json, void*) -> int {
            if (!node) return 0;
            // (elem::js::parseJSON only accepts an array or object at the top level: wrap the bare value)
            auto wrapped = elem::js::parseJSON("[" + std::string(json) + "]");
            return (*static_cast<std::shared_ptr<elem::GraphNode<float>>*>(node))->setProperty(key, wrapped.getArray().at(0));
        };
        vt.process = [](void* node, const float* const* in, size_t nIn, float* out, size_t n, int64_t sampleTime, int active, void*) {
            if (!node) return;
            int64_t t = sampleTime;
            float* outs[1] = {out};
            elem::BlockContext<float> ctx{const_cast<float const**>(in), nIn, outs, 1, n, &t, active != 0};
            (*static_cast<std::shared_ptr<elem::GraphNode<float>>*>(node))->process(ctx);
        };
        vt.reset = [](void* node, void*) { if (node) (*static_cast<std::shared_ptr<elem::GraphNode<float>>*>(node))->reset(); };
        return elemhip_register_node_type(h, type.c_str(), &vt);
    }
#endif

    elemhip_t* handle() { return h; }

private:
    template <typename F>
    static std::string text(F&& get) {
        const size_t need = get(nullptr, 0);
        std::string s(need ? need : 1, '\0');
        get(&s[0], s.size());
        s.resize(need ? need - 1 : 0);
        return s;
    }
#ifdef ELEMHIP_HAVE_ELEM_HEADERS
    struct Factory { NodeFactoryFn fn; };
    std::vector<std::unique_ptr<Factory>> factories;
#endif
    elemhip_t* h;
    int64_t implicitTime = 0;
};

} // namespace elemhip
