// tests/native/facade_host.cpp — TEST INFRASTRUCTURE. A host written against the REFERENCE's public surface
// (runtime/elem/Runtime.h:39-153, the way cli/Benchmark.cpp:31-112 uses it) that instantiates elemhip::Runtime<float>
// instead of elem::Runtime<float>: applyInstructions(js::Array) with the batch parsed by elem::js::parseJSON,
// registerNodeType with UNMODIFIED reference node classes (wasm/Metro.h, wasm/SampleTime.h: the two custom nodes
// wasm/Main.cpp:55-61 registers) under the names "cpumetro" / "cputime", process(), gc(), snapshot(),
// getSharedResourceMapKeys(), processQueuedEvents(js::Value). Built by oracle/Makefile (it needs the reference headers)
// into oracle/_ref/facade_host; tests/test_facade.py drives it.
//
//   facade_host <batch.json> <blocks> <nOut> <out.f32> [device=0] [sampleRate=44100] [mode=process|blocks]
// device -1: dry handle (host logic only; nothing is rendered, the output file holds zeros).
// mode blocks: the whole render through ONE processBlocks call (elemhip_process_blocks_host: planar host arrays).
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iterator>
#include <list>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include <elemhip/Runtime.hpp>   // pulls <elem/Value.h>, <elem/JSON.h>, <elem/GraphNode.h> from the reference tree
#include <elem/AudioBufferResource.h>
#include <SampleTime.h>
#include <Metro.h>

#ifndef ELEMHIP_HAVE_ELEM_HEADERS
#error "this host must be built with the reference headers on the include path"
#endif

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s batch.json blocks nOut out.f32 [device] [sampleRate]\n", argv[0]); return 2; }
    const size_t blocks = std::stoul(argv[2]), nOut = std::stoul(argv[3]);
    const int device = argc > 5 ? std::stoi(argv[5]) : 0;
    const double sr = argc > 6 ? std::stod(argv[6]) : 44100.0;
    const bool blocksMode = argc > 7 && std::string(argv[7]) == "blocks";
    std::ifstream f(argv[1]);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());

    elemhip::Runtime<float> runtime(sr, 512, device);
    int rc = runtime.registerNodeType("cpumetro", [](elem::NodeId id, double fs, int bs) { return std::make_shared<elem::MetronomeNode<float>>(id, fs, bs); });
    if (rc) { std::fprintf(stderr, "registerNodeType: %d\n", rc); return 1; }
    rc = runtime.registerNodeType("cputime", [](elem::NodeId id, double fs, int bs) { return std::make_shared<elem::SampleTimeNode<float>>(id, fs, bs); });
    if (rc) { std::fprintf(stderr, "registerNodeType: %d\n", rc); return 1; }
    const int dup = runtime.registerNodeType("cpumetro", [](elem::NodeId id, double fs, int bs) { return std::make_shared<elem::MetronomeNode<float>>(id, fs, bs); });
    const int dupBuiltin = runtime.registerNodeType("phasor", [](elem::NodeId id, double fs, int bs) { return std::make_shared<elem::MetronomeNode<float>>(id, fs, bs); });

    std::vector<float> table(64);
    for (size_t i = 0; i < table.size(); ++i) table[i] = (float)i / 64.0f;
    const float* chans[1] = {table.data()};
    // Runtime::addSharedResource(name, std::unique_ptr<SharedResource>) as the reference's hosts call it (Runtime.h:83)
    const bool added = runtime.addSharedResource("ramp", std::make_unique<elem::AudioBufferResource>(table.data(), table.size()));
    const bool addedTwice = runtime.addSharedResource("ramp", chans, 1, table.size());

    rc = runtime.applyInstructions(elem::js::parseJSON(text).getArray());          // cli/Benchmark.cpp:41
    if (rc) { std::fprintf(stderr, "applyInstructions: %s\n", elemhip_describe(rc)); return 1; }

    std::vector<std::vector<float>> scratch(nOut, std::vector<float>(512));
    std::vector<float*> ptrs;
    for (auto& s : scratch) ptrs.push_back(s.data());
    std::ofstream out(argv[4], std::ios::binary);
    size_t events = 0;
    if (blocksMode) {
        std::vector<std::vector<float>> whole(nOut, std::vector<float>(blocks * 512));
        std::vector<float*> wp;
        for (auto& s : whole) wp.push_back(s.data());
        int64_t t0 = 0;
        if (device >= 0) { rc = runtime.processBlocks(nullptr, 0, wp.data(), nOut, blocks * 512, &t0); if (rc) { std::fprintf(stderr, "processBlocks: %d\n", rc); return 1; } }
        for (size_t b = 0; b < blocks; ++b)
            for (auto& s : whole) out.write(reinterpret_cast<const char*>(s.data() + b * 512), 512 * sizeof(float));
    } else
    for (size_t b = 0; b < blocks; ++b) {
        int64_t t = (int64_t)(b * 512);
        if (device >= 0) runtime.process(nullptr, 0, ptrs.data(), nOut, 512, &t);    // userData = &sampleTime (wasm/Main.cpp:212)
        for (auto& s : scratch) out.write(reinterpret_cast<const char*>(s.data()), 512 * sizeof(float));
        runtime.processQueuedEvents([&](std::string const&, elem::js::Value) { ++events; });
    }
    auto snap = runtime.snapshot();
    auto keys = runtime.getSharedResourceMapKeys();
    auto pruned = runtime.gc();
    runtime.reset();
    std::printf("{\"dup\": %d, \"dup_builtin\": %d, \"added\": %d, \"added_twice\": %d, \"snapshot_nodes\": %zu, \"resource_keys\": %zu, \"first_key\": \"%s\", \"pruned\": %zu, \"events\": %zu}\n",
                dup, dupBuiltin, (int)added, (int)addedTwice, snap.size(), keys.size(), keys.empty() ? "" : keys[0].c_str(), pruned.size(), events);
    return 0;
}
