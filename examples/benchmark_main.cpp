// examples/benchmark_main.cpp — the timing protocol of the reference's cli benchmark
// (cli/Benchmark.cpp:31-112: Runtime(44100, 512), apply the instruction batch, 1 warm-up block,
// N timed 512-frame process() calls into 2 scratch channels, report total / average) on the HIP
// engine.  The reference evaluates a JS bundle in QuickJS to obtain the batch; here the batch is
// read from a JSON file (e.g. written by `python -m elementary_amd.tools dump c1 batch.json`).
//
//   g++ -std=c++17 -O2 -Iinclude examples/benchmark_main.cpp -Lelementary_amd -lelemhip -Wl,-rpath,$PWD/elementary_amd -o examples/bench_cli
// (`make -C elementary_amd/csrc` builds it; tests/test_gpu_parity.py::test_cli_benchmark_host runs it on the GPU box)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <fstream>
#include <iterator>
#include <numeric>
#include <string>
#include <vector>

#include <elemhip/Runtime.hpp>

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s batch.json [blocks=10000] [sampleRate=44100] [last_block.f32] [all_blocks.f32]\n", argv[0]); return 2; }
    const size_t blocks = argc > 2 ? std::stoul(argv[2]) : 10000;
    const double sr = argc > 3 ? std::stod(argv[3]) : 44100.0;
    std::ifstream f(argv[1]);
    std::string batch((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());

    elemhip::Runtime<float> runtime(sr, 512);
    // (r06, for graphs other than the cli benchmark's own — C3's convolution reverb: ELEMHIP_BENCH_RES="name=file.f32;name=file.f32"
    //  registers mono shared resources before the batch is applied, ELEMHIP_BENCH_IO="<inputs>,<outputs>" sets the channel counts of
    //  the process() calls; inputs are seeded noise, refreshed outside the timed calls)
    if (const char* res = std::getenv("ELEMHIP_BENCH_RES")) {
        std::string all(res);
        for (size_t at = 0; at < all.size();) {
            size_t end = all.find(';', at); if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(at, end - at); at = end + 1;
            const size_t eq = item.find('=');
            if (eq == std::string::npos) continue;
            std::ifstream rf(item.substr(eq + 1), std::ios::binary);
            std::vector<char> raw((std::istreambuf_iterator<char>(rf)), std::istreambuf_iterator<char>());
            std::vector<float> data(raw.size() / sizeof(float));
            std::memcpy(data.data(), raw.data(), data.size() * sizeof(float));
            const float* chan = data.data();
            if (!runtime.addSharedResource(item.substr(0, eq), &chan, 1, data.size())) { std::fprintf(stderr, "addSharedResource(%s) failed\n", item.c_str()); return 1; }
        }
    }
    const int rc = runtime.applyInstructionsJSON(batch);
    if (rc) { std::fprintf(stderr, "applyInstructions: %s\n", elemhip_describe(rc)); return 1; }
    size_t nIn = 0, nOut = 2;
    if (const char* io = std::getenv("ELEMHIP_BENCH_IO")) { unsigned a = 0, b = 2; if (std::sscanf(io, "%u,%u", &a, &b) == 2) { nIn = a; nOut = b; } }

    std::vector<std::vector<float>> scratch(nOut, std::vector<float>(512)), inputs(nIn, std::vector<float>(512));
    std::vector<float*> ptrs;
    for (auto& c : scratch) ptrs.push_back(c.data());
    std::vector<const float*> inPtrs;
    for (auto& c : inputs) inPtrs.push_back(c.data());
    uint32_t lcg = 12345u;
    auto refill = [&] { for (auto& c : inputs) for (auto& v : c) { lcg = lcg * 1664525u + 1013904223u; v = ((float)(lcg >> 8) / 8388608.0f - 1.0f) * 0.25f; } };
    refill();
    // (a checker may ask for EVERY rendered block, warm-up included: [blocks + 1][2][512] floats, kept outside the timed calls)
    std::vector<float> all;
    const bool keepAll = argc > 5;
    if (keepAll) all.reserve((blocks + 1) * nOut * 512);
    auto keep = [&] { if (keepAll) for (auto& c : scratch) all.insert(all.end(), c.begin(), c.end()); };
    runtime.process(nIn ? inPtrs.data() : nullptr, nIn, ptrs.data(), nOut, 512, nullptr);                 // warm-up block (:70-77)
    keep();

    std::vector<double> deltas;
    deltas.reserve(blocks);
    for (size_t i = 0; i < blocks; ++i) {
        if (nIn) refill();
        auto t0 = std::chrono::steady_clock::now();
        runtime.process(nIn ? inPtrs.data() : nullptr, nIn, ptrs.data(), nOut, 512, nullptr);
        auto t1 = std::chrono::steady_clock::now();
        deltas.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());   // ns resolution, not truncated
        keep();
    }
    if (keepAll) { std::ofstream o(argv[5], std::ios::binary); o.write(reinterpret_cast<const char*>(all.data()), (std::streamsize)(all.size() * sizeof(float))); }
    const double sum = std::accumulate(deltas.begin(), deltas.end(), 0.0);
    {   // + the distribution, as one JSON line on stderr (benchmarks/bench_configs.py c1 reads it)
        std::vector<double> sorted(deltas);
        std::sort(sorted.begin(), sorted.end());
        auto pct = [&](double q) { return sorted.empty() ? 0.0 : sorted[std::min(sorted.size() - 1, (size_t)(q * (double)sorted.size()))]; };
        std::fprintf(stderr, "{\"host\": \"native C++ over include/elemhip/Runtime.hpp\", \"blocks\": %zu, \"us_mean\": %.3f, \"us_p50\": %.3f, \"us_p99\": %.3f, \"us_max\": %.3f}\n",
                     deltas.size(), sum / (double)std::max<size_t>(1, deltas.size()), pct(0.5), pct(0.99), sorted.empty() ? 0.0 : sorted.back());
    }
    if (argc > 4 && argv[4][0]) {   // the last rendered block (outputs x 512 floats), for a checker
        std::ofstream o(argv[4], std::ios::binary);
        for (auto& c : scratch) o.write(reinterpret_cast<const char*>(c.data()), 512 * sizeof(float));
    }
    std::printf("[Running float]:\nTotal run time: %.0fus (%.3fs)\nAverage iteration time: %.2fus\nDone\n", sum, sum / 1e6, sum / deltas.size());
    return 0;
}
