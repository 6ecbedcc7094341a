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
    const int rc = runtime.applyInstructionsJSON(batch);
    if (rc) { std::fprintf(stderr, "applyInstructions: %s\n", elemhip_describe(rc)); return 1; }

    std::vector<std::vector<float>> scratch(2, std::vector<float>(512));
    std::vector<float*> ptrs = {scratch[0].data(), scratch[1].data()};
    // (a checker may ask for EVERY rendered block, warm-up included: [blocks + 1][2][512] floats, kept outside the timed calls)
    std::vector<float> all;
    const bool keepAll = argc > 5;
    if (keepAll) all.reserve((blocks + 1) * 1024);
    auto keep = [&] { if (keepAll) for (auto& c : scratch) all.insert(all.end(), c.begin(), c.end()); };
    runtime.process(nullptr, 0, ptrs.data(), 2, 512, nullptr);                 // warm-up block (:70-77)
    keep();

    std::vector<double> deltas;
    deltas.reserve(blocks);
    for (size_t i = 0; i < blocks; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        runtime.process(nullptr, 0, ptrs.data(), 2, 512, nullptr);
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
    if (argc > 4) {   // the last rendered block (2 x 512 floats), for a checker
        std::ofstream o(argv[4], std::ios::binary);
        for (auto& c : scratch) o.write(reinterpret_cast<const char*>(c.data()), 512 * sizeof(float));
    }
    std::printf("[Running float]:\nTotal run time: %.0fus (%.3fs)\nAverage iteration time: %.2fus\nDone\n", sum, sum / 1e6, sum / deltas.size());
    return 0;
}
