// jitc_main.cpp — elemhip_jitc: one hiprtc compile in a process of its own.
//
// hiprtc / comgr serialise compilations inside a process (three concurrent hiprtcCompileProgram calls take 1x, 2x and 3x the time of
// one: tools measured it in r05, and the round's first GPU run showed it as 7 s "mean compile time" with eight worker threads queueing
// behind one lock). The engine's compile workers (jit.cpp) therefore hand each island shape to this helper: N workers = N compilers.
//   elemhip_jitc <source file> <output .hsaco>        exit 0: the code object was written; otherwise the compiler log is on stderr
// Same options as the in-process path (jit.cpp falls back to it when this binary is missing): gfx950, -O3, -ffp-contract=off.
#include <hip/hiprtc.h>

#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s source.hip out.hsaco\n", argv[0]); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const std::string src((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "elemhip_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { std::fprintf(stderr, "hiprtcCreateProgram failed\n"); return 3; }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-pragma-once-outside-header"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    size_t logSize = 0;
    (void)hiprtcGetProgramLogSize(prog, &logSize);
    if (logSize > 1) { std::string log(logSize, '\0'); (void)hiprtcGetProgramLog(prog, &log[0]); std::fputs(log.c_str(), stderr); }
    if (rc != HIPRTC_SUCCESS) return 4;
    size_t sz = 0;
    (void)hiprtcGetCodeSize(prog, &sz);
    std::vector<char> code(sz);
    (void)hiprtcGetCode(prog, code.data());
    std::ofstream o(argv[2], std::ios::binary);
    if (!o) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 5; }
    o.write(code.data(), (std::streamsize)code.size());
    o.close();
    return o ? 0 : 5;
}
