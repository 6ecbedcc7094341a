// tests/native/fft4096_host.cpp — the transform core of the long-partition convolver (elementary_amd/csrc/fft4096.h) run on the HOST:
// the 256 threads of a workgroup are emulated phase by phase (every phase between two barriers is a loop over tid) and the real
// forward / inverse transforms of 8192 samples are checked against a double-precision DFT, and a whole overlap-save convolution
// step against a direct sum. Prints one JSON line; exit code 0 when every error is below its bound.
//   clang++ -std=c++17 -O2 -ffp-contract=off -I elementary_amd/csrc tests/native/fft4096_host.cpp -o /tmp/fft4096_host
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft4096.h"

using namespace lfft;
typedef std::complex<double> cd;

static std::vector<c2> gW, gTab;

static void fft4096_emulated(std::vector<c2>& buf) {           // buf: padded, holds the input at pad(i); the result lands the same way
    std::vector<c2> regs(256 * 16);
    for (uint32_t s = 0; s < 3; ++s) {
        for (uint32_t tid = 0; tid < 256; ++tid) {
            c2 v[16];
            pass_read(buf.data(), tid, v);
            pass_twiddle(v, tid, s, gTab.data());
            dft16(v);
            for (int r = 0; r < 16; ++r) regs[tid * 16 + r] = v[r];
        }
        for (uint32_t tid = 0; tid < 256; ++tid) {
            c2 v[16];
            for (int r = 0; r < 16; ++r) v[r] = regs[tid * 16 + r];
            pass_write(buf.data(), tid, s, v);
        }
    }
}

static void real_forward(const std::vector<float>& x, std::vector<c2>& U) {     // U[k] = 2 X[k], k = 0..4096
    std::vector<c2> buf(kBuf);
    for (uint32_t n = 0; n < M; ++n) buf[pad(n)] = mk(x[2 * n], x[2 * n + 1]);
    fft4096_emulated(buf);
    U.assign(kRow, mk(0, 0));
    for (uint32_t k = 0; k <= M / 2; ++k) {
        c2 a, b;
        split_forward(buf.data(), k, gW.data(), a, b);
        U[k] = a; U[M - k] = b;
    }
}

static void real_inverse(const std::vector<c2>& Y, std::vector<float>& out) {   // out = 8192 * irfft(Y)
    std::vector<c2> buf(kBuf);
    for (uint32_t k = 0; k <= M / 2; ++k) {
        c2 a, b;
        split_inverse(Y[k], Y[M - k], k, gW.data(), a, b);
        buf[pad(k)] = a;
        if (k) buf[pad(M - k)] = b;
    }
    fft4096_emulated(buf);
    out.assign(N, 0.0f);
    for (uint32_t n = 0; n < M; ++n) { const c2 f = buf[pad(n)]; out[2 * n] = f.x; out[2 * n + 1] = -f.y; }
}

int main() {
    gW.resize(N);
    for (uint32_t j = 0; j < N; ++j) { const double a = -2.0 * M_PI * (double)j / (double)N; gW[j] = mk((float)std::cos(a), (float)std::sin(a)); }
    gTab.resize(kTabSize);
    for (uint32_t i = 0; i < kTabSize; ++i) gTab[i] = gW[tab_source(i)];
    uint32_t seed = 12345u;
    auto rnd = [&] { seed = 1664525u * seed + 1013904223u; return (float)((double)seed / 2147483648.0 - 1.0); };
    // 1. forward transform vs double DFT
    std::vector<float> x(N);
    for (auto& v : x) v = 0.5f * rnd();
    std::vector<c2> U;
    real_forward(x, U);
    double errF = 0.0, peak = 0.0;
    for (uint32_t k = 0; k <= M; k += 1) {
        if (k % 7 != 0 && k != M && k != M / 2 && k > 8) continue;      // a sample of bins (the naive DFT is O(N) per bin)
        cd acc = 0;
        for (uint32_t n = 0; n < N; ++n) acc += (double)x[n] * std::polar(1.0, -2.0 * M_PI * (double)k * (double)n / (double)N);
        acc *= 2.0;
        errF = std::max(errF, std::abs(acc - cd(U[k].x, U[k].y)));
        peak = std::max(peak, std::abs(acc));
    }
    // 2. inverse of the forward gives 16384 x back
    std::vector<float> back;
    real_inverse(U, back);
    double errI = 0.0;
    for (uint32_t n = 0; n < N; ++n) errI = std::max(errI, std::fabs((double)back[n] / 16384.0 - (double)x[n]));
    // 3. one overlap-save step: y = (g * x)[4096..8191] for a 4096-tap g, through G = rfft([g | 0]) / 16384 (double, rounded to float)
    std::vector<float> g(4096);
    double e2 = 0.0;
    for (uint32_t i = 0; i < 4096; ++i) { g[i] = rnd() * (float)std::exp(-(double)i / 1500.0); e2 += (double)g[i] * g[i]; }
    for (auto& v : g) v = (float)(v / std::sqrt(e2));
    std::vector<c2> G(kRow, mk(0, 0));
    for (uint32_t k = 0; k <= M; ++k) {
        cd acc = 0;
        for (uint32_t n = 0; n < 4096; ++n) acc += (double)g[n] * std::polar(1.0, -2.0 * M_PI * (double)k * (double)n / (double)N);
        G[k] = mk((float)(acc.real() / 16384.0), (float)(acc.imag() / 16384.0));
    }
    std::vector<c2> Y(kRow);
    for (uint32_t k = 0; k <= M; ++k) Y[k] = cmul(G[k], U[k]);
    std::vector<float> y;
    real_inverse(Y, y);
    double errC = 0.0, peakC = 0.0;
    for (uint32_t n = 4096; n < N; n += 13) {
        double acc = 0.0;
        for (uint32_t t = 0; t < 4096; ++t) acc += (double)g[t] * (double)x[n - t];
        errC = std::max(errC, std::fabs(acc - (double)y[n]));
        peakC = std::max(peakC, std::fabs(acc));
    }
    const bool ok = errF <= 2e-5 * peak && errI <= 2e-6 && errC <= 5e-7;
    std::printf("{\"forward_max_err\": %.3e, \"forward_peak\": %.3e, \"roundtrip_max_err\": %.3e, \"overlap_save_max_err\": %.3e, \"overlap_save_peak\": %.3e, \"ok\": %s}\n",
                errF, peak, errI, errC, peakC, ok ? "true" : "false");
    return ok ? 0 : 1;
}
