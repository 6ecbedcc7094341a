// fftconv_oracle.h — TEST INFRASTRUCTURE (CPU checker), not part of the product path.
//
// Restates the algorithm behind the reference's `convolve` node (wasm/Convolve.h:23-92), which calls
// `fftconvolver::TwoStageFFTConvolver::init(512, 4096, ir, len)` (:49) and `process(in, out, n)` (:73-84).
// That class lives in the third-party HiFi-LoFi/FFTConvolver library, an UN-VENDORED, UN-PINNED git
// submodule of the reference (.gitmodules:1-3 -> wasm/FFTConvolver, empty in the checkout), so there is
// no source file:line to cite. What is restated here is the library's published algorithm
// (FFTConvolver.cpp / TwoStageFFTConvolver.cpp, MIT licence), from its documentation and behaviour:
//
//   FFTConvolver  : uniformly partitioned overlap-save/-add convolver. blockSize B (power of two), segment
//                   size 2B, IR split into ceil(len/B) zero-padded segments, each FFT'd once. Trailing IR
//                   samples with |h| < 1e-6 are dropped first. Per call the (possibly partially filled)
//                   input block is FFT'd again, multiplied with IR segment 0 and added to the
//                   pre-multiplied sum of the older input spectra x the later IR segments (computed when a
//                   new block starts), inverse FFT'd, and the first half + saved overlap is emitted.
//   TwoStage      : head = FFTConvolver(B=512) over ir[0:4096); tail0 = FFTConvolver(512) over
//                   ir[4096:8192) whose output is buffered and presented one 4096-frame period later;
//                   tail = FFTConvolver(4096) over ir[8192:) run once per 4096 input frames and presented
//                   two periods later (the library's optional background thread; inline by default).
//
// The library's FFT (Ooura, float) is replaced by a plain iterative radix-2 float FFT, so results agree
// with the reference to float FFT rounding (~1e-7 relative), not bit for bit. PINNING: this restatement
// is checked against outputs recorded from the reference's own prebuilt wasm engine
// (tests/golden/convolve_wasm.f32, made by tests/golden/make_convolve_golden.js) in
// tests/test_convolve_oracle.py at <= 1e-6 abs.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <memory>
#include <vector>

namespace fftconv_oracle {

struct Fft {           // complex radix-2 DIT, float data, twiddles rounded from double
    size_t n = 0;
    std::vector<std::complex<float>> tw;
    std::vector<uint32_t> rev;
    void init(size_t size) {
        n = size; tw.resize(n / 2); rev.resize(n);
        for (size_t k = 0; k < n / 2; ++k) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
            tw[k] = std::complex<float>((float)std::cos(a), (float)std::sin(a));
        }
        size_t bits = 0; while ((size_t(1) << bits) < n) ++bits;
        for (size_t i = 0; i < n; ++i) { uint32_t r = 0; for (size_t b = 0; b < bits; ++b) if (i & (size_t(1) << b)) r |= 1u << (bits - 1 - b); rev[i] = r; }
    }
    void run(std::vector<std::complex<float>>& a, bool inverse) const {
        for (size_t i = 0; i < n; ++i) if (i < rev[i]) std::swap(a[i], a[rev[i]]);
        for (size_t len = 2; len <= n; len <<= 1) {
            const size_t half = len / 2, stride = n / len;
            for (size_t s = 0; s < n; s += len)
                for (size_t k = 0; k < half; ++k) {
                    std::complex<float> w = tw[k * stride];
                    if (inverse) w = std::conj(w);
                    const std::complex<float> u = a[s + k];
                    const std::complex<float> x = a[s + k + half];
                    const std::complex<float> v(x.real() * w.real() - x.imag() * w.imag(), x.real() * w.imag() + x.imag() * w.real());
                    a[s + k] = u + v; a[s + k + half] = u - v;
                }
        }
    }
};

struct Spectrum { std::vector<float> re, im; void resize(size_t n) { re.assign(n, 0.0f); im.assign(n, 0.0f); } };

struct FftConvolver {
    size_t blockSize = 0, segSize = 0, segCount = 0, bins = 0, current = 0, fill = 0;
    Fft fft;
    std::vector<std::complex<float>> work;
    std::vector<Spectrum> segments, segmentsIR;
    Spectrum pre, conv;
    std::vector<float> overlap, inputBuffer, timeBuf;

    void forward(const float* x, size_t count, Spectrum& out) {     // zero-padded real FFT of `count` samples
        for (size_t i = 0; i < segSize; ++i) work[i] = std::complex<float>(i < count ? x[i] : 0.0f, 0.0f);
        fft.run(work, false);
        for (size_t k = 0; k < bins; ++k) { out.re[k] = work[k].real(); out.im[k] = work[k].imag(); }
    }
    void inverse(const Spectrum& in) {                              // -> timeBuf[segSize], scaled by 1/segSize
        for (size_t k = 0; k < bins; ++k) work[k] = std::complex<float>(in.re[k], in.im[k]);
        for (size_t k = bins; k < segSize; ++k) work[k] = std::conj(work[segSize - k]);
        fft.run(work, true);
        const float s = 1.0f / (float)segSize;
        for (size_t i = 0; i < segSize; ++i) timeBuf[i] = work[i].real() * s;
    }
    static void mac(Spectrum& acc, const Spectrum& a, const Spectrum& b, size_t bins) {
        for (size_t k = 0; k < bins; ++k) {
            acc.re[k] += a.re[k] * b.re[k] - a.im[k] * b.im[k];
            acc.im[k] += a.re[k] * b.im[k] + a.im[k] * b.re[k];
        }
    }
    bool init(size_t block, const float* ir, size_t irLen) {
        *this = FftConvolver();
        if (block == 0) return false;
        while (irLen > 0 && std::fabs(ir[irLen - 1]) < 0.000001f) --irLen;
        if (irLen == 0) return true;
        blockSize = 1; while (blockSize < block) blockSize <<= 1;
        segSize = 2 * blockSize; segCount = (irLen + blockSize - 1) / blockSize; bins = segSize / 2 + 1;
        fft.init(segSize); work.resize(segSize); timeBuf.assign(segSize, 0.0f);
        segments.resize(segCount); segmentsIR.resize(segCount);
        for (size_t i = 0; i < segCount; ++i) {
            segments[i].resize(bins); segmentsIR[i].resize(bins);
            const size_t remaining = irLen - i * blockSize;
            forward(ir + i * blockSize, std::min(remaining, blockSize), segmentsIR[i]);
        }
        pre.resize(bins); conv.resize(bins);
        overlap.assign(blockSize, 0.0f); inputBuffer.assign(blockSize, 0.0f);
        return true;
    }
    void process(const float* input, float* output, size_t len) {
        if (segCount == 0) { std::fill_n(output, len, 0.0f); return; }
        size_t processed = 0;
        while (processed < len) {
            const bool wasEmpty = (fill == 0);
            const size_t processing = std::min(len - processed, blockSize - fill);
            const size_t pos = fill;
            std::memcpy(inputBuffer.data() + pos, input + processed, processing * sizeof(float));
            forward(inputBuffer.data(), blockSize, segments[current]);
            if (wasEmpty) {
                pre.resize(bins);
                for (size_t i = 1; i < segCount; ++i) mac(pre, segmentsIR[i], segments[(current + i) % segCount], bins);
            }
            conv = pre;
            mac(conv, segments[current], segmentsIR[0], bins);
            inverse(conv);
            for (size_t i = 0; i < processing; ++i) output[processed + i] = timeBuf[pos + i] + overlap[pos + i];
            fill += processing;
            if (fill == blockSize) {
                std::fill(inputBuffer.begin(), inputBuffer.end(), 0.0f); fill = 0;
                std::memcpy(overlap.data(), timeBuf.data() + blockSize, blockSize * sizeof(float));
                current = (current > 0) ? (current - 1) : (segCount - 1);
            }
            processed += processing;
        }
    }
};

struct TwoStageConvolver {
    size_t headBlock = 0, tailBlock = 0, tailInputFill = 0, precalcPos = 0;
    FftConvolver head, tail0, tail;
    std::vector<float> tailOutput0, tailPrecalc0, tailOutput, tailPrecalc, tailInput, bgInput;

    bool init(size_t headBlockSize, size_t tailBlockSize, const float* ir, size_t irLen) {
        *this = TwoStageConvolver();
        if (headBlockSize == 0 || tailBlockSize == 0) return false;
        headBlockSize = std::max<size_t>(1, headBlockSize);
        if (headBlockSize > tailBlockSize) std::swap(headBlockSize, tailBlockSize);
        while (irLen > 0 && std::fabs(ir[irLen - 1]) < 0.000001f) --irLen;
        if (irLen == 0) return true;
        headBlock = 1; while (headBlock < headBlockSize) headBlock <<= 1;
        tailBlock = 1; while (tailBlock < tailBlockSize) tailBlock <<= 1;
        head.init(headBlock, ir, std::min(irLen, tailBlock));
        if (irLen > tailBlock) {
            tail0.init(headBlock, ir + tailBlock, std::min(irLen - tailBlock, tailBlock));
            tailOutput0.assign(tailBlock, 0.0f); tailPrecalc0.assign(tailBlock, 0.0f);
        }
        if (irLen > 2 * tailBlock) {
            tail.init(tailBlock, ir + 2 * tailBlock, irLen - 2 * tailBlock);
            tailOutput.assign(tailBlock, 0.0f); tailPrecalc.assign(tailBlock, 0.0f); bgInput.assign(tailBlock, 0.0f);
        }
        if (!tailPrecalc0.empty() || !tailPrecalc.empty()) tailInput.assign(tailBlock, 0.0f);
        return true;
    }
    void process(const float* input, float* output, size_t len) {
        head.process(input, output, len);
        if (tailInput.empty()) return;
        size_t processed = 0;
        while (processed < len) {
            const size_t processing = std::min(len - processed, headBlock - (tailInputFill % headBlock));
            if (!tailPrecalc0.empty()) for (size_t i = 0; i < processing; ++i) output[processed + i] += tailPrecalc0[precalcPos + i];
            if (!tailPrecalc.empty())  for (size_t i = 0; i < processing; ++i) output[processed + i] += tailPrecalc[precalcPos + i];
            precalcPos += processing;
            std::memcpy(tailInput.data() + tailInputFill, input + processed, processing * sizeof(float));
            tailInputFill += processing;
            if (!tailPrecalc0.empty() && tailInputFill % headBlock == 0) {
                const size_t off = tailInputFill - headBlock;
                tail0.process(tailInput.data() + off, tailOutput0.data() + off, headBlock);
                if (tailInputFill == tailBlock) tailPrecalc0.swap(tailOutput0);
            }
            if (!tailPrecalc.empty() && tailInputFill == tailBlock) {
                tailPrecalc.swap(tailOutput);
                bgInput = tailInput;
                tail.process(bgInput.data(), tailOutput.data(), tailBlock);
            }
            if (tailInputFill == tailBlock) { tailInputFill = 0; precalcPos = 0; }
            processed += processing;
        }
    }
};

} // namespace fftconv_oracle
