// Build stub (ours, not reference code) for the un-vendored signalsmith-stretch
// submodule that runtime/elem/builtins/SampleSeq.h:8 includes unconditionally.
// Only `sampleseq2` / `mc.sampleseq2` (out of scope) touch
// it; this stub just lets the reference headers compile for oracle/_ref.
#pragma once
namespace signalsmith { namespace stretch {
template <class Sample>
struct SignalsmithStretch {
    void presetDefault(int, Sample) {}
    void setTransposeSemitones(Sample) {}
    template <class In, class Out>
    void process(In&&, int, Out&&, int) {}
};
}}
