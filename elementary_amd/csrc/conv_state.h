// conv_state.h — layout of a `convolve` node's device state (conv.hip, conv_long.inc, engine.cpp, island.inc's epilogue).
// Kept out of device.h since r05: device.h is part of the text the run-time compiler sees (jit.cpp), so every edit there re-keys the
// whole on-disk kernel cache; nothing in the specialised island kernels looks at a convolver's state.
#pragma once
#include <stdint.h>

namespace elemhip {

// Convolver state: one device allocation per `path` assignment, zero-initialised except H.
//   header[16] | H[P][512] float2 | X[P][512] float2 | pre[2][S][512] float2 | preSlow[512] float2 | inbuf[512] | overlap[512]
// Spectra are 1024-point real-FFT bins 0..511 with the (real) Nyquist bin packed into bin 0's imaginary part.
namespace conv {
enum : uint32_t {
    kBlock = 512, kFft = 1024, kBinGroups = 8, kMaxSlices = 8, kSlicePartitions = 96,
    H_P = 0,            // partitions = ceil(trimmed IR length / 512)
    H_S = 1,            // partial-sum slices the helpers produce
    H_FILL = 2, H_BLK = 3,             // frames already in the current input block / index of that block
    H_FILL_NEXT = 4, H_BLK_NEXT = 5,   // written by the node's main workgroup, committed by the epilogue
    H_PREVALID0 = 6,    // [2]: block index whose older-partition sum pre[i] holds
    H_PRESLOW_FOR = 8,  // block index preSlow holds (only when a call straddles two input blocks)
    H_Q = 9,            // long partitions (4096 samples) of the IR, 0: the node has no long-partition spectra (conv_long.inc)
    H_HISTBLKS = 10,    // blocks of the time-domain input ring behind `overlap` (0: none)
    H_OVL_STALE = 11,   // a long-partition launch set rendered last: `overlap` is not what the next 512-partition evaluation needs (conv_long.inc)
    H_UID = 12,         // a number no other convolver state of this engine carries (never 0): whose spectra a launch set's scratch ring holds (conv_long.inc)
    kHeaderDwords = 16,
};
}

} // namespace elemhip
