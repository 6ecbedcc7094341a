"""Deterministic synthetic graphs for the BASELINE.json configs.

Each builder returns the list of per-channel root signals to hand to ``Runtime.render`` (which
wraps them in ``root`` nodes exactly like the reference frontend, Reconciler.res:88-100).
"""
from __future__ import annotations

import math
from typing import List

from . import el
from .reconciler import NodeRepr


# ---- C1: cli/BenchmarkMain shape — 2 ch el.lowpass(800, 1, el.mul(0.3, el.cycle(440 + c))) ----
def c1_graph() -> List[NodeRepr]:
    return [el.lowpass(800, 1, el.mul(0.3, el.cycle(440 + c))) for c in range(2)]


C1_SAMPLE_RATE = 44100.0


def c1_algorithmic_bytes(block: int = 512) -> int:
    """4 shared consts x 1 + per channel [const 1 + phasor 2 + mul 3 + sin 2 + mul 3 + svf 4 + root 2] = 38
    buffer touches x 2 KB + the 4 KB bus = 80 KB per block."""
    return (4 + 2 * 17) * block * 4 + 2 * block * 4


# ---- C2: 256-voice subtractive synth, 4107 nodes, sr 48 kHz -----------------------------------
C2_SAMPLE_RATE = 48000.0
C2_POLE = 0.9995


def c2_voice_params(v: int):
    f = 55.0 * 2.0 ** ((v % 48) / 12.0) * (1.0 + 0.0007 * (v // 48))
    r = 1.0 + 0.25 * (v % 16)
    return f, r


def c2_voice(v: int) -> NodeRepr:
    f, r = c2_voice_params(v)
    p = C2_POLE
    # the three per-voice consts are keyed so the reconciler's structural sharing
    # (NodeRepr.res:34-54) cannot merge the envelope chains of voices with equal rates:
    # every voice keeps its own 13 ops + 3 consts, 4107 nodes in total
    s1 = el.blepsaw(el.const({"key": f"v{v}:f1", "value": f}))
    s2 = el.blepsaw(el.const({"key": f"v{v}:f2", "value": 1.003 * f}))
    osc = el.add(s1, s2)
    gate = el.le(el.phasor(el.const({"key": f"v{v}:r", "value": r})), 0.5)   # el.train (lib/oscillators.ts:27-29)
    e = el.pole(p, el.mul(gate, 1.0 - p))           # el.smooth core (lib/filters.ts:28-31)
    fc = el.add(200, el.mul(e, 4000))
    y = el.svf({"mode": "lowpass"}, fc, 2, osc)
    return el.mul(0.05, el.tanh(el.mul(y, e)))


def c2_graph(voices: int = 256, channels: int = 2, first_voice: int = 0) -> List[NodeRepr]:
    """Channel c sums the voices v with v % channels == c (left fold, Math.h:75-85)."""
    outs = []
    for c in range(channels):
        vs = [c2_voice(v) for v in range(first_voice, first_voice + voices) if v % channels == c]
        outs.append(el.add(*vs) if len(vs) > 1 else vs[0])
    return outs


def c2_algorithmic_bytes(voices: int = 256, channels: int = 2, block: int = 512) -> int:
    """sum(fanIn + outs) * block * 4 B per block + bus RMW."""
    per_voice = 39
    mix = channels * (voices // channels + 1) + channels * 2 + 7
    return (voices * per_voice + mix) * block * 4 + channels * block * 4


def c2_level_algorithmic_bytes(voices: int = 256, channels: int = 2, block: int = 512):
    """The same figure split by launch level: [voice islands, mixers + roots] (the bus RMW belongs to the epilogue)."""
    mix = channels * (voices // channels + 1) + channels * 2 + 7
    return [voices * 39 * block * 4, mix * block * 4]


# ---- C4: independent offline-render instances (mixed filter + delay chains), 16 nodes each ----
C4_SAMPLE_RATE = 48000.0


def _rbj_lowshelf(fc: float, gain_db: float, sr: float):
    A = 10.0 ** (gain_db / 40.0)
    w0 = 2.0 * math.pi * fc / sr
    cw, sw = math.cos(w0), math.sin(w0)
    alpha = sw / 2.0 * math.sqrt(2.0)
    sq = 2.0 * math.sqrt(A) * alpha
    b0 = A * ((A + 1) - (A - 1) * cw + sq)
    b1 = 2 * A * ((A - 1) - (A + 1) * cw)
    b2 = A * ((A + 1) - (A - 1) * cw - sq)
    a0 = (A + 1) + (A - 1) * cw + sq
    a1 = -2 * ((A - 1) + (A + 1) * cw)
    a2 = (A + 1) + (A - 1) * cw - sq
    return b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0


def c4_instance(k: int) -> NodeRepr:
    fc = 300.0 * 2.0 ** ((k % 40) / 10.0)
    length = 2400 + 37 * (k % 256)
    b0, b1, b2, a1, a2 = _rbj_lowshelf(250.0, 3.0, C4_SAMPLE_RATE)
    src = el.rand({"seed": k + 1})
    if k % 2 == 0:
        x = el.svf({"mode": "lowpass"}, fc, 0.7, src)
        x = el.delay({"size": 24000}, length, 0.5, x)
        x = el.biquad(b0, b1, b2, a1, a2, x)
    else:   # "mixed": odd instances swap the svf / biquad order
        x = el.biquad(b0, b1, b2, a1, a2, src)
        x = el.delay({"size": 24000}, length, 0.5, x)
        x = el.svf({"mode": "lowpass"}, fc, 0.7, x)
    x = el.sdelay({"size": 1000 + k}, x)
    return el.tanh(x)


def c4_algorithmic_bytes(instances: int, block: int = 512) -> int:
    return instances * (31 * block * 4 + block * 4)


# ---- C3: 8-channel FFT convolution reverb, 2 s impulse responses at 48 kHz ----------
C3_SAMPLE_RATE = 48000.0
C3_CHANNELS = 8
C3_IR_LEN = 96000
C3_IR_DECAY = 0.9999375019530843        # exp(-1/16000) per sample


def _lcg_stream(seed: int, n: int):
    import numpy as np
    out = np.empty(n, dtype=np.float64)
    s = seed & 0xFFFFFFFF
    for i in range(n):
        s = (1664525 * s + 1013904223) & 0xFFFFFFFF
        out[i] = s / 2147483648.0 - 1.0
    return out


def c3_impulse_response(ch: int, length: int = C3_IR_LEN):
    """ir[n] = lcg(n) * r^n, ir[0] = 1, unit energy, float32; distinct LCG seed per channel."""
    import numpy as np
    ir = _lcg_stream(1 + ch, length) * np.power(C3_IR_DECAY, np.arange(length, dtype=np.float64))
    ir[0] = 1.0
    return (ir / np.sqrt(np.sum(ir * ir))).astype(np.float32)


def c3_graph(channels: int = C3_CHANNELS) -> List[NodeRepr]:
    """per channel: root(ch)(convolve{path: "ir<ch>"}(in{channel: ch}))"""
    return [el.convolve({"path": f"ir{ch}"}, el.in_({"channel": ch})) for ch in range(channels)]


def c3_input(channels: int, frames: int, amp: float = 0.25):
    import numpy as np
    return np.stack([(_lcg_stream(101 + ch, frames) * amp).astype(np.float32) for ch in range(channels)])


def c3_algorithmic_bytes(channels: int = C3_CHANNELS, ir_len: int = C3_IR_LEN, block: int = 512) -> int:
    """per channel-block the reference's two-stage partitioning reads (8 + 8) spectra
    of 513 bins and 1/8 of the 4096-partition tail spectra (8 B per bin), plus 4 KB of block I/O."""
    tail = max(0, ir_len - 8192)
    tail_parts = (tail + 4095) // 4096
    head_parts = min(16, (min(ir_len, 8192) + 511) // 512)
    per_ch = head_parts * 513 * 8 + tail_parts * 4097 * 8 // 8 + 2 * block * 4
    return channels * per_ch
