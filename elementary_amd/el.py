"""``el.*`` — Python mirror of the reference's JS standard library for the hot-path nodes.

Each function builds the same node tree (same kinds, props and child order, hence the same
hashes and the same instruction batches) as the reference frontend:
  js/packages/core/lib/core.ts, math.ts, oscillators.ts, filters.ts, signals.ts,
  envelopes.ts, dynamics.ts.
Composite helpers (cycle, train, smooth, adsr, ...) expand to primitives exactly as the
reference does, e.g. ``cycle(f) = sin(mul(2π, phasor(f)))`` (lib/oscillators.ts:39-41).
"""
from __future__ import annotations

import math
from typing import List, Any, Dict, Optional

from .reconciler import ElemNode, NodeRepr, create_node, resolve, unpack  # noqa: F401


def _n(kind: str, props: Optional[Dict[str, Any]], *children: ElemNode) -> NodeRepr:
    return create_node(kind, props or {}, children)


# ---- lib/core.ts ---------------------------------------------------------------------
def const(props: Dict[str, Any]) -> NodeRepr:
    return _n("const", props)


constant = const


def sr() -> NodeRepr:
    return _n("sr", {})


def time() -> NodeRepr:
    return _n("time", {})


def counter(gate: ElemNode) -> NodeRepr:
    return _n("counter", {}, gate)


def accum(xn: ElemNode, reset: ElemNode) -> NodeRepr:
    return _n("accum", {}, xn, reset)


def phasor(rate: ElemNode) -> NodeRepr:
    return _n("phasor", {}, rate)


def syncphasor(rate: ElemNode, reset: ElemNode) -> NodeRepr:
    return _n("sphasor", {}, rate, reset)


def latch(t: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("latch", {}, t, x)


def maxhold(props: Dict[str, Any], x: ElemNode, reset: ElemNode) -> NodeRepr:
    return _n("maxhold", props, x, reset)


def once(props: Dict[str, Any], x: ElemNode) -> NodeRepr:
    return _n("once", props, x)


def rand(props: Optional[Dict[str, Any]] = None) -> NodeRepr:
    return _n("rand", props or {})


def metro(props: Optional[Dict[str, Any]] = None) -> NodeRepr:
    return _n("metro", props or {})


def convolve(props: Dict[str, Any], x: ElemNode) -> NodeRepr:
    return _n("convolve", props, x)


def seq(props: Dict[str, Any], trigger: ElemNode, reset: ElemNode) -> NodeRepr:
    return _n("seq", props, trigger, reset)


def sample(props: Dict[str, Any], trigger: ElemNode, rate: ElemNode) -> NodeRepr:
    return _n("sample", props, trigger, rate)


def meter(props: Dict[str, Any], x: ElemNode) -> NodeRepr:
    return _n("meter", props, x)


def scope(props: Dict[str, Any], *args: ElemNode) -> NodeRepr:
    return _n("scope", props, *args)


def snapshot(props: Dict[str, Any], trigger: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("snapshot", props, trigger, x)


def table(props: Dict[str, Any], t: ElemNode) -> NodeRepr:
    return _n("table", props, t)


def seq2(props: Dict[str, Any], trigger: ElemNode, reset: ElemNode) -> NodeRepr:
    return _n("seq2", props, trigger, reset)


def sparseq2(props: Dict[str, Any], t: ElemNode) -> NodeRepr:
    return _n("sparseq2", props, t)


def sparseq(props: Dict[str, Any], trigger: ElemNode, reset: ElemNode) -> NodeRepr:
    """core.ts:131-144: props {seq: [{value, tickTime}], offset, loop: False | [start, end], follow, interpolate, tickInterval}."""
    return _n("sparseq", props, trigger, reset)


def capture(props: Dict[str, Any], g: ElemNode, x: ElemNode) -> NodeRepr:
    """core.ts:348-356: records x while the gate g is non-zero; the recording arrives as a "capture" event."""
    return _n("capture", props, g, x)


def sampleseq(props: Dict[str, Any], t: ElemNode) -> NodeRepr:
    return _n("sampleseq", props, t)


def pole(p: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("pole", {}, p, x)


def env(atk_pole: ElemNode, rel_pole: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("env", {}, atk_pole, rel_pole, x)


def z(x: ElemNode) -> NodeRepr:
    return _n("z", {}, x)


def delay(props: Dict[str, Any], length: ElemNode, fb: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("delay", props, length, fb, x)


def sdelay(props: Dict[str, Any], x: ElemNode) -> NodeRepr:
    return _n("sdelay", props, x)


def prewarp(fc: ElemNode) -> NodeRepr:
    return _n("prewarp", {}, fc)


def mm1p(props: Dict[str, Any], fc: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("mm1p", props, fc, x)


def svf(props: Dict[str, Any], fc: ElemNode, q: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("svf", props, fc, q, x)


def svfshelf(props: Dict[str, Any], fc: ElemNode, q: ElemNode, gain_db: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("svfshelf", props, fc, q, gain_db, x)


def biquad(b0: ElemNode, b1: ElemNode, b2: ElemNode, a1: ElemNode, a2: ElemNode, x: ElemNode) -> NodeRepr:
    return _n("biquad", {}, b0, b1, b2, a1, a2, x)


def tapIn(props: Dict[str, Any]) -> NodeRepr:
    return _n("tapIn", props)


def tapOut(props: Dict[str, Any], x: ElemNode) -> NodeRepr:
    return _n("tapOut", props, x)


# ---- lib/math.ts ---------------------------------------------------------------------
def identity(props: Dict[str, Any], x: Optional[ElemNode] = None) -> NodeRepr:
    """``el.in`` (math.ts:10-22): child only when a *node* is given."""
    if isinstance(x, NodeRepr):
        return _n("in", props, x)
    return _n("in", props)


in_ = identity


def _unary(kind: str):
    def f(x: ElemNode) -> NodeRepr:
        return _n(kind, {}, x)

    f.__name__ = kind
    return f


sin = _unary("sin")
cos = _unary("cos")
tan = _unary("tan")
tanh = _unary("tanh")
asinh = _unary("asinh")
ln = _unary("ln")
log = _unary("log")
log2 = _unary("log2")
ceil = _unary("ceil")
floor = _unary("floor")
round = _unary("round")  # noqa: A001 - mirrors el.round
sqrt = _unary("sqrt")
exp = _unary("exp")
abs = _unary("abs")  # noqa: A001


def _binary(kind: str):
    def f(a: ElemNode, b: ElemNode) -> NodeRepr:
        return _n(kind, {}, a, b)

    f.__name__ = kind
    return f


le = _binary("le")
leq = _binary("leq")
ge = _binary("ge")
geq = _binary("geq")
pow = _binary("pow")  # noqa: A001
eq = _binary("eq")
and_ = _binary("and")
or_ = _binary("or")


def _reducing(kind: str):
    def f(*args: ElemNode) -> NodeRepr:
        return _n(kind, {}, *args)

    f.__name__ = kind
    return f


add = _reducing("add")
sub = _reducing("sub")
mul = _reducing("mul")
div = _reducing("div")
mod = _reducing("mod")
min = _reducing("min")  # noqa: A001
max = _reducing("max")  # noqa: A001


# ---- lib/signals.ts ------------------------------------------------------------------
def ms2samps(t: ElemNode) -> NodeRepr:
    return mul(sr(), div(t, 1000.0))


def tau2pole(t: ElemNode) -> NodeRepr:
    return exp(div(-1.0, mul(t, sr())))


def db2gain(db: ElemNode) -> NodeRepr:
    return pow(10, mul(db, 1 / 20))


def select(g: ElemNode, a: ElemNode, b: ElemNode) -> NodeRepr:
    return add(mul(g, a), mul(sub(1, g), b))


def gain2db(gain: ElemNode) -> NodeRepr:
    return select(ge(gain, 0), max(-120, mul(20, log(gain))), -120)


def hann(t: ElemNode) -> NodeRepr:
    return mul(0.5, sub(1, cos(mul(2.0 * math.pi, t))))


# ---- lib/oscillators.ts --------------------------------------------------------------
def train(rate: ElemNode) -> NodeRepr:
    return le(phasor(rate), 0.5)


def cycle(rate: ElemNode) -> NodeRepr:
    return sin(mul(2.0 * math.pi, phasor(rate)))


def saw(rate: ElemNode) -> NodeRepr:
    return sub(mul(2, phasor(rate)), 1)


def square(rate: ElemNode) -> NodeRepr:
    return sub(mul(2, train(rate)), 1)


def triangle(rate: ElemNode) -> NodeRepr:
    return mul(2, sub(0.5, abs(saw(rate))))


def blepsaw(rate: ElemNode) -> NodeRepr:
    return _n("blepsaw", {}, rate)


def blepsquare(rate: ElemNode) -> NodeRepr:
    return _n("blepsquare", {}, rate)


def bleptriangle(rate: ElemNode) -> NodeRepr:
    return _n("bleptriangle", {}, rate)


def noise(props: Optional[Dict[str, Any]] = None) -> NodeRepr:
    return sub(mul(2, rand(props)), 1)


# ---- lib/filters.ts ------------------------------------------------------------------
def smooth(p: ElemNode, x: ElemNode) -> NodeRepr:
    return pole(p, mul(sub(1, p), x))


def sm(x: ElemNode) -> NodeRepr:
    return smooth(tau2pole(0.02), x)


def zero(b0: ElemNode, b1: ElemNode, x: ElemNode) -> NodeRepr:
    return sub(mul(b0, x), mul(b1, z(x)))


def dcblock(x: ElemNode) -> NodeRepr:
    return pole(0.995, zero(1, 1, x))


def df11(b0: ElemNode, b1: ElemNode, a1: ElemNode, x: ElemNode) -> NodeRepr:
    return pole(a1, zero(b0, b1, x))


def lowpass(fc: ElemNode, q: ElemNode, x: ElemNode) -> NodeRepr:
    return svf({"mode": "lowpass"}, fc, q, x)


def highpass(fc: ElemNode, q: ElemNode, x: ElemNode) -> NodeRepr:
    return svf({"mode": "highpass"}, fc, q, x)


def bandpass(fc: ElemNode, q: ElemNode, x: ElemNode) -> NodeRepr:
    return svf({"mode": "bandpass"}, fc, q, x)


def notch(fc: ElemNode, q: ElemNode, x: ElemNode) -> NodeRepr:
    return svf({"mode": "notch"}, fc, q, x)


def allpass(fc: ElemNode, q: ElemNode, x: ElemNode) -> NodeRepr:
    return svf({"mode": "allpass"}, fc, q, x)


def peak(fc: ElemNode, q: ElemNode, gain_db: ElemNode, x: ElemNode) -> NodeRepr:
    return svfshelf({"mode": "peak"}, fc, q, gain_db, x)


def lowshelf(fc: ElemNode, q: ElemNode, gain_db: ElemNode, x: ElemNode) -> NodeRepr:
    return svfshelf({"mode": "lowshelf"}, fc, q, gain_db, x)


def highshelf(fc: ElemNode, q: ElemNode, gain_db: ElemNode, x: ElemNode) -> NodeRepr:
    return svfshelf({"mode": "highshelf"}, fc, q, gain_db, x)


def pink(x: ElemNode) -> NodeRepr:
    def clip(lo, hi, v):
        return min(hi, max(lo, v))

    return clip(
        -1,
        1,
        mul(
            db2gain(-30),
            add(
                pole(0.99765, mul(x, 0.099046)),
                pole(0.963, mul(x, 0.2965164)),
                pole(0.57, mul(x, 1.0526913)),
                mul(0.1848, x),
            ),
        ),
    )


def pinknoise(props: Optional[Dict[str, Any]] = None) -> NodeRepr:
    return pink(noise(props))


# ---- lib/envelopes.ts ----------------------------------------------------------------
def adsr(attack_sec: ElemNode, decay_sec: ElemNode, sustain: ElemNode, release_sec: ElemNode, gate: ElemNode) -> NodeRepr:
    a, d, s, r, g = attack_sec, decay_sec, sustain, release_sec, gate
    atk_samps = mul(a, sr())
    atk_gate = le(counter(g), atk_samps)
    target = select(g, select(atk_gate, 1.0, s), 0)
    t60 = max(0.0001, select(g, select(atk_gate, a, d), r))
    p = tau2pole(div(t60, 6.91))
    return smooth(p, target)


# ---- lib/dynamics.ts -----------------------------------------------------------------
def compress(attack_ms: ElemNode, release_ms: ElemNode, threshold: ElemNode, ratio: ElemNode, sidechain: ElemNode, xn: ElemNode) -> NodeRepr:
    e = env(tau2pole(mul(0.001, attack_ms)), tau2pole(mul(0.001, release_ms)), sidechain)
    env_db = gain2db(e)
    adjusted = sub(1, div(1, ratio))
    gain = mul(adjusted, sub(threshold, env_db))
    clean = min(0, gain)
    return mul(xn, db2gain(clean))


# ---- lib/mc.ts: multi-channel nodes; each returns one NodeRepr per output channel (nodeUtils unpack) --------------
class mc:
    @staticmethod
    def _n(kind: str, props: Dict[str, Any], *children: ElemNode) -> List[NodeRepr]:
        from .reconciler import unpack
        p = dict(props)
        channels = p.pop("channels", None)
        if not isinstance(channels, (int, float)) or channels <= 0:
            raise ValueError("Must provide a positive number channels prop")
        return unpack(_n(kind, p, *children), int(channels))

    @staticmethod
    def table(props: Dict[str, Any], t: ElemNode) -> List[NodeRepr]:
        """lib/mc.ts:91-107"""
        return mc._n("mc.table", props, t)

    @staticmethod
    def sample(props: Dict[str, Any], gate: ElemNode) -> List[NodeRepr]:
        """lib/mc.ts:12-34"""
        return mc._n("mc.sample", props, gate)

    @staticmethod
    def sampleseq(props: Dict[str, Any], t: ElemNode) -> List[NodeRepr]:
        """lib/mc.ts:36-56"""
        return mc._n("mc.sampleseq", props, t)

    @staticmethod
    def capture(props: Dict[str, Any], g: ElemNode, *args: ElemNode) -> List[NodeRepr]:
        """lib/mc.ts:94-113: records every input while the gate g is non-zero (any drop of g hands the take over); output channel
        k passes args[k] through; the recording arrives as an "mc.capture" event with one array per input."""
        return mc._n("mc.capture", props, g, *args)
