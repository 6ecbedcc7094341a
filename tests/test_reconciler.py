"""Wire-format known-answer tests: our frontend mirror must emit the reference reconciler's exact
instruction batches, real int32 hashes included (js/packages/core/__tests__/core.test.js and its
snapshot, transcribed to tests/golden/core_instruction_batches.json by tests/golden/make_golden.py)."""
import json
import math
import os

from elementary_amd.reconciler import Delegate, Renderer, create_node, js_number, render_with_delegate, resolve

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "core_instruction_batches.json")))


class BatchRenderer:
    """core.test.js:9-30"""

    def __init__(self):
        self.d = Delegate()

    def render(self, *roots):
        self.d.clear()
        render_with_delegate(self.d, [resolve(r) for r in roots], 20, 20)

    def batch(self):
        return self.d.packed()


def sort_batch(b):
    """core.test.js:32-49: sort by hash inside each opcode group (JS sort is stable)."""
    out, i = [], 0
    while i < len(b):
        j = i
        while j < len(b) and b[j][0] == b[i][0]:
            j += 1
        grp = b[i:j]
        out += sorted(grp, key=lambda x: x[1]) if len(grp[0]) > 1 and not isinstance(grp[0][1], list) else grp
        i = j
    return out


def sine(key=None, freq=440, pi_literal=False):
    fq = create_node("const", {"key": key, "value": freq} if key else {"value": freq}, [])
    two_pi = 2 * math.pi if pi_literal else create_node("const", {"value": 2 * math.pi}, [])
    return create_node("sin", {}, [create_node("mul", {}, [two_pi, create_node("phasor", {}, [fq])])])


def check(name, batch):
    assert json.loads(json.dumps(sort_batch(batch))) == GOLD[name]


def test_the_basics():
    tr = BatchRenderer(); tr.render(sine("fq")); check("the basics 1", tr.batch())


def test_numeric_literals():
    tr = BatchRenderer()
    tr.render(create_node("sin", {}, [create_node("mul", {}, [2 * math.pi, create_node("phasor", {}, [440])])]))
    check("numeric literals 1", tr.batch())


def test_distinguish_by_props():
    def voice(path, seq):
        return create_node("sample", {"path": path}, [create_node("seq", {"seq": seq}, [
            create_node("le", {}, [create_node("phasor", {}, [create_node("const", {"value": 2}, [])]),
                                   create_node("const", {"value": 0.5}, [])])])])
    tr = BatchRenderer(); tr.render(voice("test/path.wav", [0, 0, 1]), voice("test/path.wav", [0, 1, 0]))
    check("distinguish by props 1", tr.batch())


def test_multi_channel_basics():
    tr = BatchRenderer(); m = sine("fq"); tr.render(m, m); check("multi-channel basics 1", tr.batch())


def test_simple_sharing():
    tr = BatchRenderer(); tr.render(sine("fq")); tr.render(create_node("tanh", {}, [sine("fq")]))
    check("simple sharing 1", tr.batch())


def test_subtrees_by_key_and_value_change():
    tr = BatchRenderer()
    voices = [("fq1", 440), ("fq2", 440), ("fq3", 440), ("fq4", 440)]
    tr.render(create_node("add", {}, [sine(k, f) for k, f in voices]))
    check("distinguished subtrees by key 1", tr.batch())
    voices[0] = ("fq1", 441)
    tr.render(create_node("add", {}, [sine(k, f) for k, f in voices]))
    check("structural equality with value change 1", tr.batch())


def test_switch_and_switch_back():
    tr = BatchRenderer()
    tr.render(sine("hi", 440)); tr.render(sine("bye", 880)); tr.render(sine("hi", 440))
    check("switch and switch back 1", tr.batch())


def test_refs():
    """core.test.js:257-282: a ref update is exactly one SET_PROPERTY and a COMMIT."""
    sent = []
    r = Renderer(lambda b: sent.append(b) or 0)
    freq, set_freq = r.create_ref("const", {"value": 440}, [])
    r.render(create_node("sin", {}, [create_node("mul", {}, [2 * math.pi, create_node("phasor", {}, [freq])])]))
    set_freq({"value": 550})
    assert sent[-1] == [[3, 1915043800, "value", 550], [5]]


def test_js_number_formatting():
    assert [js_number(x) for x in (1e-7, 1e21, 0.000001, 1.5e-10, 100, -0.05, 1e20, 2 * math.pi, 0.1 + 0.2)] == \
        ["1e-7", "1e+21", "0.000001", "1.5e-10", "100", "-0.05", "100000000000000000000", "6.283185307179586", "0.30000000000000004"]


# ---- hashing.test.js: "instruction set similarity without hash values" -----------------------------------
HASHLESS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hashless_instruction_batches.json")))


class HashlessRenderer:
    """hashing.test.js:16-104: every hash is replaced by the ordinal of its first appearance; appendChild drops the
    output channel. What is left is the SHAPE of the batch: which nodes the library functions expand to, how the
    reconciler shares them, and the order in which it visits them."""

    def __init__(self):
        self.mask = {}
        self.batch = []

    def _id(self, h):
        return self.mask.setdefault(h, len(self.mask))

    def render(self, *roots):
        d = Delegate()
        outer = self

        class Masking(Delegate):
            def create_node(self, h, kind): outer.batch.append([0, outer._id(h), kind])
            def append_child(self, parent, child, ch): outer.batch.append([2, outer._id(parent), outer._id(child)])
            def set_property(self, h, key, value): outer.batch.append([3, outer._id(h), key, value])
            def activate_roots(self, roots): outer.batch.append([4, [outer._id(r) for r in roots]])
            def commit_updates(self): outer.batch.append([5])
        m = Masking()
        m.node_map = d.node_map
        render_with_delegate(m, [resolve(r) for r in roots], 20, 20)


def sort_hashless(b):
    """hashing.test.js:106-116: by opcode, then by masked id (not stable across equal ids in JS either: compare as multisets per (op, id))."""
    return sorted(b, key=lambda x: (x[0], x[1] if len(x) > 1 and not isinstance(x[1], list) else -1))


def _canon(b):
    out = []
    for x in sort_hashless(b):
        out.append(json.loads(json.dumps(x)))
    # entries with the same (opcode, id) may come in either order from a non-stable comparator
    return sorted(out, key=lambda x: (x[0], x[1] if len(x) > 1 and not isinstance(x[1], list) else -1, json.dumps(x[2:])))


def test_hashless_cycle():
    from elementary_amd import el
    tr = HashlessRenderer()
    tr.render(el.cycle(440))
    assert _canon(tr.batch) == _canon(HASHLESS["instruction set similarity without hash values 1"])


def stranger_things_voice():
    """cli/examples/02_StrangerThings.js:10-31 == hashing.test.js:128-150 (the 69-node synth voice)."""
    from elementary_amd import el

    def synth_voice(hz):
        return el.mul(0.25, el.add(el.blepsaw(el.mul(hz, 1.001)), el.blepsquare(el.mul(hz, 0.994)),
                                   el.blepsquare(el.mul(hz, 0.501)), el.blepsaw(el.mul(hz, 0.496))))
    train = el.train(4.8)
    arp = [261.63 * 0.5 * math.pow(2, x / 12) for x in [0, 4, 7, 11, 12, 11, 7, 4]]

    def modulate(x, rate, amt):
        return el.add(x, el.mul(amt, el.cycle(rate)))
    env = el.adsr(0.01, 0.5, 0, 0.4, train)

    def filt(x):
        return el.lowpass(el.add(40, el.mul(modulate(1840, 0.05, 1800), env)), 1, x)
    return el.mul(0.25, filt(synth_voice(el.seq({"seq": arp, "hold": True}, train, 0))))


def test_hashless_synth_voice():
    tr = HashlessRenderer()
    out = stranger_things_voice()
    tr.render(out, out)
    got, want = _canon(tr.batch), _canon(HASHLESS["instruction set similarity without hash values 2 1"])
    assert len(got) == len(want) == 189

    def same(a, b):   # the arp frequencies come from the SCRIPT's Math.pow (V8) vs math.pow (glibc): equal to an ulp
        if isinstance(a, list) and isinstance(b, list):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, float) or isinstance(b, float):
            return isinstance(a, (int, float)) and isinstance(b, (int, float)) and abs(a - b) <= 4e-16 * max(abs(a), abs(b))
        return a == b
    for g, w in zip(got, want):
        assert same(g, w), (g, w)
