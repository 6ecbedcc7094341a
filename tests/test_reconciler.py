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
