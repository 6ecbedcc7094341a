#!/usr/bin/env python
"""Transcribe the reference's own golden vectors for the block-render path into small JSON fixtures.

Run in the authoring container only (needs /root/reference):  python tests/golden/make_golden.py
Sources (jest snapshot files of the reference, recorded through its wasm engine):
  js/packages/offline-renderer/__tests__/__snapshots__/{delays,tap,time,offline-renderer,sampleseq,maxhold,sparseq2,sparseq,vfs,mc}.test.js.snap
  js/packages/core/__tests__/__snapshots__/core.test.js.snap   (instruction batches with real int32 hashes)
  js/packages/core/__tests__/__snapshots__/hashing.test.js.snap (the same with masked hashes: 69-node synth voice)
Only Float32Array snapshots and instruction-batch snapshots are transcribed; the scenarios that
produce them are restated in tests/test_oracle_golden.py / tests/test_reconciler.py.
"""
import json
import os
import re

REF = "/root/reference/js/packages"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_snap(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"exports\[`(.+?)`\] = `\n(.*?)\n`;", txt, re.S):
        out[m.group(1)] = m.group(2)
    return out


def float_array(body):
    assert body.startswith("Float32Array ["), body[:40]
    return [float(x) for x in re.findall(r"^\s+(-?[0-9.eE+-]+),$", body, re.M)]


def js_value(body):
    """jest pretty-format of nested arrays/strings/numbers -> JSON."""
    s = re.sub(r"\bArray \[", "[", body)
    s = re.sub(r"\bObject \{", "{", s)
    s = re.sub(r",(\s*[\]}])", r"\1", s)
    return json.loads(s)


def main():
    audio = {}
    for name in ("delays", "tap", "time", "offline-renderer", "sampleseq", "maxhold", "sparseq2", "sparseq", "vfs", "mc"):
        snaps = parse_snap(f"{REF}/offline-renderer/__tests__/__snapshots__/{name}.test.js.snap")
        for key, body in snaps.items():
            if body.startswith("Float32Array ["):
                audio[f"{name}:{key}"] = float_array(body)
    json.dump(audio, open(os.path.join(HERE, "offline_renderer_snapshots.json"), "w"))
    batches = {}
    for key, body in parse_snap(f"{REF}/core/__tests__/__snapshots__/core.test.js.snap").items():
        try:
            batches[key] = js_value(body)
        except Exception:
            pass
    json.dump(batches, open(os.path.join(HERE, "core_instruction_batches.json"), "w"))
    # hashing.test.js: the same batches with hashes replaced by first-seen ordinals (a 69-node synth voice, rendered twice)
    hashless = {}
    for key, body in parse_snap(f"{REF}/core/__tests__/__snapshots__/hashing.test.js.snap").items():
        hashless[key] = js_value(body)
    json.dump(hashless, open(os.path.join(HERE, "hashless_instruction_batches.json"), "w"))
    print("hashless snapshots:", {k: len(v) for k, v in hashless.items()})
    print("audio snapshots:", {k: len(v) for k, v in audio.items()})
    print("batch snapshots:", list(batches))


if __name__ == "__main__":
    main()
