#!/usr/bin/env node
// Records golden outputs of the `convolve` node from the reference's own prebuilt wasm engine
// (js/packages/offline-renderer/elementary-wasm.cjs: Runtime<double> + wasm/Convolve.h + the real
// HiFi-LoFi FFTConvolver), the only place that arithmetic exists in the reference checkout.
//
// Authoring container only:  node tests/golden/make_convolve_golden.js
// Node 12 cannot parse the ES2020/2021 syntax in the emscripten glue, so a TEMP copy of the JS glue
// is patched textually (wasm payload untouched) and loaded from the OS temp dir.
// Output: tests/golden/convolve_wasm.f32 (little-endian float32, scenarios back to back) and
//         tests/golden/convolve_wasm.json (offsets + sample counts per scenario).
const fs = require('fs'), os = require('os'), path = require('path');
const HERE = __dirname;
const REF = '/root/reference/js/packages/offline-renderer/elementary-wasm.cjs';

function patchedGlue() {
  let src = fs.readFileSync(REF, 'utf8');
  src = src.replace(/globalThis\?\.crypto\?\.getRandomValues/g, '(globalThis.crypto&&globalThis.crypto.getRandomValues)');
  src = src.replace(/([A-Za-z_$][\w$]*(?:\.[A-Za-z_$][\w$]*)+)\?\.\(([^()]*)\)/g, (m, f, a) => `(${f}&&${f}(${a}))`);
  src = src.replace(/([A-Za-z_$][\w$]*)\?\?=([\w$]+)/g, (m, v, d) => `${v}=(${v}==null?${d}:${v})`);
  src = src.replace(/([A-Za-z_$][\w$]*)&&=([A-Za-z_$][\w$]*\([^()]*\))/g, (m, v, e) => `${v}=${v}&&(${e})`);
  src = src.replace(/\(X=C\.U\)\.ka\?\?\(X\.ka=\[\]\)/g, '((X=C.U).ka!=null?X.ka:(X.ka=[]))');
  const out = path.join(fs.mkdtempSync(path.join(os.tmpdir(), 'elemwasm-')), 'elementary-wasm.patched.cjs');
  fs.writeFileSync(out, src);
  return out;
}

function Lcg(seed) { let s = seed >>> 0; return () => { s = (Math.imul(1664525, s) + 1013904223) >>> 0; return s / 2147483648 - 1; }; }

function makeIr(spec) {
  const next = Lcg(spec.seed), ir = new Float64Array(spec.len);
  let d = 1.0;
  for (let n = 0; n < spec.len; ++n) { ir[n] = next() * d; d = d * spec.r; }
  ir[0] = 1.0;
  if (spec.tiny_from !== undefined) for (let n = spec.tiny_from; n < spec.len; ++n) ir[n] = ir[n] * 1e-7;
  let ss = 0.0;
  for (let n = 0; n < spec.len; ++n) ss = ss + ir[n] * ir[n];
  const norm = Math.sqrt(ss), out = new Float32Array(spec.len);
  for (let n = 0; n < spec.len; ++n) out[n] = ir[n] / norm;
  return out;
}

(async () => {
  const M = await require(patchedGlue())();
  const spec = JSON.parse(fs.readFileSync(path.join(HERE, 'convolve_scenarios.json'), 'utf8'));
  const chunks = [], manifest = {};
  let offset = 0;
  for (const sc of spec.scenarios) {
    const p = new M.ElementaryAudioProcessor(1, 1);
    p.prepare(48000, 512);
    for (const ir of sc.irs) {
      const r = p.addSharedResource(ir.name, makeIr(ir));
      if (!r.success) throw new Error(r.message);
    }
    // root(1) <- convolve(2) <- in(3); same wire format the reconciler emits (Runtime.h:115-121)
    let r = p.postMessageBatch([[0, 1, 'root'], [0, 2, 'convolve'], [0, 3, 'in'], [3, 3, 'channel', 0], [3, 1, 'channel', 0],
                                [2, 2, 3, 0], [2, 1, 2, 0], [4, [1]], [5]]);
    if (!r.success) throw new Error(r.message);
    const next = Lcg(sc.input_seed), out = [];
    for (const call of sc.calls) {
      if (call[0] === 'path') {
        r = p.postMessageBatch([[3, 2, 'path', call[1]], [5]]);
        if (!r.success) throw new Error(r.message);
      } else {
        const n = call[1];
        for (let k = 0; k < call[2]; ++k) {
          const inp = p.getInputBufferData(0);
          for (let j = 0; j < n; ++j) inp[j] = Math.fround(next() * 0.25);
          p.process(n);
          const o = p.getOutputBufferData(0);
          for (let j = 0; j < n; ++j) out.push(o[j]);
        }
      }
    }
    const f = Float32Array.from(out);
    // Runtime<double>: the root's 20 ms fade-in multiplies the (float) convolver output by a double gain, so
    // only samples after the fade are float32-exact; the stored value is the float32 rounding either way.
    let lastInexact = -1;
    for (let j = 0; j < out.length; ++j) if (f[j] !== out[j]) lastInexact = j;
    if (lastInexact >= 960) throw new Error('inexact float32 output after the root fade: ' + lastInexact);
    manifest[sc.name] = { offset, count: f.length, max_abs: f.reduce((a, b) => Math.max(a, Math.abs(b)), 0) };
    chunks.push(Buffer.from(f.buffer));
    offset += f.length;
    p.delete();
  }
  fs.writeFileSync(path.join(HERE, 'convolve_wasm.f32'), Buffer.concat(chunks));
  fs.writeFileSync(path.join(HERE, 'convolve_wasm.json'), JSON.stringify(manifest, null, 1));
  console.log(manifest);
})().catch(e => { console.error(e); process.exit(1); });
