#!/usr/bin/env node
// CPU baseline for BASELINE configs[2] (C3: 8 channels x 96 000-tap convolution reverb, 48 kHz, blockSize 512) measured on
// the reference's OWN engine: the prebuilt wasm build (js/packages/offline-renderer/elementary-wasm.cjs = Runtime<double>
// + wasm/Convolve.h + the real HiFi-LoFi TwoStageFFTConvolver), the only place that arithmetic exists in the reference
// checkout. Authoring container only (the GPU box has neither /root/reference nor the glue):
//     node benchmarks/c3_wasm_baseline.js [blocks] > profiles/r02/c3_wasm_reference_cpu.json
// Node 12 cannot parse the ES2020/2021 syntax of the emscripten glue: a TEMP copy is patched textually (wasm payload
// untouched), exactly as tests/golden/make_convolve_golden.js does.
const fs = require('fs'), os = require('os'), path = require('path');
const REF = '/root/reference/js/packages/offline-renderer/elementary-wasm.cjs';
const CH = 8, TAPS = 96000, BLOCK = 512, SR = 48000;

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
function makeIr(ch) {   // graphs.c3_impulse_response: lcg noise x 0.99993^n, unit energy
  const next = Lcg(1 + ch), ir = new Float64Array(TAPS);
  let ss = 0.0;
  for (let n = 0; n < TAPS; ++n) { ir[n] = next() * Math.pow(0.99993, n); if (n === 0) ir[0] = 1.0; ss += ir[n] * ir[n]; }
  const out = new Float32Array(TAPS), norm = Math.sqrt(ss);
  for (let n = 0; n < TAPS; ++n) out[n] = ir[n] / norm;
  return out;
}

(async () => {
  const blocks = parseInt(process.argv[2] || '3000', 10), warm = 200;
  const M = await require(patchedGlue())();
  const p = new M.ElementaryAudioProcessor(CH, CH);
  p.prepare(SR, BLOCK);
  const batch = [], roots = [];
  for (let ch = 0; ch < CH; ++ch) {
    const r = p.addSharedResource('ir' + ch, makeIr(ch));
    if (!r.success) throw new Error(r.message);
    const root = 100 + ch, conv = 200 + ch, inp = 300 + ch;   // root <- convolve{path} <- in{channel}
    batch.push([0, root, 'root'], [0, conv, 'convolve'], [0, inp, 'in'], [3, inp, 'channel', ch], [3, conv, 'path', 'ir' + ch],
               [3, root, 'channel', ch], [2, conv, inp, 0], [2, root, conv, 0]);
    roots.push(root);
  }
  batch.push([4, roots], [5]);
  const r = p.postMessageBatch(batch);
  if (!r.success) throw new Error(r.message);
  const next = Lcg(12345);
  const times = [];
  let peak = 0;
  for (let b = 0; b < warm + blocks; ++b) {
    for (let ch = 0; ch < CH; ++ch) { const inp = p.getInputBufferData(ch); for (let j = 0; j < BLOCK; ++j) inp[j] = next() * 0.25; }
    const t0 = process.hrtime.bigint();
    p.process(BLOCK);
    const t1 = process.hrtime.bigint();
    if (b >= warm) times.push(Number(t1 - t0) / 1000.0);
    const o = p.getOutputBufferData(0);
    for (let j = 0; j < BLOCK; j += 64) peak = Math.max(peak, Math.abs(o[j]));
  }
  times.sort((a, b) => a - b);
  const mean = times.reduce((a, b) => a + b, 0) / times.length;
  console.log(JSON.stringify({
    what: 'BASELINE configs[2] (C3) on the reference wasm engine (Runtime<double>, TwoStageFFTConvolver 512/4096), 1 thread',
    channels: CH, ir_taps: TAPS, block: BLOCK, sample_rate: SR, blocks_timed: blocks, warmup_blocks: warm,
    us_per_block_mean: mean, us_per_block_p50: times[times.length >> 1], us_per_block_p99: times[Math.floor(times.length * 0.99)],
    us_per_block_max: times[times.length - 1],
    channel_samples_per_s: CH * BLOCK / (mean * 1e-6),
    note: 'the tail convolver runs every 8th block (4096 / 512): p99/max show the burst the two-stage schedule leaves on one thread',
    output_peak: peak, node: process.version, host_cpu: os.cpus()[0].model, host_cores: os.cpus().length,
    measured_on: 'authoring container (not the GPU box: /root/reference is absent there)',
  }, null, 1));
  p.delete();
})().catch(e => { console.error(e); process.exit(1); });
