"""Shared parity harness: drive two engines with the same instruction batches and inputs."""
from __future__ import annotations

import numpy as np


def lcg_noise(n: int, seed: int, amp: float = 1.0) -> np.ndarray:
    """SURVEY.md §8(d) input generator: s = 1664525*s + 1013904223 mod 2^32, x = s/2^31 - 1."""
    s = np.uint64(seed)
    out = np.empty(n, dtype=np.float64)
    a, c, m = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    for i in range(n):
        s = (a * s + c) & m
        out[i] = float(s) / 2147483648.0 - 1.0
    return (out * amp).astype(np.float32)


def lcg_noise_fast(n: int, seed: int, amp: float = 1.0) -> np.ndarray:
    """lcg_noise, vectorised (the affine map composed by doubling): the same samples bit for bit, for soaks of 10^6+ frames."""
    m = np.uint64(0xFFFFFFFF)
    a = np.empty(n, dtype=np.uint64); c = np.empty(n, dtype=np.uint64)       # x_{i+1} = a[i] * seed + c[i]
    if n == 0:
        return np.empty(0, dtype=np.float32)
    a[0], c[0] = 1664525, 1013904223
    have = 1
    while have < n:
        k = min(have, n - have)
        ah, ch = a[have - 1], c[have - 1]                                    # the map "advance by `have` steps"
        a[have:have + k] = (a[:k] * ah) & m
        c[have:have + k] = (a[:k] * ch + c[:k]) & m
        have += k
    s = (a * np.uint64(seed) + c) & m
    return ((s.astype(np.float64) / 2147483648.0 - 1.0) * amp).astype(np.float32)


def render_pair(make_a, make_b, roots_fn, sample_rate=44100.0, block=512, blocks=12, n_in=0, n_out=None,
                inputs=None, resources=None, per_block=None):
    """Render `blocks` blocks of the same graph on two engines; returns (outA, outB) [blocks, n_out, block]."""
    a, b = make_a(sample_rate, block), make_b(sample_rate, block)
    for rt in (a, b):
        for name, data in (resources or {}).items():
            assert rt.add_shared_resource(name, data)
    roots = roots_fn()
    n_out = n_out if n_out is not None else len(roots)
    ra = a.render(*roots)
    rb = b.render(*roots)
    assert ra["result"] == 0 and rb["result"] == 0, (ra["result"], rb["result"])
    assert ra["batch"] == rb["batch"]
    outs_a, outs_b = [], []
    for k in range(blocks):
        x = None
        if n_in:
            x = inputs[k] if inputs is not None else np.stack([lcg_noise(block, 1 + c + 97 * k, 0.5) for c in range(n_in)])
        if per_block:
            per_block(k, a, b)
        outs_a.append(a.process(x, n_out, block))
        outs_b.append(b.process(x, n_out, block))
    return np.stack(outs_a), np.stack(outs_b)
