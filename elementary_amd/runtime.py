"""``Runtime`` — host-side mirror of ``elem::Runtime<float>`` backed by libelemhip.so (HIP, gfx950).

There is no CPU fallback: constructing a ``Runtime`` without the compiled extension or without
a usable GPU raises.  Method names/arguments follow the reference class
(runtime/elem/Runtime.h:39-153); see ``_cabi.CRuntime`` for the shared surface.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, Optional

from ._cabi import CRuntime, RETURN_CODES

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libelemhip.so")
_lib: Optional[C.CDLL] = None


class ElemHipError(RuntimeError):
    pass


def sum_buses(dst_ptr: int, partial_ptrs, num_floats: int, device: int = 0, hip_stream: int = 0) -> None:
    """``elemhip_sum_buses``: dst = ((partials[0] + partials[1]) + ...) in the order given — the rank-ordered, bit-reproducible
    sum of the ranks' output buses once they sit on one device (raw device pointers, ``num_floats`` float32 each)."""
    lib = load_library()
    arr = (C.c_void_p * len(partial_ptrs))(*[C.c_void_p(p) for p in partial_ptrs])
    rc = lib.elemhip_sum_buses(int(device), C.c_void_p(hip_stream or None), C.c_void_p(dst_ptr), arr, len(partial_ptrs), int(num_floats))
    if rc != 0:
        raise ElemHipError(f"elemhip_sum_buses failed: {describe(rc)} (code {rc})")


def load_library() -> C.CDLL:
    """Load the in-tree HIP engine. Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ElemHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C elementary_amd/csrc`). The HIP engine has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        lib.elemhip_create.argtypes = [C.c_double, C.c_int, C.c_int]
        lib.elemhip_create.restype = C.c_void_p
        lib.elemhip_last_create_error.restype = C.c_int
        lib.elemhip_describe.argtypes = [C.c_int]
        lib.elemhip_describe.restype = C.c_char_p
        lib.elemhip_process_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int64]
        lib.elemhip_process_blocks.restype = C.c_int
        _fpp = C.POINTER(C.POINTER(C.c_float))
        lib.elemhip_process_blocks_host.argtypes = [C.c_void_p, _fpp, C.c_size_t, _fpp, C.c_size_t, C.c_size_t, C.c_int64]
        lib.elemhip_process_blocks_host.restype = C.c_int
        lib.elemhip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.elemhip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        lib.elemhip_get_stats.argtypes = [C.c_void_p, C.c_void_p]
        lib.elemhip_time_launches.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_float), C.c_size_t]
        lib.elemhip_time_launches.restype = C.c_int
        lib.elemhip_describe_plan.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.elemhip_describe_plan.restype = C.c_size_t
        lib.elemhip_sum_buses.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]
        lib.elemhip_sum_buses.restype = C.c_int
        _lib = lib
    return _lib


class _Stats(C.Structure):
    _fields_ = [
        ("blocks_rendered", C.c_uint64), ("plans_built", C.c_uint64), ("last_plan_build_ms", C.c_double),
        ("num_islands", C.c_uint32), ("num_levels", C.c_uint32), ("num_tasks", C.c_uint32),
        ("num_nodes_in_plan", C.c_uint32), ("max_lds_bytes", C.c_uint32), ("num_hbm_buffers", C.c_uint32),
        ("graph_replays", C.c_uint64), ("graph_captures", C.c_uint64), ("batch_launches", C.c_uint64),
        ("spec_launches", C.c_uint64), ("spec_shapes", C.c_uint32), ("spec_islands", C.c_uint32), ("last_jit_wait_ms", C.c_double), ("last_graph_capture_ms", C.c_double),
        ("resident_launches", C.c_uint64), ("resident_blocks", C.c_uint64),
    ]


def describe(code: int) -> str:
    try:
        return load_library().elemhip_describe(int(code)).decode()
    except ElemHipError:
        return RETURN_CODES.get(code, "Return code not recognized")


class Runtime(CRuntime):
    """``elem::Runtime<float>`` on one MI355X (``device`` = HIP ordinal; -1 = dry host-logic handle)."""

    def __init__(self, sample_rate: float, block_size: int, device: int = 0):
        lib = load_library()
        h = lib.elemhip_create(float(sample_rate), int(block_size), int(device))
        if not h:
            code = lib.elemhip_last_create_error()
            raise ElemHipError(f"elemhip_create failed: {describe(code)} (code {code})")
        super().__init__(lib, "elemhip_", C.c_void_p(h), sample_rate, block_size)
        self.device = int(device)

    # -- offline / throughput path --------------------------------------------------------
    def process_blocks(self, num_blocks: int, num_outputs: int, out_ptr: int = 0, in_ptr: int = 0, num_inputs: int = 0,
                       sample_time: Optional[int] = None) -> None:
        """Render ``num_blocks`` full blocks with device-resident I/O (raw device pointers).

        ``out_ptr`` -> float32 [num_blocks][num_outputs][block_size] in HBM (0 = discard),
        ``in_ptr``  -> float32 [num_blocks][num_inputs][block_size] in HBM (0 when no inputs).
        """
        st = self.sample_time if sample_time is None else int(sample_time)
        rc = self._lib.elemhip_process_blocks(self._h, C.c_void_p(in_ptr or None), num_inputs,
                                              C.c_void_p(out_ptr or None), num_outputs, int(num_blocks), st)
        if rc != 0:
            raise ElemHipError(f"elemhip_process_blocks failed: {describe(rc)} (code {rc})")
        if sample_time is None:
            self.sample_time += int(num_blocks) * self.block_size

    def process_blocks_host(self, inputs, num_outputs: int, num_frames: Optional[int] = None, out=None,
                            sample_time: Optional[int] = None):
        """``elemhip_process_blocks_host``: the offline caller's whole block loop over HOST arrays.

        ``inputs``: float32 ``[num_inputs, num_frames]`` (or None); returns / fills ``out``: float32
        ``[num_outputs, num_frames]`` (C-contiguous rows). ``ceil(num_frames / block_size)`` full blocks are rendered.
        """
        import numpy as np
        from ._cabi import _ptr_array
        rows = []
        if inputs is not None:
            a = np.ascontiguousarray(inputs, dtype=np.float32)
            if a.ndim == 1:
                a = a[None, :]
            rows = [a[i] for i in range(a.shape[0])]
            if num_frames is None:
                num_frames = a.shape[1]
        if num_frames is None:
            num_frames = out.shape[1]
        if out is None:
            out = np.empty((num_outputs, int(num_frames)), dtype=np.float32)
        assert out.dtype == np.float32 and out.shape[0] == num_outputs and out.shape[1] >= num_frames and out.strides[1] == 4
        st = self.sample_time if sample_time is None else int(sample_time)
        rc = self._lib.elemhip_process_blocks_host(self._h, _ptr_array(rows), len(rows),
                                                   _ptr_array([out[i] for i in range(num_outputs)]), num_outputs, int(num_frames), st)
        if rc != 0:
            raise ElemHipError(f"elemhip_process_blocks_host failed: {describe(rc)} (code {rc})")
        if sample_time is None:
            nb = (int(num_frames) + self.block_size - 1) // self.block_size
            self.sample_time += nb * self.block_size
        return out

    def event_window_blocks(self) -> int:
        """Blocks a ``process_queued_events(blockwise=True)`` window may span and still equal a relay after every block."""
        f = self._lib.elemhip_event_window_blocks
        f.argtypes = [C.c_void_p]
        f.restype = C.c_uint32
        return int(f(self._h))

    def set_stream(self, hip_stream: int) -> None:
        self._lib.elemhip_set_stream(self._h, C.c_void_p(hip_stream))

    def set_option(self, key: str, value: float) -> None:
        rc = self._lib.elemhip_set_option(self._h, key.encode(), float(value))
        if rc != 0:
            raise ElemHipError(f"unknown option {key!r}")

    def time_launches(self, num_outputs: int, num_blocks: int):
        """Mean HIP-event duration (ms) of each launch level and of the epilogue kernel."""
        buf = (C.c_float * 64)()
        k = self._lib.elemhip_time_launches(self._h, num_outputs, int(num_blocks), buf, 64)
        if k < 0:
            raise ElemHipError(f"elemhip_time_launches failed: {describe(-k)}")
        self.last_event_overhead_ms = float(buf[k])   # empty event pair, already subtracted
        self.last_time_batch = max(1, int(buf[k + 1]))   # blocks per timed launch (option "time_batch")
        self.sample_time += int(num_blocks) * self.block_size * self.last_time_batch
        return [float(buf[i]) for i in range(k)]

    def launch_profile(self):
        """Per-launch-level HIP-event time of the multi-block launches issued since ``set_option('profile_launches', 1)``:
        ``{'level_ms': [...], 'epilogue_ms': x, 'launch_sets': n, 'blocks': b}`` (sums over the profiled launch sets)."""
        buf = (C.c_double * 64)()
        sets, blocks = C.c_uint64(0), C.c_uint64(0)
        f = self._lib.elemhip_get_launch_profile
        f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        f.restype = C.c_int
        k = f(self._h, buf, 64, C.byref(sets), C.byref(blocks))
        if k <= 0:
            return {"level_ms": [], "epilogue_ms": 0.0, "launch_sets": int(sets.value), "blocks": int(blocks.value)}
        if k > 64:      # more launch levels than the first buffer holds: ask again with room for all of them
            buf = (C.c_double * k)()
            k = min(k, f(self._h, buf, k, C.byref(sets), C.byref(blocks)))
        vals = [float(buf[i]) for i in range(k)]
        return {"level_ms": vals[:-1], "epilogue_ms": vals[-1], "launch_sets": int(sets.value), "blocks": int(blocks.value)}

    def spec_info(self, k: int = 0) -> Dict[str, Any]:
        """The k-th specialised island shape of the newest plan: program text, compiler log, state, islands covered."""
        f = self._lib.elemhip_spec_info
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        f.restype = C.c_int
        src, log = C.create_string_buffer(4 << 20), C.create_string_buffer(1 << 20)    # (C2 voice: 170 KB of text)
        state, isl = C.c_int(0), C.c_uint32(0)
        n = f(self._h, k, src, len(src), log, len(log), C.byref(state), C.byref(isl))
        return {"shapes": n, "source": src.value.decode(), "log": log.value.decode(errors="replace"), "state": state.value, "islands": isl.value}

    def describe_plan(self) -> Dict[str, Any]:
        import json
        buf = C.create_string_buffer(1 << 20)
        self._lib.elemhip_describe_plan(self._h, buf, len(buf))
        return json.loads(buf.value)

    def stats(self) -> Dict[str, Any]:
        s = _Stats()
        self._lib.elemhip_get_stats(self._h, C.byref(s))
        return {k: getattr(s, k) for k, _ in _Stats._fields_}
