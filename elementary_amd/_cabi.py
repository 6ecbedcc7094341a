"""ctypes plumbing shared by every engine that speaks the ``elem*_`` C-ABI (include/elemhip.h).

``CRuntime`` is the host-side mirror of ``elem::Runtime<float>`` (runtime/elem/Runtime.h:39-153):
same method names and argument meaning (snake_case), same integer return codes
(runtime/elem/Types.h:51-86).  It is parameterised by (shared library, symbol prefix) so the
product engine (``elemhip_``) and the test-only CPU checkers under ``oracle/`` can be driven by
identical test code; this module itself knows nothing about ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .reconciler import Renderer, batch_to_json

RETURN_CODES = {
    0: "Ok",
    1: "Node type not recognized",
    2: "Node not found",
    3: "Attempting to create a node that already exists",
    4: "Attempting to create a node type that already exists",
    5: "Invalid value type for the given node property",
    6: "Invalid value for the given node property",
    7: "Invariant violation",
    8: "Invalid instruction format",
}

_FPP = C.POINTER(C.POINTER(C.c_float))


_EVENT_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p, C.c_void_p)


def _ptr_array(rows: Sequence[np.ndarray]):
    arr = (C.POINTER(C.c_float) * max(1, len(rows)))()
    for i, r in enumerate(rows):
        arr[i] = r.ctypes.data_as(C.POINTER(C.c_float))
    return arr


class CRuntime:
    """Mirror of ``elem::Runtime<float>`` over a C-ABI shared library."""

    def __init__(self, lib: C.CDLL, prefix: str, handle: C.c_void_p, sample_rate: float, block_size: int):
        self._lib = lib
        self._p = prefix
        self._h = handle
        self.sample_rate = float(sample_rate)
        self.block_size = int(block_size)
        self.sample_time = 0
        self._renderer: Optional[Renderer] = None
        self._bind()

    # -- symbol binding ------------------------------------------------------------
    def _fn(self, name: str):
        return getattr(self._lib, self._p + name)

    def _bind(self) -> None:
        f = self._fn("apply_instructions_json")
        f.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        f.restype = C.c_int
        f = self._fn("process")
        f.argtypes = [C.c_void_p, _FPP, C.c_size_t, _FPP, C.c_size_t, C.c_size_t, C.c_int64]
        f.restype = C.c_int
        f = self._fn("add_shared_resource")
        f.argtypes = [C.c_void_p, C.c_char_p, _FPP, C.c_size_t, C.c_size_t]
        f.restype = C.c_int
        f = self._fn("gc")
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t]
        f.restype = C.c_size_t
        self._fn("destroy").argtypes = [C.c_void_p]
        self._fn("destroy").restype = None
        self._fn("reset").argtypes = [C.c_void_p]
        self._fn("prune_shared_resources").argtypes = [C.c_void_p]
        f = self._fn("process_queued_events")
        f.argtypes = [C.c_void_p, _EVENT_CB, C.c_void_p]
        f.restype = C.c_int

    # -- Runtime API -----------------------------------------------------------------
    def apply_instructions(self, batch: List[list]) -> int:
        """``Runtime::applyInstructions`` (Runtime.h:170-218); returns the ReturnCode."""
        s = batch_to_json(batch).encode("utf-8")
        return int(self._fn("apply_instructions_json")(self._h, s, len(s)))

    def apply_instructions_json(self, s: str) -> int:
        b = s.encode("utf-8")
        return int(self._fn("apply_instructions_json")(self._h, b, len(b)))

    def process(self, inputs: Optional[np.ndarray], num_outputs: int, num_samples: Optional[int] = None,
                sample_time: Optional[int] = None) -> np.ndarray:
        """``Runtime::process`` (Runtime.h:274-290): planar float32 in -> planar float32 out.

        ``sample_time`` plays the role of the wasm host's ``userData`` (wasm/Main.cpp:206-215);
        when omitted an internal counter advances by ``num_samples`` per call.
        """
        n = self.block_size if num_samples is None else int(num_samples)
        if inputs is None:
            rows: List[np.ndarray] = []
        else:
            a = np.ascontiguousarray(inputs, dtype=np.float32)
            if a.ndim == 1:
                a = a[None, :]
            rows = [a[i] for i in range(a.shape[0])]
            self._keep = a
        out = np.full((num_outputs, n), np.nan, dtype=np.float32)
        st = self.sample_time if sample_time is None else int(sample_time)
        rc = self._fn("process")(self._h, _ptr_array(rows), len(rows), _ptr_array([out[i] for i in range(num_outputs)]),
                                 num_outputs, n, st)
        if rc != 0:
            raise RuntimeError(f"{self._p}process failed with code {rc}")
        if sample_time is None:
            self.sample_time += n
        return out

    def add_shared_resource(self, name: str, data: np.ndarray) -> bool:
        a = np.ascontiguousarray(data, dtype=np.float32)
        if a.ndim == 1:
            a = a[None, :]
        rc = self._fn("add_shared_resource")(self._h, name.encode(), _ptr_array([a[i] for i in range(a.shape[0])]),
                                             a.shape[0], a.shape[1])
        return bool(rc)

    def prune_shared_resources(self) -> None:
        self._fn("prune_shared_resources")(self._h)

    def gc(self) -> List[int]:
        cap = 1 << 16
        buf = (C.c_int32 * cap)()
        k = int(self._fn("gc")(self._h, buf, cap))
        if k > cap:   # the reference returns the whole set (Runtime.h:220-272): fetch the rest of this pass
            f = self._fn("last_gc")   # (elemhip_ only; the test checkers never prune that many)
            f.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t]
            f.restype = C.c_size_t
            buf = (C.c_int32 * k)()
            k = int(f(self._h, buf, k))
        ids = [int(buf[i]) for i in range(k)]
        if self._renderer is not None:
            self._renderer.prune(ids)
        return ids

    def reset(self) -> None:
        self._fn("reset")(self._h)

    def process_queued_events(self, blockwise: bool = False) -> List[tuple]:
        """Runtime::processQueuedEvents (Runtime.h:64, 437-446): [(type, payload dict), ...] in relay order.
        ``blockwise`` (HIP engine only): every block's events since the last relay, in block order — what a caller that relayed
        after every block (offline-renderer/index.ts:112-120) would have collected."""
        import json
        got: List[tuple] = []

        def cb(kind, payload, _user):
            got.append((kind.decode(), json.loads(payload.decode())))
        name = "process_queued_events_blockwise" if blockwise else "process_queued_events"
        if blockwise:
            f = self._fn(name)
            f.argtypes = [C.c_void_p, _EVENT_CB, C.c_void_p]
            f.restype = C.c_int
        rc = self._fn(name)(self._h, _EVENT_CB(cb), None)
        if rc != 0:
            raise RuntimeError(f"process_queued_events failed with code {rc}")
        return got

    # -- frontend convenience (offline-renderer/index.ts:60-85) ------------------------
    @property
    def renderer(self) -> Renderer:
        if self._renderer is None:
            self._renderer = Renderer(self.apply_instructions)
        return self._renderer

    def render(self, *roots: Any) -> Dict[str, Any]:
        return self.renderer.render(*roots)

    def close(self) -> None:
        if self._h:
            self._fn("destroy")(self._h)
            self._h = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
