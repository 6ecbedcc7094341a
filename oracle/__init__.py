"""oracle/ — CPU checkers for the block-render path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``elementary_amd/`` does, and the product never falls back to it.

Two checkers, same C-ABI shape as include/elemhip.h:

* ``RefRuntime``  — oracle/_ref/libelemref.so: the UNMODIFIED reference engine
  (``elem::Runtime<float|double>`` from /root/reference/runtime) compiled in place by
  oracle/Makefile (`make ref`).  Built in the authoring container, travels to the GPU box as a
  git-ignored binary.  ``cpu_baseline.kind == "reference"`` times ``libelemref_bench.so``.
* ``PortRuntime`` — oracle/libelemoracle.so: our C++ restatement (oracle/elem_oracle.cpp), each
  function citing the reference file:line it follows; pinned against the reference's jest
  golden vectors (tests/golden) and against ``RefRuntime`` on randomized graphs.
"""
from __future__ import annotations

import ctypes as C
import os

from elementary_amd._cabi import CRuntime

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libelemref.so")
REF_BENCH_SO = os.path.join(_HERE, "_ref", "libelemref_bench.so")
PORT_SO = os.path.join(_HERE, "libelemoracle.so")


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def have_port() -> bool:
    return os.path.exists(PORT_SO)


class RefRuntime(CRuntime):
    """The reference engine itself (float by default; ``use_double`` = the wasm build's type)."""

    def __init__(self, sample_rate: float, block_size: int, use_double: bool = False, bench_build: bool = False):
        lib = C.CDLL(REF_BENCH_SO if bench_build else REF_SO)
        lib.elemref_create.argtypes = [C.c_double, C.c_int, C.c_int]
        lib.elemref_create.restype = C.c_void_p
        h = C.c_void_p(lib.elemref_create(float(sample_rate), int(block_size), int(use_double)))
        super().__init__(lib, "elemref_", h, sample_rate, block_size)


class PortRuntime(CRuntime):
    """Our CPU restatement of the path (oracle/elem_oracle.cpp)."""

    def __init__(self, sample_rate: float, block_size: int):
        lib = C.CDLL(PORT_SO)
        lib.elemoracle_create.argtypes = [C.c_double, C.c_int]
        lib.elemoracle_create.restype = C.c_void_p
        h = C.c_void_p(lib.elemoracle_create(float(sample_rate), int(block_size)))
        super().__init__(lib, "elemoracle_", h, sample_rate, block_size)
