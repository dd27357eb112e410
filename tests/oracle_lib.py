"""ctypes binding of oracle/libac_oracle.so (TEST INFRASTRUCTURE ONLY).

The oracle is the CPU restatement of the reference's algorithm
(oracle/ac_oracle.c).  Nothing under ahocorasick_rs_amd/ imports this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "libac_oracle.so")

STANDARD, LEFTMOST_FIRST, LEFTMOST_LONGEST = 0, 1, 2
KIND_NFA, KIND_DFA = 0, 2


def build_oracle() -> str:
    src = os.path.join(_ROOT, "oracle", "ac_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build_oracle())
        L.aco_build.restype = ctypes.c_void_p
        L.aco_build.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                ctypes.c_int, ctypes.c_int]
        L.aco_free.argtypes = [ctypes.c_void_p]
        for name in ("aco_num_states", "aco_num_classes", "aco_max_pattern_len",
                     "aco_min_pattern_len"):
            getattr(L, name).restype = ctypes.c_uint64
            getattr(L, name).argtypes = [ctypes.c_void_p]
        for name in ("aco_find_iter", "aco_find_overlapping_iter"):
            f = getattr(L, name)
            f.restype = ctypes.c_int64
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                          ctypes.c_void_p, ctypes.c_uint64]
        L.aco_count_iter.restype = ctypes.c_int64
        L.aco_count_iter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
        L.aco_find_str.restype = ctypes.c_int64
        L.aco_find_str.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64]
        L.aco_byte_to_code_point.argtypes = [ctypes.c_void_p, ctypes.c_uint64,
                                             ctypes.c_void_p]
        _lib = L
    return _lib


def pack_patterns(patterns: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(patterns) + 1, dtype=np.uint64)
    if patterns:
        off[1:] = np.cumsum([len(p) for p in patterns], dtype=np.uint64)
    blob = np.frombuffer(b"".join(patterns) + b"\0", dtype=np.uint8).copy()
    return blob, off


def _as_u8(hay) -> np.ndarray:
    if isinstance(hay, np.ndarray):
        assert hay.dtype == np.uint8
        return np.ascontiguousarray(hay)
    return np.frombuffer(bytes(hay) + b"\0", dtype=np.uint8)[: len(hay)]


class Oracle:
    """One compiled oracle automaton."""

    def __init__(self, patterns: Iterable[bytes], match_kind: int = STANDARD,
                 kind: int = KIND_DFA):
        pats = [bytes(p) for p in patterns]
        if any(len(p) == 0 for p in pats):
            raise ValueError("empty pattern")
        self._blob, self._off = pack_patterns(pats)
        self.match_kind = match_kind
        self._h = lib().aco_build(self._blob.ctypes.data, self._off.ctypes.data,
                                  len(pats), match_kind, kind)
        if not self._h:
            raise MemoryError("aco_build failed")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib().aco_free(h)
            self._h = None

    @property
    def num_states(self) -> int:
        return lib().aco_num_states(self._h)

    @property
    def num_classes(self) -> int:
        return lib().aco_num_classes(self._h)

    def _run(self, fn, hay, extra=()) -> np.ndarray:
        h = _as_u8(hay)
        n = len(h)
        cap = 1024
        while True:
            out = np.empty((cap, 3), dtype=np.uint64)
            r = fn(self._h, h.ctypes.data if n else None, n, *extra,
                   out.ctypes.data, cap)
            if r < 0:
                raise ValueError("overlapping search unsupported for this match kind")
            if r <= cap:
                return out[:r]
            cap = int(r)

    def find_raw(self, hay, overlapping: bool = False) -> np.ndarray:
        """(n,3) u64 array of (pattern, start, end) byte offsets."""
        fn = lib().aco_find_overlapping_iter if overlapping else lib().aco_find_iter
        return self._run(fn, hay)

    def find(self, hay, overlapping: bool = False) -> List[Tuple[int, int, int]]:
        return [tuple(int(x) for x in r) for r in self.find_raw(hay, overlapping)]

    def find_str(self, s: str, overlapping: bool = False) -> List[Tuple[int, int, int]]:
        b = s.encode("utf-8")
        arr = self._run(lib().aco_find_str, b, extra=(1 if overlapping else 0,))
        return [tuple(int(x) for x in r) for r in arr]

    def count(self, hay) -> int:
        h = _as_u8(hay)
        return int(lib().aco_count_iter(self._h, h.ctypes.data, len(h)))


def byte_to_code_point(b: bytes) -> np.ndarray:
    h = _as_u8(b)
    out = np.empty(len(h) + 1, dtype=np.uint64)
    lib().aco_byte_to_code_point(h.ctypes.data if len(h) else None, len(h),
                                 out.ctypes.data)
    return out
