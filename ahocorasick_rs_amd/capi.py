"""ctypes binding of the C ABI (include/acx.h -> libacx_hip.so).

This is the thin Python view of the drop-in boundary used by the device-pointer
workflows (bench.py, the multi-GPU harness, tests).  The reference-shaped
classes (`AhoCorasick`, `BytesAhoCorasick`) live in the C++ CPython extension
`ahocorasick_rs_amd.ahocorasick_rs`; both sit on the same shared library.

There is no CPU fallback: if libacx_hip.so is missing this module raises
ImportError, and without a HIP device every matching call raises RuntimeError.
"""
from __future__ import annotations

import ctypes
import os
import sys
import weakref
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libacx_hip.so")

OK, EINVAL, EEMPTY, EOVERLAP, ENOMEM, EDEVICE, ETOOBIG = 0, -1, -2, -3, -4, -5, -6
MATCH_STANDARD, MATCH_LEFTMOST_FIRST, MATCH_LEFTMOST_LONGEST = 0, 1, 2
IMPL_AUTO, IMPL_NONCONTIGUOUS_NFA, IMPL_CONTIGUOUS_NFA, IMPL_DFA = -1, 0, 1, 2
KERNEL_AUTO, KERNEL_DFA_WALK, KERNEL_PREFILTER = 0, 1, 2
KERNEL_NAMES = {1: "dfa_walk", 2: "prefilter"}
ABI_VERSION = 9  # ACX_VERSION of include/acx.h this binding was written against

MATCH_DTYPE = np.dtype([("pattern", "<u8"), ("start", "<u8"), ("end", "<u8")])


class Info(ctypes.Structure):
    _fields_ = [("n_patterns", ctypes.c_uint64), ("n_states", ctypes.c_uint64),
                ("n_classes", ctypes.c_uint32), ("stride", ctypes.c_uint32),
                ("min_pattern_len", ctypes.c_uint32), ("max_pattern_len", ctypes.c_uint32),
                ("table_bytes", ctypes.c_uint64), ("lds_hot_rows", ctypes.c_uint32),
                ("kernel", ctypes.c_int32), ("match_kind", ctypes.c_int32),
                ("device", ctypes.c_int32), ("filter_q", ctypes.c_uint32)]


class HostTables(ctypes.Structure):
    _fields_ = [("n_patterns", ctypes.c_uint64), ("n_states", ctypes.c_uint64),
                ("n_classes", ctypes.c_uint32), ("stride", ctypes.c_uint32),
                ("min_pattern_len", ctypes.c_uint32), ("max_pattern_len", ctypes.c_uint32),
                ("classes", ctypes.c_void_p), ("table", ctypes.c_void_p),
                ("own_off", ctypes.c_void_p), ("own_pid", ctypes.c_void_p),
                ("dlink", ctypes.c_void_p), ("level_start", ctypes.c_void_p),
                ("pattern_len", ctypes.c_void_p), ("rank", ctypes.c_void_p),
                ("filter_xy", ctypes.c_void_p), ("prefix_table", ctypes.c_void_p),
                ("prefix_lists", ctypes.c_void_p),
                ("filter_q", ctypes.c_uint32), ("filter_q2", ctypes.c_uint32),
                ("filter_entries_log2", ctypes.c_uint32), ("prefix_table_log2", ctypes.c_uint32),
                ("filter_density", ctypes.c_double),
                ("n_prefix_keys", ctypes.c_uint32), ("n_prefix_lists", ctypes.c_uint32),
                ("prefix_bitmap", ctypes.c_void_p),
                ("dense", ctypes.c_uint32), ("first_child", ctypes.c_void_p), ("in_byte", ctypes.c_void_p),
                ("fail", ctypes.c_void_p), ("state_flags", ctypes.c_void_p),
                ("walk_t3b", ctypes.c_void_p), ("walk_t3r", ctypes.c_void_p), ("walk_grec", ctypes.c_void_p),
                ("long_min_len", ctypes.c_uint32), ("n_short", ctypes.c_uint32), ("short_min_len", ctypes.c_uint32),
                ("short_xy", ctypes.c_void_p), ("short_codes", ctypes.c_void_p),
                ("max_shift", ctypes.c_uint32), ("pattern_shift", ctypes.c_void_p), ("pattern_head", ctypes.c_void_p)]


class Profile(ctypes.Structure):
    _fields_ = [("scan_ms", ctypes.c_double), ("scan_launches", ctypes.c_uint64),
                ("post_ms", ctypes.c_double), ("scan_bytes", ctypes.c_uint64),
                ("raw_occurrences", ctypes.c_uint64), ("prefix_hits", ctypes.c_uint64),
                ("small_calls", ctypes.c_uint64)]


def _preload_hip_runtime() -> None:
    # One process holds ONE HIP runtime: PyTorch wheels bundle their own libamdhip64 with the
    # same SONAME as /opt/rocm's, and whichever is loaded first serves both libraries.  Nothing
    # is imported here unless asked for (ACX_PRELOAD_TORCH=1): a plain AhoCorasick user needs
    # no torch, and a process that already imported torch already has its runtime loaded.
    if os.environ.get("ACX_PRELOAD_TORCH") != "1" or "torch" in sys.modules:
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _preload_hip_runtime()
    L = ctypes.CDLL(os.environ.get("ACX_LIB", _SO), mode=ctypes.RTLD_GLOBAL)  # ACX_LIB: experiments with variant builds
    vp, u64, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
    L.acx_version.restype = i32
    # (ACX_LIB_ANY_ABI=1 with ACX_LIB: same-box measurements against an older build of the library)
    if L.acx_version() != ABI_VERSION and not (os.environ.get("ACX_LIB") and os.environ.get("ACX_LIB_ANY_ABI")):
        raise ImportError(f"libacx_hip.so speaks C ABI version {L.acx_version()}, this binding version {ABI_VERSION}: "
                          "rebuild (python -c 'import __graft_entry__ as g; g.build()')")
    L.acx_last_error.restype = ctypes.c_char_p
    L.acx_device_count.argtypes = [ctypes.POINTER(i32)]
    L.acx_set_device.argtypes = [i32]
    L.acx_build.argtypes = [vp, vp, u64, i32, i32, ctypes.POINTER(vp)]
    L.acx_free_automaton.argtypes = [vp]
    L.acx_free_automaton.restype = None
    L.acx_automaton_info.argtypes = [vp, ctypes.POINTER(Info)]
    L.acx_set_kernel.argtypes = [vp, i32]
    L.acx_compile_host.argtypes = [vp, vp, u64, i32, ctypes.POINTER(vp)]
    L.acx_host_tables.argtypes = [vp, ctypes.POINTER(HostTables)]
    L.acx_filter_hash.argtypes = [ctypes.c_uint32]
    L.acx_filter_hash.restype = ctypes.c_uint32
    L.acx_prefix_slot.argtypes = [u64, ctypes.c_uint32, ctypes.c_uint32]
    L.acx_prefix_slot.restype = ctypes.c_uint32
    L.acx_free_host.argtypes = [vp]
    L.acx_free_host.restype = None
    L.acx_find.argtypes = [vp, vp, u64, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(u64)]
    L.acx_free_matches.argtypes = [vp]
    L.acx_free_matches.restype = None
    L.acx_find_batch.argtypes = [vp, vp, vp, u64, i32, i32, ctypes.POINTER(vp),
                                 ctypes.POINTER(u64), vp]
    L.acx_find_device.argtypes = [vp, vp, u64, vp, u64, u64, i32, i32, ctypes.POINTER(vp)]
    L.acx_replicate.argtypes = [vp, i32, ctypes.POINTER(vp)]
    L.acx_automaton_device.argtypes = [vp]
    L.acx_shard_range.argtypes = [u64, i32, i32, ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.acx_shard_range.restype = None
    L.acx_find_batch_multi.argtypes = [vp, i32, vp, vp, u64, i32, i32, ctypes.POINTER(vp), ctypes.POINTER(u64), vp]
    L.acx_result_count.argtypes = [vp]
    L.acx_result_count.restype = u64
    L.acx_result_device_matches.argtypes = [vp]
    L.acx_result_device_matches.restype = vp
    L.acx_result_device_counts.argtypes = [vp]
    L.acx_result_device_counts.restype = vp
    L.acx_result_copy.argtypes = [vp, vp]
    L.acx_result_copy_counts.argtypes = [vp, vp]
    L.acx_free_result.argtypes = [vp]
    L.acx_free_result.restype = None
    L.acx_profile_enable.argtypes = [vp, i32]
    L.acx_profile_read.argtypes = [vp, ctypes.POINTER(Profile), i32]
    L.acx_path_stats.argtypes = [vp, ctypes.POINTER(u64), i32]
    L.acx_device_alloc.argtypes = [ctypes.POINTER(vp), u64]
    L.acx_device_free.argtypes = [vp]
    L.acx_device_upload.argtypes = [vp, vp, u64]
    L.acx_device_download.argtypes = [vp, vp, u64]
    L.acx_device_synchronize.argtypes = []
    L.acx_device_synchronize_on.argtypes = [i32]
    L.acx_comm_init_all.argtypes = [ctypes.POINTER(i32), i32, ctypes.POINTER(vp)]
    L.acx_comm_unique_id.argtypes = [vp]
    L.acx_comm_init_rank.argtypes = [vp, i32, i32, i32, ctypes.POINTER(vp)]
    L.acx_comm_world.argtypes = [vp]
    L.acx_comm_local_ranks.argtypes = [vp]
    L.acx_comm_allgather_counts.argtypes = [vp, vp, vp]
    L.acx_comm_free.argtypes = [vp]
    L.acx_comm_free.restype = None
    L.acx_output_offsets.argtypes = [vp, i32, vp]
    L.acx_output_offsets.restype = None
    L.acx_generate_haystack.argtypes = [vp, vp, u64, i32, u64, u64]
    _lib = L
    return L


class AcxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"acx error {code}: {msg}")
        self.code = code
        self.msg = msg


def _check(rc: int) -> None:
    if rc != OK:
        msg = lib().acx_last_error().decode("utf-8", "replace")
        if rc in (EINVAL, EEMPTY, EOVERLAP, ETOOBIG):
            e = ValueError(msg)
            e.code = rc  # type: ignore[attr-defined]
            raise e
        if rc == ENOMEM:
            raise MemoryError(msg)
        raise AcxError(rc, msg)


def shard_range(n_items: int, shard: int, n_shards: int) -> Tuple[int, int]:
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    lib().acx_shard_range(n_items, shard, n_shards, ctypes.byref(lo), ctypes.byref(hi))
    return int(lo.value), int(hi.value)


def device_count() -> int:
    n = ctypes.c_int(0)
    rc = lib().acx_device_count(ctypes.byref(n))
    return n.value if rc == OK else 0


def set_device(ordinal: int) -> None:
    _check(lib().acx_set_device(ordinal))


def pack(patterns: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(patterns) + 1, dtype=np.uint64)
    if len(patterns):
        off[1:] = np.cumsum([len(p) for p in patterns], dtype=np.uint64)
    blob = np.frombuffer(b"".join(bytes(p) for p in patterns) + b"\0" * 16, dtype=np.uint8).copy()
    return blob, off


class HostAutomaton:
    """acx_host_automaton_t: the compiled tables on the host (no device needed).
    Arrays are numpy views valid while this object is alive."""

    def __init__(self, patterns: Sequence[bytes], match_kind: int = MATCH_STANDARD):
        blob, off = pack(patterns)
        h = ctypes.c_void_p()
        _check(lib().acx_compile_host(blob.ctypes.data, off.ctypes.data, len(patterns),
                                      match_kind, ctypes.byref(h)))
        self._h = h.value
        t = HostTables()
        _check(lib().acx_host_tables(self._h, ctypes.byref(t)))
        self.t = t

        def view(ptr, n, dtype):
            if not n:
                return np.zeros(0, dtype=dtype)
            size = n * np.dtype(dtype).itemsize
            buf = (ctypes.c_uint8 * size).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=n)

        self.n_states = int(t.n_states)
        self.stride = int(t.stride)
        self.classes = view(t.classes, 256, np.uint8)
        self.dense = bool(t.dense)
        self.table = view(t.table, self.n_states * self.stride if t.dense else 0, np.uint32).reshape(-1, self.stride)
        self.first_child = view(t.first_child, self.n_states + 1, np.uint32)
        self.in_byte = view(t.in_byte, self.n_states, np.uint8)
        self.fail = view(t.fail, self.n_states, np.uint32)
        self.state_flags = view(t.state_flags, self.n_states, np.uint8)
        self.own_off = view(t.own_off, self.n_states + 1, np.uint32)
        self.own_pid = view(t.own_pid, int(t.n_patterns), np.uint32)
        self.dlink = view(t.dlink, self.n_states, np.uint32)
        self.level_start = view(t.level_start, int(t.max_pattern_len) + 2, np.uint32)
        self.pattern_len = view(t.pattern_len, int(t.n_patterns), np.uint32)
        self.rank = view(t.rank, int(t.n_patterns), np.uint32)
        self.filter_xy = view(t.filter_xy, (2 << int(t.filter_entries_log2)) if t.filter_q else 0,
                              np.uint32).reshape(-1, 2)
        self.prefix_table = view(t.prefix_table, (4 << int(t.prefix_table_log2)) if t.filter_q else 0,
                                 np.uint32).reshape(-1, 4)
        self.prefix_lists = view(t.prefix_lists, int(t.n_prefix_lists), np.uint32)
        self.prefix_bitmap = view(t.prefix_bitmap, (8 << int(t.prefix_table_log2)) // 32 if t.filter_q else 0, np.uint32)
        # K1a's failureless walk (automata of at most 32 byte classes, else empty)
        nc = int(t.n_classes)
        self.n_classes = nc
        self.walk_t3b = view(t.walk_t3b, 33 * 1024 if t.walk_t3b else 0, np.uint32)
        self.walk_t3r = view(t.walk_t3r, 2 * nc ** 3 if t.walk_t3r else 0, np.uint32).reshape(-1, 2)
        self.walk_grec = view(t.walk_grec, 4 * self.n_states if t.walk_grec else 0, np.uint32).reshape(-1, 4)
        # K1b's side test for patterns of 1 and 2 bytes (empty without such patterns)
        self.short_xy = view(t.short_xy, 512 if t.short_xy else 0, np.uint32).reshape(-1, 2)
        self.short_codes = view(t.short_codes, 256 + 65536 if t.short_codes else 0, np.uint32)
        # anchors: where every pattern is filed (a code of the prefix table is pattern id | shift << 24)
        self.pattern_shift = view(t.pattern_shift, int(t.n_patterns), np.uint8)
        self.pattern_head = view(t.pattern_head, 4 * int(t.n_patterns) if t.pattern_head else 0, np.uint32).reshape(-1, 4)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().acx_free_host(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def filter_hash(gram: bytes) -> int:
    """level-1 hash H of a (Q-1)-byte gram: entry = H >> 18, signature bits from H >> 9."""
    return int(lib().acx_filter_hash(int.from_bytes(gram[:4], "little")))


def prefix_hash(gram: bytes, salt: int) -> int:
    """32-bit hash of the first `salt` bytes of `gram` (the slot is its top bits)."""
    return int(lib().acx_prefix_slot(int.from_bytes(gram[:salt], "little"), salt, 32))


def prefix_slot(gram: bytes, salt: int, log2: int) -> int:
    """home slot of the first `salt` bytes of `gram` in the prefix table."""
    return int(lib().acx_prefix_slot(int.from_bytes(gram[:salt], "little"), salt, log2))


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """acx_comm_unique_id: the 128 bytes rank 0 creates and every rank passes to Comm.init_rank."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _check(lib().acx_comm_unique_id(buf))
    return bytes(buf)


def output_offsets(counts: Sequence[int]) -> List[int]:
    """acx_output_offsets: exclusive prefix of the ranks' match counts (+ the total)."""
    c = np.asarray(counts, dtype=np.uint64)
    out = np.zeros(len(c) + 1, dtype=np.uint64)
    lib().acx_output_offsets(c.ctypes.data, len(c), out.ctypes.data)
    return [int(x) for x in out]


class Comm:
    """acx_comm_t: the count exchange over RCCL, without torch (include/acx.h)."""

    def __init__(self, handle: int):
        self._h = handle

    @classmethod
    def init_all(cls, devices: Sequence[int]) -> "Comm":
        arr = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        _check(lib().acx_comm_init_all(arr, len(devices), ctypes.byref(h)))
        return cls(h.value)

    @classmethod
    def init_rank(cls, uid: bytes, world: int, rank: int, device: int) -> "Comm":
        assert len(uid) == COMM_ID_BYTES
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
        h = ctypes.c_void_p()
        _check(lib().acx_comm_init_rank(buf, world, rank, device, ctypes.byref(h)))
        return cls(h.value)

    @property
    def world(self) -> int:
        return int(lib().acx_comm_world(self._h))

    @property
    def local_ranks(self) -> int:
        return int(lib().acx_comm_local_ranks(self._h))

    def allgather_counts(self, local_counts: Sequence[int]) -> List[int]:
        loc = np.asarray(local_counts, dtype=np.uint64)
        assert len(loc) == self.local_ranks
        out = np.zeros(self.world, dtype=np.uint64)
        _check(lib().acx_comm_allgather_counts(self._h, loc.ctypes.data, out.ctypes.data))
        return [int(x) for x in out]

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().acx_comm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """hipMalloc'ed bytes owned by this object (for hosts without torch)."""

    def __init__(self, nbytes: int):
        p = ctypes.c_void_p()
        _check(lib().acx_device_alloc(ctypes.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes

    def upload(self, arr: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        _check(lib().acx_device_upload(self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, nbytes: Optional[int] = None) -> np.ndarray:
        n = self.nbytes if nbytes is None else nbytes
        out = np.empty(n, dtype=np.uint8)
        _check(lib().acx_device_download(out.ctypes.data, self.ptr, n))
        return out

    def free(self) -> None:
        if self.ptr:
            lib().acx_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceResult:
    """Matches resident in HBM (acx_result_t)."""

    def __init__(self, handle: int, n_hay: int):
        self._h = handle
        self.n_hay = n_hay

    @property
    def count(self) -> int:
        return int(lib().acx_result_count(self._h))

    @property
    def device_ptr(self) -> int:
        return lib().acx_result_device_matches(self._h) or 0

    def matches(self) -> np.ndarray:
        out = np.empty(self.count, dtype=MATCH_DTYPE)
        if self.count:
            _check(lib().acx_result_copy(self._h, out.ctypes.data))
        return out

    def counts(self) -> np.ndarray:
        out = np.zeros(self.n_hay, dtype=np.uint64)
        if self.n_hay:
            _check(lib().acx_result_copy_counts(self._h, out.ctypes.data))
        return out

    def free(self) -> None:
        if self._h:
            lib().acx_free_result(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _take_matches(ptr: Optional[int], n: int) -> np.ndarray:
    """The library-owned host array of an acx_find* call as a numpy structured array, without a
    copy (a large result sits in pinned host memory): acx_free_matches runs when the array and
    every view of it are gone."""
    if not n:
        return np.empty(0, dtype=MATCH_DTYPE)
    buf = (ctypes.c_uint8 * (n * 24)).from_address(ptr)
    weakref.finalize(buf, lib().acx_free_matches, ptr)
    return np.frombuffer(buf, dtype=MATCH_DTYPE)


class Automaton:
    """acx_automaton_t: compiled patterns + device tables."""

    def __init__(self, patterns: Sequence[bytes], match_kind: int = MATCH_STANDARD,
                 implementation: int = IMPL_AUTO, kernel: Optional[int] = None):
        blob, off = pack(patterns)
        h = ctypes.c_void_p()
        _check(lib().acx_build(blob.ctypes.data, off.ctypes.data, len(patterns), match_kind,
                               implementation, ctypes.byref(h)))
        self._h = h.value
        if kernel is not None:
            self.set_kernel(kernel)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().acx_free_automaton(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def info(self) -> Info:
        i = Info()
        _check(lib().acx_automaton_info(self._h, ctypes.byref(i)))
        return i

    @property
    def max_pattern_len(self) -> int:
        return int(self.info.max_pattern_len)

    def set_kernel(self, kernel: int) -> None:
        _check(lib().acx_set_kernel(self._h, kernel))

    # ---- host-memory entry points
    def find(self, hay, overlapping: bool = False, codepoints: bool = False) -> np.ndarray:
        a = np.frombuffer(hay, dtype=np.uint8) if not isinstance(hay, np.ndarray) else hay
        a = np.ascontiguousarray(a)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        _check(lib().acx_find(self._h, a.ctypes.data if a.size else None, a.size,
                              int(overlapping), int(codepoints), ctypes.byref(out),
                              ctypes.byref(n)))
        return _take_matches(out.value, n.value)

    def find_tuples(self, hay, overlapping: bool = False,
                    codepoints: bool = False) -> List[Tuple[int, int, int]]:
        return [(int(p), int(s), int(e)) for (p, s, e) in self.find(hay, overlapping, codepoints)]

    def find_batch(self, haystacks: Sequence[bytes], overlapping: bool = False,
                   codepoints: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        blob, off = pack(haystacks)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        counts = np.zeros(len(haystacks), dtype=np.uint64)
        _check(lib().acx_find_batch(self._h, blob.ctypes.data, off.ctypes.data, len(haystacks),
                                    int(overlapping), int(codepoints), ctypes.byref(out),
                                    ctypes.byref(n), counts.ctypes.data))
        return _take_matches(out.value, n.value), counts

    def replicate(self, device: int) -> "Automaton":
        """the same automaton compiled again on another device (acx_replicate)"""
        h = ctypes.c_void_p()
        _check(lib().acx_replicate(self._h, device, ctypes.byref(h)))
        r = Automaton.__new__(Automaton)
        r._h = h.value
        return r

    def find_batch_multi(self, others: Sequence["Automaton"], haystacks: Sequence[bytes], overlapping: bool = False,
                         codepoints: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """acx_find_batch_multi over [self] + others: one host thread per handle, contiguous ranges of
        haystacks, identical to find_batch on one handle"""
        hs = [self] + list(others)
        arr = (ctypes.c_void_p * len(hs))(*[h._h for h in hs])
        blob, off = pack(haystacks)
        out, n = ctypes.c_void_p(), ctypes.c_uint64()
        counts = np.zeros(len(haystacks), dtype=np.uint64)
        _check(lib().acx_find_batch_multi(arr, len(hs), blob.ctypes.data, off.ctypes.data, len(haystacks),
                                          int(overlapping), int(codepoints), ctypes.byref(out), ctypes.byref(n),
                                          counts.ctypes.data))
        return _take_matches(out.value, n.value), counts

    # ---- device-resident entry point
    def find_device(self, d_ptr: int, nbytes: int, *, d_offsets: int = 0, n_hay: int = 0,
                    uniform_len: int = 0, overlapping: bool = False,
                    codepoints: bool = False) -> DeviceResult:
        out = ctypes.c_void_p()
        _check(lib().acx_find_device(self._h, d_ptr, nbytes, d_offsets or None, n_hay,
                                     uniform_len, int(overlapping), int(codepoints),
                                     ctypes.byref(out)))
        return DeviceResult(out.value, n_hay if (uniform_len or d_offsets) else 0)

    def generate(self, d_ptr: int, nbytes: int, kind: int, seed: int,
                 stream_offset: int = 0) -> None:
        _check(lib().acx_generate_haystack(self._h, d_ptr, nbytes, kind, seed, stream_offset))

    # ---- measurement
    def profile_enable(self, on=True) -> None:
        """True / 1: time every call's scan kernel; N > 1: every N-th call; False / 0: off."""
        _check(lib().acx_profile_enable(self._h, int(on)))

    PATH_STATS = ("sparse", "hot_calls", "hot_groups", "overflow_hits", "dense_tiles", "dense_radix", "overflow_regrown", "k0", "byte_ranges", "wide_redone", "resident_launches", "in_place")

    def path_stats(self, reset: bool = True) -> dict:
        """which way this handle's calls went (acx_path_stats): {sparse, hot_calls, hot_groups, overflow_hits,
        dense_tiles, dense_radix, overflow_regrown, k0, byte_ranges, wide_redone, resident_launches, in_place}"""
        out = (ctypes.c_uint64 * len(self.PATH_STATS))()
        _check(lib().acx_path_stats(self._h, out, int(reset)))
        return dict(zip(self.PATH_STATS, [int(v) for v in out]))

    def profile_read(self, reset: bool = True) -> Profile:
        p = Profile()
        _check(lib().acx_profile_read(self._h, ctypes.byref(p), int(reset)))
        return p
