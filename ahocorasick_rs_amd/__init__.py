"""MI355X-native Aho-Corasick matcher with the `ahocorasick_rs` Python API.

Drop-in for the reference package's façade
(/root/reference/pysrc/ahocorasick_rs/__init__.py:1-23): the same four names and the
deprecated MATCHKIND_* aliases, served by a C++ CPython extension
(`ahocorasick_rs_amd.ahocorasick_rs`, csrc/pymodule.cpp) over the C ABI of
`libacx_hip.so` (include/acx.h) and its hand-written HIP kernels.

There is NO CPU matching path: importing needs the built extension, and every
search needs a HIP device (RuntimeError otherwise).
"""
# Importing this package needs neither numpy nor torch (the reference's needs neither): the
# ctypes view of the C ABI (`ahocorasick_rs_amd.capi`, numpy-based, used by bench.py and the
# device-pointer workflows) is imported on first attribute access only.  One process holds ONE
# HIP runtime: torch wheels bundle a libamdhip64 with the same SONAME as /opt/rocm's, and
# whichever is loaded first serves both -- code that hands torch device pointers to this
# library should `import torch` first (bench.py and the tests do).
try:
    from .ahocorasick_rs import (  # noqa: E402
        AhoCorasick,
        BytesAhoCorasick,
        MatchKind,
        Implementation,
    )
except ImportError as e:  # pragma: no cover - build problem, fail loudly
    raise ImportError(
        "ahocorasick_rs_amd: the native extension is not built. Run "
        "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
        "There is no pure-Python or CPU fallback.") from e

# Backwards compatibility (reference __init__.py:10-12):
MATCHKIND_STANDARD = MatchKind.Standard
MATCHKIND_LEFTMOST_FIRST = MatchKind.LeftmostFirst
MATCHKIND_LEFTMOST_LONGEST = MatchKind.LeftmostLongest

__all__ = [
    "AhoCorasick",
    "BytesAhoCorasick",
    "MatchKind",
    "Implementation",
    # Deprecated:
    "MATCHKIND_STANDARD",
    "MATCHKIND_LEFTMOST_FIRST",
    "MATCHKIND_LEFTMOST_LONGEST",
]


def __getattr__(name):  # lazy submodules: capi (ctypes + numpy), distributed (torch.distributed)
    if name in ("capi", "distributed"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
