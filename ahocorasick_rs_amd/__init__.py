"""MI355X-native Aho-Corasick matcher with the `ahocorasick_rs` Python API.

Drop-in for the reference package's façade
(/root/reference/pysrc/ahocorasick_rs/__init__.py:1-23): the same four names and the
deprecated MATCHKIND_* aliases, served by a C++ CPython extension
(`ahocorasick_rs_amd.ahocorasick_rs`, csrc/pymodule.cpp) over the C ABI of
`libacx_hip.so` (include/acx.h) and its hand-written HIP kernels.

There is NO CPU matching path: importing needs the built extension, and every
search needs a HIP device (RuntimeError otherwise).
"""
from . import capi as _capi

# one HIP runtime per process: if torch is present let it load its bundled runtime first
_capi._preload_hip_runtime()

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
