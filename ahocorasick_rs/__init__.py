"""`ahocorasick_rs` -- the reference's package name, served by the MI355X-native build.

The reference package is a façade over its native submodule
(/root/reference/pysrc/ahocorasick_rs/__init__.py:2-7; module `ahocorasick_rs.ahocorasick_rs`
declared at /root/reference/src/lib.rs:438-445).  This package has the same shape:
`ahocorasick_rs.ahocorasick_rs` resolves to the C++ CPython extension of
`ahocorasick_rs_amd` (HIP kernels behind include/acx.h), so existing callers --
`import ahocorasick_rs; ahocorasick_rs.AhoCorasick(...)` -- run unchanged.

`__acx_amd__` marks this build: bench.py's probe for a genuine Rust wheel (BASELINE.md §2
step 1) skips any module that carries it, and searches with this repository removed from
sys.path, so a real wheel installed in site-packages is never shadowed during that probe.
"""
from .ahocorasick_rs import (
    AhoCorasick,
    BytesAhoCorasick,
    MatchKind,
    Implementation,
)

__acx_amd__ = True

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
