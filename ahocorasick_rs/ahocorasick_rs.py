"""Stand-in for the reference's native submodule `ahocorasick_rs.ahocorasick_rs`
(/root/reference/src/lib.rs:438-445): the very same four classes, re-exported from the C++
CPython extension `ahocorasick_rs_amd.ahocorasick_rs` (one extension, one set of type objects,
so `ahocorasick_rs.MatchKind.Standard is ahocorasick_rs_amd.MatchKind.Standard`)."""
from ahocorasick_rs_amd.ahocorasick_rs import (  # noqa: F401
    AhoCorasick,
    BytesAhoCorasick,
    Implementation,
    MatchKind,
)

__all__ = ["AhoCorasick", "BytesAhoCorasick", "Implementation", "MatchKind"]
